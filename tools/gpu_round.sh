#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -q -m gpu -x -k "bad_offsets or spans or fastx or fastq or whole_read or seed_whole" 2>&1 | tail -3
python bench.py --config var --steps 10 --warmup 3 --no-cpu-baseline --no-peak 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e9,1), d['ms_per_step'], d['verify']['ok'])"
