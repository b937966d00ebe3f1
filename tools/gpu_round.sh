#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3 4; do python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e9,1), round(d['roofline']['frac'],4), round(d['roofline']['peak_measured']), [(b['fill_GBps'],b['candidates_measured']) for b in d['config']['placement']['buffers']])"; done
