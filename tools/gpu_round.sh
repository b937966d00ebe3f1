#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench.py -q -m gpu -x -k "placement or bench" 2>&1 | tail -3
for i in 1 2 3; do python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-peak 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e9,1), round(d['roofline']['frac'],4), [(b['fill_GBps'], b['candidates_measured']) for b in d['config']['placement']['buffers']])"; done
