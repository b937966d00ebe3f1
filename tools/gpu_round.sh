#!/bin/bash
# scratch script for one gpurun call
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15
python bench.py --steps 20 --warmup 5 > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err
tail -c 3000 gpurun_out/r02/bench_default.json
