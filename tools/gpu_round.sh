#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "seed" 2>&1 | tail -3
SWEEP_GIB=16 timeout 900 python tools/seed_sweep.py gpurun_out/seed_sweep.json 2>&1 | tee gpurun_out/seed_sweep.txt
for sh in 31,2,3 31,3,3 31,4,2 31,6,1 48,3,2 64,2,3 64,3,1; do echo -n "== $sh "; RSB_SHAPE=$sh timeout 300 python tools/ragged_seed_bench.py 2000000 2>&1 | tail -1; done | tee gpurun_out/ragged_seed_shapes.txt
