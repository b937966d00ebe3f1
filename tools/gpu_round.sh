#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "seed" 2>&1 | tail -4
SWEEP_GIB=16 timeout 900 python tools/seed_sweep.py gpurun_out/seed_sweep.json 2>&1 | tee gpurun_out/seed_sweep.txt
