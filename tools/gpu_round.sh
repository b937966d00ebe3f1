#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_facade.py tests/test_gpu_bench.py -q -m gpu -x -k "seed or Seed or bench" 2>&1 | tail -3
SWEEP_GIB=8 SWEEP_SHAPES="250,31,6,1;250,31,2,3" timeout 900 python tools/seed_sweep.py 2>&1 | tail -2
