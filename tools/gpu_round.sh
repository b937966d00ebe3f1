#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_facade.py -q -m gpu -x -k "seed or Seed" 2>&1 | tail -3
for rot in 0 1; do echo "NO_SEED_ROT=$rot"; for sh in 31,3,1 31,4,1 31,4,2 31,3,3; do echo -n "$sh: "; NTHIP_TUNE_NO_SEED_ROT=$rot RSB_SHAPE=$sh timeout 300 python tools/ragged_seed_bench.py 2000000 2>&1 | tail -1; done; done
timeout 1500 python tools/stress_seeds.py 800 5 2>&1 | grep -v "^ok" | tail -3
