#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_facade.py -q -m gpu -x -k "seed or Seed" 2>&1 | tail -3
timeout 900 python tools/stress_seeds.py 400 21 2>&1 | tail -1
for r in 1000 50000 1,10,100,1000,50000; do echo "runs of $r"; LSB_RUNS=$r timeout 600 python tools/long_seed_bench.py 50 1 2>&1 | head -2; done
