#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_facade.py -q -m gpu -x -k "seed or Seed" 2>&1 | tail -4
SWEEP_GIB=8 SWEEP_SHAPES="250,80,2,2;300,128,1,1;250,100,1,3;250,65,2,3;150,64,2,3;250,31,2,3;150,48,2,3" timeout 900 python tools/seed_sweep.py 2>&1 | tail -8
timeout 600 python tools/stress_seeds.py 300 555 2>&1 | tail -1
