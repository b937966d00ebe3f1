#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "seed" 2>&1 | tail -3
for rot in 0 1; do echo "NO_SEED_ROT=$rot"; NTHIP_TUNE_NO_SEED_ROT=$rot SWEEP_GIB=16 SWEEP_SHAPES="250,31,3,1;250,31,3,3;250,31,4,1;250,31,4,2;250,31,4,4;150,24,3,2;250,31,2,3" timeout 900 python tools/seed_sweep.py 2>&1; done
