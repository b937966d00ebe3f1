#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for mb in -1 16 64 160; do echo "SEED_CHUNK_MB=$mb"; NTHIP_TUNE_SEED_CHUNK_MB=$mb SWEEP_GIB=8 SWEEP_SHAPES="250,31,6,1;100,64,3,1;150,48,3,2;250,31,5,4" timeout 900 python tools/seed_sweep.py 2>&1; done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "seed_passes" 2>&1 | tail -3
