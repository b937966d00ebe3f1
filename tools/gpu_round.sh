#!/bin/bash
# scratch script for one gpurun call
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "seed" 2>&1 | tail -5
ABLATE_SEEDS=1 ABLATE_SHAPE=250,31,3 python tools/ab_multi.py nohash,nostore,:NTHIP_TUNE_NO_SEED_ROT=1 8000000 10
