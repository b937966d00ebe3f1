#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "seed" 2>&1 | tail -2
for sh in "151,31,3" "250,31,3" "101,31,3" "150,31,3"; do echo "=== $sh"; ABLATE_SEEDS=1 ABLATE_SHAPE=$sh python tools/ab_multi.py ":NTHIP_TUNE_NO_SEED_ALIGN=1" 12000000 8 | cut -c1-125; done
