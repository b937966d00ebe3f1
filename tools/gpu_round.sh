#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fixed or run_split or fuzz or golden or any_k" 2>&1 | tail -2
for sh in "151,31,2" "151,31,4" "250,31,3" "100,25,2"; do echo "=== $sh"; ABLATE_SHAPE=$sh python tools/ab_multi.py ":NTHIP_TUNE_NO_ANY_K_RUNS=1" 30000000 8 | cut -c1-125; done
