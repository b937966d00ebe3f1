#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fixed_vs_oracle or golden or any_k_any_m or properties or fuzz" 2>&1 | tail -2
echo "=== 150,31,2"; ABLATE_SHAPE=150,31,2 python tools/ab_multi.py ":NTHIP_TUNE_NO_M4=1" 50000000 8 | cut -c1-125
echo "=== 150,31,3"; ABLATE_SHAPE=150,31,3 python tools/ab_multi.py ":NTHIP_TUNE_NO_M4=1" 50000000 8 | cut -c1-125
