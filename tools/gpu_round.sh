#!/bin/bash
cd "$GRAFT_REPO_ROOT"
ABLATE_SHAPE=100,64,3 timeout 600 python tools/ab_multi.py old 30000000 10 2>&1 | tail -2
ABLATE_SHAPE=151,31,2 timeout 600 python tools/ab_multi.py old 20000000 10 2>&1 | tail -2
ABLATE_SHAPE=101,31,4 timeout 600 python tools/ab_multi.py old 20000000 10 2>&1 | tail -2
ABLATE_SHAPE=151,31,5 timeout 600 python tools/ab_multi.py old 20000000 10 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "any_m or any_k or multi or general or shapes" 2>&1 | tail -3
