#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "seed" 2>&1 | tail -3
timeout 2000 python tools/stress_seeds.py 1000 77 2>&1 | grep -v "^ok" | tail -5
