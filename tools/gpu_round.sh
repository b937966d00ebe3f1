#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python tools/seed_long_wholecall.py 2>&1 | grep "LONG=0"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_facade.py -q -m gpu -x -k "seed or Seed" 2>&1 | tail -3
