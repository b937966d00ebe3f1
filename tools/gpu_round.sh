#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "seed_long" 2>&1 | tail -8
