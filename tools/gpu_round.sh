#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fixed or run_split or fuzz or golden or na_runs or dirty or padded or any_k or long" 2>&1 | tail -2
