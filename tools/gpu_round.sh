#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "whole_read or ragged or seed_passes or fastx or spans" 2>&1 | tail -3
timeout 600 python tools/ab_ragged.py mk0,mk3 10000000 12 2>&1 | tail -6
RAGGED_MOSTLY=150 timeout 600 python tools/ab_ragged.py mk0,mk3 10000000 12 2>&1 | tail -4
