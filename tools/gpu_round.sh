#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "whole_read or ragged" 2>&1 | tail -2
RAGGED_M=3 python tools/ab_ragged.py rdp 8000000 10
RAGGED_M=2 python tools/ab_ragged.py rdp 8000000 10
