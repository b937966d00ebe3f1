#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "whole_read" 2>&1 | tail -8
