#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
timeout 2400 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "windowed or any_k_inst" 2>&1 | tail -5
