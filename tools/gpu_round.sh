#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests/test_gpu_bench.py -q -m gpu -x 2>&1 | tail -8
