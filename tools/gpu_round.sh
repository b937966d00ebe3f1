#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "wave_tile_kernel_shapes" 2>&1 | tail -3
