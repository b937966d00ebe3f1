#!/bin/bash
# scratch script for one gpurun call
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -5
timeout 300 python tools/soak.py 200 99 2>&1 | tail -2
timeout 200 python tools/stress_seeds.py 300 77 2>&1 | tail -1
timeout 200 python tools/stress_shapes.py 2>&1 | tail -1
