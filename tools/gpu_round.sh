#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests -q -m gpu -x -k "seed" 2>&1 | tail -12
timeout 300 python tools/stress_seeds.py 400 21 2>&1 | tail -3
timeout 300 python tools/ragged_seed_bench.py 2>&1 | tail -3
