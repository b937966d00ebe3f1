#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python tools/ragged_seed_bench.py 2>&1 | tail -2
timeout 200 python tools/stress_seeds.py 400 555 2>&1 | tail -1
