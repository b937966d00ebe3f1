#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python tools/stress_seeds.py 1500 11 2>&1 | tail -3
NTHIP_TUNE_SEED_PASS=1 timeout 900 python tools/stress_seeds.py 500 12 2>&1 | tail -2
