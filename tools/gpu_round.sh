#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -5
timeout 200 python tools/stress_seeds.py 300 123 2>&1 | tail -1
