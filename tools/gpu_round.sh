#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -q -m gpu -x -k "seed or fastx or fastq" 2>&1 | tail -3
timeout 200 python tools/stress_seeds.py 300 8 2>&1 | tail -1
