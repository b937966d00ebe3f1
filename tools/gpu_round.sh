#!/bin/bash
# scratch script for one gpurun call
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
