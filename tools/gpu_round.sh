#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
for i in 1 2 3; do python tools/process_variance.py 100000000 4; done
