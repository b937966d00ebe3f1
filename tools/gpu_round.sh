#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
echo "=== 100,64,3"; ABLATE_SHAPE=100,64,3 python tools/ab_multi.py "h32,h32:NTHIP_TUNE_RUN_LEN=10,:NTHIP_TUNE_RUN_LEN=10" 40000000 8 | cut -c1-130
echo "=== 100,64,1"; ABLATE_SHAPE=100,64,1 python tools/ab_multi.py "h32" 40000000 8 | cut -c1-130
echo "=== 150,51,1"; ABLATE_SHAPE=150,51,1 python tools/ab_multi.py "h32,h48" 40000000 8 | cut -c1-130
echo "=== 150,40,1"; ABLATE_SHAPE=150,40,1 python tools/ab_multi.py "h32" 40000000 8 | cut -c1-130
