#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
echo "=== 150,31,4 special vs gen"; ABLATE_SHAPE=150,31,4 python tools/ab_multi.py ":NTHIP_TUNE_NO_SPECIAL=1,:NTHIP_TUNE_NO_M4=1" 20000000 8 | cut -c1-130
echo "=== 150,31,3"; ABLATE_SHAPE=150,31,3 python tools/ab_multi.py ":NTHIP_TUNE_NO_SPECIAL=1" 20000000 8 | cut -c1-130
echo "=== 100,64,3 run lengths"; ABLATE_SHAPE=100,64,3 python tools/ab_multi.py ":NTHIP_TUNE_RUN_LEN=13,:NTHIP_TUNE_RUN_LEN=19,:NTHIP_TUNE_RUN_LEN=10,:NTHIP_TUNE_WAVES=8,:NTHIP_TUNE_WAVES=12" 40000000 8 | cut -c1-130
