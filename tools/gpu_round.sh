#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
for sh in "151,31,1" "100,31,1" "76,31,1" "250,31,1" "125,31,1" "150,21,1" "150,25,1"; do
  echo "=== $sh"; ABLATE_SHAPE=$sh python tools/ab_multi.py ":NTHIP_TUNE_NO_ANY_K_RUNS=1" 40000000 10 | cut -c1-120
done
echo "=== 150,25,1 run lengths"; ABLATE_SHAPE=150,25,1 python tools/ab_multi.py ":NTHIP_TUNE_RUN_LEN=14,:NTHIP_TUNE_RUN_LEN=9,:NTHIP_TUNE_RUN_LEN=21" 40000000 10 | cut -c1-120
echo "=== 100,31,1 run lengths"; ABLATE_SHAPE=100,31,1 python tools/ab_multi.py ":NTHIP_TUNE_RUN_LEN=14,:NTHIP_TUNE_RUN_LEN=10" 40000000 10 | cut -c1-120
