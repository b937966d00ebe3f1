#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
SWEEP_GIB=4 SWEEP_PROBED=0 SWEEP_SHAPES="250,80,2,2;300,128,1,1;250,100,1,3;300,160,1,1;250,31,2,12;250,31,1,9;10000,31,2,3;100000,31,2,3;5000000,31,2,3;31,31,2,3;40,31,2,3;3000,48,3,2;1000,31,4,2" timeout 1200 python tools/seed_sweep.py 2>&1 | tee gpurun_out/seed_sweep_edges.txt
