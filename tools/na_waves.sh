#!/bin/bash
# waves per CU of the N-aware hash pass (NTHIP_TUNE_NA_WAVES) on dirty fixed-length batches (tools/dirty_bench.py)
cd "$GRAFT_REPO_ROOT"
for shape in "150,31,1" "151,31,1" "250,31,4" "100,64,3" "100,21,2" "250,48,1"; do
  for w in 8 12 16; do
    export NTHIP_TUNE_NA_WAVES=$w DIRTY_SHAPE=$shape
    n=$((2400000000 / (${shape%%,*} * 1)))
    echo "shape $shape NA_WAVES=$w: $(python tools/dirty_bench.py 8000000 2>&1 | grep optimistic | sed 's/.*skipped)//')"
  done
done
