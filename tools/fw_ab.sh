#!/bin/bash
# First-window forms (first_window.hpp) on the GPU: parity subset, then in-process A/B of grouped / scan / cost model on
# k > 64 shapes, then the k <= 64 shapes with the k-independent forms (NTHIP_TUNE_TABLE_K_MAX, one process per variant).
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-fw}
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "first_window or any_k or extend or whole_read or bloom_long or minhash" 2>&1 | tail -8 > $OUT/pytest.log
cat $OUT/pytest.log
for shape in 150,65,1:12000000 150,100,1:20000000 250,200,1:20000000 1000,500,1:2000000 150,80,2:8000000 10000,200,1:120000 300,128,1:6000000; do
  s=${shape%%:*}; n=${shape##*:}
  echo "== shape $s reads $n" | tee -a $OUT/ab.txt
  ABLATE_SHAPE=$s timeout 600 python tools/ab_multi.py ":NTHIP_TUNE_FW=1,:NTHIP_TUNE_FW=2" $n 8 2>&1 | tee -a $OUT/ab.txt
done
for v in "default" "NTHIP_TUNE_TABLE_K_MAX=32 NTHIP_TUNE_FW=1" "NTHIP_TUNE_TABLE_K_MAX=32 NTHIP_TUNE_FW=2"; do
  echo "== variant: $v" | tee -a $OUT/k64.txt
  if [ "$v" = "default" ]; then v=""; fi
  env $v SWEEP_GIB=8 SWEEP_SHAPES="100,64,3;100,64,1;150,64,1;150,51,1;150,40,1" timeout 600 python tools/shape_sweep.py 2>&1 | tee -a $OUT/k64.txt
done
