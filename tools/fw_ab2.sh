#!/bin/bash
# second pass: full GPU parity, then m > 1 copy-out (staged vs per-value), cost model check, longer runs for k > 64
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-fw2}
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -6 > $OUT/pytest.log
cat $OUT/pytest.log
for shape in 100,64,3:30000000 150,31,4:10000000 151,25,2:12000000 150,31,8:5000000 150,80,2:8000000 101,31,3:12000000; do
  s=${shape%%:*}; n=${shape##*:}
  echo "== shape $s reads $n  (base = staged; NO_STAGE = one value per lane half; NO_SPECIAL)" | tee -a $OUT/ab.txt
  NTHIP_TUNE_NO_SPECIAL=1 ABLATE_SHAPE=$s timeout 600 python tools/ab_multi.py ":NTHIP_TUNE_NO_STAGE=1;NTHIP_TUNE_NO_SPECIAL=1" $n 8 2>&1 | tee -a $OUT/ab.txt
done
echo "== 150,31,4 special kernel vs general (staged)" | tee -a $OUT/ab.txt
ABLATE_SHAPE=150,31,4 timeout 600 python tools/ab_multi.py ":NTHIP_TUNE_NO_SPECIAL=1" 10000000 8 2>&1 | tee -a $OUT/ab.txt
for shape in 150,65,1:12000000 150,100,1:20000000 250,200,1:20000000 1000,500,1:2000000 300,128,1:6000000; do
  s=${shape%%:*}; n=${shape##*:}
  echo "== shape $s reads $n  (base = model; run lengths)" | tee -a $OUT/ab.txt
  ABLATE_SHAPE=$s timeout 600 python tools/ab_multi.py ":NTHIP_TUNE_RUN_LEN=15,:NTHIP_TUNE_RUN_LEN=19,:NTHIP_TUNE_RUN_LEN=23,:NTHIP_TUNE_RUN_LEN=27,:NTHIP_TUNE_RUN_LEN=31" $n 6 2>&1 | tee -a $OUT/ab.txt
done
