#!/bin/bash
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-fw3}
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -4 > $OUT/pytest.log
cat $OUT/pytest.log
PMC_SQ_ONLY=1 bash tools/run_pmc.sh $OUT/pmc_ref shape:100,64,3 20000000 > $OUT/pmc_ref.log 2>&1
grep -v "^copy" $OUT/pmc_ref/summary.txt
PMC_SQ_ONLY=1 bash tools/run_pmc.sh $OUT/pmc_k100 shape:150,100,1 20000000 > $OUT/pmc_k100.log 2>&1
grep -v "^copy" $OUT/pmc_k100/summary.txt
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
SWEEP_GIB=8 SWEEP_SHAPES="150,65,1;150,100,1;250,200,1;1000,500,1;150,80,2;300,128,1;10000,200,1;400,255,1;100,64,3;100,64,1;150,64,1" python tools/shape_sweep.py | tee $OUT/sweep.txt
