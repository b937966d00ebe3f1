set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 600 python tools/ab_multi.py :NTHIP_TUNE_NO_SPECIAL=1\;NTHIP_TUNE_WAVES=12,gc15:NTHIP_TUNE_NO_SPECIAL=1\;NTHIP_TUNE_WAVES=12,gc15:NTHIP_TUNE_NO_SPECIAL=1\;NTHIP_TUNE_WAVES=16 60000000 5 > gpurun_out/r02/abl_gen3.txt 2>&1; cat gpurun_out/r02/abl_gen3.txt
ABLATE_SHAPE=151,31,1 timeout 600 python tools/ab_multi.py :NTHIP_TUNE_WAVES=12,gc15:NTHIP_TUNE_WAVES=12,gc15:NTHIP_TUNE_WAVES=16 60000000 5 > gpurun_out/r02/abl_gen4.txt 2>&1; cat gpurun_out/r02/abl_gen4.txt
