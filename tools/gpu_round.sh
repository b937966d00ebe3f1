set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
python tools/fill_probe.py fill30,fill8wt,fill8nt,fill30wt 16 > gpurun_out/r02/fill_probe.txt 2>&1; cat gpurun_out/r02/fill_probe.txt
python tools/ab_multi.py nohash,nh_nl,nh_stdef,nh_stnt,nh_stsc1,nh_ldplain,nohash:NTHIP_TUNE_WAVES=16,nohash:NTHIP_TUNE_WAVES=4,nohash:NTHIP_TUNE_RUN_LEN=30,nohash:NTHIP_TUNE_RUN_LEN=30\;NTHIP_TUNE_WAVES=4 100000000 8 > gpurun_out/r02/abl_nohash.txt 2>&1; cat gpurun_out/r02/abl_nohash.txt
python tools/ab_multi.py uw,uw:NTHIP_TUNE_WAVES=16,uw:NTHIP_TUNE_WAVES=12,uw:NTHIP_TUNE_RUN_LEN=30,:NTHIP_TUNE_WAVES=16 100000000 8 > gpurun_out/r02/abl_uw.txt 2>&1; cat gpurun_out/r02/abl_uw.txt
