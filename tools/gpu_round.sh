cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02
NTHASH_AMD_LIB=$PWD/nthash_amd/lib/ab/libnthash_hip_wind.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not native_library" 2>&1 | tail -3
timeout 600 python tools/ab_multi.py win,wind,wind:NTHIP_TUNE_NO_PACING=1,wind:NTHIP_TUNE_PH_PERIOD=4300,wind:NTHIP_TUNE_PH_PERIOD=4700,windnh,winnh,winddbg,winddbg:NTHIP_TUNE_PH_PERIOD=4700 100000000 5 > gpurun_out/r02/abl_win2.txt 2>&1; grep -v "^\[kmer" gpurun_out/r02/abl_win2.txt; grep "^\[kmer" gpurun_out/r02/abl_win2.txt | sort -k 30 | tail -4
