timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_query.py -x -q -k "bloom or count or sketch or consumers" 2>&1 | tail -4
timeout 200 bash tools/kstats.sh gpurun_out/ab_pieces python tools/bloom_one.py 20000000 0 3 2>&1 | grep -v "rocclr\|synth" | head -5
timeout 300 bash tools/kstats.sh gpurun_out/abq_pieces python tools/query_bench.py 20000000 1 35 0 2>&1 | grep -v "rocclr\|synth" | head -9
export NTHASH_AMD_LIB=$PWD/nthash_amd/lib/ab/libnthash_hip_tim.so
echo "== pieces"; timeout 200 python tools/bloom_one.py 20000000 0 1 2>&1 | grep "tile" | sed -n 10,14p
