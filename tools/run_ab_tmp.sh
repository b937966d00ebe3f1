timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_query.py -x -q -k "bloom or count or sketch or consumers" 2>&1 | tail -4
timeout 300 bash tools/kstats.sh gpurun_out/abq_pieces python tools/query_bench.py 20000000 1,3 35 30 2>&1 | grep -v "rocclr\|synth" | head -24
