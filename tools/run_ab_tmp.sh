timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_query.py -x -q -k "bloom or count or sketch or consumers or seed_bloom or query" 2>&1 | tail -4
timeout 300 python tools/query_bench.py 20000000 1,3 35 0 2>&1 | grep -v rocprof
NTHIP_TUNE_BLOOM_QUERY_PASSES=1 timeout 300 python tools/query_bench.py 20000000 3 35 0 2>&1 | grep "binned"
NTHIP_TUNE_BLOOM_QUERY_PASSES=2 timeout 300 python tools/query_bench.py 20000000 3 35 0 2>&1 | grep "binned"
timeout 200 python bench.py --consumers-reads 20000000 2>/dev/null | python tools/show_consumers.py /dev/stdin 2>/dev/null | head -20
