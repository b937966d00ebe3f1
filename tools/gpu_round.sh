#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/vp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench.py -q -m gpu -x -k "whole_read or ragged or fastx or fastq or spans or var or seed" 2>&1 | tail -3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/vp -o kt -- python bench.py --config var --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-peak > gpurun_out/vp/out.json 2> gpurun_out/vp/err.txt
python -c "
import json; d=json.loads(open('gpurun_out/vp/out.json').read().strip().splitlines()[-1]); print(d['value']/1e9, d['ms_per_step'], d['roofline']['kernel_avg_ms'], d.get('verify',{}).get('ok'))"
f=$(find gpurun_out/vp -name "*kernel_stats.csv" | head -1); grep -i "dirty\|kmer_reads\|prep" "$f" | cut -c1-140
find gpurun_out/vp -name "*.db" -delete; find gpurun_out/vp -name "*kernel_trace.csv" -delete
timeout 300 python tools/ragged_bench.py 2>&1 | tail -3
timeout 300 python tools/ragged_seed_bench.py 2>&1 | tail -2
