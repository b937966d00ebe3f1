#!/bin/bash
# scratch script for one gpurun call
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "ragged or offsets or long or fuzz" 2>&1 | tail -3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/reads_trace -o kt -- python tools/ragged_bench.py 10000000 > /dev/null 2>&1
f=$(find gpurun_out/r02/reads_trace -name "*kernel_stats.csv" | head -1)
cat "$f" | cut -c1-110 | grep -v "general\|copyBuffer\|synth" | head -6
find gpurun_out/r02/reads_trace -name "*.db" -delete; find gpurun_out/r02/reads_trace -name "*kernel_trace.csv" -delete
for kv in "A=1" "NTHIP_TUNE_READS_PER_TILE=64" "NTHIP_TUNE_READS_RUN_LEN=10" "NTHIP_TUNE_READS_RUN_LEN=10;NTHIP_TUNE_READS_PER_TILE=64" "RAGGED_MOSTLY=150"; do
  echo "$kv: $(env ${kv//;/ } python tools/ragged_bench.py 10000000 2>&1 | grep 'ragged run')"
done
