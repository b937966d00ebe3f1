#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
for kv in "A=1" "NTHIP_TUNE_READS_RUN_LEN=7" "NTHIP_TUNE_READS_RUN_LEN=9" "NTHIP_TUNE_READS_RUN_LEN=11" "NTHIP_TUNE_READS_RUN_LEN=13" "NTHIP_TUNE_READS_RUN_LEN=15" "NTHIP_TUNE_READS_RUN_LEN=9;NTHIP_TUNE_READS_PER_TILE=64" "NTHIP_TUNE_READS_RUN_LEN=11;NTHIP_TUNE_READS_PER_TILE=64" "NTHIP_TUNE_READS_RUN_LEN=9;NTHIP_TUNE_READS_PER_TILE=16"; do
  echo "$kv: $(env ${kv//;/ } python tools/ragged_bench.py 10000000 2>&1 | grep 'ragged run') // mostly150: $(env ${kv//;/ } RAGGED_MOSTLY=150 python tools/ragged_bench.py 10000000 2>&1 | grep 'ragged run' | cut -c60-)"
done
