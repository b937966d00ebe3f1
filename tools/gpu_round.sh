#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
timeout 600 python -m pytest tests -q -m gpu -x -k "seed_whole or seed_dirty" 2>&1 | tail -3
for kv in "A=1" "NTHIP_TUNE_SEED_RPT=8" "NTHIP_TUNE_SEED_RPT=12" "NTHIP_TUNE_SEED_RPT=24" "NTHIP_TUNE_SEED_RPT=32"; do
  echo "$kv: $(env $kv python tools/ragged_seed_bench.py 2>&1 | grep seed_wave_kernel)"
done
