#!/bin/bash
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-burst}
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "kmer and not windowed" 2>&1 | tail -5 | tee $OUT/pytest.log
AB_PROBED=1 python tools/ab_multi.py ":NTHIP_TUNE_NO_PHASES=1,:NTHIP_TUNE_WAVES=8,:NTHIP_TUNE_WAVES=10,:NTHIP_TUNE_TILE_MAP=256,:NTHIP_TUNE_TILE_MAP=8" 60000000 10 2>&1 | tee $OUT/ab_c2.txt
ABLATE_SHAPE=150,31,4 python tools/ab_multi.py ":NTHIP_TUNE_NO_PHASES=1" 15000000 8 2>&1 | tee $OUT/ab_c3.txt
