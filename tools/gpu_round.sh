#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "ragged or whole_read or offsets or fuzz or fastx or multi" 2>&1 | tail -4
python tools/ab_ragged.py pf0,pf3,:NTHIP_TUNE_READS_RUN_LEN=11,:NTHIP_TUNE_READS_PER_TILE=24 10000000 14
RAGGED_MOSTLY=150 python tools/ab_ragged.py pf0,pf3,:NTHIP_TUNE_READS_RUN_LEN=15 10000000 14
