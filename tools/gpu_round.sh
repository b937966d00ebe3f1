#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests -q -m gpu -x -k "whole_read or ragged or fastx or fastq or batch_nthash" 2>&1 | tail -3
RAGGED_MOSTLY=150 python tools/ab_ragged.py :NTHIP_TUNE_READS_RUN_LEN=9,:NTHIP_TUNE_READS_RUN_LEN=15 10000000 12
RAGGED_MOSTLY=151 python tools/ab_ragged.py :NTHIP_TUNE_READS_RUN_LEN=9,:NTHIP_TUNE_READS_RUN_LEN=15,:NTHIP_TUNE_READS_RUN_LEN=11 10000000 12
python tools/ab_ragged.py :NTHIP_TUNE_READS_RUN_LEN=9 10000000 10
