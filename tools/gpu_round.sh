#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "seed or fastx or fasta or fastq or spans" 2>&1 | tail -5
timeout 300 python tools/fastq_bench.py 2>&1 | tail -4
