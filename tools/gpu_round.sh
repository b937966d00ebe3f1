#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fixed or run_split or fuzz or golden or na_runs or dirty or padded or bloom or minhash" 2>&1 | tail -3
for sh in "101,31,1" "149,31,1" "150,51,1" "100,64,1"; do echo "=== $sh"; ABLATE_SHAPE=$sh python tools/ab_multi.py cochk 60000000 8 | cut -c1-125; done
