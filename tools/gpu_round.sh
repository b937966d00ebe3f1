#!/bin/bash
cd "$GRAFT_REPO_ROOT"
NTHASH_AMD_LIB=$PWD/nthash_amd/lib/ab/libnthash_hip_dst.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "seed" 2>&1 | tail -3
ABLATE_SEEDS=1 ABLATE_SHAPE=250,31,3 AB_PROBED=1 python tools/ab_multi.py "dst,dstnt" 16000000 8 | cut -c1-125
