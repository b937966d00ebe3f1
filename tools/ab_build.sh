#!/bin/bash
# Build a second copy of the C-ABI library with extra compiler flags, for in-process A/B timing:
#   tools/ab_build.sh b -DKRG_UNIFORM_WAVE=0            ->  nthash_amd/lib/ab/libnthash_hip_b.so
#   UNITS=capi_kmer_runs tools/ab_build.sh b -DX=1      only that unit gets the flags (fast)
#   NTHASH_AMD_LIB=nthash_amd/lib/ab/libnthash_hip_b.so python tools/ablate.py ...
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
if [ -n "$UNITS" ]; then
  python -m nthash_amd.build --tag "$tag" --flags "$*" --units "$UNITS" | tail -1
else
  python -m nthash_amd.build --tag "$tag" --flags "$*" | tail -1
fi
