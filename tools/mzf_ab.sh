#!/bin/bash
# A/B of the one-pass minimizer kernel's builds (tools/ab_build.sh tags) on the GPU box: whole call, 20 M x 150 bp
cd "$(dirname "$0")/.."
n=${N:-20000000}
for tag in "$@"; do
  lib=nthash_amd/lib/ab/libnthash_hip_$tag.so
  [ "$tag" = base ] && lib=nthash_amd/lib/libnthash_hip.so
  for w in ${WS:-10}; do
    echo -n "$tag: "; NTHASH_AMD_LIB=$lib timeout 300 python tools/minimizer_bench.py $n $w clean 2>&1 | tail -1
  done
done
