#!/bin/bash
# A/B of the XCD-aware tile groups: dirty fixed-length batch (N-aware pass) and shapes whose tiles start anywhere
cd "$(dirname "$0")/.."
for tag in "$@"; do
  lib=nthash_amd/lib/ab/libnthash_hip_$tag.so
  [ "$tag" = base ] && lib=nthash_amd/lib/libnthash_hip.so
  echo "== $tag"
  NTHASH_AMD_LIB=$lib python tools/dirty_bench.py 20000000 2>&1 | head -2
  NTHASH_AMD_LIB=$lib SWEEP_GIB=16 SWEEP_SHAPES="151,31,1;101,31,1;250,31,1;150,31,1;100,64,3;150,31,4" python tools/shape_sweep.py 2>&1 | grep "L="
done
