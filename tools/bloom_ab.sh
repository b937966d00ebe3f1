#!/bin/bash
# A/B of the binned Bloom insert's builds (tools/ab_build.sh tags): 20 M x 150 bp into a fresh filter of 2^$BITS bits
cd "$(dirname "$0")/.."
for tag in "$@"; do
  lib=nthash_amd/lib/ab/libnthash_hip_$tag.so
  [ "$tag" = base ] && lib=nthash_amd/lib/libnthash_hip.so
  echo -n "$tag: "; NTHASH_AMD_LIB=$lib timeout 300 python tools/bloom_binned_prof.py 20000000 1 ${BITS:-35} 2>&1 | tail -1
done
