#!/bin/bash
# A/B of the binned Bloom insert's partition kernels over variant libraries (tools/ab_build.sh bb_<tag> ...)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/bbab
for tag in base "$@"; do
  lib=nthash_amd/lib/ab/libnthash_hip_bb_$tag.so
  if [ "$tag" = base ]; then unset NTHASH_AMD_LIB; else export NTHASH_AMD_LIB=$PWD/$lib; fi
  echo "== $tag"
  bash tools/bbp.sh bbab/$tag 2>&1 | grep -E "^insert 3|bloom_part|bloom_apply|bloom_hist"
  if [ -z "$NO_TEST" ]; then timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bloom_binned" 2>&1 | tail -1; fi
done
