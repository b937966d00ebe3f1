#!/bin/bash
# Build a second copy of the C-ABI library with extra compiler flags, for in-process A/B timing:
#   tools/ab_build.sh b -DKRG_UNIFORM_WAVE=0   ->  nthash_amd/lib/ab/libnthash_hip_b.so
#   NTHASH_AMD_LIB=nthash_amd/lib/ab/libnthash_hip_b.so python tools/ablate.py ...
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p nthash_amd/lib/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-pass-failed -Iinclude "$@" \
  nthash_amd/csrc/nthip_capi.hip -o nthash_amd/lib/ab/libnthash_hip_$tag.so
echo nthash_amd/lib/ab/libnthash_hip_$tag.so
