"""Build the nthash_amd native libraries in-tree with hipcc (gfx950 only).

    python -m nthash_amd.build            # builds nthash_amd/lib/*.so

libnthash_hip.so  the C-ABI (include/nthash_hip.h): HIP kernels + launch logic
libnthash.so      the C++ host facade (include/nthash/nthash.hpp) on top of it
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib")
HIP_SO = os.path.join(LIB, "libnthash_hip.so")
FACADE_SO = os.path.join(LIB, "libnthash.so")
ARCH = "gfx950"


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: nthash_amd needs the ROCm toolchain to build")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _sources(*names):
    return [os.path.join(CSRC, n) for n in names]


def build(force=False, verbose=False):
    os.makedirs(LIB, exist_ok=True)
    hipcc = _hipcc()
    common = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
              "-Wno-pass-failed", f"-I{os.path.join(ROOT, 'include')}"]
    hip_deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))] + \
        [os.path.join(ROOT, "include", "nthash_hip.h")]
    if force or _newer(HIP_SO, hip_deps):
        cmd = common + [os.path.join(CSRC, "nthip_capi.hip"), "-o", HIP_SO]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    # the C++ host facade is plain host C++17 on top of the C-ABI
    facade_src = os.path.join(CSRC, "nthash_facade.cpp")
    deps = [facade_src, os.path.join(CSRC, "nt_math.hpp"), os.path.join(CSRC, "seed_parse.hpp"),
            os.path.join(ROOT, "include", "nthash", "nthash.hpp"),
            os.path.join(ROOT, "include", "nthash_hip.h"), HIP_SO]
    if force or _newer(FACADE_SO, deps):
        cxx = shutil.which("g++") or shutil.which("c++") or hipcc
        cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra",
               f"-I{os.path.join(ROOT, 'include')}", facade_src, "-o", FACADE_SO,
               f"-L{LIB}", "-lnthash_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return HIP_SO


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(HIP_SO)
