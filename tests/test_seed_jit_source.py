"""capi_seed_jit.hip without a device: nthip_seed_jit_source returns the text hiprtc would compile for a seed set and read
length (seed_psj_kernel.inc behind the shape's constants); here hipcc compiles it for gfx950 and the kernel must neither
spill nor leave the LDS of a CU -- config 4's pair x 3, six seeds of 31 x 1, a long seed of few blocks, the reference's
ignore-path seed.  Shapes without a specialised kernel say so."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT

CASES = [
    (["1010101010101010101010101010101", "1101101101101101011011011011011"], 250, 3),
    (["1111011101110010111011110111011", "1110111011011101101110110111011", "1011101110111111111110111011101",
      "1101110101110111011101011101110"[:31], "1111111111111110111111111111111", "1110011100111001110011100111001"[:31]], 150, 1),
    (["1" * 40 + "0" * 48 + "1" * 40], 250, 1),
    (["1" * 20 + "0" * 5 + "1" * 14 + "0" * 3 + "1" * 22], 100, 2),
]


def _source(lib, seeds, length, m2):
    arr = (C.c_char_p * len(seeds))(*[s.encode() for s in seeds])
    out = C.c_void_p()
    rc = lib.nthip_seed_jit_source(arr, len(seeds), len(seeds[0]), length, m2, C.byref(out))
    if rc != 0:
        return None
    text = C.string_at(out.value).decode()
    C.CDLL(None).free(out)
    return text


@pytest.mark.parametrize("case", range(len(CASES)))
def test_specialised_seed_kernel_source_compiles_without_spills(built_lib, tmp_path, case):
    seeds, length, m2 = CASES[case]
    text = _source(built_lib, seeds, length, m2)
    assert text is not None and "PSJ_T_E" in text and 'extern "C" __global__' in text
    src = os.path.join(str(tmp_path), "psj.hip")
    open(src, "w").write(text)
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed",
                        "-Rpass-analysis=kernel-resource-usage", "-include", "hip/hip_runtime.h", "--cuda-device-only", "-c", src,
                        "-o", os.path.join(str(tmp_path), "psj.o")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    num = lambda name: int(re.search(name + r"[^:]*: (\d+)", r.stderr).group(1))
    waves = int(re.search(r"#define PSJ_WAVES (\d+)u", text).group(1))
    # (a kernel that spills at the planned block size is compiled again for a smaller one at run time: capi_seed.hip)
    if num("ScratchSize") != 0:
        assert waves > 8, (waves, num("ScratchSize"))
    assert num(r"LDS Size") <= 160 * 1024


def test_shapes_without_a_specialised_kernel(built_lib):
    assert _source(built_lib, ["1" * 21], 3000, 1) is None        # 59 windows per segment
    assert _source(built_lib, ["1" * 31], 150, 9) is None          # more hashes per seed than the kernel's table
    assert _source(built_lib, ["1" * 31], 20, 1) is None           # reads shorter than k
