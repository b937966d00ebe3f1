"""Source-level drop-in check (build container only; skipped where /root/reference
is absent): the reference's OWN test program and examples must compile against
nthash_amd's include/nthash/nthash.hpp and link with libnthash.so, unchanged.
(Running them needs an MI355X; their behaviour is covered on the GPU box by
tests/test_gpu_facade.py through fixtures recorded from the real reference.)"""
import os
import subprocess
import tempfile

import pytest

from conftest import ROOT

REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "tests")), reason="reference tree not present")
@pytest.mark.parametrize("src", ["tests/tests.cpp", "examples/kmer_hashing.cpp", "examples/benchmark.cpp"])
def test_reference_sources_compile_against_our_header(built_lib, src):
    lib = os.path.join(ROOT, "nthash_amd", "lib")
    assert os.path.exists(os.path.join(lib, "libnthash.so"))
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "prog")
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", f"-I{os.path.join(ROOT, 'include')}",
                               os.path.join(REF, src), f"-L{lib}", "-lnthash", "-lnthash_hip",
                               f"-Wl,-rpath,{lib}", "-o", exe])
        assert os.path.exists(exe)


def test_facade_library_exports_the_reference_api(built_lib):
    out = subprocess.check_output(["nm", "-DC", os.path.join(ROOT, "nthash_amd", "lib", "libnthash.so")], text=True)
    for sym in ("nthash::NtHash::roll()", "nthash::NtHash::roll_back()", "nthash::NtHash::peek(char)",
                "nthash::BlindNtHash::roll(char)", "nthash::SeedNtHash::roll()", "nthash::SeedNtHash::peek_back(char)",
                "nthash::BlindSeedNtHash::roll_back(char)", "nthash::parse_seeds("):
        assert sym in out, sym
