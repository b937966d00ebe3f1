"""nthash_amd/csrc/first_window.hpp on the CPU: the header compiles for the host as it does for the kernels, and
tests/host/first_window_host.cpp runs the grouped and the prefix-scan first window (64 "lanes" in a loop, the lane
partition and the exclusive XOR scan included) against nt_math.hpp's direct hashes -- 45 000 windows, k from 3 to 4099
across the 1023 period of the split rotate, slabs that end on and off a word boundary."""
import os
import subprocess

from conftest import ROOT


def test_first_window_forms_on_the_host(tmp_path):
    exe = os.path.join(str(tmp_path), "fw_host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-Werror",
                           os.path.join(ROOT, "tests", "host", "first_window_host.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("ok ") and int(out.stdout.split()[1]) > 40000
