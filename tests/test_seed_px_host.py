"""nthash_amd/csrc/seed_px_plan.hpp on the CPU: a spaced seed as a sparse sum over scanned term arrays.
tests/host/seed_px_host.cpp evaluates the plan the way seed_px_kernel.hpp does (terms in one frame, exclusive scans of
stride 1 / d, the plan's reads, one pair of split rotates per window) against the masked direct formula of nt_math.hpp --
400 random seed sets (dense, blocky, periodic, don't-cares at both ends; k from 3 to 200), every array form forced and the
planner's own choice -- and checks the collapses the plan exists for (1010...1: two reads of the stride-2 scan)."""
import os
import subprocess

from conftest import ROOT


def test_seed_px_plan_on_the_host(tmp_path):
    exe = os.path.join(str(tmp_path), "px_host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-Werror",
                           os.path.join(ROOT, "tests", "host", "seed_px_host.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("ok ") and int(out.stdout.split()[1]) > 100000
