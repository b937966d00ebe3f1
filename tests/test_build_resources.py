"""No kernel of the library may spill registers: several kernels keep loads in flight behind inline asm that hipcc
cannot see (the next tile's slab), and a spilled register of such a load is saved before the load has landed -- wrong
hashes (found on seed_wtile_kernel<8> in round 2).  nthash_amd/build.py compiles with
-Rpass-analysis=kernel-resource-usage, refuses a unit whose kernels spill and leaves a report per unit."""
import glob
import json
import os

from conftest import ROOT


def test_no_kernel_spills(built_lib):
    reports = glob.glob(os.path.join(ROOT, "nthash_amd", "build", "capi_*.o.res.json"))
    assert len(reports) >= 10, "resource reports of the build are missing (python -m nthash_amd.build --force)"
    n = 0
    for f in reports:
        for k in json.load(open(f)):
            n += 1
            assert k.get("scratch", 0) == 0 and k.get("vgpr_spill", 0) == 0, (os.path.basename(f), k)
    assert n > 100
