#!/usr/bin/env python3
"""Write-only fill rate of 8 GiB windows across ONE big allocation: is the speed of a page set a matter of where in the
allocation (i.e. where in HBM) it lies?   python tools/region_probe.py [GiB]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nthash_amd
gib = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ctx = nthash_amd.Context(0)
GB = 1 << 30
p = ctx.malloc(gib * GB)
W = 8 * GB
rates = []
for rep in range(2):
    row = []
    for off in range(0, gib * GB - W + 1, W):
        ms = ctx.fill_bench_ptr(p + off, W, 2)
        row.append(W / ms / 1e6)
    rates.append(row)
    print("pass", rep, " ".join(f"{r:5.0f}" for r in row), flush=True)
ctx.free(p)
