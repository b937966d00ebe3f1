#!/usr/bin/env python3
"""fill rate (two fills, nthip_fill_bench) of fresh plain allocations of several sizes, all held: which classes there are"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nthash_amd
os.environ["NTHIP_TUNE_MALLOC_PROBE"] = "1"
ctx = nthash_amd.Context(0)
held = []
for gb, cnt in ((2, 6), (15, 6), (36, 3), (90, 1)):
    rates = []
    for _ in range(cnt):
        p = ctx.malloc(gb << 30); held.append(p)
        ms = ctx.fill_bench_ptr(p, gb << 30, 2)
        rates.append(round((gb << 30) / ms / 1e6))
    print(f"{gb} GiB: {rates} GB/s", flush=True)
