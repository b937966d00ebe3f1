#!/usr/bin/env python3
"""kmer_hash of 20 M variable-length reads into FIVE different plain allocations of the output (all kept alive), same process:
how much of the box-to-box spread of `var` is the output buffer's page set.   python tools/var_alloc_spread.py"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nthash_amd
ctx = nthash_amd.Context(0)
n, k = 20_000_000, 31
rng = np.random.default_rng(1)
lens = rng.integers(100, 151, n).astype(np.uint64)
offs = np.zeros(n + 1, np.uint64); offs[1:] = np.cumsum(lens)
tb = int(offs[-1])
d_in = ctx.malloc(tb + 64)
ctx.synth_reads_ptr(d_in, 0, (tb + 149) // 150, 150, 42)
d_offs = ctx.malloc((n + 1) * 8); ctx.h2d(d_offs, offs)
cap = int((lens - k + 1).sum())
outs = [ctx.malloc(cap * 8) for _ in range(5)]
fills = []
for d_out in outs:
    r = ctx.fill_bench_ptr(d_out, cap * 8, 3)
    fills.append(r)
print("fill rate of the five allocations:", fills, flush=True)
for rep in range(2):
    for i, d_out in enumerate(outs):
        ts = []
        for _ in range(6):
            t0 = time.perf_counter(); tot = ctx.kmer_hash_ptr(d_in, d_offs, n, 0, 0, k, 1, d_out, cap); ts.append(time.perf_counter() - t0)
        print(f"pass {rep} output allocation {i} at {d_out:#x}: median {statistics.median(ts[1:])*1e3:.3f} ms  {tot/statistics.median(ts[1:])/1e9:.1f} G k-mers/s", flush=True)
