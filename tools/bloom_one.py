#!/usr/bin/env python3
"""Whole-call time of nthip_kmer_bloom_insert (fresh 4 GiB filter) and nthip_kmer_count_insert (fresh 1 Gi counters) on
device-resident 150 bp reads, clean or with an N in one read of ~1000 -- the calls alone, for tools/kstats.sh.

    python tools/bloom_one.py [reads=20000000] [dirty=0] [reps=3]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nthash_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
dirty = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
L, k, m = 150, 31, 1
nwin = L - k + 1
ctx = nthash_amd.Context(0)
d_in = ctx.malloc(n * L)
ctx.synth_reads_ptr(d_in, 0, n, L, 42)
if dirty:
    rng = np.random.default_rng(1)
    enn = np.array([78], np.uint8)
    for p in rng.choice(n * L, n // 1000, replace=False):
        ctx.h2d(d_in + int(p), enn)
for name, size, ins in (("bloom 4 GiB", 1 << 32, lambda d: ctx.bloom_insert_ptr(d_in, n, L, 0, k, m, d, 1 << 35)),
                        ("count 1 Gi", 1 << 30, lambda d: ctx.count_insert_ptr(d_in, n, L, 0, k, m, d, 1 << 30))):
    d_t = ctx.malloc(size)
    ts = []
    for _ in range(reps):
        ctx.memset(d_t, 0, size)
        t0 = time.perf_counter(); tot = ins(d_t); ts.append(time.perf_counter() - t0)
    ctx.set_profiling(True); ctx.memset(d_t, 0, size); ins(d_t); kname = ctx.last_kernel_ms(); ctx.set_profiling(False)
    print(f"{name} {'dirty' if dirty else 'clean'}: {min(ts)*1e3:7.2f} ms  {tot/min(ts)/1e9:6.1f} G k-mers/s  ({tot} k-mers of {n*nwin}; {kname[1]} {kname[0]:.2f} ms)", flush=True)
    ctx.free(d_t)
