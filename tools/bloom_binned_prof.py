#!/usr/bin/env python3
"""The binned Bloom insert alone (for rocprofv3 --kernel-trace --stats): reads -> fresh filter, a few times.

    python tools/bloom_binned_prof.py [reads=20000000] [m=1] [log2 filter bits=35]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nthash_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n_bits = 1 << (int(sys.argv[3]) if len(sys.argv) > 3 else 35)
L, k = 150, 31
ctx = nthash_amd.Context(0)
d_in = ctx.malloc(n * L)
ctx.synth_reads_ptr(d_in, 0, n, L, 42)
nbytes = (n_bits + 31) // 32 * 4
d_f = ctx.malloc(nbytes)
for it in range(4):
    ctx.memset(d_f, 0, nbytes)
    t0 = time.perf_counter()
    tot = ctx.bloom_insert_ptr(d_in, n, L, 0, k, m, d_f, n_bits)
    dt = time.perf_counter() - t0
    print(f"insert {it}: {dt*1e3:.2f} ms, {tot/dt/1e9:.1f} G k-mers/s ({tot*m/dt/1e9:.1f} G values/s)", flush=True)
