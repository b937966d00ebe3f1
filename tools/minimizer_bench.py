#!/usr/bin/env python3
"""Per-read (w, k)-minimizers: whole-call rate on device-resident reads (150 bp, k = 31).

    python tools/minimizer_bench.py [reads=20000000] [w=10] [shape=clean|dirty|var]

dirty: an N in one read of 20 (the whole round takes the by-position form); var: the same reads given by offsets
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nthash_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
shape = sys.argv[3] if len(sys.argv) > 3 else "clean"
L, k = int(os.environ.get("MZ_L", "150")), 31
nwin = L - k + 1
ctx = nthash_amd.Context(0)
ctx.set_profiling(True)
d_in = ctx.malloc(n * L)
ctx.synth_reads_ptr(d_in, 0, n, L, 42)
d_offs = 0
if shape == "dirty":
    at = (np.arange(n // 20, dtype=np.int64) * 20 * L + 77)
    for a in at[:: max(1, len(at) // 2000)]:      # (2000 single-byte copies are enough to make every round dirty)
        ctx.h2d(d_in + int(a), np.frombuffer(b"N", dtype=np.uint8))
if shape == "var":
    offs = np.arange(n + 1, dtype=np.uint64) * L
    d_offs = ctx.malloc(offs.nbytes)
    ctx.h2d(d_offs, offs)
for w in ([int(sys.argv[2])] if len(sys.argv) > 2 and int(sys.argv[2]) else [5, 10, 19, 50]):
    cap = n * (2 * nwin // (w + 1) + 4)
    d_h, d_p, d_o = ctx.malloc(cap * 8), ctx.malloc(cap * 4), ctx.malloc((n + 1) * 8)
    best = 1e9
    for it in range(3):
        t0 = time.perf_counter()
        tot = ctx.minimizers_ptr(d_in, n, 0 if d_offs else L, 0, k, w, d_h, d_p, d_o, cap, offsets=d_offs)
        best = min(best, time.perf_counter() - t0)
    ms, name = ctx.last_kernel_ms()
    print(f"{shape} w={w:3d}: {best*1e3:8.2f} ms whole call, {n*nwin/best/1e9:6.1f} G k-mers/s, {tot/n:.2f} minimizers per read "
          f"(density {tot/(n*nwin):.3f}; 2/(w+1) = {2/(w+1):.3f}); {name} {ms:.2f} ms", flush=True)
    for p in (d_h, d_p, d_o):
        ctx.free(p)
