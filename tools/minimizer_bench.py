#!/usr/bin/env python3
"""Per-read (w, k)-minimizers: whole-call rate on device-resident reads (150 bp, k = 31).

    python tools/minimizer_bench.py [reads=20000000] [w=10]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nthash_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
L, k = 150, 31
nwin = L - k + 1
ctx = nthash_amd.Context(0)
ctx.set_profiling(True)
d_in = ctx.malloc(n * L)
ctx.synth_reads_ptr(d_in, 0, n, L, 42)
for w in ([int(sys.argv[2])] if len(sys.argv) > 2 else [5, 10, 19, 50]):
    cap = n * (2 * nwin // (w + 1) + 4)
    d_h, d_p, d_o = ctx.malloc(cap * 8), ctx.malloc(cap * 4), ctx.malloc((n + 1) * 8)
    best = 1e9
    for it in range(3):
        t0 = time.perf_counter()
        tot = ctx.minimizers_ptr(d_in, n, L, 0, k, w, d_h, d_p, d_o, cap)
        best = min(best, time.perf_counter() - t0)
    ms, name = ctx.last_kernel_ms()
    print(f"w={w:3d}: {best*1e3:8.2f} ms whole call, {n*nwin/best/1e9:6.1f} G k-mers/s, {tot/n:.2f} minimizers per read "
          f"(density {tot/(n*nwin):.3f}; 2/(w+1) = {2/(w+1):.3f}); {name} {ms:.2f} ms", flush=True)
    for p in (d_h, d_p, d_o):
        ctx.free(p)
