#!/usr/bin/env python3
"""nthip_kmer_extend throughput: n k-mers -> own hashes + 4 successors + 4 predecessors (device-resident)."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nthash_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1
k = int(sys.argv[3]) if len(sys.argv) > 3 else 31
ctx = nthash_amd.Context(0)
d_in = ctx.malloc(n * k)
ctx.synth_reads_ptr(d_in, 0, n, k, 3)
d_self, d_next, d_prev = ctx.malloc(n * m * 8), ctx.malloc(n * 4 * m * 8), ctx.malloc(n * 4 * m * 8)
ctx.set_profiling(True)
L = ctx.L
for _ in range(4):
    t0 = time.perf_counter()
    rc = L.nthip_kmer_extend(ctx.h, C.c_void_p(d_in), n, k, m, C.c_void_p(d_self), C.c_void_p(d_next), C.c_void_p(d_prev), 0)
    dt = time.perf_counter() - t0
    assert rc == 0
ms, name = ctx.last_kernel_ms()
gb = n * (k + 9 * m * 8) / 1e9
print(f"{name} k={k} m={m}: {ms:.3f} ms for {n} k-mers = {n/ms/1e6:.1f} G k-mers/s ({9*m*n/ms/1e6:.0f} G hashes/s, {gb/ms:.2f} TB/s)")
