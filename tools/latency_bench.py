#!/usr/bin/env python3
"""Per-call latency of the C-ABI on small device-resident batches (BASELINE config 1 shape and smaller)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nthash_amd
L, k, m = 150, 31, 1
nwin = L - k + 1
ctx = nthash_amd.Context(0)
for n in (1, 100, 10_000, 100_000, 1_000_000):
    d_in = ctx.malloc(n * L); d_out = ctx.malloc(n * nwin * 8)
    ctx.synth_reads_ptr(d_in, 0, n, L, 42)
    for _ in range(5):
        ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, m, d_out, n * nwin)
    t0 = time.perf_counter(); reps = 200
    for _ in range(reps):
        ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, m, d_out, n * nwin)
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, m, d_out, n * nwin, flags=nthash_amd.capi.NTHIP_ASYNC)
    assert ctx.take_dirty() is False
    da = (time.perf_counter() - t0) / reps
    print(f"n_reads={n:8d}  {dt*1e6:8.1f} us/call  {n*nwin/dt/1e9:8.2f} Gkmer/s   |  NTHIP_ASYNC {da*1e6:7.1f} us/call  {n*nwin/da/1e9:8.2f} Gkmer/s")
    ctx.free(d_in); ctx.free(d_out)
