#!/usr/bin/env python3
"""Spaced seeds of 65+ bases (and several-pass seed sets) on batches that leave the dense kernel: a fixed-length batch with
a few N's and a variable-length batch.  Whole call, device-resident.

    python tools/seed_dirty_long.py
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nthash_amd
ctx = nthash_amd.Context(0)
ctx.set_profiling(True)
rng = np.random.default_rng(3)
def seed(k):
    half = rng.random((k + 1) // 2) < 0.7
    s = np.concatenate([half, half[: k // 2][::-1]]); s[0] = s[-1] = True
    return "".join("1" if b else "0" for b in s)
for (L, k, ns, m2, n) in [(250, 80, 2, 2, 2_000_000), (250, 100, 1, 1, 2_000_000), (300, 160, 1, 1, 2_000_000), (250, 31, 6, 1, 2_000_000),
                          (250, 31, 2, 3, 2_000_000)]:
    seeds = nthash_amd.Seeds(ctx, [seed(k) for _ in range(ns)], k)
    nwin = L - k + 1
    d_in = ctx.malloc(n * L)
    ctx.synth_reads_ptr(d_in, 0, n, L, 11)
    host = np.zeros(n * L, np.uint8); ctx.d2h(host, d_in)
    host[rng.choice(n * L, n // 1000, replace=False)] = ord("N")   # an N in one read of ~1000
    ctx.h2d(d_in, host)
    d_out = ctx.malloc(n * nwin * ns * m2 * 8)
    best = 1e9
    for it in range(4):
        t0 = time.perf_counter()
        tot = ctx.seed_hash_ptr(d_in, 0, n, L, 0, seeds, m2, d_out, n * nwin)
        best = min(best, time.perf_counter() - t0)
    name = ctx.last_kernel_ms()[1]
    print(f"L={L} k={k} seeds={ns} m={m2}, an N in 1 read of 1000: {best*1e3:8.2f} ms, {tot/best/1e9:6.1f} G k-mers/s whole call ({name})", flush=True)
    ctx.free(d_in); ctx.free(d_out)
