#!/usr/bin/env python3
"""Throughput of the N-aware paths on a dirty fixed-length batch (device-resident)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nthash_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
L, k, m = (int(x) for x in os.environ.get("DIRTY_SHAPE", "150,31,1").split(","))
nwin = L - k + 1
ctx = nthash_amd.Context(0)
d_in = ctx.malloc(n * L); d_out = ctx.malloc(n * nwin * m * 8)
ctx.synth_reads_ptr(d_in, 0, n, L, 42)
# one N every ~1000 reads
idx = np.arange(0, n * L, 1000 * L + 17, dtype=np.int64)
for i in idx[:20000]:
    ctx.h2d(d_in + int(i), np.frombuffer(b"N", np.uint8))
d_cnt = ctx.malloc(n * 8)
for name, flags in (("optimistic+na", 0), ("read slots", nthash_amd.capi.NTHIP_OUT_READ_SLOTS), ("general", 4)):
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        tot = ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, m, d_out, n * nwin, flags=flags, counts=d_cnt if name == "read slots" else 0)
        ts.append(time.perf_counter() - t0)
    t = min(ts)
    print(f"{name:14s} total={tot} ({n*nwin-tot} skipped)  {t*1e3:.2f} ms  {tot/t/1e9:.1f} Gkmer/s (wall, whole call)")
