#!/usr/bin/env python3
"""Fused per-read MinHash signatures vs hashing to a stream (the write the consumer avoids).

    python tools/minhash_bench.py [reads] [read length]
Device-resident reads, k=31; wall time of the whole call, best of 4.
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nthash_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
k = 31
nwin = L - k + 1
ctx = nthash_amd.Context(0)
d_in = ctx.malloc(n * L)
ctx.synth_reads_ptr(d_in, 0, n, L, 42)
def best(f, reps=4):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t0)
    return min(ts), r
for m in (1, 2, 4, 8, 16):
    if n * nwin * m * 8 > 150e9: break
    d_out = ctx.malloc(n * nwin * m * 8)
    t_hash, _ = best(lambda: ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, m, d_out, n * nwin))
    ctx.free(d_out)
    d_sig = ctx.malloc(n * m * 8)
    t_mh, tot = best(lambda: ctx.minhash_ptr(d_in, n, L, 0, k, m, d_sig))
    ctx.free(d_sig)
    print(f"m={m:2d}: hash stream {t_hash*1e3:8.2f} ms {n*nwin/t_hash/1e9:7.1f} Gkmer/s | "
          f"fused minhash {t_mh*1e3:8.2f} ms {tot/t_mh/1e9:7.1f} Gkmer/s "
          f"({tot*m/t_mh/1e9:7.1f} G hashes/s, reads in at {n*L/t_mh/1e9:6.1f} GB/s)", flush=True)
