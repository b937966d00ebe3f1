#!/usr/bin/env python3
"""Fused Bloom-filter consumers vs the unfused pipeline (hash stream to HBM, then a consumer kernel).

    python tools/bloom_bench.py [reads] [m]
Device-resident reads (150 bp, k=31); wall time of the whole call, best of 4.
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nthash_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1
L, k = 150, 31
nwin = L - k + 1
ctx = nthash_amd.Context(0)
d_in = ctx.malloc(n * L)
ctx.synth_reads_ptr(d_in, 0, n, L, 42)
d_out = ctx.malloc(n * nwin * m * 8)
def best(f, reps=4):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t0)
    return min(ts), r
t_hash, _ = best(lambda: ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, m, d_out, n * nwin))
print(f"hash stream only            {t_hash*1e3:8.2f} ms  {n*nwin/t_hash/1e9:7.1f} Gkmer/s")
for n_bits in (1 << 26, (1 << 30) + 7, 1 << 32, 1 << 35):
    nbytes = (n_bits + 31) // 32 * 4
    d_f = ctx.malloc(nbytes)
    def fresh_insert():
        ctx.memset(d_f, 0, nbytes)
        t0 = time.perf_counter(); r = ctx.bloom_insert_ptr(d_in, n, L, 0, k, m, d_f, n_bits)
        return time.perf_counter() - t0, r
    t_ins, tot = min(fresh_insert() for _ in range(3))
    t_re, _ = best(lambda: ctx.bloom_insert_ptr(d_in, n, L, 0, k, m, d_f, n_bits))   # every bit already set
    t_q, (tq, found) = best(lambda: ctx.bloom_query_ptr(d_in, n, L, 0, k, m, d_f, n_bits))
    t_st, _ = best(lambda: ctx.stream_bloom_insert_ptr(d_out, n * nwin * m, d_f, n_bits))
    print(f"filter {nbytes/2**20:8.0f} MiB m={m}: fused insert {t_ins*1e3:8.2f} ms {tot/t_ins/1e9:6.1f} Gkmer/s | "
          f"re-insert {t_re*1e3:8.2f} ms {tot/t_re/1e9:6.1f} | "
          f"unfused (hash + stream insert) {(t_hash+t_st)*1e3:8.2f} ms {tot/(t_hash+t_st)/1e9:6.1f} | "
          f"fused query {t_q*1e3:8.2f} ms {tq/t_q/1e9:6.1f} Gkmer/s (found {found/tq:.3f})", flush=True)
    ctx.free(d_f)
