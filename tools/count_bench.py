#!/usr/bin/env python3
"""k-mer counting sketch: whole-call rate of nthip_kmer_count_insert on fresh sketches of several sizes, the binned
insert against the compare-and-swap kernel (NTHIP_TUNE_BLOOM_BINNED=2), and the estimate query.

    python tools/count_bench.py [reads=20000000] [m=1]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nthash_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1
L, k = 150, 31
nwin = L - k + 1
ctx = nthash_amd.Context(0)
d_in = ctx.malloc(n * L)
ctx.synth_reads_ptr(d_in, 0, n, L, 42)
d_h = ctx.malloc(n * nwin * m * 8)
ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, m, d_h, n * nwin)
d_e = ctx.malloc(n * nwin)
for n_counters in (1 << 23, 1 << 27, 1 << 30):
    d_c = ctx.malloc(n_counters)
    def fresh():
        ctx.memset(d_c, 0, n_counters)
        t0 = time.perf_counter(); tot = ctx.count_insert_ptr(d_in, n, L, 0, k, m, d_c, n_counters)
        return time.perf_counter() - t0, tot
    t_ins, tot = min(fresh() for _ in range(3))
    def q():
        t0 = time.perf_counter(); ctx.stream_count_query_ptr(d_h, n * nwin, m, d_c, n_counters, d_e)
        return time.perf_counter() - t0
    t_q = min(q() for _ in range(3))
    print(f"sketch {n_counters/2**20:6.0f} Mi counters m={m}: insert {t_ins*1e3:8.2f} ms {tot/t_ins/1e9:6.1f} G k-mers/s | "
          f"estimates of a stream {t_q*1e3:8.2f} ms {n*nwin/t_q/1e9:6.1f} G k-mers/s", flush=True)
    ctx.free(d_c)
