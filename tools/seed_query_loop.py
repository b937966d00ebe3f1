#!/usr/bin/env python3
"""nthip_seed_bloom_query / _insert ten times in a row, every call's wall time (the calls allocate their hash streams)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nthash_amd
from bench import SEED_A, SEED_B
ctx = nthash_amd.Context(0)
n4, L4 = 5_000_000, 250
d_in = ctx.malloc(n4 * L4); ctx.synth_reads_ptr(d_in, 0, n4, L4, 42)
sd = nthash_amd.Seeds(ctx, [SEED_A, SEED_B], 31)
n_bits = 1 << 35
d_f = ctx.malloc(n_bits // 8); ctx.memset(d_f, 0, n_bits // 8)
d_hits = ctx.malloc(n4 * 8)
for what in ("insert", "query"):
    ts = []
    for i in range(10):
        t0 = time.perf_counter()
        if what == "insert": ctx.seed_bloom_insert_ptr(d_in, n4, L4, 0, sd, 3, d_f, n_bits)
        else: ctx.seed_bloom_query_ptr(d_in, n4, L4, 0, sd, 3, d_f, n_bits, hits=d_hits)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(what, " ".join(f"{t:.0f}" for t in ts), "ms", flush=True)
