#!/usr/bin/env python3
"""Would probing candidate allocations with a fill and keeping the fastest avoid the slow page sets?
    python tools/alloc_pick.py [reads] [trials] [candidates]"""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nthash_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cands = int(sys.argv[3]) if len(sys.argv) > 3 else 3
L, k = 150, 31
ctx = nthash_amd.Context(0)
ctx.set_profiling(True)
in_b, out_b = n * L, n * 120 * 8


def kernel_rate(d_in, d_out):
    ts = []
    for _ in range(5):
        ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, 1, d_out, n * 120)
        ts.append(ctx.last_kernel_ms()[0])
    return n * 120 / statistics.median(ts[1:]) / 1e6


def fill_rate(p, nbytes):
    return nbytes / ctx.fill_bench_ptr(p, nbytes, 2) / 1e6


for t in range(trials):
    # plain: the first allocation
    d_in = ctx.malloc(in_b); ctx.synth_reads_ptr(d_in, 0, n, L, 42)
    pool = []
    for c in range(cands):
        p = ctx.malloc(out_b)
        pool.append((fill_rate(p, out_b), p))
    first = pool[0]
    best = max(pool)
    r_first = kernel_rate(d_in, first[1])
    r_best = kernel_rate(d_in, best[1]) if best[1] != first[1] else r_first
    print(f"trial {t}: fills {' '.join(f'{f:5.0f}' for f, _ in pool)} GB/s   kernel on the first {r_first:5.0f} G, on the fastest-fill {r_best:5.0f} G", flush=True)
    for _, p in pool:
        ctx.free(p)
    ctx.free(d_in)
