#!/usr/bin/env python3
"""Does HOW the two buffers are allocated change which class of pages they get?  Headline shape, fresh allocations in one
process, four strategies in rotation.   python tools/alloc_strategies.py [reads] [rounds]"""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nthash_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
L, k = 150, 31
ctx = nthash_amd.Context(0)
ctx.set_profiling(True)
in_b, out_b = n * L, n * 120 * 8
GB = 1 << 30


def up(x, a):
    return (x + a - 1) // a * a


def strat_two():
    a = ctx.malloc(in_b); b = ctx.malloc(out_b)
    return a, b, [a, b]


def strat_out_first():
    b = ctx.malloc(out_b); a = ctx.malloc(in_b)
    return a, b, [a, b]


def strat_arena():
    r = ctx.malloc(up(in_b, 2 << 20) + out_b)
    return r, r + up(in_b, 2 << 20), [r]


def strat_gib():
    a = ctx.malloc(up(in_b, GB)); b = ctx.malloc(up(out_b, GB))
    return a, b, [a, b]


res = {}
hold = []
for r in range(rounds):
    for name, f in (("two hipMalloc", strat_two), ("out first", strat_out_first), ("one arena", strat_arena), ("GiB-rounded", strat_gib)):
        d_in, d_out, bufs = f()
        ctx.synth_reads_ptr(d_in, 0, n, L, 42)
        ts = []
        for _ in range(6):
            ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, 1, d_out, n * 120)
            ts.append(ctx.last_kernel_ms()[0])
        res.setdefault(name, []).append(n * 120 / statistics.median(ts[1:]) / 1e6)
        hold.append(bufs)
        if len(hold) > 1:          # free the previous set only after the next one exists: new pages every time
            for p in hold.pop(0):
                ctx.free(p)
for name, v in res.items():
    print(f"{name:14s} " + " ".join(f"{x:5.0f}" for x in v) + f"   median {statistics.median(v):5.0f} G k-mers/s", flush=True)
