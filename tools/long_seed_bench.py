#!/usr/bin/env python3
"""SeedNtHash on a few LONG reads (contigs / chromosomes with runs of N): cut into independent pieces
(seed_long_kernels.hpp) against one wave per read (NTHIP_TUNE_NO_SEED_LONG=1).
    python tools/long_seed_bench.py [Mbase per read] [reads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nthash_amd
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 50
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
k, m2 = 31, 3
SEEDS = ["1010101010101010101010101010101", "1101101101101101011011011011011"]
L = mb * 1_000_000
rng = np.random.default_rng(3)
offs = (np.arange(n + 1, dtype=np.uint64) * L)
outs = {}
for name, env in (("pieces", None), ("one wave per read", "1")):
    if env:
        os.environ["NTHIP_TUNE_NO_SEED_LONG"] = env
    ctx = nthash_amd.Context(0)
    os.environ.pop("NTHIP_TUNE_NO_SEED_LONG", None)
    sd = nthash_amd.Seeds(ctx, SEEDS, k)
    d_in = ctx.malloc(n * L + 64)
    ctx.synth_reads_ptr(d_in, 0, n * L // 64 + 1, 64, 42)
    # runs of N: one in ~200 kbase, 1 .. 50 000 long, the same for both contexts
    r2 = np.random.default_rng(5)
    for at in r2.integers(0, n * L - 60_000, n * L // 200_000):
        ln = int(r2.choice([int(x) for x in os.environ.get("LSB_RUNS", "1,10,100,1000,50000").split(",")]))
        if ln == 0:
            continue
        ctx.h2d(d_in + int(at), np.full(ln, ord("N"), np.uint8))
    d_offs = ctx.malloc((n + 1) * 8)
    ctx.h2d(d_offs, offs)
    cap = n * (L - k + 1)
    d_out = ctx.malloc(cap * len(SEEDS) * m2 * 8)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        tot = ctx.seed_hash_ptr(d_in, d_offs, n, 0, 0, sd, m2, d_out, cap)
        ts.append(time.perf_counter() - t0)
    outs[name] = (tot, ctx.checksum_ptr(d_out, tot * len(SEEDS) * m2))
    print(f"{name:20s} {n} x {mb} Mbase  kmers={tot}  {min(ts) * 1e3:9.2f} ms  {tot / min(ts) / 1e9:6.2f} G k-mers/s", flush=True)
    ctx.free(d_in); ctx.free(d_out); ctx.free(d_offs); sd.close(); ctx.close()
assert outs["pieces"] == outs["one wave per read"], outs
print("same k-mer count and checksum")
