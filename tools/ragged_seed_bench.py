#!/usr/bin/env python3
"""SeedNtHash on variable-length reads (offsets) with N's: seed_wave_kernel vs the lane-per-read kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nthash_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
k, m2 = 31, 3
SEEDS = ["1010101010101010101010101010101", "1101101101101101011011011011011"]
rng = np.random.default_rng(1)
if os.environ.get("RSB_SHAPE"):  # "k,seeds,m": random symmetric care patterns
    k, ns, m2 = (int(x) for x in os.environ["RSB_SHAPE"].split(","))
    SEEDS = []
    for _ in range(ns):
        half = rng.random((k + 1) // 2) < 0.7
        sm = np.concatenate([half, half[: k // 2][::-1]])
        sm[0] = sm[-1] = True
        SEEDS.append("".join("1" if x else "0" for x in sm))
lens = rng.integers(100, 251, n).astype(np.uint64)
offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
total_bytes = int(offs[-1])
ctx = nthash_amd.Context(0)
sd = nthash_amd.Seeds(ctx, SEEDS, k)
d_in = ctx.malloc(total_bytes + 64); d_offs = ctx.malloc((n + 1) * 8)
ctx.synth_reads_ptr(d_in, 0, total_bytes // 64 + 1, 64, 42)
ctx.h2d(d_offs, offs)
for i in np.arange(0, total_bytes, 600_011, dtype=np.int64)[:20000]:
    ctx.h2d(d_in + int(i), np.frombuffer(b"N", np.uint8))
cap = int((lens - k + 1).sum())
d_out = ctx.malloc(cap * len(SEEDS) * m2 * 8)
for name, env in (("rtile + wave kernels", None),) + ((("lane-per-read kernel", "1"),) if not os.environ.get("RSB_SHAPE") else ()):
    if env: os.environ["NTHIP_TUNE_NO_SEED_WAVE"] = env
    ctx.reload_tuning()
    ts = []
    for _ in range(3 if not env else 1):
        t0 = time.perf_counter(); tot = ctx.seed_hash_ptr(d_in, d_offs, n, 0, 0, sd, m2, d_out, cap); ts.append(time.perf_counter() - t0)
    print(f"{name:22s} reads={n} kmers={tot} ({cap-tot} skipped)  {min(ts)*1e3:9.2f} ms  {tot/min(ts)/1e9:6.1f} Gkmer/s", flush=True)
