#!/usr/bin/env python3
"""Kernel-time sweep over SeedNtHash shapes (read length, k, seeds, hashes per seed): which table layout / kernel a shape
gets and what it reaches.

    python tools/seed_sweep.py [out.json]
Prints kernel name, G k-mers/s and algorithmic TB/s (input bytes + 8 * seeds * m bytes per k-mer) per shape.
"""
import json, os, statistics, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NTHIP_SEED_JIT", "1")  # steady state: compile on the spot (a job meets its kernel after the first batches)
import nthash_amd

SHAPES = [  # (L, k, seeds, m per seed)
    (250, 31, 2, 3), (250, 31, 1, 1), (250, 31, 1, 3), (250, 31, 2, 1), (250, 31, 3, 1), (250, 31, 3, 3), (250, 31, 4, 2),
    (150, 31, 2, 3), (151, 31, 2, 3), (150, 24, 2, 2), (150, 48, 2, 3), (150, 64, 2, 3), (100, 64, 3, 1), (250, 40, 2, 3),
    (250, 31, 2, 5), (250, 31, 6, 1),
]
OUT_BUDGET = int(os.environ.get("SWEEP_GIB", "16")) << 30
if os.environ.get("SWEEP_SHAPES"):
    SHAPES = [tuple(int(x) for x in t.split(",")) for t in os.environ["SWEEP_SHAPES"].split(";")]
PROBED = os.environ.get("SWEEP_PROBED", "1") != "0"


def make_seeds(k, n, rng):
    out = []
    for _ in range(n):
        half = rng.random((k + 1) // 2) < 0.7
        s = np.concatenate([half, half[: k // 2][::-1]])  # symmetric care pattern, as the reference's examples
        s[0] = s[-1] = True
        out.append("".join("1" if b else "0" for b in s))
    return out


ctx = nthash_amd.Context(0)
ctx.set_profiling(True)
alloc = (lambda nb: ctx.malloc_probed(nb, 3)[0]) if PROBED else ctx.malloc
rng = np.random.default_rng(5)
rows = []
for (L, k, ns, m2) in SHAPES:
    nwin = L - k + 1
    per = ns * m2
    n = max(1, int(OUT_BUDGET // (nwin * per * 8)))
    seeds = nthash_amd.Seeds(ctx, make_seeds(k, ns, rng), k)
    d_in = alloc(n * L)
    d_out = alloc(n * nwin * per * 8)
    ctx.synth_reads_ptr(d_in, 0, n, L, 7)
    ts, name = [], "?"
    for it in range(6):
        ctx.seed_hash_ptr(d_in, 0, n, L, 0, seeds, m2, d_out, n * nwin)
        ms, name = ctx.last_kernel_ms()
        ts.append(ms)
    ms = statistics.median(ts[1:])
    kmers = n * nwin
    alg = n * L + kmers * per * 8
    row = dict(L=L, k=k, seeds=ns, m=m2, reads=n, kernel=name, ms=round(ms, 3), gkmer_s=round(kmers / ms / 1e6, 1),
               alg_tb_s=round(alg / ms / 1e9, 3))
    rows.append(row)
    print(f"L={L:4d} k={k:2d} seeds={ns} m={m2} {name:20s} {ms:8.3f} ms {row['gkmer_s']:7.1f} Gk/s "
          f"{row['alg_tb_s']:.2f} TB/s ({row['alg_tb_s'] / 8 * 100:.0f}% of 8)", flush=True)
    ctx.free(d_in); ctx.free(d_out); seeds.close()
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
