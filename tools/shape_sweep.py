#!/usr/bin/env python3
"""Kernel-time sweep over read shapes (length, k, hashes per k-mer): finds the cliffs in the dispatch.

    python tools/shape_sweep.py [out.json]
Prints kernel name, Gk-mer/s and algorithmic TB/s (input bytes + 8*m bytes per k-mer) per shape.
"""
import json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nthash_amd

SHAPES = [  # (L, k, m)
    (150, 31, 1), (151, 31, 1), (101, 31, 1), (100, 31, 1), (76, 31, 1), (250, 31, 1), (251, 31, 1),
    (300, 31, 1), (150, 21, 1), (150, 25, 1), (150, 51, 1), (150, 63, 1), (150, 64, 1), (150, 32, 1),
    (100, 64, 3), (100, 64, 1), (150, 31, 2), (150, 31, 4), (150, 31, 8), (151, 25, 2), (1000, 31, 1),
    (10000, 31, 1), (36, 21, 1), (50, 31, 1), (150, 15, 1), (150, 11, 1), (125, 31, 1), (149, 31, 1),
]
OUT_BUDGET = int(os.environ.get("SWEEP_GIB", "6")) << 30  # bytes of hashes per shape
if os.environ.get("SWEEP_SHAPES"):  # "151,31,1;101,31,1"
    SHAPES = [tuple(int(x) for x in t.split(",")) for t in os.environ["SWEEP_SHAPES"].split(";")]

ctx = nthash_amd.Context(0)
ctx.set_profiling(True)
# buffers from the placement-aware allocator (SWEEP_PROBED=0: plain allocations)
alloc = (lambda nb: ctx.malloc_probed(nb, 3)[0]) if os.environ.get("SWEEP_PROBED", "1") != "0" else ctx.malloc
rows = []
for (L, k, m) in SHAPES:
    nwin = L - k + 1
    n = max(1, int(OUT_BUDGET // (nwin * m * 8)))
    d_in = alloc(n * L)
    d_out = alloc(n * nwin * m * 8)
    ctx.synth_reads_ptr(d_in, 0, n, L, 7)
    ts = []
    name = "?"
    for it in range(6):
        ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, m, d_out, n * nwin, flags=0)
        ms, name = ctx.last_kernel_ms()
        ts.append(ms)
    ms = statistics.median(ts[1:])
    kmers = n * nwin
    alg = n * L + kmers * m * 8
    row = dict(L=L, k=k, m=m, nwin=nwin, reads=n, kernel=name, ms=round(ms, 3),
               gkmer_s=round(kmers / ms / 1e6, 1), alg_tb_s=round(alg / ms / 1e9, 3))
    rows.append(row)
    print(f"L={L:5d} k={k:2d} m={m} nwin={nwin:5d} {name:22s} {ms:8.3f} ms {row['gkmer_s']:7.1f} Gk/s "
          f"{row['alg_tb_s']:.2f} TB/s ({row['alg_tb_s']/8*100:.0f}% of 8)", flush=True)
    ctx.free(d_in); ctx.free(d_out)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
