#!/usr/bin/env python3
"""Throughput on a FASTQ-like batch: variable-length reads (100..150 bp) with occasional N, device-resident.
RAGGED_MOSTLY=150: all reads that long except one in a thousand (trimmed to 100..149) -- an Illumina run as it comes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nthash_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
k, m = 31, 1
rng = np.random.default_rng(1)
lens = rng.integers(100, 151, n).astype(np.uint64)
if os.environ.get("RAGGED_MOSTLY"):
    full = int(os.environ["RAGGED_MOSTLY"])
    lens = np.where(rng.random(n) < 0.001, rng.integers(100, full, n), full).astype(np.uint64)
offs = np.zeros(n + 1, np.uint64); offs[1:] = np.cumsum(lens)
total_bytes = int(offs[-1])
ctx = nthash_amd.Context(0)
d_in = ctx.malloc(total_bytes + 64)
ctx.synth_reads_ptr(d_in, 0, (total_bytes + 149) // 150, 150, 42)      # random ACGT bytes
for i in range(0, total_bytes, 1_000_003):
    ctx.h2d(d_in + i, np.frombuffer(b"N", np.uint8))
d_offs = ctx.malloc((n + 1) * 8); ctx.h2d(d_offs, offs)
cap = int((lens - k + 1).sum())
d_out = ctx.malloc(cap * m * 8)
for name, flags in (("ragged run-split", 0), ("general lane/read", 4)):
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        tot = ctx.kmer_hash_ptr(d_in, d_offs, n, 0, 0, k, m, d_out, cap, flags=flags)
        ts.append(time.perf_counter() - t0)
    t = min(ts)
    print(f"{name:18s} reads={n} kmers={tot} ({cap-tot} skipped)  {t*1e3:.2f} ms  {tot/t/1e9:.1f} Gkmer/s (wall, whole call)")
