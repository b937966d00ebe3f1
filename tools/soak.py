#!/usr/bin/env python3
"""Soak test: large random batches, fast paths vs the lane-per-read general kernel by on-device checksums
(sum and xor of the whole stream) -- timing-dependent faults in the prefetch / counted-wait code would show here.

    python tools/soak.py [seconds] [seed]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nthash_amd
T = float(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = nthash_amd.Context(0)
CAP_IN, CAP_OUT = 3 << 30, 24 << 30
d_in = ctx.malloc(CAP_IN + 64); d_out = ctx.malloc(CAP_OUT); d_offs = ctx.malloc(8 * 40_000_001)
t_end = time.time() + T
it = fails = 0
while time.time() < t_end:
    k = int(rng.choice([11, 21, 25, 31, 31, 31, 32, 47, 55, 64, 96, 128]))
    m = int(rng.choice([1, 1, 1, 2, 3, 4]))
    ragged = rng.random() < 0.35
    dirty = rng.random() < 0.5
    if not ragged:
        L = int(rng.choice([k, k + 5, 100, 101, 125, 150, 150, 151, 250, 251, 300, 1000, 5003])) if rng.random() < 0.8 else int(rng.integers(k, 600))
        L = max(L, k)
        nwin = L - k + 1
        n = int(min(CAP_IN // L, CAP_OUT // (nwin * m * 8), 30_000_000))
        n = max(1, int(n * rng.uniform(0.3, 1.0)))
        ctx.synth_reads_ptr(d_in, int(rng.integers(0, 1 << 30)), n, L, int(rng.integers(0, 1 << 30)))
        total_bytes = n * L
        args = dict(seqs=d_in, offsets=0, n_reads=n, fixed_len=L, stride=0)
        cap = n * nwin
        desc = f"fixed n={n} L={L}"
    else:
        n = int(rng.integers(1_000_000, 12_000_000))
        lens = rng.integers(0, 320, n).astype(np.uint64)
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        total_bytes = int(offs[-1])
        ctx.synth_reads_ptr(d_in, 0, total_bytes // 64 + 1, 64, int(rng.integers(0, 1 << 30)))
        ctx.h2d(d_offs, offs)
        args = dict(seqs=d_in, offsets=d_offs, n_reads=n, fixed_len=0, stride=0)
        cap = int(np.maximum(lens.astype(np.int64) - k + 1, 0).sum())
        desc = f"ragged n={n} bytes={total_bytes}"
    if cap * m * 8 > CAP_OUT or cap == 0:
        continue
    if dirty:
        for pos in rng.integers(0, total_bytes, int(rng.integers(1, 200))):
            ctx.h2d(d_in + int(pos), np.frombuffer(b"N", np.uint8))
    tot = ctx.kmer_hash_ptr(hashes=d_out, capacity=cap, k=k, m=m, **args)
    s1 = ctx.checksum_ptr(d_out, tot * m)
    tot2 = ctx.kmer_hash_ptr(hashes=d_out, capacity=cap, k=k, m=m, flags=4, **args)
    s2 = ctx.checksum_ptr(d_out, tot2 * m)
    it += 1
    if (tot, s1) != (tot2, s2):
        fails += 1
        print("MISMATCH", desc, f"k={k} m={m} dirty={dirty}", tot, tot2, s1, s2, flush=True)
    elif it % 10 == 0:
        print(f"ok {it}: {desc} k={k} m={m} dirty={dirty} kmers={tot}", flush=True)
print("done:", it, "batches,", fails, "mismatches")
sys.exit(1 if fails else 0)
