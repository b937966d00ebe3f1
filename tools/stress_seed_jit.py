#!/usr/bin/env python3
"""Randomised cross-check of the run-time specialised seed kernel (NTHIP_SEED_JIT=1, segment kernel forced) against the lane-per-read
state-machine kernel, clean fixed-length batches of random shapes.   python tools/stress_seed_jit.py [iterations] [seed]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NTHIP_SEED_JIT"] = "1"
os.environ["NTHIP_TUNE_SEED_PS"] = "1"
os.environ.setdefault("NTHIP_JIT_CACHE", "")
import numpy as np
import nthash_amd
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = nthash_amd.Context(0)
ref = nthash_amd.Context(0)
alph = np.frombuffer(b"ACGTacgtUu", dtype=np.uint8)
names, fails = collections.Counter(), 0
for it in range(iters):
    k = int(rng.choice([8, 15, 16, 17, 25, 31, 32, 33, 47, 48, 63, 64, 65, 80, 100, 127, 128, 160, 200]))
    m2 = int(rng.integers(1, 5))
    ns = int(rng.integers(1, 9)) if rng.random() < 0.4 else int(rng.integers(1, 4))
    seeds = []
    for _ in range(ns):
        dens = rng.choice([0.3, 0.6, 0.9])
        sd = "".join("1" if rng.random() < dens else "0" for _ in range(k))
        seeds.append("1" + sd[1:-1] + "1" if k > 1 else "1")
    L = int(k + rng.integers(0, 400))
    n = max(1, int(rng.integers(1, 600_000 // L + 2)))
    data = alph[rng.integers(0, len(alph), n * L)]
    ctx.set_profiling(True)
    a = ctx.seed_hash(data, seeds, k, m2, fixed_len=L, n_reads=n)
    names[ctx.last_kernel_ms()[1]] += 1
    b = ref.seed_hash(data, seeds, k, m2, fixed_len=L, n_reads=n, flags=4)
    if not (a["total"] == b["total"] and (a["hashes"] == b["hashes"]).all()):
        fails += 1
        print("MISMATCH", k, m2, L, n, seeds, ctx.last_kernel_ms()[1], flush=True)
print("done:", iters, "cases,", fails, "mismatches; kernels:", dict(names))
sys.exit(1 if fails else 0)
