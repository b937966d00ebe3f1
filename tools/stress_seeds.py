#!/usr/bin/env python3
"""Randomised cross-check of the spaced-seed fast paths (seed_fixed / split / seed_wave kernels) against the
lane-per-read state-machine kernel (both on the GPU).   python tools/stress_seeds.py [iterations] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nthash_amd
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = nthash_amd.Context(0)
alph = np.frombuffer(b"ACGTacgtUu", dtype=np.uint8)
bad_alph = np.frombuffer(b"NnRYKMSWBDHV-*.\x00\x01\x07", dtype=np.uint8)
fails = 0
for it in range(iters):
    k = int(rng.choice([3, 4, 5, 8, 15, 16, 17, 25, 31, 32, 33, 47, 48, 49, 63, 64, 65, 66, 80, 100, 127, 128, 200]))
    m2 = int(rng.integers(1, 9)) if rng.random() < 0.2 else int(rng.integers(1, 6))
    seeds = []
    for _ in range(int(rng.integers(1, 8)) if rng.random() < 0.35 else int(rng.integers(1, 4))):  # (many seeds: several passes)
        dens = rng.choice([0.3, 0.6, 0.9])
        sd = "".join("1" if rng.random() < dens else "0" for _ in range(k))
        if "1" not in sd:
            sd = "1" + sd[1:]
        seeds.append(sd)
    kind = rng.integers(0, 2) if rng.random() < 0.85 else 2
    bad_rate = rng.choice([0.0, 0.0005, 0.005, 0.05])
    if kind == 0:
        L = int(k + rng.integers(0, 300))
        n = max(1, int(rng.integers(1, 400_000 // L + 2)))
        total = n * L
        data = alph[rng.integers(0, len(alph), total)]
        nb = int(total * bad_rate)
        if nb:
            data[rng.integers(0, total, nb)] = bad_alph[rng.integers(0, len(bad_alph), nb)]
        kw = dict(fixed_len=L, n_reads=n)
        desc = f"fixed n={n} L={L}"
    elif kind == 2:  # a few long reads (cut into independent pieces): runs of non-bases of every length, some at the cuts
        n = int(rng.integers(1, 5))
        fixed = bool(rng.random() < 0.4)
        lens = np.full(n, int(rng.integers(16_384, 120_000))) if fixed else rng.integers(10_000, 250_000, n)
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        total = int(offs[-1])
        data = alph[rng.integers(0, len(alph), total)]
        for _ in range(int(rng.integers(0, 60))):
            ln = int(rng.choice([1, 1, 2, max(1, k - 1), k, k + 1, 3 * k, 500, 3000, 20000]))
            ln = min(ln, total)
            at = int(rng.integers(0, max(1, total - ln)))
            data[at:at + ln] = bad_alph[rng.integers(0, len(bad_alph))] if rng.random() < 0.7 else bad_alph[rng.integers(0, len(bad_alph), ln)]
        S = max(1280, 4 * k)
        for j in range(1, total // S):
            if rng.random() < 0.1:
                data[min(total - 1, j * S + int(rng.integers(0, S)))] = bad_alph[rng.integers(0, len(bad_alph))]
        kw = dict(fixed_len=int(lens[0]), n_reads=n) if fixed else dict(offsets=offs)
        desc = f"long n={n} bytes={total} fixed={fixed}"
    else:
        n = int(rng.integers(1, 1500))
        lens = np.where(rng.random(n) < 0.1, rng.integers(0, k + 2, n), rng.integers(0, 500, n))
        if rng.random() < 0.4:
            for _ in range(int(rng.integers(1, 4))):
                lens[rng.integers(0, n)] = int(rng.choice([1984, 2047, 2048, 2049, 4031, 4032, 6000, 20000]))  # segment edges
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        total = int(offs[-1])
        data = alph[rng.integers(0, len(alph), max(total, 1))]
        nb = int(total * bad_rate)
        if nb:
            data[rng.integers(0, total, nb)] = bad_alph[rng.integers(0, len(bad_alph), nb)]
        kw = dict(offsets=offs)
        desc = f"ragged n={n} bytes={total}"
    want_pos = bool(rng.random() < 0.4)
    want_str = bool(rng.random() < 0.3)
    os.environ.pop("NTHIP_TUNE_NO_SEED_WAVE", None)
    ctx.reload_tuning()
    a = ctx.seed_hash(data, seeds, k, m2, want_pos=want_pos, want_strands=want_str, **kw)
    os.environ["NTHIP_TUNE_NO_SEED_WAVE"] = "1"
    ctx.reload_tuning()
    b = ctx.seed_hash(data, seeds, k, m2, want_pos=want_pos, want_strands=want_str, flags=4, **kw)
    ok = a["total"] == b["total"] and (a["hashes"] == b["hashes"]).all() and (a["counts"] == b["counts"]).all()
    if want_pos:
        ok = ok and (a["pos"] == b["pos"]).all()
    if want_str:
        ok = ok and (a["fwd"] == b["fwd"]).all() and (a["rev"] == b["rev"]).all()
    if not ok:
        fails += 1
        print("MISMATCH", desc, k, m2, seeds, bad_rate, a["total"], b["total"], flush=True)
    elif it % 25 == 0:
        print("ok", it, desc, f"k={k} m2={m2} seeds={len(seeds)} bad={bad_rate} pos={want_pos} kmers={a['total']}", flush=True)
print("done:", iters, "cases,", fails, "mismatches")
sys.exit(1 if fails else 0)
