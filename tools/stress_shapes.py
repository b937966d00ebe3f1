#!/usr/bin/env python3
"""Randomised cross-check of the fast paths against the lane-per-read general kernel (both on the GPU):
fixed-length shapes (dense general run-split kernel / N-aware pass), overlapping runs, ragged reads.

    python tools/stress_shapes.py [iterations] [seed]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nthash_amd
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = nthash_amd.Context(0)
alph = np.frombuffer(b"ACGTacgtUu", dtype=np.uint8)
bad_alph = np.frombuffer(b"NnRYKM-*.\x00", dtype=np.uint8)
fails = 0
for it in range(iters):
    k = int(rng.integers(3, 65)) if rng.random() < 0.6 else int(rng.integers(65, 400))
    m = int(rng.integers(1, 9)) if rng.random() < 0.8 else int(rng.integers(9, 40))
    kind = rng.integers(0, 3)
    if kind < 2:   # fixed length
        L = int(k + rng.integers(0, 40)) if rng.random() < 0.3 else int(rng.integers(k, 4000))
        n = max(1, int(rng.integers(1, 3_000_000 // L + 2)))
        stride = 0
        total = n * L
        if kind == 1 and L > k:  # overlapping runs of one sequence
            stride = L - k + 1
            total = (n - 1) * stride + L
        padded = kind == 0 and rng.random() < 0.3   # rows with padding between the reads (any bytes)
        if padded:
            stride = L + int(rng.integers(1, 200))
            total = (n - 1) * stride + L
        data = alph[rng.integers(0, len(alph), total)]
        if padded:
            if n > 1:
                np.lib.stride_tricks.as_strided(data[L:], (n - 1, stride - L), (stride, 1), writeable=True)[:] = \
                    (bad_alph if rng.random() < 0.5 else alph)[rng.integers(0, 9, (n - 1, stride - L))]
        if rng.random() < 0.5:
            nb = int(rng.integers(1, max(2, total // 2000)))
            data[rng.integers(0, total, nb)] = bad_alph[rng.integers(0, len(bad_alph) - 1, nb)]
        want_pos = bool(rng.random() < 0.3)
        want_str = bool(rng.random() < 0.25)
        a = ctx.kmer_hash(data, k, m, fixed_len=L, stride=stride, n_reads=n, want_pos=want_pos, want_strands=want_str)
        b = ctx.kmer_hash(data, k, m, fixed_len=L, stride=stride, n_reads=n, want_pos=want_pos, want_strands=want_str, flags=4)
        desc = f"fixed n={n} L={L} k={k} m={m} stride={stride} pos={want_pos}"
        if stride == 0 and L <= 2048 and L >= k and m <= 8 and rng.random() < 0.6:
            # NTHIP_OUT_READ_SLOTS on the same batch: read r's k-mers at the front of slot r * (L - k + 1), zeros behind, counts
            import nthash_amd.capi as capi
            nwin = L - k + 1
            hs, cs, ps = np.full(n * nwin * m, 0x5A, np.uint64), np.zeros(n, np.uint64), np.zeros(n * nwin, np.uint32)
            fl = capi.NTHIP_HOST_INPUT | capi.NTHIP_HOST_OUTPUT | capi.NTHIP_OUT_READ_SLOTS
            tot = ctx.kmer_hash_ptr(data.ctypes.data, 0, n, L, 0, k, m, hs.ctypes.data, n * nwin, counts=cs.ctypes.data,
                                    pos=ps.ctypes.data, flags=fl)
            hs = hs.reshape(n, nwin, m)
            bh = b["hashes"].reshape(-1, m)
            cnt = b["counts"].astype(np.int64)
            o = np.concatenate(([0], np.cumsum(cnt)))
            keep = np.arange(nwin)[None, :] < cnt[:, None]
            good = tot == n * nwin and (cs == b["counts"]).all() and (hs[keep] == bh).all() and (hs[~keep] == 0).all()
            if not good:
                fails += 1
                print("MISMATCH read slots", desc, tot, n * nwin, int((cs != b["counts"]).sum()), flush=True)
        if (stride == 0 or stride >= L - k + 1) and rng.random() < 0.5:   # the fused MinHash consumer on the same batch
            sig, tot = ctx.minhash(data, k, m, L, n, stride=stride)
            hs = b["hashes"].reshape(-1, m)
            cnt = b["counts"].astype(np.int64)
            exp = np.full((n, m), np.uint64(0xFFFFFFFFFFFFFFFF), np.uint64)
            st0 = np.concatenate(([0], np.cumsum(cnt)))[:-1]
            if (cnt > 0).any():
                exp[cnt > 0] = np.minimum.reduceat(hs, st0[cnt > 0], axis=0)
            if tot != b["total"] or not (sig == exp).all():
                fails += 1
                badix = np.argwhere(sig != exp)
                print("MISMATCH minhash", desc, tot, b["total"], "entries", len(badix), badix[:6].tolist(),
                      [(hex(int(sig[r, c])), hex(int(exp[r, c])), int(cnt[r])) for r, c in badix[:3]], flush=True)
    else:
        n = int(rng.integers(1, 4000))
        lens = np.where(rng.random(n) < 0.1, rng.integers(0, k + 2, n), rng.integers(0, 600 + 2 * k, n))
        if rng.random() < 0.2:
            lens[rng.integers(0, n)] = int(rng.integers(5000, 200000))
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        total = int(offs[-1])
        data = alph[rng.integers(0, len(alph), max(total, 1))]
        if total and rng.random() < 0.6:
            nb = int(rng.integers(1, max(2, total // 1000)))
            data[rng.integers(0, total, nb)] = bad_alph[rng.integers(0, len(bad_alph), nb)]
        want_pos = bool(rng.random() < 0.5)
        want_str = bool(rng.random() < 0.25)
        a = ctx.kmer_hash(data, k, m, offsets=offs, want_pos=want_pos, want_strands=want_str)
        b = ctx.kmer_hash(data, k, m, offsets=offs, want_pos=want_pos, want_strands=want_str, flags=4)
        desc = f"ragged n={n} bytes={total} k={k} m={m} pos={want_pos}"
    ok = a["total"] == b["total"] and (a["hashes"] == b["hashes"]).all() and (a["counts"] == b["counts"]).all()
    if want_pos:
        ok = ok and (a["pos"] == b["pos"]).all()
    if want_str:
        ok = ok and (a["fwd"] == b["fwd"]).all() and (a["rev"] == b["rev"]).all()
    if not ok:
        fails += 1
        print("MISMATCH", desc, a["total"], b["total"], flush=True)
    elif it % 25 == 0:
        print("ok", it, desc, "kmers", a["total"], flush=True)
print("done:", iters, "cases,", fails, "mismatches")
sys.exit(1 if fails else 0)
