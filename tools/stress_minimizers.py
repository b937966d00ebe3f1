#!/usr/bin/env python3
"""Minimizers: the register-table kernels against the LDS-table kernels (two independent implementations, both on the GPU)
on random shapes -- fixed-length reads clean / with non-bases, reads given by offsets -- and one large clean and one large
dirty batch (chunks of 256 reads per wave), compared through device checksums.

    python tools/stress_minimizers.py [iterations] [seed]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nthash_amd
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
ctx_reg = nthash_amd.Context(0)
os.environ["NTHIP_TUNE_MZ_TABLE"] = "1"
ctx_tab = nthash_amd.Context(0)
os.environ.pop("NTHIP_TUNE_MZ_TABLE")
alph = np.frombuffer(b"ACGTacgt", dtype=np.uint8)
fails = 0
for it in range(iters):
    k = int(rng.integers(3, 65)) if rng.random() < 0.8 else int(rng.integers(65, 200))
    w = int(rng.integers(1, 40)) if rng.random() < 0.8 else int(rng.integers(40, 300))
    var = rng.random() < 0.4
    nwin_max = int(rng.integers(1, 129)) if rng.random() < 0.6 else int(rng.integers(129, 257))
    L = k + nwin_max - 1
    n = int(rng.integers(1, 200_000 // L + 2)) if rng.random() < 0.8 else int(rng.integers(50_000, 200_000))
    if var:
        lens = rng.integers(0, L + 1, n).astype(np.uint64)
        lens[rng.integers(0, n)] = L
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    else:
        offs = None
    total = int(offs[-1]) if var else n * L
    data = alph[rng.integers(0, len(alph), max(total, 1))]
    if rng.random() < 0.3:                      # low complexity: ties
        data[: total // 2] = ord("A")
    dirty = rng.random() < 0.5
    if dirty and total:
        bad = rng.integers(0, total, max(1, total // int(rng.integers(50, 5000))))
        data[bad] = ord("N")
    dev = bool(rng.integers(0, 2))
    a = ctx_reg.minimizers(data, k, w, 0 if var else L, n, offsets=offs, device_input=dev)
    b = ctx_tab.minimizers(data, k, w, 0 if var else L, n, offsets=offs, device_input=dev)
    ok = a["total"] == b["total"] and (a["offsets"] == b["offsets"]).all() and (a["pos"] == b["pos"]).all() and (a["hashes"] == b["hashes"]).all()
    if not ok:
        fails += 1
        print(f"MISMATCH {it}: n={n} L={L} k={k} w={w} var={var} dirty={dirty} dev={dev} totals {a['total']} {b['total']}", flush=True)
    elif it % 25 == 0:
        print(f"ok {it} n={n} L={L} k={k} w={w} var={var} dirty={dirty} dev={dev} minimizers {a['total']}", flush=True)
# large batches: device checksums
n, L, k = 12_000_000, 150, 31
nwin = L - k + 1
d_in = ctx_reg.malloc(n * L)
ctx_reg.synth_reads_ptr(d_in, 0, n, L, 7)
for shape in ("clean", "dirty"):
    if shape == "dirty":
        for at in range(0, n, 4001):
            ctx_reg.h2d(d_in + at * L + (at % L), np.frombuffer(b"N", dtype=np.uint8))
    for w in (7, 24):
        cap = n * (2 * nwin // (w + 1) + 4)
        res = []
        for c in (ctx_reg, ctx_tab):
            d_h, d_p, d_o = ctx_reg.malloc(cap * 8), ctx_reg.malloc(cap * 4 + 8), ctx_reg.malloc((n + 1) * 8)
            ctx_reg.memset(d_p, 0, cap * 4 + 8)
            tot = c.minimizers_ptr(d_in, n, L, 0, k, w, d_h, d_p, d_o, cap)
            res.append((tot, c.checksum_ptr(d_h, tot), c.checksum_ptr(d_p, (tot + 1) // 2 if tot % 2 == 0 else tot // 2), c.checksum_ptr(d_o, n + 1)))
            for p in (d_h, d_p, d_o):
                ctx_reg.free(p)
        ok = res[0] == res[1]
        fails += not ok
        print(f"{'ok' if ok else 'MISMATCH'} large {shape} w={w}: {res[0][0]} minimizers, checksums {'equal' if ok else res}", flush=True)
print(f"done: {iters} cases + 4 large, {fails} mismatches")
sys.exit(1 if fails else 0)
