#!/usr/bin/env python3
"""Workload for rocprofv3 --pmc passes: one calibration copy of known size
(so FETCH_SIZE / WRITE_SIZE can be calibrated as MI355X_MICROARCH.md asks) and
then the hot kernels on a reduced batch.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python tools/pmc_probe.py [c2|c3|c4|mh|gen] [reads]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nthash_amd  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000_000
SEEDS = ["1010101010101010101010101010101", "1101101101101101011011011011011"]
if cfg.startswith("shape:"):  # shape:L,k,m -- any fixed-length NtHash shape
    _L, _k, _m = (int(x) for x in cfg[6:].split(","))
    SHAPE = {cfg: (_L, _k, _m, None)}
else:
    SHAPE = {}
L, k, m, seeds = SHAPE[cfg] if cfg in SHAPE else {"c2": (150, 31, 1, None), "c3": (150, 31, 4, None), "c4": (250, 31, 3, SEEDS),
                  "mh": (150, 31, 1, None), "gen": (150, 31, 1, None),      # mh: fused MinHash; gen: general kernel
                  "rag": (150, 31, 1, None),                                # rag: variable-length reads (100..150 bp)
                  "mz": (150, 31, 1, None),                                 # mz: (w, k)-minimizers, w = $MZ_W (10), one pass
                  "na": (150, 31, 1, None)}[cfg]                            # na: fixed-length batch with N's (N-aware pass)
if cfg == "gen":
    os.environ["NTHIP_TUNE_NO_SPECIAL"] = "1"
per = m if seeds is None else len(seeds) * m
nwin = L - k + 1
ctx = nthash_amd.Context(0)
COPY = 4 << 30
a = ctx.malloc(COPY)
b = ctx.malloc(COPY)
ctx.synth_reads_ptr(a, 0, COPY // 128, 128, 1)
print("copy 4GiB ms:", ctx.copy_bench_ptr(b, a, COPY, 2))
ctx.free(a)
ctx.free(b)
d_in = ctx.malloc(n * L)
d_out = ctx.malloc(n * nwin * per * 8)
ctx.synth_reads_ptr(d_in, 0, n, L, 42)
if cfg == "na":
    import numpy as np
    for i in range(0, n * L, 187_507):
        ctx.h2d(d_in + i, np.frombuffer(b"N", np.uint8))
if cfg == "rag":
    import numpy as np
    lens = np.random.default_rng(1).integers(100, 151, n).astype(np.uint64)
    offs = np.zeros(n + 1, np.uint64); offs[1:] = np.cumsum(lens)
    d_offs = ctx.malloc((n + 1) * 8); ctx.h2d(d_offs, offs)
ctx.set_profiling(True)
sd = nthash_amd.Seeds(ctx, seeds, k) if seeds else None
for _ in range(3):
    if cfg == "rag":
        ctx.kmer_hash_ptr(d_in, d_offs, n, 0, 0, k, m, d_out, n * nwin)
    elif cfg == "mh":
        ctx.minhash_ptr(d_in, n, L, 0, k, m, d_out)
    elif cfg == "mz":
        cap = n * nwin // 2
        ctx.minimizers_ptr(d_in, n, L, 0, k, int(os.environ.get("MZ_W", "10")), d_out, d_out + cap * 8, d_out + cap * 12, cap)
    elif sd is None:
        ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, m, d_out, n * nwin)
    else:
        ctx.seed_hash_ptr(d_in, 0, n, L, 0, sd, m, d_out, n * nwin)
    print("kernel ms:", ctx.last_kernel_ms(), "algorithmic GB:", n * nwin * (8 * per + L / nwin) / 1e9)
