#!/usr/bin/env python3
"""Randomised cross-check of the binned stream query / stream insert against the direct kernels (both on the GPU):
    python tools/stress_stream_query.py [cases, default 60] [seed]
Random table sizes (2^20 .. 2^34 bits / 2^17 .. 2^31 counters, not powers of two), m 1 .. 4, 10^5 .. 6 x 10^7 values, pieces /
shared cursors, buckets of the mean (overflow list in use), several rounds.  Every case: a filter built by the binned stream
insert == the one built by the atomic kernel; flags / estimates through the regions == the direct kernels'."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nthash_amd
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def ctx_with(env):
    for k_, v in env.items():
        os.environ[k_] = str(v)
    try:
        return nthash_amd.Context(0)
    finally:
        for k_ in env:
            os.environ.pop(k_, None)


direct = ctx_with({"NTHIP_TUNE_BLOOM_QUERY": 2, "NTHIP_TUNE_BLOOM_BINNED": 2})
L, k = 150, 31
bad = 0
for case in range(cases):
    m = int(rng.integers(1, 5))
    nk = int(10 ** rng.uniform(5, 7.3)) // (L - k + 1) * (L - k + 1)
    n_reads = nk // (L - k + 1)
    lb = rng.uniform(20, 34)
    n_bits = int(2 ** lb) + int(rng.integers(0, 1000))
    env = {"NTHIP_TUNE_BLOOM_QUERY": 1, "NTHIP_TUNE_BLOOM_BINNED": 1}
    if rng.random() < 0.3:
        env["NTHIP_TUNE_BLOOM_PIECES"] = 2
    if rng.random() < 0.3:
        env["NTHIP_TUNE_BLOOM_SLOT_TIGHT"] = 1
    if rng.random() < 0.3:
        env["NTHIP_TUNE_BLOOM_ROUND"] = int(nk * m / rng.uniform(1.5, 4))
    ctx = ctx_with(env)
    d_in = ctx.malloc(n_reads * L)
    ctx.synth_reads_ptr(d_in, int(rng.integers(0, 1 << 30)), n_reads, L, int(rng.integers(1, 1000)))
    d_h = ctx.malloc(nk * m * 8)
    assert ctx.kmer_hash_ptr(d_in, 0, n_reads, L, 0, k, m, d_h, nk) == nk
    ok = True
    # filter: insert the first 60 % of the k-mers' values through the lists and through the atomic kernel
    nbytes = (n_bits + 31) // 32 * 4
    d_f, d_g = ctx.malloc(nbytes), ctx.malloc(nbytes)
    ctx.memset(d_f, 0, nbytes); ctx.memset(d_g, 0, nbytes)
    n_ins = (nk * 6 // 10) * m
    ctx.stream_bloom_insert_ptr(d_h, n_ins, d_f, n_bits)
    direct.stream_bloom_insert_ptr(d_h, n_ins, d_g, n_bits)
    a, b = np.zeros(nbytes, np.uint8), np.zeros(nbytes, np.uint8)
    ctx.d2h(a, d_f); ctx.d2h(b, d_g)
    if not (a == b).all():
        ok = False; print("  insert differs:", int((a != b).sum()), "bytes")
    d_a, d_b = ctx.malloc(nk + 16), ctx.malloc(nk + 16)
    fa = ctx.stream_bloom_query_ptr(d_h, nk, m, d_f, n_bits, d_a)
    fb = direct.stream_bloom_query_ptr(d_h, nk, m, d_f, n_bits, d_b)
    a, b = np.zeros(nk, np.uint8), np.zeros(nk, np.uint8)
    ctx.d2h(a, d_a); ctx.d2h(b, d_b)
    if fa != fb or not (a == b).all():
        ok = False; print("  flags differ:", fa, fb, int((a != b).sum()))
    # sketch
    n_c = (n_bits // 8) // 4 * 4 + 4
    d_c, d_d = ctx.malloc(n_c), ctx.malloc(n_c)
    ctx.memset(d_c, 0, n_c); ctx.memset(d_d, 0, n_c)
    ctx.stream_count_insert_ptr(d_h, n_ins, d_c, n_c)
    direct.stream_count_insert_ptr(d_h, n_ins, d_d, n_c)
    a2, b2 = np.zeros(n_c, np.uint8), np.zeros(n_c, np.uint8)
    ctx.d2h(a2, d_c); ctx.d2h(b2, d_d)
    if not (a2 == b2).all():
        ok = False; print("  sketch differs:", int((a2 != b2).sum()))
    ctx.stream_count_query_ptr(d_h, nk, m, d_c, n_c, d_a)
    direct.stream_count_query_ptr(d_h, nk, m, d_c, n_c, d_b)
    ctx.d2h(a, d_a); ctx.d2h(b, d_b)
    if not (a == b).all():
        ok = False; print("  estimates differ:", int((a != b).sum()))
    print(f"case {case}: m={m} values={nk*m} n_bits=2^{lb:.2f} env={env} {'ok' if ok else 'FAILED'}", flush=True)
    bad += not ok
    for p in (d_in, d_h, d_f, d_g, d_a, d_b, d_c, d_d):
        ctx.free(p)
    ctx.close()
print("failed cases:", bad)
sys.exit(1 if bad else 0)
