#!/usr/bin/env python3
"""The binned query of hash STREAMS (stream_query_binned) against the direct kernels, in one process.

    python tools/stream_query_bench.py [k-mers in the stream, default 600 M] [filter log2 bits, 35]
Stream roads: nthip_stream_bloom_query (m = 1, 3), nthip_stream_count_query (m = 1), nthip_seed_bloom_query (config 4's
seed pair, 3 hashes per seed, 5 M reads of 250) and nthip_kmer_bloom_query of reads given by offsets.  Wall time, best of 3.
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nthash_amd
nk = int(sys.argv[1]) if len(sys.argv) > 1 else 600_000_000
lb = int(sys.argv[2]) if len(sys.argv) > 2 else 35
L, k = 150, 31
nwin = L - k + 1
n = nk // nwin
nk = n * nwin


def ctx_with(v):
    os.environ["NTHIP_TUNE_BLOOM_QUERY"] = str(v)
    try:
        return nthash_amd.Context(0)
    finally:
        os.environ.pop("NTHIP_TUNE_BLOOM_QUERY", None)


auto, direct = nthash_amd.Context(0), ctx_with(2)


def best(f, reps=3):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t0)
    return min(ts), r


d_in = auto.malloc(n * L)
auto.synth_reads_ptr(d_in, 0, n, L, 42)
n_bits = 1 << lb
d_f = auto.malloc(n_bits // 8)
auto.memset(d_f, 0, n_bits // 8)
auto.bloom_insert_ptr(d_in, n // 2, L, 0, k, 1, d_f, n_bits)
for m in (1, 3):
    nkm = nk // m // nwin * nwin
    d_h = auto.malloc(nkm * m * 8)
    tot = auto.kmer_hash_ptr(d_in, 0, nkm // nwin, L, 0, k, m, d_h, nkm)
    assert tot == nkm
    d_a, d_b = auto.malloc(nkm), auto.malloc(nkm)
    tb, fb = best(lambda: auto.stream_bloom_query_ptr(d_h, nkm, m, d_f, n_bits, d_a))
    td, fd = best(lambda: direct.stream_bloom_query_ptr(d_h, nkm, m, d_f, n_bits, d_b))
    a, b = np.zeros(nkm, np.uint8), np.zeros(nkm, np.uint8)
    auto.d2h(a, d_a); auto.d2h(b, d_b)
    print(f"stream_bloom_query m={m} {nkm*m/1e6:.0f} M values, {n_bits>>33} GiB filter: binned {tb*1e3:.2f} ms ({nkm*m/tb/1e9:.1f} G values/s)  "
          f"direct {td*1e3:.2f} ms ({nkm*m/td/1e9:.1f} G)  same={fb == fd and bool((a == b).all())} found={fb}", flush=True)
    if m == 1:
        n_c = 1 << 30
        d_c = auto.malloc(n_c)
        auto.memset(d_c, 0, n_c)
        auto.stream_count_insert_ptr(d_h, nkm // 2, d_c, n_c)
        tb, _ = best(lambda: auto.stream_count_query_ptr(d_h, nkm, 1, d_c, n_c, d_a))
        td, _ = best(lambda: direct.stream_count_query_ptr(d_h, nkm, 1, d_c, n_c, d_b))
        auto.d2h(a, d_a); auto.d2h(b, d_b)
        print(f"stream_count_query m=1 {nkm/1e6:.0f} M values, 1 Gi counters: binned {tb*1e3:.2f} ms ({nkm/tb/1e9:.1f} G/s)  direct {td*1e3:.2f} ms "
              f"({nkm/td/1e9:.1f} G)  same={bool((a == b).all())}", flush=True)
        auto.free(d_c)
    for p in (d_h, d_a, d_b):
        auto.free(p)
# reads given by offsets (all of 150: the lengths do not matter to the road)
offs = np.arange(n + 1, dtype=np.uint64) * L
d_o = auto.malloc(offs.nbytes)
auto.h2d(d_o, offs)
d_h1, d_h2 = auto.malloc(n * 8), auto.malloc(n * 8)
tb, rb = best(lambda: auto.bloom_query_ptr(d_in, n, 0, 0, k, 1, d_f, n_bits, hits=d_h1, offsets=d_o))
td, rd = best(lambda: direct.bloom_query_ptr(d_in, n, 0, 0, k, 1, d_f, n_bits, hits=d_h2, offsets=d_o))
h1, h2 = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
auto.d2h(h1, d_h1); auto.d2h(h2, d_h2)
print(f"kmer_bloom_query by offsets {nk/1e6:.0f} M k-mers: binned {tb*1e3:.2f} ms ({nk/tb/1e9:.1f} G k-mers/s)  direct {td*1e3:.2f} ms ({nk/td/1e9:.1f} G)  "
      f"same={rb == rd and bool((h1 == h2).all())}", flush=True)
# spaced seeds
SEED_A = "1111011101110010111001011011111"[:31]
SEED_B = SEED_A[::-1]
try:
    from bench import SEED_A, SEED_B  # noqa: F811
except Exception:
    pass
n4, L4 = 5_000_000, 250
d_in4 = auto.malloc(n4 * L4)
auto.synth_reads_ptr(d_in4, 0, n4, L4, 42)
sd_a, sd_d = nthash_amd.Seeds(auto, [SEED_A, SEED_B], 31), nthash_amd.Seeds(direct, [SEED_A, SEED_B], 31)
auto.memset(d_f, 0, n_bits // 8)
auto.seed_bloom_insert_ptr(d_in4, n4 // 2, L4, 0, sd_a, 3, d_f, n_bits)
d_h1, d_h2 = auto.malloc(n4 * 8), auto.malloc(n4 * 8)
tb, rb = best(lambda: auto.seed_bloom_query_ptr(d_in4, n4, L4, 0, sd_a, 3, d_f, n_bits, hits=d_h1))
td, rd = best(lambda: direct.seed_bloom_query_ptr(d_in4, n4, L4, 0, sd_d, 3, d_f, n_bits, hits=d_h2))
h1, h2 = np.zeros(n4, np.uint64), np.zeros(n4, np.uint64)
auto.d2h(h1, d_h1); auto.d2h(h2, d_h2)
w4 = n4 * (L4 - 30)
print(f"seed_bloom_query {w4/1e6:.0f} M windows x 6: binned {tb*1e3:.2f} ms ({w4/tb/1e9:.2f} G windows/s)  direct {td*1e3:.2f} ms ({w4/td/1e9:.2f} G)  "
      f"same={rb == rd and bool((h1 == h2).all())} {rb}", flush=True)
auto.set_profiling(True)
auto.seed_bloom_query_ptr(d_in4, n4, L4, 0, sd_a, 3, d_f, n_bits, hits=d_h1)
print("last:", auto.last_kernel_ms())
