#!/usr/bin/env python3
"""The binned read side (bloom_query_kernels.hpp) against the direct kernels, in one process.

    python tools/query_bench.py [reads] [m list, e.g. 1,3] [filter log2 bits, e.g. 35] [counters log2, e.g. 30; 0: skip]
Device-resident reads (150 bp, k = 31); wall time of the whole call, best of 3.
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nthash_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
ms = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1").split(",")]
lb = int(sys.argv[3]) if len(sys.argv) > 3 else 35
lc = int(sys.argv[4]) if len(sys.argv) > 4 else 0
L, k = 150, 31
nwin = L - k + 1


def ctx_with(v):
    os.environ["NTHIP_TUNE_BLOOM_QUERY"] = str(v)
    try:
        return nthash_amd.Context(0)
    finally:
        os.environ.pop("NTHIP_TUNE_BLOOM_QUERY", None)


binned, direct, auto = ctx_with(1), ctx_with(2), nthash_amd.Context(0)
d_in = binned.malloc(n * L)
binned.synth_reads_ptr(d_in, 0, n, L, 42)
d_hits = binned.malloc(n * 8)


def best(f, reps=3):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t0)
    return min(ts), r


n_bits = 1 << lb
nbytes = n_bits // 8
d_f = binned.malloc(nbytes)
for m in ms:
    binned.memset(d_f, 0, nbytes)
    half = n // 2
    binned.bloom_insert_ptr(d_in, half, L, 0, k, m, d_f, n_bits)        # half of the reads are in the filter
    res = {}
    for name, c in (("binned", binned), ("direct", direct), ("default", auto)):
        t, (tq, found) = best(lambda: c.bloom_query_ptr(d_in, n, L, 0, k, m, d_f, n_bits, hits=d_hits))
        res[name] = (tq, found)
        print(f"filter 2^{lb} bits m={m} {name:8s} query {t*1e3:8.2f} ms {tq/t/1e9:6.1f} Gkmer/s  tested {tq} found {found}", flush=True)
    assert res["binned"] == res["direct"] == res["default"], res
binned.free(d_f)
if lc:
    n_cnt = 1 << lc
    d_c = binned.malloc(n_cnt)
    d_e = binned.malloc(n * nwin)
    for m in ms:
        binned.memset(d_c, 0, n_cnt)
        binned.count_insert_ptr(d_in, n // 2, L, 0, k, m, d_c, n_cnt)
        sums = {}
        for name, c in (("binned", binned), ("direct", direct)):
            t, tq = best(lambda: c.count_query_ptr(d_in, n, L, 0, k, m, d_c, n_cnt, d_e))
            sums[name] = (tq, binned.checksum_ptr(d_e, n * nwin // 8))
            print(f"sketch 2^{lc} counters m={m} {name:8s} query {t*1e3:8.2f} ms {tq/t/1e9:6.1f} Gkmer/s  checksum {sums[name][1]}", flush=True)
        assert sums["binned"] == sums["direct"], sums
