"""Round 5: the binned query of a hash STREAM (ntamd::host::stream_query_binned, nthash_amd/csrc/capi_sink_query.hip): the
values are partitioned to the table's regions, answered out of LDS and sent back -- behind nthip_stream_bloom_query,
nthip_stream_count_query, nthip_kmer_bloom_query of reads given by offsets and nthip_seed_bloom_query.  Every road against
the oracle's stream (reference emission rule src/kmer.cpp:228-264, src/seed.cpp:493-544) and against the direct kernels."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED_A = "111111111100000000001111111111"[:31].ljust(31, "1")
SEED_B = "1010101010101010101010101010101"


def _ctx_with(env):
    import nthash_amd
    for k_, v in env.items():
        os.environ[k_] = str(v)
    try:
        return nthash_amd.Context(0)
    finally:
        for k_ in env:
            os.environ.pop(k_, None)


def _filter_of(hashes, n_bits):
    nbytes = (n_bits + 31) // 32 * 4
    pos = hashes.ravel() % np.uint64(n_bits)
    filt = np.zeros(nbytes, np.uint8)
    np.bitwise_or.at(filt, (pos >> np.uint64(3)).astype(np.int64), (np.uint8(1) << (pos & np.uint64(7)).astype(np.uint8)))
    return filt


def _present(filt, hashes, n_bits, m):
    pos = hashes.ravel() % np.uint64(n_bits)
    bit = (filt[(pos >> np.uint64(3)).astype(np.int64)] >> (pos & np.uint64(7)).astype(np.uint8)) & 1
    return bit.reshape(-1, m).all(axis=1)


def _last_name(ctx):
    return ctx.last_kernel_ms()[1]


@pytest.mark.parametrize("n,L,k,m,n_bits,env,binned", [
    (6000, 150, 31, 1, (1 << 22) + 77, {}, True),                                   # 5 regions, one bin: level 1 goes to the regions
    (6000, 150, 31, 3, (1 << 28) + 12_345_677, {}, True),                           # 3 bins: two levels, pieces; m = 3: answers -> flags
    (6000, 150, 31, 3, (1 << 28) + 12_345_677, {"NTHIP_TUNE_BLOOM_PIECES": 2}, True),  # ... behind shared cursors
    (6000, 150, 31, 1, (1 << 30) + 5, {"NTHIP_TUNE_BLOOM_SLOT_TIGHT": 1, "NTHIP_TUNE_BLOOM_PIECES": 2}, True),
    (40000, 150, 31, 1, (1 << 33) + 12_345, {}, True),                              # 65 bins, 4.7 M values: pieces of every bin per block
    (6000, 150, 31, 2, 1 << 29, {"NTHIP_TUNE_BLOOM_SLOT_TIGHT": 1}, True),          # buckets of the mean: the overflow list on the way out and back
    (6000, 150, 31, 2, 1 << 24, {"NTHIP_TUNE_BLOOM_ROUND": 200_000}, True),         # several rounds of values
    (6000, 150, 31, 2, 1 << 29, {"NTHIP_TUNE_BLOOM_SLOT_TIGHT": 2}, False),         # the overflow list overflows: the direct kernel answers
    (300, 60, 21, 1, 12_345, {}, True),                                             # one region, not full
])
def test_stream_bloom_query_binned_flags(oracle, n, L, k, m, n_bits, env, binned):
    """flags / found of nthip_stream_bloom_query through the regions == the bits of the oracle's stream in a filter built on
    the CPU (half of the k-mers inserted), == the direct kernel's"""
    env = dict(env)
    env["NTHIP_TUNE_BLOOM_QUERY"] = 1
    ctx = _ctx_with(env)
    direct = _ctx_with({"NTHIP_TUNE_BLOOM_QUERY": 2})
    reads = oracle.synth_reads(0, n, L, 5 + k + m)
    offs = np.arange(n + 1, dtype=np.uint64) * L
    want = oracle.kmer_batch(reads, offs, k, m, want_pos=False)
    h = want["hashes"].reshape(-1, m)
    nk = h.shape[0]
    filt = _filter_of(h[::2], n_bits)
    exp = _present(filt, h, n_bits, m)
    d_f, nbytes = ctx.bloom_new(n_bits)
    ctx.h2d(d_f, filt)
    d_h = ctx.malloc(h.nbytes)
    ctx.h2d(d_h, np.ascontiguousarray(h))
    d_fl = ctx.malloc(nk + 16)
    for c_, want_binned in ((ctx, binned), (direct, False)):
        c_.memset(d_fl, 0xEE, nk + 16)
        c_.set_profiling(True)
        found = c_.stream_bloom_query_ptr(d_h, nk, m, d_f, n_bits, d_fl)
        name = _last_name(c_)
        c_.set_profiling(False)
        flags = np.zeros(nk + 16, np.uint8)
        c_.d2h(flags, d_fl)
        assert (flags[:nk] == exp).all(), int((flags[:nk] != exp).sum())
        assert (flags[nk:] == 0xEE).all()
        assert found == int(exp.sum())
        if want_binned:
            assert "binned stream query" in name or name == "answers_per_kmer_kernel", name
        else:
            assert name == "stream_bloom_flags_kernel", name
    for p_ in (d_f, d_h, d_fl):
        ctx.free(p_)
    ctx.close()
    direct.close()


@pytest.mark.parametrize("n,L,k,m,n_counters,env", [
    (6000, 150, 31, 1, (1 << 19) + 4, {}),                                # 5 regions of 2^17 counters
    (6000, 150, 31, 3, (1 << 25) + 40, {}),                               # 3 bins: pieces
    (6000, 150, 31, 3, (1 << 25) + 40, {"NTHIP_TUNE_BLOOM_PIECES": 2}),   # ... shared cursors
    (6000, 150, 31, 2, 1 << 26, {"NTHIP_TUNE_BLOOM_SLOT_TIGHT": 1}),
    (6000, 150, 31, 4, 1 << 22, {"NTHIP_TUNE_BLOOM_ROUND": 300_000}),
])
def test_stream_count_query_binned_estimates(oracle, n, L, k, m, n_counters, env):
    """nthip_stream_count_query through the regions: the smallest of a k-mer's m counters, against numpy on the oracle's stream"""
    env = dict(env)
    env["NTHIP_TUNE_BLOOM_QUERY"] = 1
    ctx = _ctx_with(env)
    reads = oracle.synth_reads(3, n, L, 9 + m)
    offs = np.arange(n + 1, dtype=np.uint64) * L
    h = oracle.kmer_batch(reads, offs, k, m, want_pos=False)["hashes"].reshape(-1, m)
    nk = h.shape[0]
    rng = np.random.default_rng(m)
    table = rng.integers(0, 256, n_counters, dtype=np.uint8)
    exp = table[(h % np.uint64(n_counters)).astype(np.int64)].min(axis=1)
    d_t = ctx.malloc(n_counters)
    ctx.h2d(d_t, table)
    d_h = ctx.malloc(h.nbytes)
    ctx.h2d(d_h, np.ascontiguousarray(h))
    d_e = ctx.malloc(nk + 16)
    ctx.memset(d_e, 0xEE, nk + 16)
    ctx.set_profiling(True)
    ctx.stream_count_query_ptr(d_h, nk, m, d_t, n_counters, d_e)
    name = _last_name(ctx)
    ctx.set_profiling(False)
    assert "binned stream query" in name or name == "answers_per_kmer_kernel", name
    est = np.zeros(nk + 16, np.uint8)
    ctx.d2h(est, d_e)
    assert (est[:nk] == exp).all(), int((est[:nk] != exp).sum())
    assert (est[nk:] == 0xEE).all()
    for p_ in (d_t, d_h, d_e):
        ctx.free(p_)
    ctx.close()


@pytest.mark.parametrize("m,n_bits,env", [
    (2, (1 << 28) + 5, {}),
    (1, 1 << 23, {"NTHIP_TUNE_BLOOM_ROUND": 150_000}),
])
def test_bloom_query_of_reads_by_offsets_through_the_regions(oracle, m, n_bits, env):
    """nthip_kmer_bloom_query of reads of any lengths: the compact stream, its answers through the regions, hits per read
    from the reads' offsets in the stream -- against the oracle's stream (reads with non-bases, empty and short reads)"""
    env = dict(env)
    env["NTHIP_TUNE_BLOOM_QUERY"] = 1
    ctx = _ctx_with(env)
    n, k = 5000, 25
    rng = np.random.default_rng(17 + m)
    lens = rng.integers(0, 400, n).astype(np.uint64)
    lens[:4] = [0, k - 1, k, 0]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    tb = int(offs[-1])
    a = oracle.synth_reads(1, 1, tb, 23).copy()
    a[rng.choice(tb, tb // 500, replace=False)] = ord("N")
    wa = oracle.kmer_batch(a, offs, k, m, want_pos=False)
    ha = wa["hashes"].reshape(-1, m)
    filt = _filter_of(ha[::3], n_bits)
    present = _present(filt, ha, n_bits, m)
    read_of = np.repeat(np.arange(n), wa["counts"].astype(np.int64))
    want_hits = np.bincount(read_of[present], minlength=n).astype(np.uint64)
    d_f, _ = ctx.bloom_new(n_bits)
    ctx.h2d(d_f, filt)
    d_in = ctx.malloc(tb + 16)
    ctx.h2d(d_in, a)
    d_o = ctx.malloc(offs.nbytes)
    ctx.h2d(d_o, offs)
    d_hits = ctx.malloc(n * 8)
    ctx.memset(d_hits, 0xEE, n * 8)
    ctx.set_profiling(True)
    total, found = ctx.bloom_query_ptr(d_in, n, 0, 0, k, m, d_f, n_bits, hits=d_hits, offsets=d_o)
    name = _last_name(ctx)
    ctx.set_profiling(False)
    assert name == "answers_per_read_kernel", name
    hits = np.zeros(n, np.uint64)
    ctx.d2h(hits, d_hits)
    assert total == wa["total"] and found == int(want_hits.sum())
    assert (hits == want_hits).all(), int((hits != want_hits).sum())
    from nthash_amd.capi import NTHIP_HOST_INPUT, NTHIP_HOST_OUTPUT
    hits2 = np.zeros(n, np.uint64)
    t2, f2 = ctx.bloom_query_ptr(a.ctypes.data, n, 0, 0, k, m, d_f, n_bits, hits=hits2.ctypes.data, offsets=offs.ctypes.data,
                                 flags=NTHIP_HOST_INPUT | NTHIP_HOST_OUTPUT)
    assert t2 == total and f2 == found and (hits2 == hits).all()
    for p_ in (d_f, d_in, d_o, d_hits):
        ctx.free(p_)
    ctx.close()


@pytest.mark.parametrize("by_offsets,env", [(False, {}), (True, {}), (False, {"NTHIP_TUNE_BLOOM_ROUND": 400_000}),
                                            (False, {"NTHIP_TUNE_BLOOM_PIECES": 2}), (True, {"NTHIP_TUNE_BLOOM_SLOT_TIGHT": 1, "NTHIP_TUNE_BLOOM_PIECES": 2})])
def test_seed_bloom_query_through_the_regions(oracle, by_offsets, env):
    """nthip_seed_bloom_query with the answers of the seeds' hashes coming through the regions: hits per read == the windows
    whose n_seeds * m2 hashes (oracle's seed_batch stream) all hit a filter built on the CPU"""
    import nthash_amd
    env = dict(env)
    env["NTHIP_TUNE_BLOOM_QUERY"] = 1
    ctx = _ctx_with(env)
    seeds, k, m2, n, L, n_bits = [SEED_A, SEED_B], 31, 2, 3000, 200, (1 << 28) + 99
    per = len(seeds) * m2
    sd = nthash_amd.Seeds(ctx, seeds, k)
    rng = np.random.default_rng(3)
    if by_offsets:
        lens = rng.integers(0, 350, n).astype(np.uint64)
        lens[:3] = [0, k - 1, k]
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    else:
        offs = np.arange(n + 1, dtype=np.uint64) * L
    tb = int(offs[-1])
    a = oracle.synth_reads(4, 1, tb, 41).copy()
    a[rng.choice(tb, tb // 600, replace=False)] = ord("N")
    wa = oracle.seed_batch(a, offs, seeds, k, m2, want_pos=False)
    ha = wa["hashes"].reshape(-1, per)
    filt = _filter_of(ha[::2], n_bits)
    present = _present(filt, ha, n_bits, per)
    read_of = np.repeat(np.arange(n), wa["counts"].astype(np.int64))
    want_hits = np.bincount(read_of[present], minlength=n).astype(np.uint64)
    d_f, _ = ctx.bloom_new(n_bits)
    ctx.h2d(d_f, filt)
    d_in = ctx.malloc(tb + 16)
    ctx.h2d(d_in, a)
    d_o = 0
    if by_offsets:
        d_o = ctx.malloc(offs.nbytes)
        ctx.h2d(d_o, offs)
    d_hits = ctx.malloc(n * 8)
    ctx.memset(d_hits, 0xEE, n * 8)
    ctx.set_profiling(True)
    tq, found = ctx.seed_bloom_query_ptr(d_in, n, 0 if by_offsets else L, 0, sd, m2, d_f, n_bits, hits=d_hits, offsets=d_o)
    name = _last_name(ctx)
    ctx.set_profiling(False)
    assert name == "answers_per_read_kernel", name
    hits = np.zeros(n, np.uint64)
    ctx.d2h(hits, d_hits)
    assert tq == wa["total"] and found == int(want_hits.sum())
    assert (hits == want_hits).all(), int((hits != want_hits).sum())
    for p_ in (d_f, d_in, d_hits) + ((d_o,) if d_o else ()):
        ctx.free(p_)
    sd.close()
    ctx.close()


def test_stream_query_default_choice_and_full_size(ctx):
    """the default context: a small stream keeps the direct kernel; 40 M values against a 256 MiB filter go through the
    regions and the flags are the direct kernel's, bit for bit (m = 1 and m = 2)"""
    direct = _ctx_with({"NTHIP_TUNE_BLOOM_QUERY": 2})
    n, L, k = 400_000, 150, 51
    n_bits = (1 << 31) + 1_234_567
    d_in = ctx.malloc(n * L)
    ctx.synth_reads_ptr(d_in, 0, n, L, 7)
    nk = n * (L - k + 1)
    for m in (1, 2):
        d_h = ctx.malloc(nk * m * 8)
        total = ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, m, d_h, nk)
        assert total == nk
        d_f, nbytes = ctx.bloom_new(n_bits)
        ctx.stream_bloom_insert_ptr(d_h, (nk // 2) * m, d_f, n_bits)          # the first half of the k-mers
        d_a, d_b = ctx.malloc(nk), ctx.malloc(nk)
        ctx.set_profiling(True)
        fa = ctx.stream_bloom_query_ptr(d_h, nk, m, d_f, n_bits, d_a)
        name = _last_name(ctx)
        ctx.set_profiling(False)
        assert "binned stream query" in name or name == "answers_per_kmer_kernel", name
        fb = direct.stream_bloom_query_ptr(d_h, nk, m, d_f, n_bits, d_b)
        a, b = np.zeros(nk, np.uint8), np.zeros(nk, np.uint8)
        ctx.d2h(a, d_a)
        ctx.d2h(b, d_b)
        assert fa == fb and (a == b).all(), int((a != b).sum())
        assert a[: nk // 2].all() and fa >= nk // 2
        # a small stream: the direct kernel
        ctx.set_profiling(True)
        ctx.stream_bloom_query_ptr(d_h, 100_000, m, d_f, n_bits, d_a)
        assert _last_name(ctx) == "stream_bloom_flags_kernel"
        ctx.set_profiling(False)
        for p_ in (d_h, d_f, d_a, d_b):
            ctx.free(p_)
    ctx.free(d_in)
    direct.close()


@pytest.mark.parametrize("m,n_bits", [(3, (1 << 28) + 12_345), (2, (1 << 29) + 7), (4, (1 << 28) + 1)])
def test_bloom_insert_of_reads_by_offsets_from_first_hashes(oracle, m, n_bits):
    """nthip_kmer_bloom_insert of reads of any lengths into a two-level filter: the round's stream holds hashes()[0] only and the
    first partition level makes the other m - 1 values (extend_hashes, src/internal.hpp:104-118) -- the filter is the one built on
    the CPU from the oracle's full stream; and the query of the same reads finds every k-mer"""
    ctx = _ctx_with({"NTHIP_TUNE_BLOOM_BINNED": 1, "NTHIP_TUNE_BLOOM_QUERY": 1})
    n, k = 4000, 27
    rng = np.random.default_rng(m)
    lens = rng.integers(0, 300, n).astype(np.uint64)
    lens[:3] = [0, k - 1, k]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    tb = int(offs[-1])
    a = oracle.synth_reads(6, 1, tb, 31).copy()
    a[rng.choice(tb, tb // 400, replace=False)] = ord("N")
    want = oracle.kmer_batch(a, offs, k, m, want_pos=False)
    filt = _filter_of(want["hashes"], n_bits)
    d_f, nbytes = ctx.bloom_new(n_bits)
    d_in = ctx.malloc(tb + 16)
    ctx.h2d(d_in, a)
    d_o = ctx.malloc(offs.nbytes)
    ctx.h2d(d_o, offs)
    ctx.set_profiling(True)
    total = ctx.bloom_insert_ptr(d_in, n, 0, 0, k, m, d_f, n_bits, offsets=d_o)
    name = _last_name(ctx)
    ctx.set_profiling(False)
    assert "pieces" in name, name
    assert total == want["total"]
    got = np.zeros(nbytes, np.uint8)
    ctx.d2h(got, d_f)
    assert (got == filt).all(), int((got != filt).sum())
    d_hits = ctx.malloc(n * 8)
    tq, found = ctx.bloom_query_ptr(d_in, n, 0, 0, k, m, d_f, n_bits, hits=d_hits, offsets=d_o)
    hits = np.zeros(n, np.uint64)
    ctx.d2h(hits, d_hits)
    assert tq == total and found == total and (hits == want["counts"]).all()
    for p_ in (d_f, d_in, d_o, d_hits):
        ctx.free(p_)
    ctx.close()
