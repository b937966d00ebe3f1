"""Round 5: the binned read side of the Bloom filter / counting sketch (nthash_amd/csrc/bloom_query_kernels.hpp,
capi_sink_query.hip) and nthip_kmer_count_query -- every road against the oracle's hash stream
(reference emission rule src/kmer.cpp:228-264, hashes()[i] src/internal.hpp:104-118)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx_with(env):
    import nthash_amd
    for k_, v in env.items():
        os.environ[k_] = str(v)
    try:
        return nthash_amd.Context(0)
    finally:
        for k_ in env:
            os.environ.pop(k_, None)


def _bloom_expected(hashes, n_bits):
    """filter bytes after setting bit (h mod n_bits) of every hash: bit p = bit p & 7 of byte p >> 3 (filters of gigabits:
    no array of one byte per bit)"""
    nbytes = (n_bits + 31) // 32 * 4
    pos = hashes.ravel() % np.uint64(n_bits)
    filt = np.zeros(nbytes, np.uint8)
    np.bitwise_or.at(filt, (pos >> np.uint64(3)).astype(np.int64), (np.uint8(1) << (pos & np.uint64(7)).astype(np.uint8)))
    return filt


def _present(filt, hashes, n_bits, m):
    pos = hashes % np.uint64(n_bits)
    bit = (filt[(pos >> np.uint64(3)).astype(np.int64)] >> (pos & np.uint64(7)).astype(np.uint8)) & 1
    return bit.reshape(-1, m).all(axis=1)


def _dirty(rng, reads, n, L, every=700):
    bad = rng.choice(n * L, max(3, n * L // every), replace=False)
    reads[bad] = np.frombuffer(b"NnRY-", dtype=np.uint8)[rng.integers(0, 5, bad.size)]


def _query_device(ctx, reads, n, L, k, m, d_f, n_bits, stride=0):
    """device-resident reads, device hits -> (hits, total, found, kernel of record)"""
    d_in = ctx.malloc(reads.size + 16)
    ctx.h2d(d_in, reads)
    d_hits = ctx.malloc(max(8, n * 8))
    ctx.memset(d_hits, 0xEE, max(8, n * 8))          # (every read's count must be WRITTEN, not added to)
    ctx.set_profiling(True)
    total, found = ctx.bloom_query_ptr(d_in, n, L, stride, k, m, d_f, n_bits, hits=d_hits)
    name = ctx.last_kernel_ms()[1]
    ctx.set_profiling(False)
    hits = np.zeros(n, np.uint64)
    ctx.d2h(hits, d_hits)
    ctx.free(d_in)
    ctx.free(d_hits)
    return hits, total, found, name


@pytest.mark.parametrize("n,L,k,m,n_bits,tight,round_values", [
    (5000, 150, 31, 1, 1 << 22, 0, 0),                   # one bin: level 1 straight to the regions
    (5000, 150, 31, 1, (1 << 28) + 4_000_037 * 32, 0, 0),  # 3 bins, a partial last region, not a power of two
    (3000, 150, 31, 3, 1 << 29, 0, 0),                   # m = 3: a k-mer hits when all three answers are 1
    (2500, 101, 25, 2, 3_000_000_128, 0, 0),             # 23 bins, invariant modulo
    (1100, 250, 64, 1, 1 << 30, 0, 0),                   # k = 64, 8 bins, a last tile that is not full
    (4000, 150, 31, 2, 1 << 28, 1, 0),                   # buckets of the mean exactly: the overflow list answers
    (4000, 150, 31, 1, 1 << 28, 2, 0),                   # ... of half the mean and a list of 64: the round fails -> direct kernel
    (6000, 150, 31, 1, 1 << 29, 0, 150_000),             # several rounds of reads
    (300, 48, 21, 1, 1 << 28, 0, 0), (1, 150, 31, 1, 1 << 28, 0, 0),
])
def test_bloom_binned_query_hits_per_read(oracle, n, L, k, m, n_bits, tight, round_values):
    """insert batch A, query batch B (half of A's reads + new ones, with non-bases) on the binned road
    (NTHIP_TUNE_BLOOM_QUERY=1): hits per read, k-mers tested and found == what the oracle's hashes and the expected filter
    say; and == the direct kernel's (NTHIP_TUNE_BLOOM_QUERY=2) answers"""
    env = {"NTHIP_TUNE_BLOOM_QUERY": 1}
    if tight:
        env["NTHIP_TUNE_BLOOM_SLOT_TIGHT"] = tight
    if round_values:
        env["NTHIP_TUNE_BLOOM_ROUND"] = round_values
    ctx = _ctx_with(env)
    direct = _ctx_with({"NTHIP_TUNE_BLOOM_QUERY": 2})
    rng = np.random.default_rng(11 * n + L + m)
    a_reads = oracle.synth_reads(0, n, L, 5)
    b_reads = oracle.synth_reads(n // 2, n, L, 5).copy()
    b_reads[: 3 * L] = ord("C")                            # a k-mer many times over (and absent from the filter)
    if n > 10:
        _dirty(rng, b_reads, n, L)
    offs = np.arange(n + 1, dtype=np.uint64) * L
    ha = oracle.kmer_batch(a_reads, offs, k, m, want_pos=False)
    filt = _bloom_expected(ha["hashes"], n_bits)
    d_f, nbytes = ctx.bloom_new(n_bits)
    ctx.h2d(d_f, filt)
    hb = oracle.kmer_batch(b_reads, offs, k, m, want_pos=False)
    present = _present(filt, hb["hashes"], n_bits, m)
    read_of = np.repeat(np.arange(n), hb["counts"].astype(np.int64))
    want_hits = np.bincount(read_of[present], minlength=n).astype(np.uint64)
    hits, total, found, name = _query_device(ctx, b_reads, n, L, k, m, d_f, n_bits)
    if tight == 2:
        assert name == "kmer_runs_gen_kernel(bloom query)", name     # the failed round was redone
    else:
        assert name.startswith("bloom binned query"), name
    assert total == hb["total"]
    assert (hits == want_hits).all(), int((hits != want_hits).sum())
    assert found == int(want_hits.sum())
    hits2, total2, found2, name2 = _query_device(direct, b_reads, n, L, k, m, d_f, n_bits)
    assert name2 == "kmer_runs_gen_kernel(bloom query)"
    assert total2 == total and found2 == found and (hits2 == hits).all()
    # host buffers on the same context: the direct road (the binned one is for device-resident reads), same answers
    hits3, total3, found3 = ctx.bloom_query(b_reads, k, m, L, n, d_f, n_bits)
    assert total3 == total and found3 == found and (hits3 == hits).all()
    ctx.free(d_f)
    ctx.close()
    direct.close()


def test_bloom_binned_query_padded_rows_and_default_choice(ctx, oracle):
    """stride > length (one read per line of a text file) on the binned road; and the default context keeps a small batch
    on the direct kernel"""
    n, L, k, m, n_bits, stride = 3000, 100, 31, 2, 1 << 28, 101
    forced = _ctx_with({"NTHIP_TUNE_BLOOM_QUERY": 1})
    rows = np.full(n * stride, ord("\n"), np.uint8)
    reads = oracle.synth_reads(3, n, L, 9)
    rows.reshape(n, stride)[:, :L] = reads.reshape(n, L)
    offs = np.arange(n + 1, dtype=np.uint64) * L
    h = oracle.kmer_batch(reads, offs, k, m, want_pos=False)
    filt = _bloom_expected(h["hashes"][: h["hashes"].shape[0] // 2], n_bits)     # the first half of the k-mers
    d_f, _ = forced.bloom_new(n_bits)
    forced.h2d(d_f, filt)
    present = _present(filt, h["hashes"], n_bits, m)
    want = present.reshape(n, L - k + 1).sum(axis=1).astype(np.uint64)
    hits, total, found, name = _query_device(forced, rows, n, L, k, m, d_f, n_bits, stride=stride)
    assert name.startswith("bloom binned query"), name
    assert total == h["total"] and found == int(want.sum()) and (hits == want).all()
    d_f2, _ = ctx.bloom_new(n_bits)
    ctx.h2d(d_f2, filt)
    hits2, total2, found2, name2 = _query_device(ctx, rows, n, L, k, m, d_f2, n_bits, stride=stride)
    assert name2 == "kmer_runs_gen_kernel(bloom query)", name2
    assert total2 == total and found2 == found and (hits2 == hits).all()
    ctx.free(d_f2)
    forced.free(d_f)
    forced.close()


@pytest.mark.parametrize("n,L,k,m,n_counters,binned,tight,dirty", [
    (4000, 150, 31, 1, 1 << 20, True, 0, False),          # 8 regions, one bin
    (4000, 150, 31, 3, (1 << 25) + 40_000, True, 0, True),  # 3 bins, a partial last region, reads with non-bases
    (3000, 150, 31, 2, 1 << 26, True, 1, True),           # the overflow list answers
    (3000, 150, 31, 2, 1 << 26, True, 2, False),          # the round fails: the stream road takes over
    (2000, 101, 25, 2, 999_984, False, 0, True),          # the stream road from the start
    (600, 40, 31, 1, 1 << 16, False, 0, False),
])
def test_kmer_count_query_matches_stream_query_on_the_oracle_stream(oracle, n, L, k, m, n_counters, binned, tight, dirty):
    """nthip_kmer_count_query (fixed-length device-resident reads; binned or by rounds of the compact stream) == the
    smallest of each emitted k-mer's m counters at its window (== what nthip_stream_count_query answers on the ORACLE's
    stream), 0 for the windows NtHash skips"""
    env = {"NTHIP_TUNE_BLOOM_QUERY": 1 if binned else 2}
    if tight:
        env["NTHIP_TUNE_BLOOM_SLOT_TIGHT"] = tight
    ctx = _ctx_with(env)
    rng = np.random.default_rng(5 * n + L + m)
    reads = oracle.synth_reads(1, n, L, 21).copy()
    reads[: 2 * L] = ord("G")
    if dirty:
        _dirty(rng, reads, n, L, every=400)
    offs = np.arange(n + 1, dtype=np.uint64) * L
    h = oracle.kmer_batch(reads, offs, k, m, want_pos=True)
    sketch = rng.integers(0, 256, n_counters, dtype=np.int64).astype(np.uint8)
    sketch[::7] = 0
    d_c = ctx.malloc(n_counters)
    ctx.h2d(d_c, sketch)
    hs = np.ascontiguousarray(h["hashes"]).reshape(-1, m)
    n_kmers = hs.shape[0]
    # the stream query on the oracle's stream
    d_h, d_e = ctx.malloc(max(8, hs.size * 8)), ctx.malloc(max(4, n_kmers))
    ctx.h2d(d_h, hs.ravel())
    ctx.stream_count_query_ptr(d_h, n_kmers, m, d_c, n_counters, d_e)
    est_stream = np.zeros(n_kmers, np.uint8)
    ctx.d2h(est_stream, d_e)
    assert (est_stream == sketch[(hs % np.uint64(n_counters)).astype(np.int64)].min(axis=1)).all()
    nwin = L - k + 1
    want = np.zeros(n * nwin, np.uint8)
    read_of = np.repeat(np.arange(n), h["counts"].astype(np.int64))
    want[read_of * nwin + h["pos"].astype(np.int64)] = est_stream
    # device-resident reads, device estimates
    d_in, d_out = ctx.malloc(reads.size + 16), ctx.malloc(n * nwin)
    ctx.h2d(d_in, reads)
    ctx.memset(d_out, 0xEE, n * nwin)
    ctx.set_profiling(True)
    total = ctx.count_query_ptr(d_in, n, L, 0, k, m, d_c, n_counters, d_out)
    name = ctx.last_kernel_ms()[1]
    ctx.set_profiling(False)
    got = np.zeros(n * nwin, np.uint8)
    ctx.d2h(got, d_out)
    if binned and tight != 2:
        assert name.startswith("count binned query"), name
    else:
        assert not name.startswith("count binned query"), name
    assert total == h["total"]
    assert (got == want).all(), int((got != want).sum())
    # host buffers: the stream road, same answers
    got2, total2 = ctx.count_query(reads, k, m, L, n, d_c, n_counters)
    assert total2 == total and (got2 == want).all()
    for p in (d_in, d_out, d_h, d_e, d_c):
        ctx.free(p)
    ctx.close()


def test_kmer_count_query_reads_by_offsets(ctx, oracle):
    """reads of any lengths: read r's windows at the sum of the windows of the reads before it"""
    n, k, m, n_counters = 700, 25, 2, 1 << 18
    rng = np.random.default_rng(99)
    lens = rng.integers(0, 400, n).astype(np.uint64)
    lens[:3] = [0, k - 1, k]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    total_b = int(offs[-1])
    reads = oracle.synth_reads(4, 1, total_b, 3).copy()
    reads[rng.choice(total_b, total_b // 300, replace=False)] = ord("N")
    h = oracle.kmer_batch(reads, offs, k, m, want_pos=True)
    sketch = rng.integers(0, 256, n_counters, dtype=np.int64).astype(np.uint8)
    d_c = ctx.malloc(n_counters)
    ctx.h2d(d_c, sketch)
    wins = np.maximum(lens.astype(np.int64) - k + 1, 0)
    slot = np.concatenate([[0], np.cumsum(wins)])
    hs = np.ascontiguousarray(h["hashes"]).reshape(-1, m)
    est = sketch[(hs % np.uint64(n_counters)).astype(np.int64)].min(axis=1)
    want = np.zeros(int(slot[-1]), np.uint8)
    read_of = np.repeat(np.arange(n), h["counts"].astype(np.int64))
    want[slot[read_of] + h["pos"].astype(np.int64)] = est
    got, total = ctx.count_query(reads, k, m, 0, n, d_c, n_counters, offsets=offs)
    assert total == h["total"]
    assert (got == want).all(), int((got != want).sum())
    # device-resident batch
    d_in, d_o, d_out = ctx.malloc(total_b + 16), ctx.malloc(offs.nbytes), ctx.malloc(max(4, want.size))
    ctx.h2d(d_in, reads)
    ctx.h2d(d_o, offs)
    total2 = ctx.count_query_ptr(d_in, n, 0, 0, k, m, d_c, n_counters, d_out, offsets=d_o)
    got2 = np.zeros(want.size, np.uint8)
    ctx.d2h(got2, d_out)
    assert total2 == total and (got2 == want).all()
    for p in (d_in, d_o, d_out, d_c):
        ctx.free(p)


def test_kmer_count_query_argument_errors(ctx):
    import nthash_amd
    d_c, d_e = ctx.malloc(1024), ctx.malloc(4096)
    data = np.frombuffer(b"ACGT" * 50, dtype=np.uint8)
    d_in = ctx.malloc(256)
    ctx.h2d(d_in, data)
    for bad in (lambda: ctx.count_query_ptr(d_in, 1, 200, 0, 31, 1, 0, 1024, d_e),         # NULL sketch
                lambda: ctx.count_query_ptr(d_in, 1, 200, 0, 31, 1, d_c + 1, 1024, d_e),   # unaligned
                lambda: ctx.count_query_ptr(d_in, 1, 200, 0, 31, 1, d_c, 1022, d_e),       # not a multiple of 4
                lambda: ctx.count_query_ptr(d_in, 1, 200, 0, 31, 1, d_c, 1024, 0),         # NULL estimates
                lambda: ctx.count_query_ptr(d_in, 1, 200, 0, 0, 1, d_c, 1024, d_e),
                lambda: ctx.count_query_ptr(d_in, 1, 200, 0, 31, 0, d_c, 1024, d_e)):
        with pytest.raises(nthash_amd.NtHipError):
            bad()
    assert ctx.count_query_ptr(d_in, 10, 20, 0, 31, 1, d_c, 1024, d_e) == 0      # reads shorter than k: nothing to estimate
    for p in (d_c, d_e, d_in):
        ctx.free(p)


# ---- spaced-seed consumers: nthip_seed_bloom_insert / _query -----------------------------------------------------------------
SEED_A = "1010101010101010101010101010101"       # BASELINE config 4's pair
SEED_B = "1101101101101101011011011011011"
SEEDS_K48 = ["110110110110110110110110011011011011011011011011", "101101101101101101101101101101101101101101101101",
             "111100001111000011110000000011110000111100001111"]


@pytest.mark.parametrize("seeds,k,m2,n,L,n_bits,rounds,by_offsets", [
    ([SEED_A, SEED_B], 31, 3, 3000, 250, (1 << 28) + 12_345_677, False, False),   # config 4's shape; 3 bins: the binned insert's two levels
    ([SEED_A, SEED_B], 31, 3, 2000, 250, 1 << 22, True, False),                   # several rounds of reads
    (SEEDS_K48, 48, 2, 1500, 150, 40_000_003, False, False),                      # three seeds of 48
    (SEEDS_K48, 48, 1, 1200, 0, 1 << 24, False, True),                            # reads of any lengths
    ([SEED_A], 31, 1, 400, 40, 1 << 20, False, False),
    # the insert from hashes()[0] alone, the first partition level making the other m2 - 1 values (forced through the lists)
    ([SEED_A, SEED_B], 31, 3, 3000, 250, (1 << 28) + 12_345_677, "binned", False),
    ([SEED_A, SEED_B], 31, 2, 2500, 250, (1 << 29) + 5, "binned", False),
    (SEEDS_K48, 48, 4, 1500, 150, (1 << 28) + 77, "binned", True),
])
def test_seed_bloom_insert_and_query_match_oracle_seed_stream(oracle, seeds, k, m2, n, L, n_bits, rounds, by_offsets):
    """the filter after nthip_seed_bloom_insert == the filter built on the CPU from the oracle's seed_batch stream (every one
    of the n_seeds * m2 hashes of every window the reference's SeedNtHash emits, src/seed.cpp:493-544 -- reads with
    non-bases included); nthip_seed_bloom_query's hits per read == the windows whose hashes all hit that filter"""
    env = {"NTHIP_TUNE_BLOOM_BINNED": 1} if rounds == "binned" else {"NTHIP_TUNE_BLOOM_ROUND": 700_000} if rounds else {}
    ctx = _ctx_with(env)
    import nthash_amd
    sd = nthash_amd.Seeds(ctx, seeds, k)
    per = len(seeds) * m2
    rng = np.random.default_rng(n + k + m2)
    if by_offsets:
        lens = rng.integers(0, 300, n).astype(np.uint64)
        lens[:3] = [0, k - 1, k]
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    else:
        offs = np.arange(n + 1, dtype=np.uint64) * L
    total_b = int(offs[-1])
    a = oracle.synth_reads(2, 1, total_b, 17 + k).copy()
    bad = rng.choice(total_b, max(3, total_b // 600), replace=False)
    a[bad] = np.frombuffer(b"NnRY-", dtype=np.uint8)[rng.integers(0, 5, bad.size)]
    want = oracle.seed_batch(a, offs, seeds, k, m2, want_pos=False)
    filt = _bloom_expected(want["hashes"], n_bits)
    d_f, nbytes = ctx.bloom_new(n_bits)
    d_in = ctx.malloc(total_b + 16)
    ctx.h2d(d_in, a)
    d_o = 0
    if by_offsets:
        d_o = ctx.malloc(offs.nbytes)
        ctx.h2d(d_o, offs)
    total = ctx.seed_bloom_insert_ptr(d_in, n, 0 if by_offsets else L, 0, sd, m2, d_f, n_bits, offsets=d_o)
    assert total == want["total"]
    got = np.zeros(nbytes, np.uint8)
    ctx.d2h(got, d_f)
    assert (got == filt).all(), int((got != filt).sum())
    # query another batch: the second half of the first + new bases, its own non-bases; against a filter with bits missing
    b = np.concatenate([a[total_b // 2:], oracle.synth_reads(9, 1, total_b // 2 + 1, 99)])[:total_b].copy()
    b[rng.choice(total_b, max(3, total_b // 500), replace=False)] = ord("N")
    wb = oracle.seed_batch(b, offs, seeds, k, m2, want_pos=False)
    present = _present(filt, wb["hashes"], n_bits, per)
    read_of = np.repeat(np.arange(n), wb["counts"].astype(np.int64))
    want_hits = np.bincount(read_of[present], minlength=n).astype(np.uint64)
    ctx.h2d(d_in, b)
    d_hits = ctx.malloc(n * 8)
    ctx.memset(d_hits, 0xEE, n * 8)
    tq, found = ctx.seed_bloom_query_ptr(d_in, n, 0 if by_offsets else L, 0, sd, m2, d_f, n_bits, hits=d_hits, offsets=d_o)
    hits = np.zeros(n, np.uint64)
    ctx.d2h(hits, d_hits)
    assert tq == wb["total"]
    assert (hits == want_hits).all(), int((hits != want_hits).sum())
    assert found == int(want_hits.sum())
    # host buffers
    from nthash_amd.capi import NTHIP_HOST_INPUT, NTHIP_HOST_OUTPUT
    hits2 = np.zeros(n, np.uint64)
    tq2, found2 = ctx.seed_bloom_query_ptr(b.ctypes.data, n, 0 if by_offsets else L, 0, sd, m2, d_f, n_bits, hits=hits2.ctypes.data,
                                          flags=NTHIP_HOST_INPUT | NTHIP_HOST_OUTPUT, offsets=offs.ctypes.data if by_offsets else 0)
    assert tq2 == tq and found2 == found and (hits2 == hits).all()
    for p_ in (d_f, d_in, d_hits) + ((d_o,) if d_o else ()):
        ctx.free(p_)
    sd.close()
    ctx.close()


def test_seed_bloom_argument_errors(ctx):
    import nthash_amd
    sd = nthash_amd.Seeds(ctx, [SEED_A], 31)
    d_f, _ = ctx.bloom_new(1 << 16)
    d_in = ctx.malloc(256)
    for bad in (lambda: ctx.seed_bloom_insert_ptr(d_in, 1, 200, 0, sd, 1, 0, 1 << 16),          # NULL filter
                lambda: ctx.seed_bloom_insert_ptr(d_in, 1, 200, 0, sd, 1, d_f + 1, 1 << 16),    # unaligned
                lambda: ctx.seed_bloom_insert_ptr(d_in, 1, 200, 0, sd, 0, d_f, 1 << 16),        # no hash per seed
                lambda: ctx.seed_bloom_query_ptr(d_in, 1, 200, 0, sd, 1, d_f, 0)):
        with pytest.raises(nthash_amd.NtHipError):
            bad()
    assert ctx.seed_bloom_insert_ptr(d_in, 5, 20, 0, sd, 2, d_f, 1 << 16) == 0                # reads shorter than the seeds
    ctx.free(d_in)
    ctx.free(d_f)
    sd.close()


# ---- the sharded query over all-gathered tables ---------------------------------------------------------------------------------
@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0, 0, 0]])
def test_multi_device_query_of_all_gathered_tables(ctx, oracle, devices):
    """nthip_multi_kmer_bloom_insert / _count_insert with NTHIP_MULTI_ALLGATHER leave the merged table on every device;
    nthip_multi_kmer_bloom_query / _count_query then shard a SECOND batch over the devices, every device asks its own copy,
    the answers stay with the reads.  The box has one GPU, so it is listed 2-5 times (separate contexts, threads, tables: the
    code an 8-GPU node runs).  Against one device asking one table about all the reads; uneven shards, an empty one."""
    import nthash_amd
    from nthash_amd import capi
    G = len(devices)
    n, L, k, m = 8000, 150, 31, 2
    nwin = L - k + 1
    a = oracle.synth_reads(5, n, L, 31)
    b = oracle.synth_reads(5 + n // 2, n, L, 31).copy()          # half of A's reads, half new ones
    rng = np.random.default_rng(8)
    _dirty(rng, b, n, L, every=900)
    n_bits, n_cnt = 1 << 22, 1 << 16
    # one device: insert A, ask about B
    d_a, d_b = ctx.malloc(n * L), ctx.malloc(n * L)
    ctx.h2d(d_a, a)
    ctx.h2d(d_b, b)
    d_f, _ = ctx.bloom_new(n_bits)
    d_c = ctx.malloc(n_cnt)
    ctx.memset(d_c, 0, n_cnt)
    ctx.bloom_insert_ptr(d_a, n, L, 0, k, m, d_f, n_bits)
    ctx.count_insert_ptr(d_a, n, L, 0, k, m, d_c, n_cnt)
    d_h, d_e = ctx.malloc(n * 8), ctx.malloc(n * nwin)
    tot1, found1 = ctx.bloom_query_ptr(d_b, n, L, 0, k, m, d_f, n_bits, hits=d_h)
    totc1 = ctx.count_query_ptr(d_b, n, L, 0, k, m, d_c, n_cnt, d_e)
    hits1, est1 = np.zeros(n, np.uint64), np.zeros(n * nwin, np.uint8)
    ctx.d2h(hits1, d_h)
    ctx.d2h(est1, d_e)
    for p_ in (d_a, d_b, d_f, d_c, d_h, d_e):
        ctx.free(p_)
    # the set
    cuts = [0] + sorted(int(x) for x in rng.integers(1, n, G - 1)) + [n]
    if G > 2:
        cuts[2] = cuts[1]
    mm = nthash_amd.Multi(devices)
    cs = [mm.ctx(g) for g in range(G)]
    sh_a, sh_b, flt, cnt, hit, est, owned = [], [], [], [], [], [], []
    for g in range(G):
        r0, r1 = cuts[g], cuts[g + 1]
        da, db = cs[g].malloc(max((r1 - r0) * L, 16)), cs[g].malloc(max((r1 - r0) * L, 16))
        if r1 > r0:
            cs[g].h2d(da, a[r0 * L: r1 * L])
            cs[g].h2d(db, b[r0 * L: r1 * L])
        sh_a.append((da, 0, r1 - r0, L, 0))
        sh_b.append((db, 0, r1 - r0, L, 0))
        f, c_ = cs[g].malloc(n_bits // 8), cs[g].malloc(n_cnt)
        cs[g].memset(f, 0, n_bits // 8)
        cs[g].memset(c_, 0, n_cnt)
        h_, e_ = cs[g].malloc(max((r1 - r0) * 8, 16)), cs[g].malloc(max((r1 - r0) * nwin, 16))
        flt.append(f); cnt.append(c_); hit.append(h_); est.append(e_)
        owned += [(g, da), (g, db), (g, f), (g, c_), (g, h_), (g, e_)]
    mm.bloom_insert(sh_a, k, m, flt, n_bits, capi.NTHIP_MULTI_ALLGATHER)
    mm.count_insert(sh_a, k, m, cnt, n_cnt, capi.NTHIP_MULTI_ALLGATHER)
    tot, found = mm.bloom_query(sh_b, k, m, flt, n_bits, hit)
    assert tot == tot1 and found == found1
    totc = mm.count_query(sh_b, k, m, cnt, n_cnt, est)
    assert totc == totc1
    for g in range(G):
        r0, r1 = cuts[g], cuts[g + 1]
        if r1 == r0:
            continue
        hg, eg = np.zeros(r1 - r0, np.uint64), np.zeros((r1 - r0) * nwin, np.uint8)
        cs[g].d2h(hg, hit[g])
        cs[g].d2h(eg, est[g])
        assert (hg == hits1[r0:r1]).all(), g
        assert (eg == est1[r0 * nwin: r1 * nwin]).all(), g
    assert mm.bloom_query(sh_b, k, m, flt, n_bits) == (tot1, found1)          # no per-read hits asked for
    for g, p_ in owned:
        cs[g].free(p_)
    mm.close()


@pytest.mark.gpu
def test_scratch_limit_bounds_the_kept_buffers_same_filter():
    """nthip_ctx_set_scratch_limit: config 4's seed pair x 3 hashes on 1.5 M x 250 bp (2 G values: 40 GB of
    stream and lists when a round may take what is free) under a 1 GiB limit -- the rounds shrink to fit, the context never
    holds more than the limit, and the filter is the unlimited context's, word for word (as is the filter of rounds cut by
    NTHIP_TUNE_BLOOM_ROUND: tools/seed_insert_rounds_check.py's check); the same for the k-mer insert and the queries' hits"""
    import os
    import nthash_amd
    seeds = ["1010101010101010101010101010101", "1101101101101101011011011011011"]
    n, L, k, n_bits = 1_500_000, 250, 31, 1 << 33
    os.environ["NTHIP_TUNE_BLOOM_ROUND"] = str(300_000_000)
    try:
        c_rounds = nthash_amd.Context(0)
    finally:
        os.environ.pop("NTHIP_TUNE_BLOOM_ROUND", None)
    c_free, c_lim = nthash_amd.Context(0), nthash_amd.Context(0)
    lim = 1 << 30
    c_lim.set_scratch_limit(lim)
    assert c_lim.scratch_info()[1] == lim and c_free.scratch_info()[1] > (16 << 30)
    d_in = c_free.malloc(n * L)
    c_free.synth_reads_ptr(d_in, 0, n, L, 42)
    sums, hits = [], []
    for c in (c_free, c_lim, c_rounds):
        sd = nthash_amd.Seeds(c, seeds, k)
        d_f = c.malloc(n_bits // 8)
        c.memset(d_f, 0, n_bits // 8)
        tot = c.seed_bloom_insert_ptr(d_in, n, L, 0, sd, 3, d_f, n_bits)
        assert tot == n * (L - k + 1)
        kept, _ = c.scratch_info()
        if c is c_lim:
            assert kept <= lim, (kept, lim)
        sums.append(c.checksum_ptr(d_f, n_bits // 64))
        d_h = c.malloc(n * 8)
        tq, found = c.seed_bloom_query_ptr(d_in, n, L, 0, sd, 3, d_f, n_bits, hits=d_h)
        assert tq == found == tot
        if c is c_lim:
            assert c.scratch_info()[0] <= lim
        hs = np.zeros(n, np.uint64)
        c.d2h(hs, d_h)
        hits.append(hs)
        # the k-mer consumers under the same limit
        c.memset(d_f, 0, n_bits // 8)
        tk = c.bloom_insert_ptr(d_in, n, L, 0, k, 3, d_f, n_bits)
        assert tk == n * (L - k + 1)
        sums.append(c.checksum_ptr(d_f, n_bits // 64))
        if c is c_lim:
            assert c.scratch_info()[0] <= lim
        c.free(d_f); c.free(d_h); sd.close()
    assert sums[0:2] == sums[2:4] == sums[4:6], sums
    assert (hits[0] == hits[1]).all() and (hits[0] == hits[2]).all() and (hits[0] == L - k + 1).all()
    # a new limit under what is held releases it at once; 0 restores the default
    kept_free = c_free.scratch_info()[0]
    assert kept_free > lim
    c_free.set_scratch_limit(lim)
    assert c_free.scratch_info()[0] <= lim
    c_free.set_scratch_limit(0)
    assert c_free.scratch_info()[1] > (16 << 30)
    with pytest.raises(nthash_amd.NtHipError):
        c_free.set_scratch_limit(1 << 20)
    c_free.free(d_in)
    for c in (c_free, c_lim, c_rounds):
        c.close()
