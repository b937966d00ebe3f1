"""GPU parity tests (run with -m gpu on an MI355X): every call goes through the
C-ABI (include/nthash_hip.h) and is compared bit-for-bit with
  * the golden fixtures generated from the real reference (tests/golden),
  * the oracle (oracle/nthash_oracle.c) on seeded inputs,
and, at sizes where no oracle run is affordable, through size-independent
properties (strand symmetry, full-care seed == k-mer hash, fast kernel ==
general kernel, shard concatenation, checksums of checksums).
Nothing here reads /root/reference.
"""
import os

import numpy as np
import pytest

from conftest import load_golden
from oracle.pyoracle import concat_reads

pytestmark = pytest.mark.gpu

SEED_A = "1010101010101010101010101010101"
SEED_B = "1101101101101101011011011011011"


def h2i(xs):
    return np.array([int(x, 16) for x in xs], dtype=np.uint64)


def rc_bytes(a):
    tab = np.arange(256, dtype=np.uint8)
    for x, y in zip(b"ACGTacgt", b"TGCAtgca"):
        tab[x] = y
    return tab[a[::-1]]


def test_native_library_is_loaded(ctx):
    """the HIP extension, not a fallback, is what runs"""
    import nthash_amd
    assert nthash_amd.device_count() >= 1
    maps = open("/proc/self/maps").read()
    assert "libnthash_hip.so" in maps


# ---------------------------------------------------------------------------
# golden fixtures (generated from the real reference)
# ---------------------------------------------------------------------------
def test_golden_kmer_cases(ctx):
    for c in load_golden("kmer_cases.json"):
        d, offs = concat_reads(c["reads"])
        r = ctx.kmer_hash(d, c["k"], c["m"], offsets=offs, want_pos=True, want_strands=True)
        assert r["counts"].tolist() == c["counts"], c["reads"]
        assert r["pos"].tolist() == c["pos"]
        assert (r["hashes"].ravel() == h2i(c["hashes"])).all()
        assert (r["fwd"] == h2i(c["fwd"])).all() and (r["rev"] == h2i(c["rev"])).all()


def test_golden_seed_cases(ctx):
    for c in load_golden("seed_cases.json"):
        d, offs = concat_reads(c["reads"])
        r = ctx.seed_hash(d, c["seeds"], c["k"], c["m2"], offsets=offs, want_pos=True)
        assert r["counts"].tolist() == c["counts"], (c["reads"], c["seeds"])
        assert r["pos"].tolist() == c["pos"]
        assert (r["hashes"].ravel() == h2i(c["hashes"])).all()


def test_golden_synth_checksums_device_resident(ctx, oracle):
    """BASELINE config 1 (10k x 150 bp, k=31) and friends, inputs generated ON the
    device by the counter-based generator, full stream compared by checksum and
    head values with what the real reference produced for the same reads."""
    for c in load_golden("synth_checksums.json"):
        n, L, k = c["n_reads"], c["len"], c["k"]
        d_in = ctx.malloc(n * L)
        ctx.synth_reads_ptr(d_in, 0, n, L, c["seed"])
        host = np.zeros(n * L, np.uint8)
        ctx.d2h(host, d_in)
        assert (host == oracle.synth_reads(0, n, L, c["seed"])).all()  # generator parity
        nwin = L - k + 1
        per = c["m"] if c["kind"] == "kmer" else len(c["seeds"]) * c["m2"]
        d_out = ctx.malloc(n * nwin * per * 8)
        if c["kind"] == "kmer":
            tot = ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, c["m"], d_out, n * nwin)
        else:
            import nthash_amd
            sd = nthash_amd.Seeds(ctx, c["seeds"], k)
            tot = ctx.seed_hash_ptr(d_in, 0, n, L, 0, sd, c["m2"], d_out, n * nwin)
        assert tot == c["total"]
        s, x = ctx.checksum_ptr(d_out, tot * per)
        assert format(s, "016x") == c["sum"] and format(x, "016x") == c["xor"]
        head = np.zeros(len(c["head"]), np.uint64)
        ctx.d2h(head, d_out)
        assert (head == h2i(c["head"])).all()
        ctx.free(d_in)
        ctx.free(d_out)


# ---------------------------------------------------------------------------
# oracle on seeded inputs: fixed-length fast kernels
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("n,L,k,m", [
    (10000, 150, 31, 1),   # BASELINE config 1 shape: full bit-exact compare
    (3000, 150, 31, 4),    # multi-hash (config 3 shape)
    (1000, 150, 31, 2), (777, 150, 31, 8),
    (513, 100, 64, 3),     # examples/benchmark.cpp shape (k > 32: d mod 31/33 wrap)
    (300, 151, 21, 3), (257, 97, 17, 5), (255, 64, 33, 1), (100, 31, 31, 1), (64, 40, 3, 2),
    (1, 150, 31, 1), (3, 250, 16, 1), (1000, 250, 47, 1), (2, 500, 32, 7),
])
def test_kmer_fixed_vs_oracle(ctx, oracle, n, L, k, m):
    data = oracle.synth_reads(5, n, L, 1234 + L)
    offs = np.arange(n + 1, dtype=np.uint64) * L
    want = oracle.kmer_batch(data, offs, k, m, want_pos=False)
    got = ctx.kmer_hash(data, k, m, fixed_len=L, n_reads=n)
    assert got["total"] == want["total"] == n * (L - k + 1)
    assert (got["hashes"] == want["hashes"]).all()
    assert (got["counts"] == want["counts"]).all()
    # the N-aware general kernel and the row-per-read kernel must give the same stream
    gen = ctx.kmer_hash(data, k, m, fixed_len=L, n_reads=n, flags=4, want_pos=True, want_strands=True)
    assert (gen["hashes"] == want["hashes"]).all()
    rows = ctx.kmer_hash(data, k, m, fixed_len=L, n_reads=n, flags=8)
    assert (rows["hashes"] == want["hashes"]).all()


@pytest.mark.parametrize("n,L,k,m", [
    (4097, 101, 31, 1),    # 71 windows (prime): ragged last run
    (1000, 151, 25, 2),    # 127 windows (prime), odd value offsets with m = 2
    (999, 100, 64, 3),     # 37 windows, k = 64 (examples/benchmark.cpp shape)
    (333, 251, 31, 1), (777, 149, 31, 1), (500, 76, 31, 1), (1001, 36, 21, 1), (129, 53, 50, 5),
    (65, 1000, 31, 1), (7, 10000, 31, 1), (3, 10007, 64, 2), (1, 5003, 17, 1), (2, 70001, 31, 3),
    (5000, 33, 31, 1), (5000, 31, 31, 8), (63, 47, 40, 7), (64, 48, 40, 1), (200, 150, 32, 1),
])
def test_kmer_general_run_split_shapes_vs_oracle(ctx, oracle, n, L, k, m):
    """shapes whose window count has no convenient divisor, very long reads, odd
    stream offsets: the general run-split kernel (kmer_runs_gen_kernel.hpp)"""
    data = oracle.synth_reads(11, n, L, 4321 + L + k)
    offs = np.arange(n + 1, dtype=np.uint64) * L
    want = oracle.kmer_batch(data, offs, k, m, want_pos=False)
    ctx.set_profiling(True)
    got = ctx.kmer_hash(data, k, m, fixed_len=L, n_reads=n)
    name = ctx.last_kernel_ms()[1]
    ctx.set_profiling(False)
    assert name in ("kmer_runs_gen_kernel", "kmer_runs_kernel"), name
    assert got["total"] == want["total"] == n * (L - k + 1)
    assert (got["hashes"] == want["hashes"]).all()
    assert (got["counts"] == want["counts"]).all()


def test_kmer_headline_kernel_any_k_instantiations_vs_oracle(ctx, oracle):
    """kmer_runs_kernel with k at run time and a compile-time run length (m = 1, 17 <= k <= 32, run length dividing the
    window count): every compiled run length, k at both ends of the table width, short and long slabs (dword tail on
    and off), an unaligned buffer, and a batch with an N (dense pass gives way to the N-aware one)"""
    rng = np.random.default_rng(99)
    for C in (10, 11, 12, 13, 14, 15, 16, 19, 20, 22, 23, 25):
        for k in (17, 21, 25, 31, 32):
            for rpr in (1, 3, 7):
                if k == 31 and C == 15:
                    continue                       # (the k = 31 instantiation)
                L = rpr * C + k - 1
                n = int(rng.integers(65, 400))
                data = oracle.synth_reads(int(rng.integers(0, 100)), n, L, int(rng.integers(0, 1 << 30)))
                offs = np.arange(n + 1, dtype=np.uint64) * L
                want = oracle.kmer_batch(data, offs, k, 1, want_pos=False)
                ctx.set_profiling(True)
                got = ctx.kmer_hash(data, k, 1, fixed_len=L, n_reads=n)
                name = ctx.last_kernel_ms()[1]
                ctx.set_profiling(False)
                # (the plan takes the largest divisor <= 16 of the window count: rpr * C may have a larger one than C)
                assert name in ("kmer_runs_kernel", "kmer_runs_gen_kernel"), name
                if rpr == 1 and "NTHASH_AMD_LIB" not in os.environ:   # (an experiment build may leave these shapes to the general kernel)
                    assert name == "kmer_runs_kernel", (name, C, k)
                assert got["total"] == want["total"] and (got["hashes"] == want["hashes"]).all(), (C, k, rpr, name)
    n, L, k = 3000, 151, 31                        # 121 windows = 11 x 11
    data = oracle.synth_reads(1, n, L, 5).copy()
    offs = np.arange(n + 1, dtype=np.uint64) * L
    d_in = ctx.malloc(n * L + 16)
    d_out = ctx.malloc(n * 121 * 8)
    try:
        for shift in (0, 5):
            ctx.h2d(d_in + shift, data)
            ctx.set_profiling(True)
            tot = ctx.kmer_hash_ptr(d_in + shift, 0, n, L, 0, k, 1, d_out, n * 121)
            assert ctx.last_kernel_ms()[1] == "kmer_runs_kernel" or "NTHASH_AMD_LIB" in os.environ
            ctx.set_profiling(False)
            got = np.zeros(n * 121, np.uint64)
            ctx.d2h(got, d_out)
            assert tot == n * 121 and (got.reshape(-1, 1) == oracle.kmer_batch(data, offs, k, 1, want_pos=False)["hashes"]).all()
    finally:
        ctx.free(d_in)
        ctx.free(d_out)
    data[7 * L + 40] = ord("N")
    want = oracle.kmer_batch(data, offs, k, 1, want_pos=False)
    got = ctx.kmer_hash(data, k, 1, fixed_len=L, n_reads=n)
    assert got["total"] == want["total"] == n * 121 - 31 and (got["hashes"] == want["hashes"]).all()


def test_kmer_general_run_split_unaligned_and_overlapping(ctx, oracle):
    """base pointer off the 16-byte grid; stride < len (a long sequence cut into
    overlapping runs) -- both with a ragged run split"""
    n, L, k = 301, 101, 31
    data = oracle.synth_reads(0, n, L, 19)
    offs = np.arange(n + 1, dtype=np.uint64) * L
    want = oracle.kmer_batch(data, offs, k, 1, want_pos=False)["hashes"].ravel()
    for shift in (1, 5, 15):
        d_in = ctx.malloc(n * L + 64)
        ctx.h2d(d_in + shift, data)
        d_out = ctx.malloc(n * (L - k + 1) * 8)
        tot = ctx.kmer_hash_ptr(d_in + shift, 0, n, L, 0, k, 1, d_out, n * (L - k + 1))
        got = np.zeros(tot, np.uint64)
        ctx.d2h(got, d_out)
        assert (got == want).all()
        ctx.free(d_in)
        ctx.free(d_out)
    # one 50 kb sequence as runs of R = 997 windows: stride R, len R + k - 1
    R, k, m = 997, 25, 2
    n_runs = 50
    seq = oracle.synth_reads(0, 1, R * n_runs + k - 1, 5)
    one = oracle.kmer_batch(seq, np.array([0, seq.size], np.uint64), k, m, want_pos=False)["hashes"]
    got = ctx.kmer_hash(seq, k, m, fixed_len=R + k - 1, n_reads=n_runs, stride=R)
    assert got["total"] == R * n_runs and (got["hashes"] == one).all()


def test_kmer_general_run_split_detects_every_non_base(ctx, oracle):
    """a single non-base byte anywhere (first / last byte of the batch, a read's short
    last run, the last read) must send the batch to the N-aware paths"""
    n, L, k, m = 193, 101, 31, 1
    clean = oracle.synth_reads(0, n, L, 23)
    offs = np.arange(n + 1, dtype=np.uint64) * L
    rng = np.random.default_rng(5)
    spots = [0, 1, 15, 16, L - 1, L, n * L - 1, n * L - 2, (n - 1) * L, 70 + 17 * L, 100 + 64 * L]
    spots += [int(x) for x in rng.integers(0, n * L, 24)]
    for sp in spots:
        data = clean.copy()
        data[sp] = ord("N")
        want = oracle.kmer_batch(data, offs, k, m, want_pos=False)
        got = ctx.kmer_hash(data, k, m, fixed_len=L, n_reads=n)
        assert got["total"] == want["total"] < n * (L - k + 1), sp
        assert (got["hashes"] == want["hashes"]).all(), sp
        assert (got["counts"] == want["counts"]).all(), sp


def test_kmer_fixed_unaligned_base_pointer(ctx, oracle):
    """device buffer that does not start on a 16-byte boundary (edge vectors)"""
    n, L, k = 700, 150, 31
    data = oracle.synth_reads(0, n, L, 9)
    offs = np.arange(n + 1, dtype=np.uint64) * L
    want = oracle.kmer_batch(data, offs, k, 1, want_pos=False)["hashes"]
    for shift in (1, 7, 13):
        d_in = ctx.malloc(n * L + 64)
        ctx.h2d(d_in + shift, data)
        d_out = ctx.malloc(n * (L - k + 1) * 8)
        tot = ctx.kmer_hash_ptr(d_in + shift, 0, n, L, 0, k, 1, d_out, n * (L - k + 1))
        got = np.zeros(tot, np.uint64)
        ctx.d2h(got, d_out)
        assert (got == want.ravel()).all()
        ctx.free(d_in)
        ctx.free(d_out)


def test_kmer_lowercase_and_rna(ctx, oracle):
    n, L, k = 500, 150, 31
    data = oracle.synth_reads(0, n, L, 3)
    lower = data | 0x20
    rna = data.copy()
    rna[rna == ord("T")] = ord("U")
    a = ctx.kmer_hash(data, k, 2, fixed_len=L, n_reads=n)["hashes"]
    assert (ctx.kmer_hash(lower, k, 2, fixed_len=L, n_reads=n)["hashes"] == a).all()
    assert (ctx.kmer_hash(rna, k, 2, fixed_len=L, n_reads=n)["hashes"] == a).all()


def test_kmer_dirty_fixed_len_falls_back_exactly(ctx, oracle):
    """fixed-length batch with N / IUPAC bytes: optimistic kernel must notice and
    the device-side N-aware path must reproduce the reference's skipping"""
    rng = np.random.default_rng(3)
    n, L, k = 2000, 150, 31
    data = oracle.synth_reads(0, n, L, 77).copy()
    bad = rng.choice(n * L, 60, replace=False)
    data[bad] = np.frombuffer(b"NnRY-*", dtype=np.uint8)[rng.integers(0, 6, 60)]
    data[0] = ord("N")
    data[-1] = ord("N")
    offs = np.arange(n + 1, dtype=np.uint64) * L
    want = oracle.kmer_batch(data, offs, k, 3)
    got = ctx.kmer_hash(data, k, 3, fixed_len=L, n_reads=n, want_pos=True)
    assert got["total"] == want["total"] < n * (L - k + 1)
    assert (got["counts"] == want["counts"]).all()
    assert (got["pos"] == want["pos"]).all()
    assert (got["hashes"] == want["hashes"]).all()
    # without pos the optimistic fixed-length kernel runs first, sees the dirt, and falls back
    got = ctx.kmer_hash(data, k, 3, fixed_len=L, n_reads=n)
    assert got["total"] == want["total"] and (got["hashes"] == want["hashes"]).all()
    assert (got["counts"] == want["counts"]).all()


@pytest.mark.parametrize("n,L,k,m,frac", [
    (5000, 150, 31, 1, 0.002), (3000, 150, 31, 4, 0.01), (2000, 150, 31, 1, 0.2),
    (1500, 100, 21, 2, 0.01), (1000, 250, 31, 3, 0.005), (700, 151, 25, 1, 0.01), (513, 64, 33, 1, 0.02),
    (400, 120, 64, 2, 0.003),
    # any shape: prime window counts, k > 50 (validity over more than 64 bases), long reads
    (2000, 101, 31, 1, 0.004), (900, 151, 25, 2, 0.01), (1500, 100, 64, 3, 0.002), (600, 150, 63, 1, 0.003),
    (40, 5003, 31, 1, 0.001), (5, 40001, 57, 2, 0.0005), (3000, 36, 21, 1, 0.01), (700, 33, 31, 1, 0.01),
])
def test_kmer_na_runs_path_vs_oracle(ctx, oracle, n, L, k, m, frac):
    """fixed-length reads sprinkled with non-bases: count pass -> scan -> compact hash pass
    (kmer_runs_na_kernel) must reproduce the reference's emitted set, order, positions and counts"""
    rng = np.random.default_rng(n + L)
    data = oracle.synth_reads(3, n, L, 7 * L + k).copy()
    nbad = max(1, int(frac * n * L))
    where = rng.choice(n * L, nbad, replace=False)
    data[where] = np.frombuffer(b"NnRYKM-*.", dtype=np.uint8)[rng.integers(0, 9, nbad)]
    data[:3] = ord("N")            # first window of the batch
    data[-2:] = ord("n")           # last window of the batch
    data[L * 7:L * 8] = ord("N")   # a read with no valid window at all
    offs = np.arange(n + 1, dtype=np.uint64) * L
    want = oracle.kmer_batch(data, offs, k, m)
    got = ctx.kmer_hash(data, k, m, fixed_len=L, n_reads=n, want_pos=True)
    assert got["total"] == want["total"]
    assert (got["counts"] == want["counts"]).all()
    assert (got["pos"] == want["pos"]).all()
    assert (got["hashes"] == want["hashes"]).all()
    got2 = ctx.kmer_hash(data, k, m, fixed_len=L, n_reads=n)  # optimistic dense kernel first, then N-aware
    assert got2["total"] == want["total"] and (got2["hashes"] == want["hashes"]).all()
    gen = ctx.kmer_hash(data, k, m, fixed_len=L, n_reads=n, flags=4, want_pos=True)
    assert (gen["hashes"] == want["hashes"]).all() and (gen["pos"] == want["pos"]).all()
    # strand hashes (get_forward_hash / get_reverse_hash) from the run-split pass: value selector
    ws = oracle.kmer_batch(data, offs, k, m, want_strands=True)
    gs = ctx.kmer_hash(data, k, m, fixed_len=L, n_reads=n, want_pos=True, want_strands=True)
    assert gs["total"] == ws["total"]
    for key in ("hashes", "pos", "fwd", "rev"):
        assert (gs[key] == ws[key]).all(), key


def test_kmer_ragged_reads_vs_oracle(ctx, oracle):
    rng = np.random.default_rng(21)
    alph = np.frombuffer(b"ACGTacgtUuNnRYKM-*", dtype=np.uint8)
    for k, m in [(31, 1), (5, 3), (64, 2), (17, 4)]:
        reads = []
        for _ in range(400):
            L = int(rng.integers(0, 300))
            if rng.random() < 0.5:
                idx = rng.integers(0, 4, L)
            else:
                idx = np.where(rng.random(L) < 0.95, rng.integers(0, 10, L), rng.integers(10, len(alph), L))
            reads.append(alph[idx].tobytes())
        reads += [b"", b"A", b"ACGT" * 100, b"N" * 50]
        d, offs = concat_reads(reads)
        want = oracle.kmer_batch(d, offs, k, m, want_strands=True)
        got = ctx.kmer_hash(d, k, m, offsets=offs, want_pos=True, want_strands=True)
        assert got["total"] == want["total"]
        for key in ("counts", "pos", "hashes", "fwd", "rev"):
            assert (got[key] == want[key]).all(), (k, m, key)


def test_kmer_ragged_fast_path_vs_oracle(ctx, oracle):
    """variable-length reads through the run-split ragged kernel (no strand outputs requested):
    reads shorter than k, empty reads, long reads spanning many tiles, bursts of tiny reads between
    real ones, non-bases everywhere -- stream, positions and per-read counts must be the reference's"""
    rng = np.random.default_rng(314)
    alph = np.frombuffer(b"ACGTacgtUuNnRYKM-*", dtype=np.uint8)
    for k, m in [(31, 1), (31, 4), (5, 2), (21, 1), (50, 1), (17, 3), (64, 1), (57, 2)]:
        reads = []
        for _ in range(600):
            p = rng.random()
            L = int(rng.integers(0, 40)) if p < 0.15 else int(rng.integers(40, 400)) if p < 0.97 else int(rng.integers(3000, 20000))
            if rng.random() < 0.5:
                idx = rng.integers(0, 4, L)
            else:
                idx = np.where(rng.random(L) < 0.97, rng.integers(0, 10, L), rng.integers(10, len(alph), L))
            reads.append(alph[idx].tobytes())
        reads[5:5] = [b"AC"] * 300            # a burst of reads without any window
        reads += [b"", b"A" * k, b"ACGT" * 2000, b"N" * 200, b"ACGTN" * 50]
        d, offs = concat_reads(reads)
        want = oracle.kmer_batch(d, offs, k, m)
        got = ctx.kmer_hash(d, k, m, offsets=offs, want_pos=True)
        assert got["total"] == want["total"], (k, m)
        for key in ("counts", "pos", "hashes"):
            assert (got[key] == want[key]).all(), (k, m, key)
        got = ctx.kmer_hash(d, k, m, offsets=offs)          # without positions
        assert (got["hashes"] == want["hashes"]).all() and (got["counts"] == want["counts"]).all()
        gen = ctx.kmer_hash(d, k, m, offsets=offs, flags=4)  # the lane-per-read kernel agrees
        assert (gen["hashes"] == want["hashes"]).all()
    # degenerate batches
    for reads in ([b""], [b"AC", b"A"], [b"ACGTACGTAC"] * 3, [b"N" * 100]):
        d, offs = concat_reads(reads)
        want = oracle.kmer_batch(d, offs, 5, 2)
        got = ctx.kmer_hash(d, 5, 2, offsets=offs, want_pos=True)
        assert got["total"] == want["total"] and (got["counts"] == want["counts"]).all()
        assert (got["hashes"] == want["hashes"]).all()


def test_kmer_whole_read_tiles_vs_oracle(oracle):
    """variable-length SHORT reads on kmer_reads_kernel (tiles of whole reads; clean reads rolled with an overlapping
    last run, reads with a non-base on the wave-per-read kernel that fills the holes they leave): read counts around
    the tile size, reads with fewer windows than a run, whole neighbourhoods of reads with N's (holes larger than the
    output tile's slack), every table width and the any-k Horner start, positions, strands, several hashes per
    k-mer -- against the oracle and against round 1's kmer_ragged_kernel (NTHIP_TUNE_NO_KMER_READS=1), and the spans
    entry (sequence lines with other bytes between them)"""
    import os
    import nthash_amd
    rng = np.random.default_rng(2024)
    new = nthash_amd.Context(0)
    os.environ["NTHIP_TUNE_NO_KMER_READS"] = "1"
    try:
        old = nthash_amd.Context(0)
    finally:
        os.environ.pop("NTHIP_TUNE_NO_KMER_READS", None)
    alph = np.frombuffer(b"ACGTacgtUuNnRYKM-*", dtype=np.uint8)

    def make_reads(n, lo, hi, p_dirty, cluster=False):
        reads = []
        for i in range(n):
            L = int(rng.integers(lo, hi + 1))
            dirty = (rng.random() < p_dirty) or (cluster and 40 <= i < 75)
            idx = np.where(rng.random(L) < 0.96, rng.integers(0, 10, L), rng.integers(10, len(alph), L)) if dirty \
                else rng.integers(0, 10, L)
            reads.append(alph[idx].tobytes())
        return reads

    cases = []
    for n in (1, 2, 31, 32, 33, 63, 64, 65, 1000):
        cases.append((n, 60, 151, 0.02, False, 31, 1))
    cases += [(700, 0, 60, 0.1, False, 31, 1),        # most reads have fewer windows than a run, or none
              (700, 100, 150, 0.0, True, 31, 1),      # 35 consecutive reads with non-bases: holes beyond the tile's slack
              (500, 100, 150, 0.3, False, 31, 2),
              (600, 20, 300, 0.05, False, 3, 1), (600, 20, 300, 0.05, False, 16, 3), (600, 20, 300, 0.05, False, 17, 1),
              (600, 40, 300, 0.05, False, 32, 4), (600, 40, 300, 0.05, False, 33, 1), (400, 70, 400, 0.05, False, 64, 2),
              (300, 120, 500, 0.05, False, 100, 1),   # any-k start (NW = 0)
              (200, 1500, 2048, 0.02, False, 31, 1)]  # the longest reads this path takes
    for (n, lo, hi, p_dirty, cluster, k, m) in cases:
        reads = make_reads(n, lo, hi, p_dirty, cluster)
        d, offs = concat_reads(reads)
        want = oracle.kmer_batch(d, offs, k, m, want_strands=True)
        new.set_profiling(True)
        got = new.kmer_hash(d, k, m, offsets=offs, want_pos=True)
        name = new.last_kernel_ms()[1]
        new.set_profiling(False)
        if want["total"] and max(len(r) for r in reads) >= k:
            assert name == "kmer_reads_kernel", (name, n, k)
        assert got["total"] == want["total"], (n, lo, hi, k, m)
        for key in ("counts", "pos", "hashes"):
            assert (got[key] == want[key]).all(), (n, lo, hi, k, m, key)
        got = new.kmer_hash(d, k, m, offsets=offs, want_strands=True)            # no positions; strands
        for key in ("counts", "hashes", "fwd", "rev"):
            assert (got[key] == want[key]).all(), (n, lo, hi, k, m, key)
        ref = old.kmer_hash(d, k, m, offsets=offs, want_pos=True)               # round 1's kernel agrees
        assert (ref["hashes"] == want["hashes"]).all() and (ref["pos"] == want["pos"]).all()
    # spans: reads with other bytes between them (what the FASTQ indexer produces), device-resident
    reads = make_reads(777, 50, 151, 0.03)
    buf = bytearray(b"@")
    starts, ends = [], []
    for r in reads:
        buf += b"hdr " + bytes(rng.integers(33, 127, int(rng.integers(0, 40))).astype(np.uint8)) + b"\n"
        starts.append(len(buf)); buf += r; ends.append(len(buf))
        buf += b"\n+\n" + bytes(rng.integers(33, 127, len(r)).astype(np.uint8)) + b"\n@"
    raw = np.frombuffer(bytes(buf), dtype=np.uint8)
    d, offs = concat_reads(reads)
    for k, m in ((31, 1), (21, 2)):
        want = oracle.kmer_batch(d, offs, k, m)
        cap = max(1, want["total"])
        d_buf = new.malloc(raw.size + 16); d_s = new.malloc(8 * len(reads)); d_e = new.malloc(8 * len(reads))
        d_h = new.malloc(cap * m * 8); d_c = new.malloc(8 * len(reads)); d_p = new.malloc(4 * cap)
        try:
            new.h2d(d_buf, raw); new.h2d(d_s, np.array(starts, np.uint64)); new.h2d(d_e, np.array(ends, np.uint64))
            new.set_profiling(True)
            tot = new.kmer_hash_spans_ptr(d_buf, raw.size, d_s, d_e, len(reads), k, m, d_h, cap, counts=d_c, pos=d_p)
            name = new.last_kernel_ms()[1]
            new.set_profiling(False)
            assert name == "kmer_reads_kernel" and tot == want["total"]
            h = np.zeros(cap * m, np.uint64); cts = np.zeros(len(reads), np.uint64); ps = np.zeros(cap, np.uint32)
            new.d2h(h, d_h); new.d2h(cts, d_c); new.d2h(ps, d_p)
            assert (h[: tot * m].reshape(-1, m) == want["hashes"]).all() and (cts == want["counts"]).all()
            assert (ps[:tot] == want["pos"]).all()
            # the same spans in REVERSE order: not this path's (tiles are contiguous slabs) -- the round-1 kernel takes them
            rev = list(range(len(reads)))[::-1]
            d2, offs2 = concat_reads([reads[i] for i in rev])
            want2 = oracle.kmer_batch(d2, offs2, k, m)
            new.h2d(d_s, np.array([starts[i] for i in rev], np.uint64)); new.h2d(d_e, np.array([ends[i] for i in rev], np.uint64))
            new.set_profiling(True)
            tot2 = new.kmer_hash_spans_ptr(d_buf, raw.size, d_s, d_e, len(reads), k, m, d_h, cap, counts=d_c, pos=d_p)
            assert new.last_kernel_ms()[1] == "kmer_ragged_kernel"
            new.set_profiling(False)
            new.d2h(h, d_h); new.d2h(cts, d_c)
            assert tot2 == want2["total"] and (h[: tot2 * m].reshape(-1, m) == want2["hashes"]).all()
            assert (cts == want2["counts"]).all()
            new.h2d(d_s, np.array(starts, np.uint64)); new.h2d(d_e, np.array(ends, np.uint64))
        finally:
            for ptr in (d_buf, d_s, d_e, d_h, d_c, d_p):
                new.free(ptr)
    new.close()
    old.close()


def test_kmer_long_reads_are_segmented(ctx, oracle):
    """reads far longer than a segment (1024 windows) next to short and empty ones: the general
    kernel cuts them into segments that restart the roll; stream, positions and counts must not change"""
    rng = np.random.default_rng(5)
    alph = np.frombuffer(b"ACGTACGTACGTNacgtn", dtype=np.uint8)
    reads = [alph[rng.integers(0, len(alph), L)].tobytes()
             for L in (300_000, 40, 0, 5000, 1023 + 30, 1024 + 30, 1025 + 30, 150, 70_001)]
    reads.append(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 200_000)].tobytes())
    d, offs = concat_reads(reads)
    for k, m in ((31, 2), (64, 1), (5, 1)):
        want = oracle.kmer_batch(d, offs, k, m, want_strands=True)
        got = ctx.kmer_hash(d, k, m, offsets=offs, want_pos=True, want_strands=True)
        assert got["total"] == want["total"]
        for key in ("counts", "pos", "hashes", "fwd", "rev"):
            assert (got[key] == want[key]).all(), (k, m, key)
    # one long fixed-length "read" (no offsets) goes the same way
    one = np.frombuffer(reads[-1], dtype=np.uint8)
    want = oracle.kmer_batch(one, np.array([0, len(one)], dtype=np.uint64), 31, 1)
    got = ctx.kmer_hash(one, 31, 1, fixed_len=len(one), n_reads=1, want_pos=True, want_strands=True)
    assert (got["hashes"] == want["hashes"]).all() and (got["pos"] == want["pos"]).all()


def test_kmer_long_sequence_as_overlapping_runs(ctx, oracle):
    """one long sequence hashed as runs of R windows overlapping by k-1 bases
    (stride < fixed_len) == the sequence hashed as a single read"""
    rng = np.random.default_rng(8)
    k, R = 31, 120
    n_runs = 1500
    N = n_runs * R + k - 1
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, N)]
    offs = np.array([0, N], dtype=np.uint64)
    want = oracle.kmer_batch(seq, offs, k, 1, want_pos=False)["hashes"]
    got = ctx.kmer_hash(seq, k, 1, fixed_len=R + k - 1, stride=R, n_reads=n_runs)
    assert got["total"] == n_runs * R == len(want)
    assert (got["hashes"] == want).all()


def test_kmer_argument_errors(ctx):
    import nthash_amd
    from nthash_amd import capi
    data = np.frombuffer(b"ACGTACGTAC", dtype=np.uint8)
    with pytest.raises(nthash_amd.NtHipError) as e:
        ctx.kmer_hash(data, 0, 1, fixed_len=10, n_reads=1)  # k == 0: src/kmer.cpp:212
    assert e.value.code == capi.NTHIP_ERR_ARG
    with pytest.raises(nthash_amd.NtHipError) as e:
        ctx.kmer_hash(data, 2, 1, fixed_len=10, n_reads=1)
    assert e.value.code == capi.NTHIP_ERR_UNSUPPORTED
    # reads shorter than k emit nothing (the iterator would exit(1), src/kmer.cpp:215)
    r = ctx.kmer_hash(data, 11, 1, fixed_len=10, n_reads=1)
    assert r["total"] == 0 and r["counts"].tolist() == [0]
    # capacity too small: error code + required size
    d_in = ctx.malloc(1000)
    d_out = ctx.malloc(8 * 10)
    ctx.h2d(d_in, np.frombuffer(b"ACGT" * 250, dtype=np.uint8))
    with pytest.raises(nthash_amd.NtHipError) as e:
        ctx.kmer_hash_ptr(d_in, 0, 10, 100, 0, 31, 1, d_out, 10)
    assert e.value.code == capi.NTHIP_ERR_CAPACITY and e.value.total == 10 * 70
    ctx.free(d_in)
    ctx.free(d_out)
    # seed length != k: src/seed.cpp:90-95
    with pytest.raises(nthash_amd.NtHipError) as e:
        nthash_amd.Seeds(ctx, ["110011"], 5)
    assert e.value.code == capi.NTHIP_ERR_ARG


# ---------------------------------------------------------------------------
# spaced seeds
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("n,L,seeds,m2", [
    (4000, 250, [SEED_A, SEED_B], 3),      # BASELINE config 4 shape
    (1000, 150, [SEED_A], 1),
    (600, 100, ["11100111"], 3),
    (500, 120, ["111110000000011111", "111111100001111111"], 2),
    (300, 150, ["1" * 31], 4),
    (300, 130, ["1111111111111110111111111111111"], 2),   # ignore-path description
    (200, 200, [("1101" * 8) + ("1101" * 8)[::-1]], 2),          # k = 64 (NW = 4)
    (129, 150, ["10" * 24 + "01" * 24], 1),    # k = 96 > 64: general kernel
])
def test_seed_fixed_vs_oracle(ctx, oracle, n, L, seeds, m2):
    k = len(seeds[0])
    data = oracle.synth_reads(11, n, L, 99 + L)
    offs = np.arange(n + 1, dtype=np.uint64) * L
    want = oracle.seed_batch(data, offs, seeds, k, m2, want_pos=False)
    got = ctx.seed_hash(data, seeds, k, m2, fixed_len=L, n_reads=n)
    assert got["total"] == want["total"] == n * (L - k + 1)
    assert (got["hashes"] == want["hashes"]).all()
    gen = ctx.seed_hash(data, seeds, k, m2, fixed_len=L, n_reads=n, flags=4, want_pos=True)
    assert (gen["hashes"] == want["hashes"]).all()


def test_seed_wave_tile_kernel_shapes_vs_oracle(ctx, oracle):
    """the dense spaced-seed kernel (one wave per tile of reads, seed_wtile_kernel): read counts that are no multiple
    of its tile, strides below the read length (overlapping rows), unaligned buffers, short and long k, one to
    three seeds, and the same stream from the block-tile kernel it replaced"""
    import os
    import nthash_amd
    rng = np.random.default_rng(4242)
    for (n, L, stride, seeds, m2) in [(1, 250, 0, [SEED_A, SEED_B], 3), (17, 250, 0, [SEED_A, SEED_B], 3),
                                      (5003, 151, 0, [SEED_A, SEED_B], 3),   # 121 windows: tiles that start between lines
                                      (2501, 101, 0, [SEED_B], 3), (1999, 77, 0, [SEED_A[:21]], 1),
                                      (1000, 250, 0, [SEED_A, SEED_B], 3), (4099, 100, 0, ["1101011"], 2),
                                      (3001, 151, 0, ["1" * 20 + "0" * 9 + "1" * 20], 1),
                                      (777, 64, 0, ["1" * 64], 4), (513, 300, 41, ["10101", "11011", "01110"], 2),
                                      (2500, 1000, 0, ["110" * 10 + "1"], 1), (300, 5000, 0, [SEED_A], 2)]:
        k = len(seeds[0])
        st_ = stride or L
        total = (n - 1) * st_ + L
        raw = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, total + 3)]
        for shift in (0, 3):    # unaligned start of the buffer
            data = raw[shift: shift + total]
            reads = [data[i * st_: i * st_ + L].tobytes() for i in range(n)]
            d, offs = concat_reads(reads)
            want = oracle.seed_batch(d, offs, seeds, k, m2, want_pos=False)
            d_in = ctx.malloc(total + 16)
            d_out = ctx.malloc(max(1, want["total"]) * len(seeds) * m2 * 8)
            try:
                ctx.h2d(d_in + shift, np.ascontiguousarray(data))
                sd = nthash_amd.Seeds(ctx, seeds, k)
                ctx.set_profiling(True)
                tot = ctx.seed_hash_ptr(d_in + shift, 0, n, L, stride, sd, m2, d_out, want["total"])
                name = ctx.last_kernel_ms()[1]
                ctx.set_profiling(False)
                assert tot == want["total"]
                got = np.zeros(want["hashes"].size, np.uint64)
                ctx.d2h(got, d_out)
                assert (got.reshape(want["hashes"].shape) == want["hashes"]).all(), (n, L, stride, seeds, m2, shift, name)
                if L == 250:
                    assert name == "seed_wtile_kernel", (name, n, L)
            finally:
                ctx.free(d_in)
                ctx.free(d_out)


def test_seed_rotated_slot_tables_vs_oracle(oracle):
    """the rotated-slot layout of the byte tables in LDS (seed_wtile_kernel<4, seeds, m2>: one or two seeds, k <= 32,
    m2 <= 4): every (seeds, m2) instantiation, k from 4 to 32 (tables past ceil(k/4) are zero), read counts that
    leave a ragged last tile and a ragged last 64-window group -- against the oracle and against the plain
    [table][entry] layout (NTHIP_TUNE_NO_SEED_ROT=1), which also covers m2 = 5 and three seeds staying on it"""
    import os
    import nthash_amd
    rng = np.random.default_rng(777)
    plain_env = "NTHIP_TUNE_NO_SEED_ROT"
    os.environ.pop(plain_env, None)
    rot = nthash_amd.Context(0)
    os.environ[plain_env] = "1"
    try:
        plain = nthash_amd.Context(0)
    finally:
        os.environ.pop(plain_env, None)

    def mask(k, density):
        m = (rng.random(k) < density).astype(int)
        m[0] = m[-1] = 1
        return "".join(str(int(x)) for x in m)

    cases = []
    for k in (4, 7, 8, 9, 16, 17, 24, 25, 31, 32):
        for n_seeds in (1, 2):
            cases.append((int(rng.integers(1, 400)), int(rng.integers(k, 300)), [mask(k, 0.6) for _ in range(n_seeds)],
                          int(rng.integers(1, 5))))
    for m2 in (1, 2, 3, 4, 5):
        cases.append((1237, 250, [SEED_A, SEED_B], m2))
        cases.append((641, 150, [SEED_B], m2))
    cases.append((500, 250, [SEED_A, SEED_B, SEED_A[::-1]], 2))
    for (n, L, seeds, m2) in cases:
        k = len(seeds[0])
        data = np.frombuffer(b"ACGTacgtUu", dtype=np.uint8)[rng.integers(0, 10, n * L)]
        offs = np.arange(n + 1, dtype=np.uint64) * L
        want = oracle.seed_batch(data, offs, seeds, k, m2, want_pos=False)
        for c in (rot, plain):
            got = c.seed_hash(data, seeds, k, m2, fixed_len=L, n_reads=n)
            assert got["total"] == want["total"] == n * (L - k + 1)
            assert (got["hashes"] == want["hashes"]).all(), (n, L, seeds, m2, c is rot)
    rot.close()
    plain.close()


def test_seed_dirty_and_ragged_vs_oracle(ctx, oracle):
    rng = np.random.default_rng(33)
    alph = np.frombuffer(b"ACGTacgtUuNnRYKMSW-*\x00", dtype=np.uint8)
    for seeds, m2 in [([SEED_A, SEED_B], 3), (["110011"], 2), (["1111111111111110111111111111111"], 2),
                      (["10101", "11011", "01110"], 1)]:
        k = len(seeds[0])
        reads = []
        for _ in range(300):
            L = int(rng.integers(0, 260))
            p = rng.random()
            if p < 0.4:
                idx = rng.integers(0, 4, L)
            elif p < 0.9:
                idx = np.where(rng.random(L) < 0.95, rng.integers(0, 10, L), rng.integers(10, len(alph) - 1, L))
            else:
                idx = np.where(rng.random(L) < 0.9, rng.integers(0, 10, L), rng.integers(10, len(alph), L))
            reads.append(alph[idx].tobytes())
        d, offs = concat_reads(reads)
        want = oracle.seed_batch(d, offs, seeds, k, m2)
        got = ctx.seed_hash(d, seeds, k, m2, offsets=offs, want_pos=True)
        assert got["total"] == want["total"]
        for key in ("counts", "pos", "hashes"):
            assert (got[key] == want[key]).all(), (seeds, key)
    # fixed-length but dirty: optimistic kernel -> exact fallback
    n, L = 1500, 250
    data = oracle.synth_reads(0, n, L, 5).copy()
    data[rng.choice(n * L, 40, replace=False)] = ord("N")
    offs = np.arange(n + 1, dtype=np.uint64) * L
    want = oracle.seed_batch(data, offs, [SEED_A, SEED_B], 31, 3)
    got = ctx.seed_hash(data, [SEED_A, SEED_B], 31, 3, fixed_len=L, n_reads=n, want_pos=True)
    assert got["total"] == want["total"]
    assert (got["pos"] == want["pos"]).all() and (got["hashes"] == want["hashes"]).all()


def test_seed_whole_read_tiles_vs_oracle(ctx, oracle):
    """SeedNtHash on variable-length short reads: clean reads on seed_rtile_kernel (tiles of whole reads, rotated-slot
    or plain tables), reads with a non-base on the wave-per-read kernel into the holes they leave -- read counts around
    the tile size, tiny reads, neighbourhoods of dirty reads, one to three seeds, k up to 64, positions, and the spans
    entry; against the oracle's reference state machine"""
    import nthash_amd
    rng = np.random.default_rng(555)
    alph = np.frombuffer(b"ACGTacgtUuNnRYKM-*\x00", dtype=np.uint8)

    def mask(k, density):
        m_ = (rng.random(k) < density).astype(int)
        m_[0] = m_[-1] = 1
        return "".join(str(int(x)) for x in m_)

    def make_reads(n, lo, hi, p_dirty, cluster=False):
        reads = []
        for i in range(n):
            L = int(rng.integers(lo, hi + 1))
            dirty = (rng.random() < p_dirty) or (cluster and 20 <= i < 45)
            idx = np.where(rng.random(L) < 0.96, rng.integers(0, 10, L), rng.integers(10, len(alph), L)) if dirty \
                else rng.integers(0, 10, L)
            reads.append(alph[idx].tobytes())
        return reads

    cases = [(n, 60, 151, 0.03, False, [SEED_A, SEED_B], 3) for n in (1, 15, 16, 17, 33, 500)]
    cases += [(400, 0, 50, 0.1, False, [SEED_A], 2), (400, 100, 150, 0.0, True, [SEED_A, SEED_B], 3),
              (300, 100, 250, 0.3, False, [SEED_B], 4), (300, 20, 300, 0.05, False, [mask(5, 0.7)], 1),
              (300, 20, 300, 0.05, False, [mask(17, 0.6), mask(17, 0.8), mask(17, 0.5)], 2),
              (300, 40, 300, 0.05, False, [mask(32, 0.6), mask(32, 0.9)], 5),
              (200, 70, 400, 0.05, False, [mask(47, 0.6)], 2), (200, 70, 400, 0.05, False, [mask(64, 0.7), mask(64, 0.4)], 1),
              (100, 1000, 2048, 0.02, False, [SEED_A, SEED_B], 3)]
    for (n, lo, hi, p_dirty, cluster, seeds, m2) in cases:
        k = len(seeds[0])
        reads = make_reads(n, lo, hi, p_dirty, cluster)
        d, offs = concat_reads(reads)
        want = oracle.seed_batch(d, offs, seeds, k, m2)
        ctx.set_profiling(True)
        got = ctx.seed_hash(d, seeds, k, m2, offsets=offs, want_pos=True)
        name = ctx.last_kernel_ms()[1]
        ctx.set_profiling(False)
        if max(len(r) for r in reads) >= k:
            assert name == "seed_rtile_kernel", (name, n, k)
        assert got["total"] == want["total"], (n, lo, hi, seeds, m2)
        for key in ("counts", "pos", "hashes"):
            assert (got[key] == want[key]).all(), (n, lo, hi, seeds, m2, key)
        got = ctx.seed_hash(d, seeds, k, m2, offsets=offs)
        assert (got["hashes"] == want["hashes"]).all() and (got["counts"] == want["counts"]).all()
    # spans with other bytes between the reads
    reads = make_reads(333, 50, 151, 0.03)
    buf = bytearray(b"@")
    starts, ends = [], []
    for r in reads:
        buf += b"hdr\n"; starts.append(len(buf)); buf += r; ends.append(len(buf))
        buf += b"\n+\n" + bytes(rng.integers(33, 127, len(r)).astype(np.uint8)) + b"\n@"
    raw = np.frombuffer(bytes(buf), dtype=np.uint8)
    d, offs = concat_reads(reads)
    want = oracle.seed_batch(d, offs, [SEED_A, SEED_B], 31, 3)
    cap = max(1, want["total"])
    sd = nthash_amd.Seeds(ctx, [SEED_A, SEED_B], 31)
    d_buf = ctx.malloc(raw.size + 16); d_s = ctx.malloc(8 * len(reads)); d_e = ctx.malloc(8 * len(reads))
    d_h = ctx.malloc(cap * 6 * 8); d_c = ctx.malloc(8 * len(reads)); d_p = ctx.malloc(4 * cap)
    try:
        ctx.h2d(d_buf, raw); ctx.h2d(d_s, np.array(starts, np.uint64)); ctx.h2d(d_e, np.array(ends, np.uint64))
        tot = ctx.seed_hash_spans_ptr(d_buf, raw.size, d_s, d_e, len(reads), sd, 3, d_h, cap, counts=d_c, pos=d_p)
        assert tot == want["total"]
        h = np.zeros(cap * 6, np.uint64); cts = np.zeros(len(reads), np.uint64); ps = np.zeros(cap, np.uint32)
        ctx.d2h(h, d_h); ctx.d2h(cts, d_c); ctx.d2h(ps, d_p)
        assert (h[: tot * 6].reshape(-1, 6) == want["hashes"]).all() and (cts == want["counts"]).all()
        assert (ps[:tot] == want["pos"]).all()
    finally:
        for ptr in (d_buf, d_s, d_e, d_h, d_c, d_p):
            ctx.free(ptr)
        sd.close()


def test_seed_asymmetric_flag(ctx):
    import nthash_amd
    assert nthash_amd.Seeds(ctx, ["1101"], 4).asymmetric       # src/seed.cpp:96-102 warns
    assert not nthash_amd.Seeds(ctx, ["1001"], 4).asymmetric


# ---------------------------------------------------------------------------
# properties at sizes the oracle does not cover
# ---------------------------------------------------------------------------
def test_properties_at_scale(ctx, oracle):
    """4M x 150 bp on the device (480M k-mers; not the 100M-read config, but big
    enough to cross every tile/grid-stride boundary many times)."""
    import nthash_amd
    n, L, k = 4_000_000, 150, 31
    nwin = L - k + 1
    d_in = ctx.malloc(n * L)
    ctx.synth_reads_ptr(d_in, 0, n, L, 42)
    d_out = ctx.malloc(n * nwin * 8)
    assert ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, 1, d_out, n * nwin) == n * nwin
    s_all, x_all = ctx.checksum_ptr(d_out, n * nwin)
    # (1) shard property: hashing [0,n/2) and [n/2,n) separately gives the same stream
    half = n // 2
    d_o2 = ctx.malloc(half * nwin * 8)
    s2 = x2 = 0
    for part in range(2):
        assert ctx.kmer_hash_ptr(d_in + part * half * L, 0, half, L, 0, k, 1, d_o2, half * nwin) == half * nwin
        s, x = ctx.checksum_ptr(d_o2, half * nwin)
        s2 = (s2 + s) & (2**64 - 1)
        x2 ^= x
    assert (s2, x2) == (s_all, x_all)
    # (2) spot-check reads against the oracle (first, last, tile edges)
    for r0 in (0, 255, 256, 257, 1_000_003, n - 1):
        got = np.zeros(nwin, np.uint64)
        ctx.d2h(got, d_out + r0 * nwin * 8)
        data = oracle.synth_reads(r0, 1, L, 42)
        want = oracle.kmer_batch(data, np.array([0, L], dtype=np.uint64), k, 1, want_pos=False)["hashes"].ravel()
        assert (got == want).all(), r0
    # (3) full-care spaced seed == k-mer hash (tests/tests.cpp:447-463) on 1M reads
    n3 = 1_000_000
    sd = nthash_amd.Seeds(ctx, ["1" * k], k)
    assert ctx.seed_hash_ptr(d_in, 0, n3, L, 0, sd, 1, d_o2, n3 * nwin) == n3 * nwin
    s3, x3 = ctx.checksum_ptr(d_o2, n3 * nwin)
    assert (s3, x3) == ctx.checksum_ptr(d_out, n3 * nwin)
    # (4) multi-hash: h[0] of the m=4 stream is the m=1 stream
    n4 = 200_000
    d_o4 = ctx.malloc(n4 * nwin * 4 * 8)
    assert ctx.kmer_hash_ptr(d_in, 0, n4, L, 0, k, 4, d_o4, n4 * nwin) == n4 * nwin
    a = np.zeros(n4 * nwin * 4, np.uint64)
    ctx.d2h(a, d_o4)
    b = np.zeros(n4 * nwin, np.uint64)
    ctx.d2h(b, d_out)
    a = a.reshape(-1, 4)
    assert (a[:, 0] == b).all()
    mult = [(i ^ (31 * 0x90b45d39fb6da1fa)) & (2**64 - 1) for i in range(4)]
    for i in (1, 2, 3):
        t = b * np.uint64(mult[i])
        assert (a[:, i] == (t ^ (t >> np.uint64(27)))).all()
    for p in (d_in, d_out, d_o2, d_o4):
        ctx.free(p)


def test_full_size_config2_properties(ctx, oracle):
    """BASELINE.json configs[1] at FULL size (100 M x 150 bp, k=31, m=1: 15 GB in, 96 GB out):
    no oracle run is affordable, so parity rests on size-independent properties --
    (1) hashing the two halves separately gives the same checksum-of-checksums as the
    whole batch, (2) reads sampled across the batch (tile, block-range and batch edges)
    are bit-exact against the oracle, (3) the m=4 stream's h[0] column is the m=1 stream."""
    n, L, k = 100_000_000, 150, 31
    nwin = L - k + 1
    try:
        d_in = ctx.malloc(n * L)
        d_out = ctx.malloc(n * nwin * 8)
    except Exception as e:  # a smaller GPU: the 4 M-read test above still runs
        pytest.skip(f"not enough device memory for the full-size config: {e}")
    ctx.synth_reads_ptr(d_in, 0, n, L, 42)
    assert ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, 1, d_out, n * nwin) == n * nwin
    s_all, x_all = ctx.checksum_ptr(d_out, n * nwin)
    rng = np.random.default_rng(0)
    sample = [0, 1, 255, 256, 4095, 4096, n // 2 - 1, n // 2, n - 2, n - 1] + \
        [int(x) for x in rng.integers(0, n, 40)]
    for r0 in sample:
        got = np.zeros(nwin, np.uint64)
        ctx.d2h(got, d_out + r0 * nwin * 8)
        data = oracle.synth_reads(r0, 1, L, 42)
        want = oracle.kmer_batch(data, np.array([0, L], dtype=np.uint64), k, 1, want_pos=False)["hashes"].ravel()
        assert (got == want).all(), r0
    half = n // 2
    s2 = x2 = 0
    for part in range(2):  # reuse the first half of d_out as the destination
        assert ctx.kmer_hash_ptr(d_in + part * half * L, 0, half, L, 0, k, 1, d_out, half * nwin) == half * nwin
        s, x = ctx.checksum_ptr(d_out, half * nwin)
        s2 = (s2 + s) & (2**64 - 1)
        x2 ^= x
    assert (s2, x2) == (s_all, x_all)
    # d_out now holds the m=1 stream of the second half; hash its first 12.5 M reads with m=4
    n4 = 12_500_000
    d_o4 = ctx.malloc(n4 * nwin * 4 * 8)
    assert ctx.kmer_hash_ptr(d_in + half * L, 0, n4, L, 0, k, 4, d_o4, n4 * nwin) == n4 * nwin
    a = np.zeros(1_000_000 * 4, np.uint64)
    b = np.zeros(1_000_000, np.uint64)
    off = (n4 * nwin - 1_000_000)
    ctx.d2h(a, d_o4 + off * 32)
    ctx.d2h(b, d_out + off * 8)
    assert (a.reshape(-1, 4)[:, 0] == b).all()
    for p_ in (d_in, d_out, d_o4):
        ctx.free(p_)


def test_strand_symmetry_on_device(ctx, oracle):
    """canonical hashing (tests/tests.cpp:119-133): a read and its reverse
    complement give mirrored streams"""
    n, L, k = 2000, 150, 31
    data = oracle.synth_reads(0, n, L, 17).reshape(n, L)
    rcd = np.stack([rc_bytes(r) for r in data])
    a = ctx.kmer_hash(data.ravel(), k, 3, fixed_len=L, n_reads=n)["hashes"].reshape(n, L - k + 1, 3)
    b = ctx.kmer_hash(rcd.ravel(), k, 3, fixed_len=L, n_reads=n)["hashes"].reshape(n, L - k + 1, 3)
    assert (a == b[:, ::-1, :]).all()
    sa = ctx.seed_hash(data.ravel(), [SEED_A, SEED_B], k, 2, fixed_len=L, n_reads=n)["hashes"]
    sb = ctx.seed_hash(rcd.ravel(), [SEED_A, SEED_B], k, 2, fixed_len=L, n_reads=n)["hashes"]
    assert (sa.reshape(n, L - k + 1, 4) == sb.reshape(n, L - k + 1, 4)[:, ::-1, :]).all()


# ---------------------------------------------------------------------------
# randomised shapes: every dispatch corner (run-split / row / N-aware / general
# kernels, odd strides, unaligned bases, tiny and prime window counts)
# ---------------------------------------------------------------------------
def test_fuzz_fixed_length_shapes(ctx, oracle):
    rng = np.random.default_rng(2026)
    for it in range(120):
        k = int(rng.choice([3, 4, 5, 8, 15, 16, 17, 21, 31, 32, 33, 48, 63, 64, 65, 100]))
        L = int(k + rng.choice([0, 1, 2, 3, 6, 11, 14, 15, 29, 30, 59, 60, 96, 119, 120, 127, 219]))
        n = int(rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 1000, 1537]))
        m = int(rng.choice([1, 1, 2, 3, 4, 8]))
        dirty = rng.random() < 0.4
        data = oracle.synth_reads(int(rng.integers(0, 1000)), n, L, int(rng.integers(0, 1 << 30))).copy()
        if dirty:
            nb = max(1, (n * L) // 300)
            data[rng.choice(n * L, nb, replace=False)] = ord("N")
        offs = np.arange(n + 1, dtype=np.uint64) * L
        want = oracle.kmer_batch(data, offs, k, m)
        shift = int(rng.choice([0, 0, 1, 5, 8, 15]))
        d_in = ctx.malloc(n * L + 32)
        ctx.h2d(d_in + shift, data)
        cap = n * (L - k + 1)
        d_out = ctx.malloc(cap * m * 8 + 16)
        d_cnt = ctx.malloc(n * 8)
        d_pos = ctx.malloc(cap * 4 + 16)
        want_pos = rng.random() < 0.3
        tot = ctx.kmer_hash_ptr(d_in + shift, 0, n, L, 0, k, m, d_out, cap, counts=d_cnt,
                                pos=d_pos if want_pos else 0)
        assert tot == want["total"], (it, n, L, k, m, dirty)
        got = np.zeros(tot * m, np.uint64)
        cnt = np.zeros(n, np.uint64)
        ctx.d2h(got, d_out)
        ctx.d2h(cnt, d_cnt)
        assert (got == want["hashes"].ravel()).all(), (it, n, L, k, m, dirty, shift)
        assert (cnt == want["counts"]).all(), (it, n, L, k, m, dirty)
        if want_pos:
            pos = np.zeros(tot, np.uint32)
            ctx.d2h(pos, d_pos)
            assert (pos == want["pos"]).all(), (it, n, L, k, m, dirty)
        for p_ in (d_in, d_out, d_cnt, d_pos):
            ctx.free(p_)


def test_fuzz_seed_shapes(ctx, oracle):
    import nthash_amd
    rng = np.random.default_rng(77)
    for it in range(60):
        k = int(rng.choice([4, 8, 16, 17, 31, 32, 33, 48, 64, 65, 80]))
        L = int(k + rng.choice([0, 1, 7, 30, 64, 119, 219]))
        n = int(rng.choice([1, 3, 64, 129, 500]))
        m2 = int(rng.choice([1, 2, 3, 5]))
        seeds = []
        for _ in range(int(rng.integers(1, 4))):
            half = "".join("1" if rng.random() < 0.6 else "0" for _ in range((k + 1) // 2))
            seeds.append(half + half[: k // 2][::-1])
        data = oracle.synth_reads(0, n, L, int(rng.integers(0, 1 << 30))).copy()
        if rng.random() < 0.3:
            data[rng.choice(n * L, max(1, n * L // 400), replace=False)] = ord("N")
        offs = np.arange(n + 1, dtype=np.uint64) * L
        want = oracle.seed_batch(data, offs, seeds, k, m2)
        got = ctx.seed_hash(data, seeds, k, m2, fixed_len=L, n_reads=n, want_pos=bool(rng.random() < 0.3))
        assert got["total"] == want["total"], (it, n, L, k, m2, seeds)
        assert (got["hashes"] == want["hashes"]).all(), (it, n, L, k, m2, seeds)
        assert (got["counts"] == want["counts"]).all()


@pytest.mark.parametrize("n,L,gap,k,m,dirty,pad", [
    (3000, 120, 1, 31, 1, False, b"\n"), (3000, 120, 1, 31, 1, True, b"\n"), (1500, 151, 5, 25, 3, True, b"ACGTN"),
    (900, 100, 28, 64, 2, False, b"A"), (40, 5003, 117, 31, 1, True, b"\r\n"), (2000, 64, 64, 21, 1, False, b"#"),
])
def test_kmer_padded_rows_vs_oracle(ctx, oracle, n, L, gap, k, m, dirty, pad):
    """rows with padding between the reads (stride > fixed_len; e.g. one read per line of a text file):
    the padding -- bases, newlines, anything -- is never hashed; positions and per-read counts as for packed reads"""
    rng = np.random.default_rng(n + gap)
    reads = oracle.synth_reads(6, n, L, 31 + gap).reshape(n, L).copy()
    if dirty:
        reads.ravel()[rng.integers(0, n * L, max(2, n * L // 3000))] = ord("N")
    rows = np.empty((n, L + gap), np.uint8)
    rows[:, :L] = reads
    rows[:, L:] = np.frombuffer(pad, np.uint8)[rng.integers(0, len(pad), (n, gap))]
    offs = np.arange(n + 1, dtype=np.uint64) * L
    want = oracle.kmer_batch(reads.ravel(), offs, k, m, want_pos=True)
    flat = rows.ravel()[: (n - 1) * (L + gap) + L]           # the batch ends with the last read, not its padding
    got = ctx.kmer_hash(flat, k, m, fixed_len=L, stride=L + gap, n_reads=n, want_pos=True)
    assert got["total"] == want["total"]
    assert (got["counts"] == want["counts"]).all()
    assert (got["hashes"] == want["hashes"]).all()
    assert (got["pos"] == want["pos"]).all()


# ---------------------------------------------------------------------------
# batched graph-extension query (BlindNtHash::peek / peek_back for all 4 bases)
# ---------------------------------------------------------------------------
def test_kmer_extend_golden(ctx):
    for c in load_golden("extend_cases.json"):
        kmer = np.frombuffer(c["kmer"].encode("latin-1"), dtype=np.uint8)
        r = ctx.kmer_extend(kmer, c["k"], c["m"])
        assert (r["self"][0] == h2i(c["self"])).all(), c["kmer"]
        for b in range(4):
            assert (r["next"][0, b] == h2i(c["next"][b])).all(), (c["kmer"], "next", b)
            assert (r["prev"][0, b] == h2i(c["prev"][b])).all(), (c["kmer"], "prev", b)


def test_seed_extend_golden(ctx):
    """nthip_seed_extend against the fixtures recorded from the real BlindSeedNtHash::roll(c) / roll_back(c)
    (tests/golden/gen_golden.py section 7): seeds with monomers, whose roll_back reads them from the window it leaves,
    the don't-care description, several seeds, k to 100"""
    for c in load_golden("seed_extend_cases.json"):
        kmer = np.frombuffer(c["kmer"].encode("latin-1"), dtype=np.uint8)
        r = ctx.seed_extend(kmer, c["seeds"], c["k"], c["m2"])
        assert (r["self"][0] == h2i(c["self"])).all(), (c["kmer"], c["seeds"])
        for b in range(4):
            assert (r["next"][0, b] == h2i(c["next"][b])).all(), (c["kmer"], c["seeds"], "next", b)
            assert (r["prev"][0, b] == h2i(c["prev"][b])).all(), (c["kmer"], c["seeds"], "prev", b)


@pytest.mark.parametrize("n,k,n_seeds,m2", [(5000, 31, 2, 3), (3001, 31, 1, 1), (777, 17, 3, 2), (500, 64, 2, 2), (300, 65, 1, 1),
                                            (200, 100, 2, 2), (64, 6, 1, 1), (1, 48, 2, 1), (100003, 31, 2, 3), (70, 33, 4, 1),
                                            (1000, 128, 1, 2), (1000, 21, 5, 4)])
def test_seed_extend_batch_vs_oracle(ctx, oracle, n, k, n_seeds, m2):
    """a batch of windows against the oracle's nto_seed_extend (pinned to the real reference, tests/test_oracle.py), and
    the successors against the hash stream: successor b of window i == SeedNtHash of (window[1:] + b)"""
    rng = np.random.default_rng(k + n_seeds)
    seeds = ["".join("10"[int(x)] for x in rng.integers(0, 2, k)) for _ in range(n_seeds)]
    if n_seeds > 1:
        seeds[1] = "1" * k if k % 2 else "11" + "0" * (k - 4) + "11"
    kmers = np.frombuffer(b"ACGTacgtUu", dtype=np.uint8)[rng.integers(0, 10, (n, k))]
    r = ctx.seed_extend(kmers.ravel(), seeds, k, m2)
    per = n_seeds * m2
    for i in list(range(min(n, 200))) + [n - 1]:
        me, nx, pv = oracle.seed_extend(kmers[i].tobytes(), seeds, m2)
        assert (r["self"][i] == me).all(), i
        assert (r["next"][i] == nx).all(), i
        assert (r["prev"][i] == pv).all(), i
    # every successor through the stream kernels: window[1:] + b as a read of k bases
    succ = np.empty((n, 4, k), np.uint8)
    succ[:, :, : k - 1] = kmers[:, None, 1:]
    succ[:, :, k - 1] = np.frombuffer(b"ACGT", np.uint8)[None, :]
    offs = np.arange(4 * n + 1, dtype=np.uint64) * k
    want = oracle.seed_batch(succ.ravel(), offs, seeds, k, m2, want_pos=False)["hashes"]
    assert (r["next"].reshape(4 * n, per) == want).all()
    only = ctx.seed_extend(kmers.ravel(), seeds, k, m2, want_self=False, want_next=False)
    assert (only["prev"] == r["prev"]).all()


@pytest.mark.parametrize("n,k,m", [(5000, 31, 3), (3001, 31, 1), (777, 17, 1), (500, 64, 2), (300, 65, 1), (200, 100, 2),
                                   (64, 4, 1), (1, 48, 1), (600033, 31, 1), (70, 33, 1), (4001, 31, 8), (1000, 21, 5), (300000, 25, 2),
                                   (5003, 96, 1), (700, 200, 3), (131, 1000, 2), (70, 2500, 1), (3, 9000, 1)])
def test_kmer_extend_batch_consistency(ctx, oracle, n, k, m):
    """a batch: successor b of k-mer i == the hash stream entry of (kmer[1:] + b); self == k-mer hash
    (k <= 64: byte tables; k > 64: Horner over a staged 2-bit stream; very long k: the lane-per-k-mer kernel)"""
    rng = np.random.default_rng(12 + k)
    kmers = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (n, k))]
    r = ctx.kmer_extend(kmers.ravel(), k, m)
    offs = np.arange(n + 1, dtype=np.uint64) * k
    want_self = oracle.kmer_batch(kmers.ravel(), offs, k, m, want_pos=False)["hashes"]
    assert (r["self"] == want_self).all()
    for b, ch in enumerate(b"ACGT"):
        nxt = np.concatenate([kmers[:, 1:], np.full((n, 1), ch, np.uint8)], axis=1)
        prv = np.concatenate([np.full((n, 1), ch, np.uint8), kmers[:, :-1]], axis=1)
        assert (r["next"][:, b] == oracle.kmer_batch(nxt.ravel(), offs, k, m, want_pos=False)["hashes"]).all()
        assert (r["prev"][:, b] == oracle.kmer_batch(prv.ravel(), offs, k, m, want_pos=False)["hashes"]).all()


# ---------------------------------------------------------------------------
# fused consumers of the hash stream: Bloom filter insert / query
# ---------------------------------------------------------------------------
def _bloom_expected(hashes, n_bits):
    """filter bytes after setting bit (h mod n_bits) of every hash: bit p = bit p&7 of byte p>>3"""
    nbytes = (n_bits + 31) // 32 * 4
    pos = hashes.ravel() % np.uint64(n_bits)
    bits = np.zeros(nbytes * 8, np.uint8)
    bits[pos.astype(np.int64)] = 1
    return np.packbits(bits, bitorder="little")


@pytest.mark.parametrize("n,L,k,m,n_bits,dirty", [
    (3000, 150, 31, 1, 1 << 22, False),          # power of two: mask
    (3000, 150, 31, 4, 4_000_037, False),        # prime size: invariant modulo
    (2000, 150, 31, 3, 12_345_678, True),        # reads with N: only emitted k-mers are consumed
    (1500, 101, 25, 2, 999_983, True), (700, 100, 64, 3, (1 << 20) + 32, False), (50, 5003, 31, 1, 1 << 18, True),
    (1, 150, 31, 8, 64, False), (300, 36, 21, 1, 33, False),
])
def test_bloom_insert_matches_oracle_hash_stream(ctx, oracle, n, L, k, m, n_bits, dirty):
    """the filter the fused kernel builds == the filter built on the CPU from the oracle's hash stream
    (OR is order-free, so the comparison is bit-exact); the unfused stream consumer agrees"""
    rng = np.random.default_rng(n + L)
    data = oracle.synth_reads(2, n, L, 99 + k).copy()
    if dirty:
        bad = rng.choice(n * L, max(3, n * L // 500), replace=False)
        data[bad] = np.frombuffer(b"NnRY-", dtype=np.uint8)[rng.integers(0, 5, bad.size)]
    offs = np.arange(n + 1, dtype=np.uint64) * L
    want = oracle.kmer_batch(data, offs, k, m, want_pos=False)
    expect = _bloom_expected(want["hashes"], n_bits)
    d_f, nbytes = ctx.bloom_new(n_bits)
    total = ctx.bloom_insert(data, k, m, L, n, d_f, n_bits)
    assert total == want["total"]
    got = np.zeros(nbytes, np.uint8)
    ctx.d2h(got, d_f)
    assert (got == expect).all()
    # inserting again changes nothing; the stream consumer on the materialised hashes gives the same filter
    assert ctx.bloom_insert(data, k, m, L, n, d_f, n_bits) == total
    ctx.d2h(got, d_f)
    assert (got == expect).all()
    d_f2, _ = ctx.bloom_new(n_bits)
    hs = np.ascontiguousarray(want["hashes"]).ravel()
    if hs.size:
        d_h = ctx.malloc(hs.size * 8)
        ctx.h2d(d_h, hs)
        ctx.stream_bloom_insert_ptr(d_h, hs.size, d_f2, n_bits)
        ctx.free(d_h)
    got2 = np.zeros(nbytes, np.uint8)
    ctx.d2h(got2, d_f2)
    assert (got2 == expect).all()
    ctx.free(d_f)
    ctx.free(d_f2)


@pytest.mark.parametrize("n,L,k,m,n_bits,dirty", [
    (3000, 150, 31, 1, 1 << 22, False),              # 4 regions, one bin: straight to the regions
    (3000, 150, 31, 4, 4_000_037, True),             # prime size: invariant modulo, a partial last region; reads with N
    (700, 100, 64, 3, (1 << 20) + 32, False),        # a second region of 32 bits
    (1, 150, 31, 8, 64, False), (300, 36, 21, 1, 33, False),   # filters smaller than a vector
    (4000, 150, 31, 2, (1 << 28) + 12_345, False),   # 3 bins (the last one partial): both partition levels
    (2500, 250, 31, 1, 1 << 30, True),               # 8 bins
    (1200, 150, 31, 3, (1 << 33) - 1_234_567, False),  # 64 bins, the last one partial
    (9000, 150, 31, 2, (1 << 28) + 77, True),        # several rounds (NTHIP_TUNE_BLOOM_ROUND below: 300 000 values each)
])
def test_bloom_binned_insert_matches_oracle_hash_stream(oracle, n, L, k, m, n_bits, dirty):
    """the binned insert (histogram -> region lists -> one workgroup per 128 KiB region, no device atomics;
    NTHIP_TUNE_BLOOM_BINNED=1 takes it whatever the batch size) builds the filter the CPU builds from the oracle's hash
    stream -- from reads and from a materialised stream, on a filter that already holds bits, with repeated k-mers"""
    import os
    import nthash_amd
    os.environ["NTHIP_TUNE_BLOOM_BINNED"] = "1"
    if n == 9000:
        os.environ["NTHIP_TUNE_BLOOM_ROUND"] = "300000"
    try:
        ctx = nthash_amd.Context(0)
    finally:
        os.environ.pop("NTHIP_TUNE_BLOOM_BINNED", None)
        os.environ.pop("NTHIP_TUNE_BLOOM_ROUND", None)
    rng = np.random.default_rng(n + L + m)
    data = oracle.synth_reads(2, n, L, 99 + k).copy()
    data[: 3 * L] = ord("A")                      # low-complexity reads: one value many times
    if n > 10:
        data[5 * L: 6 * L] = data[4 * L: 5 * L]   # a duplicated read
    if dirty:
        bad = rng.choice(n * L, max(3, n * L // 500), replace=False)
        data[bad] = np.frombuffer(b"NnRY-", dtype=np.uint8)[rng.integers(0, 5, bad.size)]
    offs = np.arange(n + 1, dtype=np.uint64) * L
    want = oracle.kmer_batch(data, offs, k, m, want_pos=False)
    nbytes = (n_bits + 31) // 32 * 4

    def expect_words(prior=None):
        pos = (want["hashes"].ravel() % np.uint64(n_bits)).astype(np.int64)
        words = np.zeros(nbytes // 4, np.uint32) if prior is None else prior.copy()
        np.bitwise_or.at(words, pos >> 5, (np.uint32(1) << (pos & 31).astype(np.uint32)))
        return words

    d_f, _ = ctx.bloom_new(n_bits)
    ctx.set_profiling(True)
    total = ctx.bloom_insert(data, k, m, L, n, d_f, n_bits)
    name = ctx.last_kernel_ms()[1]
    ctx.set_profiling(False)
    assert name.startswith("bloom binned insert"), name
    assert total == want["total"]
    got = np.zeros(nbytes // 4, np.uint32)
    ctx.d2h(got, d_f)
    exp = expect_words()
    assert (got == exp).all(), (int((got != exp).sum()), "words differ")
    # a filter that already holds bits (every 7th word all ones) keeps them; the stream entry, in two uneven calls
    prior = np.zeros(nbytes // 4, np.uint32)
    prior[::7] = 0xFFFFFFFF
    ctx.h2d(d_f, prior)
    hs = np.ascontiguousarray(want["hashes"]).ravel()
    d_h = ctx.malloc(max(8, hs.size * 8))
    ctx.h2d(d_h, hs)
    cut = hs.size // 3 | 1
    ctx.stream_bloom_insert_ptr(d_h, cut, d_f, n_bits)
    ctx.stream_bloom_insert_ptr(d_h + 8 * cut, hs.size - cut, d_f, n_bits)
    ctx.d2h(got, d_f)
    exp = expect_words(prior)
    assert (got == exp).all(), (int((got != exp).sum()), "words differ")
    ctx.free(d_h)
    ctx.free(d_f)
    ctx.close()


@pytest.mark.parametrize("n,L,k,m,n_bits,n_cnt,dirty,knobs,expect", [
    (3000, 150, 31, 1, 1 << 22, 1 << 16, False, {}, "slots"),                            # one bin: level 1 straight to the regions
    (4000, 150, 31, 2, (1 << 28) + 12_345, (1 << 23) + 12_344, False, {}, "slots"),      # 3 bins, the last one partial: both levels
    (1200, 150, 31, 3, (1 << 33) - 1_234_567, 1 << 26, False, {}, "slots"),              # 64 bins / 16 bins
    (4000, 150, 31, 2, (1 << 28) + 12_345, (1 << 23) + 12_344, False, {"NTHIP_TUNE_BLOOM_SLOT_TIGHT": "1"}, "slots"),   # the overflow list in use
    (3000, 150, 31, 1, 1 << 22, 1 << 16, False, {"NTHIP_TUNE_BLOOM_SLOT_TIGHT": "1"}, "slots"),
    (4000, 150, 31, 2, (1 << 28) + 12_345, (1 << 23) + 12_344, False, {"NTHIP_TUNE_BLOOM_SLOT_TIGHT": "2"}, "fused insert ("),  # the round fails: exact lists
    (4000, 150, 31, 2, (1 << 28) + 12_345, (1 << 23) + 12_344, False, {"NTHIP_TUNE_BLOOM_SLOTS": "2"}, "fused insert ("),      # never slots
    (9000, 150, 31, 2, (1 << 28) + 77, 1 << 24, False, {"NTHIP_TUNE_BLOOM_ROUND": "300000"}, "slots"),                         # several rounds
    (2500, 250, 31, 1, 1 << 30, 1 << 26, True, {}, "slots"),                             # reads with non-bases: their threads count the bases since the last one
    (3000, 150, 31, 4, 4_000_037, 4_000_036, True, {"NTHIP_TUNE_BLOOM_SLOTS": "2"}, "fused insert ("),   # the same on the exact lists (both passes)
    (3000, 151, 25, 2, 1 << 24, 1 << 20, True, {"NTHIP_TUNE_BLOOM_ROUND": "200000"}, "slots"),
    (700, 100, 64, 3, (1 << 20) + 32, 40_000, False, {"NTHIP_TUNE_BLOOM_FUSED": "1"}, "slots"),
    (4000, 150, 31, 2, (1 << 28) + 12_345, (1 << 23) + 12_344, True, {"NTHIP_TUNE_BLOOM_PIECES": "2"}, "slots ("),   # round 5: two levels behind shared cursors (not pieces)
    (5000, 150, 31, 1, (1 << 30) + 64, 1 << 27, True, {}, "pieces"),                    # round 5: block-private pieces, whole lines; reads with non-bases
    (5000, 150, 31, 3, (1 << 29) - 8_000_000, 1 << 25, False, {"NTHIP_TUNE_BLOOM_SLOT_TIGHT": "1"}, "pieces"),   # ... the overflow list in use
])
def test_binned_insert_without_a_hash_stream(oracle, n, L, k, m, n_bits, n_cnt, dirty, knobs, expect):
    """Device-resident fixed-length reads into a Bloom filter / a counting sketch through the lists, with no hash stream:
    slots mode (every bucket owns mean + 8 sigma entries, the reads hashed once, what does not fit through the overflow list),
    buckets of exactly the mean (the overflow list really used), of half the mean (the round fails, the table is untouched
    and the exact lists -- count, scan, part, apply -- redo it), slots mode switched off, several rounds, reads with
    non-bases (the fused pass skips their windows itself, in both modes).  The table the CPU builds from the oracle's hash
    stream, on tables that already hold something; low-complexity reads (one value hundreds of times) included."""
    import os
    import nthash_amd
    os.environ["NTHIP_TUNE_BLOOM_BINNED"] = "1"
    os.environ.update(knobs)
    try:
        ctx = nthash_amd.Context(0)
    finally:
        os.environ.pop("NTHIP_TUNE_BLOOM_BINNED", None)
        for key in knobs:
            os.environ.pop(key, None)
    rng = np.random.default_rng(n + L + m)
    data = oracle.synth_reads(2, n, L, 99 + k).copy()
    data[: 4 * L] = ord("A")                      # one k-mer ~500 times
    data[5 * L: 6 * L] = data[4 * L: 5 * L]       # a duplicated read
    if dirty:
        bad = rng.choice(n * L, max(3, n * L // 500), replace=False)
        data[bad] = np.frombuffer(b"NnRY-", dtype=np.uint8)[rng.integers(0, 5, bad.size)]
        data[7 * L] = ord("N")                    # a read's first base, another's last, a read of non-bases, two neighbours
        data[9 * L - 1] = ord("n")
        data[10 * L: 11 * L] = ord("N")
        data[12 * L - 1] = data[12 * L] = ord("-")
    offs = np.arange(n + 1, dtype=np.uint64) * L
    want = oracle.kmer_batch(data, offs, k, m, want_pos=False)
    hs = np.ascontiguousarray(want["hashes"]).ravel()
    d_in = ctx.malloc(n * L)
    ctx.h2d(d_in, data)
    # Bloom filter that already holds bits
    nbytes = (n_bits + 31) // 32 * 4
    prior = np.zeros(nbytes // 4, np.uint32)
    prior[::7] = 0x80000001
    d_f = ctx.malloc(nbytes)
    ctx.h2d(d_f, prior)
    ctx.set_profiling(True)
    total = ctx.bloom_insert_ptr(d_in, n, L, 0, k, m, d_f, n_bits)
    name = ctx.last_kernel_ms()[1]
    # (round 5: a slots-mode round of a two-level table runs on block-private pieces -- "slots" rows accept either label)
    assert (expect in name or (expect == "slots" and "pieces" in name)) and name.startswith("bloom"), name
    assert total == want["total"]
    pos = (hs % np.uint64(n_bits)).astype(np.int64)
    exp = prior.copy()
    np.bitwise_or.at(exp, pos >> 5, (np.uint32(1) << (pos & 31).astype(np.uint32)))
    got = np.zeros(nbytes // 4, np.uint32)
    ctx.d2h(got, d_f)
    assert (got == exp).all(), (int((got != exp).sum()), "words differ")
    ctx.free(d_f)
    # counting sketch that already holds counts, some near the top
    priorc = rng.integers(0, 4, n_cnt, dtype=np.int64)
    priorc[::5] = 250
    d_c = ctx.malloc(n_cnt)
    ctx.h2d(d_c, priorc.astype(np.uint8))
    total = ctx.count_insert_ptr(d_in, n, L, 0, k, m, d_c, n_cnt)
    name = ctx.last_kernel_ms()[1]
    okc = expect in name or (expect in ("slots", "pieces", "slots (") and ("pieces" in name or "slots" in name))
    assert okc and name.startswith("count"), name
    assert total == want["total"]
    tally = np.bincount((hs % np.uint64(n_cnt)).astype(np.int64), minlength=n_cnt).astype(np.int64)
    expc = np.minimum(255, priorc + tally).astype(np.uint8)
    gotc = np.zeros(n_cnt, np.uint8)
    ctx.d2h(gotc, d_c)
    assert (gotc == expc).all(), (int((gotc != expc).sum()), "counters differ")
    assert expc.max() == 255
    ctx.free(d_c)
    ctx.free(d_in)
    ctx.close()


def _minimizers_brute(pos, hashes, nwin, w):
    """positions picked by the windows of w positions of one read (its emitted k-mers: pos ascending, one hash each)"""
    w = min(w, nwin)
    picked = set()
    for s in range(0, nwin - w + 1):
        lo, hi = np.searchsorted(pos, s), np.searchsorted(pos, s + w)
        if hi > lo:
            picked.add(int(pos[lo + int(np.argmin(hashes[lo:hi]))]))   # (argmin: the first of equal values)
    return sorted(picked)


@pytest.mark.parametrize("n,L,k,w,dirty,rounds", [
    (300, 150, 31, 10, False, False), (300, 150, 31, 10, True, False), (200, 100, 21, 1, True, False),
    (150, 120, 31, 64, True, False),     # windows wider than most stretches between non-bases
    (100, 60, 31, 50, False, False),     # fewer windows than w: the read is one window
    (64, 700, 64, 19, True, False), (40, 1500, 101, 25, True, False),
    (2500, 150, 31, 12, True, True),     # several rounds of reads (NTHIP_TUNE_BLOOM_ROUND)
    (5000, 150, 31, 10, True, False), (3000, 101, 21, 5, True, False), (2000, 158, 31, 128, True, False),
    (3000, 250, 31, 10, True, False), (1500, 286, 31, 200, True, False), (1000, 200, 21, 65, True, False),   # four register sets
])
def test_minimizers_match_brute_force_on_oracle_stream(oracle, n, L, k, w, dirty, rounds):
    """nthip_kmer_minimizers: per read, of every w consecutive window positions the emitted k-mer with the smallest
    canonical hash (ties: the leftmost) -- against a brute force over the oracle's stream (positions + hashes per read),
    reads with non-bases, low-complexity reads (runs of equal hashes), reads shorter than k"""
    import os
    import nthash_amd
    if rounds:
        os.environ["NTHIP_TUNE_BLOOM_ROUND"] = "60000"
    try:
        ctx = nthash_amd.Context(0)
    finally:
        os.environ.pop("NTHIP_TUNE_BLOOM_ROUND", None)
    rng = np.random.default_rng(n + L + w)
    data = oracle.synth_reads(3, n, L, 5 + k).copy()
    data[2 * L: 3 * L] = ord("A")                                                   # every hash of the read is the same
    data[3 * L: 4 * L] = np.frombuffer(b"AC" * L, dtype=np.uint8)[:L]               # period 2: two values alternate
    if dirty:
        bad = rng.choice(n * L, max(3, n * L // 300), replace=False)
        data[bad] = np.frombuffer(b"NnRY-", dtype=np.uint8)[rng.integers(0, 5, bad.size)]
    offs = np.arange(n + 1, dtype=np.uint64) * L
    want = oracle.kmer_batch(data, offs, k, 1, want_pos=True)
    nwin = L - k + 1
    exp_off, exp_pos, exp_h = [0], [], []
    o = 0
    for r in range(n):
        c = int(want["counts"][r])
        p, h = want["pos"][o:o + c].astype(np.int64), want["hashes"][o:o + c].ravel()
        picked = _minimizers_brute(p, h, nwin, w)
        look = dict(zip(p.tolist(), h.tolist()))
        exp_pos += picked
        exp_h += [look[q] for q in picked]
        exp_off.append(len(exp_pos))
        o += c
    got = ctx.minimizers(data, k, w, L, n)
    assert got["total"] == len(exp_pos)
    assert (got["offsets"] == np.array(exp_off, np.uint64)).all()
    assert (got["pos"] == np.array(exp_pos, np.uint32)).all()
    assert (got["hashes"] == np.array(exp_h, np.uint64)).all()
    # device-resident reads: the dense pass alone first, a round with a non-base once more under the read-slots contract
    dev = ctx.minimizers(data, k, w, L, n, device_input=True)
    assert dev["total"] == got["total"] and (dev["offsets"] == got["offsets"]).all()
    assert (dev["pos"] == got["pos"]).all() and (dev["hashes"] == got["hashes"]).all()
    if got["total"] > 1:   # too small a capacity: the need is reported
        with pytest.raises(nthash_amd.NtHipError) as ei:
            ctx.minimizers(data, k, w, L, n, capacity=got["total"] - 1)
        assert ei.value.code == nthash_amd.capi.NTHIP_ERR_CAPACITY and ei.value.total == got["total"]
    ctx.close()


@pytest.mark.parametrize("n,L,k,w", [
    (700, 150, 31, 10), (700, 150, 31, 1), (300, 150, 31, 5), (300, 150, 31, 19), (300, 150, 31, 64), (200, 150, 31, 120),
    (200, 150, 31, 119), (200, 150, 31, 500), (300, 158, 31, 33),   # 128 windows: the two register sets full
    (300, 94, 31, 16), (300, 95, 31, 17), (300, 93, 31, 63),        # 64 / 65 / 63 windows
    (500, 31, 31, 4), (400, 32, 31, 2), (300, 100, 64, 37), (300, 250, 160, 12),
    (70000, 36, 21, 7),                                             # chunks of several reads per wave
    (600, 250, 31, 10), (300, 250, 31, 1), (300, 250, 31, 64), (300, 250, 31, 129), (200, 250, 31, 220), (200, 250, 31, 900),
    (300, 286, 31, 100), (300, 159, 31, 17), (300, 222, 31, 128), (20000, 250, 31, 19),   # 129 ... 256 windows: four register sets
])
@pytest.mark.parametrize("table", [False, True])
def test_minimizers_of_clean_short_reads(oracle, n, L, k, w, table):
    """clean fixed-length reads of at most 128 windows take the register tables (minimizer_dense_kernel: chunks compacted
    in place + gather); NTHIP_TUNE_MZ_TABLE=1 sends the same batch through the LDS tables: both against the brute force.
    Reads of one repeated base / period 2 (ties: the leftmost), a too small capacity, several rounds"""
    import os
    import nthash_amd
    if table:
        os.environ["NTHIP_TUNE_MZ_TABLE"] = "1"
    if n == 700:
        os.environ["NTHIP_TUNE_BLOOM_ROUND"] = "30000"   # rounds of 250 reads
    try:
        ctx = nthash_amd.Context(0)
    finally:
        os.environ.pop("NTHIP_TUNE_MZ_TABLE", None)
        os.environ.pop("NTHIP_TUNE_BLOOM_ROUND", None)
    data = oracle.synth_reads(11, n, L, 3 + k + w).copy()
    data[2 * L: 3 * L] = ord("A")
    data[3 * L: 4 * L] = np.frombuffer(b"AC" * L, dtype=np.uint8)[:L]
    data[5 * L: 6 * L] = np.frombuffer(b"ACG" * L, dtype=np.uint8)[:L]
    offs = np.arange(n + 1, dtype=np.uint64) * L
    want = oracle.kmer_batch(data, offs, k, 1, want_pos=False)
    nwin = L - k + 1
    hs = want["hashes"].ravel().reshape(n, nwin)
    weff = min(w, nwin)
    if n > 5000:    # vectorised brute force: window s picks the first argmin of hs[:, s:s+w]
        from numpy.lib.stride_tricks import sliding_window_view
        arg = sliding_window_view(hs, weff, axis=1).argmin(axis=2) + np.arange(nwin - weff + 1)[None, :]
        pick = np.zeros((n, nwin), dtype=bool)
        np.put_along_axis(pick, arg, True, axis=1)
        exp_off = np.concatenate([[0], np.cumsum(pick.sum(axis=1))]).astype(np.uint64)
        rr, pp = np.nonzero(pick)
        exp_pos, exp_h = pp.astype(np.uint32), hs[rr, pp]
    else:
        eo, ep, eh = [0], [], []
        pos = np.arange(nwin, dtype=np.int64)
        for r in range(n):
            picked = _minimizers_brute(pos, hs[r], nwin, w)
            ep += picked
            eh += [hs[r][q] for q in picked]
            eo.append(len(ep))
        exp_off, exp_pos, exp_h = np.array(eo, np.uint64), np.array(ep, np.uint32), np.array(eh, np.uint64)
    got = ctx.minimizers(data, k, w, L, n)
    assert got["total"] == len(exp_pos)
    assert (got["offsets"] == exp_off).all()
    assert (got["pos"] == exp_pos).all()
    assert (got["hashes"] == exp_h).all()
    dev = ctx.minimizers(data, k, w, L, n, device_input=True)
    assert dev["total"] == got["total"] and (dev["offsets"] == got["offsets"]).all()
    assert (dev["pos"] == got["pos"]).all() and (dev["hashes"] == got["hashes"]).all()
    if got["total"] > 1 and n <= 700:
        with pytest.raises(nthash_amd.NtHipError) as ei:
            ctx.minimizers(data, k, w, L, n, capacity=got["total"] - 1)
        assert ei.value.code == nthash_amd.capi.NTHIP_ERR_CAPACITY and ei.value.total == got["total"]
    ctx.close()


@pytest.mark.parametrize("n,L,k,w,C", [
    (3000, 150, 31, 10, 0), (3000, 150, 31, 10, 8), (3000, 150, 31, 10, 7), (3000, 150, 31, 10, 5), (3000, 150, 31, 10, 2),
    (2000, 150, 31, 2, 0), (2000, 150, 31, 3, 0), (2000, 150, 31, 5, 0), (2000, 150, 31, 16, 0), (2000, 150, 31, 17, 0),
    (2000, 150, 31, 19, 0), (2000, 150, 31, 19, 9), (2000, 150, 31, 33, 0), (2000, 150, 31, 50, 0), (2000, 150, 31, 50, 8),
    (2000, 150, 31, 64, 16), (1000, 150, 31, 120, 0), (1000, 150, 31, 119, 0), (1000, 150, 31, 100, 15),
    (3000, 151, 31, 10, 0), (3000, 101, 21, 7, 0), (3000, 100, 64, 12, 0), (3000, 96, 48, 9, 0), (3000, 76, 15, 11, 0),
    (2000, 250, 31, 10, 0), (2000, 250, 31, 25, 0), (1000, 1054, 31, 16, 16), (500, 600, 17, 40, 0), (5000, 40, 31, 10, 0),
    (5000, 36, 21, 7, 0), (3000, 150, 31, 128, 0),
    (3000, 150, 31, 4, 0), (3000, 150, 31, 6, 0), (3000, 150, 31, 7, 0), (3000, 150, 31, 8, 0), (3000, 150, 31, 9, 0),
    (3000, 150, 31, 11, 0), (3000, 150, 31, 12, 0), (3000, 150, 31, 13, 0), (3000, 150, 31, 14, 0), (3000, 150, 31, 15, 0),
    (3000, 100, 21, 10, 0), (3000, 250, 31, 14, 0), (3000, 45, 31, 15, 0), (3000, 46, 31, 15, 0), (3000, 40, 16, 5, 0),
    (3000, 61, 12, 10, 0), (2000, 1055, 32, 16, 0),
    (200000, 150, 31, 10, 0), (300000, 100, 25, 5, 0),         # many rounds of tiles: the look-back over the block-rounds
])
def test_minimizers_fused_one_pass(oracle, n, L, k, w, C):
    """minimizer_fused_kernel (round 4): device-resident clean fixed-length reads are hashed and their minimizers picked in
    ONE kernel -- no hash stream in HBM.  Against the vectorised brute force over the oracle's stream (first argmin of every
    window of w), for run lengths (= blocks of the sliding minimum) chosen by the plan and forced (NTHIP_TUNE_MZ_C),
    w from 2 to 128 (with and without whole blocks between a window's ends), k across the table widths, reads of one
    repeated base / short periods (ties: the leftmost), too small a capacity, and the round-3 kernels on the same batch
    (NTHIP_TUNE_MZ_FUSED=2).  A batch with a non-base comes back through the N-aware path with the same answers."""
    import os
    import nthash_amd
    from numpy.lib.stride_tricks import sliding_window_view
    if C:
        os.environ["NTHIP_TUNE_MZ_C"] = str(C)           # (a forced run length: of the any-run-length form)
        os.environ["NTHIP_TUNE_MZ_FUSED"] = "1"
    try:
        ctx = nthash_amd.Context(0)
    finally:
        os.environ.pop("NTHIP_TUNE_MZ_C", None)
        os.environ.pop("NTHIP_TUNE_MZ_FUSED", None)
    os.environ["NTHIP_TUNE_MZ_FUSED"] = "2"
    try:
        old = nthash_amd.Context(0)
        os.environ["NTHIP_TUNE_MZ_FUSED"] = "1"     # the any-run-length form where the record form (run length = w) would run
        gen = nthash_amd.Context(0)
    finally:
        os.environ.pop("NTHIP_TUNE_MZ_FUSED", None)
    data = oracle.synth_reads(5, n, L, 17 + k + w).copy()
    data[2 * L: 3 * L] = ord("A")
    data[3 * L: 4 * L] = np.frombuffer(b"AC" * L, dtype=np.uint8)[:L]
    data[5 * L: 6 * L] = np.frombuffer(b"ACG" * L, dtype=np.uint8)[:L]
    data[(n - 1) * L:] = ord("t")                               # the batch's last read too
    offs = np.arange(n + 1, dtype=np.uint64) * L
    nwin = L - k + 1
    hs = oracle.kmer_batch(data, offs, k, 1, want_pos=False)["hashes"].ravel().reshape(n, nwin)
    weff = min(w, nwin)
    arg = sliding_window_view(hs, weff, axis=1).argmin(axis=2) + np.arange(nwin - weff + 1)[None, :]
    pick = np.zeros((n, nwin), dtype=bool)
    np.put_along_axis(pick, arg, True, axis=1)
    exp_off = np.concatenate([[0], np.cumsum(pick.sum(axis=1))]).astype(np.uint64)
    rr, pp = np.nonzero(pick)
    exp_pos, exp_h = pp.astype(np.uint32), hs[rr, pp]
    ctx.set_profiling(True)
    got = ctx.minimizers(data, k, w, L, n, device_input=True)
    record_form = C == 0 and 4 <= w <= 16 and k <= 32 and w <= nwin and -(-nwin // w) <= 64
    if record_form:
        assert ctx.last_kernel_ms()[1] == "minimizer_w_kernel"
    elif w <= 100 and w <= nwin:    # (w beyond 7 blocks of 16 windows, or beyond the read: the round-3 kernels)
        assert ctx.last_kernel_ms()[1] == "minimizer_fused_kernel"
    assert got["total"] == len(exp_pos)
    assert (got["offsets"] == exp_off).all()
    assert (got["pos"] == exp_pos).all()
    assert (got["hashes"] == exp_h).all()
    if n <= 5000:
        gen.set_profiling(True)
        for other in (old, gen):
            ref = other.minimizers(data, k, w, L, n, device_input=True)
            if other is gen and w <= 100 and w <= nwin:    # (w beyond 7 blocks of 16 windows, or beyond the read: the round-3 kernels)
                assert gen.last_kernel_ms()[1] == "minimizer_fused_kernel"
            assert ref["total"] == got["total"] and (ref["offsets"] == got["offsets"]).all()
            assert (ref["pos"] == got["pos"]).all() and (ref["hashes"] == got["hashes"]).all()
        with pytest.raises(nthash_amd.NtHipError) as ei:        # too small a capacity: the need is reported, nothing past it written
            ctx.minimizers(data, k, w, L, n, capacity=got["total"] - 1, device_input=True)
        assert ei.value.code == nthash_amd.capi.NTHIP_ERR_CAPACITY and ei.value.total == got["total"]
        # a non-base: the one-pass kernel reports it, the N-aware path answers
        dirty = data.copy()
        dirty[7 * L + L // 2] = ord("N")
        a = ctx.minimizers(dirty, k, w, L, n, device_input=True)
        b = old.minimizers(dirty, k, w, L, n, device_input=True)
        assert a["total"] == b["total"] and (a["offsets"] == b["offsets"]).all()
        assert (a["pos"] == b["pos"]).all() and (a["hashes"] == b["hashes"]).all()
    ctx.close()
    old.close()
    gen.close()


@pytest.mark.parametrize("w", [4, 10, 16, 21])
def test_minimizers_one_pass_when_every_window_picks(ctx, oracle, w):
    """reads of ONE repeated base: all hashes equal, the leftmost wins, so every window picks a new position -- the densest
    output there is.  In the record form every tile overflows the stash that parks a tile's picks for a round and writes
    them itself (the wave claims its round's look-back early); mixed with ordinary reads so that both ways meet in one block"""
    n, L, k = 40000, 150, 31
    data = oracle.synth_reads(9, n, L, 5).copy().reshape(n, L)
    data[: n // 2] = ord("A")
    data[n // 2 + 7:: 50] = ord("c")
    data = data.ravel()
    nwin = L - k + 1
    offs = np.arange(n + 1, dtype=np.uint64) * L
    from numpy.lib.stride_tricks import sliding_window_view
    hs = oracle.kmer_batch(data, offs, k, 1, want_pos=False)["hashes"].ravel().reshape(n, nwin)
    arg = sliding_window_view(hs, w, axis=1).argmin(axis=2) + np.arange(nwin - w + 1)[None, :]
    pick = np.zeros((n, nwin), dtype=bool)
    np.put_along_axis(pick, arg, True, axis=1)
    rr, pp = np.nonzero(pick)
    got = ctx.minimizers(data, k, w, L, n, device_input=True)
    assert got["total"] == len(pp) and got["total"] >= (n // 2) * (nwin - w + 1)
    assert (got["offsets"] == np.concatenate([[0], np.cumsum(pick.sum(axis=1))]).astype(np.uint64)).all()
    assert (got["pos"] == pp.astype(np.uint32)).all()
    assert (got["hashes"] == hs[rr, pp]).all()


@pytest.mark.parametrize("n,lmax,k,w,dirty", [
    (400, 300, 31, 10, True), (300, 180, 21, 300, False),     # w beyond every read: one minimizer per read
    (60, 3000, 31, 19, True),                                  # reads on both sides of the 1024-window limit of the wave tables
    (500, 90, 25, 4, False),
    (600, 150, 31, 10, True), (600, 158, 31, 1, True), (300, 120, 31, 200, True),   # at most 128 windows: the register tables
    (600, 280, 31, 10, True), (400, 286, 31, 130, False), (400, 250, 25, 1, True),      # at most 256: four register sets
])
def test_minimizers_of_reads_given_by_offsets(oracle, ctx, n, lmax, k, w, dirty):
    """the same brute force, reads of any lengths (offsets): empty reads, reads shorter than k, reads of exactly k bases"""
    rng = np.random.default_rng(n + lmax + w)
    lens = rng.integers(0, lmax + 1, n).astype(np.uint64)
    lens[:4] = [0, k - 1, k, k + w - 1]
    lens[4] = lmax
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    total = int(offs[-1])
    data = oracle.synth_reads(3, 1, total, 9 + k).copy()
    if dirty:
        bad = rng.choice(total, max(3, total // 300), replace=False)
        data[bad] = np.frombuffer(b"NnRY-", dtype=np.uint8)[rng.integers(0, 5, bad.size)]
    want = oracle.kmer_batch(data, offs, k, 1, want_pos=True)
    exp_off, exp_pos, exp_h = [0], [], []
    o = 0
    for r in range(n):
        c = int(want["counts"][r])
        p, h = want["pos"][o:o + c].astype(np.int64), want["hashes"][o:o + c].ravel()
        nwin = max(int(lens[r]) - k + 1, 0)
        picked = _minimizers_brute(p, h, nwin, w) if nwin else []
        look = dict(zip(p.tolist(), h.tolist()))
        exp_pos += picked
        exp_h += [look[q] for q in picked]
        exp_off.append(len(exp_pos))
        o += c
    got = ctx.minimizers(data, k, w, 0, n, offsets=offs, device_input=(n % 200 == 0))
    assert got["total"] == len(exp_pos)
    assert (got["offsets"] == np.array(exp_off, np.uint64)).all()
    assert (got["pos"] == np.array(exp_pos, np.uint32)).all()
    assert (got["hashes"] == np.array(exp_h, np.uint64)).all()


def test_minimizers_argument_errors_and_short_reads(ctx):
    import nthash_amd
    data = np.frombuffer(b"ACGT" * 50, dtype=np.uint8)
    got = ctx.minimizers(data, 31, 5, 20, 10)          # reads shorter than k: no minimizers, offsets all zero
    assert got["total"] == 0 and (got["offsets"] == 0).all()
    with pytest.raises(nthash_amd.NtHipError):
        ctx.minimizers(data, 31, 0, 200, 1)            # w == 0
    with pytest.raises(nthash_amd.NtHipError):
        ctx.minimizers(data, 0, 5, 200, 1)             # k == 0


@pytest.mark.parametrize("n,L,k,m,n_counters,dirty,binned", [
    (3000, 150, 31, 1, 1 << 16, False, True),            # 2 regions: counters pile up (saturation at 255 matters)
    (3000, 150, 31, 4, 4_000_036, True, True),           # not a power of two, a partial last region; reads with N
    (1, 150, 31, 8, 64, False, True), (300, 36, 21, 1, 36, False, True),
    (4000, 150, 31, 2, (1 << 23) + 12_344, False, True), # 3 bins: both partition levels
    (2500, 250, 31, 1, 1 << 26, True, True),             # 16 bins
    (3000, 150, 31, 3, 1 << 14, False, False), (2000, 101, 25, 2, 999_984, True, False),   # the compare-and-swap kernel
])
def test_count_sketch_insert_and_query_match_oracle_hash_stream(oracle, n, L, k, m, n_counters, dirty, binned):
    """k-mer counting sketch (count-min, one-byte saturating counters): the table after nthip_kmer_count_insert /
    nthip_stream_count_insert == min(255, prior + number of stream values with h mod n_counters == slot), built on the
    CPU from the oracle's hash stream; nthip_stream_count_query == the smallest of each k-mer's m counters.  Both insert
    paths: the lists of the binned insert (NTHIP_TUNE_BLOOM_BINNED=1) and the compare-and-swap kernel (=2)"""
    import os
    import nthash_amd
    os.environ["NTHIP_TUNE_BLOOM_BINNED"] = "1" if binned else "2"
    try:
        ctx = nthash_amd.Context(0)
    finally:
        os.environ.pop("NTHIP_TUNE_BLOOM_BINNED", None)
    rng = np.random.default_rng(n + L + m)
    data = oracle.synth_reads(2, n, L, 77 + k).copy()
    data[: 4 * L] = ord("A")                       # one k-mer ~500 times: its counters saturate
    if dirty:
        bad = rng.choice(n * L, max(3, n * L // 500), replace=False)
        data[bad] = np.frombuffer(b"NnRY-", dtype=np.uint8)[rng.integers(0, 5, bad.size)]
    offs = np.arange(n + 1, dtype=np.uint64) * L
    want = oracle.kmer_batch(data, offs, k, m, want_pos=False)
    hs = np.ascontiguousarray(want["hashes"]).ravel()
    slots = (hs % np.uint64(n_counters)).astype(np.int64)
    tally = np.bincount(slots, minlength=n_counters).astype(np.int64)

    d_c = ctx.malloc(n_counters)
    ctx.memset(d_c, 0, n_counters)
    ctx.set_profiling(True)
    total = ctx.count_insert(data, k, m, L, n, d_c, n_counters)
    name = ctx.last_kernel_ms()[1]
    ctx.set_profiling(False)
    assert name.startswith("count binned insert" if binned else "count_atomic_kernel"), name
    assert total == want["total"]
    got = np.zeros(n_counters, np.uint8)
    ctx.d2h(got, d_c)
    exp = np.minimum(255, tally).astype(np.uint8)
    assert (got == exp).all(), (int((got != exp).sum()), "counters differ")
    assert n < 1000 or exp.max() == 255           # (the saturating case is really exercised)
    # on a sketch that already holds counts (some near the top), from a materialised stream in two uneven calls
    prior = rng.integers(0, 256, n_counters, dtype=np.int64)
    prior[::5] = 250
    ctx.h2d(d_c, prior.astype(np.uint8))
    d_h = ctx.malloc(max(8, hs.size * 8))
    ctx.h2d(d_h, hs)
    cut = hs.size // 3 | 1
    ctx.stream_count_insert_ptr(d_h, cut, d_c, n_counters)
    ctx.stream_count_insert_ptr(d_h + 8 * cut, hs.size - cut, d_c, n_counters)
    ctx.d2h(got, d_c)
    exp2 = np.minimum(255, prior + tally).astype(np.uint8)
    assert (got == exp2).all(), (int((got != exp2).sum()), "counters differ")
    # estimates: the smallest of a k-mer's m counters
    n_kmers = hs.size // m
    d_e = ctx.malloc(max(4, n_kmers))
    ctx.stream_count_query_ptr(d_h, n_kmers, m, d_c, n_counters, d_e)
    est = np.zeros(n_kmers, np.uint8)
    if n_kmers:
        ctx.d2h(est, d_e)
    assert (est == exp2[slots.reshape(n_kmers, m)].min(axis=1)).all()
    for p in (d_e, d_h, d_c):
        ctx.free(p)
    ctx.close()


def test_count_sketch_argument_errors(ctx):
    d_c = ctx.malloc(1024)
    data = np.frombuffer(b"ACGT" * 50, dtype=np.uint8)
    import nthash_amd
    for bad in (lambda: ctx.count_insert(data, 31, 1, 200, 1, 0, 1024),          # NULL sketch
                lambda: ctx.count_insert(data, 31, 1, 200, 1, d_c + 1, 1024),    # unaligned
                lambda: ctx.count_insert(data, 31, 1, 200, 1, d_c, 1022),        # not a multiple of 4
                lambda: ctx.count_insert(data, 31, 1, 200, 1, d_c, 0),
                lambda: ctx.count_insert(data, 0, 1, 200, 1, d_c, 1024),
                lambda: ctx.count_insert(data, 31, 0, 200, 1, d_c, 1024)):
        with pytest.raises(nthash_amd.NtHipError):
            bad()
    assert ctx.count_insert(data, 31, 1, 20, 10, d_c, 1024) == 0    # reads shorter than k: nothing counted
    ctx.free(d_c)


@pytest.mark.parametrize("n,L,k,m,n_bits", [
    (2000, 150, 31, 1, 1 << 20), (2000, 150, 31, 3, 3_000_017), (1200, 101, 25, 2, 700_001), (40, 5003, 31, 2, 1 << 21),
    (900, 100, 64, 1, 1 << 19),
])
def test_bloom_query_hits_per_read(ctx, oracle, n, L, k, m, n_bits):
    """insert one batch, query another that shares half of its reads and has N's: per-read hit counts
    == those computed on the CPU from the oracle's hashes and the expected filter"""
    rng = np.random.default_rng(7 * n + L)
    a_reads = oracle.synth_reads(0, n, L, 5)
    b_reads = oracle.synth_reads(n // 2, n, L, 5).copy()       # second half of A + n/2 new reads
    bad = rng.choice(n * L, max(3, n * L // 800), replace=False)
    b_reads[bad] = ord("N")
    offs = np.arange(n + 1, dtype=np.uint64) * L
    ha = oracle.kmer_batch(a_reads, offs, k, m, want_pos=False)
    filt = _bloom_expected(ha["hashes"], n_bits)
    d_f, nbytes = ctx.bloom_new(n_bits)
    ctx.bloom_insert(a_reads, k, m, L, n, d_f, n_bits)
    hb = oracle.kmer_batch(b_reads, offs, k, m, want_pos=False)
    bits = np.unpackbits(filt, bitorder="little")
    present = bits[(hb["hashes"] % np.uint64(n_bits)).astype(np.int64)].reshape(-1, m).all(axis=1)
    read_of = np.repeat(np.arange(n), hb["counts"].astype(np.int64))
    want_hits = np.bincount(read_of[present], minlength=n).astype(np.uint64)
    hits, total, found = ctx.bloom_query(b_reads, k, m, L, n, d_f, n_bits)
    assert total == hb["total"]
    assert found == int(want_hits.sum())
    assert (hits == want_hits).all()
    assert int(hits[: n // 2].sum()) >= int(hb["counts"][: n // 2].sum()) * 0  # shared reads: all their clean k-mers hit
    clean_shared = (hb["counts"][: n // 2] == L - k + 1)
    assert (hits[: n // 2][clean_shared] == L - k + 1).all()
    ctx.free(d_f)


@pytest.mark.parametrize("n,lmax,k,m,w,round_bases,device_input", [
    (3000, 250, 31, 2, 10, 20000, True), (3000, 250, 31, 2, 10, 20000, False), (500, 3000, 25, 1, 19, 9000, True),
    (60, 9000, 31, 3, 5, 4000, False),        # reads longer than a round: each alone
    (2000, 180, 21, 1, 300, 1000, True),
])
def test_consumers_by_offsets_without_a_batch_ceiling(ctx, oracle, n, lmax, k, m, w, round_bases, device_input):
    """Round 4: the consumers on reads given by offsets took the batch's compact stream in ONE round of device scratch and
    refused what did not fit (NTHIP_ERR_UNSUPPORTED: split the batch).  Now the batch is cut into rounds of reads whose stream
    fits (offsets_in_rounds; here NTHIP_TUNE_BLOOM_ROUND makes the rounds tiny -- a batch 20-300 x the bound) and every
    consumer carries its result across them: the filter and the sketch accumulate, hits and signatures land at their reads,
    the minimizers' CSR offsets run on.  Bit for bit what one round gives (itself checked against the oracle elsewhere)."""
    import os
    import nthash_amd
    os.environ["NTHIP_TUNE_BLOOM_ROUND"] = str(round_bases)
    try:
        small = nthash_amd.Context(0)
    finally:
        os.environ.pop("NTHIP_TUNE_BLOOM_ROUND", None)
    rng = np.random.default_rng(n + lmax)
    lens = rng.integers(0, lmax + 1, n).astype(np.uint64)
    lens[:3] = [0, k - 1, k]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    total_b = int(offs[-1])
    assert total_b > 4 * round_bases
    a = oracle.synth_reads(5, 1, total_b, 3 + k).copy()
    a[rng.choice(total_b, max(3, total_b // 900), replace=False)] = ord("N")
    n_bits, n_cnt = (1 << 22) + 64, 1 << 16
    res = []
    for c in (ctx, small):
        d_f, nbytes = c.bloom_new(n_bits)
        d_c = c.malloc(n_cnt)
        c.memset(d_c, 0, n_cnt)
        if device_input:
            d_a, d_o, d_hits = c.malloc(total_b + 16), c.malloc(offs.nbytes), c.malloc(n * 8)
            c.h2d(d_a, a)
            c.h2d(d_o, offs)
            t_i = c.bloom_insert_ptr(d_a, n, 0, 0, k, m, d_f, n_bits, offsets=d_o)
            t_q, found = c.bloom_query_ptr(d_a, n, 0, 0, k, m, d_f, n_bits, hits=d_hits, offsets=d_o)
            hits = np.zeros(n, np.uint64)
            c.d2h(hits, d_hits)
            t_c = c.count_insert_ptr(d_a, n, 0, 0, k, m, d_c, n_cnt, offsets=d_o)
            for p_ in (d_a, d_o, d_hits):
                c.free(p_)
        else:
            t_i = c.bloom_insert(a, k, m, 0, n, d_f, n_bits, offsets=offs)
            hits, t_q, found = c.bloom_query(a, k, m, 0, n, d_f, n_bits, offsets=offs)
            t_c = c.count_insert(a, k, m, 0, n, d_c, n_cnt, offsets=offs)
        filt, cnt = np.zeros(nbytes, np.uint8), np.zeros(n_cnt, np.uint8)
        c.d2h(filt, d_f)
        c.d2h(cnt, d_c)
        c.free(d_f)
        c.free(d_c)
        sig, t_s = c.minhash(a, k, m, 0, n, offsets=offs)
        mz = c.minimizers(a, k, w, 0, n, offsets=offs, device_input=device_input)
        res.append(dict(t=(t_i, t_q, found, t_c, t_s, mz["total"]), filt=filt, cnt=cnt, hits=hits, sig=sig, mz=mz))
    one, many = res
    want = oracle.kmer_batch(a, offs, k, m, want_pos=False)
    assert one["t"][0] == want["total"] and one["t"] == many["t"]
    assert (one["hits"] == want["counts"]).all()        # (every k-mer of the batch is in the filter)
    for key in ("filt", "cnt", "hits", "sig"):
        assert (one[key] == many[key]).all(), key
    for key in ("offsets", "pos", "hashes"):
        assert (one["mz"][key] == many["mz"][key]).all(), key
    small.close()


@pytest.mark.parametrize("n,lmax,k,m,n_bits,device_input", [
    (1500, 250, 31, 1, 1 << 22, False), (1200, 180, 25, 3, 3_000_017, True), (600, 400, 64, 2, (1 << 23) + 5, False),
    (40, 6000, 31, 2, 1 << 21, True),
])
def test_consumers_take_reads_given_by_offsets(ctx, oracle, n, lmax, k, m, n_bits, device_input):
    """Bloom insert / query and the counting sketch on reads of any lengths (offsets): the batch's compact stream in one
    round, then the stream consumers -- filter, per-read hits and counters against the oracle's stream (empty reads, reads
    shorter than k, reads with non-bases)"""
    from nthash_amd.capi import NTHIP_HOST_OUTPUT
    rng = np.random.default_rng(n + lmax + m)
    lens = rng.integers(0, lmax + 1, n).astype(np.uint64)
    lens[:3] = [0, k - 1, k]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    total_b = int(offs[-1])
    a = oracle.synth_reads(5, 1, total_b, 3 + k).copy()
    b = a.copy()                                    # the query batch: the second half is other sequence, some N
    half = int(offs[n // 2])
    b[half:] = oracle.synth_reads(6, 1, total_b, 4 + k)[half:]
    bad = rng.choice(total_b, max(3, total_b // 700), replace=False)
    a[bad[::2]] = ord("N")
    b[bad[1::2]] = ord("n")
    ha = oracle.kmer_batch(a, offs, k, m, want_pos=False)
    hb = oracle.kmer_batch(b, offs, k, m, want_pos=False)
    filt = _bloom_expected(ha["hashes"], n_bits)
    d_f, nbytes = ctx.bloom_new(n_bits)

    def dev(arr):
        d = ctx.malloc(max(16, arr.nbytes + 16))
        ctx.h2d(d, arr)
        return d
    if device_input:
        d_a, d_b, d_o = dev(a), dev(b), dev(offs)
        assert ctx.bloom_insert_ptr(d_a, n, 0, 0, k, m, d_f, n_bits, offsets=d_o) == ha["total"]
    else:
        assert ctx.bloom_insert(a, k, m, 0, n, d_f, n_bits, offsets=offs) == ha["total"]
    got = np.zeros(nbytes, np.uint8)
    ctx.d2h(got, d_f)
    assert (got == filt).all()
    bits = np.unpackbits(filt, bitorder="little")
    present = bits[(hb["hashes"] % np.uint64(n_bits)).astype(np.int64)].reshape(-1, m).all(axis=1)
    read_of = np.repeat(np.arange(n), hb["counts"].astype(np.int64))
    want_hits = np.bincount(read_of[present], minlength=n).astype(np.uint64)
    if device_input:
        d_hits = ctx.malloc(n * 8)
        total, found = ctx.bloom_query_ptr(d_b, n, 0, 0, k, m, d_f, n_bits, hits=d_hits, offsets=d_o)
        hits = np.zeros(n, np.uint64)
        ctx.d2h(hits, d_hits)
        ctx.free(d_hits)
    else:
        hits, total, found = ctx.bloom_query(b, k, m, 0, n, d_f, n_bits, offsets=offs)
    assert total == hb["total"] and found == int(want_hits.sum())
    assert (hits == want_hits).all()
    # the same question per k-mer on the stream (nthip_stream_bloom_query)
    hs = np.ascontiguousarray(hb["hashes"]).ravel()
    if hs.size:
        d_hs, d_fl = ctx.malloc(hs.nbytes), ctx.malloc(hb["total"] + 16)
        ctx.h2d(d_hs, hs)
        assert ctx.stream_bloom_query_ptr(d_hs, hb["total"], m, d_f, n_bits, d_fl) == int(present.sum())
        fl = np.zeros(hb["total"], np.uint8)
        ctx.d2h(fl, d_fl)
        assert (fl == present.astype(np.uint8)).all()
        ctx.free(d_hs); ctx.free(d_fl)
    # counting sketch
    n_counters = 1 << 18
    d_c = ctx.malloc(n_counters)
    ctx.memset(d_c, 0, n_counters)
    if device_input:
        assert ctx.count_insert_ptr(d_a, n, 0, 0, k, m, d_c, n_counters, offsets=d_o) == ha["total"]
    else:
        assert ctx.count_insert(a, k, m, 0, n, d_c, n_counters, offsets=offs) == ha["total"]
    tally = np.bincount((np.ascontiguousarray(ha["hashes"]).ravel() % np.uint64(n_counters)).astype(np.int64), minlength=n_counters)
    cnt = np.zeros(n_counters, np.uint8)
    ctx.d2h(cnt, d_c)
    assert (cnt == np.minimum(tally, 255).astype(np.uint8)).all()
    # per-read MinHash signatures
    sig, tot = ctx.minhash(a, k, m, 0, n, offsets=offs)
    assert tot == ha["total"]
    assert (sig == _minhash_expected(ha, n, m)).all()
    for d in ([d_a, d_b, d_o] if device_input else []) + [d_f, d_c]:
        ctx.free(d)


def test_bloom_argument_errors(ctx):
    import nthash_amd
    d_f, _ = ctx.bloom_new(1024)
    data = np.frombuffer(b"ACGT" * 50, dtype=np.uint8)
    with pytest.raises(nthash_amd.NtHipError):
        ctx.bloom_insert(data, 31, 1, 200, 1, 0, 1024)             # NULL filter
    with pytest.raises(nthash_amd.NtHipError):
        ctx.bloom_insert(data, 31, 1, 200, 1, d_f + 1, 1024)       # unaligned filter
    with pytest.raises(nthash_amd.NtHipError):
        ctx.bloom_insert(data, 31, 1, 200, 1, d_f, 0)              # empty filter
    with pytest.raises(nthash_amd.NtHipError):
        ctx.bloom_insert(data, 0, 1, 200, 1, d_f, 1024)            # k == 0
    assert ctx.bloom_insert(data, 31, 1, 20, 10, d_f, 1024) == 0    # reads shorter than k: nothing consumed
    ctx.free(d_f)


def _minhash_expected(want, n, m):
    """per-read minimum of every hash column of the oracle's stream; all ones for a read without k-mers"""
    hs = np.ascontiguousarray(want["hashes"]).reshape(-1, m)
    counts = want["counts"].astype(np.int64)
    sig = np.full((n, m), np.uint64(0xFFFFFFFFFFFFFFFF), np.uint64)
    start = np.concatenate(([0], np.cumsum(counts)))[:-1]
    has = counts > 0
    if has.any():
        sig[has] = np.minimum.reduceat(hs, start[has], axis=0)
    return sig


@pytest.mark.parametrize("n,L,k,m,bad_every", [
    (3000, 150, 31, 1, 0), (3000, 150, 31, 1, 211), (2500, 150, 31, 4, 0), (1500, 151, 31, 3, 97),
    (1000, 100, 64, 3, 401), (700, 250, 21, 8, 53), (500, 149, 31, 12, 307), (60, 5003, 31, 2, 1009),
    (300, 400, 101, 9, 997), (4000, 35, 31, 2, 41), (129, 31, 31, 5, 7), (900, 64, 5, 1, 3),
])
def test_minhash_signatures_match_oracle_stream(ctx, oracle, n, L, k, m, bad_every):
    """fused consumer: signatures[r][i] == min over the oracle's k-mers of read r of hash i (bit-exact;
    min is order-free); reads without a valid k-mer keep UINT64_MAX"""
    data = oracle.synth_reads(4, n, L, 1234 + L).copy()
    if bad_every:
        data[bad_every // 2::bad_every] = ord("N")
        data[L * (n // 3): L * (n // 3 + 1)] = ord("n")     # one read without any k-mer
    offs = np.arange(n + 1, dtype=np.uint64) * L
    want = oracle.kmer_batch(data, offs, k, m, want_pos=False)
    sig, total = ctx.minhash(data, k, m, L, n)
    assert total == want["total"]
    expect = _minhash_expected(want, n, m)
    assert (sig == expect).all()


@pytest.mark.parametrize("n,L,k,m,stride", [(13529, 92, 55, 3, 38), (20000, 151, 31, 2, 0), (9000, 100, 64, 1, 37)])
def test_minhash_sparse_non_bases_across_tile_boundaries(ctx, oracle, n, L, k, m, stride):
    """a read whose runs sit in two wave tiles, with a non-base only in the part the second tile recomputes:
    the recomputed windows must not enter the minimum (found by tools/stress_shapes.py)"""
    rng = np.random.default_rng(n)
    total = n * L if stride == 0 else (n - 1) * stride + L
    data = np.frombuffer(b"ACGTacgtUu", dtype=np.uint8)[rng.integers(0, 10, total)]
    data[rng.integers(0, total, total // 2500)] = ord("N")
    step = stride if stride else L
    reads = np.lib.stride_tricks.as_strided(data, (n, L), (step, 1)).copy()
    offs = np.arange(n + 1, dtype=np.uint64) * L
    want = oracle.kmer_batch(reads.ravel(), offs, k, m, want_pos=False)
    sig, total_k = ctx.minhash(data, k, m, L, n, stride=stride)
    assert total_k == want["total"]
    assert (sig == _minhash_expected(want, n, m)).all()


def test_minhash_device_io_stride_and_errors(ctx, oracle):
    import nthash_amd
    from nthash_amd.capi import NTHIP_HOST_INPUT
    n, L, stride, k, m = 800, 120, 128, 25, 3
    rows = oracle.synth_reads(9, n, L, 77).reshape(n, L)
    padded = np.full((n, stride), ord("N"), np.uint8)
    padded[:, :L] = rows
    offs = np.arange(n + 1, dtype=np.uint64) * L
    want = oracle.kmer_batch(rows.ravel(), offs, k, m, want_pos=False)
    d_in = ctx.malloc(padded.size)
    ctx.h2d(d_in, padded.ravel())
    d_sig = ctx.malloc(n * m * 8)
    total = ctx.minhash_ptr(d_in, n, L, stride, k, m, d_sig)
    got = np.zeros((n, m), np.uint64)
    ctx.d2h(got, d_sig)
    assert total == want["total"] and (got == _minhash_expected(want, n, m)).all()
    ctx.free(d_in)
    ctx.free(d_sig)
    short, tot = ctx.minhash(rows.ravel()[: 10 * 20], 31, 2, 20, 10)          # reads shorter than k
    assert tot == 0 and (short == np.uint64(0xFFFFFFFFFFFFFFFF)).all()
    with pytest.raises(nthash_amd.NtHipError):
        ctx.minhash(rows.ravel(), 0, 1, L, n)
    with pytest.raises(nthash_amd.NtHipError):
        ctx.minhash_ptr(rows.ctypes.data, n, L, 0, k, m, 0, flags=NTHIP_HOST_INPUT)    # NULL signatures


# ---------------------------------------------------------------------------
# FASTQ / FASTA -> device batches: device indexer, spans, file streaming
# ---------------------------------------------------------------------------
def _make_fastx(rng, n, fmt, crlf=False, lo=0, hi=320, bad_frac=0.01):
    """(file bytes, list of sequences) -- variable-length reads, some shorter than k, some with N"""
    alph = np.frombuffer(b"ACGTacgtN", dtype=np.uint8)
    nl = b"\r\n" if crlf else b"\n"
    parts, seqs = [], []
    for i in range(n):
        L = int(rng.integers(lo, hi))
        idx = np.where(rng.random(L) < bad_frac, 8, rng.integers(0, 8, L))
        sq = alph[idx].tobytes()
        seqs.append(sq)
        if fmt == 4:
            qual = bytes(rng.integers(33, 74, L, dtype=np.uint8))   # may start with '@' or '+'
            parts.append(b"@r%d some text" % i + nl + sq + nl + b"+" + nl + qual + nl)
        else:
            parts.append(b">s%d" % i + nl + sq + nl)
    return b"".join(parts), seqs


def _py_spans(buf, fmt):
    """reference parser: spans of the sequence lines of the complete records, consumed bytes"""
    starts, ends, pos, consumed = [], [], 0, 0
    while True:
        lines = []
        p = pos
        for _ in range(fmt):
            q = buf.find(b"\n", p)
            if q < 0:
                lines = None
                break
            lines.append((p, q))
            p = q + 1
        if lines is None:
            break
        s, e = lines[1]
        if e > s and buf[e - 1:e] == b"\r":
            e -= 1
        starts.append(s)
        ends.append(e)
        pos = consumed = p
    return np.array(starts, np.uint64), np.array(ends, np.uint64), consumed


@pytest.mark.parametrize("fmt,crlf,cut", [(4, False, 0), (4, True, 0), (2, False, 0), (4, False, 777), (2, True, 5)])
def test_fastx_index_and_spans_vs_python_parser(ctx, oracle, fmt, crlf, cut):
    rng = np.random.default_rng(fmt * 10 + cut)
    buf, seqs = _make_fastx(rng, 3000, fmt, crlf)
    if cut:
        buf = buf[: len(buf) - cut]           # the chunk ends inside a record
    raw = np.frombuffer(buf, dtype=np.uint8)
    w_starts, w_ends, w_consumed = _py_spans(buf, fmt)
    d_buf = ctx.malloc(raw.size + 64)
    ctx.h2d(d_buf + 3, raw)                    # not even 4-byte aligned
    cap = w_starts.size + 5
    d_s, d_e = ctx.malloc(cap * 8), ctx.malloc(cap * 8)
    n_rec, consumed, bad = ctx.fastx_index_ptr(d_buf + 3, raw.size, fmt, d_s, d_e, cap)
    assert (n_rec, consumed, bad) == (w_starts.size, w_consumed, 0)
    g_s, g_e = np.zeros(n_rec, np.uint64), np.zeros(n_rec, np.uint64)
    ctx.d2h(g_s, d_s)
    ctx.d2h(g_e, d_e)
    assert (g_s == w_starts).all() and (g_e == w_ends).all()
    # hash the sequence lines where they lie; oracle on the parsed reads
    data, offs = concat_reads(seqs[:n_rec])
    for k, m in ((31, 2), (64, 1), (5, 3)):
        want = oracle.kmer_batch(data, offs, k, m)
        capk = max(int(want["total"]), 1)
        d_h, d_c, d_p = ctx.malloc(capk * m * 8), ctx.malloc(n_rec * 8), ctx.malloc(capk * 4)
        tot = ctx.kmer_hash_spans_ptr(d_buf + 3, raw.size, d_s, d_e, n_rec, k, m, d_h, capk, counts=d_c, pos=d_p)
        assert tot == want["total"]
        h, cnt, pos = np.zeros(tot * m, np.uint64), np.zeros(n_rec, np.uint64), np.zeros(tot, np.uint32)
        ctx.d2h(h, d_h); ctx.d2h(cnt, d_c); ctx.d2h(pos, d_p)
        assert (h == want["hashes"].ravel()).all() and (cnt == want["counts"]).all() and (pos == want["pos"]).all()
        for d in (d_h, d_c, d_p):
            ctx.free(d)
    for d in (d_buf, d_s, d_e):
        ctx.free(d)


@pytest.mark.parametrize("k,m,n,lo,hi,bad_frac", [(31, 1, 5000, 20, 400, 0.002), (31, 2, 3000, 100, 151, 0.0), (21, 1, 4000, 1, 90, 0.01),
                                                   (64, 3, 1500, 60, 300, 0.001), (100, 1, 800, 90, 500, 0.002), (25, 1, 64, 150, 151, 0.05)])
def test_kmer_read_slots_contract_vs_oracle(ctx, oracle, k, m, n, lo, hi, bad_frac):
    """NTHIP_OUT_READ_SLOTS: one pass over the bases -- read r's k-mers at the front of the slot its LENGTH implies, the
    exact count in counts[r], zeros behind, *total = every window of every read -- through the offsets entry (reads back
    to back) and the spans entry (sequence lines of a FASTQ text: headers and qualities between the reads, never bases);
    reads with non-bases, reads shorter than k, empty reads; positions"""
    import nthash_amd
    from nthash_amd.capi import NTHIP_OUT_READ_SLOTS
    rng = np.random.default_rng(k * 7 + n)
    buf, seqs = _make_fastx(rng, n, 4, False, lo=lo, hi=hi, bad_frac=bad_frac)
    seqs[n // 3] = b""
    data, offs = concat_reads(seqs)
    want = oracle.kmer_batch(data, offs, k, m)
    lens = np.diff(offs.astype(np.int64))
    nwin = np.maximum(lens - k + 1, 0)
    slot = np.concatenate([[0], np.cumsum(nwin)])
    w_off = np.concatenate([[0], np.cumsum(want["counts"].astype(np.int64))])

    def check(got_h, got_c, got_p, total):
        assert total == slot[-1]
        assert (got_c == want["counts"]).all()
        got_h = got_h.reshape(-1, m)
        for r in range(n):
            c = int(want["counts"][r])
            a = int(slot[r])
            assert (got_h[a:a + c] == want["hashes"][w_off[r]:w_off[r] + c]).all(), r
            assert (got_p[a:a + c] == want["pos"][w_off[r]:w_off[r] + c]).all(), r
            assert (got_h[a + c:int(slot[r + 1])] == 0).all(), r

    cap = int(slot[-1])
    h, c_, p_ = np.zeros(max(cap, 1) * m, np.uint64), np.zeros(n, np.uint64), np.zeros(max(cap, 1), np.uint32)
    flags = nthash_amd.capi.NTHIP_HOST_INPUT | nthash_amd.capi.NTHIP_HOST_OUTPUT | NTHIP_OUT_READ_SLOTS
    total = ctx.kmer_hash_ptr(data.ctypes.data, offs.ctypes.data, n, 0, 0, k, m, h.ctypes.data, cap, counts=c_.ctypes.data,
                              pos=p_.ctypes.data, flags=flags)
    check(h, c_, p_, total)
    with pytest.raises(nthash_amd.NtHipError) as ei:   # the slot array must fit
        ctx.kmer_hash_ptr(data.ctypes.data, offs.ctypes.data, n, 0, 0, k, m, h.ctypes.data, cap - 1, counts=c_.ctypes.data, flags=flags)
    assert ei.value.code == nthash_amd.capi.NTHIP_ERR_CAPACITY and ei.value.total == cap
    # the same reads where they lie in the FASTQ text
    st2 = np.zeros(n, np.uint64)
    en2 = np.zeros(n, np.uint64)
    at = 0
    for r, sq in enumerate(seqs):
        at = buf.index(b"\n", at) + 1          # past the header line
        st2[r], en2[r] = at, at + len(sq)
        if r == n // 3:                          # (the record whose sequence we emptied: an empty span inside its line)
            en2[r] = at
        for _ in range(3):
            at = buf.index(b"\n", at) + 1
    raw = np.frombuffer(buf, dtype=np.uint8)
    d_buf, d_s, d_e = ctx.malloc(raw.size + 64), ctx.malloc(n * 8), ctx.malloc(n * 8)
    d_h, d_c, d_p = ctx.malloc(max(cap, 1) * m * 8), ctx.malloc(n * 8), ctx.malloc(max(cap, 1) * 4)
    ctx.h2d(d_buf, raw); ctx.h2d(d_s, st2); ctx.h2d(d_e, en2)
    total = ctx.kmer_hash_spans_ptr(d_buf, raw.size, d_s, d_e, n, k, m, d_h, cap, counts=d_c, pos=d_p, flags=NTHIP_OUT_READ_SLOTS)
    h2, c2, p2 = np.zeros(max(cap, 1) * m, np.uint64), np.zeros(n, np.uint64), np.zeros(max(cap, 1), np.uint32)
    ctx.d2h(h2, d_h); ctx.d2h(c2, d_c); ctx.d2h(p2, d_p)
    check(h2, c2, p2, total)
    for d in (d_buf, d_s, d_e, d_h, d_c, d_p):
        ctx.free(d)


@pytest.mark.parametrize("n,L,k,m,bad_every", [
    (4000, 150, 31, 1, 997),      # the headline kernel (burst path: pieces of 8 tiles = 64 reads)
    (4000, 150, 31, 4, 500),      # compile-time m = 4
    (3000, 151, 31, 1, 400),      # the run-length-11 instantiation of the headline kernel
    (3000, 100, 64, 3, 300),      # general kernel, forward-half tables
    (2500, 250, 21, 2, 700), (1500, 150, 100, 1, 200), (900, 301, 200, 2, 150),   # general kernel; k beyond the position tables
    (700, 36, 21, 1, 50), (64, 2048, 31, 1, 5),
    (3000, 150, 31, 1, 0),        # a clean batch: nothing is redone
    (2000, 150, 31, 1, 3),        # a non-base in most reads
])
def test_kmer_read_slots_fixed_length_vs_oracle(ctx, oracle, n, L, k, m, bad_every):
    """NTHIP_OUT_READ_SLOTS on fixed-length reads (offsets == NULL): slot r = r * (L - k + 1) holds the k-mers NtHash emits
    for read r at its front, counts[r] says how many, zeros behind; *total = n * (L - k + 1).  The dense kernels run as if
    the batch were clean and mark the 16-byte vectors that hold a non-base; the reads those touch are redone afterwards.
    Non-bases at read ends, in neighbouring reads, in the first and the last read, several per read; positions"""
    import nthash_amd
    from nthash_amd.capi import NTHIP_OUT_READ_SLOTS, NTHIP_HOST_INPUT, NTHIP_HOST_OUTPUT
    rng = np.random.default_rng(n + L + k)
    data = oracle.synth_reads(4, n, L, 11 + k).copy()
    if bad_every:
        n_bad = max(4, n * L // (bad_every * L))
        where = rng.choice(n * L, n_bad, replace=False)
        data[where] = np.frombuffer(b"NnRY-.", dtype=np.uint8)[rng.integers(0, 6, n_bad)]
        data[0] = ord("N")                      # the first byte of the batch
        data[n * L - 1] = ord("N")              # the last one
        data[5 * L - 1] = ord("N")              # the last base of a read ...
        data[5 * L] = ord("N")                  # ... and the first of the next
        data[(n // 2) * L + k - 1] = ord("N")
        data[(n // 2) * L + k + 3] = ord("n")   # two in one read
    offs = np.arange(n + 1, dtype=np.uint64) * L
    want = oracle.kmer_batch(data, offs, k, m)
    nwin = L - k + 1
    cap = n * nwin
    h, c_, p_ = np.full(cap * m, 0xAB, np.uint64), np.zeros(n, np.uint64), np.full(cap, 7, np.uint32)
    flags = NTHIP_HOST_INPUT | NTHIP_HOST_OUTPUT | NTHIP_OUT_READ_SLOTS
    total = ctx.kmer_hash_ptr(data.ctypes.data, 0, n, L, 0, k, m, h.ctypes.data, cap, counts=c_.ctypes.data, pos=p_.ctypes.data,
                              flags=flags)
    assert total == cap
    assert (c_ == want["counts"]).all()
    hh = h.reshape(n, nwin, m)
    pp = p_.reshape(n, nwin)
    w_off = np.concatenate([[0], np.cumsum(want["counts"].astype(np.int64))])
    wh = want["hashes"].reshape(-1, m)
    for r in range(n):
        c = int(want["counts"][r])
        assert (hh[r, :c] == wh[w_off[r]:w_off[r] + c]).all(), r
        assert (pp[r, :c] == want["pos"][w_off[r]:w_off[r] + c]).all(), r
        assert (hh[r, c:] == 0).all(), r
    # without positions; the slot array must fit
    h2, c2 = np.zeros(cap * m, np.uint64), np.zeros(n, np.uint64)
    assert ctx.kmer_hash_ptr(data.ctypes.data, 0, n, L, 0, k, m, h2.ctypes.data, cap, counts=c2.ctypes.data, flags=flags) == cap
    assert (h2 == h).all() and (c2 == c_).all()
    with pytest.raises(nthash_amd.NtHipError) as ei:
        ctx.kmer_hash_ptr(data.ctypes.data, 0, n, L, 0, k, m, h2.ctypes.data, cap - 1, counts=c2.ctypes.data, flags=flags)
    assert ei.value.code == nthash_amd.capi.NTHIP_ERR_CAPACITY and ei.value.total == cap
    with pytest.raises(nthash_amd.NtHipError):   # counts are part of the contract
        ctx.kmer_hash_ptr(data.ctypes.data, 0, n, L, 0, k, m, h2.ctypes.data, cap, flags=flags)


def test_fastx_index_flags_malformed_input(ctx):
    for buf, fmt in ((b"@a\nACGT\n-\nIIII\n", 4), (b"a\nACGT\n+\nIIII\n", 4), (b"@a\nAC\n+\nII\nxx\nAC\n+\nII\n", 4),
                     (b">a\nACGT\nACGT\n>b\nAC\n", 2)):
        raw = np.frombuffer(buf, dtype=np.uint8)
        d_buf, d_s, d_e = ctx.malloc(raw.size + 16), ctx.malloc(64), ctx.malloc(64)
        ctx.h2d(d_buf, raw)
        _n, _c, bad = ctx.fastx_index_ptr(d_buf, raw.size, fmt, d_s, d_e, 8)
        assert bad != 0, buf
        for d in (d_buf, d_s, d_e):
            ctx.free(d)


@pytest.mark.parametrize("fmt,chunk,final_newline", [(4, 1 << 16, True), (4, 100_000, False), (2, 1 << 16, True),
                                                     (4, 1 << 22, True)])
def test_fastx_file_stream_vs_oracle(ctx, oracle, tmp_path, fmt, chunk, final_newline):
    """a file far larger than the chunk size: many batches, records straddling every chunk boundary"""
    rng = np.random.default_rng(chunk % 1000 + fmt)
    buf, seqs = _make_fastx(rng, 6000, fmt, lo=20, hi=400)
    if not final_newline:
        buf = buf[:-1]
    path = tmp_path / ("reads.fq" if fmt == 4 else "reads.fa")
    path.write_bytes(buf)
    k, m = 31, 2
    data, offs = concat_reads(seqs)
    want = oracle.kmer_batch(data, offs, k, m, want_pos=False)
    got_h, got_c, firsts = [], [], []

    def on_batch(b):
        h, cnt = np.zeros(b.n_kmers * m, np.uint64), np.zeros(b.n_reads, np.uint64)
        if h.size:
            ctx.d2h(h, b.hashes)
        ctx.d2h(cnt, b.counts)
        got_h.append(h); got_c.append(cnt); firsts.append(b.first_read)

    st = ctx.fastx_kmer_hash_file(path, fmt, k, m, chunk_bytes=chunk, on_batch=on_batch)
    assert st.reads == len(seqs) and st.kmers == want["total"] and st.file_bytes == len(buf)
    if chunk < len(buf) // 4:
        assert st.batches > 4
    assert firsts == list(np.cumsum([0] + [c.size for c in got_c[:-1]]))
    assert (np.concatenate(got_c) == want["counts"]).all()
    assert (np.concatenate(got_h) == want["hashes"].ravel()).all()


@pytest.mark.parametrize("n,lo,hi,k,w", [(3000, 20, 150, 31, 10), (1500, 30, 280, 25, 19), (400, 100, 900, 31, 12)])
def test_minimizers_of_spans_of_a_raw_fastq_buffer(ctx, oracle, n, lo, hi, k, w):
    """nthip_kmer_minimizers_spans on the sequence lines of a raw FASTQ buffer (indexed on the device) == nthip_kmer_minimizers
    on the parsed reads given by offsets (itself tested against the brute force): register tables (at most 256 windows) and
    LDS tables"""
    rng = np.random.default_rng(n + hi)
    buf, seqs = _make_fastx(rng, n, 4, lo=lo, hi=hi)
    data, offs = concat_reads(seqs)
    want = ctx.minimizers(data, k, w, 0, n, offsets=offs)
    raw = np.frombuffer(buf, dtype=np.uint8)
    d_buf, d_s, d_e = ctx.malloc(raw.size + 16), ctx.malloc(n * 8 + 8), ctx.malloc(n * 8 + 8)
    ctx.h2d(d_buf, raw)
    n_rec, _cons, bad = ctx.fastx_index_ptr(d_buf, raw.size, 4, d_s, d_e, n)
    assert n_rec == n and not bad
    cap = max(int(data.size), 1)
    d_h, d_p, d_o = ctx.malloc(cap * 8), ctx.malloc(cap * 4), ctx.malloc((n + 1) * 8)
    total = ctx.minimizers_spans_ptr(d_buf, raw.size, d_s, d_e, n, k, w, d_h, d_p, d_o, cap)
    assert total == want["total"]
    o, h, p = np.zeros(n + 1, np.uint64), np.zeros(total, np.uint64), np.zeros(total, np.uint32)
    ctx.d2h(o, d_o); ctx.d2h(h, d_h); ctx.d2h(p, d_p)
    assert (o == want["offsets"]).all() and (h == want["hashes"]).all() and (p == want["pos"]).all()
    for d in (d_buf, d_s, d_e, d_h, d_p, d_o):
        ctx.free(d)


def test_fastx_file_batches_into_the_stream_consumers(ctx, oracle, tmp_path):
    """a FASTQ file end to end into a Bloom filter and a counting sketch: the stream consumers called on every batch from
    inside the driver's callback, with the driver's own context -- filter and counters == those of the
    oracle's stream over the parsed reads"""
    rng = np.random.default_rng(321)
    buf, seqs = _make_fastx(rng, 5000, 4, lo=20, hi=300)
    path = tmp_path / "reads.fq"
    path.write_bytes(buf)
    k, m, n_bits, n_counters = 31, 2, 3_000_017, 1 << 18
    data, offs = concat_reads(seqs)
    want = oracle.kmer_batch(data, offs, k, m, want_pos=False)
    d_f, nbytes = ctx.bloom_new(n_bits)
    d_c = ctx.malloc(n_counters)
    ctx.memset(d_c, 0, n_counters)

    def on_batch(b):
        if b.n_kmers:
            ctx.stream_bloom_insert_ptr(b.hashes, b.n_kmers * m, d_f, n_bits)
            ctx.stream_count_insert_ptr(b.hashes, b.n_kmers * m, d_c, n_counters)

    st = ctx.fastx_kmer_hash_file(path, 4, k, m, chunk_bytes=1 << 17, on_batch=on_batch)
    assert st.kmers == want["total"] and st.batches > 2
    got = np.zeros(nbytes, np.uint8)
    ctx.d2h(got, d_f)
    assert (got == _bloom_expected(want["hashes"], n_bits)).all()
    tally = np.bincount((np.ascontiguousarray(want["hashes"]).ravel() % np.uint64(n_counters)).astype(np.int64), minlength=n_counters)
    cnt = np.zeros(n_counters, np.uint8)
    ctx.d2h(cnt, d_c)
    assert (cnt == np.minimum(tally, 255).astype(np.uint8)).all()
    ctx.free(d_f); ctx.free(d_c)


@pytest.mark.parametrize("fmt,crlf,chunk,final_newline,devices", [
    (4, False, 1 << 16, True, [0, 0, 0]), (4, True, 70_000, False, [0, 0]), (2, False, 1 << 16, True, [0, 0, 0, 0]),
    (4, False, 1 << 22, True, [0, 0, 0]), (4, False, 1 << 16, True, None)])
def test_multi_device_fastx_file_vs_oracle(ctx, oracle, tmp_path, fmt, crlf, chunk, final_newline, devices):
    """nthip_multi_fastx_kmer_hash_file: the file cut into record-aligned pieces on the host (quality lines that begin
    with '@' or '+' must not fool the cut), piece j on device j mod N -- the box's one GPU listed several times: separate
    contexts, reader threads and pinned rings, the code an 8-GPU node runs --, every batch delivered once, in file order,
    with first_read counting through the file; hashes and counts == the oracle on the parsed reads"""
    import nthash_amd
    rng = np.random.default_rng(chunk % 977 + fmt + (len(devices) if devices else 9))
    buf, seqs = _make_fastx(rng, 7000, fmt, crlf, lo=20, hi=400)
    if not final_newline:
        buf = buf[:-2] if crlf else buf[:-1]
    path = tmp_path / ("reads.fq" if fmt == 4 else "reads.fa")
    path.write_bytes(buf)
    k, m = 31, 2
    data, offs = concat_reads(seqs)
    want = oracle.kmer_batch(data, offs, k, m, want_pos=False)
    got_h, got_c, firsts, devs = [], [], [], []

    def on_batch(b):
        h, cnt = np.zeros(b.n_kmers * m, np.uint64), np.zeros(b.n_reads, np.uint64)
        if h.size:
            ctx.d2h(h, b.hashes)
        ctx.d2h(cnt, b.counts)
        got_h.append(h); got_c.append(cnt); firsts.append(b.first_read); devs.append(b.device)

    mg = nthash_amd.Multi(devices)
    try:
        st = mg.fastx_kmer_hash_file(path, fmt, k, m, chunk_bytes=chunk, on_batch=on_batch)
    finally:
        mg.close()
    assert st.reads == len(seqs) and st.kmers == want["total"] and st.file_bytes == len(buf)
    if chunk < len(buf) // 8:
        assert st.batches >= 8
    assert firsts == list(np.cumsum([0] + [c.size for c in got_c[:-1]]))
    assert set(devs) == {0}
    assert (np.concatenate(got_c) == want["counts"]).all()
    assert (np.concatenate(got_h) == want["hashes"].ravel()).all()


def test_multi_device_fastx_file_errors(ctx, tmp_path):
    import nthash_amd
    mg = nthash_amd.Multi([0, 0])
    try:
        with pytest.raises(nthash_amd.NtHipError):
            mg.fastx_kmer_hash_file(tmp_path / "nope.fq", 4, 31, 1)
        bad = tmp_path / "bad.fq"
        bad.write_bytes(b"@a\nACGT\n+\nIIII\n" * 3000 + b"@b\nACGT\n-\nIIII\n" + b"@a\nACGT\n+\nIIII\n" * 3000)
        with pytest.raises(nthash_amd.NtHipError):
            mg.fastx_kmer_hash_file(bad, 4, 31, 1, chunk_bytes=1 << 16)
        seen = []

        def stop_after_two(b):
            seen.append(b.first_read)
            if len(seen) == 2:
                raise RuntimeError("enough")
        ok = tmp_path / "ok.fq"
        ok.write_bytes(b"@a\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n" * 20000)
        with pytest.raises(RuntimeError):
            mg.fastx_kmer_hash_file(ok, 4, 31, 1, chunk_bytes=1 << 16, on_batch=stop_after_two)
        assert len(seen) == 2
    finally:
        mg.close()


def test_fastx_file_stream_reuses_and_trims_context_buffers(ctx, oracle, tmp_path):
    """the streaming buffers live in the context: same file again (reuse), bigger chunks (regrow), after
    nthip_ctx_trim (fresh) and with more hashes per k-mer (regrow of the hash buffer) -- same stream every time"""
    rng = np.random.default_rng(77)
    buf, seqs = _make_fastx(rng, 3000, 4, lo=20, hi=300)
    path = tmp_path / "again.fq"
    path.write_bytes(buf)
    data, offs = concat_reads(seqs)
    for m, chunk, trim in ((1, 1 << 16, False), (1, 1 << 16, False), (1, 1 << 20, False), (1, 1 << 16, True),
                           (3, 1 << 16, False), (1, 1 << 18, True)):
        if trim:
            ctx.trim()
        want = oracle.kmer_batch(data, offs, 31, m, want_pos=False)
        got = []

        def on_batch(b, m=m, got=got):
            h = np.zeros(b.n_kmers * m, np.uint64)
            if h.size:
                ctx.d2h(h, b.hashes)
            got.append(h)

        st = ctx.fastx_kmer_hash_file(path, 4, 31, m, chunk_bytes=chunk, on_batch=on_batch)
        assert st.reads == len(seqs) and st.kmers == want["total"]
        assert (np.concatenate(got) == want["hashes"].ravel()).all()
    # the batch paths still work after a trim (their scratch is reallocated on demand)
    ctx.trim()
    one = ctx.kmer_hash(data, 31, 1, offsets=offs)
    assert one["total"] == oracle.kmer_batch(data, offs, 31, 1, want_pos=False)["total"]


def test_fastx_file_errors(ctx, tmp_path):
    import nthash_amd
    with pytest.raises(nthash_amd.NtHipError):
        ctx.fastx_kmer_hash_file(tmp_path / "missing.fq", 4, 31, 1)
    p = tmp_path / "bad.fq"
    p.write_bytes(b"@a\nACGT\n+\nIIII\n@b\nACGT\n+\n")       # truncated last record
    with pytest.raises(nthash_amd.NtHipError):
        ctx.fastx_kmer_hash_file(p, 4, 3, 1)
    p.write_bytes(b"")
    assert ctx.fastx_kmer_hash_file(p, 4, 31, 1).reads == 0


def _make_multiline_fasta(rng, n_rec, crlf=False, max_len=5000, blank_lines=True):
    alph = np.frombuffer(b"ACGTacgtNn", dtype=np.uint8)
    nl = b"\r\n" if crlf else b"\n"
    parts, seqs = [], []
    for i in range(n_rec):
        L = int(rng.integers(0, max_len)) if rng.random() < 0.9 else 0
        sq = alph[np.where(rng.random(L) < 0.01, rng.integers(8, 10, L), rng.integers(0, 8, L))].tobytes()
        seqs.append(sq)
        width = int(rng.choice([1, 7, 60, 61, 63, 64, 65, 80, 127, 128, 129, 1000]))
        parts.append(b">chr%d some > text with \t tabs" % i + nl)
        for p in range(0, L, width):
            parts.append(sq[p:p + width] + nl)
            if blank_lines and rng.random() < 0.02:
                parts.append(nl)
    return b"".join(parts), seqs


@pytest.mark.parametrize("n_rec,crlf,max_len", [(300, False, 5000), (200, True, 3000), (3, False, 400_000), (1, False, 70),
                                                (2000, False, 40)])
def test_fasta_multiline_compaction_vs_python(ctx, oracle, n_rec, crlf, max_len):
    rng = np.random.default_rng(n_rec + max_len)
    buf, seqs = _make_multiline_fasta(rng, n_rec, crlf, max_len)
    raw = np.frombuffer(buf, dtype=np.uint8)
    d_raw = ctx.malloc(raw.size + 64)
    ctx.h2d(d_raw + 1, raw)                                   # odd base address
    d_seqs, d_offs = ctx.malloc(raw.size + 64), ctx.malloc((n_rec + 1) * 8)
    import nthash_amd
    with pytest.raises(nthash_amd.NtHipError) as ei:           # too small an offsets array: count is reported
        ctx.fasta_compact_ptr(d_raw + 1, raw.size, d_seqs, d_offs, n_rec - 1 if n_rec > 1 else 0)
    assert ei.value.n_records == n_rec
    got_n, got_bytes = ctx.fasta_compact_ptr(d_raw + 1, raw.size, d_seqs, d_offs, n_rec)
    data, offs = concat_reads(seqs)
    assert got_n == n_rec and got_bytes == data.size
    g_offs = np.zeros(n_rec + 1, np.uint64)
    ctx.d2h(g_offs, d_offs)
    assert (g_offs == offs).all()
    if data.size:
        g_data = np.zeros(data.size, np.uint8)
        ctx.d2h(g_data, d_seqs)
        assert (g_data == data).all()
    for d in (d_raw, d_seqs, d_offs):
        ctx.free(d)


def test_fasta_multiline_file_vs_oracle(ctx, oracle, tmp_path):
    from nthash_amd.capi import NTHIP_FASTA_MULTILINE
    rng = np.random.default_rng(77)
    buf, seqs = _make_multiline_fasta(rng, 40, max_len=60_000)
    path = tmp_path / "genome.fa"
    path.write_bytes(buf[:-1])                                  # no newline at the end of the file
    k, m = 31, 2
    data, offs = concat_reads(seqs)
    want = oracle.kmer_batch(data, offs, k, m, want_pos=False)
    got = {}

    def on_batch(b):
        h, cnt = np.zeros(b.n_kmers * m, np.uint64), np.zeros(b.n_reads, np.uint64)
        ctx.d2h(h, b.hashes); ctx.d2h(cnt, b.counts)
        got["h"], got["c"] = h, cnt

    st = ctx.fastx_kmer_hash_file(path, NTHIP_FASTA_MULTILINE, k, m, on_batch=on_batch)
    assert st.reads == len(seqs) and st.kmers == want["total"] and st.batches == 1
    assert (got["c"] == want["counts"]).all() and (got["h"] == want["hashes"].ravel()).all()
    import nthash_amd
    (tmp_path / "bad.fa").write_bytes(b"ACGT\n>x\nAC\n")
    with pytest.raises(nthash_amd.NtHipError):
        ctx.fastx_kmer_hash_file(tmp_path / "bad.fa", NTHIP_FASTA_MULTILINE, k, m)


def _gzip_members(buf, cuts, level=6):
    """buf as a gzip file of len(cuts) + 1 concatenated members (what bgzip / `cat a.gz b.gz` make)"""
    import gzip
    edges = [0] + list(cuts) + [len(buf)]
    return b"".join(gzip.compress(buf[a:b], compresslevel=level) for a, b in zip(edges[:-1], edges[1:]))


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,chunk,members,final_newline,use_seeds", [
    (4, 1 << 16, 1, True, False),        # many chunks of the inflated stream, records across every chunk boundary
    (4, 100_000, 5, False, False),       # concatenated members cut at arbitrary bytes, no newline at the end
    (2, 1 << 16, 3, True, False),
    (4, 1 << 26, 1, True, False),        # the whole stream in one chunk
    (4, 1 << 16, 2, True, True),         # the spaced-seed stream
])
def test_fastx_gzip_file_stream_vs_oracle(ctx, oracle, tmp_path, fmt, chunk, members, final_newline, use_seeds):
    """a gzip-compressed FASTQ / FASTA file (recognised by its magic, whatever its name) through the same pipeline: hashes,
    counts, batch order == the oracle on the parsed reads; and == the plain file's own stream, batch for batch"""
    rng = np.random.default_rng(chunk % 991 + fmt + members)
    buf, seqs = _make_fastx(rng, 6000, fmt, lo=20, hi=400)
    if not final_newline:
        buf = buf[:-1]
    cuts = sorted(int(x) for x in rng.integers(1, len(buf) - 1, members - 1))
    gz = _gzip_members(buf, cuts)
    path = tmp_path / "reads.anything"
    path.write_bytes(gz)
    plain = tmp_path / "reads.plain"
    plain.write_bytes(buf)
    k, m = 31, 2
    data, offs = concat_reads(seqs)
    sd = None
    if use_seeds:
        import nthash_amd
        seeds = ["1010101010101010101010101010101", "1101101101101101011011011011011"]
        want = oracle.seed_batch(data, offs, seeds, k, m, want_pos=False)
        sd = nthash_amd.Seeds(ctx, seeds, k)
        per = len(seeds) * m
    else:
        want = oracle.kmer_batch(data, offs, k, m, want_pos=False)
        per = m

    def run(p):
        got_h, got_c, firsts = [], [], []

        def on_batch(b):
            h, cnt = np.zeros(b.n_kmers * per, np.uint64), np.zeros(b.n_reads, np.uint64)
            if h.size:
                ctx.d2h(h, b.hashes)
            ctx.d2h(cnt, b.counts)
            got_h.append(h); got_c.append(cnt); firsts.append(b.first_read)

        st = ctx.fastx_kmer_hash_file(p, fmt, k, m, chunk_bytes=chunk, on_batch=on_batch, seeds=sd)
        return st, got_h, got_c, firsts

    st, got_h, got_c, firsts = run(path)
    assert st.reads == len(seqs) and st.kmers == want["total"] and st.file_bytes == len(gz)
    if chunk < len(buf) // 4:
        assert st.batches > 4
    assert firsts == list(np.cumsum([0] + [c.size for c in got_c[:-1]]))
    assert (np.concatenate(got_c) == want["counts"]).all()
    assert (np.concatenate(got_h) == want["hashes"].ravel()).all()
    st2, h2, c2, f2 = run(plain)
    assert st2.batches == st.batches and f2 == firsts
    assert all((a == b).all() for a, b in zip(h2, got_h)) and all((a == b).all() for a, b in zip(c2, got_c))


@pytest.mark.gpu
def test_fastx_gzip_chunk_edges_and_errors(ctx, oracle, tmp_path):
    """the inflated stream an exact multiple of the chunk (the last chunk is known as the last by the byte read ahead), an
    empty stream, a file cut short, a corrupt member, a gzip magic with nothing behind it"""
    import gzip
    import nthash_amd
    rec = b"@r\nACGTACGTACGTACGTACGTACGTACGTAC\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n"      # 67 bytes: 30 bases
    assert len(rec) == 67
    chunk = 1 << 16
    recs = []  # 67-byte records and one with a padded header: exactly two chunks
    total = 0
    while total + 2 * len(rec) < 2 * chunk:
        recs.append(rec); total += len(rec)
    pad = 2 * chunk - total - len(rec)
    recs.append(b"@r" + b"x" * pad + rec[2:])
    buf = b"".join(recs)
    assert len(buf) == 2 * chunk
    p = tmp_path / "exact.fq.gz"
    p.write_bytes(gzip.compress(buf))
    st = ctx.fastx_kmer_hash_file(p, 4, 31, 1, chunk_bytes=chunk)
    assert st.reads == len(recs) and st.kmers == 0 and st.batches == 2
    st = ctx.fastx_kmer_hash_file(p, 4, 30, 1, chunk_bytes=chunk)
    assert st.reads == len(recs) and st.kmers == len(recs)
    # an empty stream
    p = tmp_path / "empty.fq.gz"
    p.write_bytes(gzip.compress(b""))
    st = ctx.fastx_kmer_hash_file(p, 4, 31, 1)
    assert st.reads == 0 and st.batches == 0
    # cut short: inside the deflate data, and inside the trailer
    whole = gzip.compress(buf)
    for cut in (len(whole) // 2, len(whole) - 3):
        p = tmp_path / "short.fq.gz"
        p.write_bytes(whole[:cut])
        with pytest.raises(nthash_amd.NtHipError):
            ctx.fastx_kmer_hash_file(p, 4, 31, 1, chunk_bytes=chunk)
    # a flipped byte in the deflate data (crc or structure error)
    bad = bytearray(whole)
    bad[len(bad) // 2] ^= 0x5A
    p = tmp_path / "corrupt.fq.gz"
    p.write_bytes(bytes(bad))
    with pytest.raises(nthash_amd.NtHipError):
        ctx.fastx_kmer_hash_file(p, 4, 31, 1, chunk_bytes=chunk)
    p = tmp_path / "magic.fq.gz"
    p.write_bytes(b"\x1f\x8b")
    with pytest.raises(nthash_amd.NtHipError):
        ctx.fastx_kmer_hash_file(p, 4, 31, 1)
    # the context is usable afterwards
    p = tmp_path / "ok.fq"
    p.write_bytes(rec)
    assert ctx.fastx_kmer_hash_file(p, 4, 30, 1).kmers == 1


def _bgzf(buf, block=65280, eof=True, level=6):
    """buf as a BGZF file (bgzip / htslib): gzip members of `block` inflated bytes with the 'BC' extra field that holds the
    member's size, and the 28-byte end-of-file marker"""
    import struct, zlib
    out = []
    for a in range(0, len(buf), block):
        raw = buf[a:a + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        payload = co.compress(raw) + co.flush()
        bsize = 18 + len(payload) + 8
        assert bsize <= 65536
        out.append(struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, 66, 67, 2, bsize - 1) + payload
                   + struct.pack("<II", zlib.crc32(raw), len(raw)))
    if eof:
        out.append(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    return b"".join(out)


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,chunk,block,eof,final_newline", [
    (4, 1 << 16, 65280, True, True),      # bgzip's own block size: one block per chunk
    (4, 1 << 17, 1000, True, False),      # many small blocks per chunk, no newline at the end of the stream
    (2, 100_000, 4096, False, True),      # no end-of-file marker
    (4, 1 << 26, 30_000, True, True),     # the whole file in one chunk
])
def test_fastx_bgzf_file_stream_vs_oracle(ctx, oracle, tmp_path, fmt, chunk, block, eof, final_newline):
    """a BGZF file: its blocks inflated side by side by the reader threads (sizes from the block headers, crc32 checked) ==
    the oracle on the parsed reads == the same file through the one-thread gzread path (NTHIP_TUNE_NO_BGZF=1)"""
    import gzip, os
    rng = np.random.default_rng(chunk % 983 + fmt + block)
    buf, seqs = _make_fastx(rng, 6000, fmt, lo=20, hi=400)
    if not final_newline:
        buf = buf[:-1]
    bg = _bgzf(buf, block, eof)
    assert gzip.decompress(bg) == buf                      # (the writer above makes what gzip readers accept)
    path = tmp_path / "reads.fq.gz"
    path.write_bytes(bg)
    k, m = 31, 2
    data, offs = concat_reads(seqs)
    want = oracle.kmer_batch(data, offs, k, m, want_pos=False)

    def run():
        got_h, got_c, firsts = [], [], []

        def on_batch(b):
            h, cnt = np.zeros(b.n_kmers * m, np.uint64), np.zeros(b.n_reads, np.uint64)
            if h.size:
                ctx.d2h(h, b.hashes)
            ctx.d2h(cnt, b.counts)
            got_h.append(h); got_c.append(cnt); firsts.append(b.first_read)

        st = ctx.fastx_kmer_hash_file(path, fmt, k, m, chunk_bytes=chunk, on_batch=on_batch)
        assert firsts == list(np.cumsum([0] + [c.size for c in got_c[:-1]]))
        return st, np.concatenate(got_h), np.concatenate(got_c)

    st, h, cnt = run()
    assert st.reads == len(seqs) and st.kmers == want["total"] and st.file_bytes == len(bg)
    if chunk < len(buf) // 4:
        assert st.batches > 4
    assert (cnt == want["counts"]).all() and (h == want["hashes"].ravel()).all()
    os.environ["NTHIP_TUNE_NO_BGZF"] = "1"
    try:
        st1, h1, c1 = run()
    finally:
        del os.environ["NTHIP_TUNE_NO_BGZF"]
    assert st1.reads == st.reads and (h1 == h).all() and (c1 == cnt).all()


@pytest.mark.gpu
def test_fastx_bgzf_errors(ctx, tmp_path):
    """a block whose data, crc32 or size field was damaged, a file cut inside a block, an ordinary gzip member behind BGZF
    blocks: NTHIP_ERR_ARG, and the context works afterwards"""
    import gzip
    import nthash_amd
    rng = np.random.default_rng(5)
    buf, seqs = _make_fastx(rng, 3000, 4, lo=50, hi=200)
    good = _bgzf(buf, 20_000)
    first = 18 + int.from_bytes(good[16:18], "little") + 1 - 18      # size of the first block
    cases = {}
    b = bytearray(good); b[first // 2] ^= 0x10; cases["data"] = bytes(b)
    b = bytearray(good); b[first - 8] ^= 0x01; cases["crc"] = bytes(b)
    b = bytearray(good); b[first - 4] ^= 0x01; cases["isize"] = bytes(b)
    cases["cut"] = good[:len(good) - 28 - 100]
    cases["mixed"] = _bgzf(buf[:40_000], 20_000, eof=False) + gzip.compress(buf[40_000:])
    for name, blob in cases.items():
        p = tmp_path / (name + ".fq.gz")
        p.write_bytes(blob)
        with pytest.raises(nthash_amd.NtHipError):
            ctx.fastx_kmer_hash_file(p, 4, 31, 1, chunk_bytes=1 << 16)
    p = tmp_path / "good.fq.gz"
    p.write_bytes(good)
    assert ctx.fastx_kmer_hash_file(p, 4, 31, 1, chunk_bytes=1 << 16).reads == len(seqs)
    # only the end-of-file marker: an empty stream
    p.write_bytes(_bgzf(b""))
    st = ctx.fastx_kmer_hash_file(p, 4, 31, 1)
    assert st.reads == 0 and st.batches == 0


@pytest.mark.gpu
def test_fastx_gzip_multiline_fasta_and_multi_device(ctx, oracle, tmp_path):
    """NTHIP_FASTA_MULTILINE from a gzip file (inflated on the host, then the one-batch path), and the multi-device driver
    on a gzip file: one inflating thread, so the first device's pipeline takes the whole file -- same batches, same order"""
    import gzip
    import nthash_amd
    from nthash_amd.capi import NTHIP_FASTA_MULTILINE
    rng = np.random.default_rng(78)
    buf, seqs = _make_multiline_fasta(rng, 40, max_len=60_000)
    path = tmp_path / "genome.fa.gz"
    path.write_bytes(_gzip_members(buf[:-1], [len(buf) // 3]))
    k, m = 31, 2
    data, offs = concat_reads(seqs)
    want = oracle.kmer_batch(data, offs, k, m, want_pos=False)
    got = {}

    def on_genome(b):
        h, cnt = np.zeros(b.n_kmers * m, np.uint64), np.zeros(b.n_reads, np.uint64)
        ctx.d2h(h, b.hashes); ctx.d2h(cnt, b.counts)
        got["h"], got["c"] = h, cnt

    st = ctx.fastx_kmer_hash_file(path, NTHIP_FASTA_MULTILINE, k, m, on_batch=on_genome)
    assert st.reads == len(seqs) and st.kmers == want["total"] and st.batches == 1
    assert (got["c"] == want["counts"]).all() and (got["h"] == want["hashes"].ravel()).all()
    # the same genome bgzipped: its blocks inflated side by side before the one-batch path
    got.clear()
    path = tmp_path / "genome.bgzf.fa.gz"
    path.write_bytes(_bgzf(buf[:-1], 30_000))
    st = ctx.fastx_kmer_hash_file(path, NTHIP_FASTA_MULTILINE, k, m, on_batch=on_genome)
    assert st.reads == len(seqs) and st.kmers == want["total"] and st.batches == 1
    assert (got["c"] == want["counts"]).all() and (got["h"] == want["hashes"].ravel()).all()
    bad = bytearray(_bgzf(buf[:-1], 30_000)); bad[5000] ^= 0x40
    path.write_bytes(bytes(bad))
    with pytest.raises(nthash_amd.NtHipError):
        ctx.fastx_kmer_hash_file(path, NTHIP_FASTA_MULTILINE, k, m)
    # multi-device
    buf, seqs = _make_fastx(rng, 5000, 4, lo=20, hi=300)
    path = tmp_path / "reads.fq.gz"
    path.write_bytes(gzip.compress(buf))
    data, offs = concat_reads(seqs)
    want = oracle.kmer_batch(data, offs, k, m, want_pos=False)
    got_h, got_c, firsts = [], [], []

    def on_batch(b):
        h, cnt = np.zeros(b.n_kmers * m, np.uint64), np.zeros(b.n_reads, np.uint64)
        if h.size:
            ctx.d2h(h, b.hashes)
        ctx.d2h(cnt, b.counts)
        got_h.append(h); got_c.append(cnt); firsts.append(b.first_read)

    mg = nthash_amd.Multi([0, 0, 0])
    try:
        st = mg.fastx_kmer_hash_file(path, 4, k, m, chunk_bytes=1 << 17, on_batch=on_batch)
    finally:
        mg.close()
    assert st.reads == len(seqs) and st.kmers == want["total"] and st.batches > 4
    assert firsts == list(np.cumsum([0] + [c.size for c in got_c[:-1]]))
    assert (np.concatenate(got_c) == want["counts"]).all()
    assert (np.concatenate(got_h) == want["hashes"].ravel()).all()


def test_fastx_seed_stream_vs_oracle(ctx, oracle, tmp_path):
    """SeedNtHash over a streamed FASTQ (spans + the reference's position state machine on reads with N)"""
    import nthash_amd
    rng = np.random.default_rng(91)
    buf, seqs = _make_fastx(rng, 2500, 4, lo=10, hi=300, bad_frac=0.004)
    path = tmp_path / "reads.fq"
    path.write_bytes(buf)
    seeds = ["1010101010101010101010101010101", "1101101101101101011011011011011"]
    k, m2 = 31, 2
    data, offs = concat_reads(seqs)
    want = oracle.seed_batch(data, offs, seeds, k, m2, want_pos=False)
    sd = nthash_amd.Seeds(ctx, seeds, k)
    per = len(seeds) * m2
    got_h, got_c = [], []

    def on_batch(b):
        h, cnt = np.zeros(b.n_kmers * per, np.uint64), np.zeros(b.n_reads, np.uint64)
        if h.size:
            ctx.d2h(h, b.hashes)
        ctx.d2h(cnt, b.counts)
        got_h.append(h); got_c.append(cnt)

    st = ctx.fastx_kmer_hash_file(path, 4, k, m2, chunk_bytes=1 << 17, on_batch=on_batch, seeds=sd)
    assert st.reads == len(seqs) and st.kmers == want["total"] and st.batches > 3
    assert (np.concatenate(got_c) == want["counts"]).all()
    assert (np.concatenate(got_h) == want["hashes"].ravel()).all()


def test_fastx_single_line_fasta_long_sequences(ctx, oracle, tmp_path):
    """2-line FASTA whose sequence lines are almost the whole chunk (capacity: one k-mer per byte)"""
    rng = np.random.default_rng(5)
    buf, seqs = _make_fastx(rng, 40, 2, lo=20_000, hi=60_000)
    path = tmp_path / "contigs.fa"
    path.write_bytes(buf)
    data, offs = concat_reads(seqs)
    want = oracle.kmer_batch(data, offs, 31, 1, want_pos=False)
    got = []
    st = ctx.fastx_kmer_hash_file(path, 2, 31, 1, chunk_bytes=1 << 18,
                                  on_batch=lambda b: got.append(_d2h_u64(ctx, b.hashes, b.n_kmers)))
    assert st.kmers == want["total"] and (np.concatenate(got) == want["hashes"].ravel()).all()


def _d2h_u64(ctx, dptr, n):
    out = np.zeros(n, np.uint64)
    if n:
        ctx.d2h(out, dptr)
    return out


# ---------------------------------------------------------------------------
# any k, any m: k > 64 hashes a run's first window by Horner, m > 8 computes its multipliers
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("n,L,k,m", [
    (500, 300, 65, 1), (300, 250, 96, 2), (200, 400, 128, 3), (150, 401, 127, 1), (100, 1000, 255, 1),
    (40, 3000, 1000, 2), (6, 9000, 4099, 1), (800, 150, 31, 12), (300, 150, 31, 255), (64, 200, 100, 9),
    (700, 130, 66, 1), (33, 150, 150, 4),
])
def test_kmer_any_k_any_m_vs_oracle(ctx, oracle, n, L, k, m):
    rng = np.random.default_rng(n + k)
    clean = oracle.synth_reads(4, n, L, 31 * k + m)
    offs = np.arange(n + 1, dtype=np.uint64) * L
    want = oracle.kmer_batch(clean, offs, k, m, want_pos=False)
    ctx.set_profiling(True)
    got = ctx.kmer_hash(clean, k, m, fixed_len=L, n_reads=n)                     # dense pass
    name = ctx.last_kernel_ms()[1]
    ctx.set_profiling(False)
    assert name.startswith("kmer_runs_gen_kernel"), name
    assert got["total"] == want["total"] == n * (L - k + 1)
    assert (got["hashes"] == want["hashes"]).all()
    dirty = clean.copy()
    bad = rng.choice(n * L, max(4, n * L // 3000), replace=False)
    dirty[bad] = np.frombuffer(b"NnRY*", dtype=np.uint8)[rng.integers(0, 5, bad.size)]
    dirty[0] = ord("N"); dirty[-1] = ord("N")
    want = oracle.kmer_batch(dirty, offs, k, m)
    got = ctx.kmer_hash(dirty, k, m, fixed_len=L, n_reads=n, want_pos=True)      # N-aware pass
    assert got["total"] == want["total"]
    for key in ("counts", "pos", "hashes"):
        assert (got[key] == want[key]).all(), key
    # the same reads as a ragged batch (trimmed to random lengths): the run-split ragged kernel
    lens = rng.integers(0, L + 1, n)
    reads = [dirty[i * L:i * L + int(lens[i])].tobytes() for i in range(n)]
    d, o = concat_reads(reads)
    want = oracle.kmer_batch(d, o, k, m)
    got = ctx.kmer_hash(d, k, m, offsets=o, want_pos=True)
    assert got["total"] == want["total"]
    for key in ("counts", "pos", "hashes"):
        assert (got[key] == want[key]).all(), ("ragged", key)


@pytest.mark.parametrize("fw,table_k_max", [("1", None), ("2", None), ("1", "16"), ("2", "16")])
def test_kmer_first_window_forms_vs_oracle(oracle, fw, table_k_max):
    """first_window.hpp: the grouped (16 bases per step) and the prefix-scan (k-independent) first window of a run, each
    forced with NTHIP_TUNE_FW on every shape -- k across 31 / 33 / 64 / 65 / 1023 (the period of the split rotate) and
    beyond, short and long reads, read lengths that put a slab's end on and off a word boundary -- through the dense
    pass, the N-aware pass (positions, counts), MinHash, Bloom insert and the batched extension query; with
    NTHIP_TUNE_TABLE_K_MAX=16 the k <= 64 shapes take the k-independent forms too"""
    import os
    import nthash_amd
    os.environ["NTHIP_TUNE_FW"] = fw
    if table_k_max:
        os.environ["NTHIP_TUNE_TABLE_K_MAX"] = table_k_max
    try:
        c = nthash_amd.Context(0)
        shapes = [(400, 150, 65, 1), (300, 150, 100, 2), (200, 250, 200, 1), (60, 1000, 500, 1), (40, 1500, 1023, 1),
                  (20, 3000, 1024, 3), (12, 6000, 2047, 1), (3, 70000, 300, 1), (300, 160, 65, 1), (257, 177, 129, 1)]
        if table_k_max:
            shapes = [(600, 150, 31, 1), (500, 150, 33, 2), (500, 100, 64, 3), (400, 151, 17, 1), (300, 250, 48, 1)]
        for n, L, k, m in shapes:
            rng = np.random.default_rng(7 * n + k)
            clean = oracle.synth_reads(9, n, L, 5 * k + m)
            offs = np.arange(n + 1, dtype=np.uint64) * L
            want = oracle.kmer_batch(clean, offs, k, m, want_pos=False)
            c.set_profiling(True)
            got = c.kmer_hash(clean, k, m, fixed_len=L, n_reads=n)
            name = c.last_kernel_ms()[1]
            c.set_profiling(False)
            assert name.startswith("kmer_runs_gen_kernel"), (name, n, L, k, m)
            assert got["total"] == want["total"] == n * (L - k + 1)
            assert (got["hashes"] == want["hashes"]).all(), ("dense", n, L, k, m)
            dirty = clean.copy()
            bad = rng.choice(n * L, max(3, n * L // 4000), replace=False)
            dirty[bad] = np.frombuffer(b"NnRY*", dtype=np.uint8)[rng.integers(0, 5, bad.size)]
            want = oracle.kmer_batch(dirty, offs, k, m)
            got = c.kmer_hash(dirty, k, m, fixed_len=L, n_reads=n, want_pos=True)
            assert got["total"] == want["total"]
            for key in ("counts", "pos", "hashes"):
                assert (got[key] == want[key]).all(), (key, n, L, k, m)
            sig, _consumed = c.minhash(dirty, k, m, L, n)
            o = want["counts"].cumsum() - want["counts"]
            for r in (0, n // 2, n - 1):
                h = want["hashes"].reshape(-1, m)[int(o[r]):int(o[r] + want["counts"][r])]
                exp = h.min(axis=0) if len(h) else np.full(m, 2**64 - 1, np.uint64)
                assert (sig.reshape(n, m)[r] == exp).all(), ("minhash", n, L, k, m, r)
        # the batched extension query (BlindNtHash::peek / peek_back) hashes each k-mer with the grouped form
        for k, m in ((31, 1), (64, 2)) if table_k_max else ((65, 1), (129, 2), (1023, 1)):
            n = 300
            kmers = oracle.synth_reads(3, n, k, k)
            got = c.kmer_extend(kmers, k, m)
            offs = np.arange(n + 1, dtype=np.uint64) * k
            assert (got["self"].reshape(-1) == oracle.kmer_batch(kmers, offs, k, m, want_pos=False)["hashes"].reshape(-1)).all()
        c.close()
    finally:
        os.environ.pop("NTHIP_TUNE_FW", None)
        os.environ.pop("NTHIP_TUNE_TABLE_K_MAX", None)
        nthash_amd.Context(0).close()  # (the table limit is process-wide: a fresh context restores the default)


@pytest.mark.parametrize("n,L,stride", [(3000, 150, 0), (777, 151, 0), (500, 100, 0), (64, 31, 0), (300, 250, 220), (400, 150, 160),
                                        (40, 3000, 0), (5, 70000, 0)])
def test_kmer_packed_input_vs_oracle(ctx, oracle, n, L, stride):
    """nthip_pack_reads + NTHIP_PACKED_INPUT: the batch as a 2-bit code stream and a validity stream, made once, hashed at
    several k and m -- clean batches (NTHIP_PACKED_CLEAN: the headline instantiation for k = 31 / 150 bp, the general
    run-split kernel elsewhere, the k-independent first window beyond k = 64) and batches with non-bases (N-aware passes
    fed by the validity stream: counts, positions), overlapping runs and padded rows included -- against the oracle on
    the ASCII reads"""
    import nthash_amd
    rng = np.random.default_rng(n * 31 + L)
    st = stride or L
    total_bytes = (n - 1) * st + L
    clean = oracle.synth_reads(11, 1, total_bytes, n + L).copy()
    if st > L:  # padded rows: the padding is never hashed, whatever it holds
        for r in range(n):
            clean[r * st + L:(r + 1) * st] = ord("\n")
    offs_s = np.arange(n, dtype=np.uint64) * st
    reads = lambda buf: concat_reads([buf[int(o):int(o) + L].tobytes() for o in offs_s])
    d_pk, bad = ctx.pack_reads(clean, L, n, stride=stride)
    assert bad == (n - 1) * (st - L if st > L else 0)
    try:
        ks = [(31, 1), (21, 2), (64, 3), (17, 1)] + ([(100, 1), (129, 2)] if L >= 150 else []) + ([(1023, 1)] if L >= 3000 else [])
        for k, m in ks:
            if k > L:
                continue
            if st < L - k + 1:  # reads overlapping by more than k - 1 bases: not a packed shape
                with pytest.raises(nthash_amd.NtHipError) as ei:
                    ctx.kmer_hash_packed(d_pk, k, m, L, n, stride=stride, clean=True)
                assert ei.value.code == nthash_amd.capi.NTHIP_ERR_UNSUPPORTED
                continue
            d, o = reads(clean)
            want = oracle.kmer_batch(d, o, k, m)
            ctx.set_profiling(True)
            got = ctx.kmer_hash_packed(d_pk, k, m, L, n, stride=stride, clean=True, want_pos=True)
            name = ctx.last_kernel_ms()[1]
            ctx.set_profiling(False)
            if (L, k, m, stride) == (150, 31, 1, 0) and not os.environ.get("NTHASH_AMD_LIB"):  # (A/B builds may differ)
                assert name == "kmer_runs_kernel", name
            assert got["total"] == want["total"] == n * (L - k + 1)
            for key in ("hashes", "counts", "pos"):
                assert (got[key] == want[key]).all(), (key, k, m)
            # without the promise the same buffer goes through the N-aware passes: same answer
            got = ctx.kmer_hash_packed(d_pk, k, m, L, n, stride=stride, clean=False, want_pos=True)
            assert got["total"] == want["total"]
            for key in ("hashes", "counts", "pos"):
                assert (got[key] == want[key]).all(), ("unpromised", key, k, m)
    finally:
        ctx.free(d_pk)
    dirty = clean.copy()
    where = rng.choice(total_bytes, max(5, total_bytes // 2500), replace=False)
    dirty[where] = np.frombuffer(b"NnRY*\x00", dtype=np.uint8)[rng.integers(0, 6, where.size)]
    dirty[0] = ord("N"); dirty[-1] = ord("n")
    d_pk, bad = ctx.pack_reads(dirty, L, n, stride=stride)
    assert bad == int((~np.isin(dirty, np.frombuffer(b"ACGTUacgtu", dtype=np.uint8))).sum())
    try:
        for k, m in [(31, 1), (25, 3), (64, 1)] + ([(96, 2)] if L >= 150 else []):
            if k > L or st < L - k + 1:
                continue
            d, o = reads(dirty)
            want = oracle.kmer_batch(d, o, k, m)
            got = ctx.kmer_hash_packed(d_pk, k, m, L, n, stride=stride, clean=False, want_pos=True)
            assert got["total"] == want["total"]
            for key in ("hashes", "counts", "pos"):
                assert (got[key] == want[key]).all(), ("dirty", key, k, m)
            with pytest.raises(nthash_amd.NtHipError) as ei:  # too small an output: the need is reported
                ctx.kmer_hash_packed(d_pk, k, m, L, n, stride=stride, capacity=want["total"] - 1)
            assert ei.value.code == nthash_amd.capi.NTHIP_ERR_CAPACITY and ei.value.total == want["total"]
    finally:
        ctx.free(d_pk)


def test_kmer_packed_input_refusals(ctx, oracle):
    import nthash_amd
    data = oracle.synth_reads(0, 10, 100, 1)
    d_pk, _ = ctx.pack_reads(data, 100, 10)
    try:
        offs = np.arange(11, dtype=np.uint64) * 100
        d_offs = ctx.malloc(offs.nbytes); ctx.h2d(d_offs, offs)
        d_out = ctx.malloc(10 * 70 * 8)
        with pytest.raises(nthash_amd.NtHipError) as ei:   # variable-length reads are not a packed shape (yet)
            ctx.kmer_hash_ptr(d_pk, d_offs, 10, 0, 0, 31, 1, d_out, 700, flags=nthash_amd.capi.NTHIP_PACKED_INPUT)
        assert ei.value.code == nthash_amd.capi.NTHIP_ERR_UNSUPPORTED
        with pytest.raises(nthash_amd.NtHipError) as ei:   # a misaligned buffer is refused, not misread
            ctx.kmer_hash_ptr(d_pk + 4, 0, 10, 100, 0, 31, 1, d_out, 700, flags=nthash_amd.capi.NTHIP_PACKED_INPUT)
        assert ei.value.code == nthash_amd.capi.NTHIP_ERR_ARG
        # round 4: the validity stream's place follows from the number of bases that were packed; a call that describes
        # other reads (a subset of them) would take its validity bits from elsewhere -- refused unless it says CLEAN
        with pytest.raises(nthash_amd.NtHipError) as ei:
            ctx.kmer_hash_ptr(d_pk, 0, 9, 100, 0, 31, 1, d_out, 700, flags=nthash_amd.capi.NTHIP_PACKED_INPUT)
        assert ei.value.code == nthash_amd.capi.NTHIP_ERR_ARG and "not the batch" in str(ei.value)
        clean = nthash_amd.capi.NTHIP_PACKED_INPUT | nthash_amd.capi.NTHIP_PACKED_CLEAN
        assert ctx.kmer_hash_ptr(d_pk, 0, 9, 100, 0, 31, 1, d_out, 700, flags=clean) == 9 * 70
        assert ctx.kmer_hash_ptr(d_pk, 0, 10, 100, 0, 31, 1, d_out, 700, flags=nthash_amd.capi.NTHIP_PACKED_INPUT) == 700
        ctx.free(d_offs); ctx.free(d_out)
    finally:
        ctx.free(d_pk)


@pytest.mark.parametrize("knob", ["", "NTHIP_TUNE_NO_DIRTY_MEMORY", "NTHIP_TUNE_NO_NA_SPECIAL", "NTHIP_TUNE_NO_TILES_FLAG"])
def test_fixed_length_batches_with_non_bases_every_way(oracle, knob):
    """a fixed-length batch with a non-base: the dense pass gives up, the count pass finds every tile's place in the compact
    stream, the tiles that lost no window go through the specialised kernel at those places (kmer_runs_kernel's compact
    mode: tiles built where they start in a 128-byte line), the others through the N-aware kernel from a list; the next
    batch of the shape skips the dense pass (the context remembers), a batch that loses nothing takes the shape back.
    With either knob off the older paths run.  Shapes of the specialised kernel (150 / 31, run length 15; 151 / 31: 11)
    and one that is not; N's at a read's first and last base, in neighbouring reads, a read of N's only; batch sizes
    that end inside a tile and inside a piece of 8 tiles"""
    import os
    import nthash_amd
    if knob:
        os.environ[knob] = "1"
    try:
        c = nthash_amd.Context(0)
    finally:
        os.environ.pop(knob, None)
    rng = np.random.default_rng(41)
    for n, L, k in ((3000, 150, 31), (70001, 150, 31), (4099, 151, 31), (2000, 100, 25), (5, 150, 31)):
        clean = oracle.synth_reads(77, n, L, 9)
        dirty = clean.copy()
        at = [0, L - 1, 5 * L + 40, 6 * L, (n - 1) * L + L - 1] if n > 10 else [L + 3]
        at += [int(x) for x in rng.choice(n * L, max(1, n // 400), replace=False)]
        dirty[at] = ord("N")
        if n > 100:
            dirty[50 * L:51 * L] = ord("N")
        offs = np.arange(n + 1, dtype=np.uint64) * L
        want = {"clean": oracle.kmer_batch(clean, offs, k, 1, want_pos=False),
                "dirty": oracle.kmer_batch(dirty, offs, k, 1, want_pos=False)}
        for which in ("clean", "dirty", "dirty", "clean", "clean", "dirty"):
            got = c.kmer_hash(clean if which == "clean" else dirty, k, 1, fixed_len=L, n_reads=n)
            assert got["total"] == want[which]["total"], (which, n, L)
            assert (got["hashes"][:got["total"]] == want[which]["hashes"][:got["total"]]).all(), (which, n, L)
            assert (got["counts"] == want[which]["counts"]).all()
        if n == 70001:   # device-resident reads that start off a 16-byte boundary
            for shift in (5, 16):
                d_in, d_out = c.malloc(n * L + 64), c.malloc(n * (L - k + 1) * 8)
                c.h2d(d_in + shift, dirty)
                tot = c.kmer_hash_ptr(d_in + shift, 0, n, L, 0, k, 1, d_out, n * (L - k + 1))
                assert tot == want["dirty"]["total"]
                got_h = np.zeros(tot, np.uint64)
                c.d2h(got_h, d_out)
                assert (got_h == want["dirty"]["hashes"].ravel()[:tot]).all()
                c.free(d_in); c.free(d_out)


def test_seed_rolled_run_by_run_vs_oracle(oracle):
    """seed_roll_kernel: seeds are rolled the way the reference rolls them (NTMSM64, src/seed.cpp:177-207) -- one base in
    and one out per care run and window, a lane per segment of 16 / 8 / 4 windows whose first one is hashed directly --
    instead of ceil(k / 4) lookups per window: long seeds of a few solid
    blocks, k beyond 128, two seeds, several hashes per seed, monomers, a seed that starts and ends with don't-cares, reads
    that are not a multiple of 16 long, fewer segments than a tile, reads longer than a tile, six seeds, strided reads
    (not this kernel's).  Forced with NTHIP_TUNE_SEED_ROLL=1 so that the cost model cannot hide a shape; a batch
    with an N falls back to the kernels that know the position state machine"""
    import os
    import nthash_amd
    rng = np.random.default_rng(23)
    os.environ["NTHIP_TUNE_SEED_ROLL"] = "1"
    try:
        c = nthash_amd.Context(0)
    finally:
        os.environ.pop("NTHIP_TUNE_SEED_ROLL", None)

    def blocky(k, gaps):  # care everywhere but in the gaps [(start, length)]
        s = np.ones(k, dtype=bool)
        for a, n in gaps:
            s[a:a + n] = False
        return "".join("1" if b else "0" for b in s)

    cases = [  # (seeds, m2, L, n_reads)
        ([blocky(128, [(40, 48)])], 1, 250, 700),
        ([blocky(160, [(30, 20), (110, 20)])], 1, 300, 600),
        ([blocky(128, [(20, 5), (60, 8), (100, 9)]), blocky(128, [(64, 1)])], 2, 251, 515),
        ([blocky(64, [(10, 44)])], 3, 150, 1000),
        ([blocky(31, [(0, 3), (15, 1), (28, 3)])], 1, 100, 300),   # don't-cares at both ends, a gap of one
        ([blocky(48, [(5, 1), (7, 1), (9, 30), (41, 1), (43, 1)])], 2, 97, 129),  # monomers at 6, 8, 40, 42
        ([blocky(200, [(50, 100)]), blocky(200, [(10, 180)])], 1, 1000, 70),
        (["1" * 40], 1, 77, 2000),                                # no gap at all: a k-mer
        ([blocky(24, [(8, 8)])], 8, 24, 400),                      # one window per read
        ([blocky(40, [(10, 5)])], 1, 3000, 9),                     # a read is several tiles of segments
        ([blocky(31, [(3 + i, 2), (20, 4)]) for i in range(6)], 1, 150, 300),  # six seeds: segments of 4 windows
        ([blocky(64, [(20, 10)]), blocky(64, [(5, 5), (50, 3)]), blocky(64, [(31, 2)])], 1, 200, 257),  # three: 8 windows
        (["".join("10"[i & 1] for i in range(63)) + "1"], 1, 150, 200),  # 32 monomers
    ]
    for seeds, m2, L, n in cases:
        k = len(seeds[0])
        data = oracle.synth_reads(29, n, L, k + len(seeds))
        offs = np.arange(n + 1, dtype=np.uint64) * L
        want = oracle.seed_batch(data, offs, seeds, k, m2, want_pos=False)
        c.set_profiling(True)
        got = c.seed_hash(data, seeds, k, m2, fixed_len=L, n_reads=n)
        name = c.last_kernel_ms()[1]
        c.set_profiling(False)
        assert name == "seed_roll_kernel", (name, k, seeds)
        assert got["total"] == want["total"] == n * (L - k + 1)
        assert (got["hashes"] == want["hashes"]).all(), (k, seeds, m2)
        dirty = data.copy()
        dirty[rng.choice(n * L, 3, replace=False)] = ord("N")
        want = oracle.seed_batch(dirty, offs, seeds, k, m2)
        got = c.seed_hash(dirty, seeds, k, m2, fixed_len=L, n_reads=n, want_pos=True)
        assert got["total"] == want["total"]
        for key in ("counts", "pos", "hashes"):
            assert (got[key] == want[key]).all(), (key, k, seeds)
    # strided reads (overlapping windows of one long sequence)
    seeds = [blocky(96, [(32, 32)])]
    L, stride, n = 200, 50, 333
    data = oracle.synth_reads(31, 1, stride * (n - 1) + L, 5)
    rows = np.concatenate([data[i * stride:i * stride + L] for i in range(n)])
    want = oracle.seed_batch(rows, np.arange(n + 1, dtype=np.uint64) * L, seeds, 96, 2, want_pos=False)
    got = c.seed_hash(data, seeds, 96, 2, fixed_len=L, n_reads=n, stride=stride)
    assert (got["hashes"] == want["hashes"]).all()
    # a seed set outside the kernel (more runs than it takes) is not sent there even when forced
    many = ["".join("10"[i & 1] for i in range(63)) + "1"] * 3
    data = oracle.synth_reads(33, 200, 150, 3)
    want = oracle.seed_batch(data, np.arange(201, dtype=np.uint64) * 150, many, 64, 1, want_pos=False)
    c.set_profiling(True)
    got = c.seed_hash(data, many, 64, 1, fixed_len=150, n_reads=200)
    assert c.last_kernel_ms()[1] != "seed_roll_kernel"
    c.set_profiling(False)
    assert (got["hashes"] == want["hashes"]).all()


@pytest.mark.parametrize("forced", [False, True])
def test_seed_any_form_vs_oracle(oracle, forced):
    """seed_wtile_kernel<0>: any seed set of any k in ONE pass from the k-independent tables (care positions as a mask per
    16-base group, the masked-out positions' code-0 contribution XOR-ed off) -- the dense path of seeds beyond 128 bases
    (one wave per read until round 3), and, forced with NTHIP_TUNE_SEED_ANY=1, every dense batch: six seeds of k = 31,
    three of k = 64, asymmetric seeds, k not a multiple of 16, several hashes per seed; a batch with an N falls back"""
    import os
    import nthash_amd
    rng = np.random.default_rng(17 + forced)
    if forced:
        os.environ["NTHIP_TUNE_SEED_ANY"] = "1"
    os.environ["NTHIP_TUNE_SEED_ROLL"] = "2"  # (seeds beyond 128 bases of up to 64 runs may be rolled: not this test's form)
    try:
        c = nthash_amd.Context(0)
    finally:
        os.environ.pop("NTHIP_TUNE_SEED_ANY", None)
        os.environ.pop("NTHIP_TUNE_SEED_ROLL", None)

    def rand_seed(k, sym=True):
        half = rng.random((k + 1) // 2) < 0.7
        s = np.concatenate([half, half[: k // 2][::-1]]) if sym else rng.random(k) < 0.6
        s[0] = s[-1] = True
        return "".join("1" if b else "0" for b in s)

    cases = [(160, 1, 1, 300, 400), (200, 2, 2, 260, 300), (129, 3, 1, 150, 500), (333, 1, 3, 1000, 60)]
    if forced:
        cases = [(31, 6, 1, 250, 900), (64, 3, 1, 100, 1200), (31, 2, 3, 250, 700), (17, 5, 2, 60, 800), (100, 2, 2, 150, 500),
                 (48, 4, 1, 151, 640)]
    for k, ns, m2, L, n in cases:
        seeds = [rand_seed(k, sym=(i % 2 == 0)) for i in range(ns)]
        data = oracle.synth_reads(21, n, L, k + ns)
        offs = np.arange(n + 1, dtype=np.uint64) * L
        want = oracle.seed_batch(data, offs, seeds, k, m2, want_pos=False)
        c.set_profiling(True)
        got = c.seed_hash(data, seeds, k, m2, fixed_len=L, n_reads=n)
        name = c.last_kernel_ms()[1]
        c.set_profiling(False)
        assert name == "seed_wtile_kernel(any seed set)", (name, k, ns)
        assert got["total"] == want["total"] == n * (L - k + 1)
        assert (got["hashes"] == want["hashes"]).all(), (k, ns, m2)
        # a batch with non-bases: the clean reads on tiles of whole reads (seed_rtile_kernel<0>, the same form), the
        # others one by one -- not one wave per read for the whole batch (round 2: seeds beyond 64 bases)
        dirty = data.copy()
        dirty[rng.choice(n * L, 3, replace=False)] = ord("N")
        want = oracle.seed_batch(dirty, offs, seeds, k, m2)
        c.set_profiling(True)
        got = c.seed_hash(dirty, seeds, k, m2, fixed_len=L, n_reads=n, want_pos=True)
        name = c.last_kernel_ms()[1]
        c.set_profiling(False)
        if k > 64:
            assert name == "seed_rtile_kernel(any seed set)", (name, k, ns)
        assert got["total"] == want["total"]
        for key in ("counts", "pos", "hashes"):
            assert (got[key] == want[key]).all(), (key, k, ns)
        # variable-length reads (some shorter than k, one with an N)
        lens = rng.integers(max(1, k - 5), L + 1, 300)
        reads = [np.frombuffer(b"ACGTacgt", dtype=np.uint8)[rng.integers(0, 8, int(x))].tobytes() for x in lens]
        reads[150] = reads[150][: len(reads[150]) // 2] + b"N" + reads[150][len(reads[150]) // 2 + 1:]
        d, roffs = concat_reads(reads)
        want = oracle.seed_batch(d, roffs, seeds, k, m2)
        c.set_profiling(True)
        got = c.seed_hash(d, seeds, k, m2, offsets=roffs, want_pos=True)
        name = c.last_kernel_ms()[1]
        c.set_profiling(False)
        if k > 64 or forced:
            assert name == "seed_rtile_kernel(any seed set)", (name, k, ns)
        assert got["total"] == want["total"]
        for key in ("counts", "pos", "hashes"):
            assert (got[key] == want[key]).all(), (key, k, ns, "variable-length")
    c.close()


def test_bloom_long_k_many_hashes(ctx, oracle):
    n, L, k, m, n_bits = 300, 400, 101, 11, 3_000_017
    data = oracle.synth_reads(1, n, L, 8).copy()
    data[5::997] = ord("N")
    offs = np.arange(n + 1, dtype=np.uint64) * L
    want = oracle.kmer_batch(data, offs, k, m, want_pos=False)
    d_f, nbytes = ctx.bloom_new(n_bits)
    assert ctx.bloom_insert(data, k, m, L, n, d_f, n_bits) == want["total"]
    got = np.zeros(nbytes, np.uint8)
    ctx.d2h(got, d_f)
    assert (got == _bloom_expected(want["hashes"], n_bits)).all()
    hits, total, found = ctx.bloom_query(data, k, m, L, n, d_f, n_bits)
    assert total == found == want["total"] and (hits == want["counts"]).all()
    ctx.free(d_f)


@pytest.mark.parametrize("n,L,seeds,m2,stride", [
    (3000, 250, [SEED_A, SEED_B], 3, 0),          # BASELINE config 4 shape with N's
    (2000, 150, [SEED_A], 1, 0),                   # one seed, one hash: odd record size
    (1500, 100, ["110011", "101101", "111111"], 1, 0), (800, 64, ["1111111111111110111111111111111"], 2, 0),
    (500, 250, [SEED_A, SEED_B], 2, 220),          # overlapping runs: a bad byte lies in two reads
    (70, 40, ["10101", "11011"], 4, 0),
])
def test_seed_dirty_fixed_length_split_path(ctx, oracle, n, L, seeds, m2, stride):
    """fixed-length batch with non-bases (N, IUPAC, NUL): the reads that have one go through the reference's
    position state machine, the others stay on the fast kernel, writing into one compact stream"""
    rng = np.random.default_rng(n + L)
    k = len(seeds[0])
    total_bytes = n * L if not stride else (n - 1) * stride + L
    data = oracle.synth_reads(0, 1, total_bytes, 5).copy()
    nbad = max(5, total_bytes // 4000)
    data[rng.choice(total_bytes, nbad, replace=False)] = np.frombuffer(b"NnRYKMSW-\x00", dtype=np.uint8)[rng.integers(0, 10, nbad)]
    data[0] = ord("N"); data[-1] = ord("n")
    st_ = stride or L
    reads = [data[i * st_: i * st_ + L].tobytes() for i in range(n)]
    d, offs = concat_reads(reads)
    want = oracle.seed_batch(d, offs, seeds, k, m2, want_pos=True)
    for want_pos in (False, True):   # positions: window indices for the clean reads, the state machine's for the others
        ctx.set_profiling(True)
        got = ctx.seed_hash(data, seeds, k, m2, fixed_len=L, stride=stride, n_reads=n, want_pos=want_pos)
        name = ctx.last_kernel_ms()[1]
        ctx.set_profiling(False)
        assert name == "seed_fixed_kernel", name          # not the all-reads general fallback
        assert got["total"] == want["total"]
        assert (got["counts"] == want["counts"]).all()
        assert (got["hashes"] == want["hashes"]).all()
        if want_pos:
            assert (got["pos"] == want["pos"]).all()
    # a clean batch with positions stays on the dense kernel too
    clean = oracle.synth_reads(3, n, L, 9)
    offs_c = np.arange(n + 1, dtype=np.uint64) * L
    want_c = oracle.seed_batch(clean, offs_c, seeds, k, m2, want_pos=True)
    ctx.set_profiling(True)
    got_c = ctx.seed_hash(clean, seeds, k, m2, fixed_len=L, n_reads=n, want_pos=True)
    name = ctx.last_kernel_ms()[1]
    ctx.set_profiling(False)
    assert name in ("seed_fixed_kernel", "seed_wtile_kernel"), name   # (clean: the wave-tile dense kernel)
    assert (got_c["hashes"] == want_c["hashes"]).all() and (got_c["pos"] == want_c["pos"]).all()


@pytest.mark.parametrize("L,k,m", [(151, 31, 1), (125, 31, 1), (96, 64, 1), (76, 31, 2)])
def test_autotuned_run_length_gives_the_same_stream(oracle, L, k, m, monkeypatch):
    """a batch of >= 2^30 k-mers of a shape the context has not seen: the general dense kernel times a few run
    lengths on a slice and keeps the fastest -- whatever it picks, the stream is the one the model's choice gives
    (checksums of the whole stream + the first reads against the oracle)"""
    import nthash_amd
    nwin = L - k + 1
    n = (1 << 30) // nwin + 1000
    tuned, plain = nthash_amd.Context(0), nthash_amd.Context(0)
    d_in = tuned.malloc(n * L)
    d_out = tuned.malloc(n * nwin * m * 8)
    tuned.synth_reads_ptr(d_in, 0, n, L, 11)
    monkeypatch.delenv("NTHIP_TUNE_NO_AUTOTUNE", raising=False)
    assert tuned.kmer_hash_ptr(d_in, 0, n, L, 0, k, m, d_out, n * nwin) == n * nwin
    sums_tuned = tuned.checksum_ptr(d_out, n * nwin * m)
    head = np.zeros(2000 * nwin * m, np.uint64)
    tuned.d2h(head, d_out)
    monkeypatch.setenv("NTHIP_TUNE_NO_AUTOTUNE", "1")
    assert plain.kmer_hash_ptr(d_in, 0, n, L, 0, k, m, d_out, n * nwin) == n * nwin
    assert plain.checksum_ptr(d_out, n * nwin * m) == sums_tuned
    reads = oracle.synth_reads(0, 2000, L, 11)
    want = oracle.kmer_batch(reads, np.arange(2001, dtype=np.uint64) * L, k, m, want_pos=False)
    assert (head == want["hashes"].ravel()).all()
    # the second call of the tuned context reuses its choice
    assert tuned.kmer_hash_ptr(d_in, 0, n, L, 0, k, m, d_out, n * nwin) == n * nwin
    assert tuned.checksum_ptr(d_out, n * nwin * m) == sums_tuned
    # the same with N's: the N-aware passes tune their own run length (count -> scan -> hash on a slice)
    for i in range(0, n * L, 150_001):
        tuned.h2d(d_in + i, np.frombuffer(b"N", np.uint8))
    d_cnt = tuned.malloc(n * 8)
    monkeypatch.delenv("NTHIP_TUNE_NO_AUTOTUNE", raising=False)
    tot_t = tuned.kmer_hash_ptr(d_in, 0, n, L, 0, k, m, d_out, n * nwin, counts=d_cnt)
    res_t = (tuned.checksum_ptr(d_out, tot_t * m), tuned.checksum_ptr(d_cnt, n))
    monkeypatch.setenv("NTHIP_TUNE_NO_AUTOTUNE", "1")
    tot_p = plain.kmer_hash_ptr(d_in, 0, n, L, 0, k, m, d_out, n * nwin, counts=d_cnt)
    assert tot_p == tot_t < n * nwin
    assert (plain.checksum_ptr(d_out, tot_p * m), plain.checksum_ptr(d_cnt, n)) == res_t
    raw = np.zeros(3000 * L, np.uint8)
    tuned.d2h(raw, d_in)
    want_d = oracle.kmer_batch(raw, np.arange(3001, dtype=np.uint64) * L, k, m, want_pos=False)
    head_d = np.zeros(want_d["total"] * m, np.uint64)
    monkeypatch.delenv("NTHIP_TUNE_NO_AUTOTUNE", raising=False)
    assert tuned.kmer_hash_ptr(d_in, 0, n, L, 0, k, m, d_out, n * nwin, counts=d_cnt) == tot_t
    tuned.d2h(head_d, d_out)
    assert (head_d == want_d["hashes"].ravel()).all()
    for d in (d_in, d_out, d_cnt):
        tuned.free(d)


def test_async_dense_batches(ctx, oracle):
    """NTHIP_ASYNC: many small device-resident batches back to back without a round trip per call;
    nthip_ctx_take_dirty tells afterwards whether every stream is valid"""
    import nthash_amd
    from nthash_amd.capi import NTHIP_ASYNC
    n, L, k, m, nb = 500, 150, 31, 2, 12
    nwin = L - k + 1
    d_in, d_out = ctx.malloc(nb * n * L), ctx.malloc(nb * n * nwin * m * 8)
    ctx.synth_reads_ptr(d_in, 0, nb * n, L, 77)
    for b in range(nb):
        tot = ctx.kmer_hash_ptr(d_in + b * n * L, 0, n, L, 0, k, m, d_out + b * n * nwin * m * 8, n * nwin, flags=NTHIP_ASYNC)
        assert tot == n * nwin
    assert ctx.take_dirty() is False
    got = np.zeros(nb * n * nwin * m, np.uint64)
    ctx.d2h(got, d_out)
    data = oracle.synth_reads(0, nb * n, L, 77)
    want = oracle.kmer_batch(data, np.arange(nb * n + 1, dtype=np.uint64) * L, k, m, want_pos=False)["hashes"]
    assert (got == want.ravel()).all()
    # a batch with a non-base: reported, and synchronous calls are refused until it has been taken
    ctx.h2d(d_in + 5 * n * L + 77, np.frombuffer(b"N", np.uint8))
    for b in range(nb):
        ctx.kmer_hash_ptr(d_in + b * n * L, 0, n, L, 0, k, m, d_out + b * n * nwin * m * 8, n * nwin, flags=NTHIP_ASYNC)
    with pytest.raises(nthash_amd.NtHipError):
        ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, m, d_out, n * nwin)
    assert ctx.take_dirty() is True
    assert ctx.take_dirty() is False
    # not a plain dense call
    d_offs = ctx.malloc(16)
    with pytest.raises(nthash_amd.NtHipError):
        ctx.kmer_hash_ptr(d_in, d_offs, 1, 0, 0, k, m, d_out, n * nwin, flags=NTHIP_ASYNC)
    # and the ordinary path still works
    tot = ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, m, d_out, n * nwin)
    assert tot == n * nwin
    for d in (d_in, d_out, d_offs):
        ctx.free(d)


def test_uniform_offsets_take_the_fixed_length_kernels(ctx, oracle):
    """offsets of equal-length, back-to-back reads (what nthash::BatchNtHash sends for Illumina reads) are recognised
    and hashed by the fixed-stride kernels; one odd read anywhere keeps the variable-length path; same stream either way"""
    rng = np.random.default_rng(5)
    n, L, k, m = 5000, 150, 31, 2
    data = oracle.synth_reads(17, n, L, 3)
    offs = np.arange(n + 1, dtype=np.uint64) * L
    want = oracle.kmer_batch(data, offs, k, m, want_pos=True)
    for host in (True, False):
        ctx.set_profiling(True)
        if host:
            got = ctx.kmer_hash(data, k, m, offsets=offs, want_pos=True)
        else:
            d_in = ctx.malloc(data.size + 64); d_off = ctx.malloc(offs.nbytes); d_out = ctx.malloc(want["total"] * m * 8)
            ctx.h2d(d_in + 5, data); ctx.h2d(d_off, offs + np.uint64(5))   # offsets need not start at 0
            tot = ctx.kmer_hash_ptr(d_in, d_off, n, 0, 0, k, m, d_out, want["total"])
            h = np.zeros(want["hashes"].size, np.uint64); ctx.d2h(h, d_out)
            got = {"total": tot, "hashes": h.reshape(want["hashes"].shape)}
            ctx.free(d_in); ctx.free(d_off); ctx.free(d_out)
        name = ctx.last_kernel_ms()[1]
        ctx.set_profiling(False)
        assert name.startswith("kmer_runs"), name
        assert got["total"] == want["total"] and (got["hashes"] == want["hashes"]).all()
        if host:
            assert (got["pos"] == want["pos"]).all() and (got["counts"] == want["counts"]).all()
    # one read one base shorter: not uniform
    lens = np.full(n, L); lens[n // 2] = L - 1
    offs2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    d2 = data[: int(offs2[-1])]
    want2 = oracle.kmer_batch(d2, offs2, k, m, want_pos=False)
    ctx.set_profiling(True)
    got2 = ctx.kmer_hash(d2, k, m, offsets=offs2)
    name = ctx.last_kernel_ms()[1]
    ctx.set_profiling(False)
    assert name in ("kmer_reads_kernel", "kmer_ragged_kernel"), name
    assert got2["total"] == want2["total"] and (got2["hashes"] == want2["hashes"]).all()


def test_bad_offsets_are_refused(ctx):
    """a decreasing pair of offsets must come back as NTHIP_ERR_ARG, not as a wild read (the last offset IS the
    buffer size in this layout, so it cannot be checked against anything; spans are checked against buf_bytes)"""
    import nthash_amd
    n, L = 2000, 100
    data = np.frombuffer(b"ACGT" * (n * L // 4), dtype=np.uint8)
    for bad_at, bad_val in ((700, 10), (1500, 0)):
        offs = np.arange(n + 1, dtype=np.uint64) * L
        offs[1] = L - 1           # (not uniform: the offsets are really used)
        offs[bad_at] = bad_val
        d_in = ctx.malloc(data.size); d_off = ctx.malloc(offs.nbytes); d_out = ctx.malloc(n * L * 8)
        try:
            ctx.h2d(d_in, data); ctx.h2d(d_off, offs)
            with pytest.raises(nthash_amd.NtHipError) as e:
                ctx.kmer_hash_ptr(d_in, d_off, n, 0, 0, 31, 1, d_out, n * L)
            assert e.value.code == nthash_amd.capi.NTHIP_ERR_ARG
        finally:
            ctx.free(d_in); ctx.free(d_off); ctx.free(d_out)


def test_windowed_build_of_the_headline_kernel_is_bit_exact():
    """-DKR_CHUNKED=1 (kmer_runs_kernel.hpp: chip-wide read windows paced by the 100 MHz clock; off by default because
    it is not faster) must stay bit-exact: this file's k-mer tests again, through that build of the library"""
    import os
    import subprocess
    import sys

    from conftest import ROOT
    lib = os.path.join(ROOT, "nthash_amd", "lib", "ab", "libnthash_hip_win.so")
    if not os.path.exists(lib):  # (__graft_entry__.build() makes it; a tree that was not built that way compiles it here)
        subprocess.check_call([sys.executable, "-m", "nthash_amd.build", "--tag", "win", "--flags", "-DKR_CHUNKED=1", "--units",
                               "capi_kmer_runs"], cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    env = dict(os.environ, NTHASH_AMD_LIB=lib)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q", "-x",
                        "-k", "kmer and not windowed and not native_library and not read_slots_fixed"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("devices,allgather", [([0, 0], False), ([0, 0], True), ([0, 0, 0], True), ([0, 0, 0, 0], False),
                                               ([0, 0, 0, 0, 0], True)])
def test_multi_device_consumers_merge_over_peer_copies(ctx, oracle, devices, allgather):
    """nthip_multi_kmer_bloom_insert / _count_insert / _minhash_set (round 4): every device of the set consumes its
    device-resident shard into its own table, then the tables are merged by the ring reduce-scatter (+ all-gather) over
    peer copies -- OR, saturating add of one-byte counters, minimum of 64-bit entries.  The box has one GPU, so it is listed
    2-5 times (peer-to-self copies; separate contexts, threads, tables and staging buffers: the code an 8-GPU node runs).
    Against ONE device consuming all the reads: bit for bit, what the tables held before included; uneven shards, an empty
    shard, reads with non-bases; the merge alone on random tables; the device-resident nthip_multi_kmer_hash_shards."""
    import nthash_amd
    from nthash_amd import capi
    G = len(devices)
    n, L, k, m = 9000, 150, 31, 2
    nwin = L - k + 1
    data = oracle.synth_reads(77, n, L, 3).copy()
    data[5 * L + 40] = ord("N")
    data[(n - 1) * L + 3] = ord("n")
    offs = np.arange(n + 1, dtype=np.uint64) * L
    want = oracle.kmer_batch(data, offs, k, m, want_pos=False)
    n_bits, n_cnt = 1 << 21, 9008          # (2.1 M values on 9008 counters: some saturate, most do not)
    rng = np.random.default_rng(4)
    prior_f = (rng.integers(0, 256, n_bits // 8, dtype=np.uint8) & rng.integers(0, 256, n_bits // 8, dtype=np.uint8)
               & rng.integers(0, 256, n_bits // 8, dtype=np.uint8))
    prior_c = rng.integers(0, 3, n_cnt, dtype=np.uint8)
    # one device, all the reads
    d_in, d_f, d_c = ctx.malloc(n * L), ctx.malloc(n_bits // 8), ctx.malloc(n_cnt)
    ctx.h2d(d_in, data)
    ctx.h2d(d_f, prior_f)
    ctx.h2d(d_c, prior_c)
    tot1 = ctx.bloom_insert_ptr(d_in, n, L, 0, k, m, d_f, n_bits)
    ctx.count_insert_ptr(d_in, n, L, 0, k, m, d_c, n_cnt)
    one_f, one_c = np.zeros(n_bits // 8, np.uint8), np.zeros(n_cnt, np.uint8)
    ctx.d2h(one_f, d_f)
    ctx.d2h(one_c, d_c)
    assert one_c.max() == 255          # (hot counters saturate: the merge must saturate the same way)
    for p in (d_in, d_f, d_c):
        ctx.free(p)
    one_sig = want["hashes"].reshape(-1, m).min(axis=0)
    # the set: uneven shards, the second one empty when there are more than two devices
    cuts = [0] + sorted(int(x) for x in rng.integers(1, n, G - 1)) + [n]
    if G > 2:
        cuts[2] = cuts[1]
    mm = nthash_amd.Multi(devices)
    assert mm.device_count() == G
    cs = [mm.ctx(g) for g in range(G)]
    shards, owned = [], []
    flt, cnt, sig, outs = [], [], [], []
    pad_m = (m + 1) & ~1
    for g in range(G):
        r0, r1 = cuts[g], cuts[g + 1]
        d = cs[g].malloc(max((r1 - r0) * L, 16))
        if r1 > r0:
            cs[g].h2d(d, data[r0 * L: r1 * L])
        shards.append((d, 0, r1 - r0, L, 0))
        f, c_, s_ = cs[g].malloc(n_bits // 8), cs[g].malloc(n_cnt), cs[g].malloc(pad_m * 8)
        o = cs[g].malloc(max((r1 - r0) * nwin * m * 8, 16))
        if g == 0:
            cs[g].h2d(f, prior_f)
            cs[g].h2d(c_, prior_c)
        else:
            cs[g].memset(f, 0, n_bits // 8)
            cs[g].memset(c_, 0, n_cnt)
        flt.append(f); cnt.append(c_); sig.append(s_); outs.append((o, (r1 - r0) * nwin))
        owned += [(g, d), (g, f), (g, c_), (g, s_), (g, o)]
    flags = capi.NTHIP_MULTI_ALLGATHER if allgather else 0
    assert mm.bloom_insert(shards, k, m, flt, n_bits, flags) == tot1 == want["total"]
    assert mm.count_insert(shards, k, m, cnt, n_cnt, flags) == tot1
    assert mm.minhash_set(shards, k, m, sig, flags) == tot1
    for g in (range(G) if allgather else [0]):
        gf, gc, gs = np.zeros(n_bits // 8, np.uint8), np.zeros(n_cnt, np.uint8), np.zeros(pad_m, np.uint64)
        cs[g].d2h(gf, flt[g]); cs[g].d2h(gc, cnt[g]); cs[g].d2h(gs, sig[g])
        assert (gf == one_f).all(), g
        assert (gc == one_c).all(), g
        assert (gs[:m] == one_sig).all(), g
    # the device-resident hash call: every device's stream is its shard of the single call's stream
    tots = mm.kmer_hash_shards(shards, k, m, outs)
    at = 0
    for g in range(G):
        got = np.zeros(tots[g] * m, np.uint64)
        if tots[g]:
            cs[g].d2h(got, outs[g][0])
        assert (got.reshape(-1, m) == want["hashes"].reshape(-1, m)[at: at + tots[g]]).all()
        at += tots[g]
    assert at == want["total"]
    # the merge alone, the three operators on random tables of an awkward size
    nb = 16 * 1031
    for op, name in ((capi.NTHIP_MERGE_OR, "or"), (capi.NTHIP_MERGE_ADD_SAT_U8, "add"), (capi.NTHIP_MERGE_MIN_U64, "min")):
        tabs = [rng.integers(0, 120 if name == "add" else 256, nb, dtype=np.uint8) for _ in range(G)]
        ptrs = []
        for g in range(G):
            p_ = cs[g].malloc(nb)
            cs[g].h2d(p_, tabs[g])
            ptrs.append(p_)
            owned.append((g, p_))
        mm.merge(ptrs, nb, op, flags)
        if name == "or":
            exp = np.bitwise_or.reduce(np.stack(tabs), axis=0)
        elif name == "add":
            exp = np.minimum(np.stack(tabs).astype(np.int64).sum(axis=0), 255).astype(np.uint8)
        else:
            exp = np.stack([t.view(np.uint64) for t in tabs]).min(axis=0).view(np.uint8)
        for g in (range(G) if allgather else [0]):
            got = np.zeros(nb, np.uint8)
            cs[g].d2h(got, ptrs[g])
            assert (got == exp).all(), (name, g)
    for g, p_ in owned:
        cs[g].free(p_)
    mm.close()


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0, 0, 0]])
def test_multi_device_shards_give_the_single_call_stream(oracle, devices):
    """nthip_multi_*: a host batch cut into one shard of reads per device (here the box's GPU listed several times: one
    context and one host thread each), hashed concurrently, stitched back -- must be exactly the oracle's stream,
    N-skipping, counts and positions included, for fixed-length and variable-length reads and for spaced seeds"""
    import nthash_amd
    mu = nthash_amd.Multi(devices)
    assert mu.device_count() == len(devices)
    rng = np.random.default_rng(len(devices))
    n, L, k, m = 3001, 120, 25, 2
    data = oracle.synth_reads(5, n, L, 77).copy()
    data[rng.choice(n * L, 40, replace=False)] = ord("N")       # gaps in the dense layout: shards must move down
    offs = np.arange(n + 1, dtype=np.uint64) * L
    want = oracle.kmer_batch(data, offs, k, m, want_pos=True)
    got = mu.kmer_hash(data, k, m, fixed_len=L, n_reads=n, want_pos=True)
    assert got["total"] == want["total"] and (got["counts"] == want["counts"]).all()
    assert (got["hashes"] == want["hashes"]).all() and (got["pos"] == want["pos"]).all()
    # variable lengths (shards balanced by bytes), reads shorter than k, an empty read
    lens = rng.integers(0, 200, 2000)
    alph = np.frombuffer(b"ACGTACGTN", dtype=np.uint8)
    reads = [alph[rng.integers(0, len(alph), int(x))].tobytes() for x in lens]
    d, o = concat_reads(reads)
    want = oracle.kmer_batch(d, o, 21, 1, want_pos=True)
    got = mu.kmer_hash(d, 21, 1, offsets=o, want_pos=True)
    assert got["total"] == want["total"] and (got["counts"] == want["counts"]).all()
    assert (got["hashes"] == want["hashes"]).all() and (got["pos"] == want["pos"]).all()
    # spaced seeds
    wants = oracle.seed_batch(data, offs, [SEED_A[:25], SEED_B[3:28]], 25, 2, want_pos=True)
    gots = mu.seed_hash(data, [SEED_A[:25], SEED_B[3:28]], 25, 2, fixed_len=L, n_reads=n, want_pos=True)
    assert gots["total"] == wants["total"] and (gots["hashes"] == wants["hashes"]).all() and (gots["pos"] == wants["pos"]).all()
    mu.close()


def test_placement_aware_allocation(ctx):
    """nthip_malloc_probed: several candidate allocations measured with the write-only fill, the fastest returned --
    a usable buffer (hashing into it gives the usual stream), the rate reported, small buffers allocated without a probe"""
    p, gbps, tried = ctx.malloc_probed(512 << 20, 3)
    try:
        assert p and tried >= 1 and 500 < gbps < 8000, (gbps, tried)
        n, L, k = 2000, 150, 31
        d_in = ctx.malloc(n * L)
        ctx.synth_reads_ptr(d_in, 0, n, L, 42)
        tot = ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, 1, p, n * 120)
        s1 = ctx.checksum_ptr(p, tot)
        q = ctx.malloc(n * 120 * 8)
        ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, 1, q, n * 120)
        assert tot == n * 120 and ctx.checksum_ptr(q, tot) == s1
        ctx.free(q); ctx.free(d_in)
    finally:
        ctx.free(p)
    p, gbps, tried = ctx.malloc_probed(1 << 20, 3)
    assert p and tried == 1 and gbps == 0
    ctx.free(p)
    # candidates after the first are mapped from small physical pieces (virtual-memory API): an odd size, copies both
    # ways over the whole range, many allocations and frees in a row, a context closed with one still alive
    import nthash_amd
    nb = (300 << 20) + 12345
    for it in range(6):
        p, gbps, tried = ctx.malloc_probed(nb, 4)
        assert p and tried >= 2
        src = np.random.default_rng(it).integers(0, 256, nb, dtype=np.uint8)
        ctx.h2d(p, src)
        back = np.empty(nb, np.uint8)
        ctx.d2h(back, p)
        assert (back == src).all()
        ctx.free(p)
    other = nthash_amd.Context(0)
    p, _, _ = other.malloc_probed(256 << 20, 3)
    assert p
    other.close()


def test_seed_offsets_are_surveyed(ctx, oracle):
    """nthip_seed_hash with offsets: reads that all have one length and lie back to back take the dense kernel (as in
    nthip_kmer_hash), offsets that decrease are refused instead of being read through"""
    import nthash_amd
    n, L, k, m2 = 3000, 150, 31, 3
    data = oracle.synth_reads(2, n, L, 7)
    offs = np.arange(n + 1, dtype=np.uint64) * L
    want = oracle.seed_batch(data, offs, [SEED_A, SEED_B], k, m2)
    ctx.set_profiling(True)
    got = ctx.seed_hash(data, [SEED_A, SEED_B], k, m2, offsets=offs, want_pos=True)
    name = ctx.last_kernel_ms()[1]
    ctx.set_profiling(False)
    assert name == "seed_wtile_kernel", name
    assert got["total"] == want["total"] and (got["hashes"] == want["hashes"]).all()
    assert (got["pos"] == want["pos"]).all() and (got["counts"] == want["counts"]).all()
    bad = offs.copy()
    bad[1500] = 10
    with pytest.raises(nthash_amd.NtHipError) as e:
        ctx.seed_hash(data, [SEED_A, SEED_B], k, m2, offsets=bad)
    assert e.value.code == nthash_amd.capi.NTHIP_ERR_ARG


def test_kmer_whole_read_tiles_many_hashes(ctx, oracle):
    """the whole-read-tile path with more hashes per k-mer than the multiplier table of the fixed kernels holds (m = 20,
    m = 255: the multipliers are computed), positions included"""
    rng = np.random.default_rng(9)
    alph = np.frombuffer(b"ACGTacgtN", dtype=np.uint8)
    for k, m in ((31, 20), (17, 255), (40, 9)):
        reads = [alph[rng.integers(0, len(alph) - (0 if rng.random() < 0.1 else 1), int(rng.integers(0, 200)))].tobytes()
                 for _ in range(300)]
        d, offs = concat_reads(reads)
        want = oracle.kmer_batch(d, offs, k, m)
        ctx.set_profiling(True)
        got = ctx.kmer_hash(d, k, m, offsets=offs, want_pos=True)
        name = ctx.last_kernel_ms()[1]
        ctx.set_profiling(False)
        assert name == "kmer_reads_kernel", name
        assert got["total"] == want["total"]
        for key in ("counts", "pos", "hashes"):
            assert (got[key] == want[key]).all(), (k, m, key)


def test_seed_passes_vs_oracle(oracle):
    """seed sets hashed a few seeds per pass of seed_wtile_kernel (every pass writes its part of each record): more than
    two seeds of k <= 32 on the rotated-slot layout, seed sets whose byte tables do not fit in LDS together (k = 64:
    three seeds), many seeds; the same batches with one seed per pass forced (NTHIP_TUNE_SEED_PASS=1) and with an 'N'
    in one read (the batch leaves the dense kernel)"""
    import os
    import nthash_amd
    rng = np.random.default_rng(4242)
    planned = nthash_amd.Context(0)
    os.environ["NTHIP_TUNE_SEED_PASS"] = "1"
    try:
        single = nthash_amd.Context(0)
    finally:
        os.environ.pop("NTHIP_TUNE_SEED_PASS", None)

    def mask(k, density):
        m = (rng.random(k) < density).astype(int)
        m[0] = m[-1] = 1
        return "".join(str(int(x)) for x in m)

    shapes = [(31, 3, 1), (31, 3, 3), (31, 4, 2), (31, 6, 1), (20, 5, 4), (32, 7, 2), (48, 3, 2), (48, 5, 1), (64, 2, 3),
              (64, 3, 1), (57, 4, 5), (31, 2, 3), (12, 9, 1),
              (65, 1, 1), (80, 2, 2), (96, 3, 1), (100, 1, 3), (113, 2, 1), (127, 1, 2), (128, 2, 2)]  # (long seeds: 8 lookups at a time)
    for (k, n_seeds, m2) in shapes:
        seeds = [mask(k, 0.65) for _ in range(n_seeds)]
        n, L = int(rng.integers(100, 900)), int(rng.integers(k, 260))
        data = np.frombuffer(b"ACGTacgt", dtype=np.uint8)[rng.integers(0, 8, n * L)].copy()
        offs = np.arange(n + 1, dtype=np.uint64) * L
        for dirty in (False, True):
            if dirty:
                data[(n // 2) * L + L // 2] = ord("N")
            want = oracle.seed_batch(data, offs, seeds, k, m2, want_pos=False)
            for c in (planned, single):
                c.set_profiling(True)
                got = c.seed_hash(data, seeds, k, m2, fixed_len=L, n_reads=n)
                name = c.last_kernel_ms()[1]
                c.set_profiling(False)
                assert got["total"] == want["total"], (k, n_seeds, m2, dirty)
                assert (got["hashes"] == want["hashes"]).all(), (k, n_seeds, m2, dirty, c is planned)
                # (an 'N' in the batch: the split pass of the block-tile kernel when its LDS plan has room, otherwise tiles of
                #  whole reads over spans made on the device -- never the lane-per-read kernel for the whole batch)
                # (the planned context sends seed sets of several passes to the any-seed form where that is ahead)
                assert dirty or name == "seed_wtile_kernel" or (c is planned and name == "seed_wtile_kernel(any seed set)"), \
                    (name, k, n_seeds, m2)
                assert name != "seed_general_kernel", (name, k, n_seeds, m2, dirty)
        # reads of several lengths (tiles of whole reads, seed_rtile_kernel), one of them with an 'N'
        alph = np.frombuffer(b"ACGTacgt", dtype=np.uint8)
        reads = [alph[rng.integers(0, 8, int(rng.integers(max(1, k - 3), max(200, 2 * k))))].tobytes() for _ in range(700)]
        reads[350] = reads[350][: len(reads[350]) // 2] + b"N" + reads[350][len(reads[350]) // 2 + 1:]
        d, roffs = concat_reads(reads)
        want = oracle.seed_batch(d, roffs, seeds, k, m2)
        for c in (planned, single):
            c.set_profiling(True)
            got = c.seed_hash(d, seeds, k, m2, offsets=roffs, want_pos=True)
            name = c.last_kernel_ms()[1]
            c.set_profiling(False)
            # (seeds beyond 64 bases, and -- on the planned context -- seed sets of several passes: the any-seed form)
            assert name in (("seed_rtile_kernel", "seed_rtile_kernel(any seed set)") if k <= 64 and c is planned else
                            ("seed_rtile_kernel",) if k <= 64 else ("seed_rtile_kernel(any seed set)",)), (name, k, n_seeds, m2)
            assert got["total"] == want["total"]
            for key in ("counts", "pos", "hashes"):
                assert (got[key] == want[key]).all(), (k, n_seeds, m2, key, c is planned)
    planned.close()
    single.close()


def test_seed_long_reads_cut_into_pieces(ctx, oracle):
    """SeedNtHash on long reads (nthip_seed_hash, offsets and fixed length): the reads are cut into independent pieces at
    positions with 2k bases around them (seed_long_kernels.hpp) -- counts, positions and hashes against the oracle's
    sequential walk; non-bases single, in runs of k - 1 / k / k + 1 / thousands, right at the nominal cuts, NUL bytes,
    a clean read, a read of non-bases only, k up to 100"""
    import os
    import nthash_amd
    rng = np.random.default_rng(20260929)
    alph = np.frombuffer(b"ACGTacgt", dtype=np.uint8)

    def mask(k, density):
        m = (rng.random(k) < density).astype(int)
        m[0] = m[-1] = 1
        return "".join(str(int(x)) for x in m)

    def long_read(n, k, kind):
        d = alph[rng.integers(0, 8, n)].copy()
        if kind == "clean":
            return d
        if kind == "all_n":
            d[:] = ord("N")
            return d
        S = max(1280, 4 * k)
        for _ in range(int(rng.integers(3, 40))):  # runs of non-bases of telling lengths
            ln = int(rng.choice([1, 1, 2, k - 1, k, k + 1, 2 * k, 3 * k + 5, 700, 5000]))
            at = int(rng.integers(0, max(1, n - ln)))
            d[at:at + ln] = ord("N")
        for j in range(1, n // S):  # and right around the nominal cuts
            if rng.random() < 0.3:
                at = j * S + int(rng.integers(-2 * k - 2, S // 2 + 2 * k))
                if 0 <= at < n:
                    d[at] = rng.choice(np.frombuffer(b"NnRY-\x00", dtype=np.uint8))
        return d

    cases = [(31, 2, 2, [(60_000, "dirty"), (200_000, "dirty"), (300, "dirty"), (20_000, "clean"), (17_000, "all_n"), (0, "clean")]),
             (64, 1, 3, [(150_000, "dirty"), (16_384, "dirty")]),
             (100, 2, 1, [(90_000, "dirty")]),
             (17, 3, 1, [(120_000, "dirty"), (40_000, "dirty"), (90, "clean")])]
    for (k, n_seeds, m2, reads) in cases:
        seeds = [mask(k, 0.7) for _ in range(n_seeds)]
        parts = [long_read(n, k, kind) for (n, kind) in reads]
        d = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
        offs = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.uint64)
        want = oracle.seed_batch(d, offs, seeds, k, m2)
        ctx.set_profiling(True)
        got = ctx.seed_hash(d, seeds, k, m2, offsets=offs, want_pos=True)
        name = ctx.last_kernel_ms()[1]
        ctx.set_profiling(False)
        # (the pieces are spans in order that overlap by k - 1: tiles of whole pieces unless one of them -- a long run of
        #  non-bases has no cut -- is longer than a tile's slab, then a wave per piece)
        assert name in ("seed_wave_kernel", "seed_rtile_kernel"), name
        assert got["total"] == want["total"], (k, n_seeds, m2)
        for key in ("counts", "pos", "hashes"):
            assert (got[key] == want[key]).all(), (k, n_seeds, m2, key)
    # long runs of the letter N (cuts at the positions where the walk restarts): beginning at the read's start, before
    # position k, behind bases, behind another non-base / a NUL (no cut there), upper and lower case, another character
    # inside the run, two runs k - 1 bases apart, a run that reaches the read's end
    k, m2, seeds = 31, 2, [SEED_A, SEED_B]

    def with_runs(n, runs):
        d = alph[rng.integers(0, 8, n)].copy()
        for (at, ln, ch, before) in runs:
            d[at:at + ln] = np.frombuffer(ch, np.uint8)[np.arange(ln) % len(ch)]
            if before is not None and at > 0:
                d[at - 1] = before
        return d

    crafted = [with_runs(40_000, [(0, 30_000, b"N", None)]),
               with_runs(40_000, [(7, 30_000, b"N", None)]),
               with_runs(40_000, [(30, 30_000, b"n", None)]),
               with_runs(60_000, [(5_000, 40_000, b"Nn", None)]),
               with_runs(60_000, [(5_000, 40_000, b"N", ord("R"))]),
               with_runs(60_000, [(5_000, 40_000, b"N", 0)]),
               with_runs(60_000, [(5_000, 20_000, b"N", None), (25_000, 1, b"-", None), (25_001, 20_000, b"N", None)]),
               with_runs(60_000, [(5_000, 20_000, b"N", None), (25_000 + k - 1, 20_000, b"N", None)]),
               with_runs(50_000, [(20_000, 30_000, b"N", None)]),
               with_runs(50_000, [(3, 49_997, b"N", None)])]
    d = np.concatenate(crafted)
    offs = np.concatenate([[0], np.cumsum([len(x) for x in crafted])]).astype(np.uint64)
    want = oracle.seed_batch(d, offs, seeds, k, m2)
    got = ctx.seed_hash(d, seeds, k, m2, offsets=offs, want_pos=True)
    assert got["total"] == want["total"]
    for key in ("counts", "pos", "hashes"):
        assert (got[key] == want[key]).all(), ("runs of N", key)
    # fixed length: four reads of 100 kbase with non-bases; one clean read of 1 Mbase (beyond the block-tile kernel's LDS)
    k, m2, seeds = 31, 3, [SEED_A, SEED_B]
    L = 100_000
    d = np.concatenate([long_read(L, k, "dirty") for _ in range(4)])
    offs = np.arange(5, dtype=np.uint64) * L
    want = oracle.seed_batch(d, offs, seeds, k, m2)
    got = ctx.seed_hash(d, seeds, k, m2, fixed_len=L, n_reads=4, want_pos=True)
    assert got["total"] == want["total"]
    for key in ("counts", "pos", "hashes"):
        assert (got[key] == want[key]).all(), key
    d = long_read(1_000_000, k, "clean")
    offs = np.array([0, len(d)], dtype=np.uint64)
    want = oracle.seed_batch(d, offs, seeds, k, m2, want_pos=False)
    got = ctx.seed_hash(d, seeds, k, m2, fixed_len=len(d), n_reads=1)
    assert got["total"] == want["total"] == len(d) - k + 1
    assert (got["hashes"] == want["hashes"]).all()
    # the pieces against one wave per read (NTHIP_TUNE_NO_SEED_LONG=1), a read with many cuts refused
    os.environ["NTHIP_TUNE_NO_SEED_LONG"] = "1"
    try:
        whole = nthash_amd.Context(0)
    finally:
        os.environ.pop("NTHIP_TUNE_NO_SEED_LONG", None)
    d = long_read(400_000, k, "dirty")
    offs = np.array([0, len(d)], dtype=np.uint64)
    g1 = ctx.seed_hash(d, seeds, k, m2, offsets=offs, want_pos=True)
    g2 = whole.seed_hash(d, seeds, k, m2, offsets=offs, want_pos=True)
    assert g1["total"] == g2["total"]
    for key in ("counts", "pos", "hashes"):
        assert (g1[key] == g2[key]).all(), key
    whole.close()


def test_seed_long_reads_as_spans(ctx, oracle):
    """nthip_seed_hash_spans (what the FASTA / FASTQ driver calls) with contig-sized records between header lines: the
    long ones are cut into pieces as in nthip_seed_hash; counts, positions, hashes against the oracle"""
    import nthash_amd
    rng = np.random.default_rng(31337)
    alph = np.frombuffer(b"ACGT", dtype=np.uint8)
    k, m2, seeds = 31, 2, [SEED_A, SEED_B]
    reads = []
    for n in (70_000, 150, 33_000, 0, 16_384, 2_500):
        d = alph[rng.integers(0, 4, n)].copy()
        for _ in range(n // 9000):
            ln = int(rng.choice([1, k, 400, 3000]))
            at = int(rng.integers(0, max(1, n - ln)))
            d[at:at + ln] = ord("N")
        reads.append(d.tobytes())
    buf = bytearray()
    starts, ends = [], []
    for i, r in enumerate(reads):
        buf += b">contig%d some text\n" % i
        starts.append(len(buf)); buf += r; ends.append(len(buf))
        buf += b"\n"
    raw = np.frombuffer(bytes(buf), dtype=np.uint8)
    d, offs = concat_reads(reads)
    want = oracle.seed_batch(d, offs, seeds, k, m2)
    cap = max(1, want["total"])
    per = len(seeds) * m2
    sd = nthash_amd.Seeds(ctx, seeds, k)
    d_buf = ctx.malloc(raw.size + 16); d_s = ctx.malloc(8 * len(reads)); d_e = ctx.malloc(8 * len(reads))
    d_h = ctx.malloc(cap * per * 8); d_c = ctx.malloc(8 * len(reads)); d_p = ctx.malloc(4 * cap)
    try:
        ctx.h2d(d_buf, raw); ctx.h2d(d_s, np.array(starts, np.uint64)); ctx.h2d(d_e, np.array(ends, np.uint64))
        tot = ctx.seed_hash_spans_ptr(d_buf, raw.size, d_s, d_e, len(reads), sd, m2, d_h, cap, counts=d_c, pos=d_p)
        assert tot == want["total"]
        h = np.zeros(cap * per, np.uint64); cts = np.zeros(len(reads), np.uint64); ps = np.zeros(cap, np.uint32)
        ctx.d2h(h, d_h); ctx.d2h(cts, d_c); ctx.d2h(ps, d_p)
        assert (cts == want["counts"]).all()
        assert (ps[:tot] == want["pos"]).all()
        assert (h[: tot * per].reshape(-1, per) == want["hashes"]).all()
    finally:
        for ptr in (d_buf, d_s, d_e, d_h, d_c, d_p):
            ctx.free(ptr)
        sd.close()
