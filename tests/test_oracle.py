"""CPU tests that pin the oracle (oracle/nthash_oracle.c):
  * against the golden vectors the reference's own tests hold
    (tests/tests.cpp:54-57 and :236-240, restated here as data),
  * against fixtures generated from the real reference (tests/golden/*.json),
  * against the real reference itself (oracle/_ref) on randomised inputs,
  * and through the domain's size-independent properties.
"""
import numpy as np
import pytest

from conftest import load_golden
from oracle.pyoracle import concat_reads


def h2i(xs):
    return np.array([int(x, 16) for x in xs], dtype=np.uint64)


def rc(s):
    return s[::-1].translate(str.maketrans("ACGTacgt", "TGCAtgca"))


def test_reference_golden_kmer_vector(oracle):
    # tests/tests.cpp:50-57: ACATGCATGCA, k=5, 3 hashes, positions 1 and 2
    d, offs = concat_reads(["ACATGCATGCA"])
    r = oracle.kmer_batch(d, offs, 5, 3)
    assert [hex(x) for x in r["hashes"][1]] == ["0x38cc00f940aebdae", "0xab7e1b110e086fc6", "0x11a1818bcfdd553"]
    assert [hex(x) for x in r["hashes"][2]] == ["0x603a48c5a11c794a", "0xe66016e61816b9c4", "0xc5b13cb146996ffe"]


def test_reference_golden_seed_vector(oracle):
    # tests/tests.cpp:231-240: seed 11100111, 3 hashes, positions 0..2
    d, offs = concat_reads(["ACATGCATGCA"])
    r = oracle.seed_batch(d, offs, ["11100111"], 8, 3)
    want = [["0x10be4904ad8de5d", "0x3e29e4f4c991628c", "0x3f35c984b13feb20"],
            ["0x8200a7aa3eaf17c8", "0x344198402f4c2a9c", "0xb6423fe62e69c40c"],
            ["0x3ce8adcbeaa56532", "0x162e91a4dbedbf11", "0x53173f786a031f45"]]
    for i in range(3):
        assert [hex(x) for x in r["hashes"][i]] == want[i]


def test_survey_known_answers(oracle):
    # SURVEY.md App. C (captured from the reference): palindrome, k=31 m=4 row 0
    d, offs = concat_reads(["ACGT"])
    r = oracle.kmer_batch(d, offs, 4, 1, want_strands=True)
    assert hex(r["fwd"][0]) == "0x4b21efd76bfc8c8a" and r["fwd"][0] == r["rev"][0]
    assert hex(r["hashes"][0][0]) == "0x9643dfaed7f91914"
    d, offs = concat_reads(["CACTCGGCCACACACACACACACACACCCTCACACACACAAAACGCACAC"])
    r = oracle.kmer_batch(d, offs, 31, 4, want_strands=True)
    assert hex(r["fwd"][0]) == "0xa8df5e7192727f56" and hex(r["rev"][0]) == "0x2589789de530587c"
    assert [hex(x) for x in r["hashes"][0]] == ["0xce68d70f77a2d7d2", "0x6847659ef9498c",
                                                "0x952dc22588b15b2d", "0x6396994ad678dd4e"]
    # multipliers at k=31 (i ^ k*MULTISEED)
    assert [hex(int(oracle.extend(1, 0, 31, 4)[i])) for i in (1, 2, 3)] == [
        hex((0x85d74a0572469d47 ^ (0x85d74a0572469d47 >> 27))),
        hex((0x85d74a0572469d44 ^ (0x85d74a0572469d44 >> 27))),
        hex((0x85d74a0572469d45 ^ (0x85d74a0572469d45 >> 27)))]


def test_split_rotate_identities(oracle):
    rng = np.random.default_rng(7)
    for x in rng.integers(0, 2**63, 200, dtype=np.uint64):
        x = int(x) * 2 + 1 & (2**64 - 1)
        assert oracle.sror(oracle.srol(x)) == x
        y = x
        for d in range(0, 70):
            assert oracle.srol_n(x, d) == y
            y = oracle.srol(y)
        assert oracle.srol_n(x, 1023) == x  # period lcm(31,33)


def test_golden_kmer_cases(oracle):
    for c in load_golden("kmer_cases.json"):
        d, offs = concat_reads(c["reads"])
        r = oracle.kmer_batch(d, offs, c["k"], c["m"], want_strands=True)
        assert r["counts"].tolist() == c["counts"]
        assert r["pos"].tolist() == c["pos"]
        assert (r["hashes"].ravel() == h2i(c["hashes"])).all()
        assert (r["fwd"] == h2i(c["fwd"])).all() and (r["rev"] == h2i(c["rev"])).all()


def test_golden_seed_cases(oracle):
    for c in load_golden("seed_cases.json"):
        d, offs = concat_reads(c["reads"])
        r = oracle.seed_batch(d, offs, c["seeds"], c["k"], c["m2"])
        assert r["counts"].tolist() == c["counts"], c["reads"]
        assert r["pos"].tolist() == c["pos"]
        assert (r["hashes"].ravel() == h2i(c["hashes"])).all()


def test_golden_synth_checksums(oracle):
    for c in load_golden("synth_checksums.json"):
        data = oracle.synth_reads(0, c["n_reads"], c["len"], c["seed"])
        offs = np.arange(c["n_reads"] + 1, dtype=np.uint64) * c["len"]
        if c["kind"] == "kmer":
            assert data[: c["len"]].tobytes().decode() == c["first_read"]
            r = oracle.kmer_batch(data, offs, c["k"], c["m"], want_pos=False)
        else:
            r = oracle.seed_batch(data, offs, c["seeds"], c["k"], c["m2"], want_pos=False)
        s, x = oracle.checksum(r["hashes"])
        assert r["total"] == c["total"]
        assert format(s, "016x") == c["sum"] and format(x, "016x") == c["xor"]


def test_golden_nthash_scripts(oracle):
    n = 0
    for c in load_golden("api_scripts.json"):
        if c["cls"] != "NtHash":
            continue
        res = oracle.nthash_script(c["seq"], c["m"], c["k"], c["pos0"], c["ops"])
        assert [a[0] for a in res] == c["ret"]
        assert [a[1] for a in res] == c["pos"], c
        for a, f, r_, hs in zip(res, c["fwd"], c["rev"], c["hashes"]):
            assert a[2] == int(f, 16) and a[3] == int(r_, 16)
            if hs is not None:  # recorded only for calls that returned true
                assert (a[4] == h2i(hs)).all()
        n += 1
    assert n >= 10


def test_oracle_vs_reference_random(oracle, reference):
    rng = np.random.default_rng(11)
    alph = np.frombuffer(b"ACGTacgtUuNnRYKM-*\x00", dtype=np.uint8)
    for it in range(400):
        k = int(rng.integers(3, 70))
        m = int(rng.integers(1, 6))
        reads = []
        for _ in range(int(rng.integers(1, 5))):
            L = int(rng.integers(0, 150))
            p = rng.random()
            if p < 0.5:
                idx = rng.integers(0, 4, L)
            else:
                idx = np.where(rng.random(L) < 0.9, rng.integers(0, 10, L), rng.integers(10, len(alph), L))
            reads.append(alph[idx].tobytes())
        d, offs = concat_reads(reads)
        a = oracle.kmer_batch(d, offs, k, m, want_strands=True)
        b = reference.kmer_batch(d, offs, k, m, want_strands=True)
        assert a["total"] == b["total"]
        for key in ("hashes", "pos", "fwd", "rev", "counts"):
            assert (a[key] == b[key]).all(), (it, key)
        seeds = []
        for _s in range(int(rng.integers(1, 4))):
            half = "".join("1" if rng.random() < 0.6 else "0" for _ in range((k + 1) // 2))
            seeds.append(half + half[: k // 2][::-1])
        a = oracle.seed_batch(d, offs, seeds, k, m)
        b = reference.seed_batch(d, offs, seeds, k, m)
        assert a["total"] == b["total"]
        for key in ("hashes", "pos", "counts"):
            assert (a[key] == b[key]).all(), (it, key, seeds)


def test_oracle_vs_reference_long_k_many_hashes(oracle, reference):
    """the restatement against the REAL reference beyond k = 64 and m = 8 (the GPU's Horner first window, computed
    multipliers and long-k validity are checked against this oracle): k up to 1100 crosses the 1023 period of the
    split rotate, m up to 255, spaced seeds up to k = 200, with non-bases"""
    rng = np.random.default_rng(12)
    alph = np.frombuffer(b"ACGTacgtUuNnRY-", dtype=np.uint8)
    for it, (k, m) in enumerate([(65, 1), (96, 2), (100, 12), (127, 3), (128, 1), (200, 9), (255, 2), (300, 40),
                                 (1000, 1), (1023, 2), (1024, 1), (1100, 3), (31, 255), (64, 100)]):
        reads = []
        for _ in range(4):
            L = int(rng.integers(0, k + 400))
            idx = np.where(rng.random(L) < 0.995, rng.integers(0, 10, L), rng.integers(10, len(alph), L))
            reads.append(alph[idx].tobytes())
        d, offs = concat_reads(reads)
        a = oracle.kmer_batch(d, offs, k, m, want_strands=True)
        b = reference.kmer_batch(d, offs, k, m, want_strands=True)
        assert a["total"] == b["total"], (k, m)
        for key in ("hashes", "pos", "fwd", "rev", "counts"):
            assert (a[key] == b[key]).all(), (k, m, key)
        if k <= 200:
            seeds = []
            for _s in range(2):
                half = "".join("1" if rng.random() < 0.6 else "0" for _ in range((k + 1) // 2))
                seeds.append(half + half[: k // 2][::-1])
            m2 = min(m, 5)
            a = oracle.seed_batch(d, offs, seeds, k, m2)
            b = reference.seed_batch(d, offs, seeds, k, m2)
            assert a["total"] == b["total"], (k, "seeds")
            for key in ("hashes", "pos", "counts"):
                assert (a[key] == b[key]).all(), (k, key, "seeds")


def test_properties_canonical_and_full_care_seed(oracle):
    rng = np.random.default_rng(5)
    for _ in range(50):
        L = int(rng.integers(40, 120))
        k = int(rng.integers(3, 40))
        s = "".join("ACGT"[i] for i in rng.integers(0, 4, L))
        d1, o1 = concat_reads([s])
        d2, o2 = concat_reads([rc(s)])
        a = oracle.kmer_batch(d1, o1, k, 3)
        b = oracle.kmer_batch(d2, o2, k, 3)
        # strand symmetry: hashes of the reverse complement are the reversed stream
        assert (a["hashes"] == b["hashes"][::-1]).all()
        # a full-care seed is the k-mer hash (tests/tests.cpp:447-463)
        c = oracle.seed_batch(d1, o1, ["1" * k], k, 3)
        assert (a["hashes"] == c["hashes"]).all()


def test_get_blocks_examples(oracle):
    # SURVEY.md 8(a16) examples, verified against the reference there
    assert oracle.get_blocks("1010101010101010101010101010101") == ([], list(range(0, 31, 2)))
    assert oracle.get_blocks("1101101101101101011011011011011") == (
        [(0, 31)], [2, 5, 8, 11, 14, 16, 19, 22, 25, 28])
    assert oracle.get_blocks("11100111") == ([(0, 3), (5, 8)], [])
    assert oracle.get_blocks("101101") == ([(2, 4)], [0, 5])


def test_bench_checksum_fixture_small_entries_vs_oracle(oracle):
    """tests/golden/bench_checksums.json (made by the REAL reference, gen_bench_checksums.py) is what bench.py
    verifies the full-size streams against; its small entries pin the file's format and the shard layout
    (first_read = rank * 125 M) against the C restatement."""
    from oracle.pyoracle import var_reads
    for e in load_golden("bench_checksums.json"):
        if e.get("len_min"):  # the rule of ref_synth_var_checksum (variable lengths; or one length: "c2_dirty"), restated in numpy
            if e["n_reads"] > 50_000:
                continue
            n = e["n_reads"]
            lens, has_n, n_pos = var_reads(e["first_read"], n, e["len_min"], e["len"], e["seed"])
            full = oracle.synth_reads(e["first_read"], n, e["len"], e["seed"]).reshape(n, e["len"]).copy()
            for r in np.nonzero(has_n)[0]:
                full[r, int(n_pos[r])] = ord("N")
            d, offs = concat_reads([full[r, : int(lens[r])].tobytes() for r in range(n)])
            r = oracle.kmer_batch(d, offs, e["k"], e["m"], want_pos=False)
            s, x = oracle.checksum(r["hashes"])
            assert (int(r["total"]), format(s, "016x"), format(x, "016x")) == (e["total"], e["sum"], e["xor"])
            assert has_n.sum() > 0 and r["total"] < int(np.maximum(lens.astype(np.int64) - e["k"] + 1, 0).sum())
            continue
        if e["n_reads"] > 50_000:
            assert e["total"] == e["n_reads"] * (e["len"] - e["k"] + 1)
            continue
        data = oracle.synth_reads(e["first_read"], e["n_reads"], e["len"], e["seed"])
        offs = np.arange(e["n_reads"] + 1, dtype=np.uint64) * e["len"]
        if e["seeds"]:
            r = oracle.seed_batch(data, offs, e["seeds"], e["k"], e["m"], want_pos=False)
        else:
            r = oracle.kmer_batch(data, offs, e["k"], e["m"], want_pos=False)
        s, x = oracle.checksum(r["hashes"])
        assert (int(r["total"]), format(s, "016x"), format(x, "016x")) == (e["total"], e["sum"], e["xor"])


def test_reference_synth_checksum_matches_batch_checksum(oracle, reference):
    """the OpenMP on-the-fly checksum helper of the reference shim == checksum of its batch output"""
    if not reference.has_synth:
        pytest.skip("prebuilt oracle/_ref without the synthetic-workload helpers")
    for (first, n, L, k, m, seeds) in [(7, 3000, 150, 31, 2, None), (10**9, 500, 250, 31, 3, ["1" * 15 + "0" + "1" * 15])]:
        data = reference.synth_reads(first, n, L, 42)
        offs = np.arange(n + 1, dtype=np.uint64) * L
        r = (reference.seed_batch(data, offs, seeds, k, m, want_pos=False) if seeds
             else reference.kmer_batch(data, offs, k, m, want_pos=False))
        s, x = reference.checksum(r["hashes"])
        assert reference.synth_checksum(first, n, L, k, m, seeds=seeds, threads=3) == (s, x, int(r["total"]))


def test_golden_seed_extend_cases(oracle, reference):
    """nto_seed_extend -- the 4 successors / predecessors of a window under BlindSeedNtHash::roll(c) / roll_back(c)
    (src/seed.cpp:701-737) -- against the fixtures recorded from the real reference (tests/golden/gen_golden.py, section 7:
    seeds with monomers, whose roll_back reads them from the window it leaves; the don't-care description; k to 100), and
    against the real reference itself on fresh random cases where it is built"""
    cases = load_golden("seed_extend_cases.json")
    assert len(cases) >= 40
    for c in cases:
        me, nx, pv = oracle.seed_extend(c["kmer"], c["seeds"], c["m2"])
        assert (me == h2i(c["self"])).all(), c["kmer"]
        for b in range(4):
            assert (nx[b] == h2i(c["next"][b])).all(), (c["kmer"], c["seeds"], "next", b)
            assert (pv[b] == h2i(c["prev"][b])).all(), (c["kmer"], c["seeds"], "prev", b)
    if reference is None:
        return
    rng = np.random.default_rng(3)
    for _ in range(60):
        k = int(rng.integers(4, 70))
        seeds = ["".join("10"[int(x)] for x in rng.integers(0, 2, k)) for _ in range(int(rng.integers(1, 4)))]
        m2 = int(rng.integers(1, 4))
        kmer = "".join("ACGT"[j] for j in rng.integers(0, 4, k))
        me, nx, pv = oracle.seed_extend(kmer, seeds, m2)
        assert (me == reference.blindseed_script(kmer, seeds, m2, k, 0, "")[0][3]).all()
        for b, ch in enumerate("ACGT"):
            assert (nx[b] == reference.blindseed_script(kmer, seeds, m2, k, 0, "R" + ch)[1][3]).all()
            assert (pv[b] == reference.blindseed_script(kmer, seeds, m2, k, 0, "B" + ch)[1][3]).all()
