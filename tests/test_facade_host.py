"""CPU tests of the host-side logic of the C++ facade: the classes that hash one
caller-chosen base per call (BlindNtHash, BlindSeedNtHash) and parse_seeds never
touch the device, so their reference-recorded API scripts run here without a GPU.
(NtHash / SeedNtHash need the device for roll(): tests/test_gpu_facade.py.)"""
import numpy as np

from conftest import load_golden


def h2i(xs):
    return np.array([int(x, 16) for x in xs], dtype=np.uint64)


def test_blind_nthash_scripts_on_host(facade):
    n = 0
    for c in load_golden("api_scripts.json"):
        if c["cls"] != "BlindNtHash":
            continue
        res = facade.blind_script(c["seq"], c["m"], c["k"], c["pos0"], c["ops"])
        assert [a[0] for a in res] == c["pos"]
        for a, f, r_, hs in zip(res, c["fwd"], c["rev"], c["hashes"]):
            assert a[1] == int(f, 16) and a[2] == int(r_, 16)
            assert (a[3] == h2i(hs)).all()
        n += 1
    assert n >= 3


def test_blind_seed_nthash_scripts_on_host(facade):
    n = 0
    for c in load_golden("api_scripts.json"):
        if c["cls"] != "BlindSeedNtHash":
            continue
        res = facade.blindseed_script(c["seq"], c["seeds"], c["m2"], c["k"], c["pos0"], c["ops"])
        assert [a[0] for a in res] == c["pos"]
        for a, f, r_, hs in zip(res, c["fwd"], c["rev"], c["hashes"]):
            assert (a[1] == h2i(f)).all() and (a[2] == h2i(r_)).all()
            assert (a[3] == h2i(hs)).all()
        n += 1
    assert n >= 3


def test_parse_seeds_on_host(facade):
    for c in load_golden("parse_seeds.json"):
        assert facade.parse_seeds(c["seed"]) == c["dont_care"]


def test_blind_vs_reference_random(facade, reference):
    """random Blind* call sequences, facade (host recurrences of nt_math.hpp) vs the real reference"""
    rng = np.random.default_rng(4)
    for _ in range(200):
        k = int(rng.integers(3, 70))
        seq = "".join("ACGTacgu"[i] for i in rng.integers(0, 8, k + 5))
        ops = "".join(rng.choice(list("RRRBPQ")) + "ACGTN"[int(rng.integers(0, 5))] for _ in range(30))
        m = int(rng.integers(1, 5))
        a = facade.blind_script(seq, m, k, 0, ops)
        b = reference.blind_script(seq, m, k, 0, ops)
        for x, y in zip(a, b):
            assert x[0] == y[0] and x[1] == y[1] and x[2] == y[2] and (x[3] == y[3]).all(), (seq, k, ops)
        half = "".join("1" if rng.random() < 0.6 else "0" for _ in range((k + 1) // 2))
        seeds = [half + half[: k // 2][::-1], "1" * k]
        ops2 = "".join(rng.choice(list("RRB")) + "ACGT"[int(rng.integers(0, 4))] for _ in range(20))
        a = facade.blindseed_script(seq, seeds, m, k, 0, ops2)
        b = reference.blindseed_script(seq, seeds, m, k, 0, ops2)
        for x, y in zip(a, b):
            assert x[0] == y[0] and (x[1] == y[1]).all() and (x[2] == y[2]).all() and (x[3] == y[3]).all(), (seeds, ops2)
