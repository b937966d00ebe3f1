"""GPU tests of the C++ host facade (include/nthash/nthash.hpp + libnthash.so).

The driver is oracle/ref_shim.cpp -- the same extern "C" script runner that
drives the REAL reference when fixtures are generated -- compiled here against
nthash_amd's header and library instead.  That it compiles unchanged is the
source-level drop-in check; the recorded API scripts (tests/golden/api_scripts.json,
produced by the real reference) are the behavioural one: every return value,
get_pos(), strand hash and hashes() array must match call for call.
"""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden
from oracle.pyoracle import concat_reads

pytestmark = pytest.mark.gpu


def h2i(xs):
    return np.array([int(x, 16) for x in xs], dtype=np.uint64)


def test_facade_is_ours(facade):
    assert facade.fn_name() == "ntHash_v2"
    maps = open("/proc/self/maps").read()
    assert "libnthash.so" in maps and "libnthash_hip.so" in maps


def test_nthash_scripts(facade):
    n = 0
    for c in load_golden("api_scripts.json"):
        if c["cls"] != "NtHash":
            continue
        res = facade.nthash_script(c["seq"], c["m"], c["k"], c["pos0"], c["ops"])
        assert [a[0] for a in res] == c["ret"], c["seq"]
        assert [a[1] for a in res] == c["pos"], (c["seq"], c["ops"])
        for a, f, r_, hs, ret in zip(res, c["fwd"], c["rev"], c["hashes"], c["ret"]):
            assert a[2] == int(f, 16) and a[3] == int(r_, 16)
            if hs is not None:
                assert (a[4] == h2i(hs)).all()
        n += 1
    assert n >= 10


def test_blind_scripts(facade):
    for c in load_golden("api_scripts.json"):
        if c["cls"] != "BlindNtHash":
            continue
        res = facade.blind_script(c["seq"], c["m"], c["k"], c["pos0"], c["ops"])
        assert [a[0] for a in res] == c["pos"]
        for a, f, r_, hs in zip(res, c["fwd"], c["rev"], c["hashes"]):
            assert a[1] == int(f, 16) and a[2] == int(r_, 16)
            assert (a[3] == h2i(hs)).all()


def test_seed_scripts(facade):
    for c in load_golden("api_scripts.json"):
        if c["cls"] != "SeedNtHash":
            continue
        res = facade.seed_script(c["seq"], c["seeds"], c["m2"], c["k"], c["pos0"], c["ops"])
        assert [a[0] for a in res] == c["ret"]
        assert [a[1] for a in res] == c["pos"]
        for a, f, r_, hs in zip(res, c["fwd"], c["rev"], c["hashes"]):
            if hs is None:
                continue
            assert (a[2] == h2i(f)).all() and (a[3] == h2i(r_)).all(), (c["seq"], c["ops"])
            assert (a[4] == h2i(hs)).all()


def test_blindseed_scripts(facade):
    for c in load_golden("api_scripts.json"):
        if c["cls"] != "BlindSeedNtHash":
            continue
        res = facade.blindseed_script(c["seq"], c["seeds"], c["m2"], c["k"], c["pos0"], c["ops"])
        assert [a[0] for a in res] == c["pos"]
        for a, f, r_, hs in zip(res, c["fwd"], c["rev"], c["hashes"]):
            assert (a[1] == h2i(f)).all() and (a[2] == h2i(r_)).all()
            assert (a[3] == h2i(hs)).all()


def test_facade_batches_match_golden(facade):
    """while (h.roll()) over whole reads through the facade == reference fixtures"""
    for c in load_golden("kmer_cases.json")[:25]:
        d, offs = concat_reads(c["reads"])
        r = facade.kmer_batch(d, offs, c["k"], c["m"], want_strands=True)
        assert r["counts"].tolist() == c["counts"] and r["pos"].tolist() == c["pos"]
        assert (r["hashes"].ravel() == h2i(c["hashes"])).all()
        assert (r["fwd"] == h2i(c["fwd"])).all() and (r["rev"] == h2i(c["rev"])).all()
    import contextlib
    for c in load_golden("seed_cases.json")[:25]:
        d, offs = concat_reads(c["reads"])
        r = facade.seed_batch(d, offs, c["seeds"], c["k"], c["m2"])
        assert r["counts"].tolist() == c["counts"] and r["pos"].tolist() == c["pos"]
        assert (r["hashes"].ravel() == h2i(c["hashes"])).all()


def test_parse_seeds(facade):
    for c in load_golden("parse_seeds.json"):
        assert facade.parse_seeds(c["seed"]) == c["dont_care"]


def test_facade_vs_oracle_scripts_random(facade, oracle):
    """random call sequences on NtHash: facade (GPU stream + host recurrences) vs the oracle"""
    rng = np.random.default_rng(99)
    for _ in range(60):
        L = int(rng.integers(20, 120))
        k = int(rng.integers(3, min(L, 40)))
        alph = "ACGTACGTACGTNacgu"
        seq = "".join(alph[i] for i in rng.integers(0, len(alph), L))
        ops = "".join(rng.choice(list("rrrrrbpq")) for _ in range(80))
        m = int(rng.integers(1, 4))
        pos0 = int(rng.integers(0, L - k + 1))
        a = facade.nthash_script(seq, m, k, pos0, ops)
        b = oracle.nthash_script(seq, m, k, pos0, ops)
        seen_true = False
        for x, y in zip(a, b):
            if y[1] > L - k:
                break  # a failed skip left pos past the last window: the reference reads out of bounds from here on
            assert x[0] == y[0] and x[1] == y[1], (seq, k, ops)
            seen_true |= bool(y[0])
            if seen_true:  # hashes()/strands are undefined before the first success
                assert x[2] == y[2] and x[3] == y[3] and (x[4] == y[4]).all(), (seq, k, ops)


def test_reference_own_test_program_passes_on_our_library():
    """the reference's tests/tests.cpp, unchanged, compiled against include/nthash/nthash.hpp and linked
    with libnthash.so (oracle/Makefile target ref_tests; the binary travels, the source does not):
    every block of it must pass with the hashes coming from the MI355X"""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_tests_on_facade")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_tests_on_facade not built (needs the reference tree at build time)")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])


def test_long_sequence_streams_in_windows(facade, oracle):
    """NtHash / SeedNtHash over a 30 kbase sequence with non-bases: the facade fetches the device stream window by
    window (NTHASH_AMD_WINDOW; the re-run below uses 1024 positions, i.e. ~30 window changes) -- every roll() must
    agree with the oracle's iterator"""
    rng = np.random.default_rng(2024)
    L, k = 30_000, 31
    alph = np.frombuffer(b"ACGT", dtype=np.uint8)
    seq = alph[rng.integers(0, 4, L)].copy()
    seq[rng.choice(L, 40, replace=False)] = ord("N")
    seq[12_000:12_050] = ord("N")           # a stretch longer than k
    s = seq.tobytes().decode()
    ops = "r" * (L - k + 40)
    a = facade.nthash_script(s, 2, k, 0, ops)
    b = oracle.nthash_script(s, 2, k, 0, ops)
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x[0] == y[0] and x[1] == y[1]
        if y[0]:
            assert x[2] == y[2] and x[3] == y[3] and (x[4] == y[4]).all()
    # roll back across a window boundary and forward again
    ops2 = "r" * 2100 + "b" * 1500 + "r" * 3000
    a = facade.nthash_script(s, 1, k, 500, ops2)
    b = oracle.nthash_script(s, 1, k, 500, ops2)
    for x, y in zip(a, b):
        assert x[0] == y[0] and x[1] == y[1] and (not y[0] or (x[2] == y[2] and (x[4] == y[4]).all()))
    # spaced seeds: against the batch oracle (positions and hashes of every emitted window)
    seeds = ["1101101101101101011011011011011", "1010101010101010101010101010101"]
    res = facade.seed_script(s, seeds, 2, k, 0, "r" * (L - k + 10))
    want = oracle.seed_batch(seq, np.array([0, L], dtype=np.uint64), seeds, k, 2, want_pos=True)
    got_pos = [r_[1] for r_ in res if r_[0]]
    assert got_pos == [int(p) for p in want["pos"]]
    got_h = np.array([r_[4] for r_ in res if r_[0]], dtype=np.uint64).reshape(-1, 4)
    assert (got_h == want["hashes"].reshape(-1, 4)).all()


def _rerun_suite(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([os.sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_facade.py"), "-m", "gpu",
                        "-q", "-x", "-k", "not again and not benchmark"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_facade_suite_again_in_host_roll_mode():
    """the default mode: sequences of up to 32768 bases never go to the device (host recurrences); everything above
    must give the same answers"""
    _rerun_suite({"NTHASH_AMD_FORCE_DEVICE": "0"})


def test_facade_suite_again_with_tiny_windows():
    """every sequence on the device, in windows of 1024 positions"""
    _rerun_suite({"NTHASH_AMD_FORCE_DEVICE": "1", "NTHASH_AMD_WINDOW": "1024"})


def test_facade_suite_again_without_the_prefetch_thread():
    """windows of 1024 positions fetched one after the other on the user thread (NTHASH_AMD_PREFETCH=0); the run above has
    the helper thread hash window w + 1 while w is walked"""
    _rerun_suite({"NTHASH_AMD_FORCE_DEVICE": "1", "NTHASH_AMD_WINDOW": "1024", "NTHASH_AMD_PREFETCH": "0"})


BATCH_DRIVER = r"""
#include <nthash/nthash.hpp>
#include <cstdio>
#include <iostream>
#include <string>
int main(int argc, char** argv) {
  const unsigned m = std::stoi(argv[1]), k = std::stoi(argv[2]);
  nthash::BatchNtHash b(m, k);
  std::string line;
  while (std::getline(std::cin, line)) b.add(line);
  b.run();
  std::printf("%zu %llu\n", b.size(), (unsigned long long)b.total());
  for (size_t r = 0; r < b.size(); ++r) {
    std::printf("%zu", b.count(r));
    for (size_t j = 0; j < b.count(r); ++j) {
      std::printf(" %u", b.positions(r)[j]);
      for (unsigned h = 0; h < m; ++h) std::printf(" %016llx", (unsigned long long)b.hashes(r)[j * m + h]);
    }
    std::printf("\n");
  }
  b.clear();
  b.add("ACGTACGTACGTACGTACGTACGTACGTACGTACGTAC");
  b.run();
  std::printf("%zu %llu\n", b.size(), (unsigned long long)b.total());
  return 0;
}
"""


@pytest.mark.parametrize("devices", [None, "0,0,0", "all"])
def test_batch_nthash_helper(built_lib, oracle, tmp_path, devices):
    """nthash::BatchNtHash (our addition to the header): many reads, one device call, results in roll() order; with
    NTHASH_AMD_DEVICES the batch is cut over several devices (nthip_multi_*; here the one GPU several times)"""
    env = dict(os.environ)
    env.pop("NTHASH_AMD_DEVICES", None)
    if devices:
        env["NTHASH_AMD_DEVICES"] = devices
    lib = os.path.join(ROOT, "nthash_amd", "lib")
    src = tmp_path / "batch.cpp"
    src.write_text(BATCH_DRIVER)
    exe = tmp_path / "batch"
    subprocess.check_call(["g++", "-std=c++17", "-O1", f"-I{os.path.join(ROOT, 'include')}", str(src), "-o", str(exe),
                           f"-L{lib}", "-lnthash", "-lnthash_hip", f"-Wl,-rpath,{lib}"])
    rng = np.random.default_rng(7)
    alph = "ACGTACGTACGTNacgtu"
    reads = ["".join(alph[i] for i in rng.integers(0, len(alph), int(rng.integers(0, 300)))) for _ in range(400)]
    reads[3] = ""            # (an empty line is an empty read)
    m, k = 3, 21
    r = subprocess.run([str(exe), str(m), str(k)], input="\n".join(reads) + "\n", capture_output=True, text=True, timeout=300,
                       env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().split("\n")
    d, offs = concat_reads([x.encode() for x in reads])
    want = oracle.kmer_batch(d, offs, k, m, want_pos=True)
    n, total = (int(x) for x in lines[0].split())
    assert n == len(reads) and total == want["total"]
    first = 0
    for i in range(n):
        f = lines[1 + i].split()
        cnt = int(f[0])
        assert cnt == int(want["counts"][i])
        for j in range(cnt):
            base = 1 + j * (1 + m)
            assert int(f[base]) == int(want["pos"][first + j])
            assert [int(x, 16) for x in f[base + 1: base + 1 + m]] == [int(x) for x in want["hashes"][first + j]]
        first += cnt
    assert lines[1 + n].split() == ["1", str(38 - k + 1)]


def test_reference_benchmark_program_runs_fast_on_our_library():
    """the reference's examples/benchmark.cpp (1 M objects of 100 bp, NtHash(seq, 3, 64)), unchanged, on our
    library in its default mode: one object per short read must not cost a device round trip each"""
    import time
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_benchmark_on_facade")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_benchmark_on_facade not built (needs the reference tree at build time)")
    env = dict(os.environ)
    env["NTHASH_AMD_FORCE_DEVICE"] = "0"
    t0 = time.time()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    dt = time.time() - t0
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    elapsed = float(r.stdout.split()[0])   # the program prints the seconds of its hashing loop, then its checksum
    assert elapsed < 2.0, (elapsed, dt)
