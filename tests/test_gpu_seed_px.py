"""seed_px_kernel (nthash_amd/csrc/seed_px_kernel.hpp): spaced seeds as sparse sums over scanned term arrays -- the
reference's SeedNtHash stream (src/seed.cpp:449-544; masked formula SURVEY.md App. A.4) with a cost that follows the seed:
2 reads per care run from the prefix XOR, a handful for a seed that repeats under a shift.  Every array form the planner
can choose is forced against the oracle's seed_batch (NTHIP_TUNE_SEED_PX_ARRAY), as is the planner's own choice."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx(**env):
    import nthash_amd
    for k, v in env.items():
        os.environ[k] = str(v)
    try:
        return nthash_amd.Context(0)
    finally:
        for k in env:
            os.environ.pop(k, None)


def blocky(k, gaps):  # care everywhere but in the gaps [(start, length)]
    s = np.ones(k, dtype=bool)
    for a, n in gaps:
        s[a:a + n] = False
    return "".join("1" if b else "0" for b in s)


def periodic(k, period, on):
    return "".join("1" if (i % period) < on else "0" for i in range(k))


C4 = ["1010101010101010101010101010101", "1101101101101101011011011011011"]
CASES = [  # (seeds, m2, L, n_reads)
    (C4, 3, 250, 1000),
    (C4, 1, 150, 777),
    ([blocky(128, [(40, 48)])], 1, 250, 700),
    ([blocky(160, [(30, 20), (110, 20)])], 1, 300, 600),
    ([blocky(128, [(20, 5), (60, 8), (100, 9)]), blocky(128, [(64, 1)])], 2, 251, 515),
    ([blocky(64, [(10, 44)])], 3, 150, 1000),
    ([blocky(31, [(0, 3), (15, 1), (28, 3)])], 1, 100, 300),   # don't-cares at both ends, a gap of one
    ([blocky(48, [(5, 1), (7, 1), (9, 30), (41, 1), (43, 1)])], 2, 97, 129),  # monomers
    ([blocky(200, [(50, 100)]), blocky(200, [(10, 180)])], 1, 1000, 70),
    (["1" * 40], 1, 77, 2000),                                # no gap at all: a k-mer
    ([blocky(24, [(8, 8)])], 8, 24, 400),                      # one window per read
    ([blocky(31, [(3 + i, 2), (20, 4)]) for i in range(6)], 1, 150, 300),   # six seeds
    ([blocky(64, [(20, 10)]), blocky(64, [(5, 5), (50, 3)]), blocky(64, [(31, 2)])], 1, 100, 257),
    ([periodic(64, 2, 1)[:63] + "1"], 1, 150, 200),            # 32 monomers
    ([periodic(121, 11, 8)], 1, 250, 400),                     # blocks of period 11
    ([periodic(31, 3, 2), periodic(31, 5, 3), "1" * 31], 2, 151, 333),
    ([blocky(40, [(10, 5)])], 1, 1900, 9),                     # one read per tile
]


def _check(c, oracle, seeds, m2, L, n, expect_px=True, kernel="seed_px_kernel"):
    k = len(seeds[0])
    data = oracle.synth_reads(29, n, L, k + len(seeds))
    offs = np.arange(n + 1, dtype=np.uint64) * L
    want = oracle.seed_batch(data, offs, seeds, k, m2, want_pos=False)
    c.set_profiling(True)
    got = c.seed_hash(data, seeds, k, m2, fixed_len=L, n_reads=n)
    name = c.last_kernel_ms()[1]
    c.set_profiling(False)
    if expect_px:
        assert name == kernel, (name, k, seeds)
    assert got["total"] == want["total"] == n * (L - k + 1)
    bad = np.nonzero(got["hashes"].ravel() != want["hashes"].ravel())[0]
    assert bad.size == 0, (k, seeds, m2, L, n, bad[:8], bad.size)
    return data, offs


@pytest.mark.parametrize("form", ["planned", "terms", "prefix", "mixed", "prefix_d2", "prefix_d3", "prefix_d11", "terms_d2",
                                  "terms_d5", "prefix_d16", "terms_d33"])
def test_seed_px_every_array_form_vs_oracle(oracle, form):
    force = {"planned": None, "terms": 0, "prefix": 1, "mixed": 2}.get(form, -1)
    if force == -1:
        kind, d = form.split("_d")
        force = (3 if kind == "prefix" else 100) + int(d)
    env = {"NTHIP_TUNE_SEED_PX": 1, "NTHIP_TUNE_SEED_PS": 2}
    if force is not None:
        env["NTHIP_TUNE_SEED_PX_ARRAY"] = force + 1
    c = _ctx(**env)
    rng = np.random.default_rng(23)
    for seeds, m2, L, n in CASES:
        k = len(seeds[0])
        if force is not None and force >= 3 and (force - 100 if force >= 100 else force - 3) >= k:
            continue
        data, offs = _check(c, oracle, seeds, m2, L, n)
        # a batch with an N falls back to the kernels that know SeedNtHash's position state machine (App. B Q3)
        dirty = data.copy()
        dirty[rng.choice(n * L, 3, replace=False)] = ord("N")
        want = oracle.seed_batch(dirty, offs, seeds, k, m2)
        got = c.seed_hash(dirty, seeds, k, m2, fixed_len=L, n_reads=n, want_pos=True)
        assert got["total"] == want["total"]
        for key in ("counts", "pos", "hashes"):
            assert (got[key] == want[key]).all(), (key, k, seeds)
    c.close()


def test_seed_px_tile_shapes_and_alignment(oracle):
    """reads per tile from 1 to the plan's most, one to eight waves, device buffers that start off a 16-byte boundary,
    batches smaller than a tile, values per window that leave a tile's first value anywhere in its 128-byte line"""
    import nthash_amd
    for reads, waves in ((1, 1), (2, 3), (3, 8), (5, 2), (0, 0)):
        env = {"NTHIP_TUNE_SEED_PX": 1, "NTHIP_TUNE_SEED_PS": 2}
        if reads:
            env.update(NTHIP_TUNE_SEED_PX_READS=reads, NTHIP_TUNE_SEED_PX_WAVES=waves)
        c = _ctx(**env)
        for seeds, m2, L, n in ((C4, 3, 250, 301), (C4[:1], 1, 150, 1), ([blocky(64, [(10, 44)])] * 3, 1, 100, 67),
                                ([blocky(31, [(4, 9)])], 5, 35, 999), ([blocky(96, [(32, 32)])], 2, 131, 250)):
            _check(c, oracle, seeds, m2, L, n)
        seeds, m2, L, n, k = C4, 3, 250, 4001, 31
        data = oracle.synth_reads(3, n, L, 8)
        want = oracle.seed_batch(data, np.arange(n + 1, dtype=np.uint64) * L, seeds, k, m2, want_pos=False)["hashes"].ravel()
        sd = nthash_amd.Seeds(c, seeds, k)
        for shift in (0, 5, 16, 31):
            d_in, d_out = c.malloc(n * L + 64), c.malloc(n * (L - k + 1) * 6 * 8)
            c.h2d(d_in + shift, data)
            tot = c.seed_hash_ptr(d_in + shift, 0, n, L, 0, sd, m2, d_out, n * (L - k + 1))
            assert tot == n * (L - k + 1)
            got = np.zeros(tot * 6, np.uint64)
            c.d2h(got, d_out)
            assert (got == want).all(), shift
            c.free(d_in); c.free(d_out)
        sd.close()
        c.close()


def test_seed_px_not_taken_outside_its_shapes(oracle):
    """strided reads, more hashes per seed than the kernel's runtime table and reads longer than a slab are not sent
    there even when forced; the streams are the oracle's all the same"""
    c = _ctx(NTHIP_TUNE_SEED_PX=1, NTHIP_TUNE_SEED_PS=2)
    seeds = [blocky(96, [(32, 32)])]
    L, stride, n = 200, 50, 333
    data = oracle.synth_reads(31, 1, stride * (n - 1) + L, 5)
    rows = np.concatenate([data[i * stride:i * stride + L] for i in range(n)])
    want = oracle.seed_batch(rows, np.arange(n + 1, dtype=np.uint64) * L, seeds, 96, 2, want_pos=False)
    c.set_profiling(True)
    got = c.seed_hash(data, seeds, 96, 2, fixed_len=L, n_reads=n, stride=stride)
    assert c.last_kernel_ms()[1] != "seed_px_kernel"
    assert (got["hashes"] == want["hashes"]).all()
    c.set_profiling(False)
    _check(c, oracle, C4, 9, 150, 200, expect_px=False)
    c.set_profiling(True)
    data = oracle.synth_reads(5, 9, 3000, 1)
    seeds = [blocky(40, [(10, 5)])]
    want = oracle.seed_batch(data, np.arange(10, dtype=np.uint64) * 3000, seeds, 40, 1, want_pos=False)
    got = c.seed_hash(data, seeds, 40, 1, fixed_len=3000, n_reads=9)
    assert c.last_kernel_ms()[1] != "seed_px_kernel"
    assert (got["hashes"] == want["hashes"]).all()
    c.close()


PS_CASES = CASES[:-1] + [
    ([blocky(40, [(10, 5)])], 1, 1900, 9),                     # 59 windows per segment: not this kernel's
    ([blocky(64, [(20, 10)])] * 3, 1, 100, 1003),              # 37 windows per read
    ([blocky(31, [(4, 9)])], 5, 35, 999),                      # 5 windows per read
    (["1" * 21], 2, 501, 130),
]


@pytest.mark.parametrize("form,lanes", [("planned", 0), ("terms", 0), ("prefix", 0), ("mixed", 0), ("planned", 16), ("mixed", 32),
                                        ("prefix", 64)])
def test_seed_ps_segment_kernel_vs_oracle(oracle, form, lanes):
    """seed_ps_kernel: the same sums from per-read arrays in the transposed layout, a lane per segment of W positions /
    windows -- the planner's choice and each array form forced, 16 / 32 / 64 window lanes per read forced and planned,
    1 / 2 / many values per window, batches that end inside a tile, reads with as few as 1 and 5 windows; a batch with
    an N falls back to the kernels that know SeedNtHash's position state machine (App. B Q3)"""
    env = {"NTHIP_TUNE_SEED_PS": 1}
    force = {"planned": None, "terms": 0, "prefix": 1, "mixed": 2}[form]
    if force is not None:
        env["NTHIP_TUNE_SEED_PX_ARRAY"] = force + 1
    if lanes:
        env["NTHIP_TUNE_SEED_PS_LANES"] = lanes
    c = _ctx(**env)
    rng = np.random.default_rng(29)
    taken = 0
    for seeds, m2, L, n in PS_CASES:
        k = len(seeds[0])
        data = oracle.synth_reads(31, n, L, k + len(seeds))
        offs = np.arange(n + 1, dtype=np.uint64) * L
        want = oracle.seed_batch(data, offs, seeds, k, m2, want_pos=False)
        c.set_profiling(True)
        got = c.seed_hash(data, seeds, k, m2, fixed_len=L, n_reads=n)
        taken += c.last_kernel_ms()[1] == "seed_ps_kernel"
        c.set_profiling(False)
        bad = np.nonzero(got["hashes"].ravel() != want["hashes"].ravel())[0]
        assert bad.size == 0, (k, seeds, m2, L, n, bad[:8], bad.size, c.last_kernel_ms)
        dirty = data.copy()
        dirty[rng.choice(n * L, 3, replace=False)] = ord("N")
        want = oracle.seed_batch(dirty, offs, seeds, k, m2)
        got = c.seed_hash(dirty, seeds, k, m2, fixed_len=L, n_reads=n, want_pos=True)
        assert got["total"] == want["total"]
        for key in ("counts", "pos", "hashes"):
            assert (got[key] == want[key]).all(), (key, k, seeds)
    assert taken >= (len(PS_CASES) - 6 if lanes != 64 else 4), taken
    # device buffers that start off a 16-byte boundary
    import nthash_amd
    seeds, m2, L, n, k = C4, 3, 250, 4001, 31
    data = oracle.synth_reads(3, n, L, 8)
    want = oracle.seed_batch(data, np.arange(n + 1, dtype=np.uint64) * L, seeds, k, m2, want_pos=False)["hashes"].ravel()
    sd = nthash_amd.Seeds(c, seeds, k)
    for shift in (0, 5, 16, 31):
        d_in, d_out = c.malloc(n * L + 64), c.malloc(n * (L - k + 1) * 6 * 8)
        c.h2d(d_in + shift, data)
        tot = c.seed_hash_ptr(d_in + shift, 0, n, L, 0, sd, m2, d_out, n * (L - k + 1))
        got = np.zeros(tot * 6, np.uint64)
        c.d2h(got, d_out)
        assert (got == want).all(), shift
        c.free(d_in); c.free(d_out)
    sd.close()
    c.close()


@pytest.mark.parametrize("form", ["planned", "prefix", "mixed"])
def test_seed_jit_specialised_kernel_vs_oracle(oracle, form, tmp_path):
    """the kernel specialisation cache (capi_seed_jit.hip): seed_psj_kernel.inc compiled by hiprtc for
    the very seed set and read shape (NTHIP_SEED_JIT=1: whatever the batch size) -- the same stream as the oracle's
    seed_batch for every case the segment kernel takes; the second context finds the code objects in the disk cache"""
    force = {"planned": None, "prefix": 1, "mixed": 2}[form]
    env = {"NTHIP_TUNE_SEED_PS": 1, "NTHIP_SEED_JIT": 1, "NTHIP_JIT_CACHE": str(tmp_path), "NTHIP_JIT_VERBOSE": 1}
    if force is not None:
        env["NTHIP_TUNE_SEED_PX_ARRAY"] = force + 1
    for round_ in range(2 if form == "planned" else 1):
        for k_, v_ in env.items():
            os.environ[k_] = str(v_)
        try:
            import nthash_amd
            c = nthash_amd.Context(0)
            taken = 0
            for seeds, m2, L, n in PS_CASES:
                k = len(seeds[0])
                data = oracle.synth_reads(37, n, L, k + len(seeds))
                offs = np.arange(n + 1, dtype=np.uint64) * L
                want = oracle.seed_batch(data, offs, seeds, k, m2, want_pos=False)
                c.set_profiling(True)
                got = c.seed_hash(data, seeds, k, m2, fixed_len=L, n_reads=n)
                name = c.last_kernel_ms()[1]
                taken += name == "seed_psj_kernel"
                c.set_profiling(False)
                bad = np.nonzero(got["hashes"].ravel() != want["hashes"].ravel())[0]
                assert bad.size == 0, (name, k, seeds, m2, L, n, bad[:8], bad.size)
                dirty = data.copy()
                dirty[[7, n * L - 3]] = ord("N")
                want = oracle.seed_batch(dirty, offs, seeds, k, m2)
                got = c.seed_hash(dirty, seeds, k, m2, fixed_len=L, n_reads=n, want_pos=True)
                for key in ("counts", "pos", "hashes"):
                    assert (got[key] == want[key]).all(), (key, k, seeds)
            assert taken >= len(PS_CASES) - 6, taken
            # a batch of many tiles whose last one is short (the specialised kernel has tiles of its own)
            for n in (100003, 7):
                seeds, m2, L = [blocky(128, [(40, 48)])], 1, 250
                data = oracle.synth_reads(41, n, L, 3)
                want = oracle.seed_batch(data, np.arange(n + 1, dtype=np.uint64) * L, seeds, 128, m2, want_pos=False)
                got = c.seed_hash(data, seeds, 128, m2, fixed_len=L, n_reads=n)
                assert (got["hashes"] == want["hashes"]).all(), n
            c.close()
        finally:
            for k_ in env:
                os.environ.pop(k_, None)
        if form == "planned":   # (the other forms meet shapes the process has loaded already: nothing new on the disk)
            assert len(list(tmp_path.glob("psj_*.hsaco"))) >= 6


def test_seed_jit_compiles_in_the_background(oracle, tmp_path):
    """the default (NTHIP_SEED_JIT unset): the first batch of a seed set the specialised kernel would take starts its compile
    on a thread of its own and is hashed by the precompiled kernels; some batch later the code object is there and the
    specialised kernel takes over -- the stream is the oracle's before, while and after"""
    import time
    import nthash_amd
    os.environ["NTHIP_JIT_CACHE"] = str(tmp_path)
    os.environ.pop("NTHIP_SEED_JIT", None)
    try:
        c = nthash_amd.Context(0)
        rng = np.random.default_rng(77)
        seeds = []
        for _ in range(6):   # six seeds of 31 x 1 hash: a shape the specialised kernel wins
            half = rng.random(16) < 0.7
            s = np.concatenate([half, half[:15][::-1]])
            s[0] = s[-1] = True
            seeds.append("".join("1" if b else "0" for b in s))
        n, L, k = 20000, 250, 31
        data = oracle.synth_reads(5, n, L, 11)
        want = oracle.seed_batch(data, np.arange(n + 1, dtype=np.uint64) * L, seeds, k, 1, want_pos=False)["hashes"]
        c.set_profiling(True)
        names, t0 = [], time.time()
        while time.time() - t0 < 120:
            got = c.seed_hash(data, seeds, k, 1, fixed_len=L, n_reads=n)
            names.append(c.last_kernel_ms()[1])
            assert (got["hashes"] == want).all(), names[-1]
            if names[-1] == "seed_psj_kernel":
                break
            time.sleep(0.2)
        assert names[0] != "seed_psj_kernel" and names[-1] == "seed_psj_kernel", names
        assert len(list(tmp_path.glob("psj_*.hsaco"))) == 1
        c.close()
    finally:
        os.environ.pop("NTHIP_JIT_CACHE", None)


_EXIT_SCRIPT = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
import nthash_amd
c = nthash_amd.Context(0)
seeds = {seeds!r}
n, L, k = 20000, 250, 31
data = np.frombuffer(np.random.default_rng(3).choice(np.frombuffer(b"ACGT", dtype=np.uint8), n * L).tobytes(), dtype=np.uint8)
c.set_profiling(True)
c.seed_hash(data, seeds, k, 1, fixed_len=L, n_reads=n)
print("KERNEL", c.last_kernel_ms()[1])
"""


def test_seed_jit_process_that_ends_mid_compile_and_the_next_one(tmp_path):
    """a process that ends while its compile thread is inside the compiler waits for it at exit (no crash, the code object on
    the disk); the NEXT process finds the code object and hashes its FIRST batch with the specialised kernel"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(78)
    seeds = []
    for _ in range(6):
        half = rng.random(16) < 0.7
        s = np.concatenate([half, half[:15][::-1]])
        s[0] = s[-1] = True
        seeds.append("".join("1" if b else "0" for b in s))
    env = dict(os.environ)
    env["NTHIP_JIT_CACHE"] = str(tmp_path)
    env.pop("NTHIP_SEED_JIT", None)
    script = _EXIT_SCRIPT.format(root=root, seeds=seeds)
    first = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=300)
    assert first.returncode == 0, first.stderr[-2000:]
    assert "KERNEL" in first.stdout and "seed_psj_kernel" not in first.stdout, first.stdout
    assert len(list(tmp_path.glob("psj_*.hsaco"))) == 1
    second = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=300)
    assert second.returncode == 0, second.stderr[-2000:]
    assert "KERNEL seed_psj_kernel" in second.stdout, second.stdout


@pytest.mark.parametrize("cache", ["unwritable", "off"])
def test_seed_jit_without_a_disk_cache(oracle, cache):
    """a cache directory that cannot be made (or NTHIP_JIT_CACHE= : none wanted) costs the next process a compile, nothing else:
    the specialised kernel is compiled, loaded and hashes the oracle's stream"""
    import nthash_amd
    os.environ["NTHIP_JIT_CACHE"] = "/proc/nthash_amd_no_such_dir/cache" if cache == "unwritable" else ""
    os.environ["NTHIP_SEED_JIT"] = "1"
    os.environ["NTHIP_TUNE_SEED_PS"] = "1"
    try:
        c = nthash_amd.Context(0)
        seeds = [blocky(40, [(9, 6), (22, 9)] if cache == "unwritable" else [(5, 8), (19, 12)])]   # (shapes no other test compiles)
        n, L, k = 3000, 173, 40
        data = oracle.synth_reads(43, n, L, 9)
        want = oracle.seed_batch(data, np.arange(n + 1, dtype=np.uint64) * L, seeds, k, 2, want_pos=False)
        c.set_profiling(True)
        got = c.seed_hash(data, seeds, k, 2, fixed_len=L, n_reads=n)
        assert c.last_kernel_ms()[1] == "seed_psj_kernel"
        assert (got["hashes"] == want["hashes"]).all()
        c.close()
    finally:
        for k_ in ("NTHIP_JIT_CACHE", "NTHIP_SEED_JIT", "NTHIP_TUNE_SEED_PS"):
            os.environ.pop(k_, None)


def test_seed_jit_threads_and_contexts(oracle, tmp_path):
    """four threads, a context each, two seed sets between them, background compiles (the default): the registry hands every
    thread the same code object once it is there -- the streams are the oracle's before and after, in every thread"""
    import threading
    import time
    import nthash_amd
    os.environ["NTHIP_JIT_CACHE"] = str(tmp_path)
    os.environ.pop("NTHIP_SEED_JIT", None)
    rng = np.random.default_rng(79)
    sets = []
    for _ in range(2):
        seeds = []
        for _ in range(5):
            half = rng.random(16) < 0.7
            s = np.concatenate([half, half[:15][::-1]])
            s[0] = s[-1] = True
            seeds.append("".join("1" if b else "0" for b in s))
        sets.append(seeds)
    n, L, k = 20000, 250, 31
    data = oracle.synth_reads(6, n, L, 12)
    wants = [oracle.seed_batch(data, np.arange(n + 1, dtype=np.uint64) * L, s, k, 1, want_pos=False)["hashes"] for s in sets]
    errors, finals = [], [None] * 4

    def work(i):
        try:
            c = nthash_amd.Context(0)
            c.set_profiling(True)
            t0 = time.time()
            while time.time() - t0 < 120:
                got = c.seed_hash(data, sets[i % 2], k, 1, fixed_len=L, n_reads=n)
                name = c.last_kernel_ms()[1]
                if not (got["hashes"] == wants[i % 2]).all():
                    errors.append((i, name))
                    break
                finals[i] = name
                if name == "seed_psj_kernel":
                    break
                time.sleep(0.05)
            c.close()
        except Exception as e:  # noqa: BLE001 (reported below, in the test's thread)
            errors.append((i, repr(e)))

    try:
        threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        assert finals == ["seed_psj_kernel"] * 4, finals
        assert len(list(tmp_path.glob("psj_*.hsaco"))) == 2
    finally:
        os.environ.pop("NTHIP_JIT_CACHE", None)


def test_seed_jit_random_shapes():
    """tools/stress_seed_jit.py: 60 clean fixed-length batches of random read length, k, seed sets and hashes per seed through the
    specialised (or, where no code object can be made, the precompiled) segment kernel against the lane-per-read kernel"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_seed_jit.py"), "60", "17"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "0 mismatches" in r.stdout and "seed_psj_kernel" in r.stdout, r.stdout[-500:]
