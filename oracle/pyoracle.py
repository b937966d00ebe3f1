"""oracle/pyoracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes/numpy bindings for oracle/liboracle.so (the plain-C restatement) and, if
present, oracle/_ref/libnthash_ref.so (the real reference, built by
oracle/Makefile).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; nothing under nthash_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(_HERE, "liboracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libnthash_ref.so")

_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)


def build(force=False):
    """Compile the C restatement (and the reference, when its tree is here)."""
    if force or not os.path.exists(_ORACLE_SO) or (
        os.path.getmtime(_ORACLE_SO) < os.path.getmtime(os.path.join(_HERE, "nthash_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src") and (
        force or not os.path.exists(_REF_SO)
        or os.path.getmtime(_REF_SO) < os.path.getmtime(os.path.join(_HERE, "ref_shim.cpp"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    # the reference's own test program linked with OUR facade library (run on the GPU box by
    # tests/test_gpu_facade.py); rebuilt whenever the facade library is newer
    facade = os.path.join(os.path.dirname(_HERE), "nthash_amd", "lib", "libnthash.so")
    exe = os.path.join(_HERE, "_ref", "ref_tests_on_facade")
    if os.path.isfile("/root/reference/tests/tests.cpp") and os.path.exists(facade) and (
        force or not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(facade)
    ):
        subprocess.check_call(["make", "-C", _HERE, "ref_tests"], stdout=subprocess.DEVNULL)
    bexe = os.path.join(_HERE, "_ref", "ref_benchmark_on_facade")
    if os.path.isfile("/root/reference/examples/benchmark.cpp") and os.path.exists(facade) and (
        force or not os.path.exists(bexe) or os.path.getmtime(bexe) < os.path.getmtime(facade)
    ):
        subprocess.check_call(["make", "-C", _HERE, "ref_bench"], stdout=subprocess.DEVNULL)


def _ptr(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def _as_bytes(seqs):
    if isinstance(seqs, (bytes, bytearray)):
        return np.frombuffer(bytes(seqs), dtype=np.uint8)
    if isinstance(seqs, str):
        return np.frombuffer(seqs.encode("latin-1"), dtype=np.uint8)
    return np.ascontiguousarray(seqs, dtype=np.uint8)


def concat_reads(reads):
    """list of str/bytes -> (uint8 array, uint64 offsets[n+1])."""
    bs = [r.encode("latin-1") if isinstance(r, str) else bytes(r) for r in reads]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    data = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
    if data.size == 0:
        data = np.zeros(1, np.uint8)
    return data, offs


def _max_kmers(offs, k):
    lens = (offs[1:] - offs[:-1]).astype(np.int64)
    return int(np.maximum(lens - k + 1, 0).sum())


class _SeedArr:
    def __init__(self, seeds):
        self.bs = [s.encode("latin-1") if isinstance(s, str) else bytes(s) for s in seeds]
        self.arr = (C.c_char_p * len(self.bs))(*self.bs)
        self.n = len(self.bs)


class Oracle:
    """The C restatement (kind = "port")."""

    kind = "port"

    def __init__(self):
        build()
        L = C.CDLL(_ORACLE_SO)
        self.L = L
        for name in ("nto_srol", "nto_sror"):
            getattr(L, name).restype = C.c_uint64
            getattr(L, name).argtypes = [C.c_uint64]
        L.nto_srol_n.restype = C.c_uint64
        L.nto_srol_n.argtypes = [C.c_uint64, C.c_uint]
        L.nto_seed_fwd.restype = C.c_uint64
        L.nto_seed_fwd.argtypes = [C.c_ubyte]
        L.nto_seed_rc.restype = C.c_uint64
        L.nto_seed_rc.argtypes = [C.c_ubyte]
        L.nto_base_fwd.restype = C.c_uint64
        L.nto_base_fwd.argtypes = [C.c_char_p, C.c_uint]
        L.nto_base_rev.restype = C.c_uint64
        L.nto_base_rev.argtypes = [C.c_char_p, C.c_uint]
        L.nto_kmer_batch.restype = C.c_uint64
        L.nto_kmer_batch.argtypes = [C.c_void_p, _u64p, C.c_uint64, C.c_uint, C.c_uint,
                                     _u64p, _u32p, _u64p, _u64p, _u64p]
        L.nto_seed_batch.restype = C.c_uint64
        L.nto_seed_batch.argtypes = [C.c_void_p, _u64p, C.c_uint64, C.POINTER(C.c_char_p),
                                     C.c_uint, C.c_uint, C.c_uint, _u64p, _u32p, _u64p]
        L.nto_get_blocks.restype = C.c_uint
        L.nto_get_blocks.argtypes = [C.c_char_p, C.c_uint, _u32p, _u32p, _u32p]
        L.nto_seed_window.restype = None
        L.nto_seed_window.argtypes = [C.c_char_p, C.c_char_p, C.c_uint, _u64p, _u64p]
        L.nto_seed_extend.restype = None
        L.nto_seed_extend.argtypes = [C.c_char_p, C.c_uint, C.POINTER(C.c_char_p), C.c_uint, C.c_uint, _u64p, _u64p, _u64p]
        L.nto_splitmix64.restype = C.c_uint64
        L.nto_splitmix64.argtypes = [C.c_uint64]
        L.nto_synth_reads.restype = None
        L.nto_synth_reads.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint, C.c_uint64]
        L.nto_checksum.restype = None
        L.nto_checksum.argtypes = [_u64p, C.c_uint64, _u64p, _u64p]
        L.nto_bench_kmer.restype = C.c_uint64
        L.nto_bench_kmer.argtypes = [C.c_void_p, C.c_uint64, C.c_uint, C.c_uint, C.c_uint, _u64p]
        L.nto_extend.restype = None
        L.nto_extend.argtypes = [C.c_uint64, C.c_uint64, C.c_uint, C.c_uint, _u64p]
        # iterator emulation
        L.nto_nthash_init.restype = C.c_int
        for n in ("roll", "roll_back", "peek", "peek_back"):
            getattr(L, "nto_nthash_" + n).restype = C.c_int
            getattr(L, "nto_nthash_" + n).argtypes = [C.c_void_p]
        for n in ("peek_char", "peek_back_char"):
            getattr(L, "nto_nthash_" + n).restype = C.c_int
            getattr(L, "nto_nthash_" + n).argtypes = [C.c_void_p, C.c_char]

    # -- primitives -------------------------------------------------------
    def srol(self, x):
        return self.L.nto_srol(x)

    def sror(self, x):
        return self.L.nto_sror(x)

    def srol_n(self, x, d):
        return self.L.nto_srol_n(x, d)

    def extend(self, fwd, rev, k, m):
        out = np.zeros(m, np.uint64)
        self.L.nto_extend(fwd, rev, k, m, _ptr(out, _u64p))
        return out

    # -- batches ----------------------------------------------------------
    def kmer_batch(self, data, offs, k, m, want_pos=True, want_strands=False):
        data = _as_bytes(data)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = len(offs) - 1
        cap = max(_max_kmers(offs, k), 1)
        hashes = np.zeros(cap * m, np.uint64)
        pos = np.zeros(cap, np.uint32) if want_pos else None
        fwd = np.zeros(cap, np.uint64) if want_strands else None
        rev = np.zeros(cap, np.uint64) if want_strands else None
        counts = np.zeros(max(n, 1), np.uint64)
        tot = self._kmer_batch(data, offs, n, k, m, hashes, pos, fwd, rev, counts)
        out = {"total": tot, "hashes": hashes[: tot * m].reshape(tot, m),
               "counts": counts[:n]}
        if want_pos:
            out["pos"] = pos[:tot]
        if want_strands:
            out["fwd"] = fwd[:tot]
            out["rev"] = rev[:tot]
        return out

    def _kmer_batch(self, data, offs, n, k, m, hashes, pos, fwd, rev, counts):
        return self.L.nto_kmer_batch(data.ctypes.data, _ptr(offs, _u64p), n, k, m,
                                     _ptr(hashes, _u64p), _ptr(pos, _u32p), _ptr(fwd, _u64p),
                                     _ptr(rev, _u64p), _ptr(counts, _u64p))

    def seed_batch(self, data, offs, seeds, k, m2, want_pos=True):
        data = _as_bytes(data)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = len(offs) - 1
        sa = _SeedArr(seeds)
        per = sa.n * m2
        cap = max(_max_kmers(offs, k), 1)
        hashes = np.zeros(cap * per, np.uint64)
        pos = np.zeros(cap, np.uint32) if want_pos else None
        counts = np.zeros(max(n, 1), np.uint64)
        tot = self._seed_batch(data, offs, n, sa, k, m2, hashes, pos, counts)
        out = {"total": tot, "hashes": hashes[: tot * per].reshape(tot, per),
               "counts": counts[:n]}
        if want_pos:
            out["pos"] = pos[:tot]
        return out

    def _seed_batch(self, data, offs, n, sa, k, m2, hashes, pos, counts):
        return self.L.nto_seed_batch(data.ctypes.data, _ptr(offs, _u64p), n, sa.arr, sa.n, k, m2,
                                     _ptr(hashes, _u64p), _ptr(pos, _u32p), _ptr(counts, _u64p))

    def seed_extend(self, kmer, seeds, m2):
        """BlindSeedNtHash::roll(c) / roll_back(c), c = A, C, G, T, from the window `kmer` (src/seed.cpp:701-737):
        -> (self [n_seeds * m2], next [4][n_seeds * m2], prev [4][n_seeds * m2])"""
        kb = kmer.encode("latin-1") if isinstance(kmer, str) else bytes(kmer)
        sa = _SeedArr(seeds)
        per = sa.n * m2
        me, nx, pv = np.zeros(per, np.uint64), np.zeros(4 * per, np.uint64), np.zeros(4 * per, np.uint64)
        self.L.nto_seed_extend(kb, len(kb), sa.arr, sa.n, m2, _ptr(me, _u64p), _ptr(nx, _u64p), _ptr(pv, _u64p))
        return me, nx.reshape(4, per), pv.reshape(4, per)

    def get_blocks(self, seed):
        k = len(seed)
        blocks = np.zeros(2 * (k + 2), np.uint32)
        monos = np.zeros(k + 2, np.uint32)
        nm = C.c_uint32(0)
        nb = self.L.nto_get_blocks(seed.encode(), k, _ptr(blocks, _u32p), _ptr(monos, _u32p),
                                   C.byref(nm))
        return [tuple(blocks[2 * i: 2 * i + 2].tolist()) for i in range(nb)], monos[: nm.value].tolist()

    # -- synthetic reads / checksums -------------------------------------
    def synth_reads(self, first_read, n_reads, length, seed=42):
        buf = np.zeros(n_reads * length, np.uint8)
        self.L.nto_synth_reads(buf.ctypes.data, first_read, n_reads, length, seed)
        return buf

    def checksum(self, v):
        v = np.ascontiguousarray(v, dtype=np.uint64).ravel()
        s, x = C.c_uint64(0), C.c_uint64(0)
        self.L.nto_checksum(_ptr(v, _u64p), v.size, C.byref(s), C.byref(x))
        return s.value, x.value

    def bench_kmer(self, data, n_reads, length, k, m, threads=1):
        data = _as_bytes(data)
        nk = C.c_uint64(0)
        acc = self.L.nto_bench_kmer(data.ctypes.data, n_reads, length, k, m, C.byref(nk))
        return acc, nk.value

    # -- iterator script (same op language as the reference shim) ----------
    def nthash_script(self, seq, m, k, pos0, ops):
        class IT(C.Structure):
            _fields_ = [("seq", C.c_char_p), ("len", C.c_size_t), ("k", C.c_uint), ("m", C.c_uint),
                        ("pos", C.c_size_t), ("initialized", C.c_int), ("fwd", C.c_uint64),
                        ("rev", C.c_uint64), ("h", _u64p)]
        sb = seq.encode("latin-1") if isinstance(seq, str) else bytes(seq)
        hbuf = np.zeros(max(m, 1), np.uint64)
        it = IT()
        self.L.nto_nthash_init.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint, C.c_uint,
                                           C.c_size_t, _u64p]
        if self.L.nto_nthash_init(C.byref(it), sb, len(sb), m, k, pos0, _ptr(hbuf, _u64p)) != 0:
            raise ValueError("reference would raise_error")
        res = []
        i = 0
        ob = ops.encode("latin-1") if isinstance(ops, str) else bytes(ops)
        while i < len(ob):
            o = chr(ob[i])
            if o == "r":
                ok = self.L.nto_nthash_roll(C.byref(it))
            elif o == "b":
                ok = self.L.nto_nthash_roll_back(C.byref(it))
            elif o == "p":
                ok = self.L.nto_nthash_peek(C.byref(it))
            elif o == "q":
                ok = self.L.nto_nthash_peek_back(C.byref(it))
            elif o == "P":
                i += 1
                ok = self.L.nto_nthash_peek_char(C.byref(it), ob[i:i + 1])
            elif o == "Q":
                i += 1
                ok = self.L.nto_nthash_peek_back_char(C.byref(it), ob[i:i + 1])
            else:
                raise ValueError(o)
            res.append((int(ok), int(it.pos), int(it.fwd), int(it.rev), hbuf.copy()))
            i += 1
        return res


class Reference(Oracle):
    """The REAL reference, through oracle/_ref/libnthash_ref.so (kind = "reference")."""

    kind = "reference"

    @staticmethod
    def available():
        build()
        return os.path.exists(_REF_SO)

    def __init__(self, so_path=None):
        """so_path: another library exporting the ref_* driver symbols (tests build
        oracle/ref_shim.cpp against nthash_amd's own header to drive its C++ facade)"""
        super().__init__()
        so_path = so_path or _REF_SO
        if not os.path.exists(so_path):
            raise FileNotFoundError(so_path)
        R = C.CDLL(so_path)
        self.R = R
        R.ref_fn_name.restype = C.c_char_p
        R.ref_kmer_batch.restype = C.c_uint64
        R.ref_kmer_batch.argtypes = self.L.nto_kmer_batch.argtypes
        R.ref_seed_batch.restype = C.c_uint64
        R.ref_seed_batch.argtypes = self.L.nto_seed_batch.argtypes
        R.ref_nthash_script.restype = C.c_uint64
        R.ref_nthash_script.argtypes = [C.c_char_p, C.c_uint64, C.c_uint, C.c_uint, C.c_uint64,
                                        C.c_char_p, C.c_uint64, _i32p, _u64p, _u64p, _u64p, _u64p]
        R.ref_blind_script.restype = C.c_uint64
        R.ref_blind_script.argtypes = [C.c_char_p, C.c_uint, C.c_uint, C.c_int64, C.c_char_p,
                                       C.c_uint64, _i64p, _u64p, _u64p, _u64p]
        R.ref_seed_script.restype = C.c_uint64
        R.ref_seed_script.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(C.c_char_p), C.c_uint,
                                      C.c_uint, C.c_uint, C.c_uint64, C.c_char_p, C.c_uint64,
                                      _i32p, _u64p, _u64p, _u64p, _u64p]
        R.ref_blindseed_script.restype = C.c_uint64
        R.ref_blindseed_script.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_uint, C.c_uint,
                                           C.c_uint, C.c_int64, C.c_char_p, C.c_uint64, _i64p,
                                           _u64p, _u64p, _u64p]
        R.ref_parse_seeds.restype = C.c_uint64
        R.ref_parse_seeds.argtypes = [C.c_char_p, _u32p, C.c_uint64]
        R.ref_bench_kmer.restype = C.c_uint64
        R.ref_bench_kmer.argtypes = [C.c_void_p, C.c_uint64, C.c_uint, C.c_uint, C.c_uint, C.c_int,
                                     _u64p]
        R.ref_bench_seed.restype = C.c_uint64
        R.ref_bench_seed.argtypes = [C.c_void_p, C.c_uint64, C.c_uint, C.POINTER(C.c_char_p),
                                     C.c_uint, C.c_uint, C.c_uint, C.c_int, _u64p]
        R.ref_max_threads.restype = C.c_int
        # older prebuilt _ref libraries lack the synthetic-workload helpers
        self.has_synth = hasattr(R, "ref_synth_checksum") and hasattr(R, "ref_bench_synth")
        if self.has_synth:
            if hasattr(R, "ref_synth_var_checksum"):
                R.ref_synth_var_checksum.restype = None
                R.ref_synth_var_checksum.argtypes = [C.c_uint64, C.c_uint64, C.c_uint, C.c_uint, C.c_uint64, C.c_uint, C.c_uint,
                                                     C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                                     C.POINTER(C.c_uint64)]
            R.ref_synth_checksum.restype = None
            R.ref_synth_checksum.argtypes = [C.c_uint64, C.c_uint64, C.c_uint, C.c_uint64, C.POINTER(C.c_char_p),
                                             C.c_uint, C.c_uint, C.c_uint, C.c_int, _u64p, _u64p, _u64p]
            R.ref_bench_synth.restype = C.c_double
            R.ref_bench_synth.argtypes = [C.c_uint64, C.c_uint64, C.c_uint, C.c_uint64, C.POINTER(C.c_char_p),
                                          C.c_uint, C.c_uint, C.c_uint, C.c_int, C.c_int, _u64p, _u64p,
                                          C.POINTER(C.c_int)]

    def fn_name(self):
        return self.R.ref_fn_name().decode()

    def _kmer_batch(self, data, offs, n, k, m, hashes, pos, fwd, rev, counts):
        return self.R.ref_kmer_batch(data.ctypes.data, _ptr(offs, _u64p), n, k, m,
                                     _ptr(hashes, _u64p), _ptr(pos, _u32p), _ptr(fwd, _u64p),
                                     _ptr(rev, _u64p), _ptr(counts, _u64p))

    def _seed_batch(self, data, offs, n, sa, k, m2, hashes, pos, counts):
        return self.R.ref_seed_batch(data.ctypes.data, _ptr(offs, _u64p), n, sa.arr, sa.n, k, m2,
                                     _ptr(hashes, _u64p), _ptr(pos, _u32p), _ptr(counts, _u64p))

    @staticmethod
    def _count_ops(ob, two_byte):
        n, i = 0, 0
        while i < len(ob):
            i += 2 if chr(ob[i]) in two_byte else 1
            n += 1
        return n

    def nthash_script(self, seq, m, k, pos0, ops):
        sb = seq.encode("latin-1") if isinstance(seq, str) else bytes(seq)
        if k == 0 or len(sb) < k or pos0 > len(sb) - k:
            raise ValueError("reference would raise_error")
        ob = ops.encode("latin-1") if isinstance(ops, str) else bytes(ops)
        n = self._count_ops(ob, "PQ")
        ret = np.zeros(n, np.int32)
        pos = np.zeros(n, np.uint64)
        fwd = np.zeros(n, np.uint64)
        rev = np.zeros(n, np.uint64)
        hs = np.zeros(n * m, np.uint64)
        got = self.R.ref_nthash_script(sb, len(sb), m, k, pos0, ob, len(ob), _ptr(ret, _i32p),
                                       _ptr(pos, _u64p), _ptr(fwd, _u64p), _ptr(rev, _u64p),
                                       _ptr(hs, _u64p))
        assert got == n
        return [(int(ret[i]), int(pos[i]), int(fwd[i]), int(rev[i]), hs[i * m:(i + 1) * m].copy())
                for i in range(n)]

    def blind_script(self, seq, m, k, pos0, ops):
        sb = seq.encode("latin-1") if isinstance(seq, str) else bytes(seq)
        ob = ops.encode("latin-1") if isinstance(ops, str) else bytes(ops)
        n = len(ob) // 2 + 1
        pos = np.zeros(n, np.int64)
        fwd = np.zeros(n, np.uint64)
        rev = np.zeros(n, np.uint64)
        hs = np.zeros(n * m, np.uint64)
        got = self.R.ref_blind_script(sb, m, k, pos0, ob, len(ob), _ptr(pos, _i64p),
                                      _ptr(fwd, _u64p), _ptr(rev, _u64p), _ptr(hs, _u64p))
        assert got == n
        return [(int(pos[i]), int(fwd[i]), int(rev[i]), hs[i * m:(i + 1) * m].copy())
                for i in range(n)]

    def seed_script(self, seq, seeds, m2, k, pos0, ops):
        sb = seq.encode("latin-1") if isinstance(seq, str) else bytes(seq)
        ob = ops.encode("latin-1") if isinstance(ops, str) else bytes(ops)
        sa = _SeedArr(seeds)
        per = sa.n * m2
        n = self._count_ops(ob, "PQ")
        ret = np.zeros(n, np.int32)
        pos = np.zeros(n, np.uint64)
        fwd = np.zeros(n * sa.n, np.uint64)
        rev = np.zeros(n * sa.n, np.uint64)
        hs = np.zeros(n * per, np.uint64)
        got = self.R.ref_seed_script(sb, len(sb), sa.arr, sa.n, m2, k, pos0, ob, len(ob),
                                     _ptr(ret, _i32p), _ptr(pos, _u64p), _ptr(fwd, _u64p),
                                     _ptr(rev, _u64p), _ptr(hs, _u64p))
        assert got == n
        return [(int(ret[i]), int(pos[i]), fwd[i * sa.n:(i + 1) * sa.n].copy(),
                 rev[i * sa.n:(i + 1) * sa.n].copy(), hs[i * per:(i + 1) * per].copy())
                for i in range(n)]

    def blindseed_script(self, seq, seeds, m2, k, pos0, ops):
        sb = seq.encode("latin-1") if isinstance(seq, str) else bytes(seq)
        ob = ops.encode("latin-1") if isinstance(ops, str) else bytes(ops)
        sa = _SeedArr(seeds)
        per = sa.n * m2
        n = len(ob) // 2 + 1
        pos = np.zeros(n, np.int64)
        fwd = np.zeros(n * sa.n, np.uint64)
        rev = np.zeros(n * sa.n, np.uint64)
        hs = np.zeros(n * per, np.uint64)
        got = self.R.ref_blindseed_script(sb, sa.arr, sa.n, m2, k, pos0, ob, len(ob),
                                          _ptr(pos, _i64p), _ptr(fwd, _u64p), _ptr(rev, _u64p),
                                          _ptr(hs, _u64p))
        assert got == n
        return [(int(pos[i]), fwd[i * sa.n:(i + 1) * sa.n].copy(),
                 rev[i * sa.n:(i + 1) * sa.n].copy(), hs[i * per:(i + 1) * per].copy())
                for i in range(n)]

    def parse_seeds(self, seed):
        out = np.zeros(len(seed) + 1, np.uint32)
        n = self.R.ref_parse_seeds(seed.encode(), _ptr(out, _u32p), out.size)
        return out[:n].tolist()

    def bench_kmer(self, data, n_reads, length, k, m, threads=1):
        data = _as_bytes(data)
        nk = C.c_uint64(0)
        acc = self.R.ref_bench_kmer(data.ctypes.data, n_reads, length, k, m, threads, C.byref(nk))
        return acc, nk.value

    def bench_seed(self, data, n_reads, length, seeds, k, m2, threads=1):
        data = _as_bytes(data)
        sa = _SeedArr(seeds)
        nk = C.c_uint64(0)
        acc = self.R.ref_bench_seed(data.ctypes.data, n_reads, length, sa.arr, sa.n, k, m2, threads,
                                    C.byref(nk))
        return acc, nk.value

    def max_threads(self):
        return self.R.ref_max_threads()

    def synth_checksum(self, first_read, n_reads, length, k, m, seeds=None, seed=42, threads=0):
        """(sum, xor, total) over EVERY hash of the synthetic reads [first_read, first_read + n_reads):
        NtHash(k, m), or SeedNtHash(seeds, m per seed).  OpenMP over reads, no read buffer."""
        sa = _SeedArr(seeds or [])
        s_, x_, t_ = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self.R.ref_synth_checksum(first_read, n_reads, length, seed, sa.arr if sa.n else None, sa.n, k, m, threads,
                                  C.byref(s_), C.byref(x_), C.byref(t_))
        return s_.value, x_.value, t_.value

    def synth_var_checksum(self, first_read, n_reads, len_min, len_max, k, m, seed=42, threads=0):
        """(sum, xor, total) over every NtHash(k, m) hash of the variable-length synthetic reads (ref_shim.cpp:
        ref_synth_var_checksum states the rule; var_reads() below is its numpy twin)"""
        s_, x_, t_ = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self.R.ref_synth_var_checksum(C.c_uint64(first_read), C.c_uint64(n_reads), len_min, len_max, C.c_uint64(seed),
                                      k, m, threads, C.byref(s_), C.byref(x_), C.byref(t_))
        return s_.value, x_.value, t_.value

    def bench_synth(self, first_read, n_reads, length, k, m, seeds=None, seed=42, threads=1, repeats=1):
        """Seconds the reference needs for the reads (timed inside the library: warm thread pool, reads first
        touched by the thread that hashes them); returns (seconds, k-mers, threads used)."""
        sa = _SeedArr(seeds or [])
        nk, acc, used = C.c_uint64(0), C.c_uint64(0), C.c_int(0)
        sec = self.R.ref_bench_synth(first_read, n_reads, length, seed, sa.arr if sa.n else None, sa.n, k, m, threads,
                                     repeats, C.byref(nk), C.byref(acc), C.byref(used))
        if sec < 0:
            raise MemoryError("ref_bench_synth: allocation failed")
        return sec, nk.value, used.value


def _splitmix64(x):
    """numpy uint64 splitmix64 (wrapping arithmetic), the generator of the synthetic workloads"""
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def var_reads(first_read, n_reads, len_min, len_max, seed=42):
    """The variable-length synthetic workload's shape: (lengths, has_n, n_pos) of reads [first_read, +n_reads) --
    read r = the first lengths[r] bytes of synthetic read r of length len_max, byte n_pos[r] replaced by 'N' where
    has_n[r] (oracle/ref_shim.cpp: ref_synth_var_checksum)."""
    with np.errstate(over="ignore"):
        r = np.arange(first_read, first_read + n_reads, dtype=np.uint64)
        x = _splitmix64(np.uint64(seed) + np.uint64(0xABCDEF) + r)
    lens = np.uint64(len_min) + x % np.uint64(len_max - len_min + 1)
    has_n = (x >> np.uint64(32)) % np.uint64(997) == 0
    n_pos = (x >> np.uint64(16)) % lens
    return lens, has_n, n_pos
