"""ctypes binding of the nthash_amd C-ABI (include/nthash_hip.h).

This is plumbing for the tests and bench.py: it loads nthash_amd/lib/libnthash_hip.so,
declares every symbol of the header, and offers thin numpy/torch-pointer
helpers.  It contains no hashing logic and no CPU fallback: if the shared
library is missing it raises, and if there is no HIP device every call that
touches the device returns NTHIP_ERR_NODEVICE, which is raised as NtHipError.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# NTHASH_AMD_LIB: A/B builds of the same C-ABI (tools/ab_build.sh); default is the in-tree library
LIB_PATH = os.environ.get("NTHASH_AMD_LIB") or os.path.join(_HERE, "lib", "libnthash_hip.so")

NTHIP_OK = 0
NTHIP_ERR_ARG = -1
NTHIP_ERR_HIP = -2
NTHIP_ERR_NODEVICE = -3
NTHIP_ERR_CAPACITY = -4
NTHIP_ERR_UNSUPPORTED = -5
NTHIP_HOST_INPUT = 0x1
NTHIP_HOST_OUTPUT = 0x2
NTHIP_ASYNC = 0x10
NTHIP_PACKED_INPUT = 0x20
NTHIP_PACKED_CLEAN = 0x40
NTHIP_OUT_READ_SLOTS = 0x80
NTHIP_FORCE_GENERAL = 0x4
NTHIP_FORCE_ROWS = 0x8

# every exported symbol of include/nthash_hip.h (kept in sync by tests/test_abi.py)
SYMBOLS = [
    "nthip_version", "nthip_last_error", "nthip_device_count", "nthip_ctx_create",
    "nthip_ctx_destroy", "nthip_ctx_set_stream", "nthip_ctx_synchronize", "nthip_ctx_take_dirty",
    "nthip_ctx_set_profiling", "nthip_last_kernel_ms", "nthip_malloc", "nthip_free",
    "nthip_ctx_trim", "nthip_ctx_set_scratch_limit", "nthip_ctx_scratch_info", "nthip_memcpy_h2d", "nthip_memcpy_d2h", "nthip_memset", "nthip_kmer_hash", "nthip_seeds_create",
    "nthip_seeds_destroy", "nthip_seed_jit_source", "nthip_seed_hash", "nthip_kmer_extend", "nthip_kmer_bloom_insert",
    "nthip_kmer_bloom_query", "nthip_kmer_minhash", "nthip_stream_bloom_insert", "nthip_kmer_hash_spans", "nthip_fastx_index",
    "nthip_fastx_kmer_hash_file", "nthip_fastx_seed_hash_file", "nthip_seed_hash_spans", "nthip_fasta_compact", "nthip_synth_reads", "nthip_checksum",
    "nthip_copy_bench", "nthip_fill_bench", "nthip_malloc_probed", "nthip_ctx_reload_tuning",
    "nthip_multi_create", "nthip_multi_destroy", "nthip_multi_device_count", "nthip_multi_kmer_hash",
    "nthip_multi_seeds_create", "nthip_multi_seeds_destroy", "nthip_multi_seed_hash",
    "nthip_packed_size", "nthip_pack_reads", "nthip_multi_fastx_kmer_hash_file", "nthip_host_alloc", "nthip_host_free",
    "nthip_kmer_count_insert", "nthip_stream_count_insert", "nthip_stream_count_query", "nthip_kmer_minimizers",
    "nthip_stream_bloom_query", "nthip_kmer_minimizers_spans",
    "nthip_multi_ctx", "nthip_multi_kmer_hash_shards", "nthip_multi_kmer_bloom_insert", "nthip_multi_kmer_count_insert",
    "nthip_multi_kmer_minhash_set", "nthip_multi_merge", "nthip_seed_extend", "nthip_kmer_count_query", "nthip_seed_bloom_insert", "nthip_seed_bloom_query",
    "nthip_multi_kmer_bloom_query", "nthip_multi_kmer_count_query",
]
NTHIP_MULTI_ALLGATHER = 0x100
NTHIP_MERGE_OR, NTHIP_MERGE_ADD_SAT_U8, NTHIP_MERGE_MIN_U64 = 0, 1, 2


class NtHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"nthip error {code}: {msg}")
        self.code = code


class Reads(C.Structure):
    _fields_ = [("seqs", C.c_void_p), ("offsets", C.c_void_p), ("n_reads", C.c_uint64),
                ("fixed_len", C.c_uint32), ("stride", C.c_uint32)]


class Out(C.Structure):
    _fields_ = [("hashes", C.c_void_p), ("capacity", C.c_uint64), ("counts", C.c_void_p),
                ("pos", C.c_void_p), ("fwd", C.c_void_p), ("rev", C.c_void_p)]


class FastxBatch(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_kmers", C.c_uint64), ("hashes", C.c_void_p), ("counts", C.c_void_p),
                ("raw", C.c_void_p), ("starts", C.c_void_p), ("ends", C.c_void_p), ("first_read", C.c_uint64),
                ("device", C.c_int32), ("reserved", C.c_uint32)]


class FastxStats(C.Structure):
    _fields_ = [("file_bytes", C.c_uint64), ("reads", C.c_uint64), ("kmers", C.c_uint64), ("batches", C.c_uint64),
                ("seconds", C.c_double), ("read_seconds", C.c_double), ("gpu_seconds", C.c_double)]


FASTX_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(FastxBatch))
NTHIP_FASTQ, NTHIP_FASTA, NTHIP_FASTA_MULTILINE = 4, 2, 1

_lib = None


def load():
    """dlopen the C-ABI library; loud failure if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m nthash_amd.build` "
            "(nthash_amd has no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
    L.nthip_version.restype = C.c_char_p
    L.nthip_last_error.restype = C.c_char_p
    L.nthip_device_count.argtypes = [C.POINTER(C.c_int)]
    L.nthip_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.nthip_ctx_destroy.argtypes = [vp]
    L.nthip_ctx_trim.argtypes = [vp]
    L.nthip_ctx_set_scratch_limit.argtypes = [vp, C.c_size_t]
    L.nthip_ctx_scratch_info.argtypes = [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.nthip_ctx_set_stream.argtypes = [vp, vp]
    L.nthip_ctx_synchronize.argtypes = [vp]
    L.nthip_ctx_take_dirty.argtypes = [vp, C.POINTER(C.c_int)]
    L.nthip_ctx_set_profiling.argtypes = [vp, C.c_int]
    L.nthip_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_char_p)]
    L.nthip_malloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.nthip_free.argtypes = [vp, vp]
    L.nthip_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    L.nthip_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    L.nthip_memset.argtypes = [vp, vp, C.c_int, C.c_size_t]
    L.nthip_kmer_hash.argtypes = [vp, C.POINTER(Reads), C.c_uint16, C.c_uint8, C.POINTER(Out),
                                  C.POINTER(u64), u32]
    L.nthip_seeds_create.argtypes = [vp, C.POINTER(C.c_char_p), u32, C.c_uint16, C.POINTER(vp),
                                     C.POINTER(C.c_int)]
    L.nthip_seeds_destroy.argtypes = [vp]
    L.nthip_seed_jit_source.argtypes = [C.POINTER(C.c_char_p), u32, C.c_uint16, u32, C.c_uint8, C.POINTER(vp)]
    L.nthip_seed_hash.argtypes = [vp, C.POINTER(Reads), vp, C.c_uint8, C.POINTER(Out),
                                  C.POINTER(u64), u32]
    L.nthip_kmer_extend.argtypes = [vp, vp, u64, C.c_uint16, C.c_uint8, vp, vp, vp, u32]
    L.nthip_seed_extend.argtypes = [vp, vp, u64, vp, C.c_uint8, vp, vp, vp, u32]
    L.nthip_kmer_bloom_insert.argtypes = [vp, C.POINTER(Reads), C.c_uint16, C.c_uint8, vp, u64, C.POINTER(u64), u32]
    L.nthip_kmer_bloom_query.argtypes = [vp, C.POINTER(Reads), C.c_uint16, C.c_uint8, vp, u64, vp,
                                         C.POINTER(u64), C.POINTER(u64), u32]
    L.nthip_kmer_minhash.argtypes = [vp, C.POINTER(Reads), C.c_uint16, C.c_uint8, vp, C.POINTER(u64), u32]
    L.nthip_stream_bloom_insert.argtypes = [vp, vp, u64, vp, u64]
    L.nthip_kmer_count_insert.argtypes = [vp, C.POINTER(Reads), C.c_uint16, C.c_uint8, vp, u64, C.POINTER(u64), u32]
    L.nthip_stream_count_insert.argtypes = [vp, vp, u64, vp, u64]
    L.nthip_stream_bloom_query.argtypes = [vp, vp, u64, C.c_uint8, vp, u64, vp, C.POINTER(u64)]
    L.nthip_stream_count_query.argtypes = [vp, vp, u64, C.c_uint8, vp, u64, vp]
    L.nthip_seed_bloom_insert.argtypes = [vp, C.POINTER(Reads), vp, C.c_uint8, vp, u64, C.POINTER(u64), u32]
    L.nthip_seed_bloom_query.argtypes = [vp, C.POINTER(Reads), vp, C.c_uint8, vp, u64, vp, C.POINTER(u64), C.POINTER(u64), u32]
    L.nthip_kmer_count_query.argtypes = [vp, C.POINTER(Reads), C.c_uint16, C.c_uint8, vp, u64, vp, C.POINTER(u64), u32]
    L.nthip_kmer_minimizers.argtypes = [vp, C.POINTER(Reads), C.c_uint16, u32, vp, vp, vp, u64, C.POINTER(u64), u32]
    L.nthip_kmer_minimizers_spans.argtypes = [vp, vp, u64, vp, vp, u64, C.c_uint16, u32, vp, vp, vp, u64, C.POINTER(u64)]
    L.nthip_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.nthip_host_free.argtypes = [vp]
    L.nthip_kmer_hash_spans.argtypes = [vp, vp, u64, vp, vp, u64, C.c_uint16, C.c_uint8, C.POINTER(Out),
                                        C.POINTER(u64), u32]
    L.nthip_fastx_index.argtypes = [vp, vp, u64, u32, vp, vp, u64, C.POINTER(u64), C.POINTER(u64),
                                    C.POINTER(C.c_int)]
    L.nthip_seed_hash_spans.argtypes = [vp, vp, u64, vp, vp, u64, vp, C.c_uint8, C.POINTER(Out), C.POINTER(u64), u32]
    L.nthip_fastx_seed_hash_file.argtypes = [vp, C.c_char_p, u32, vp, C.c_uint8, u64, FASTX_FN, vp,
                                             C.POINTER(FastxStats)]
    L.nthip_fasta_compact.argtypes = [vp, vp, u64, vp, vp, u64, C.POINTER(u64), C.POINTER(u64)]
    L.nthip_fastx_kmer_hash_file.argtypes = [vp, C.c_char_p, u32, C.c_uint16, C.c_uint8, u64, FASTX_FN, vp,
                                             C.POINTER(FastxStats)]
    L.nthip_synth_reads.argtypes = [vp, vp, u64, u64, u32, u64]
    L.nthip_checksum.argtypes = [vp, vp, u64, C.POINTER(u64), C.POINTER(u64)]
    L.nthip_copy_bench.argtypes = [vp, vp, vp, C.c_size_t, C.c_int, C.POINTER(C.c_float)]
    L.nthip_fill_bench.argtypes = [vp, vp, C.c_size_t, C.c_int, C.POINTER(C.c_float)]
    L.nthip_ctx_reload_tuning.argtypes = [vp]
    L.nthip_malloc_probed.argtypes = [vp, C.c_size_t, C.c_int, C.POINTER(vp), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.nthip_multi_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.nthip_multi_destroy.argtypes = [vp]
    L.nthip_multi_device_count.argtypes = [vp, C.POINTER(C.c_int)]
    L.nthip_multi_kmer_hash.argtypes = [vp, vp, C.c_uint16, C.c_uint8, vp, C.POINTER(u64)]
    L.nthip_multi_seeds_create.argtypes = [vp, C.POINTER(C.c_char_p), u32, C.c_uint16, C.POINTER(vp), C.POINTER(C.c_int)]
    L.nthip_multi_seeds_destroy.argtypes = [vp]
    L.nthip_multi_seed_hash.argtypes = [vp, vp, vp, C.c_uint8, vp, C.POINTER(u64)]
    L.nthip_multi_ctx.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.nthip_multi_kmer_hash_shards.argtypes = [vp, vp, C.c_uint16, C.c_uint8, vp, C.POINTER(u64), u32]
    L.nthip_multi_kmer_bloom_insert.argtypes = [vp, vp, C.c_uint16, C.c_uint8, C.POINTER(vp), u64, C.POINTER(u64), u32]
    L.nthip_multi_kmer_count_insert.argtypes = [vp, vp, C.c_uint16, C.c_uint8, C.POINTER(vp), u64, C.POINTER(u64), u32]
    L.nthip_multi_kmer_minhash_set.argtypes = [vp, vp, C.c_uint16, C.c_uint8, C.POINTER(vp), C.POINTER(u64), u32]
    L.nthip_multi_merge.argtypes = [vp, C.POINTER(vp), u64, C.c_int, u32]
    L.nthip_multi_kmer_bloom_query.argtypes = [vp, vp, C.c_uint16, C.c_uint8, C.POINTER(vp), u64, C.POINTER(vp), C.POINTER(u64), C.POINTER(u64), u32]
    L.nthip_multi_kmer_count_query.argtypes = [vp, vp, C.c_uint16, C.c_uint8, C.POINTER(vp), u64, C.POINTER(vp), C.POINTER(u64), u32]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if name not in ("nthip_version", "nthip_last_error"):
            fn.restype = C.c_int
    _lib = L
    return L


def _chk(rc):
    if rc != NTHIP_OK:
        raise NtHipError(rc, load().nthip_last_error().decode(errors="replace"))


def device_count():
    n = C.c_int(0)
    rc = load().nthip_device_count(C.byref(n))
    return n.value if rc == NTHIP_OK else 0


class Seeds:
    def __init__(self, ctx, seeds, k):
        self.ctx = ctx
        self.strings = [s.encode() if isinstance(s, str) else bytes(s) for s in seeds]
        arr = (C.c_char_p * len(self.strings))(*self.strings)
        h = C.c_void_p()
        asym = C.c_int(0)
        _chk(load().nthip_seeds_create(ctx.h, arr, len(self.strings), k, C.byref(h), C.byref(asym)))
        self.h = h
        self.asymmetric = bool(asym.value)
        self.n = len(self.strings)
        self.k = k

    def close(self):
        if self.h:
            load().nthip_seeds_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One device + one stream.  Raises NtHipError(NTHIP_ERR_NODEVICE) without a GPU."""

    def __init__(self, device=0):
        self.L = load()
        h = C.c_void_p()
        _chk(self.L.nthip_ctx_create(device, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.nthip_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- raw pointer API (device pointers as ints; e.g. torch tensor.data_ptr()) --
    def set_stream(self, stream_ptr):
        _chk(self.L.nthip_ctx_set_stream(self.h, C.c_void_p(stream_ptr)))

    def synchronize(self):
        _chk(self.L.nthip_ctx_synchronize(self.h))

    def set_profiling(self, on=True):
        _chk(self.L.nthip_ctx_set_profiling(self.h, 1 if on else 0))

    def last_kernel_ms(self):
        ms = C.c_float(0)
        name = C.c_char_p()
        _chk(self.L.nthip_last_kernel_ms(self.h, C.byref(ms), C.byref(name)))
        return ms.value, (name.value or b"").decode()

    def malloc(self, nbytes):
        p = C.c_void_p()
        _chk(self.L.nthip_malloc(self.h, nbytes, C.byref(p)))
        return p.value

    def free(self, ptr):
        _chk(self.L.nthip_free(self.h, C.c_void_p(ptr)))

    def h2d(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        _chk(self.L.nthip_memcpy_h2d(self.h, C.c_void_p(dptr), arr.ctypes.data, arr.nbytes))

    def d2h(self, arr, dptr):
        assert arr.flags["C_CONTIGUOUS"]
        _chk(self.L.nthip_memcpy_d2h(self.h, arr.ctypes.data, C.c_void_p(dptr), arr.nbytes))

    def take_dirty(self):
        """synchronise; True if an NTHIP_ASYNC batch since the last call met a non-base (its stream is invalid)"""
        d = C.c_int(0)
        _chk(self.L.nthip_ctx_take_dirty(self.h, C.byref(d)))
        return bool(d.value)

    def memset(self, dptr, value, nbytes):
        _chk(self.L.nthip_memset(self.h, C.c_void_p(dptr), value, nbytes))

    def kmer_hash_ptr(self, seqs, offsets, n_reads, fixed_len, stride, k, m, hashes, capacity,
                      counts=0, pos=0, fwd=0, rev=0, flags=0):
        rd = Reads(seqs, offsets or None, n_reads, fixed_len, stride)
        out = Out(hashes, capacity, counts or None, pos or None, fwd or None, rev or None)
        total = C.c_uint64(0)
        rc = self.L.nthip_kmer_hash(self.h, C.byref(rd), k, m, C.byref(out), C.byref(total), flags)
        if rc != NTHIP_OK:
            err = NtHipError(rc, self.L.nthip_last_error().decode(errors="replace"))
            err.total = total.value
            raise err
        return total.value

    # -- packed input: 2 bits per base + a validity stream, made once, hashed at any number of k ----------------
    def packed_size(self, n_bases):
        """bytes of the packed buffer for a batch of n_bases bytes -> (total, offset of the validity stream)"""
        tot, off = C.c_size_t(0), C.c_size_t(0)
        _chk(self.L.nthip_packed_size(C.c_uint64(n_bases), C.byref(tot), C.byref(off)))
        return tot.value, off.value

    def pack_reads_ptr(self, seqs, offsets, n_reads, fixed_len, stride, d_packed, flags=0):
        """device (or, with NTHIP_HOST_INPUT, host) reads -> the packed buffer at d_packed; -> bytes that are not bases"""
        rd = Reads(seqs, offsets or None, n_reads, fixed_len, stride)
        bad = C.c_uint64(0)
        _chk(self.L.nthip_pack_reads(self.h, C.byref(rd), C.c_void_p(d_packed), C.byref(bad), flags))
        return bad.value

    def pack_reads(self, data, fixed_len, n_reads, stride=0):
        """host reads -> (device pointer of the packed buffer [free it with .free()], invalid bytes)"""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        n_bases = (n_reads - 1) * (stride or fixed_len) + fixed_len if n_reads else 0
        tot, _ = self.packed_size(n_bases)
        d = self.malloc(tot)
        try:
            bad = self.pack_reads_ptr(data.ctypes.data, 0, n_reads, fixed_len, stride, d, flags=NTHIP_HOST_INPUT)
        except Exception:
            self.free(d)
            raise
        return d, bad

    def kmer_hash_packed(self, d_packed, k, m, fixed_len, n_reads, stride=0, clean=False, want_pos=False, capacity=None):
        """nthip_kmer_hash on a packed buffer (device pointer); results to the host like kmer_hash()"""
        nwin = max(fixed_len - k + 1, 0)
        cap = n_reads * nwin if capacity is None else capacity
        hashes = np.zeros(max(cap, 1) * m, np.uint64)
        counts = np.zeros(n_reads, np.uint64)
        pos = np.zeros(max(cap, 1), np.uint32) if want_pos else None
        flags = NTHIP_PACKED_INPUT | NTHIP_HOST_OUTPUT | (NTHIP_PACKED_CLEAN if clean else 0)
        total = self.kmer_hash_ptr(d_packed, 0, n_reads, fixed_len, stride, k, m, hashes.ctypes.data, cap,
                                   counts=counts.ctypes.data, pos=pos.ctypes.data if want_pos else 0, flags=flags)
        out = {"total": total, "hashes": hashes[: total * m].reshape(-1, m), "counts": counts}
        if want_pos:
            out["pos"] = pos[:total]
        return out

    def seed_hash_ptr(self, seqs, offsets, n_reads, fixed_len, stride, seeds, m2, hashes, capacity,
                      counts=0, pos=0, flags=0, fwd=0, rev=0):
        rd = Reads(seqs, offsets or None, n_reads, fixed_len, stride)
        out = Out(hashes, capacity, counts or None, pos or None, fwd or None, rev or None)
        total = C.c_uint64(0)
        rc = self.L.nthip_seed_hash(self.h, C.byref(rd), seeds.h, m2, C.byref(out), C.byref(total), flags)
        if rc != NTHIP_OK:
            err = NtHipError(rc, self.L.nthip_last_error().decode(errors="replace"))
            err.total = total.value
            raise err
        return total.value

    def kmer_extend(self, kmers, k, m, want_self=True, want_next=True, want_prev=True):
        """host k-mers (n*k bytes) -> dict of self [n,m], next [n,4,m], prev [n,4,m] (base order ACGT)"""
        kmers = np.ascontiguousarray(kmers, dtype=np.uint8)
        n = kmers.size // k
        se = np.zeros(n * m, np.uint64) if want_self else None
        nx = np.zeros(n * 4 * m, np.uint64) if want_next else None
        pv = np.zeros(n * 4 * m, np.uint64) if want_prev else None
        p = lambda a: a.ctypes.data if a is not None else None
        _chk(self.L.nthip_kmer_extend(self.h, kmers.ctypes.data, n, k, m, p(se), p(nx), p(pv),
                                      NTHIP_HOST_INPUT | NTHIP_HOST_OUTPUT))
        out = {}
        if want_self:
            out["self"] = se.reshape(n, m)
        if want_next:
            out["next"] = nx.reshape(n, 4, m)
        if want_prev:
            out["prev"] = pv.reshape(n, 4, m)
        return out

    def seed_extend(self, kmers, seeds, k, m2, want_self=True, want_next=True, want_prev=True):
        """host windows (n*k bytes) through the spaced seeds -> dict of self [n, n_seeds*m2], next / prev [n, 4, n_seeds*m2]
        (base order ACGT; what BlindSeedNtHash::roll / roll_back return from each window)"""
        kmers = np.ascontiguousarray(kmers, dtype=np.uint8)
        n = kmers.size // k
        sd = seeds if isinstance(seeds, Seeds) else Seeds(self, seeds, k)
        per = sd.n * m2
        se = np.zeros(n * per, np.uint64) if want_self else None
        nx = np.zeros(n * 4 * per, np.uint64) if want_next else None
        pv = np.zeros(n * 4 * per, np.uint64) if want_prev else None
        p = lambda a: a.ctypes.data if a is not None else None
        try:
            _chk(self.L.nthip_seed_extend(self.h, kmers.ctypes.data, n, sd.h, m2, p(se), p(nx), p(pv),
                                          NTHIP_HOST_INPUT | NTHIP_HOST_OUTPUT))
        finally:
            if sd is not seeds:
                sd.close()
        out = {}
        if want_self:
            out["self"] = se.reshape(n, per)
        if want_next:
            out["next"] = nx.reshape(n, 4, per)
        if want_prev:
            out["prev"] = pv.reshape(n, 4, per)
        return out

    # -- fused Bloom-filter consumers (filter: device memory, ceil(n_bits/32)*4 bytes) ---------
    def bloom_insert_ptr(self, seqs, n_reads, fixed_len, stride, k, m, d_filter, n_bits, flags=0, offsets=0):
        rd = Reads(seqs, offsets or None, n_reads, fixed_len, stride)
        total = C.c_uint64(0)
        _chk(self.L.nthip_kmer_bloom_insert(self.h, C.byref(rd), k, m, C.c_void_p(d_filter), n_bits,
                                            C.byref(total), flags))
        return total.value

    def bloom_query_ptr(self, seqs, n_reads, fixed_len, stride, k, m, d_filter, n_bits, hits=0, flags=0, offsets=0):
        rd = Reads(seqs, offsets or None, n_reads, fixed_len, stride)
        total, found = C.c_uint64(0), C.c_uint64(0)
        _chk(self.L.nthip_kmer_bloom_query(self.h, C.byref(rd), k, m, C.c_void_p(d_filter), n_bits,
                                           C.c_void_p(hits) if hits else None, C.byref(total), C.byref(found),
                                           flags))
        return total.value, found.value

    def trim(self):
        """release the buffers the context caches between calls (streaming driver, scan scratch)"""
        _chk(self.L.nthip_ctx_trim(self.h))

    def set_scratch_limit(self, nbytes):
        """bound what the consumers' rounds plan with and the context keeps between calls (0: the default, half of the device)"""
        _chk(self.L.nthip_ctx_set_scratch_limit(self.h, nbytes))

    def scratch_info(self):
        """-> (bytes the context holds now, the limit in force)"""
        kept, lim = C.c_size_t(0), C.c_size_t(0)
        _chk(self.L.nthip_ctx_scratch_info(self.h, C.byref(kept), C.byref(lim)))
        return kept.value, lim.value

    def minhash_ptr(self, seqs, n_reads, fixed_len, stride, k, m, sig, flags=0, offsets=0):
        """per-read MinHash signatures into sig[n_reads * m]; -> k-mers consumed"""
        rd = Reads(seqs, offsets or None, n_reads, fixed_len, stride)
        total = C.c_uint64(0)
        _chk(self.L.nthip_kmer_minhash(self.h, C.byref(rd), k, m, C.c_void_p(sig), C.byref(total), flags))
        return total.value

    def minhash(self, data, k, m, fixed_len, n_reads, stride=0, offsets=None):
        """-> (signatures [n_reads, m], k-mers consumed)"""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        if offsets is not None:
            offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        sig = np.zeros((n_reads, m), np.uint64)
        total = self.minhash_ptr(data.ctypes.data, n_reads, fixed_len, stride, k, m, sig.ctypes.data,
                                 flags=NTHIP_HOST_INPUT | NTHIP_HOST_OUTPUT,
                                 offsets=offsets.ctypes.data if offsets is not None else 0)
        return sig, total

    def stream_bloom_insert_ptr(self, d_hashes, n_values, d_filter, n_bits):
        _chk(self.L.nthip_stream_bloom_insert(self.h, C.c_void_p(d_hashes), n_values, C.c_void_p(d_filter), n_bits))

    # -- per-read (w, k)-minimizers ---------------------------------------------------------------
    def minimizers_ptr(self, seqs, n_reads, fixed_len, stride, k, w, d_hashes, d_pos, d_offsets, capacity, flags=0, offsets=0):
        """-> number of minimizers (NtHipError with .total set when capacity is too small)"""
        rd = Reads(seqs, offsets or None, n_reads, fixed_len, stride)
        total = C.c_uint64(0)
        rc = self.L.nthip_kmer_minimizers(self.h, C.byref(rd), k, w, C.c_void_p(d_hashes), C.c_void_p(d_pos) if d_pos else None,
                                          C.c_void_p(d_offsets), C.c_uint64(capacity), C.byref(total), flags)
        if rc != NTHIP_OK:
            err = NtHipError(rc, self.L.nthip_last_error().decode(errors="replace"))
            err.total = total.value
            raise err
        return total.value

    def minimizers_spans_ptr(self, d_buf, buf_bytes, d_starts, d_ends, n_reads, k, w, d_hashes, d_pos, d_offsets, capacity):
        """minimizers of reads given as spans of a device buffer; -> number of minimizers"""
        total = C.c_uint64(0)
        rc = self.L.nthip_kmer_minimizers_spans(self.h, C.c_void_p(d_buf), C.c_uint64(buf_bytes), C.c_void_p(d_starts), C.c_void_p(d_ends),
                                                C.c_uint64(n_reads), C.c_uint16(k), C.c_uint32(w), C.c_void_p(d_hashes),
                                                C.c_void_p(d_pos) if d_pos else None, C.c_void_p(d_offsets), C.c_uint64(capacity),
                                                C.byref(total))
        if rc != NTHIP_OK:
            err = NtHipError(rc, self.L.nthip_last_error().decode(errors="replace"))
            err.total = total.value
            raise err
        return total.value

    def minimizers(self, data, k, w, fixed_len, n_reads, stride=0, capacity=None, offsets=None, device_input=False):
        """host convenience: -> dict(offsets [n_reads + 1], pos, hashes); offsets: reads of any lengths instead of fixed_len;
        device_input: the reads (and offsets) are copied to the device first and handed over as device-resident buffers"""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        nwin = max(fixed_len - k + 1, 0)
        cap = (n_reads * nwin if offsets is None else int(data.size)) if capacity is None else capacity
        if offsets is not None:
            offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        d_h, d_p, d_o = self.malloc(max(8, cap * 8)), self.malloc(max(4, cap * 4)), self.malloc((n_reads + 1) * 8)
        d_in = d_of = 0
        try:
            if device_input:
                d_in = self.malloc(max(16, data.size + 16))
                self.h2d(d_in, data)
                if offsets is not None:
                    d_of = self.malloc(offsets.nbytes)
                    self.h2d(d_of, offsets)
                total = self.minimizers_ptr(d_in, n_reads, fixed_len, stride, k, w, d_h, d_p, d_o, cap, flags=0, offsets=d_of)
            else:
                total = self.minimizers_ptr(data.ctypes.data, n_reads, fixed_len, stride, k, w, d_h, d_p, d_o, cap,
                                            flags=NTHIP_HOST_INPUT, offsets=offsets.ctypes.data if offsets is not None else 0)
            offs = np.zeros(n_reads + 1, np.uint64)
            self.d2h(offs, d_o)
            hs, ps = np.zeros(total, np.uint64), np.zeros(total, np.uint32)
            if total:
                self.d2h(hs, d_h)
                self.d2h(ps, d_p)
            return dict(total=total, offsets=offs, pos=ps, hashes=hs)
        finally:
            for p in (d_h, d_p, d_o, d_in, d_of):
                if p:
                    self.free(p)

    # -- counting sketch (count-min, one-byte saturating counters) ---------------------------------
    def count_insert_ptr(self, seqs, n_reads, fixed_len, stride, k, m, d_counters, n_counters, flags=0, offsets=0):
        rd = Reads(seqs, offsets or None, n_reads, fixed_len, stride)
        total = C.c_uint64(0)
        _chk(self.L.nthip_kmer_count_insert(self.h, C.byref(rd), k, m, C.c_void_p(d_counters), C.c_uint64(n_counters),
                                            C.byref(total), flags))
        return total.value

    def count_insert(self, data, k, m, fixed_len, n_reads, d_counters, n_counters, stride=0, offsets=None):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        if offsets is not None:
            offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        return self.count_insert_ptr(data.ctypes.data, n_reads, fixed_len, stride, k, m, d_counters, n_counters,
                                     flags=NTHIP_HOST_INPUT, offsets=offsets.ctypes.data if offsets is not None else 0)

    def seed_bloom_insert_ptr(self, seqs, n_reads, fixed_len, stride, seeds, m2, d_filter, n_bits, flags=0, offsets=0):
        """every hash of every window SeedNtHash emits into the filter; -> windows consumed"""
        rd = Reads(seqs, offsets or None, n_reads, fixed_len, stride)
        total = C.c_uint64(0)
        _chk(self.L.nthip_seed_bloom_insert(self.h, C.byref(rd), seeds.h, m2, C.c_void_p(d_filter), C.c_uint64(n_bits), C.byref(total), flags))
        return total.value

    def seed_bloom_query_ptr(self, seqs, n_reads, fixed_len, stride, seeds, m2, d_filter, n_bits, hits=0, flags=0, offsets=0):
        """-> (windows tested, windows whose n_seeds * m2 bits are all set); hits: per read"""
        rd = Reads(seqs, offsets or None, n_reads, fixed_len, stride)
        total, found = C.c_uint64(0), C.c_uint64(0)
        _chk(self.L.nthip_seed_bloom_query(self.h, C.byref(rd), seeds.h, m2, C.c_void_p(d_filter), C.c_uint64(n_bits), C.c_void_p(hits or None),
                                           C.byref(total), C.byref(found), flags))
        return total.value, found.value

    def count_query_ptr(self, seqs, n_reads, fixed_len, stride, k, m, d_counters, n_counters, estimates, flags=0, offsets=0):
        """estimates: one byte per window of the batch (read r's at the windows of the reads before it); -> k-mers emitted"""
        rd = Reads(seqs, offsets or None, n_reads, fixed_len, stride)
        total = C.c_uint64(0)
        _chk(self.L.nthip_kmer_count_query(self.h, C.byref(rd), k, m, C.c_void_p(d_counters), C.c_uint64(n_counters),
                                           C.c_void_p(estimates), C.byref(total), flags))
        return total.value

    def count_query(self, data, k, m, fixed_len, n_reads, d_counters, n_counters, stride=0, offsets=None):
        """host reads -> (estimates per window as a numpy array, k-mers emitted)"""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        if offsets is not None:
            offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
            lens = (offsets[1:] - offsets[:-1]).astype(np.int64)
            n_win = int(np.maximum(lens - k + 1, 0).sum())
        else:
            n_win = n_reads * max(fixed_len - k + 1, 0)
        est = np.zeros(max(n_win, 1), np.uint8)
        total = self.count_query_ptr(data.ctypes.data, n_reads, fixed_len, stride, k, m, d_counters, n_counters, est.ctypes.data,
                                     flags=NTHIP_HOST_INPUT | NTHIP_HOST_OUTPUT, offsets=offsets.ctypes.data if offsets is not None else 0)
        return est[:n_win], total

    def stream_bloom_query_ptr(self, d_hashes, n_kmers, m, d_filter, n_bits, d_flags):
        """d_flags[i] = 1 when all m bits of k-mer i are set; -> the number of such k-mers"""
        found = C.c_uint64(0)
        _chk(self.L.nthip_stream_bloom_query(self.h, C.c_void_p(d_hashes), C.c_uint64(n_kmers), C.c_uint8(m), C.c_void_p(d_filter),
                                             C.c_uint64(n_bits), C.c_void_p(d_flags), C.byref(found)))
        return found.value

    def stream_count_insert_ptr(self, d_hashes, n_values, d_counters, n_counters):
        _chk(self.L.nthip_stream_count_insert(self.h, C.c_void_p(d_hashes), C.c_uint64(n_values), C.c_void_p(d_counters),
                                              C.c_uint64(n_counters)))

    def stream_count_query_ptr(self, d_hashes, n_kmers, m, d_counters, n_counters, d_estimates):
        _chk(self.L.nthip_stream_count_query(self.h, C.c_void_p(d_hashes), C.c_uint64(n_kmers), C.c_uint8(m),
                                             C.c_void_p(d_counters), C.c_uint64(n_counters), C.c_void_p(d_estimates)))

    def bloom_new(self, n_bits):
        """zeroed device filter of n_bits bits; returns (device pointer, bytes)"""
        nbytes = (n_bits + 31) // 32 * 4
        d = self.malloc(nbytes)
        self.memset(d, 0, nbytes)
        return d, nbytes

    def bloom_insert(self, data, k, m, fixed_len, n_reads, d_filter, n_bits, stride=0, offsets=None):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        if offsets is not None:
            offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        return self.bloom_insert_ptr(data.ctypes.data, n_reads, fixed_len, stride, k, m, d_filter, n_bits,
                                     flags=NTHIP_HOST_INPUT, offsets=offsets.ctypes.data if offsets is not None else 0)

    def bloom_query(self, data, k, m, fixed_len, n_reads, d_filter, n_bits, stride=0, offsets=None):
        """-> (hits per read, k-mers tested, k-mers found)"""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        if offsets is not None:
            offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        hits = np.zeros(n_reads, np.uint64)
        total, found = self.bloom_query_ptr(data.ctypes.data, n_reads, fixed_len, stride, k, m, d_filter, n_bits,
                                            hits=hits.ctypes.data, flags=NTHIP_HOST_INPUT | NTHIP_HOST_OUTPUT,
                                            offsets=offsets.ctypes.data if offsets is not None else 0)
        return hits, total, found

    # -- FASTQ / FASTA: device indexer, spans, file streaming ------------------------------------
    def fastx_index_ptr(self, d_buf, n_bytes, fmt, d_starts, d_ends, capacity):
        """-> (records, consumed bytes, malformed flag)"""
        n, cons, bad = C.c_uint64(0), C.c_uint64(0), C.c_int(0)
        _chk(self.L.nthip_fastx_index(self.h, C.c_void_p(d_buf), n_bytes, fmt, C.c_void_p(d_starts),
                                      C.c_void_p(d_ends), capacity, C.byref(n), C.byref(cons), C.byref(bad)))
        return n.value, cons.value, bad.value

    def fasta_compact_ptr(self, d_raw, n_bytes, d_seqs, d_offsets, capacity):
        """-> (records, sequence bytes)"""
        n, sb = C.c_uint64(0), C.c_uint64(0)
        rc = self.L.nthip_fasta_compact(self.h, C.c_void_p(d_raw), n_bytes, C.c_void_p(d_seqs), C.c_void_p(d_offsets),
                                        capacity, C.byref(n), C.byref(sb))
        if rc != NTHIP_OK:
            err = NtHipError(rc, self.L.nthip_last_error().decode(errors="replace"))
            err.n_records = n.value
            raise err
        return n.value, sb.value

    def kmer_hash_spans_ptr(self, d_buf, buf_bytes, d_starts, d_ends, n_reads, k, m, hashes, capacity,
                            counts=0, pos=0, flags=0):
        out = Out(hashes, capacity, counts or None, pos or None, None, None)
        total = C.c_uint64(0)
        rc = self.L.nthip_kmer_hash_spans(self.h, C.c_void_p(d_buf), buf_bytes, C.c_void_p(d_starts),
                                          C.c_void_p(d_ends), n_reads, k, m, C.byref(out), C.byref(total), flags)
        if rc != NTHIP_OK:
            err = NtHipError(rc, self.L.nthip_last_error().decode(errors="replace"))
            err.total = total.value
            raise err
        return total.value

    def seed_hash_spans_ptr(self, d_buf, buf_bytes, d_starts, d_ends, n_reads, seeds, m2, hashes, capacity,
                            counts=0, pos=0, flags=0):
        out = Out(hashes, capacity, counts or None, pos or None, None, None)
        total = C.c_uint64(0)
        _chk(self.L.nthip_seed_hash_spans(self.h, C.c_void_p(d_buf), buf_bytes, C.c_void_p(d_starts),
                                          C.c_void_p(d_ends), n_reads, seeds.h, m2, C.byref(out), C.byref(total), flags))
        return total.value

    def fastx_kmer_hash_file(self, path, fmt, k, m, chunk_bytes=0, on_batch=None, seeds=None):
        """stream a file; on_batch(FastxBatch) is called per batch (device pointers).  -> FastxStats
        seeds: a Seeds object -> SeedNtHash with m hashes per seed (k is the seeds' k)"""
        stats = FastxStats()
        err = []

        def tramp(_user, bp):
            try:
                if on_batch is not None:
                    on_batch(bp.contents)
                return 0
            except Exception as e:  # noqa: BLE001 -- must not unwind through the C frame
                err.append(e)
                return 1
        cb = FASTX_FN(tramp)
        if seeds is not None:
            rc = self.L.nthip_fastx_seed_hash_file(self.h, os.fsencode(path), fmt, seeds.h, m, chunk_bytes, cb, None,
                                                   C.byref(stats))
        else:
            rc = self.L.nthip_fastx_kmer_hash_file(self.h, os.fsencode(path), fmt, k, m, chunk_bytes, cb, None,
                                                   C.byref(stats))
        if err:
            raise err[0]
        _chk(rc)
        return stats

    def synth_reads_ptr(self, dptr, first_read, n_reads, length, seed=42):
        _chk(self.L.nthip_synth_reads(self.h, C.c_void_p(dptr), first_read, n_reads, length, seed))

    def checksum_ptr(self, dptr, n):
        s, x = C.c_uint64(0), C.c_uint64(0)
        _chk(self.L.nthip_checksum(self.h, C.c_void_p(dptr), n, C.byref(s), C.byref(x)))
        return s.value, x.value

    def copy_bench_ptr(self, dst, src, nbytes, reps=5):
        ms = C.c_float(0)
        _chk(self.L.nthip_copy_bench(self.h, C.c_void_p(dst), C.c_void_p(src), nbytes, reps, C.byref(ms)))
        return ms.value

    def fill_bench_ptr(self, dst, nbytes, reps=5):
        ms = C.c_float(0)
        _chk(self.L.nthip_fill_bench(self.h, C.c_void_p(dst), nbytes, reps, C.byref(ms)))
        return ms.value

    def malloc_probed(self, nbytes, candidates=3):
        """placement-aware allocation (nthip_malloc_probed): -> (device pointer, fill rate in GB/s, candidates measured)"""
        p, g, t = C.c_void_p(), C.c_double(0), C.c_int(0)
        _chk(self.L.nthip_malloc_probed(self.h, nbytes, candidates, C.byref(p), C.byref(g), C.byref(t)))
        return p.value, g.value, t.value

    def reload_tuning(self):
        """Re-read the NTHIP_TUNE_* environment knobs (they are read once, when the context is created)."""
        _chk(self.L.nthip_ctx_reload_tuning(self.h))

    # -- numpy convenience (host buffers, staged by the library) ------------------
    @staticmethod
    def _cap(offsets, fixed_len, stride, n_reads, k):
        if offsets is not None:
            lens = (offsets[1:] - offsets[:-1]).astype(np.int64)
            return int(np.maximum(lens - k + 1, 0).sum())
        return int(n_reads * max(fixed_len - k + 1, 0))

    def kmer_hash(self, data, k, m, offsets=None, fixed_len=0, stride=0, n_reads=None,
                  want_pos=False, want_strands=False, want_counts=True, flags=0):
        """Hash host reads (staged by the library); returns a dict of numpy arrays."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        if offsets is not None:
            offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
            n_reads = len(offsets) - 1
        cap = max(self._cap(offsets, fixed_len, stride, n_reads, k), 1)
        hashes = np.zeros(cap * m, np.uint64)
        counts = np.zeros(max(n_reads, 1), np.uint64) if want_counts else None
        pos = np.zeros(cap, np.uint32) if want_pos else None
        fwd = np.zeros(cap, np.uint64) if want_strands else None
        rev = np.zeros(cap, np.uint64) if want_strands else None
        p = lambda a: a.ctypes.data if a is not None else 0
        total = self.kmer_hash_ptr(p(data), p(offsets), n_reads, fixed_len, stride, k, m, p(hashes), cap,
                                   p(counts), p(pos), p(fwd), p(rev),
                                   flags | NTHIP_HOST_INPUT | NTHIP_HOST_OUTPUT)
        out = {"total": total, "hashes": hashes[: total * m].reshape(total, m)}
        if want_counts:
            out["counts"] = counts[:n_reads]
        if want_pos:
            out["pos"] = pos[:total]
        if want_strands:
            out["fwd"] = fwd[:total]
            out["rev"] = rev[:total]
        return out

    def seed_hash(self, data, seeds, k, m2, offsets=None, fixed_len=0, stride=0, n_reads=None,
                  want_pos=False, want_counts=True, want_strands=False, flags=0):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        if offsets is not None:
            offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
            n_reads = len(offsets) - 1
        sd = seeds if isinstance(seeds, Seeds) else Seeds(self, seeds, k)
        per = sd.n * m2
        cap = max(self._cap(offsets, fixed_len, stride, n_reads, k), 1)
        hashes = np.zeros(cap * per, np.uint64)
        counts = np.zeros(max(n_reads, 1), np.uint64) if want_counts else None
        pos = np.zeros(cap, np.uint32) if want_pos else None
        fwd = np.zeros(cap * sd.n, np.uint64) if want_strands else None
        rev = np.zeros(cap * sd.n, np.uint64) if want_strands else None
        p = lambda a: a.ctypes.data if a is not None else 0
        total = self.seed_hash_ptr(p(data), p(offsets), n_reads, fixed_len, stride, sd, m2, p(hashes), cap,
                                   p(counts), p(pos), flags | NTHIP_HOST_INPUT | NTHIP_HOST_OUTPUT,
                                   p(fwd), p(rev))
        out = {"total": total, "hashes": hashes[: total * per].reshape(total, per)}
        if want_strands:
            out["fwd"] = fwd[: total * sd.n].reshape(total, sd.n)
            out["rev"] = rev[: total * sd.n].reshape(total, sd.n)
        if want_counts:
            out["counts"] = counts[:n_reads]
        if want_pos:
            out["pos"] = pos[:total]
        return out


class Multi:
    """Several devices of one node (nthip_multi_*): host-resident batches, cut into contiguous shards of reads, one per
    device, hashed concurrently.  devices=None: every visible device; a device may be listed more than once."""

    def __init__(self, devices=None):
        self.L = load()
        self.h = C.c_void_p()
        if devices is None:
            _chk(self.L.nthip_multi_create(None, 0, C.byref(self.h)))
        else:
            arr = (C.c_int * len(devices))(*devices)
            _chk(self.L.nthip_multi_create(arr, len(devices), C.byref(self.h)))

    def close(self):
        if self.h:
            self.L.nthip_multi_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_count(self):
        n = C.c_int(0)
        _chk(self.L.nthip_multi_device_count(self.h, C.byref(n)))
        return n.value

    # -- device-resident shards, consumers, the merge of their tables over peer copies (round 4) --
    def ctx(self, index):
        """the context of the index-th device of the set (owned by the set: do not close it)"""
        h = C.c_void_p()
        _chk(self.L.nthip_multi_ctx(self.h, index, C.byref(h)))
        c = Context.__new__(Context)
        c.L = self.L
        c.h = h
        c.close = lambda: None
        return c

    def _shards(self, shards):
        arr = (Reads * len(shards))()
        for i, (seqs, offsets, n_reads, fixed_len, stride) in enumerate(shards):
            arr[i] = Reads(seqs, offsets or None, n_reads, fixed_len, stride)
        return arr

    def _tables(self, ptrs):
        return (C.c_void_p * len(ptrs))(*[C.c_void_p(p) for p in ptrs])

    def kmer_hash_shards(self, shards, k, m, outs, flags=0):
        """shards: per device (seqs, offsets, n_reads, fixed_len, stride) with device pointers of that device; outs: per device
        (d_hashes, capacity).  -> per-device totals"""
        oa = (Out * len(outs))()
        for i, (d_h, cap) in enumerate(outs):
            oa[i] = Out(d_h, cap, None, None, None, None)
        tots = (C.c_uint64 * len(shards))()
        _chk(self.L.nthip_multi_kmer_hash_shards(self.h, self._shards(shards), k, m, oa, tots, flags))
        return list(tots)

    def bloom_insert(self, shards, k, m, d_filters, n_bits, flags=0):
        total = C.c_uint64(0)
        _chk(self.L.nthip_multi_kmer_bloom_insert(self.h, self._shards(shards), k, m, self._tables(d_filters), n_bits, C.byref(total), flags))
        return total.value

    def count_insert(self, shards, k, m, d_counters, n_counters, flags=0):
        total = C.c_uint64(0)
        _chk(self.L.nthip_multi_kmer_count_insert(self.h, self._shards(shards), k, m, self._tables(d_counters), n_counters, C.byref(total), flags))
        return total.value

    def minhash_set(self, shards, k, m, d_sigs, flags=0):
        total = C.c_uint64(0)
        _chk(self.L.nthip_multi_kmer_minhash_set(self.h, self._shards(shards), k, m, self._tables(d_sigs), C.byref(total), flags))
        return total.value

    def merge(self, d_tables, nbytes, op, flags=0):
        _chk(self.L.nthip_multi_merge(self.h, self._tables(d_tables), nbytes, op, flags))

    def bloom_query(self, shards, k, m, d_filters, n_bits, d_hits=None, flags=0):
        """every device asks its copy of the filter about its shard; d_hits: per device, a pointer (memory of that device) or 0.
        -> (k-mers tested, k-mers found) over all devices"""
        total, found = C.c_uint64(0), C.c_uint64(0)
        hits = self._tables(d_hits) if d_hits is not None else None
        _chk(self.L.nthip_multi_kmer_bloom_query(self.h, self._shards(shards), k, m, self._tables(d_filters), n_bits, hits,
                                                 C.byref(total), C.byref(found), flags))
        return total.value, found.value

    def count_query(self, shards, k, m, d_counters, n_counters, d_estimates, flags=0):
        total = C.c_uint64(0)
        _chk(self.L.nthip_multi_kmer_count_query(self.h, self._shards(shards), k, m, self._tables(d_counters), n_counters,
                                                 self._tables(d_estimates), C.byref(total), flags))
        return total.value

    def fastx_kmer_hash_file(self, path, fmt, k, m, chunk_bytes=0, on_batch=None):
        """stream a FASTQ / single-line FASTA file over the devices; on_batch(FastxBatch) sees every batch once, in file
        order (device pointers on batch.device, that device current on the calling thread).  -> FastxStats"""
        stats = FastxStats()
        err = []

        def tramp(_user, bp):
            try:
                if on_batch is not None:
                    on_batch(bp.contents)
                return 0
            except Exception as e:  # noqa: BLE001 -- must not unwind through the C frame
                err.append(e)
                return 1
        cb = FASTX_FN(tramp)
        rc = self.L.nthip_multi_fastx_kmer_hash_file(self.h, os.fsencode(path), fmt, k, m, chunk_bytes, cb, None,
                                                     C.byref(stats))
        if err:
            raise err[0]
        _chk(rc)
        return stats

    def kmer_hash(self, data, k, m, offsets=None, fixed_len=0, stride=0, n_reads=None, want_pos=False):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        if offsets is not None:
            offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
            n_reads = len(offsets) - 1
        cap = max(Context._cap(offsets, fixed_len, stride, n_reads, k), 1)
        hashes = np.zeros(cap * m, np.uint64)
        counts = np.zeros(max(n_reads, 1), np.uint64)
        pos = np.zeros(cap, np.uint32) if want_pos else None
        p = lambda a: a.ctypes.data if a is not None else 0
        rd = Reads(p(data), p(offsets) or None, n_reads, fixed_len, stride)
        out = Out(p(hashes), cap, p(counts), p(pos) or None, None, None)
        total = C.c_uint64(0)
        _chk(self.L.nthip_multi_kmer_hash(self.h, C.byref(rd), k, m, C.byref(out), C.byref(total)))
        res = {"total": total.value, "hashes": hashes[: total.value * m].reshape(total.value, m), "counts": counts[:n_reads]}
        if want_pos:
            res["pos"] = pos[: total.value]
        return res

    def seed_hash(self, data, seeds, k, m2, offsets=None, fixed_len=0, stride=0, n_reads=None, want_pos=False):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        if offsets is not None:
            offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
            n_reads = len(offsets) - 1
        bs = [s_.encode("latin-1") for s_ in seeds]
        arr = (C.c_char_p * len(bs))(*bs)
        ms = C.c_void_p()
        _chk(self.L.nthip_multi_seeds_create(self.h, arr, len(bs), k, C.byref(ms), None))
        try:
            per = len(bs) * m2
            cap = max(Context._cap(offsets, fixed_len, stride, n_reads, k), 1)
            hashes = np.zeros(cap * per, np.uint64)
            counts = np.zeros(max(n_reads, 1), np.uint64)
            pos = np.zeros(cap, np.uint32) if want_pos else None
            p = lambda a: a.ctypes.data if a is not None else 0
            rd = Reads(p(data), p(offsets) or None, n_reads, fixed_len, stride)
            out = Out(p(hashes), cap, p(counts), p(pos) or None, None, None)
            total = C.c_uint64(0)
            _chk(self.L.nthip_multi_seed_hash(self.h, C.byref(rd), ms, m2, C.byref(out), C.byref(total)))
        finally:
            self.L.nthip_multi_seeds_destroy(ms)
        res = {"total": total.value, "hashes": hashes[: total.value * per].reshape(total.value, per), "counts": counts[:n_reads]}
        if want_pos:
            res["pos"] = pos[: total.value]
        return res
