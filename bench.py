#!/usr/bin/env python3
"""bench.py -- k-mers hashed/sec of the ntHash hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config c2|c3|c4|ref] [--reads R]

One "step" = one pass of the hot path (nthip_kmer_hash / nthip_seed_hash through the C-ABI) over the rank's
device-resident batch of synthetic reads.

N = 1: BASELINE.json configs[1] -- NtHash k=31 canonical, 1 hash/k-mer, 100 M x 150 bp reads on one MI355X
(15 GB in, 96 GB of hashes out, both resident in HBM).
N > 1: BASELINE.json configs[4] -- the 1 G x 150 bp job in shards of 125 M reads per GPU; rank r hashes reads
[r*125 M, (r+1)*125 M).  Reads are independent (include/nthash/nthash.hpp:196-204 of the reference: all state is
per object), so there is NO data-path collective ("weak" scaling); torch.distributed (RCCL) carries the barrier, the
max-over-ranks time and the per-rank results.  `python bench.py --gpus N` launches its own N ranks (one process per
GPU under torch.distributed.run); under an external launcher (WORLD_SIZE set) it just runs its rank.

Rank 0 prints ONE JSON line: BASELINE.json's metric plus
  roofline      algorithmic HBM bytes per launch / HIP-event duration of the dominant kernel, against the 8 TB/s
                spec peak AND against the write / copy rates measured on this box in the same process,
  verify        the on-device checksum of the WHOLE hash stream against the reference's
                (tests/golden/bench_checksums.json, made by the real reference library),
  secondary     (N = 1) the other single-GPU configs of BASELINE.json and a variable-length batch, same clock, 3 steps each,
  cpu_baseline  the reference CPU library (oracle/_ref, kind "reference") or the C restatement (kind "port")
                timed on this host on a bounded sample.
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED_A = "1010101010101010101010101010101"
SEED_B = "1101101101101101011011011011011"
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
SHARD_READS_MULTI = 125_000_000  # BASELINE config 5: 1 G reads over 8 GPUs
TRAFFIC_FILE = "profiles/r05_traffic.json"

CONFIGS = {
    # name: description, read length, k, hashes per k-mer (m, or m per seed), default reads per GPU
    "c2": dict(desc="NtHash k=31 canonical, 1 hash/k-mer, 100M x 150bp", L=150, k=31, m=1, seeds=None,
               reads=100_000_000),
    # (launches: the full-size stream does not fit the device -- 384 / 528 GB -- and is made in that many equal launches into
    #  one buffer, PINNED so that the driver's run and a rocprofv3 run launch the same shapes whatever memory is free)
    "c3": dict(desc="NtHash k=31, m=4 hashes/k-mer (multi-hash), 100M x 150bp", L=150, k=31, m=4,
               seeds=None, reads=100_000_000, launches=2),
    "c4": dict(desc="SeedNtHash 2 spaced seeds k=31, m=3, 50M x 250bp", L=250, k=31, m=3,
               seeds=[SEED_A, SEED_B], reads=50_000_000, launches=4),
    # the same reads as c2, packed once (2 bits per base + a validity stream: nthip_pack_reads) and hashed from the packed
    # buffer: SURVEY 8(d)'s 8.3125 B per k-mer.  The pack pass is timed apart (pack_ms): it is paid once per batch, not per k
    "c2_packed": dict(desc="NtHash k=31 canonical, 1 hash/k-mer, 100M x 150bp, 2-bit packed input (nthip_pack_reads once)",
                      L=150, k=31, m=1, seeds=None, reads=100_000_000, packed=True, checksum_as="c2"),
    # the reference's own harness shape (examples/benchmark.cpp:9-39: 100 bp reads, NtHash(seq, 3, 64)),
    # scaled from its 1 M reads to a batch that fills the GPU
    "ref": dict(desc="NtHash k=64, m=3, 100bp reads (examples/benchmark.cpp shape), 100M reads", L=100, k=64,
                m=3, seeds=None, reads=100_000_000),
    # what a FASTQ batch looks like: reads of different lengths (spans of one buffer), one in ~1000 with an N --
    # the whole call (mark pass, scan, hash pass, reads with an N) is on the clock, not only the hash kernel
    "var": dict(desc="NtHash k=31, 1 hash/k-mer, 20M variable-length reads of 100-150 bp (spans), an N in 1 read of ~1000",
                L=150, lmin=100, k=31, m=1, seeds=None, reads=20_000_000),
    # the same batch under the one-pass output contract (NTHIP_OUT_READ_SLOTS: a read's k-mers at the slot its length
    # implies, counts[r] valid, zeros behind): no pass that marks the reads with non-bases, no compaction
    "var_slots": dict(desc="NtHash k=31, 1 hash/k-mer, 20M variable-length reads of 100-150 bp (spans), an N in 1 read of ~1000, "
                           "one-pass read-slots output (NTHIP_OUT_READ_SLOTS)",
                      L=150, lmin=100, k=31, m=1, seeds=None, reads=20_000_000, slots=True, checksum_as="var"),
    # fixed-length reads as a sequencer writes them: config 2's shape with an N in one read of ~1000 (the rule of "var" with
    # one length).  Compact stream: the optimistic dense pass gives up at the first N, then count -> scan -> hash (the next
    # batch of the shape starts at the count pass: the context remembers);
    # read slots: ONE dense pass that marks the vectors holding a non-base, the reads they touch redone in their slots
    "c2_dirty": dict(desc="NtHash k=31, 1 hash/k-mer, 20M x 150bp fixed-length reads, an N in 1 read of ~1000 (compact stream)",
                     L=150, lmin=150, k=31, m=1, seeds=None, reads=20_000_000, fixed=True),
    "c2_dirty_slots": dict(desc="NtHash k=31, 1 hash/k-mer, 20M x 150bp fixed-length reads, an N in 1 read of ~1000, read-slots "
                                "output (NTHIP_OUT_READ_SLOTS): one dense pass + the reads with an N redone",
                           L=150, lmin=150, k=31, m=1, seeds=None, reads=20_000_000, fixed=True, slots=True, checksum_as="c2_dirty"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))  # (c2_packed: see CONFIGS)
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (default: the config's size)")
    ap.add_argument("--chunk-reads", type=int, default=0,
                    help="reads per launch (outputs of c3/c4 exceed HBM: a ring buffer is reused)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the c3 / c4 / ref lines at N = 1")
    ap.add_argument("--no-peak", action="store_true", help="skip the measured fill / copy ceiling")
    ap.add_argument("--no-placement", action="store_true",
                    help="(the default since round 5) one allocation per buffer through nthip_malloc: no candidates measured")
    ap.add_argument("--placement", type=int, default=1,
                    help="candidates nthip_malloc_probed measures per big buffer (1, the default since round 5: one plain allocation)")
    ap.add_argument("--no-plain-pass", action="store_true",
                    help="skip the extra pass on plain hipMalloc buffers (roofline.frac_plain_alloc)")
    ap.add_argument("--cpu-sample-reads", type=int, default=0)
    ap.add_argument("--dist-consumer-reads", type=int, default=20_000_000,
                    help="reads per rank of the `dist_consumer` line (tests: reduced; the filter shrinks with it)")
    ap.add_argument("--no-dist-consumer", action="store_true",
                    help="skip the Bloom-insert-and-merge line of the N-GPU runs (`dist_consumer`)")
    ap.add_argument("--consumers-reads", type=int, default=0,
                    help="only the `consumers` object (Bloom / counting sketch / minimizers / MinHash) on this many reads")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------
# launching: `python bench.py --gpus N` becomes N ranks
# ---------------------------------------------------------------------------------------------------------
def self_launch(args):
    """Re-exec under torch.distributed.run with one process per GPU (never returns)."""
    share = os.environ.get("NTHASH_BENCH_SHARE_GPU") == "1"
    if not share:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} HIP device(s) visible -- refusing to report "
                             f"a {args.gpus}-GPU number from fewer GPUs")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    sys.stdout.flush()
    os.execvpe(sys.executable, cmd, env)


# ---------------------------------------------------------------------------------------------------------
# CPU baseline
# ---------------------------------------------------------------------------------------------------------
def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cgroup_cpu_max():
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            return open(p).read().strip()
        except OSError:
            continue
    return None


def cpu_baseline(cfg, sample_reads):
    """Time the reference CPU library on a bounded sample of the same workload (same reads, same k / m)."""
    import numpy as np

    from oracle.pyoracle import Oracle, Reference
    L, k, m, seeds = cfg["L"], cfg["k"], cfg["m"], cfg["seeds"]
    usable = len(os.sched_getaffinity(0))
    host = {"host_cpus": os.cpu_count(), "usable_cpus": usable, "cgroup_cpu_max": _cgroup_cpu_max(),
            "cpu_model": _cpu_model()}
    ref = Reference() if Reference.available() else None
    if ref is not None and ref.has_synth:
        sec, nk, _ = ref.bench_synth(0, sample_reads, L, k, m, seeds=seeds, threads=1)
        out = {"value": nk / sec, "unit": "kmers/s", "cores": 1, "kind": "reference",
               "sample": f"{sample_reads} x {L}bp synthetic reads (same generator, seed 42), {nk} k-mers in "
                         f"{sec:.2f}s, one iterator per read, every hash consumed, timed inside the library"}
        out.update(host)
        # the reference has no threaded path: this is OUR OpenMP parallel-for over reads around it.  One thread
        # per usable CPU, reads first-touched by the thread that hashes them, pool warm, sample >= 3 s.
        # (threads the cgroup quota grants, not the CPUs the OS shows: a 16-CPU lease on a 256-CPU host)
        quota = usable
        try:
            q, per = (_cgroup_cpu_max() or "max").split()[:2]
            if q != "max":
                quota = max(1, int(q) // int(per))
        except Exception:
            pass
        nt = max(1, min(usable, quota))
        if nt > 1:
            n_mt = min(cfg["reads"], max(sample_reads, 250_000 * nt))
            sec1, nk1, used = ref.bench_synth(0, n_mt, L, k, m, seeds=seeds, threads=nt)  # calibration pass
            reps = max(1, min(64, int(3.5 / max(sec1, 1e-3)) + 1))
            sec2, nk2, used = ref.bench_synth(0, n_mt, L, k, m, seeds=seeds, threads=nt, repeats=reps)
            out["openmp"] = {"value": nk2 / sec2, "unit": "kmers/s", "cores": used,
                             "speedup_vs_1_thread": (nk2 / sec2) / out["value"],
                             "sample": f"{n_mt} reads x {reps} passes, {nk2} k-mers in {sec2:.2f}s",
                             "note": "OpenMP parallel-for over reads added by the harness; threads = cgroup cpu.max"}
        return out
    impl = ref if ref is not None else Oracle()
    data = impl.synth_reads(0, sample_reads, L, 42)
    t0 = time.perf_counter()
    if seeds is None:
        _acc, nk = impl.bench_kmer(data, sample_reads, L, k, m, threads=1)
    elif impl.kind == "reference":
        _acc, nk = impl.bench_seed(data, sample_reads, L, seeds, k, m, threads=1)
    else:
        offs = np.arange(sample_reads + 1, dtype=np.uint64) * L
        nk = impl.seed_batch(data, offs, seeds, k, m, want_pos=False)["total"]
    t1 = time.perf_counter() - t0
    out = {"value": nk / t1, "unit": "kmers/s", "cores": 1, "kind": impl.kind,
           "sample": f"{sample_reads} x {L}bp synthetic reads (same generator, seed 42), {nk} k-mers in {t1:.2f}s, "
                     f"iterator per read, every hash consumed"}
    out.update(host)
    return out


# ---------------------------------------------------------------------------------------------------------
# one workload on one rank
# ---------------------------------------------------------------------------------------------------------
_CHECKSUMS = None


def reference_checksum(name, first_read, n_reads):
    """The reference's (sum, xor, total) for this shard, from the committed fixture; None when it holds none."""
    global _CHECKSUMS
    if _CHECKSUMS is None:
        try:
            _CHECKSUMS = {(e["workload"], e["first_read"], e["n_reads"]): e
                          for e in json.load(open(os.path.join(ROOT, "tests", "golden", "bench_checksums.json")))}
        except (OSError, ValueError):
            _CHECKSUMS = {}
    return _CHECKSUMS.get((name, first_read, n_reads))


def _splitmix64(x):
    import numpy as np
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def var_reads(first_read, n_reads, len_min, len_max, seed=42):
    """Shape of the variable-length workload (the rule tests/golden/gen_bench_checksums.py's reference run follows,
    oracle/ref_shim.cpp ref_synth_var_checksum): read r = the first len_r bytes of synthetic read r of length len_max,
    len_r = len_min + x % (len_max - len_min + 1), x = splitmix64(seed + 0xABCDEF + r); if (x >> 32) % 997 == 0 its byte
    (x >> 16) % len_r is an 'N'."""
    import numpy as np
    with np.errstate(over="ignore"):
        r = np.arange(first_read, first_read + n_reads, dtype=np.uint64)
        x = _splitmix64(np.uint64(seed) + np.uint64(0xABCDEF) + r)
    lens = np.uint64(len_min) + x % np.uint64(len_max - len_min + 1)
    return lens, (x >> np.uint64(32)) % np.uint64(997) == 0, (x >> np.uint64(16)) % lens


SLOW_FILL_GBPS = float(os.environ.get("NTHASH_BENCH_SLOW_FILL_GBPS", "6600"))  # (the env: to exercise the second round)  # write-only fill of a buffer in the slow placement class: 5.1-6.1 TB/s; the others 6.9-7.2 (r02 notes 11)
PLACE_CANDIDATES = 1  # allocations nthip_malloc_probed measures per big buffer (--placement N); 1: one plain allocation per buffer


class _DevView:
    """__cuda_array_interface__ over a raw device pointer, so that torch can slice / copy a buffer the library allocated"""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


class Workload:
    """Device-resident reads + output ring of one config on one rank."""

    def _alloc(self, what, nbytes):
        """The big buffers come from the library's placement-aware allocator: which pages hipMalloc hands out moves the
        headline kernel by up to 15 % on this GPU (profiles/r02_notes.md 11); a pipeline would allocate its ring this way."""
        cands = PLACE_CANDIDATES if nbytes >= (48 << 30) or PLACE_CANDIDATES <= 1 else PLACE_CANDIDATES + 2
        if self.plain:
            cands = 1  # one plain hipMalloc, unmeasured: what a caller who brings their own buffers gets
        free_b, _tot = self.torch.cuda.mem_get_info(self.dev)
        if nbytes > 0.45 * free_b:
            cands = 1  # no room for a second candidate
        ptr, gbps, tried = self.ctx.malloc_probed(nbytes, cands)  # (small buffers: two more candidates cost a few ms)
        if cands > 1 and gbps and gbps < SLOW_FILL_GBPS:
            # every candidate landed in the slow class (one default line in seven, profiles/r04_notes.md): a second round of
            # candidates while this one is HELD -- they cannot be handed the same pages -- if there is room for both
            free_b, _tot = self.torch.cuda.mem_get_info(self.dev)
            if nbytes + (16 << 30) < free_b:
                try:
                    ptr2, gbps2, tried2 = self.ctx.malloc_probed(nbytes, cands)
                except Exception:  # noqa: BLE001 -- no room after all: the first round's buffer stands
                    ptr2, gbps2, tried2 = None, None, 0
                tried += tried2
                if ptr2 and gbps2 and gbps2 > gbps:
                    self.ctx.free(ptr)
                    ptr, gbps = ptr2, gbps2
                elif ptr2:
                    self.ctx.free(ptr2)
        self.placement.append({"buffer": what, "GiB": round(nbytes / 2**30, 2), "fill_GBps": round(gbps, 1) if gbps else None,
                               "candidates_measured": tried})
        self._owned.append(ptr)
        return ptr

    def __init__(self, torch, ctx, dev, name, cfg, n_reads, first_read, chunk_reads=0, plain=False):
        import nthash_amd
        self.torch, self.ctx, self.dev, self.name, self.cfg = torch, ctx, dev, name, cfg
        self.plain = plain
        self.n_reads, self.first_read = n_reads, first_read
        L, k, m = cfg["L"], cfg["k"], cfg["m"]
        self.L, self.k, self.m = L, k, m
        self.nwin = L - k + 1
        self.per = m if cfg["seeds"] is None else len(cfg["seeds"]) * m
        self.var = None
        self.d_packed, self.pack = 0, None
        self.placement, self._owned = [], []
        if cfg.get("lmin"):  # variable-length reads: spans [r*L, r*L + len_r) of the fixed-length buffer
            self._init_var(torch, ctx, dev, n_reads, first_read)
            return
        # launches per step: outputs larger than the free memory are produced chunk by chunk into one buffer
        free_b, _tot_b = torch.cuda.mem_get_info(dev)
        out_bytes_per_read = self.nwin * self.per * 8
        # (a ring that takes most of the memory cannot be probed -- and need not be: it contains every page set there is)
        budget = int(free_b * 0.85) - n_reads * L
        pinned = 0
        if not chunk_reads and cfg.get("launches") and n_reads == cfg["reads"]:
            pinned = -(-n_reads // cfg["launches"])
            if pinned * out_bytes_per_read > budget:   # (a smaller device: as many launches as it takes)
                pinned = 0
        chunk = chunk_reads or pinned or min(n_reads, max(1, budget // out_bytes_per_read))
        chunk = min(chunk, n_reads)
        if chunk < n_reads and (n_reads % chunk or not (chunk_reads or pinned)):
            # keep chunks a multiple of the kernels' read tiles -- unless the caller named an exact divisor of the job
            # (equal launches: what a per-kernel profile of one config wants, see tools/profile_round.sh)
            chunk = max(256, chunk // 256 * 256)
        self.chunk = chunk
        self.n_chunks = (n_reads + chunk - 1) // chunk
        self.d_in = torch.as_tensor(_DevView(self._alloc("reads", n_reads * L), n_reads * L, "|u1"), device=dev)
        n_out = chunk * self.nwin * self.per
        self.d_out = torch.as_tensor(_DevView(self._alloc("hashes", n_out * 8), n_out, "<i8"), device=dev)
        ctx.synth_reads_ptr(self.d_in.data_ptr(), first_read, n_reads, L, 42)
        self.seeds = nthash_amd.Seeds(ctx, cfg["seeds"], k) if cfg["seeds"] else None
        torch.cuda.synchronize(dev)
        if cfg.get("packed"):
            tot, _off = ctx.packed_size(n_reads * L)
            self.d_packed = self._alloc("packed reads", tot)
            ctx.set_profiling(True)
            t0 = time.perf_counter()
            bad = ctx.pack_reads_ptr(self.d_in.data_ptr(), 0, n_reads, L, 0, self.d_packed)
            wall = (time.perf_counter() - t0) * 1e3
            ms, _name = ctx.last_kernel_ms()
            self.pack = {"pack_kernel_ms": ms, "pack_call_ms": wall, "invalid_bytes": bad, "packed_GiB": round(tot / 2**30, 2),
                         "pack_GBps_in": n_reads * L / (ms * 1e-3) / 1e9}
            assert bad == 0
        self.kernel_ms = []

    def _init_var(self, torch, ctx, dev, n_reads, first_read):
        import numpy as np
        L, k = self.L, self.k
        lens, has_n, n_pos = var_reads(first_read, n_reads, self.cfg["lmin"], L)
        starts = np.arange(n_reads, dtype=np.int64) * L
        ends = starts + lens.astype(np.int64)
        self.total_kmers = int(np.maximum(lens.astype(np.int64) - k + 1, 0).sum())
        self.total_bases = int(lens.sum())
        self.chunk, self.n_chunks = n_reads, 1
        self.d_in = torch.as_tensor(_DevView(self._alloc("reads", n_reads * L), n_reads * L, "|u1"), device=dev)
        ctx.synth_reads_ptr(self.d_in.data_ptr(), first_read, n_reads, L, 42)
        idx = torch.from_numpy((starts + n_pos.astype(np.int64))[has_n]).to(dev)
        self.d_in[idx] = ord("N")
        self.d_starts = torch.from_numpy(starts).to(dev)
        self.d_ends = torch.from_numpy(ends).to(dev)
        n_out = self.total_kmers * self.per
        self.d_out = torch.as_tensor(_DevView(self._alloc("hashes", n_out * 8), n_out, "<i8"), device=dev)
        self.d_counts = torch.zeros(n_reads, dtype=torch.int64, device=dev) if self.cfg.get("slots") else None
        self.var = dict(lens=lens, has_n=has_n, n_pos=n_pos)
        self.seeds = None
        torch.cuda.synchronize(dev)
        self.kernel_ms = []

    def launch(self, c):
        if self.var is not None and self.cfg.get("fixed"):  # the same batch through the fixed-length entry
            import nthash_amd.capi as capi
            if self.cfg.get("slots"):
                return self.ctx.kmer_hash_ptr(self.d_in.data_ptr(), 0, self.n_reads, self.L, 0, self.k, self.m, self.d_out.data_ptr(),
                                              self.total_kmers, counts=self.d_counts.data_ptr(), flags=capi.NTHIP_OUT_READ_SLOTS)
            return self.ctx.kmer_hash_ptr(self.d_in.data_ptr(), 0, self.n_reads, self.L, 0, self.k, self.m, self.d_out.data_ptr(),
                                          self.total_kmers)
        if self.var is not None and self.cfg.get("slots"):
            import nthash_amd.capi as capi
            # (total_kmers = every window of every read: the extent of the slot array)
            return self.ctx.kmer_hash_spans_ptr(self.d_in.data_ptr(), self.n_reads * self.L, self.d_starts.data_ptr(),
                                                self.d_ends.data_ptr(), self.n_reads, self.k, self.m, self.d_out.data_ptr(),
                                                self.total_kmers, counts=self.d_counts.data_ptr(), flags=capi.NTHIP_OUT_READ_SLOTS)
        if self.var is not None:
            return self.ctx.kmer_hash_spans_ptr(self.d_in.data_ptr(), self.n_reads * self.L, self.d_starts.data_ptr(),
                                                self.d_ends.data_ptr(), self.n_reads, self.k, self.m,
                                                self.d_out.data_ptr(), self.total_kmers)
        r0 = c * self.chunk
        nr = min(self.chunk, self.n_reads - r0)
        if self.d_packed:
            import nthash_amd.capi as capi
            assert self.n_chunks == 1
            return self.ctx.kmer_hash_ptr(self.d_packed, 0, nr, self.L, 0, self.k, self.m, self.d_out.data_ptr(),
                                          self.chunk * self.nwin, flags=capi.NTHIP_PACKED_INPUT | capi.NTHIP_PACKED_CLEAN)
        if self.seeds is None:
            return self.ctx.kmer_hash_ptr(self.d_in.data_ptr() + r0 * self.L, 0, nr, self.L, 0, self.k, self.m,
                                          self.d_out.data_ptr(), self.chunk * self.nwin)
        return self.ctx.seed_hash_ptr(self.d_in.data_ptr() + r0 * self.L, 0, nr, self.L, 0, self.seeds, self.m,
                                      self.d_out.data_ptr(), self.chunk * self.nwin)

    def step(self, record):
        done = 0
        for c in range(self.n_chunks):
            tot = self.launch(c)
            if record:
                ms, name = self.ctx.last_kernel_ms()
                self.kernel_ms.append((ms, name, tot))
            done += tot
        return done

    def verify(self):
        """Outside the timed region: one more pass, the WHOLE stream checksummed on the device and compared with the
        reference's checksum of the same reads; plus a full compare of the last chunk's first reads with the oracle."""
        import numpy as np
        out = {"method": "on-device wrapping sum + XOR of every hash of the stream vs the reference library's "
                         "(tests/golden/bench_checksums.json)", "ok": None}
        s = x = tot = 0
        for c in range(self.n_chunks):
            t = self.launch(c)
            cs, cx = self.ctx.checksum_ptr(self.d_out.data_ptr(), t * self.per)
            s = (s + cs) & 0xFFFFFFFFFFFFFFFF
            x ^= cx
            tot += t
        if self.cfg.get("slots"):  # the slot array holds zeros behind a read's k-mers: its checksum is the stream's; the
            out["slots"] = tot       # emitted total is the sum of the per-read counts
            tot = int(self.d_counts.sum().item())
        out.update(sum=format(s, "016x"), xor=format(x, "016x"), total=tot)
        want = reference_checksum(self.cfg.get("checksum_as", self.name), self.first_read, self.n_reads)
        if want is not None:
            out["ok"] = bool(want["sum"] == out["sum"] and want["xor"] == out["xor"] and want["total"] == tot)
            out["reference"] = {"sum": want["sum"], "xor": want["xor"], "total": want["total"]}
        else:
            out["note"] = "no committed reference checksum for this (workload, shard): spot check only"
        spot = None
        try:
            from oracle.pyoracle import Oracle
            orc = Oracle()
            if self.var is not None:
                from oracle.pyoracle import concat_reads
                nv = min(2000, self.n_reads)
                full = orc.synth_reads(self.first_read, nv, self.L, 42).reshape(nv, self.L).copy()
                for r in np.nonzero(self.var["has_n"][:nv])[0]:
                    full[r, int(self.var["n_pos"][r])] = ord("N")
                d, offs = concat_reads([full[r, : int(self.var["lens"][r])].tobytes() for r in range(nv)])
                w = orc.kmer_batch(d, offs, self.k, self.m, want_pos=False)
                if self.cfg.get("slots"):  # read r at the slot its length implies, its count in counts[r], zeros behind
                    nwin = np.maximum(self.var["lens"][:nv].astype(np.int64) - self.k + 1, 0)
                    slot = np.concatenate([[0], np.cumsum(nwin)])
                    host = self.d_out[: int(slot[-1]) * self.per].cpu().numpy().view(np.uint64).reshape(-1, self.per)
                    cnt = self.d_counts[:nv].cpu().numpy().astype(np.uint64)
                    exp = np.zeros_like(host)
                    wo = np.concatenate([[0], np.cumsum(w["counts"].astype(np.int64))])
                    for r in range(nv):
                        exp[slot[r]:slot[r] + int(w["counts"][r])] = w["hashes"][wo[r]:wo[r + 1]]
                    out["spot_vs_oracle"] = bool((host == exp).all() and (cnt == w["counts"]).all())
                    return out
                host = self.d_out[: w["total"] * self.per].cpu().numpy().view(np.uint64).reshape(-1, self.per)
                out["spot_vs_oracle"] = bool((host == w["hashes"]).all())
                return out
            last_r0 = (self.n_chunks - 1) * self.chunk  # d_out holds the last chunk
            nv = min(2000, self.n_reads - last_r0)
            host = self.d_out[: nv * self.nwin * self.per].cpu().numpy().view(np.uint64).reshape(-1, self.per)
            data = orc.synth_reads(self.first_read + last_r0, nv, self.L, 42)
            offs = np.arange(nv + 1, dtype=np.uint64) * self.L
            if self.seeds is None:
                w = orc.kmer_batch(data, offs, self.k, self.m, want_pos=False)["hashes"]
            else:
                w = orc.seed_batch(data, offs, self.cfg["seeds"], self.k, self.m, want_pos=False)["hashes"]
            spot = bool((host == w).all())
        except Exception as e:  # the oracle is a checker, not a dependency of the measurement
            spot = f"not checked: {e}"
        out["spot_vs_oracle"] = spot
        return out

    def roofline(self):
        b_in = 0.25 if self.d_packed else 1.0               # bytes per base read: ASCII, or 2-bit packed
        b_per_kmer = 8.0 * self.per + b_in * self.L / self.nwin  # 8*H + b_in*L/(L-k+1)
        if self.var is not None:  # the bases of the reads, once, and the hashes (the hash kernel; the mark pass reads them again)
            b_per_kmer = 8.0 * self.per + self.total_bases / self.total_kmers
        ms_list = [q[0] for q in self.kernel_ms]
        avg_ms = sum(ms_list) / len(ms_list)
        kmers_per_launch = sum(q[2] for q in self.kernel_ms) / len(self.kernel_ms)
        achieved = kmers_per_launch * b_per_kmer / (avg_ms * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "kernel": self.kernel_ms[0][1],
                "kernel_avg_ms": avg_ms, "bytes_per_kmer": b_per_kmer, "kmers_per_launch": kmers_per_launch}

    def free(self):
        if self.seeds is not None:
            self.seeds.close()
        self.d_in = self.d_out = self.d_starts = self.d_ends = None
        self.torch.cuda.synchronize(self.dev)
        for ptr in self._owned:
            self.ctx.free(ptr)
        self._owned = []
        self.torch.cuda.empty_cache()


def consumers(torch, ctx, dev, n_reads=20_000_000):
    """What the reference's callers do with hashes(), on the device: whole-call rates on 20 M x 150 bp
    (k = 31), each with a cheap exact or necessary check.  Not part of `value`; N = 1 only."""
    import numpy as np
    L, k = 150, 31
    nwin = L - k + 1
    kmers = n_reads * nwin
    out = {"reads": n_reads, "len": L, "k": k, "kmers": kmers, "unit": "kmers/s (whole call, device-resident reads)"}
    d_in = ctx.malloc(n_reads * L)
    owned = [d_in]
    try:
        ctx.synth_reads_ptr(d_in, 0, n_reads, L, 42)
        ctx.set_profiling(True)   # (HIP events around the kernel of record of every call: `roofline.kernel_ms` below)

        first_call_ms = {}

        def best(f, reps=3, name=None):
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                r = f()
                ts.append(time.perf_counter() - t0)
            if name:   # (the first call of a consumer in a context builds the buffers the context keeps: seconds, not on the clock)
                first_call_ms[name] = ts[0] * 1e3
            return min(ts), r

        def roof(alg_bytes, call_s, what, note=None):
            """a consumer's roofline: algorithmic bytes = the bases it must read (L / nwin B per k-mer) + what it must write;
            `frac` over the whole call (every kernel of it), `kernel` / `kernel_ms` = the call's kernel of record"""
            try:
                kms, kname = ctx.last_kernel_ms()
            except Exception:  # noqa: BLE001
                kms, kname = None, None
            r = {"bound": "hbm", "bytes": what, "algorithmic_bytes": alg_bytes, "achieved": alg_bytes / call_s / 1e9, "peak": HBM_PEAK_GBPS,
                 "unit": "GB/s", "frac": alg_bytes / call_s / 1e9 / HBM_PEAK_GBPS, "kernel": kname, "kernel_ms": kms, "traffic": None}
            if note:
                r["note"] = note
            return r
        in_bytes = n_reads * L
        # Bloom filter: a fresh 4 GiB filter per repetition, one hash per k-mer; every inserted k-mer must then be found
        n_bits = 1 << 35
        d_f = ctx.malloc(n_bits // 8)
        owned.append(d_f)

        def ins():
            ctx.memset(d_f, 0, n_bits // 8)
            t0 = time.perf_counter()
            tot = ctx.bloom_insert_ptr(d_in, n_reads, L, 0, k, 1, d_f, n_bits)
            return time.perf_counter() - t0, tot
        t_ins, tot = min(ins() for _ in range(3))
        d_hits = ctx.malloc(n_reads * 8)
        owned.append(d_hits)
        t_q, (tq, found) = best(lambda: ctx.bloom_query_ptr(d_in, n_reads, L, 0, k, 1, d_f, n_bits, hits=d_hits))
        hits_sum = int(torch.as_tensor(_DevView(d_hits, n_reads, "<i8"), device=dev).sum().item())
        # the binned query: what it moves per k-mer through its lists, both ways (bytes: DESIGN "bench line legend")
        list_bytes = 32   # 4+2 | 4+4+2 | 4+1 | 2+1+1 | 2+1 (entries, 16-bit places, answer bytes through the five kernels)
        out["bloom_query_4GiB"] = {"value": tq / t_q, "ms": t_q * 1e3, "check": "hits per read add up to the k-mers found; every inserted k-mer is found",
                                   "ok": bool(hits_sum == found == kmers),
                                   "roofline": roof(in_bytes + 8 * n_reads + n_bits // 8, t_q, "bases in + 8 B per read (hits) + the filter read once")}
        out["bloom_query_4GiB"]["roofline"]["list_traffic"] = {"bytes_per_kmer": list_bytes, "GBps": tq * list_bytes / t_q / 1e9,
                                                               "frac_of_peak": tq * list_bytes / t_q / 1e9 / HBM_PEAK_GBPS}
        try:   # the kernel it replaced, same filter, same process: one filter load per k-mer, a 128-byte line each
            os.environ["NTHIP_TUNE_BLOOM_QUERY"] = "2"
            import nthash_amd
            direct = nthash_amd.Context(torch.cuda.current_device())
            os.environ.pop("NTHIP_TUNE_BLOOM_QUERY", None)
            t_d, (tqd, foundd) = best(lambda: direct.bloom_query_ptr(d_in, n_reads, L, 0, k, 1, d_f, n_bits), reps=2)
            direct.close()
            out["bloom_query_4GiB"]["direct_kernel"] = {"value": tqd / t_d, "ms": t_d * 1e3, "same_answer": bool(foundd == found),
                                                        "line_traffic_GBps": tqd * 128 / t_d / 1e9}
        except Exception as e:  # noqa: BLE001
            out["bloom_query_4GiB"]["direct_kernel"] = {"error": str(e)}
        finally:
            os.environ.pop("NTHIP_TUNE_BLOOM_QUERY", None)
        t_q3, (tq3, found3) = best(lambda: ctx.bloom_query_ptr(d_in, n_reads, L, 0, k, 3, d_f, n_bits), reps=2)
        out["bloom_query_4GiB_m3"] = {"value": tq3 / t_q3, "ms": t_q3 * 1e3, "x_m1": t_q3 / t_q, "found": found3,
                                      "roofline": roof(in_bytes + 8 * n_reads + n_bits // 8, t_q3, "as m = 1")}
        # the same two queries on a batch HALF of whose k-mers are in the filter with all three hashes (the first half of the reads
        # inserted with m = 3): the second pass of m = 3 asks hashes()[1 ...] only for the k-mers whose first hash hit
        try:
            ctx.memset(d_f, 0, n_bits // 8)
            ctx.bloom_insert_ptr(d_in, n_reads // 2, L, 0, k, 3, d_f, n_bits)
            t_h1, (_t, f_h1) = best(lambda: ctx.bloom_query_ptr(d_in, n_reads, L, 0, k, 1, d_f, n_bits), reps=2)
            t_h3, (_t, f_h3) = best(lambda: ctx.bloom_query_ptr(d_in, n_reads, L, 0, k, 3, d_f, n_bits), reps=2)
            out["bloom_query_4GiB_m3"]["half_hit"] = {"x_m1": t_h3 / t_h1, "ms_m1": t_h1 * 1e3, "ms_m3": t_h3 * 1e3, "found_m1": f_h1, "found_m3": f_h3,
                                                      "ok": bool(f_h3 >= (n_reads // 2) * nwin and f_h3 <= f_h1)}
        except Exception as e:  # noqa: BLE001
            out["bloom_query_4GiB_m3"]["half_hit"] = {"error": str(e)}
        ctx.free(d_hits)
        owned.remove(d_hits)
        ctx.memset(d_f, 0, n_bits // 8)
        ctx.bloom_insert_ptr(d_in, n_reads, L, 0, k, 1, d_f, n_bits)   # (the kernel of record of an insert, for the line below)
        out["bloom_insert_fresh_4GiB"] = {"value": tot / t_ins, "ms": t_ins * 1e3, "check": "every inserted k-mer is found",
                                          "ok": bool(tot == kmers and tq == kmers and found == kmers),
                                          "roofline": roof(in_bytes + 2 * (n_bits // 8), t_ins, "bases in + the filter read and written once")}
        ctx.free(d_f)
        owned.remove(d_f)
        # counting sketch: 1 Gi one-byte counters, fresh; no counter saturates here, so the bytes add up to the k-mers
        n_cnt = 1 << 30
        d_c = ctx.malloc(n_cnt)
        owned.append(d_c)

        def cins():
            ctx.memset(d_c, 0, n_cnt)
            t0 = time.perf_counter()
            tot = ctx.count_insert_ptr(d_in, n_reads, L, 0, k, 1, d_c, n_cnt)
            return time.perf_counter() - t0, tot
        t_c, totc = min(cins() for _ in range(3))
        view = torch.as_tensor(_DevView(d_c, n_cnt, "|u1"), device=dev)
        s_bytes = int(view.sum(dtype=torch.int64).item())
        top = int(view.max().item())
        del view
        out["count_insert_fresh_1Gi_counters"] = {"value": totc / t_c, "ms": t_c * 1e3,
                                                  "check": "sum of the counters == k-mers inserted (largest counter %d)" % top,
                                                  "ok": bool(totc == kmers and (s_bytes == kmers or top == 255)),
                                                  "roofline": roof(in_bytes + 2 * n_cnt, t_c, "bases in + the counters read and written once")}
        # the sketch's read side on the reads: an estimate per window (binned, as the filter's query)
        d_e = ctx.malloc(n_reads * nwin)
        owned.append(d_e)
        t_cq, totq = best(lambda: ctx.count_query_ptr(d_in, n_reads, L, 0, k, 1, d_c, n_cnt, d_e))
        ev = torch.as_tensor(_DevView(d_e, n_reads * nwin, "|u1"), device=dev)
        e_min = int(ev.min().item())
        del ev
        out["count_query_1Gi_counters"] = {"value": totq / t_cq, "ms": t_cq * 1e3, "check": "every inserted k-mer's estimate is at least 1",
                                           "ok": bool(totq == kmers and e_min >= 1),
                                           "roofline": roof(in_bytes + n_reads * nwin + n_cnt, t_cq, "bases in + 1 B per window out + the counters read once")}
        ctx.free(d_e)
        owned.remove(d_e)
        ctx.free(d_c)
        owned.remove(d_c)
        # spaced seeds into a filter: BASELINE config 4's seed pair, 3 hashes per seed, 250 bp reads (a tenth of config 4's batch)
        import nthash_amd as _na
        n4, L4 = 5_000_000, 250
        d_in4 = ctx.malloc(n4 * L4)
        owned.append(d_in4)
        ctx.synth_reads_ptr(d_in4, 0, n4, L4, 42)
        sd = _na.Seeds(ctx, [SEED_A, SEED_B], 31)
        n_bits4 = 1 << 35
        d_f4 = ctx.malloc(n_bits4 // 8)
        owned.append(d_f4)
        ctx.memset(d_f4, 0, n_bits4 // 8)
        # (the first call builds the buffers the context keeps -- the round's hash stream, the lists: seconds when the driver has
        #  memory to give back first, profiles/r05_notes.md §12 -- so it is not the call on the clock: second and third on a fresh filter)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        ctx.seed_bloom_insert_ptr(d_in4, n4, L4, 0, sd, 3, d_f4, n_bits4)
        t_first4 = time.perf_counter() - t0
        t_s4, tot4 = 1e9, 0
        for _ in range(2):
            ctx.memset(d_f4, 0, n_bits4 // 8)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            tot4 = ctx.seed_bloom_insert_ptr(d_in4, n4, L4, 0, sd, 3, d_f4, n_bits4)
            t_s4 = min(t_s4, time.perf_counter() - t0)
        roof_s4 = roof(n4 * L4 + 2 * (n_bits4 // 8), t_s4, "bases in + the filter read and written once")  # (before the query runs:
        torch.cuda.synchronize(dev)                                                                      #  its kernel of record)
        t0 = time.perf_counter()
        ctx.seed_bloom_query_ptr(d_in4, n4, L4, 0, sd, 3, d_f4, n_bits4)
        t_firstq4 = time.perf_counter() - t0
        t_sq, (tq4, found4) = best(lambda: ctx.seed_bloom_query_ptr(d_in4, n4, L4, 0, sd, 3, d_f4, n_bits4), reps=2)
        win4 = n4 * (L4 - 31 + 1)
        out["seed_bloom_insert_c4_seeds"] = {"value": tot4 / t_s4, "ms": t_s4 * 1e3, "unit": "windows/s (6 hashes each)", "values_per_s": 6 * tot4 / t_s4,
                                             "check": "every window is consumed and found again", "ok": bool(tot4 == win4 and tq4 == win4 and found4 == win4),
                                             "first_call_ms": t_first4 * 1e3,
                                             "query": {"value": tq4 / t_sq, "ms": t_sq * 1e3, "values_per_s": 6 * tq4 / t_sq, "first_call_ms": t_firstq4 * 1e3,
                                                       "how": "the seeds' hashes through the regions of the filter (stream_query_binned)"},
                                             "roofline": roof_s4}
        sd.close()
        for p in (d_in4, d_f4):
            ctx.free(p)
            owned.remove(p)
        # (w, k)-minimizers, w = 10: density close to 2 / (w + 1) on random reads, offsets ascending
        w = 10
        cap = n_reads * (2 * nwin // (w + 1) + 4)
        d_h, d_p, d_o = ctx.malloc(cap * 8), ctx.malloc(cap * 4), ctx.malloc((n_reads + 1) * 8)
        owned += [d_h, d_p, d_o]
        t_m, totm = best(lambda: ctx.minimizers_ptr(d_in, n_reads, L, 0, k, w, d_h, d_p, d_o, cap))
        offs = torch.as_tensor(_DevView(d_o, n_reads + 1, "<i8"), device=dev)
        mono = bool((offs[1:] >= offs[:-1]).all().item()) and int(offs[-1].item()) == totm
        del offs
        dens = totm / kmers
        out["minimizers_w10"] = {"value": kmers / t_m, "ms": t_m * 1e3, "minimizers": totm, "density": dens,
                                 "check": "density within 10 % of 2 / (w + 1); offsets ascending, last == total",
                                 "ok": bool(mono and abs(dens - 2 / (w + 1)) < 0.1 * 2 / (w + 1)),
                                 "roofline": roof(in_bytes + 12 * totm + 8 * (n_reads + 1), t_m,
                                                  "bases in + 12 B per minimizer (hash, position) + 8 B per read (offsets)")}
        # the same reads given by offsets (what a FASTQ batch looks like to the consumer): the same minimizers
        d_of = ctx.malloc((n_reads + 1) * 8)
        owned.append(d_of)
        ctx.h2d(d_of, np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(L))
        t_mo, totmo = best(lambda: ctx.minimizers_ptr(d_in, n_reads, 0, 0, k, w, d_h, d_p, d_o, cap, offsets=d_of))
        out["minimizers_w10_offsets"] = {"value": kmers / t_mo, "ms": t_mo * 1e3, "minimizers": totmo,
                                         "check": "as many minimizers as the fixed-length call", "ok": bool(totmo == totm),
                                         "roofline": roof(in_bytes + 12 * totmo + 16 * (n_reads + 1), t_mo,
                                                          "bases + offsets in, 12 B per minimizer + 8 B per read out")}
        for p in (d_h, d_p, d_o, d_of):
            ctx.free(p)
            owned.remove(p)
        # per-read MinHash, 4 hashes per k-mer (fused: no stream is written)
        d_s = ctx.malloc(n_reads * 4 * 8)
        owned.append(d_s)
        t_s, tots = best(lambda: ctx.minhash_ptr(d_in, n_reads, L, 0, k, 4, d_s))
        out["minhash_m4"] = {"value": tots / t_s, "ms": t_s * 1e3, "ok": bool(tots == kmers),
                             "roofline": roof(in_bytes + 32 * n_reads, t_s, "bases in + 4 x 8 B per read (signatures)")}
    finally:
        torch.cuda.synchronize(dev)
        for p in owned:
            ctx.free(p)
    return out


def dist_consumer_line(torch, dist, ctx, dev, rank, world, share, barrier, n_reads=20_000_000, n_bits=1 << 33):
    """Bloom insert over the ranks: rank r inserts reads [r * n_reads, (r + 1) * n_reads) of the counter-based set into its own
    1 GiB filter (device-resident, the single-device call), then the filters are OR-merged over the ring (reduce-scatter +
    all-gather of nthash_amd/sharding.py on the job's process group) so that every rank holds the filter of ALL reads.
    Checked: every rank finds every k-mer of its own shard in the merged filter, and all ranks hold the same number of set
    bits.  `value` = all ranks' k-mers / (slowest insert + slowest merge)."""
    from nthash_amd.sharding import ring_merge_dist
    L, k = 150, 31
    nwin = L - k + 1
    kmers = n_reads * nwin
    d_in = ctx.malloc(n_reads * L)
    try:
        ctx.synth_reads_ptr(d_in, rank * n_reads, n_reads, L, 42)
        filt = torch.zeros(n_bits // 8, dtype=torch.uint8, device=dev)
        barrier()
        t0 = time.perf_counter()
        tot = ctx.bloom_insert_ptr(d_in, n_reads, L, 0, k, 1, filt.data_ptr(), n_bits)
        torch.cuda.synchronize(dev)
        t_ins = time.perf_counter() - t0
        barrier()
        t0 = time.perf_counter()
        if share:   # (test mode: the ranks share one GPU and talk over gloo, which moves host tensors)
            host = filt.cpu()
            ring_merge_dist(host, "or")
            filt.copy_(host)
        else:
            ring_merge_dist(filt, "or")
        torch.cuda.synchronize(dev)
        t_merge = time.perf_counter() - t0
        barrier()
        tq, found = ctx.bloom_query_ptr(d_in, n_reads, L, 0, k, 1, filt.data_ptr(), n_bits)
        bits = _popcount_u8(torch, filt)
        cpu_dev = "cpu" if share else dev
        t = torch.tensor([t_ins, t_merge], dtype=torch.float64, device=cpu_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        g = [torch.zeros(2, dtype=torch.float64, device=cpu_dev) for _ in range(world)]
        dist.all_gather(g, torch.tensor([float(bits), 1.0 if (tot == kmers and tq == kmers and found == kmers) else 0.0],
                                        dtype=torch.float64, device=cpu_dev))
        t_ins, t_merge = float(t[0]), float(t[1])
        moved = 2 * (world - 1) / world * (n_bits // 8)   # bytes every rank sends (and receives) over its ring links
        return {"what": "Bloom insert, one 1 GiB filter per rank, OR-merged over the ring (reduce-scatter + all-gather)",
                "reads_per_rank": n_reads, "kmers": kmers * world, "n_bits": n_bits,
                "value": kmers * world / (t_ins + t_merge), "unit": "kmers/s (all ranks; insert + merge)",
                "insert_ms": t_ins * 1e3, "merge_ms": t_merge * 1e3,
                "merge_GBps_per_rank": (moved / t_merge / 1e9) if world > 1 and t_merge > 0 else None,
                "set_bits": [int(float(q[0])) for q in g],
                "check": "every rank finds all k-mers of its shard in the merged filter; all ranks hold the same filter population",
                "ok": bool(all(float(q[1]) == 1.0 for q in g) and len({int(float(q[0])) for q in g}) == 1),
                "hardware_note": None if world > 1 else "world size 1: the ring has no step to take (the N > 1 path is the gloo / "
                                                        "peer-to-self tests' until the driver runs it on a node)"}
    finally:
        ctx.free(d_in)


def _popcount_u8(torch, t, chunk=1 << 28):
    """set bits of a uint8 tensor: a 256-entry table, a quarter GiB at a time (torch has no popcount; summing the BYTES,
    as this line once did, is a checksum and needs 8 x the filter as int64 -- ADVICE r04)"""
    lut = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int64, device=t.device)
    total = 0
    for i in range(0, t.numel(), chunk):
        total += int(lut[t[i:i + chunk].to(torch.int64)].sum().item())
    return total


def measured_peak(torch, ctx, dev, own_fill=()):
    """Write-only and copy rates of this box, same process, same clock: the achievable ceilings next to the spec.
    (32 GiB per launch: an 8 GiB fill lasts 1.2 ms and measures 5.6 TB/s on a box whose 24 GiB fill runs at 7.0.)
    hipMalloc hands out allocations of different speed classes (profiles/r05_notes.md 12), so ONE buffer can be slower
    than the workload's own: three separately made 32 GiB allocations are measured (each freed before the next), and the
    fill rate of the workload's own output buffer(s) -- the ceiling is the best of all of them."""
    nbytes = 32 << 30
    fills, copies = [], []
    for _ in range(3):
        ptr = ctx.malloc(nbytes)
        try:
            fills.append(nbytes / (ctx.fill_bench_ptr(ptr, nbytes, 5) * 1e-3) / 1e9)
            half = nbytes // 2
            copies.append(2 * half / (ctx.copy_bench_ptr(ptr + half, ptr, half, 5) * 1e-3) / 1e9)
        finally:
            ctx.free(ptr)
    own = list(own_fill)   # (the hash stream the timed kernel wrote, filled right after its verdict)
    best = max(fills + copies + own)
    return {"fill_GBps": max(fills), "copy_GBps": max(copies), "best_GBps": best, "fills_GBps": fills, "copies_GBps": copies,
            "own_output_buffer_fill_GBps": own, "allocations_measured": 3 + len(own),
            "how": "nthip_fill_bench (write-only, the kernels' copy-out pattern) on 32 GiB / nthip_copy_bench (50 % reads) on "
                   "16 GiB, best of 5, in this process, on three separately made allocations and on the workload's own output "
                   "buffer; the best of all is the ceiling"}


def main():
    args = parse()
    global PLACE_CANDIDATES
    PLACE_CANDIDATES = 1 if args.no_placement else max(1, args.placement)
    if args.consumers_reads:  # (tests: the consumers object alone, at a reduced size)
        import torch
        import nthash_amd
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        ctx = nthash_amd.Context(0)
        print(json.dumps({"consumers": consumers(torch, ctx, dev, args.consumers_reads)}), flush=True)
        ctx.close()
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    cfg = dict(CONFIGS[args.config])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(or run `python bench.py --gpus {args.gpus}` and let it launch them)")

    import torch
    import torch.distributed as dist

    import nthash_amd
    from nthash_amd.sharding import weak_shard

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # NTHASH_BENCH_SHARE_GPU=1 (testing only): all ranks on GPU 0 with a gloo group, so that the
    # multi-rank control flow can be exercised on a 1-GPU box; real runs use one GPU per rank + RCCL
    share = os.environ.get("NTHASH_BENCH_SHARE_GPU") == "1"
    n_dev = torch.cuda.device_count()
    if share:
        local_rank = 0
    elif n_dev == 1 and world > 1 and os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES")):
        local_rank = 0  # a launcher that pins one GPU per rank leaves every rank with device 0 only
    elif local_rank >= n_dev:
        raise SystemExit(f"rank {rank}: local rank {local_rank} but only {n_dev} HIP device(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # NTHASH_BENCH_FORCE_DIST=1: the process group is made at world size 1 as well (RCCL unless the GPU is shared), so that
    # the branch an N-GPU run takes -- init, barrier, MAX all-reduce, all-gather -- also runs on a one-GPU box
    use_dist = world > 1 or os.environ.get("NTHASH_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(s.getsockname()[1]))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if share:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)
    cpu_group_dev = "cpu" if share else dev

    n_reads = args.reads or (SHARD_READS_MULTI if (world > 1 and args.config == "c2") else cfg["reads"])
    ctx = nthash_amd.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    ctx.set_profiling(True)
    # this rank's shard of the global read set [rank*n_reads, (rank+1)*n_reads)
    first_read, _ = weak_shard(rank, n_reads)
    wl = Workload(torch, ctx, dev, args.config, cfg, n_reads, first_read, args.chunk_reads)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        wl.step(False)
    barrier()
    t0 = time.perf_counter()
    kmers = 0
    for _ in range(args.steps):
        kmers += wl.step(True)
    torch.cuda.synchronize(dev)
    my_dt = time.perf_counter() - t0
    barrier()
    dt = time.perf_counter() - t0
    if wl.var is None:  # (a variable-length batch emits what its reads hold: checked against the reference's total below)
        assert kmers == args.steps * n_reads * wl.nwin, (kmers, args.steps * n_reads * wl.nwin)

    verify = wl.verify()
    ok_local = (verify["ok"] is not False) and (verify["spot_vs_oracle"] is True)
    # the same workload once more on PLAIN hipMalloc buffers (no placement probe): the rate callers who bring their
    # own buffers see (torch tensors, a pipeline's ring).  Outside the timed region, 1 warm-up + 3 steps, kernel time.
    roof = wl.roofline()
    placement = wl.placement
    own_fill = []   # the ceiling of the buffer the timed kernel wrote (its verdict is in: the buffer may be overwritten)
    if world == 1 and not args.no_peak and getattr(wl, "d_out", None) is not None:
        try:
            nb = (wl.d_out.numel() * 8) & ~((1 << 20) - 1)
            if nb >= (4 << 30):
                own_fill.append(nb / (ctx.fill_bench_ptr(wl.d_out.data_ptr(), nb, 3) * 1e-3) / 1e9)
        except Exception:  # noqa: BLE001
            pass
    wl.free()
    plain_roof = None
    if PLACE_CANDIDATES > 1 and not args.no_plain_pass:
        try:
            wp = Workload(torch, ctx, dev, args.config, cfg, n_reads, first_read, args.chunk_reads, plain=True)
            wp.step(False)
            for _ in range(3):
                wp.step(True)
            torch.cuda.synchronize(dev)
            plain_roof = wp.roofline()
            wp.free()
        except Exception as e:
            plain_roof = {"error": str(e)}
    # the N-GPU consumer line (SURVEY 8e + 8f-1): every rank inserts its shard into its OWN Bloom filter; what crosses xGMI is
    # the filter, merged by the ring of nthash_amd/sharding.py (RCCL send / recv + a local OR: RCCL has no bitwise reduce)
    dist_consumer = None
    if use_dist and not args.no_dist_consumer:
        try:
            dcr = args.dist_consumer_reads
            dist_consumer = dist_consumer_line(torch, dist, ctx, dev, rank, world, share, barrier, dcr,
                                               1 << 33 if dcr >= 20_000_000 else 1 << 28)
        except Exception as e:  # noqa: BLE001
            dist_consumer = {"error": str(e)}
    per_rank = [kmers / my_dt]
    all_ok = ok_local
    checked_full = verify["ok"] is True
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=cpu_group_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        g = [torch.zeros(3, dtype=torch.float64, device=cpu_group_dev) for _ in range(world)]
        dist.all_gather(g, torch.tensor([kmers / my_dt, 1.0 if ok_local else 0.0, 1.0 if checked_full else 0.0],
                                        dtype=torch.float64, device=cpu_group_dev))
        per_rank = [float(q[0]) for q in g]
        all_ok = all(float(q[1]) == 1.0 for q in g)
        checked_full = all(float(q[2]) == 1.0 for q in g)

    if rank == 0:
        L, k = cfg["L"], cfg["k"]
        total_kmers = kmers * world
        if plain_roof is not None:
            roof["frac_plain_alloc"] = plain_roof.get("frac")
            roof["plain_alloc"] = ({"achieved": plain_roof["achieved"], "kernel_avg_ms": plain_roof["kernel_avg_ms"]}
                                   if "frac" in plain_roof else plain_roof)
        # HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE x2 gfx950 correction +
        # WRITE_SIZE, separate rocprofv3 --pmc runs of the same kernel); counters cannot be read inside this process
        try:
            tj = json.load(open(os.path.join(ROOT, TRAFFIC_FILE)))
            if args.config in tj:
                roof["traffic"] = tj[args.config]["bytes_per_kmer_measured"] * roof["kmers_per_launch"]
                roof["traffic_source"] = TRAFFIC_FILE + " (offline --pmc passes, scaled)"
        except Exception:
            roof["traffic"] = None
        workload = cfg["desc"]
        if world > 1 and args.config == "c2" and n_reads == SHARD_READS_MULTI:
            workload = (f"NtHash k=31 canonical, 1 hash/k-mer, {world} x 125M x 150bp shards of the 1B-read job "
                        f"(BASELINE config 5)")
        elif n_reads != cfg["reads"]:
            workload = cfg["desc"] + f" [REDUCED to {n_reads} reads/GPU]"
        res = {
            "metric": "k-mers hashed/sec (canonical, k=%d, %dbp reads)" % (k, L),
            "value": total_kmers / dt,
            "unit": "kmers/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": workload, "reads_per_gpu": n_reads, "read_len": L, "k": k,
                       "hashes_per_kmer": wl.per, "launches_per_step": wl.n_chunks, "input": "ASCII, device-resident",
                       "parallelism": "reads sharded by rank, no data-path collective",
                       "placement": {"allocator": ("nthip_malloc_probed, %d candidates" % PLACE_CANDIDATES) if PLACE_CANDIDATES > 1
                                     else "nthip_malloc (hipMalloc), one allocation per buffer",
                                     "buffers": placement}},
            "roofline": roof,
            "verify": verify,
            "verified_vs_oracle": bool(all_ok),
            "verified_full_stream_all_ranks": bool(checked_full),
            "per_rank_kmers_per_s": per_rank,
            "dist": {"process_group": ("gloo (ranks share one GPU: test mode)" if share else "nccl (RCCL)") if use_dist else None,
                     "collectives": "barrier, all_reduce(MAX) of the step time, all_gather of per-rank rate / verdicts"
                                    if use_dist else None,
                     "forced_at_world_1": bool(use_dist and world == 1)},
        }
        if dist_consumer is not None:
            res["dist_consumer"] = dist_consumer
    # ---- N = 1 extras: measured ceiling, the other single-GPU configs, the CPU beside it ----------------------
    if world == 1:
        if not args.no_peak:
            try:
                res["roofline"]["peak_measured_pending"] = True   # (measured after the other lines: three 32 GiB allocations)
            except Exception as e:
                res["roofline"]["peak_measured"] = None
                res["roofline"]["peak_measured_error"] = str(e)
        if not args.no_secondary and args.config == "c2" and not args.reads:
            sec = {}
            for name in ("c2_packed", "c3", "c4", "ref", "var", "var_slots", "c2_dirty", "c2_dirty_slots"):
                try:
                    c2 = dict(CONFIGS[name])
                    w2 = Workload(torch, ctx, dev, name, c2, c2["reads"], 0)
                    w2.step(False)
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    km = 0
                    for _ in range(3):
                        km += w2.step(True)
                    torch.cuda.synchronize(dev)
                    d2 = time.perf_counter() - t0
                    r2 = w2.roofline()
                    v2 = w2.verify()
                    sec[name] = {"workload": c2["desc"], "value": km / d2, "unit": "kmers/s", "steps": 3,
                                 "ms_per_step": d2 / 3 * 1e3, "launches_per_step": w2.n_chunks,
                                 "kernel": r2["kernel"], "kernel_ms_per_step": r2["kernel_avg_ms"] * w2.n_chunks,
                                 "bytes_per_kmer": r2["bytes_per_kmer"], "achieved_GBps": r2["achieved"],
                                 "frac": r2["frac"], "verify_ok": v2["ok"], "spot_vs_oracle": v2["spot_vs_oracle"],
                                 "sum": v2["sum"], "xor": v2["xor"],
                                 "placement_fill_GBps": [b["fill_GBps"] for b in w2.placement]}
                    if w2.pack:
                        sec[name]["pack"] = w2.pack
                    w2.free()
                except Exception as e:
                    sec[name] = {"error": str(e)}
            res["secondary"] = sec
            try:
                res["consumers"] = consumers(torch, ctx, dev)
            except Exception as e:
                res["consumers"] = {"error": str(e)}
        if res["roofline"].pop("peak_measured_pending", False):
            try:
                pk = measured_peak(torch, ctx, dev, own_fill)
                res["roofline"]["peak_measured"] = pk["best_GBps"]
                res["roofline"]["frac_of_measured"] = res["roofline"]["achieved"] / pk["best_GBps"]
                res["roofline"]["frac_of_measured_ok"] = bool(res["roofline"]["frac_of_measured"] <= 1.02)
                res["roofline"]["peak_measured_detail"] = pk
            except Exception as e:
                res["roofline"]["peak_measured"] = None
                res["roofline"]["peak_measured_error"] = str(e)
        if not args.no_cpu_baseline:
            try:
                sample = args.cpu_sample_reads or (12_000_000 if cfg["seeds"] is None and cfg["m"] == 1 else
                                                   6_000_000 if cfg["seeds"] is None else 600_000)
                sample = min(sample, n_reads)
                res["cpu_baseline"] = cpu_baseline(cfg, sample)
            except Exception as e:
                res["cpu_baseline"] = {"value": None, "error": str(e)}
    if rank == 0:
        # what the driver keeps of a line is its LAST 2000 characters: the numbers of every part of the bench once more, short, at
        # the very end (G k-mers/s, roofline fraction; the words of every entry: DESIGN.md "bench line legend")
        def g(v):
            return None if v is None else round(v / 1e9, 1)

        def f3(v):
            return None if v is None else round(v, 3)
        summ = {"c2": [g(res["value"]), f3(res["roofline"].get("frac")), f3(res["roofline"].get("frac_plain_alloc"))]}
        if isinstance(res.get("secondary"), dict):
            # [G k-mers/s of the whole call, the hash KERNEL's roofline fraction, the whole CALL's (value x bytes per k-mer / peak)]
            summ["sec"] = {n_: ([g(v.get("value")), f3(v.get("frac")),
                                 f3(None if v.get("value") is None or not v.get("bytes_per_kmer") else v["value"] * v["bytes_per_kmer"] / 1e9 / HBM_PEAK_GBPS)]
                                if "error" not in v else "error") for n_, v in res["secondary"].items()}
        if isinstance(res.get("consumers"), dict) and "error" not in res["consumers"]:
            summ["cons"] = {n_: [g(v.get("value")), f3(v.get("roofline", {}).get("frac")), v.get("ok")]
                            for n_, v in res["consumers"].items() if isinstance(v, dict)}
            q = res["consumers"].get("bloom_query_4GiB", {})
            summ["query"] = {"binned_G": g(q.get("value")), "direct_G": g(q.get("direct_kernel", {}).get("value")),
                             "list_traffic_GBps": f3(q.get("roofline", {}).get("list_traffic", {}).get("GBps")),
                             "m3_x_m1": f3(res["consumers"].get("bloom_query_4GiB_m3", {}).get("x_m1")),
                             "m3_x_m1_half_hit": f3(res["consumers"].get("bloom_query_4GiB_m3", {}).get("half_hit", {}).get("x_m1")),
                             "seed_query_G": g(res["consumers"].get("seed_bloom_insert_c4_seeds", {}).get("query", {}).get("value")),
                             "seed_first_call_ms": [f3(res["consumers"].get("seed_bloom_insert_c4_seeds", {}).get("first_call_ms")),
                                                    f3(res["consumers"].get("seed_bloom_insert_c4_seeds", {}).get("query", {}).get("first_call_ms"))]}
            summ["legend"] = "cons: [G/s whole call, algorithmic frac, self-consistency check (parity is the GPU tests')]; sec: [G/s, kernel frac, call frac]"
        if isinstance(res.get("cpu_baseline"), dict):
            cb = res["cpu_baseline"]
            summ["cpu"] = {"kind": cb.get("kind"), "one_core_M": None if cb.get("value") is None else round(cb["value"] / 1e6, 1),
                           "openmp_M": None if not cb.get("openmp") else round(cb["openmp"]["value"] / 1e6, 1),
                           "threads": None if not cb.get("openmp") else cb["openmp"].get("cores")}
        res["summary"] = summ
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
