#!/usr/bin/env python3
"""bench.py -- k-mers hashed/sec of the ntHash hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config c2|c3|c4] [--reads R]

One "step" = one pass of the hot path (nthip_kmer_hash / nthip_seed_hash through
the C-ABI) over the rank's device-resident batch of synthetic reads.  Default
workload = BASELINE.json configs[1]: NtHash k=31 canonical, 1 hash/k-mer,
100M x 150 bp reads on one MI355X (15 GB in, 96 GB of hashes out, both resident
in HBM).  With N > 1 (launched by torch.distributed.run, one rank per GPU) every
rank hashes its own shard of N*R reads -- reads are independent, so there is no
data-path collective ("weak" scaling); torch.distributed is used only for the
barrier and the max-over-ranks time.

Rank 0 prints ONE JSON line: BASELINE.json's metric plus
  roofline      algorithmic HBM bytes per launch / HIP-event duration of the
                dominant kernel, against the 8 TB/s spec peak,
  cpu_baseline  the reference CPU library (oracle/_ref, kind "reference") or the
                C restatement (kind "port") timed on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED_A = "1010101010101010101010101010101"
SEED_B = "1101101101101101011011011011011"
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy rate

CONFIGS = {
    # name: (description, read_len, k, hashes/k-mer kind, default reads per GPU)
    "c2": dict(desc="NtHash k=31 canonical, 1 hash/k-mer, 100M x 150bp", L=150, k=31, m=1, seeds=None,
               reads=100_000_000),
    "c3": dict(desc="NtHash k=31, m=4 hashes/k-mer (multi-hash), 100M x 150bp", L=150, k=31, m=4,
               seeds=None, reads=100_000_000),
    "c4": dict(desc="SeedNtHash 2 spaced seeds k=31, m=3, 50M x 250bp", L=250, k=31, m=3,
               seeds=[SEED_A, SEED_B], reads=50_000_000),
    # the reference's own harness shape (examples/benchmark.cpp:9-39: 100 bp reads, NtHash(seq, 3, 64)),
    # scaled from its 1 M reads to a batch that fills the GPU
    "ref": dict(desc="NtHash k=64, m=3, 100bp reads (examples/benchmark.cpp shape), 100M reads", L=100, k=64,
                m=3, seeds=None, reads=100_000_000),
}


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (default: the config's size)")
    ap.add_argument("--chunk-reads", type=int, default=0,
                    help="reads per launch (outputs of c3/c4 exceed HBM: a ring buffer is reused)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-reads", type=int, default=0)
    return ap.parse_args()


def cpu_baseline(cfg, sample_reads):
    """Time the reference CPU library on a bounded sample of the same workload."""
    import numpy as np  # noqa: F401

    from oracle.pyoracle import Oracle, Reference
    impl = Reference() if Reference.available() else Oracle()
    L, k, m = cfg["L"], cfg["k"], cfg["m"]
    data = impl.synth_reads(0, sample_reads, L, 42)
    t0 = time.perf_counter()
    if cfg["seeds"] is None:
        _acc, nk = impl.bench_kmer(data, sample_reads, L, k, m, threads=1)
    elif impl.kind == "reference":
        _acc, nk = impl.bench_seed(data, sample_reads, L, cfg["seeds"], k, m, threads=1)
    else:
        offs = np.arange(sample_reads + 1, dtype=np.uint64) * L
        nk = impl.seed_batch(data, offs, cfg["seeds"], k, m, want_pos=False)["total"]
    t1 = time.perf_counter() - t0
    out = {"value": nk / t1, "unit": "kmers/s", "cores": 1, "kind": impl.kind,
           "sample": f"{sample_reads} x {L}bp synthetic reads (same generator, seed 42), "
                     f"{nk} k-mers in {t1:.2f}s, iterator per read, every hash consumed",
           "host_cpus": os.cpu_count(), "cpu_model": _cpu_model()}
    if impl.kind == "reference":
        # the reference has no threaded path; this is OUR OpenMP parallel-for over reads
        nt = impl.max_threads()
        if nt > 1:
            t0 = time.perf_counter()
            if cfg["seeds"] is None:
                _acc, nk2 = impl.bench_kmer(data, sample_reads, L, k, m, threads=nt)
            else:
                _acc, nk2 = impl.bench_seed(data, sample_reads, L, cfg["seeds"], k, m, threads=nt)
            t2 = time.perf_counter() - t0
            out["openmp"] = {"value": nk2 / t2, "unit": "kmers/s", "cores": nt,
                             "note": "OpenMP parallel-for over reads added by the harness"}
    return out


def main():
    args = parse()
    cfg = dict(CONFIGS[args.config])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist

    import nthash_amd
    from nthash_amd.sharding import weak_shard

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # NTHASH_BENCH_SHARE_GPU=1 (testing only): all ranks on GPU 0 with a gloo group, so that the
    # multi-rank control flow can be exercised on a 1-GPU box; real runs use one GPU per rank + RCCL
    share = os.environ.get("NTHASH_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    # a launcher that pins one GPU per rank (HIP_VISIBLE_DEVICES) leaves every rank with device 0 only
    if local_rank >= torch.cuda.device_count():
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    L, k, m = cfg["L"], cfg["k"], cfg["m"]
    nwin = L - k + 1
    n_reads = args.reads or cfg["reads"]
    per = m if cfg["seeds"] is None else len(cfg["seeds"]) * m
    # launches per step: outputs larger than ~100 GB are produced chunk by chunk into one buffer
    free_b, _tot_b = torch.cuda.mem_get_info(dev)
    out_bytes_per_read = nwin * per * 8
    budget = int(free_b * 0.85) - n_reads * L
    chunk = args.chunk_reads or min(n_reads, max(1, budget // out_bytes_per_read))
    chunk = min(chunk, n_reads)
    if chunk < n_reads:  # keep chunks a multiple of the kernel's 256-read tile
        chunk = max(256, chunk // 256 * 256)
    n_chunks = (n_reads + chunk - 1) // chunk

    ctx = nthash_amd.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    ctx.set_profiling(True)
    d_in = torch.empty(n_reads * L, dtype=torch.uint8, device=dev)
    d_out = torch.empty(chunk * nwin * per, dtype=torch.int64, device=dev)
    # this rank's shard of the global read set [rank*n_reads, (rank+1)*n_reads)
    first_read, _ = weak_shard(rank, n_reads)
    ctx.synth_reads_ptr(d_in.data_ptr(), first_read, n_reads, L, 42)
    seeds = nthash_amd.Seeds(ctx, cfg["seeds"], k) if cfg["seeds"] else None
    torch.cuda.synchronize(dev)

    kernel_ms = []

    def step(record):
        done = 0
        for c in range(n_chunks):
            r0 = c * chunk
            nr = min(chunk, n_reads - r0)
            if seeds is None:
                tot = ctx.kmer_hash_ptr(d_in.data_ptr() + r0 * L, 0, nr, L, 0, k, m, d_out.data_ptr(),
                                        chunk * nwin)
            else:
                tot = ctx.seed_hash_ptr(d_in.data_ptr() + r0 * L, 0, nr, L, 0, seeds, m,
                                        d_out.data_ptr(), chunk * nwin)
            if record:
                ms, name = ctx.last_kernel_ms()
                kernel_ms.append((ms, name, tot))
            done += tot
        return done

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    kmers = 0
    for _ in range(args.steps):
        kmers += step(True)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert kmers == args.steps * n_reads * nwin, (kmers, args.steps * n_reads * nwin)

    # ---- post-run verification (outside the timed region) -------------------
    verified = None
    try:
        import numpy as np

        from oracle.pyoracle import Oracle
        orc = Oracle()
        last_r0 = (n_chunks - 1) * chunk  # d_out holds the last chunk
        nv = min(2000, n_reads - last_r0)
        host = d_out[: nv * nwin * per].cpu().numpy().view(np.uint64).reshape(-1, per)
        data = orc.synth_reads(first_read + last_r0, nv, L, 42)
        offs = np.arange(nv + 1, dtype=np.uint64) * L
        if seeds is None:
            want = orc.kmer_batch(data, offs, k, m, want_pos=False)["hashes"]
        else:
            want = orc.seed_batch(data, offs, cfg["seeds"], k, m, want_pos=False)["hashes"]
        verified = bool((host == want).all())
    except Exception as e:  # the oracle is a checker, not a dependency of the measurement
        verified = f"not checked: {e}"

    if rank == 0:
        total_kmers = kmers * world
        b_per_kmer = 8.0 * per + L / nwin  # 8*H + b_in*L/(L-k+1), ASCII input
        ms_list = [x[0] for x in kernel_ms]
        avg_ms = sum(ms_list) / len(ms_list)
        kmers_per_launch = sum(x[2] for x in kernel_ms) / len(kernel_ms)
        achieved = kmers_per_launch * b_per_kmer / (avg_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE x2
        # gfx950 correction + WRITE_SIZE, collected in separate rocprofv3 --pmc runs on the same
        # kernel); counters cannot be read inside this process, so null when no summary exists
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            if args.config in tj:
                traffic = tj[args.config]["bytes_per_kmer_measured"] * kmers_per_launch
        except Exception:
            traffic = None
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
            "config": {"workload": cfg["desc"] if n_reads == cfg["reads"] else
                       cfg["desc"] + f" [REDUCED to {n_reads} reads/GPU]",
                       "reads_per_gpu": n_reads, "read_len": L, "k": k, "hashes_per_kmer": per,
                       "launches_per_step": n_chunks, "input": "ASCII, device-resident",
                       "parallelism": "reads sharded by rank, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": kernel_ms[0][1], "kernel_avg_ms": avg_ms,
                         "bytes_per_kmer": b_per_kmer, "kmers_per_launch": kmers_per_launch},
            "verified_vs_oracle": verified,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                sample = args.cpu_sample_reads or (12_000_000 if cfg["seeds"] is None and m == 1 else
                                                   6_000_000 if cfg["seeds"] is None else 600_000)
                sample = min(sample, n_reads)
                res["cpu_baseline"] = cpu_baseline(cfg, sample)
            except Exception as e:
                res["cpu_baseline"] = {"value": None, "error": str(e)}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
