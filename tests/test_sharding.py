"""CPU tests of the multi-GPU path (no GPU needed): reads shard by contiguous
ranges with no data-path collective, so (1) the shard arithmetic must tile the
read set exactly, (2) hashing the shards separately and concatenating must give
the stream of the whole set (checked on the oracle), and (3) a 2-rank job
(gloo, world_size 2) laid out like `bench.py --gpus 2`: every rank hashes its own
shard, the only communication is barrier + max-over-ranks time + the gathered
counts and checksums, and each shard's checksum equals the real reference's."""
import os
import socket

import numpy as np
import pytest

from nthash_amd.sharding import shard_of, weak_shard


def test_shard_of_tiles_exactly():
    for n in (0, 1, 7, 100, 1000003):
        for world in (1, 2, 3, 4, 8):
            nxt = 0
            for r in range(world):
                first, cnt = shard_of(r, world, n)
                assert first == nxt and cnt >= 0
                nxt += cnt
            assert nxt == n
            sizes = [shard_of(r, world, n)[1] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_weak_shards_are_disjoint_and_contiguous():
    per = 12345
    assert [weak_shard(r, per) for r in range(4)] == [(0, per), (per, per), (2 * per, per), (3 * per, per)]


def test_sharded_stream_equals_whole_stream(oracle):
    """8-way split hashed shard by shard == the whole batch (the parity argument
    for multi-GPU runs: no cross-shard state exists)"""
    n, L, k, m = 1000, 150, 31, 2
    data = oracle.synth_reads(0, n, L, 42)
    offs = np.arange(n + 1, dtype=np.uint64) * L
    whole = oracle.kmer_batch(data, offs, k, m, want_pos=False)["hashes"]
    parts = []
    for r in range(8):
        first, cnt = shard_of(r, 8, n)
        # every rank regenerates its own reads from the counter-based generator
        d = oracle.synth_reads(first, cnt, L, 42)
        assert (d == data[first * L:(first + cnt) * L]).all()
        o = np.arange(cnt + 1, dtype=np.uint64) * L
        parts.append(oracle.kmer_batch(d, o, k, m, want_pos=False)["hashes"])
    assert (np.concatenate(parts) == whole).all()
    s_all, x_all = oracle.checksum(whole)
    s = x = 0
    for p in parts:  # checksum of checksums, as bench-scale verification does
        ps, px = oracle.checksum(p)
        s = (s + ps) & (2**64 - 1)
        x ^= px
    assert (s, x) == (s_all, x_all)


def _worker(rank, world, port, q):
    """One rank of a 2-rank job laid out like `bench.py --gpus 2`: rank r owns the reads starting at r * 125 M
    (BASELINE config 5's shard), hashes the head of its shard -- here through the oracle, on the GPU box through
    the C-ABI (tests/test_gpu_bench.py) -- and the ranks exchange only times, counts and checksums."""
    import torch
    import torch.distributed as dist

    from oracle.pyoracle import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    orc = Oracle()
    n, L, k = 20_000, 150, 31
    first, _ = weak_shard(rank, 125_000_000)
    data = orc.synth_reads(first, n, L, 42)
    offs = np.arange(n + 1, dtype=np.uint64) * L
    dist.barrier()
    r = orc.kmer_batch(data, offs, k, 1, want_pos=False)
    s, x = orc.checksum(r["hashes"])
    dist.barrier()
    t = torch.tensor([0.010 * (rank + 1)], dtype=torch.float64)  # this rank's elapsed time
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    kmers = torch.tensor([int(r["total"])], dtype=torch.int64)
    dist.all_reduce(kmers, op=dist.ReduceOp.SUM)
    # checksums travel as 32-bit halves (int64 tensors): every rank sees every shard's
    mine = torch.tensor([s >> 32, s & 0xFFFFFFFF, x >> 32, x & 0xFFFFFFFF], dtype=torch.int64)
    got = [torch.zeros(4, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(got, mine)
    sums = [(int(g[0]) << 32) | int(g[1]) for g in got]
    xors = [(int(g[2]) << 32) | int(g[3]) for g in got]
    dist.barrier()
    q.put((rank, first, int(r["total"]), float(t.item()), int(kmers.item()), sums, xors))
    dist.destroy_process_group()


def test_two_rank_gloo_shards_real_work():
    import torch.multiprocessing as mp

    from conftest import load_golden
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [0, 125_000_000]
    assert all(abs(r[3] - 0.020) < 1e-12 for r in res)      # max over ranks
    assert all(r[4] == 2 * 20_000 * 120 for r in res)        # whole-job k-mers
    # every rank's shard checksum == what the REAL reference produced for those reads (fixture), and both
    # ranks saw the same gathered values
    gold = {(e["workload"], e["first_read"], e["n_reads"]): e for e in load_golden("bench_checksums.json")}
    for r in res:
        assert r[5] == res[0][5] and r[6] == res[0][6]
    for rank in range(2):
        e = gold[("c2", rank * 125_000_000, 20_000)]
        assert format(res[0][5][rank], "016x") == e["sum"]
        assert format(res[0][6][rank], "016x") == e["xor"]
