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


# ---- the merge of the consumers' tables: the one inter-GPU step (nthash_amd/sharding.py, capi_multi_sink.hip) ----------
def _tables(op, world, nbytes, seed):
    rng = np.random.default_rng(seed)
    hi = 120 if op == "add_sat_u8" else 256       # (sums that saturate and sums that do not)
    return [rng.integers(0, hi, nbytes, dtype=np.uint8) for _ in range(world)]


def _reduce(op, tabs):
    if op == "or":
        return np.bitwise_or.reduce(np.stack(tabs), axis=0)
    if op == "add_sat_u8":
        return np.minimum(np.stack(tabs).astype(np.int64).sum(axis=0), 255).astype(np.uint8)
    return np.stack([t.view(np.uint64) for t in tabs]).min(axis=0).view(np.uint8)


@pytest.mark.parametrize("op", ["or", "add_sat_u8", "min_u64"])
@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_ring_merge_schedule_in_one_process(op, world):
    """the ring's schedule and the three fold operators, ranks as threads exchanging through queues: every rank ends with
    the element-wise reduction of all tables (all-gather), or with its finished segment (reduce-scatter alone)"""
    import queue
    import threading

    import torch

    from nthash_amd import sharding as sh
    for nbytes in (16 * world, 48, 4096, 16 * 1031):
        tabs = _tables(op, world, nbytes, 5 * world + nbytes)
        want = _reduce(op, tabs)
        for allgather in (True, False):
            qs = {(a, b): queue.Queue() for a in range(world) for b in range(world)}
            outs = [None] * world

            def run(r):
                def exchange(out, dst, inp, src):
                    qs[(r, dst)].put(out.clone())
                    inp.copy_(qs[(src, r)].get(timeout=60))
                outs[r] = sh.ring_merge(torch.from_numpy(tabs[r].copy()), op, r, world, exchange, allgather).numpy()
            th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
            [t.start() for t in th]
            [t.join() for t in th]
            seg = sh.ring_segments(nbytes, world)
            for r in range(world):
                if allgather:
                    assert (outs[r] == want).all()
                else:
                    i = (r + 1) % world
                    assert (outs[r][seg[i]:seg[i + 1]] == want[seg[i]:seg[i + 1]]).all()


def _merge_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist

    from nthash_amd import sharding as sh
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    res = {}
    for op in ("or", "add_sat_u8", "min_u64"):
        tabs = _tables(op, world, 1 << 16, 99)
        t = torch.from_numpy(tabs[rank].copy())
        sh.ring_merge_dist(t, op)
        res[op] = bool((t.numpy() == _reduce(op, tabs)).all())
    dist.barrier()
    q.put((rank, res))
    dist.destroy_process_group()


def test_two_rank_gloo_merge_of_the_consumers_tables():
    """world size 2, one process per rank as on the GPUs: filter OR, saturating counter add and signature minimum travel as
    ring segments over send / recv and are folded locally -- both ranks end with the table one device would have built"""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_merge_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(all(r[1].values()) for r in res), res
