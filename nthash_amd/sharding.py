"""Read-level sharding across GPUs: every read is hashed
independently, so rank g of G owns one contiguous range of the global read set and
nothing is exchanged on the data path.  bench.py runs "weak" scaling: every rank
brings reads_per_gpu reads of its own; `shard_of` is the strong-scaling split
used when a fixed read set is divided.
"""


def weak_shard(rank, reads_per_gpu):
    """(first_read, n_reads) of `rank` when every rank owns reads_per_gpu reads."""
    return rank * reads_per_gpu, reads_per_gpu


def shard_of(rank, world, n_reads):
    """(first_read, n_reads) of `rank` for a fixed set of n_reads: contiguous,
    disjoint, covering, sizes differing by at most one."""
    base, extra = divmod(n_reads, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


# ---- the merge of the consumers' tables (SURVEY.md 5 "distributed", 8f-1): the one inter-GPU step -----------------------
# Every rank consumes its shard into its own Bloom filter / counting sketch / signature; only those RESULTS cross xGMI.
# The fold is element-wise with an operator RCCL's collectives do not have for these types, so it is a ring
# reduce-scatter made of point-to-point transfers + a local fold, then a ring all-gather -- the schedule
# nthash_amd/csrc/capi_multi_sink.hip runs between the devices of ONE process (hipMemcpyPeerAsync + merge_kernel), stated
# here for one process per GPU (torch.distributed send / recv: RCCL on GPUs, gloo in the CPU tests).

def ring_segments(nbytes, world):
    """byte offsets [o_0 = 0, ..., o_world = nbytes] of the ring's segments, cut at multiples of 16 bytes"""
    assert nbytes % 16 == 0
    return [(nbytes // 16 * s // world) * 16 for s in range(world + 1)]


def ring_reduce_scatter_steps(rank, world):
    """[(left, segment)] per step: in step s `rank` receives `segment` of its left neighbour's table and folds it into its
    own; after world - 1 steps it holds the finished segment (rank + 1) % world"""
    left = (rank - 1) % world
    return [(left, (left - s) % world) for s in range(world - 1)]


def ring_all_gather_steps(rank, world):
    """[(left, segment)] per step: `rank` receives the finished `segment` from its left neighbour"""
    left = (rank - 1) % world
    return [(left, (left + 1 - s) % world) for s in range(world - 1)]


def fold(op, mine, theirs):
    """mine (op)= theirs on uint8 torch tensors of equal length (a multiple of 8 for 'min_u64'), in place.
    'or': Bloom filters; 'add_sat_u8': one-byte counters saturating at 255; 'min_u64': little-endian 64-bit entries"""
    import torch
    if op == "or":
        mine.bitwise_or_(theirs)
    elif op == "add_sat_u8":
        mine.copy_((mine.to(torch.int16) + theirs.to(torch.int16)).clamp_(max=255).to(torch.uint8))
    elif op == "min_u64":
        a, b = mine.view(torch.int64), theirs.view(torch.int64)
        flip = torch.tensor(-0x8000000000000000, dtype=torch.int64, device=mine.device)   # unsigned order through signed min
        a.copy_(torch.minimum(a ^ flip, b ^ flip) ^ flip)
    else:
        raise ValueError(op)
    return mine


def ring_merge(table, op, rank, world, exchange, allgather=True):
    """Merge `table` (1-D uint8 torch tensor, the same size on every rank) over the ring.  exchange(out, dst, inp, src) sends
    the tensor `out` to rank dst and fills `inp` from rank src (both neighbours, one call per step: a backend that must
    post the pair together can).  With allgather every rank ends with the merged table, else only segment
    (rank + 1) % world of it is final."""
    import torch
    if world == 1:
        return table
    seg = ring_segments(table.numel(), world)
    right = (rank + 1) % world
    tmp = torch.empty(max(seg[i + 1] - seg[i] for i in range(world)), dtype=torch.uint8, device=table.device)
    for s, (left, idx) in enumerate(ring_reduce_scatter_steps(rank, world)):
        out_idx = (rank - s) % world                           # what the right neighbour folds in this step
        part = tmp[: seg[idx + 1] - seg[idx]]
        exchange(table[seg[out_idx]:seg[out_idx + 1]], right, part, left)
        fold(op, table[seg[idx]:seg[idx + 1]], part)
    if allgather:
        for s, (left, idx) in enumerate(ring_all_gather_steps(rank, world)):
            out_idx = (rank + 1 - s) % world
            exchange(table[seg[out_idx]:seg[out_idx + 1]], right, table[seg[idx]:seg[idx + 1]], left)
    return table


def ring_merge_dist(table, op, group=None, allgather=True):
    """ring_merge over a torch.distributed process group (one process per GPU; "nccl" = RCCL over xGMI, gloo on the CPU):
    a step's send and receive are posted together (batch_isend_irecv), as RCCL's point-to-point calls ask"""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)

    def peer(r):   # P2POp wants GLOBAL ranks; ring_merge counts inside the group
        return dist.get_global_rank(group, r) if group is not None else r

    def exchange(out, dst, inp, src):
        ops = []
        if out.numel():
            ops.append(dist.P2POp(dist.isend, out, peer(dst), group))
        if inp.numel():
            ops.append(dist.P2POp(dist.irecv, inp, peer(src), group))
        for req in (dist.batch_isend_irecv(ops) if ops else []):
            req.wait()
    return ring_merge(table, op, rank, world, exchange, allgather)
