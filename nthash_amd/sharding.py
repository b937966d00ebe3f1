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
