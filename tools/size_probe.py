#!/usr/bin/env python3
"""Is the rate of the headline kernel a property of the BATCH size or of the ALLOCATION it runs in?
One process: batches of several sizes inside allocations of several sizes (plain hipMalloc each).
Answer (profiles/r02_notes.md section 25): of the allocation -- a 30 GiB batch runs at 553-557 G k-mers/s in allocations of 31,
32, 34, 36, 48, 50, 56 GiB and at 606-611 G in its own 30 GiB one or a 38 GiB one, reproducibly inside a process."""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nthash_amd
L, k = 150, 31
nwin = L - k + 1
ctx = nthash_amd.Context(0)
ctx.set_profiling(True)


def run(batch_gib, alloc_gib, reps=5):
    n = int((batch_gib * (1 << 30)) // (nwin * 8))
    na = int((alloc_gib * (1 << 30)) // (nwin * 8))
    d_in = ctx.malloc(na * L)
    d_out = ctx.malloc(na * nwin * 8)
    ctx.synth_reads_ptr(d_in, 0, n, L, 7)
    ts = []
    for _ in range(reps + 1):
        ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, 1, d_out, n * nwin)
        ts.append(ctx.last_kernel_ms()[0])
    ms = statistics.median(ts[1:])
    f_out = na * nwin * 8 / ctx.fill_bench_ptr(d_out, na * nwin * 8, 3) / 1e6  # (the call returns ms per fill)
    f_in = na * L / ctx.fill_bench_ptr(d_in, na * L, 3) / 1e6
    print(f"batch {batch_gib:5.1f} GiB in allocation {alloc_gib:6.2f} GiB: {ms:7.3f} ms  {n * nwin / ms / 1e6:6.1f} G k-mers/s   "
          f"fill out {f_out:6.0f} in {f_in:6.0f} GB/s", flush=True)
    ctx.free(d_in); ctx.free(d_out)


for (b, a) in [(32, 32), (30, 30), (30, 32), (30, 31), (30, 34), (30, 36), (30, 38), (30, 44), (30, 48), (30, 50), (30, 56),
               (16, 16), (16, 17), (16, 18), (16, 22), (16, 24), (16, 28), (8, 8), (8, 9), (8, 10), (8, 12), (8, 14), (8, 15)]:
    run(b, a)
