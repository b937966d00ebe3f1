#!/usr/bin/env python3
"""What do the slow allocations have in common?  Per fresh pair of buffers: the headline kernel, a pure fill of the output
buffer, a pure read of ... (copy of the first GiBs), the no-hash stream.   python tools/alloc_classes.py [reads] [allocs]"""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nthash_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
L, k = 150, 31
ctx = nthash_amd.Context(0)
ctx.set_profiling(True)
in_b, out_b = n * L, n * 120 * 8
hold = []
for r in range(reps):
    d_in = ctx.malloc(in_b); d_out = ctx.malloc(out_b)
    ctx.synth_reads_ptr(d_in, 0, n, L, 42)
    ts = []
    for _ in range(6):
        ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, 1, d_out, n * 120)
        ts.append(ctx.last_kernel_ms()[0])
    t = statistics.median(ts[1:])
    fill = ctx.fill_bench_ptr(d_out, out_b // 2, 3)          # write-only over the first half of the output buffer
    fill2 = ctx.fill_bench_ptr(d_out + out_b // 2, out_b // 2, 3)
    cp = ctx.copy_bench_ptr(d_out, d_in, in_b, 3)             # read the input buffer (and write as much)
    print(f"alloc {r}: in {d_in:#x} out {d_out:#x}  kernel {t:7.3f} ms {n*120/t/1e6:5.0f} G   fill {out_b/2/fill/1e6:5.0f} / "
          f"{out_b/2/fill2/1e6:5.0f} GB/s   copy(in->out) {2*in_b/cp/1e6:5.0f} GB/s", flush=True)
    hold.append((d_in, d_out))
    if len(hold) > 1:
        for p in hold.pop(0):
            ctx.free(p)
