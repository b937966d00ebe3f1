#!/usr/bin/env python3
"""Does the headline kernel's time depend on the PROCESS?  Runs the C2 shape in this process: fresh context and buffers
several times in a row (REPS), prints the median kernel time of each.   python tools/process_variance.py [reads] [reps]"""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nthash_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L, k = 150, 31
out = []
for r in range(reps):
    ctx = nthash_amd.Context(0)
    pad = ctx.malloc((r * 7 + 1) << 20)      # shifts the following allocations
    d_in = ctx.malloc(n * L); d_out = ctx.malloc(n * 120 * 8)
    ctx.synth_reads_ptr(d_in, 0, n, L, 42)
    ctx.set_profiling(True)
    ts = []
    for _ in range(8):
        ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, 1, d_out, n * 120)
        ts.append(ctx.last_kernel_ms()[0])
    out.append(statistics.median(ts[2:]))
    ctx.free(d_in); ctx.free(d_out); ctx.free(pad); ctx.close()
print("pid", os.getpid(), " ".join(f"{t:.3f}" for t in out), "ms  ->", " ".join(f"{n*120/t/1e6:.0f}" for t in out), "G k-mers/s", flush=True)
