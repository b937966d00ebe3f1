#!/usr/bin/env python3
"""Headline kernel time against the placement of its two buffers inside ONE big allocation (one process, one context).
    python tools/placement_probe2.py [reads]"""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nthash_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
L, k = 150, 31
ctx = nthash_amd.Context(0)
in_b, out_b = n * L, n * 120 * 8
GB = 1 << 30
region = ctx.malloc(in_b + out_b + 6 * GB)
print("region at", hex(region), "in", in_b / GB, "GiB out", out_b / GB, "GiB", flush=True)
ctx.set_profiling(True)


def run(off_in, off_out):
    d_in, d_out = region + off_in, region + off_out
    ctx.synth_reads_ptr(d_in, 0, n, L, 42)
    ts = []
    for _ in range(6):
        ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, 1, d_out, n * 120)
        ts.append(ctx.last_kernel_ms()[0])
    return statistics.median(ts[1:])


base_out = (in_b + 2 * GB) // (2 << 20) * (2 << 20)
for name, oi, oo in [("in 0, out 2MiB-aligned", 0, base_out)] + \
        [(f"out + {d}", 0, base_out + d) for d in (256, 4096, 65536, 1 << 20, (1 << 20) + 4096, 2 << 20, 16 << 20, 128 << 20, 1 << 30, (1 << 30) + (3 << 20))] + \
        [(f"in + {d}", d, base_out) for d in (16, 256, 4096, 65536, 1 << 20, 2 << 20, 16 << 20, 128 << 20)] + \
        [("again: in 0, out aligned", 0, base_out)]:
    t = run(oi, oo)
    print(f"{name:32s} {t:7.3f} ms  {n*120/t/1e6:6.1f} G k-mers/s", flush=True)
