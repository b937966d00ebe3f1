#!/usr/bin/env python3
"""In-process A/B of a tune knob on the Bloom / counting-sketch insert (20 M x 150 bp device-resident reads, fresh 4 GiB
filter / 1 Gi counters): two contexts on the same buffers, calls interleaved, whole-call minimum and median of each.

    python tools/bloom_ab.py NTHIP_TUNE_BLOOM_L2_MAJOR=2 [reads=20000000] [reps=7]
"""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nthash_amd
knob, val = sys.argv[1].split("=")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 7
L, k, m = 150, 31, 1
a = nthash_amd.Context(0)
os.environ[knob] = val
b = nthash_amd.Context(0)
os.environ.pop(knob)
d_in = a.malloc(n * L)
a.synth_reads_ptr(d_in, 0, n, L, 42)
for name, size, call in (("bloom 4 GiB", 1 << 32, lambda c, d: c.bloom_insert_ptr(d_in, n, L, 0, k, m, d, 1 << 35)),
                         ("count 1 Gi", 1 << 30, lambda c, d: c.count_insert_ptr(d_in, n, L, 0, k, m, d, 1 << 30))):
    d_t = a.malloc(size)
    ts = {"default": [], sys.argv[1]: []}
    for it in range(reps + 1):
        for label, c in (("default", a), (sys.argv[1], b)):
            a.memset(d_t, 0, size)
            t0 = time.perf_counter(); tot = call(c, d_t); dt = time.perf_counter() - t0
            if it:
                ts[label].append(dt)
    for label, t in ts.items():
        print(f"{name} {label:32s} min {min(t)*1e3:7.3f} ms  median {statistics.median(t)*1e3:7.3f} ms  {tot/min(t)/1e9:6.1f} G k-mers/s", flush=True)
    a.free(d_t)
