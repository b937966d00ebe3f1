#!/usr/bin/env python3
"""In-process A/B of a tune knob on a dirty fixed-length batch (device-resident, one N per ~1000 reads): two contexts on
the same buffers, calls interleaved, whole-call minimum and median of each.

    python tools/dirty_ab.py NTHIP_TUNE_NO_TILES_SCAN [reads=20000000] [reps=15]
"""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nthash_amd
knob = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 15
L, k, m = (int(x) for x in os.environ.get("DIRTY_SHAPE", "150,31,1").split(","))
nwin = L - k + 1
a = nthash_amd.Context(0)
os.environ[knob] = "1"
b = nthash_amd.Context(0)
os.environ.pop(knob)
d_in = a.malloc(n * L); d_out = a.malloc(n * nwin * m * 8)
a.synth_reads_ptr(d_in, 0, n, L, 42)
for i in np.arange(0, n * L, 1000 * L + 17, dtype=np.int64)[:20000]:
    a.h2d(d_in + int(i), np.frombuffer(b"N", np.uint8))
ts = {"default": [], knob + "=1": []}
for it in range(reps + 2):
    for name, c in (("default", a), (knob + "=1", b)):
        t0 = time.perf_counter()
        tot = c.kmer_hash_ptr(d_in, 0, n, L, 0, k, m, d_out, n * nwin)
        if it >= 2:
            ts[name].append(time.perf_counter() - t0)
for name, t in ts.items():
    print(f"{name:34s} min {min(t)*1e3:6.3f} ms  median {statistics.median(t)*1e3:6.3f} ms  {tot/min(t)/1e9:6.1f} G k-mers/s", flush=True)
