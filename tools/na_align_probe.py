#!/usr/bin/env python3
"""What does the N-aware pass lose to tiles that start off a line of the stream?  20 M x 150 bp, ONE N: in the batch's last
read (every tile before it starts on a line: deficit 0), in its first read (every tile behind it is shifted by 31 k-mers),
or shifted by 16 k-mers (two N's 15 apart... a multiple of a line again).  Kernel of record = the N-aware hash pass.

    python tools/na_align_probe.py [reads=20000000]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nthash_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
L, k = 150, 31
nwin = L - k + 1
ctx = nthash_amd.Context(0)
ctx.set_profiling(True)
d_in = ctx.malloc(n * L); d_out = ctx.malloc(n * nwin * 8)
N = np.frombuffer(b"N", np.uint8)
for name, where in (("N in the last read", [(n - 1) * L + 75]), ("N in the first read", [75]),
                    ("two N's in the first read: 32 k-mers lost (two lines)", [40, 41]),
                    ("N in the first read at base 15: 16 k-mers lost (a line)", [15])):
    ctx.synth_reads_ptr(d_in, 0, n, L, 42)
    for w in where:
        ctx.h2d(d_in + w, N)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        tot = ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, 1, d_out, n * nwin)
        ts.append(time.perf_counter() - t0)
    ms, kn = ctx.last_kernel_ms()
    print(f"{name:62s} lost {n*nwin-tot:3d}: whole call {min(ts)*1e3:6.2f} ms, {kn} {ms:6.3f} ms", flush=True)
