#!/usr/bin/env python3
"""Is the per-allocation spread of the kernel time stable (placement) or temporal (clocks)?"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nthash_amd
n = 20_000_000
L, k, m = 150, 31, 1
nwin = L - k + 1
ctx = nthash_amd.Context(0)
ctx.set_profiling(True)
def meas(a, b, reps=5):
    ts = []
    for _ in range(reps):
        ctx.kmer_hash_ptr(a, 0, n, L, 0, k, m, b, n * nwin); ts.append(ctx.last_kernel_ms()[0])
    return statistics.median(ts[1:])
def rnd(x, g): return (x + g - 1) // g * g
pairs = []
for i, g in enumerate((1, 1, 2 << 20, 2 << 20, 1 << 30, 1 << 30, 1, 1)):
    a = ctx.malloc(rnd(n * L, g)); b = ctx.malloc(rnd(n * nwin * 8, g))
    ctx.synth_reads_ptr(a, 0, n, L, 42)
    pairs.append((a, b, g))
for rnd_i in range(3):
    print(" ".join(f"{meas(a, b):.3f}" for a, b, g in pairs), flush=True)
# cross pairs: input of pair i with output of pair j
print("in0/outJ:", " ".join(f"{meas(pairs[0][0], pairs[j][1]):.3f}" for j in range(len(pairs))))
print("inJ/out0:", " ".join(f"{meas(pairs[j][0], pairs[0][1]):.3f}" for j in range(len(pairs))))
