#!/usr/bin/env python3
"""Does a buffer mapped from physical pieces (NTHIP_TUNE_MALLOC_PIECES) keep what is written to it when buffers of the
same sizes were mapped, written and released just before?  (round 5: 125 M reads synthesised into such a buffer came back
with non-bases on the second use.)   python tools/vmm_reuse_check.py [pieces MiB] [cycles]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pieces = sys.argv[1] if len(sys.argv) > 1 else "8"
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 4
os.environ["NTHIP_TUNE_MALLOC_PIECES"] = pieces
import nthash_amd
ctx = nthash_amd.Context(0)
n, L, k = 125_000_000, 150, 31
nwin = L - k + 1
bad = 0
for cyc in range(cycles):
    d_in, d_out = ctx.malloc(n * L), ctx.malloc(n * nwin * 8)
    ctx.synth_reads_ptr(d_in, cyc * n, n, L, 42)
    tot = ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, 1, d_out, n * nwin)
    s = ctx.checksum_ptr(d_out, tot)
    print(f"   in {d_in:#x} out {d_out:#x}")
    print(f"cycle {cyc}: k-mers {tot} of {n * nwin}  {'OK' if tot == n * nwin else 'NON-BASES SEEN'}  checksum {s}", flush=True)
    bad += tot != n * nwin
    ctx.free(d_in)
    ctx.free(d_out)
print("bad cycles:", bad)
