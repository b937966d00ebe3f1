#!/usr/bin/env python3
"""Within-process interleaved A/B timing of kernel variants (HIP-event kernel time).

    python tools/ablate.py [reads] [variant,variant,...] [rounds]
variants: runs (default kernel), gen / genC<n> (general run-split kernel, run length n), runs_vec (NTHIP_TUNE_NO_DWORD_TAIL=1), rows (flag 8), general (flag 4)
"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NTHIP_TUNE_NO_AUTOTUNE", "1")  # A/B hygiene: the same run length on both sides
import nthash_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
variants = (sys.argv[2] if len(sys.argv) > 2 else "runs,runs_vec,rows").split(",")
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 10
L, k, m = (int(x) for x in os.environ.get("ABLATE_SHAPE", "150,31,1").split(","))
nwin = L - k + 1
ctx = nthash_amd.Context(0)
d_in = ctx.malloc(n * L); d_out = ctx.malloc(n * nwin * m * 8)
ctx.synth_reads_ptr(d_in, 0, n, L, 42)
ctx.set_profiling(True)
gb = n * nwin * (8 * m + L / nwin) / 1e9
res = {v: [] for v in variants}
for rnd in range(rounds):
    for v in variants:
        os.environ["NTHIP_TUNE_NO_DWORD_TAIL"] = "1" if v == "runs_vec" else "0"
        os.environ.pop("NTHIP_TUNE_RUN_LEN", None)
        os.environ.pop("NTHIP_TUNE_WAVES", None)
        if v.startswith("C"):      # e.g. C20 or C20w12
            c, _, w = v[1:].partition("w")
            os.environ["NTHIP_TUNE_RUN_LEN"] = c
            if w:
                os.environ["NTHIP_TUNE_WAVES"] = w
        if v.startswith("map"):
            os.environ["NTHIP_TUNE_TILE_MAP"] = v[3:]
        else:
            os.environ.pop("NTHIP_TUNE_TILE_MAP", None)
        os.environ["NTHIP_TUNE_NO_SPECIAL"] = "1" if v.startswith("gen") else "0"
        if v.startswith("genC"):
            os.environ["NTHIP_TUNE_RUN_LEN"] = v[4:]
        if v.endswith("nom4"):
            os.environ["NTHIP_TUNE_NO_M4"] = "1"
        else:
            os.environ.pop("NTHIP_TUNE_NO_M4", None)
        ctx.reload_tuning()  # the knobs are read once per context
        flags = 8 if v == "rows" else 4 if v == "general" else 0
        ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, m, d_out, n * nwin, flags=flags)
        ms, name = ctx.last_kernel_ms()
        res[v].append(ms)
for v in variants:
    t = res[v][2:] or res[v]
    med, mn = statistics.median(t), min(t)
    print(f"{v:9s} median {med:.3f} ms ({gb/med:.0f} GB/s alg, {n*nwin/med/1e6:.1f} Gkmer/s)  min {mn:.3f} ms ({gb/mn:.0f} GB/s)")
