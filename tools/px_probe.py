#!/usr/bin/env python3
"""One spaced-seed shape on seed_px_kernel, kernel time only: python tools/px_probe.py L k runs|seedstring[,seed...] m [reads]
(NTHIP_TUNE_SEED_PX_READS / _WAVES / _ARRAY from the environment); for rocprofv3 --pmc passes and quick A/B."""
import os, statistics, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("NTHIP_SEED_JIT", "1")  # steady state: compile on the spot (a job meets its kernel after the first batches)
import nthash_amd

L, k, spec, m2 = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
if spec.isdigit() and len(spec) < 3:
    def blocky(k, gaps):
        s = np.ones(k, dtype=bool)
        for a, n in gaps:
            s[a:a + n] = False
        return "".join("1" if b else "0" for b in s)
    def evenly(k, runs):
        unit = k / (runs + (runs - 1) / 3)
        gaps, p = [], unit
        for _ in range(runs - 1):
            gaps.append((int(round(p)), max(1, int(round(unit / 3)))))
            p += unit + unit / 3
        return blocky(k, gaps)
    seeds = [evenly(k, int(spec))]
elif spec.startswith("rnd"):
    rng = np.random.default_rng(5)
    seeds = []
    for _ in range(int(spec[3:])):
        half = rng.random((k + 1) // 2) < 0.7
        s = np.concatenate([half, half[: k // 2][::-1]]); s[0] = s[-1] = True
        seeds.append("".join("1" if b else "0" for b in s))
else:
    seeds = spec.split(",")
if "NTHIP_TUNE_SEED_PX" not in os.environ:
    os.environ.setdefault("NTHIP_TUNE_SEED_PS", "1")
ctx = nthash_amd.Context(0)
ctx.set_profiling(True)
nwin, per = L - k + 1, len(seeds) * m2
n = int(sys.argv[5]) if len(sys.argv) > 5 else max(1, int((4 << 30) // (nwin * per * 8)))
sd = nthash_amd.Seeds(ctx, seeds, k)
d_in, d_out = ctx.malloc(n * L), ctx.malloc(n * nwin * per * 8)
ctx.synth_reads_ptr(d_in, 0, n, L, 7)
ts = []
for it in range(5):
    ctx.seed_hash_ptr(d_in, 0, n, L, 0, sd, m2, d_out, n * nwin)
    ms, name = ctx.last_kernel_ms()
    ts.append(ms)
ms = statistics.median(ts[1:])
print(f"L={L} k={k} seeds={len(seeds)} m={m2} {name} {ms:.3f} ms {n * nwin / ms / 1e6:.1f} G windows/s "
      f"frac {(n * L + n * nwin * per * 8) / ms / 1e9 / 8:.3f}", flush=True)
