#!/usr/bin/env python3
"""Seeds of few runs: the run-by-run rolling kernel (seed_roll_kernel.hpp) against the masked direct forms, kernel time.

    python tools/seed_roll_sweep.py [out.json]
Every shape is hashed by a context with NTHIP_TUNE_SEED_PX=1 (sparse sums over scanned arrays, seed_px_kernel.hpp), one with
NTHIP_TUNE_SEED_ROLL=1 (rolled whenever the kernel takes the seed set), one with both =2 (neither) and one without a knob
(the cost model's choice).  Prints G k-mers/s and the fraction of the 8 TB/s roofline
(input bytes + 8 * seeds * m bytes per k-mer).
"""
import json, os, statistics, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NTHIP_SEED_JIT", "1")  # steady state: compile on the spot (a job meets its kernel after the first batches)
import nthash_amd


def blocky(k, gaps):
    s = np.ones(k, dtype=bool)
    for a, n in gaps:
        s[a:a + n] = False
    return "".join("1" if b else "0" for b in s)


def evenly(k, runs):  # `runs` care runs with gaps a third of a run wide
    unit = k / (runs + (runs - 1) / 3)
    gaps, p = [], unit
    for _ in range(runs - 1):
        gaps.append((int(round(p)), max(1, int(round(unit / 3)))))
        p += unit + unit / 3
    return blocky(k, gaps)


SHAPES = [  # (L, seeds, m per seed)
    (250, [blocky(128, [(40, 48)])], 1),
    (250, [evenly(128, 3)], 1),
    (250, [evenly(128, 5)], 1),
    (250, [evenly(128, 8)], 1),
    (250, [evenly(128, 12)], 1),
    (300, [blocky(160, [(30, 20), (110, 20)])], 1),
    (300, [evenly(160, 6)], 1),
    (250, [evenly(128, 3), evenly(128, 4)], 1),
    (250, [evenly(128, 3)], 2),
    (250, [evenly(64, 2)], 1),
    (250, [evenly(64, 4)], 1),
    (250, [evenly(64, 8)], 1),
    (150, [evenly(64, 3)], 1),
    (250, [evenly(31, 2)], 1),
    (250, [evenly(31, 4)], 1),
    (250, [evenly(31, 3), evenly(31, 2)], 3),
    (250, [evenly(48, 3), evenly(48, 5)], 1),
]
OUT_BUDGET = int(os.environ.get("SWEEP_GIB", "8")) << 30


def make_ctx(**env):
    for key, v in env.items():
        os.environ[key] = v
    try:
        c = nthash_amd.Context(0)
    finally:
        for key in env:
            os.environ.pop(key, None)
    c.set_profiling(True)
    return c


ctxs = {"ps": make_ctx(NTHIP_TUNE_SEED_PS="1"), "px": make_ctx(NTHIP_TUNE_SEED_PX="1"), "rolled": make_ctx(NTHIP_TUNE_SEED_ROLL="1"),
        "direct": make_ctx(NTHIP_TUNE_SEED_ROLL="2", NTHIP_TUNE_SEED_PX="2", NTHIP_TUNE_SEED_PS="2"), "default": make_ctx()}
if os.environ.get("SWEEP_ONLY"):
    ctxs = {t: c for t, c in ctxs.items() if t in os.environ["SWEEP_ONLY"].split(",") or t == "default"}
rows = []
for (L, seeds, m2) in SHAPES:
    k, ns = len(seeds[0]), len(seeds)
    nwin, per = L - k + 1, ns * m2
    n = max(1, int(OUT_BUDGET // (nwin * per * 8)))
    runs = sum(sum(1 for i, ch in enumerate(s) if ch == "1" and (i == 0 or s[i - 1] == "0")) for s in seeds)
    row = dict(L=L, k=k, seeds=ns, m=m2, runs=runs, reads=n)
    c0 = ctxs["default"]
    d_in, d_out = c0.malloc(n * L), c0.malloc(n * nwin * per * 8)
    c0.synth_reads_ptr(d_in, 0, n, L, 7)
    for tag, c in ctxs.items():
        sd = nthash_amd.Seeds(c, seeds, k)
        ts, name = [], "?"
        for it in range(5):
            c.seed_hash_ptr(d_in, 0, n, L, 0, sd, m2, d_out, n * nwin)
            ms, name = c.last_kernel_ms()
            ts.append(ms)
        ms = statistics.median(ts[1:])
        row[tag] = dict(kernel=name, ms=round(ms, 3), gkmer_s=round(n * nwin / ms / 1e6, 1),
                        frac=round((n * L + n * nwin * per * 8) / ms / 1e9 / 8, 3))
        sd.close()
    c0.free(d_in); c0.free(d_out)
    rows.append(row)
    print(f"L={L:4d} k={k:3d} seeds={ns} m={m2} runs={runs:2d}  " + "  ".join(
        f"{t}: {row[t]['gkmer_s']:6.1f} G ({row[t]['frac']:.2f}) {row[t]['kernel'][5:14]}" for t in ctxs), flush=True)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
