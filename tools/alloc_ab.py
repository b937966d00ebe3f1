#!/usr/bin/env python3
"""Two tuning settings of the library on the headline shape over several FRESH allocations in one process: does a setting's
advantage depend on which pages hipMalloc hands out?   python tools/alloc_ab.py "ENV=V;ENV=V" [reads] [allocations]"""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nthash_amd
env = dict(e.split("=", 1) for e in sys.argv[1].split(";") if e)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40_000_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
L, k = 150, 31
base = nthash_amd.Context(0)
os.environ.update(env)
alt = nthash_amd.Context(0)
for key in env:
    os.environ.pop(key, None)
for c in (base, alt):
    c.set_profiling(True)
keep = []
for r in range(reps):
    d_in = base.malloc(n * L); d_out = base.malloc(n * 120 * 8)
    base.synth_reads_ptr(d_in, 0, n, L, 42)
    res = {}
    for name, c in (("base", base), ("alt", alt), ("base", base), ("alt", alt)):
        ts = []
        for _ in range(5):
            c.kmer_hash_ptr(d_in, 0, n, L, 0, k, 1, d_out, n * 120)
            ts.append(c.last_kernel_ms()[0])
        res.setdefault(name, []).append(statistics.median(ts[1:]))
    b, a = min(res["base"]), min(res["alt"])
    print(f"alloc {r}: base {b:7.3f} ms ({n*120/b/1e6:5.0f} G)   alt {a:7.3f} ms ({n*120/a/1e6:5.0f} G)   alt/base {a/b:.3f}", flush=True)
    if r % 2 == 0:
        keep.append((d_in, d_out))      # hold some allocations so that the next ones land elsewhere
    else:
        base.free(d_in); base.free(d_out)
    if len(keep) > 2:
        i, o = keep.pop(0); base.free(i); base.free(o)
