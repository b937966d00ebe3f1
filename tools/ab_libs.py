#!/usr/bin/env python3
"""A/B two builds of the C-ABI library INSIDE ONE PROCESS (process-to-process spread on a box is ~5 %,
in-process repeatability ~0.5 %): the in-tree library against nthash_amd/lib/ab/libnthash_hip_<tag>.so.

    tools/ab_build.sh x -DSOME_FLAG=1 && python tools/ab_libs.py x [reads] [rounds]   (ABLATE_SHAPE=L,k,m; ABLATE_SEEDS=1;
    ABLATE_DIRTY=1: reads with N's)
"""
import importlib.util, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000_000
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 12
L, k, m = (int(x) for x in os.environ.get("ABLATE_SHAPE", "150,31,1").split(","))
nwin = L - k + 1


def load(path, name):
    if path:
        os.environ["NTHASH_AMD_LIB"] = path
    else:
        os.environ.pop("NTHASH_AMD_LIB", None)
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "nthash_amd", "capi.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.load()
    return mod


A = load(None, "capi_a")
B = load(os.path.join(ROOT, "nthash_amd", "lib", "ab", f"libnthash_hip_{tag}.so"), "capi_b")
ca, cb = A.Context(0), B.Context(0)
SEEDS = ["1010101010101010101010101010101", "1101101101101101011011011011011"] if os.environ.get("ABLATE_SEEDS") else None
per = m * (len(SEEDS) if SEEDS else 1)
sa = A.Seeds(ca, SEEDS, k) if SEEDS else None
sb = B.Seeds(cb, SEEDS, k) if SEEDS else None
d_in = ca.malloc(n * L); d_out = ca.malloc(n * nwin * per * 8)
ca.synth_reads_ptr(d_in, 0, n, L, 42)
if os.environ.get("ABLATE_DIRTY"):   # one N per ~190 kB: the batch takes the N-aware path
    import numpy as np
    for i in range(0, n * L, 187_507):
        ca.h2d(d_in + i, np.frombuffer(b"N", np.uint8))
res = {"base": [], tag: []}
for c in (ca, cb):
    c.set_profiling(True)
for r in range(rounds):
    for name, c in (("base", ca), (tag, cb)):
        if SEEDS:
            c.seed_hash_ptr(d_in, 0, n, L, 0, sa if c is ca else sb, m, d_out, n * nwin)
        else:
            c.kmer_hash_ptr(d_in, 0, n, L, 0, k, m, d_out, n * nwin)
        res[name].append(c.last_kernel_ms()[0])
for name in res:
    t = res[name][2:]
    print(f"{name:8s} median {statistics.median(t):.3f} ms  min {min(t):.3f}  ({n*nwin/statistics.median(t)/1e6:.1f} Gkmer/s)  {c.last_kernel_ms()[1]}")
print(f"ratio {tag}/base = {statistics.median(res[tag][2:]) / statistics.median(res['base'][2:]):.4f}")
