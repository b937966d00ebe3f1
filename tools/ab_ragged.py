#!/usr/bin/env python3
"""A/B builds of the library on a variable-length batch INSIDE ONE PROCESS (whole-call wall time, interleaved):
    python tools/ab_ragged.py tag1,tag2[:ENV=V;ENV=V] [reads] [rounds]      (RAGGED_MOSTLY=150 as tools/ragged_bench.py)"""
import importlib.util, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
tags = [t for t in sys.argv[1].split(",") if t]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 12
k, m = 31, int(os.environ.get("RAGGED_M", "1"))


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


mods = [("base", load(None, "capi_base"), {})]
for i, t in enumerate(tags):
    lib, _, envs = t.partition(":")
    env = dict(e.split("=", 1) for e in envs.split(";") if e)
    path = os.path.join(ROOT, "nthash_amd", "lib", "ab", f"libnthash_hip_{lib}.so") if lib else None
    mods.append((t, load(path, f"capi_{i}"), env))
ctxs = []
for name, mod, env in mods:
    os.environ.update(env)
    ctxs.append((name, mod.Context(0)))
    for key in env:
        os.environ.pop(key, None)
rng = np.random.default_rng(1)
lens = rng.integers(100, 151, n).astype(np.uint64)
if os.environ.get("RAGGED_MOSTLY"):
    full = int(os.environ["RAGGED_MOSTLY"])
    lens = np.where(rng.random(n) < 0.001, rng.integers(100, full, n), full).astype(np.uint64)
offs = np.zeros(n + 1, np.uint64); offs[1:] = np.cumsum(lens)
total_bytes = int(offs[-1])
c0 = ctxs[0][1]
d_in = c0.malloc(total_bytes + 64)
c0.synth_reads_ptr(d_in, 0, (total_bytes + 149) // 150, 150, 42)
for i in range(0, total_bytes, 1_000_003):
    c0.h2d(d_in + i, np.frombuffer(b"N", np.uint8))
d_offs = c0.malloc((n + 1) * 8); c0.h2d(d_offs, offs)
cap = int((lens - k + 1).sum())
d_out = c0.malloc(cap * m * 8)
res = {name: [] for name, _ in ctxs}
tot = 0
for r in range(rounds):
    for name, c in (ctxs if r % 2 == 0 else ctxs[::-1]):
        t0 = time.perf_counter()
        tot = c.kmer_hash_ptr(d_in, d_offs, n, 0, 0, k, m, d_out, cap)
        res[name].append(time.perf_counter() - t0)
base = statistics.median(res["base"][2:])
for name in res:
    t = statistics.median(res[name][2:])
    print(f"{name:40s} median {t*1e3:7.3f} ms  min {min(res[name])*1e3:7.3f}  {tot/t/1e9:6.1f} Gkmer/s  ratio {t/base:.4f}", flush=True)
