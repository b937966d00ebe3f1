#!/usr/bin/env python3
"""A/B builds of the library on SeedNtHash over a variable-length batch inside one process (whole call, interleaved):
    python tools/ab_ragged_seeds.py tag1,tag2[:ENV=V] [reads] [rounds]"""
import importlib.util, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
tags = [t for t in sys.argv[1].split(",") if t]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 10
SEEDS = ["1010101010101010101010101010101", "1101101101101101011011011011011"]
k, m2 = 31, 3


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
    c = mod.Context(0)
    ctxs.append((name, c, mod.Seeds(c, SEEDS, k)))
    for key in env:
        os.environ.pop(key, None)
rng = np.random.default_rng(1)
lens = rng.integers(100, 251, n).astype(np.uint64)
offs = np.zeros(n + 1, np.uint64); offs[1:] = np.cumsum(lens)
total_bytes = int(offs[-1])
c0 = ctxs[0][1]
d_in = c0.malloc(total_bytes + 64)
c0.synth_reads_ptr(d_in, 0, (total_bytes + 249) // 250, 250, 42)
for i in range(0, total_bytes, 600_011):
    c0.h2d(d_in + i, np.frombuffer(b"N", np.uint8))
d_offs = c0.malloc((n + 1) * 8); c0.h2d(d_offs, offs)
cap = int((lens - k + 1).sum())
d_out = c0.malloc(cap * 6 * 8)
res = {name: [] for name, _, _ in ctxs}
tot = 0
for r in range(rounds):
    for name, c, sd in (ctxs if r % 2 == 0 else ctxs[::-1]):
        t0 = time.perf_counter()
        tot = c.seed_hash_ptr(d_in, d_offs, n, 0, 0, sd, m2, d_out, cap)
        res[name].append(time.perf_counter() - t0)
base = statistics.median(res["base"][2:])
for name in res:
    t = statistics.median(res[name][2:])
    print(f"{name:30s} median {t*1e3:7.3f} ms  min {min(res[name])*1e3:7.3f}  {tot/t/1e9:6.1f} Gkmer/s  ratio {t/base:.4f}", flush=True)
