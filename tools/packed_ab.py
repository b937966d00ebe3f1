#!/usr/bin/env python3
"""In-process A/B of builds of the C-ABI library on the PACKED-input headline path (see tools/ab_multi.py).

    python tools/packed_ab.py tag1,tag2 [reads] [rounds]      (ABLATE_SHAPE=L,k,m)
"""
import importlib.util, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tags = [t for t in sys.argv[1].split(",") if t]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40_000_000
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 10
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


mods = [("base", load(None, "capi_base"), {})]
for i, t in enumerate(tags):
    lib, _, envs = t.partition(":")
    env = dict(e.split("=", 1) for e in envs.split(";") if e)
    path = os.path.join(ROOT, "nthash_amd", "lib", "ab", f"libnthash_hip_{lib}.so") if lib else None
    mods.append((t, load(path, f"capi_{i}"), env))
ctxs = []
for name, mod, env in mods:
    os.environ.update(env)
    ctxs.append((name, mod, mod.Context(0)))
    for key in env:
        os.environ.pop(key, None)
c0, m0 = ctxs[0][2], ctxs[0][1]
d_in, g_in, _ = c0.malloc_probed(n * L, 5)
d_out, g_out, _ = c0.malloc_probed(n * nwin * m * 8, 3)
tot, _ = c0.packed_size(n * L)
d_pk, g_pk, _ = c0.malloc_probed(tot, 5)
print(f"probed buffers: reads {g_in:.0f}, hashes {g_out:.0f}, packed {g_pk:.0f} GB/s fill", flush=True)
c0.synth_reads_ptr(d_in, 0, n, L, 42)
assert c0.pack_reads_ptr(d_in, 0, n, L, 0, d_pk) == 0
res = {name: [] for name, _, _ in ctxs}
res["ascii(base)"] = []
kern = {}
for _, _, c in ctxs:
    c.set_profiling(True)
for r in range(rounds):
    order = ctxs if r % 2 == 0 else ctxs[::-1]
    for name, mod, c in order:
        c.kmer_hash_ptr(d_pk, 0, n, L, 0, k, m, d_out, n * nwin, flags=mod.NTHIP_PACKED_INPUT | mod.NTHIP_PACKED_CLEAN)
        ms, kn = c.last_kernel_ms()
        res[name].append(ms)
        kern[name] = kn
    c0.kmer_hash_ptr(d_in, 0, n, L, 0, k, m, d_out, n * nwin)
    res["ascii(base)"].append(c0.last_kernel_ms()[0])
    kern["ascii(base)"] = c0.last_kernel_ms()[1]
base = statistics.median(res["base"][2:])
for name in res:
    t = res[name][2:]
    med = statistics.median(t)
    print(f"{name:34s} median {med:8.3f} ms  min {min(t):8.3f}  {n*nwin/med/1e6:7.1f} Gkmer/s  ratio {med/base:.4f}  {kern[name]}", flush=True)
