#!/usr/bin/env python3
"""A/B several builds of the C-ABI library INSIDE ONE PROCESS, interleaved (process-to-process spread on a box is ~5 %,
in-process repeatability ~0.5 %): the in-tree library against nthash_amd/lib/ab/libnthash_hip_<tag>.so for every tag.

    UNITS=capi_kmer_runs tools/ab_build.sh x -DSOME_FLAG=1
    python tools/ab_multi.py x,y,z [reads] [rounds]     (ABLATE_SHAPE=L,k,m; ABLATE_SEEDS=1: the two bench seeds)
A tag may carry tuning knobs for its context: x:NTHIP_TUNE_WAVES=16;NTHIP_TUNE_RUN_LEN=30  (":..." alone = base library).
"""
import importlib.util, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NTHIP_TUNE_NO_AUTOTUNE", "1")
tags = [t for t in sys.argv[1].split(",") if t]
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


mods = [("base", load(None, "capi_base"), {})]
for i, t in enumerate(tags):
    lib, _, envs = t.partition(":")
    env = dict(e.split("=", 1) for e in envs.split(";") if e)
    path = os.path.join(ROOT, "nthash_amd", "lib", "ab", f"libnthash_hip_{lib}.so") if lib else None
    mods.append((t, load(path, f"capi_{i}"), env))
ctxs = []
for name, mod, env in mods:
    os.environ.update(env)
    ctxs.append((name, mod, mod.Context(0)))  # the knobs are read when the context is created
    for key in env:
        os.environ.pop(key, None)
SEEDS = None
if os.environ.get("ABLATE_SEEDS"):  # "1": the two bench seeds; otherwise a comma-separated list of masks
    SEEDS = (["1010101010101010101010101010101", "1101101101101101011011011011011"] if os.environ["ABLATE_SEEDS"] == "1"
             else os.environ["ABLATE_SEEDS"].split(","))
per = m * (len(SEEDS) if SEEDS else 1)
seeds = {name: (mod.Seeds(c, SEEDS, k) if SEEDS else None) for name, mod, c in ctxs}
c0 = ctxs[0][2]
if os.environ.get("AB_PROBED"):  # buffers from the placement-aware allocator (fast page sets)
    d_in, g_in, _ = c0.malloc_probed(n * L, 5)
    d_out, g_out, _ = c0.malloc_probed(n * nwin * per * 8, 3)
    print(f"probed buffers: reads {g_in:.0f} GB/s fill, hashes {g_out:.0f} GB/s fill", flush=True)
else:
    d_in = c0.malloc(n * L)
    d_out = c0.malloc(n * nwin * per * 8)
c0.synth_reads_ptr(d_in, 0, n, L, 42)
res = {name: [] for name, _, _ in ctxs}
kern = {}
for _, _, c in ctxs:
    c.set_profiling(True)
for r in range(rounds):
    order = ctxs if r % 2 == 0 else ctxs[::-1]
    for name, mod, c in order:
        if SEEDS:
            c.seed_hash_ptr(d_in, 0, n, L, 0, seeds[name], m, d_out, n * nwin)
        else:
            c.kmer_hash_ptr(d_in, 0, n, L, 0, k, m, d_out, n * nwin)
        ms, kn = c.last_kernel_ms()
        res[name].append(ms)
        kern[name] = kn
base = statistics.median(res["base"][2:])
gb = n * nwin * (8 * per + L / nwin) / 1e9
for name in res:
    t = res[name][2:]
    med = statistics.median(t)
    print(f"{name:34s} median {med:8.3f} ms  min {min(t):8.3f}  {n*nwin/med/1e6:7.1f} Gkmer/s  {gb/med*1e3:6.0f} GB/s alg  "
          f"ratio {med/base:.4f}  {kern[name]}", flush=True)
