#!/usr/bin/env python3
"""Write-only / copy yardsticks of several builds in one process:  python tools/fill_probe.py tag,tag,... [GiB]"""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tags = [""] + [t for t in (sys.argv[1] if len(sys.argv) > 1 else "").split(",") if t]
gib = int(sys.argv[2]) if len(sys.argv) > 2 else 16
nbytes = gib << 30
bufs = None
for i, t in enumerate(tags):
    if t:
        os.environ["NTHASH_AMD_LIB"] = os.path.join(ROOT, "nthash_amd", "lib", "ab", f"libnthash_hip_{t}.so")
    else:
        os.environ.pop("NTHASH_AMD_LIB", None)
    spec = importlib.util.spec_from_file_location(f"capi_{i}", os.path.join(ROOT, "nthash_amd", "capi.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.load()
    c = mod.Context(0)
    if bufs is None:
        bufs = (c.malloc(nbytes), c.malloc(nbytes))
    f = c.fill_bench_ptr(bufs[0], nbytes, 8)
    cp = c.copy_bench_ptr(bufs[1], bufs[0], nbytes, 8)
    print(f"{t or 'base':12s} fill {nbytes/f/1e6:7.0f} GB/s   copy {2*nbytes/cp/1e6:7.0f} GB/s (r+w)", flush=True)
