#!/usr/bin/env python3
"""What a large hipMalloc / hipFree pair costs on this box (the consumers' rounds allocate their hash streams per call)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nthash_amd
ctx = nthash_amd.Context(0)
for gb in (1, 8, 50, 50, 50):
    t0 = time.perf_counter(); p = ctx.malloc(gb << 30); t1 = time.perf_counter(); ctx.memset(p, 0, gb << 30); ctx.sync() if hasattr(ctx, "sync") else None
    t2 = time.perf_counter(); ctx.free(p); t3 = time.perf_counter()
    print(f"{gb:3d} GiB: malloc {1e3*(t1-t0):8.2f} ms  first touch (memset) {1e3*(t2-t1):8.2f} ms  free {1e3*(t3-t2):8.2f} ms", flush=True)
