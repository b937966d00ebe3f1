import sys, time
sys.path.insert(0, '/root/repo')
import nthash_amd
ctx = nthash_amd.Context(0)
for gb in (2, 16, 50):
    for cand in (1, 2, 3):
        t0 = time.perf_counter(); p, g, t = ctx.malloc_probed(gb << 30, cand); t1 = time.perf_counter(); ctx.free(p); t2 = time.perf_counter()
        print(f"{gb} GiB candidates {cand}: probed {1e3*(t1-t0):.1f} ms (tried {t}, {g:.0f} GB/s), free {1e3*(t2-t1):.1f} ms", flush=True)
