import sys, time
sys.path.insert(0, '/root/repo')
import nthash_amd
ctx = nthash_amd.Context(0)
n, L, k = 20_000_000, 150, 31
d_in = ctx.malloc(n * L); ctx.synth_reads_ptr(d_in, 0, n, L, 42)
n_bits = 1 << 35
d_f = ctx.malloc(n_bits // 8); ctx.memset(d_f, 0, n_bits // 8)
ctx.bloom_insert_ptr(d_in, n // 2, L, 0, k, 1, d_f, n_bits)
d_hits = ctx.malloc(n * 8)
for prof in (False, True, False, True):
    ctx.set_profiling(prof)
    for _ in range(3):
        t0 = time.perf_counter()
        r = ctx.bloom_query_ptr(d_in, n, L, 0, k, 1, d_f, n_bits, hits=d_hits)
        w = (time.perf_counter() - t0) * 1e3
        print(f"profiling={prof} wall {w:.2f} ms", ctx.last_kernel_ms() if prof else "", r)
