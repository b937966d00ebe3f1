import sys, time
sys.path.insert(0, '/root/repo')
import nthash_amd
from bench import SEED_A, SEED_B
ctx = nthash_amd.Context(0)
n4, L4 = 5_000_000, 250
d_in4 = ctx.malloc(n4 * L4)
ctx.synth_reads_ptr(d_in4, 0, n4, L4, 42)
sd = nthash_amd.Seeds(ctx, [SEED_A, SEED_B], 31)
n_bits = 1 << 35
d_f = ctx.malloc(n_bits // 8)
ctx.memset(d_f, 0, n_bits // 8)
for i in range(3):
    t0 = time.perf_counter()
    tot = ctx.seed_bloom_insert_ptr(d_in4, n4, L4, 0, sd, 3, d_f, n_bits)
    print("insert", (time.perf_counter() - t0) * 1e3, "ms", tot)
