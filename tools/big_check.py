import sys, numpy as np
sys.path.insert(0, "/root/repo")
import nthash_amd
ctx = nthash_amd.Context(0)
ctx.set_profiling(True)
def run(n, L, k, m, dirty):
    nwin = L - k + 1
    d_in = ctx.malloc(n * L); d_out = ctx.malloc(n * nwin * m * 8)
    ctx.synth_reads_ptr(d_in, 0, n, L, 7)
    if dirty:
        for off in range(1000, n * L, n * L // 5000):
            ctx.h2d(d_in + off, np.frombuffer(b"N", np.uint8))
    tot = ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, m, d_out, n * nwin)
    name = ctx.last_kernel_ms()[1]
    s_all, x_all = ctx.checksum_ptr(d_out, tot * m)
    parts = 5
    s_sum, x_sum, t_sum = 0, 0, 0
    step = n // parts
    for i in range(parts):
        r0 = i * step; nr = step if i < parts - 1 else n - r0
        t = ctx.kmer_hash_ptr(d_in + r0 * L, 0, nr, L, 0, k, m, d_out, nr * nwin)
        s, x = ctx.checksum_ptr(d_out, t * m)
        s_sum = (s_sum + s) & (2**64 - 1); x_sum ^= x; t_sum += t
    ok = (tot, s_all, x_all) == (t_sum, s_sum, x_sum)
    print(f"n={n} L={L} k={k} m={m} dirty={dirty} kernel={name} total={tot} {'OK' if ok else 'MISMATCH'}", flush=True)
    ctx.free(d_in); ctx.free(d_out)
    return ok
ok = True
ok &= run(50_000_000, 151, 31, 1, False)
ok &= run(50_000_000, 151, 31, 1, True)
ok &= run(30_000_000, 101, 25, 2, False)
ok &= run(40_000_000, 150, 31, 1, True)
ok &= run(200, 30_000_001, 31, 1, False)      # very long reads: 6 G k-mers, tiles cross reads
ok &= run(20_000_000, 300, 128, 1, True)
sys.exit(0 if ok else 1)
