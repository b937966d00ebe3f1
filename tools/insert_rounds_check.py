#!/usr/bin/env python3
"""nthip_kmer_bloom_insert / _count_insert of 20 M x 150 bp with m = 3 (7.2 G values): one long pieces-mode round against rounds of
1.5 G values -- the tables word for word (checksums), and the times."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nthash_amd
def ctx_with(env):
    for k_, v in env.items(): os.environ[k_] = str(v)
    try: return nthash_amd.Context(0)
    finally:
        for k_ in env: os.environ.pop(k_, None)
a, b = ctx_with({}), ctx_with({"NTHIP_TUNE_BLOOM_ROUND": 1_500_000_000})
n, L, k, m = 20_000_000, 150, 31, 3
d_in = a.malloc(n * L); a.synth_reads_ptr(d_in, 0, n, L, 42)
n_bits = 1 << 35
for what in ("filter", "sketch"):
    res = []
    for name, c in (("long rounds", a), ("rounds of 1.5 G", b)):
        nb = n_bits // 8 if what == "filter" else 1 << 30
        d_t = c.malloc(nb)
        ts = []
        for i in range(3):
            c.memset(d_t, 0, nb)
            t0 = time.perf_counter()
            tot = c.bloom_insert_ptr(d_in, n, L, 0, k, m, d_t, n_bits) if what == "filter" else c.count_insert_ptr(d_in, n, L, 0, k, m, d_t, 1 << 30)
            ts.append((time.perf_counter() - t0) * 1e3)
        cs = c.checksum_ptr(d_t, nb // 8)
        res.append(cs)
        print(what, name, tot, "ms", " ".join(f"{t:.1f}" for t in ts), cs, flush=True)
        c.free(d_t)
    print(what, "same table:", res[0] == res[1], flush=True)
