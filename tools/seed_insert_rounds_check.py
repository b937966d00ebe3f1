#!/usr/bin/env python3
"""nthip_seed_bloom_insert of config 4's seed pair on 5 M x 250 bp: the filter of ONE round of 6.6 G values (pieces mode, rounds as
long as the memory allows) against the filter built in rounds of 1.5 G values, word for word (checksums of the 4 GiB); timings."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nthash_amd
from bench import SEED_A, SEED_B
def ctx_with(env):
    for k_, v in env.items(): os.environ[k_] = str(v)
    try: return nthash_amd.Context(0)
    finally:
        for k_ in env: os.environ.pop(k_, None)
a, b = ctx_with({}), ctx_with({"NTHIP_TUNE_BLOOM_ROUND": 1_500_000_000})
n4, L4 = 5_000_000, 250
d_in = a.malloc(n4 * L4); a.synth_reads_ptr(d_in, 0, n4, L4, 42)
n_bits = 1 << 35
res = []
for name, c in (("one round", a), ("rounds of 1.5 G", b)):
    sd = nthash_amd.Seeds(c, [SEED_A, SEED_B], 31)
    d_f = c.malloc(n_bits // 8); c.memset(d_f, 0, n_bits // 8)
    ts = []
    for i in range(4):
        t0 = time.perf_counter(); tot = c.seed_bloom_insert_ptr(d_in, n4, L4, 0, sd, 3, d_f, n_bits); ts.append((time.perf_counter() - t0) * 1e3)
    cs = c.checksum_ptr(d_f, n_bits // 64)
    res.append(cs)
    print(name, "windows", tot, "ms", " ".join(f"{t:.1f}" for t in ts), "filter checksum", cs, flush=True)
    c.set_profiling(True); c.seed_bloom_insert_ptr(d_in, n4, L4, 0, sd, 3, d_f, n_bits); print("  last:", c.last_kernel_ms()); c.set_profiling(False)
    c.free(d_f)
print("same filter:", res[0] == res[1])
