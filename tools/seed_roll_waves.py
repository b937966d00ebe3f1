import os, sys, statistics
sys.path.insert(0, "/root/repo")
import numpy as np, nthash_amd
def blocky(k, gaps):
    s = np.ones(k, dtype=bool)
    for a, n in gaps: s[a:a + n] = False
    return "".join("1" if b else "0" for b in s)
c = nthash_amd.Context(0); c.set_profiling(True)
for (L, seeds, m2) in [(250, [blocky(128, [(40, 48)])], 1), (250, [blocky(31, [(10, 8)])], 1)]:
    k = len(seeds[0]); nwin = L - k + 1; n = (4 << 30) // (nwin * 8 * m2)
    d_in, d_out = c.malloc(n * L), c.malloc(n * nwin * 8 * m2)
    c.synth_reads_ptr(d_in, 0, n, L, 7)
    sd = nthash_amd.Seeds(c, seeds, k)
    ts = []
    for it in range(5):
        c.seed_hash_ptr(d_in, 0, n, L, 0, sd, m2, d_out, n * nwin); ms, name = c.last_kernel_ms(); ts.append(ms)
    ms = statistics.median(ts[1:]); print(os.environ.get("NTHIP_TUNE_SEED_ROLL_WAVES"), k, name, round(n * nwin / ms / 1e6, 1), "G", flush=True)
    c.free(d_in); c.free(d_out); sd.close()
