import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, nthash_amd
k, m2 = 31, 3
SEEDS = ["1010101010101010101010101010101", "1101101101101101011011011011011"]
for L in (20000, 40000, 100000, 300000):
    for nl in ("0", "1"):
        os.environ["NTHIP_TUNE_NO_SEED_LONG"] = nl
        ctx = nthash_amd.Context(0)
        sd = nthash_amd.Seeds(ctx, SEEDS, k)
        nwin = L - k + 1
        n = max(1, (4 << 30) // (nwin * 48))
        d_in = ctx.malloc(n * L); d_out = ctx.malloc(n * nwin * 48)
        ctx.synth_reads_ptr(d_in, 0, n, L, 7)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); tot = ctx.seed_hash_ptr(d_in, 0, n, L, 0, sd, m2, d_out, n * nwin); ts.append(time.perf_counter() - t0)
        print(f"L={L} NO_SEED_LONG={nl}: whole call {min(ts)*1e3:.3f} ms  {tot/min(ts)/1e9:.1f} G k-mers/s", flush=True)
        ctx.free(d_in); ctx.free(d_out); sd.close(); ctx.close()
