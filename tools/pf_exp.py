#!/usr/bin/env python3
"""The input-prefetch experiment on the headline shape (one process, one pair of buffers): kernel time of config 2's reads
with NTHIP_TUNE_PF_GBPS off / at several rates, leads and chunk sizes (nthip_ctx_reload_tuning between settings).

    python tools/pf_exp.py [reads=60000000]
"""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nthash_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60_000_000
L, k = 150, 31
nwin = L - k + 1
ctx = nthash_amd.Context(0)
ctx.set_profiling(True)
d_in, _ = ctx.malloc_probed(n * L, 3)[0], None
d_out = ctx.malloc_probed(n * nwin * 8, 3)[0]
ctx.synth_reads_ptr(d_in, 0, n, L, 42)
def run(env):
    for key in ("NTHIP_TUNE_PF_GBPS", "NTHIP_TUNE_PF_LEAD_KB", "NTHIP_TUNE_PF_CHUNK_KB"):
        os.environ.pop(key, None)
    os.environ.update(env)
    ctx.reload_tuning()
    ts = []
    for it in range(7):
        ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, 1, d_out, n * nwin)
        ts.append(ctx.last_kernel_ms()[0])
    return statistics.median(ts[2:]), min(ts[2:])
base = run({})
print(f"off                                   median {base[0]:.3f} ms  min {base[1]:.3f}", flush=True)
for gbps in [int(x) for x in os.environ.get("PF_RATES", "700,800,850,900,1000,1500").split(",")]:
    for lead in (1024, 4096, 16384):
        for chunk in (128, 1024):
            med, mn = run({"NTHIP_TUNE_PF_GBPS": str(gbps), "NTHIP_TUNE_PF_LEAD_KB": str(lead), "NTHIP_TUNE_PF_CHUNK_KB": str(chunk)})
            print(f"rate {gbps:5d} GB/s lead {lead:6d} KB chunk {chunk:5d} KB: median {med:.3f} ms  min {mn:.3f}  ({(base[0]/med-1)*100:+.1f} %)", flush=True)
    b2 = run({})
    print(f"off again                             median {b2[0]:.3f} ms  min {b2[1]:.3f}", flush=True)
