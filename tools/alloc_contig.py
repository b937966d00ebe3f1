#!/usr/bin/env python3
"""hipMalloc against hipExtMallocWithFlags(hipDeviceMallocContiguous): headline kernel and fill rate per fresh pair of buffers.
    python tools/alloc_contig.py [reads] [rounds]"""
import ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nthash_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
L, k = 150, 31
ctx = nthash_amd.Context(0)
ctx.set_profiling(True)
hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipFree.argtypes = [C.c_void_p]
in_b, out_b = n * L, n * 120 * 8


def alloc(nbytes, flags):
    p = C.c_void_p()
    rc = hip.hipExtMallocWithFlags(C.byref(p), nbytes, flags)
    if rc != 0:
        raise RuntimeError(f"hipExtMallocWithFlags({nbytes}, {flags}) -> {rc}")
    return p.value


hold = []
for r in range(rounds):
    for name, flags in (("default", 0), ("contiguous", 4)):
        try:
            d_in = alloc(in_b, flags); d_out = alloc(out_b, flags)
        except RuntimeError as e:
            print(name, "failed:", e, flush=True)
            continue
        ctx.synth_reads_ptr(d_in, 0, n, L, 42)
        ts = []
        for _ in range(6):
            ctx.kmer_hash_ptr(d_in, 0, n, L, 0, k, 1, d_out, n * 120)
            ts.append(ctx.last_kernel_ms()[0])
        t = statistics.median(ts[1:])
        fill = ctx.fill_bench_ptr(d_out, out_b, 3)
        print(f"{name:11s} kernel {t:7.3f} ms {n*120/t/1e6:5.0f} G k-mers/s   fill {out_b/fill/1e6:5.0f} GB/s", flush=True)
        hold.append((d_in, d_out))
        if len(hold) > 1:
            for p in hold.pop(0):
                hip.hipFree(p)
