import sys, time, threading
sys.path.insert(0, '/root/repo')
import nthash_amd
a, b = nthash_amd.Context(0), nthash_amd.Context(0)
n, L, k = 20_000_000, 150, 31
d_in = a.malloc(n * L); a.synth_reads_ptr(d_in, 0, n, L, 42)
nk = n * (L - k + 1)
d_out = a.malloc(nk * 8)
d_chk = b.malloc(3_000_000_000)
b.memset(d_chk, 1, 3_000_000_000)
def hash_loop(reps):
    for _ in range(reps): a.kmer_hash_ptr(d_in, 0, n, L, 0, k, 1, d_out, nk)
def sweep_loop(reps):
    for _ in range(reps): b.checksum_ptr(d_chk, 3_000_000_000 // 8)
hash_loop(3); sweep_loop(3)
R = 20
t0 = time.perf_counter(); hash_loop(R); th = (time.perf_counter() - t0) / R
t0 = time.perf_counter(); sweep_loop(R); ts = (time.perf_counter() - t0) / R
t0 = time.perf_counter()
x = threading.Thread(target=hash_loop, args=(R,)); y = threading.Thread(target=sweep_loop, args=(R,))
x.start(); y.start(); x.join(); y.join()
tb = (time.perf_counter() - t0) / R
print(f"hash alone {th*1e3:.3f} ms  sweep alone {ts*1e3:.3f} ms  sum {(th+ts)*1e3:.3f}  both at once {tb*1e3:.3f} ms per pair")
