#!/usr/bin/env python3
"""Multi-line FASTA (genome-like: few long sequences, 60 bases per line) -> k-mer hashes on the device.

    python tools/fasta_bench.py [total_Mbp] [n_sequences]
"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nthash_amd
from nthash_amd.capi import NTHIP_FASTA_MULTILINE
mbp = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n_seq = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W = 60
lines_per_seq = mbp * 1_000_000 // n_seq // W
ctx = nthash_amd.Context(0)
path = os.path.join(tempfile.gettempdir(), "nthash_bench.fa")
t0 = time.perf_counter()
with open(path, "wb") as f:
    for s in range(n_seq):
        d = ctx.malloc(lines_per_seq * W)
        ctx.synth_reads_ptr(d, s * lines_per_seq, lines_per_seq, W, 42)
        rows = np.empty((lines_per_seq, W + 1), np.uint8)
        flat = np.zeros(lines_per_seq * W, np.uint8)
        ctx.d2h(flat, d)
        ctx.free(d)
        rows[:, :W] = flat.reshape(lines_per_seq, W)
        rows[:, W] = ord("\n")
        f.write(b">chr%d synthetic\n" % s)
        rows.tofile(f)
size = os.path.getsize(path)
print(f"wrote {size/1e9:.2f} GB ({n_seq} sequences of {lines_per_seq*W/1e6:.1f} Mbp) in {time.perf_counter()-t0:.1f} s", flush=True)
ctx.set_profiling(True)
for it in range(3):
    st = ctx.fastx_kmer_hash_file(path, NTHIP_FASTA_MULTILINE, 31, 1)
    print(f"run {it}: {st.seconds*1e3:8.1f} ms  {st.file_bytes/st.seconds/1e9:6.2f} GB/s of file  {st.kmers/st.seconds/1e9:6.2f} G k-mers/s"
          f"  (load {st.read_seconds*1e3:.0f} ms, compact+hash {st.gpu_seconds*1e3:.0f} ms, last kernel {ctx.last_kernel_ms()})", flush=True)
if len(sys.argv) > 3 and sys.argv[3] == "gz":
    # the same genome bgzipped (65280-byte blocks, level 1): blocks inflated by the reader threads, then the one-batch path;
    # NTHIP_TUNE_NO_BGZF=1: one inflating thread
    import struct, zlib
    bg = path + ".gz"
    t0 = time.perf_counter()
    with open(path, "rb") as fi, open(bg, "wb") as fo:
        while True:
            raw = fi.read(65280)
            if not raw:
                break
            co = zlib.compressobj(1, zlib.DEFLATED, -15)
            payload = co.compress(raw) + co.flush()
            fo.write(struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, 66, 67, 2, 18 + len(payload) + 8 - 1))
            fo.write(payload)
            fo.write(struct.pack("<II", zlib.crc32(raw), len(raw)))
        fo.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    print(f"BGZF: {os.path.getsize(bg)/1e9:.2f} GB in {time.perf_counter()-t0:.1f} s", flush=True)
    for tag, env in (("bgzf", None), ("bgzf, one thread", "1")):
        if env:
            os.environ["NTHIP_TUNE_NO_BGZF"] = env
        for it in range(2):
            st = ctx.fastx_kmer_hash_file(bg, NTHIP_FASTA_MULTILINE, 31, 1)
            print(f"{tag} run {it}: {st.seconds*1e3:8.1f} ms  {size/st.seconds/1e9:6.2f} GB/s inflated  {st.kmers/st.seconds/1e9:6.2f} G k-mers/s", flush=True)
        os.environ.pop("NTHIP_TUNE_NO_BGZF", None)
    os.remove(bg)
os.remove(path)
