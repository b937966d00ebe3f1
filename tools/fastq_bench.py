#!/usr/bin/env python3
"""End-to-end FASTQ file -> k-mer hashes on the device (nthip_fastx_kmer_hash_file).

    python tools/fastq_bench.py [reads] [chunk_MiB] [m] [seeds|kmers] [gz]
("gz": the file is also written gzip-compressed -- `gzip -1`, Phred scores drawn from 8 values so that it compresses like a
sequencer's output, not like a constant -- and streamed through the same call: one inflating host thread feeds the ring)
Writes a synthetic FASTQ (150 bp reads, 321-byte records) to $TMPDIR, streams it twice (page cache warm),
prints file GB/s, reads/s, k-mers/s and the stage times the driver reports.
"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nthash_amd
from nthash_amd.capi import NTHIP_FASTQ
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
chunk = (int(sys.argv[2]) if len(sys.argv) > 2 else 256) << 20
m = int(sys.argv[3]) if len(sys.argv) > 3 else 1
use_gz = len(sys.argv) > 5 and sys.argv[5] == "gz"
use_seeds = len(sys.argv) > 4 and sys.argv[4] == "seeds"   # SeedNtHash, BASELINE config 4 seeds, m hashes per seed
L, k = 150, 31
ctx = nthash_amd.Context(0)
d_in = ctx.malloc(n * L)
ctx.synth_reads_ptr(d_in, 0, n, L, 42)
seqs = np.zeros(n * L, np.uint8)
ctx.d2h(seqs, d_in)
ctx.free(d_in)
rec = np.empty((n, 17 + L + 1 + 2 + L + 1), np.uint8)
rec[:, 0] = ord("@")
ids = np.arange(n, dtype=np.uint64)
for d in range(15):
    rec[:, 15 - d] = (ids // np.uint64(10 ** d) % np.uint64(10)).astype(np.uint8) + ord("0")
rec[:, 16] = ord("\n")
rec[:, 17:17 + L] = seqs.reshape(n, L)
rec[:, 17 + L] = ord("\n")
rec[:, 18 + L] = ord("+")
rec[:, 19 + L] = ord("\n")
rec[:, 20 + L:20 + 2 * L] = ord("I")
if use_gz:
    rec[:, 20 + L:20 + 2 * L] = np.frombuffer(b"#,5:AFI?", np.uint8)[np.random.default_rng(1).integers(0, 8, (n, L), dtype=np.uint8)]
rec[:, 20 + 2 * L] = ord("\n")
path = os.path.join(tempfile.gettempdir(), "nthash_bench.fq")
t0 = time.perf_counter()
rec.tofile(path)
print(f"wrote {rec.nbytes/1e9:.2f} GB in {time.perf_counter()-t0:.1f} s -> {path}", flush=True)
del rec, seqs
sd = nthash_amd.Seeds(ctx, ["1010101010101010101010101010101", "1101101101101101011011011011011"], k) if use_seeds else None
for it in range(3):
    st = ctx.fastx_kmer_hash_file(path, NTHIP_FASTQ, k, m, chunk_bytes=chunk, seeds=sd)
    print(f"run {it}: {st.seconds*1e3:8.1f} ms  {st.file_bytes/st.seconds/1e9:6.2f} GB/s of file  "
          f"{st.reads/st.seconds/1e6:7.1f} M reads/s  {st.kmers/st.seconds/1e9:6.2f} G k-mers/s  "
          f"(batches {st.batches}, pread {st.read_seconds*1e3:.0f} ms, index+hash {st.gpu_seconds*1e3:.0f} ms)", flush=True)
if use_gz:
    import subprocess
    t0 = time.perf_counter()
    subprocess.run(["gzip", "-1", "-k", "-f", path], check=True)
    gz = path + ".gz"
    print(f"gzip -1: {os.path.getsize(gz)/1e9:.2f} GB in {time.perf_counter()-t0:.1f} s", flush=True)
    raw_bytes = os.path.getsize(path)
    for it in range(2):
        st = ctx.fastx_kmer_hash_file(gz, NTHIP_FASTQ, k, m, chunk_bytes=chunk, seeds=sd)
        print(f"gz run {it}: {st.seconds*1e3:8.1f} ms  {raw_bytes/st.seconds/1e9:6.2f} GB/s inflated ({st.file_bytes/st.seconds/1e9:5.2f} on disk)  "
              f"{st.reads/st.seconds/1e6:7.1f} M reads/s  {st.kmers/st.seconds/1e9:6.2f} G k-mers/s  "
              f"(batches {st.batches}, inflate {st.read_seconds*1e3:.0f} ms, index+hash {st.gpu_seconds*1e3:.0f} ms)", flush=True)
    os.remove(gz)
    # the same bytes as BGZF (bgzip's layout: 65280-byte blocks, level 1 here): blocks inflated by the reader threads
    import struct, zlib
    bg = path + ".bgzf.gz"
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
    for tag, env in (("bgzf", None), ("bgzf, one thread (NTHIP_TUNE_NO_BGZF=1)", "1")):
        if env:
            os.environ["NTHIP_TUNE_NO_BGZF"] = env
        for it in range(2):
            st = ctx.fastx_kmer_hash_file(bg, NTHIP_FASTQ, k, m, chunk_bytes=chunk, seeds=sd)
            print(f"{tag} run {it}: {st.seconds*1e3:8.1f} ms  {raw_bytes/st.seconds/1e9:6.2f} GB/s inflated ({st.file_bytes/st.seconds/1e9:5.2f} on disk)  "
                  f"{st.reads/st.seconds/1e6:7.1f} M reads/s  {st.kmers/st.seconds/1e9:6.2f} G k-mers/s  "
                  f"(batches {st.batches}, inflate {st.read_seconds*1e3:.0f} ms, index+hash {st.gpu_seconds*1e3:.0f} ms)", flush=True)
        os.environ.pop("NTHIP_TUNE_NO_BGZF", None)
    os.remove(bg)
os.remove(path)
