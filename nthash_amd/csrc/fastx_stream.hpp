// fastx_stream.hpp -- host side of the FASTQ/FASTA path (included by capi_fastx.hip):
// nthip_kmer_hash_spans, nthip_fastx_index and the streaming driver
// nthip_fastx_kmer_hash_file (reader threads -> pinned buffers -> copy stream -> index + hash).  The reader threads pread a
// plain file; a gzip file is inflated by one of them (gzread), a BGZF file block-wise by all of them (zlib through dlopen).
#pragma once

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "fastx_kernels.hpp"

extern "C" int nthip_kmer_hash_spans(nthip_ctx* c, const char* d_buf, uint64_t buf_bytes, const uint64_t* d_starts,
                                     const uint64_t* d_ends, uint64_t n_reads, uint16_t k16, uint8_t m8,
                                     const nthip_out* out, uint64_t* total_out, uint32_t flags)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  if (!out || !out->hashes) return fail(NTHIP_ERR_ARG, "out->hashes is NULL");
  if (n_reads && (!d_buf || !d_starts || !d_ends)) return fail(NTHIP_ERR_ARG, "buffer / spans are NULL");
  if (flags & NTHIP_HOST_INPUT) return fail(NTHIP_ERR_UNSUPPORTED, "spans are device pointers");
  if (out->fwd || out->rev) return fail(NTHIP_ERR_UNSUPPORTED, "strand outputs are not available for spans");
  const uint32_t k = k16, m = m8;
  if (k == 0) return fail(NTHIP_ERR_ARG, "k must be greater than 0");
  if (k < 3) return fail(NTHIP_ERR_UNSUPPORTED, "k < 3 is undefined in the reference (src/kmer.cpp:47)");
  if (m == 0) return fail(NTHIP_ERR_UNSUPPORTED, "num_hashes must be >= 1");
  HIPCHK(hipSetDevice(c->device));
  if (total_out) *total_out = 0;
  if (n_reads == 0) return NTHIP_OK;
  Staged st;
  st.seqs = (const uint8_t*)d_buf;
  NTCHK(stage_outputs(c, out, flags, n_reads, m, st));
  uint64_t total = 0;
  bool handled = false;
  // (spans inside the buffer, start <= end: checked by the path that takes them, before anything is read through them)
  int rc;
  if (flags & NTHIP_OUT_READ_SLOTS) { // one pass: read r's k-mers at the slot its length implies (capi_kmer_reads.hip)
    rc = run_kmer_reads(c, st, d_starts, d_ends, n_reads, buf_bytes, k, m, out->capacity, &total, &handled, nullptr, true);
    if (rc == NTHIP_OK && !handled)
      rc = fail(NTHIP_ERR_UNSUPPORTED, "NTHIP_OUT_READ_SLOTS: short reads (<= 2048 bases) in order only");
  } else {
    rc = run_kmer_ragged(c, st, d_starts, d_ends, n_reads, buf_bytes, k, m, out->capacity, &total, &handled, nullptr,
                         /*checked*/ false);
  }
  if (rc == NTHIP_ERR_CAPACITY && total_out) *total_out = total;
  NTCHK(rc);
  if (!handled) return fail(NTHIP_ERR_UNSUPPORTED, "shape outside the run-split ragged kernel");
  if (total_out) *total_out = total;
  NTCHK(unstage_outputs(c, out, flags, n_reads, m, total, st));
  HIPCHK(hipStreamSynchronize(c->stream));
  return NTHIP_OK;
}

extern "C" int nthip_seed_hash_spans(nthip_ctx* c, const char* d_buf, uint64_t buf_bytes, const uint64_t* d_starts,
                                     const uint64_t* d_ends, uint64_t n_reads, const nthip_seeds* sd, uint8_t m28,
                                     const nthip_out* out, uint64_t* total_out, uint32_t flags)
{
  if (!c || !sd) return fail(NTHIP_ERR_ARG, "ctx/seeds is NULL");
  if (!out || !out->hashes) return fail(NTHIP_ERR_ARG, "out->hashes is NULL");
  if (n_reads && (!d_buf || !d_starts || !d_ends)) return fail(NTHIP_ERR_ARG, "buffer / spans are NULL");
  if (flags & NTHIP_HOST_INPUT) return fail(NTHIP_ERR_UNSUPPORTED, "spans are device pointers");
  const uint32_t m2 = m28;
  if (m2 == 0) return fail(NTHIP_ERR_UNSUPPORTED, "num_hashes_per_seed must be >= 1");
  HIPCHK(hipSetDevice(c->device));
  if (total_out) *total_out = 0;
  if (n_reads == 0) return NTHIP_OK;
  const uint32_t per = sd->n_seeds * m2;
  uint64_t max_len = 0; // spans inside the buffer, start <= end; and how long the longest one is
  NTCHK(check_offsets_device(c, d_starts, d_ends, n_reads, buf_bytes, false, &max_len));
  if (out->pos && max_len > 0xFFFFFFFFull)
    return fail(NTHIP_ERR_UNSUPPORTED, "out->pos is 32 bits wide: a read of %llu bases cannot report its positions",
                (unsigned long long)max_len);
  Staged st;
  st.seqs = (const uint8_t*)d_buf;
  st.offsets = d_starts;
  NTCHK(stage_outputs(c, out, flags, n_reads, per, st, sd->n_seeds));
  nthip_reads rd = {d_buf, d_starts, n_reads, 0, 0};
  uint64_t total = 0;
  int rc = NTHIP_OK;
  bool long_done = false;
  // (the pieces only pay for chromosome-sized reads: a FASTQ chunk of short reads goes straight to the variable-length path)
  if (!(flags & NTHIP_FORCE_GENERAL) && max_len >= SEED_LONG_MIN)
    rc = run_seed_long(c, st, &rd, sd, m2, out->capacity, &total, &long_done, d_ends);
  if (rc == NTHIP_OK && !long_done) rc = run_seed_general(c, st, &rd, sd, m2, out->capacity, &total, d_ends);
  if (rc == NTHIP_ERR_CAPACITY && total_out) *total_out = total;
  NTCHK(rc);
  if (total_out) *total_out = total;
  NTCHK(unstage_outputs(c, out, flags, n_reads, per, total, st, sd->n_seeds));
  HIPCHK(hipStreamSynchronize(c->stream));
  return NTHIP_OK;
}

extern "C" int nthip_fastx_index(nthip_ctx* c, const char* d_buf, uint64_t n_bytes, uint32_t format,
                                 uint64_t* d_starts, uint64_t* d_ends, uint64_t capacity, uint64_t* n_records,
                                 uint64_t* consumed, int* malformed)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  if (format != NTHIP_FASTQ && format != NTHIP_FASTA) return fail(NTHIP_ERR_ARG, "format must be NTHIP_FASTQ or NTHIP_FASTA");
  if (n_bytes && (!d_buf || !d_starts || !d_ends)) return fail(NTHIP_ERR_ARG, "buffer / span arrays are NULL");
  HIPCHK(hipSetDevice(c->device));
  if (n_records) *n_records = 0;
  if (consumed) *consumed = 0;
  if (malformed) *malformed = 0;
  if (n_bytes == 0) return NTHIP_OK;
  const uint64_t nb = (n_bytes + FX_BLOCK_BYTES - 1) / FX_BLOCK_BYTES;
  const uint64_t nbs = (nb + SCAN_TILE - 1) / SCAN_TILE;
  NTCHK(ensure_scratch(c, 2 * nb + nbs + 16));
  uint64_t* d_cnt = c->d_scratch;
  uint64_t* d_base = c->d_scratch + nb;
  uint64_t* d_sums = c->d_scratch + 2 * nb;
  uint64_t* d_total = (uint64_t*)(c->d_small + 8);
  FastxIndexOut* d_out = (FastxIndexOut*)(c->d_small + 32);
  HIPCHK(hipMemsetAsync(d_out, 0, sizeof(FastxIndexOut), c->stream));
  hipLaunchKernelGGL(fastx_count_kernel, dim3((unsigned)nb), dim3(FX_THREADS), 0, c->stream, (const uint8_t*)d_buf,
                     n_bytes, d_cnt);
  NTCHK(device_exclusive_scan(c, d_cnt, d_base, nb, d_sums, d_total));
  prof_begin(c, "fastx_index_kernel");
  hipLaunchKernelGGL(fastx_index_kernel, dim3((unsigned)nb), dim3(FX_THREADS), 0, c->stream, (const uint8_t*)d_buf,
                     n_bytes, d_base, d_total, format, (uint8_t)(format == NTHIP_FASTQ ? '@' : '>'), d_starts, d_ends,
                     capacity, d_out);
  prof_end(c);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(c->h_small + 8, d_total, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(c->h_small + 32, d_out, sizeof(FastxIndexOut), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  uint64_t newlines = 0;
  FastxIndexOut ho;
  memcpy(&newlines, c->h_small + 8, 8);
  memcpy(&ho, c->h_small + 32, sizeof ho);
  const uint64_t nrec = newlines / format;
  if (n_records) *n_records = nrec;
  if (consumed) *consumed = ho.consumed;
  if (malformed) *malformed = (int)ho.malformed;
  if (nrec > capacity)
    return fail(NTHIP_ERR_CAPACITY, "span capacity %llu records < %llu in the chunk", (unsigned long long)capacity,
                (unsigned long long)nrec);
  return NTHIP_OK;
}

extern "C" int nthip_fasta_compact(nthip_ctx* c, const char* d_raw, uint64_t n_bytes, char* d_seqs,
                                   uint64_t* d_offsets, uint64_t capacity, uint64_t* n_records, uint64_t* seq_bytes)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  if (n_bytes && (!d_raw || !d_seqs || !d_offsets)) return fail(NTHIP_ERR_ARG, "buffers are NULL");
  HIPCHK(hipSetDevice(c->device));
  if (n_records) *n_records = 0;
  if (seq_bytes) *seq_bytes = 0;
  if (n_bytes == 0) return NTHIP_OK;
  char first = 0;
  HIPCHK(hipMemcpyAsync(&first, d_raw, 1, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (first != '>') return fail(NTHIP_ERR_ARG, "FASTA text does not begin with '>'");
  const uint64_t nb = (n_bytes + FX_BLOCK_BYTES - 1) / FX_BLOCK_BYTES;
  const uint64_t nbs = (nb + SCAN_TILE - 1) / SCAN_TILE;
  NTCHK(ensure_scratch(c, 5 * nb + nbs + 32));
  uint64_t* d_kept = c->d_scratch;
  uint64_t* d_hdrs = c->d_scratch + nb;
  uint64_t* d_kept_base = c->d_scratch + 2 * nb;
  uint64_t* d_hdr_base = c->d_scratch + 3 * nb;
  uint32_t* d_last = (uint32_t*)(c->d_scratch + 4 * nb);
  uint32_t* d_carry = d_last + nb;
  uint64_t* d_sums = c->d_scratch + 5 * nb + 8;
  uint64_t* d_total = (uint64_t*)(c->d_small + 8);
  const uint8_t* raw = (const uint8_t*)d_raw;
  hipLaunchKernelGGL(fasta_events_kernel, dim3((unsigned)nb), dim3(FX_THREADS), 0, c->stream, raw, n_bytes, d_last);
  hipLaunchKernelGGL(fasta_carry_kernel, dim3(1), dim3(1024), 0, c->stream, d_last, nb, d_carry);
  hipLaunchKernelGGL(fasta_count_kernel, dim3((unsigned)nb), dim3(FX_THREADS), 0, c->stream, raw, n_bytes, d_carry,
                     d_kept, d_hdrs);
  HIPCHK(hipGetLastError());
  NTCHK(device_exclusive_scan(c, d_kept, d_kept_base, nb, d_sums, d_total));
  HIPCHK(hipMemcpyAsync(c->h_small + 8, d_total, 8, hipMemcpyDeviceToHost, c->stream));
  NTCHK(device_exclusive_scan(c, d_hdrs, d_hdr_base, nb, d_sums, d_total));
  HIPCHK(hipMemcpyAsync(c->h_small + 40, d_total, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  uint64_t kept = 0, nrec = 0;
  memcpy(&kept, c->h_small + 8, 8);
  memcpy(&nrec, c->h_small + 40, 8);
  if (n_records) *n_records = nrec;
  if (seq_bytes) *seq_bytes = kept;
  if (nrec > capacity)
    return fail(NTHIP_ERR_CAPACITY, "offset capacity %llu records < %llu in the text", (unsigned long long)capacity,
                (unsigned long long)nrec);
  prof_begin(c, "fasta_scatter_kernel");
  hipLaunchKernelGGL(fasta_scatter_kernel, dim3((unsigned)nb), dim3(FX_THREADS), 0, c->stream, raw, n_bytes, d_carry,
                     d_kept_base, d_hdr_base, (uint8_t*)d_seqs, d_offsets, capacity);
  prof_end(c);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(d_offsets + nrec, &kept, 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return NTHIP_OK;
}

namespace {


// ---- gzip input -------------------------------------------------------------------------------------------------------------
// A file that begins with the gzip magic (1f 8b) is inflated on the host by the system's zlib, loaded at run time
// (dlopen("libz.so.1"): the library keeps libamdhip64 as its only link-time dependency, and a host without zlib gets
// NTHIP_ERR_UNSUPPORTED for .gz files, nothing else).  gzread follows concatenated members (bgzip, cat a.gz b.gz).
// A deflate stream has no place to seek to, so ONE host thread feeds the pinned ring; everything behind it is the plain path.
struct ZLib {
  void* (*gzdopen)(int, const char*) = nullptr;
  int (*gzbuffer)(void*, unsigned) = nullptr;
  int (*gzread)(void*, void*, unsigned) = nullptr;
  int (*gzclose)(void*) = nullptr;
  const char* (*gzerror)(void*, int*) = nullptr;
  // block-wise (BGZF) inflate: the stream interface, on zlib's own z_stream layout (stable through zlib 1.x; LP64)
  const char* (*zlibVersion)() = nullptr;
  int (*inflateInit2_)(void*, int, const char*, int) = nullptr;
  int (*inflate)(void*, int) = nullptr;
  int (*inflateReset)(void*) = nullptr;
  int (*inflateEnd)(void*) = nullptr;
  unsigned long (*crc32)(unsigned long, const unsigned char*, unsigned) = nullptr;
  bool ok = false, ok_raw = false;
};
struct NtZStream { // == z_stream of zlib.h
  const unsigned char* next_in;
  unsigned avail_in;
  unsigned long total_in;
  unsigned char* next_out;
  unsigned avail_out;
  unsigned long total_out;
  const char* msg;
  void* state;
  void* zalloc;
  void* zfree;
  void* opaque;
  int data_type;
  unsigned long adler;
  unsigned long reserved;
};
static_assert(sizeof(NtZStream) == 112, "z_stream layout (LP64)");

inline const ZLib& zlib_api()
{
  static const ZLib z = [] {
    ZLib r;
    void* h = dlopen("libz.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("libz.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return r;
    r.gzdopen = (void* (*)(int, const char*))dlsym(h, "gzdopen");
    r.gzbuffer = (int (*)(void*, unsigned))dlsym(h, "gzbuffer");
    r.gzread = (int (*)(void*, void*, unsigned))dlsym(h, "gzread");
    r.gzclose = (int (*)(void*))dlsym(h, "gzclose");
    r.gzerror = (const char* (*)(void*, int*))dlsym(h, "gzerror");
    r.ok = r.gzdopen && r.gzbuffer && r.gzread && r.gzclose && r.gzerror;
    r.zlibVersion = (const char* (*)())dlsym(h, "zlibVersion");
    r.inflateInit2_ = (int (*)(void*, int, const char*, int))dlsym(h, "inflateInit2_");
    r.inflate = (int (*)(void*, int))dlsym(h, "inflate");
    r.inflateReset = (int (*)(void*))dlsym(h, "inflateReset");
    r.inflateEnd = (int (*)(void*))dlsym(h, "inflateEnd");
    r.crc32 = (unsigned long (*)(unsigned long, const unsigned char*, unsigned))dlsym(h, "crc32");
    r.ok_raw = r.zlibVersion && r.inflateInit2_ && r.inflate && r.inflateReset && r.inflateEnd && r.crc32 &&
               r.zlibVersion()[0] == '1';
    return r;
  }();
  return z;
}

// does the file behind fd begin with the gzip magic?  (< 0: read error)
inline int fd_is_gzip(int fd)
{
  uint8_t magic[2] = {0, 0};
  const ssize_t r = pread(fd, magic, 2, 0);
  if (r < 0) return -1;
  return r == 2 && magic[0] == 0x1f && magic[1] == 0x8b ? 1 : 0;
}

// an inflating reader over a dup of fd (the caller keeps fd); read() fills dst with up to want bytes, fewer only at the
// end of the stream; < 0: corrupt or truncated input
struct GzSource {
  void* gz = nullptr;
  std::string err;
  int open_fd(int fd)
  {
    const ZLib& z = zlib_api();
    if (!z.ok) return NTHIP_ERR_UNSUPPORTED;
    const int d = dup(fd);
    if (d < 0) return NTHIP_ERR_ARG;
    if (lseek(d, 0, SEEK_SET) < 0) { ::close(d); return NTHIP_ERR_ARG; }
    gz = z.gzdopen(d, "rb");
    if (!gz) { ::close(d); return NTHIP_ERR_ARG; }
    (void)z.gzbuffer(gz, 1u << 20);
    return NTHIP_OK;
  }
  int64_t read(uint8_t* dst, uint64_t want)
  {
    const ZLib& z = zlib_api();
    uint64_t got = 0;
    while (got < want) {
      const unsigned piece = (unsigned)(want - got < (1ull << 30) ? want - got : (1ull << 30));
      const int r = z.gzread(gz, dst + got, piece);
      if (r < 0) { note(); return -1; }
      if (r == 0) break;
      got += (uint64_t)r;
    }
    if (got < want) { // the end of the stream: a file cut short ends here too, and says so (Z_BUF_ERROR)
      int num = 0;
      (void)z.gzerror(gz, &num);
      if (num < 0) { note(); return -1; }
    }
    return (int64_t)got;
  }
  void note()
  {
    int num = 0;
    const char* msg = zlib_api().gzerror(gz, &num);
    err = msg && *msg ? msg : "truncated or corrupt gzip stream";
    if (num == -5 /* Z_BUF_ERROR */) err = "gzip stream ends before its end-of-stream mark (truncated file)";
  }
  ~GzSource()
  {
    if (gz) (void)zlib_api().gzclose(gz);
  }
};

// ---- BGZF (bgzip, samtools/htslib): gzip members of at most 64 KiB, each with its compressed size in an extra field
// ('B' 'C' 2 BSIZE) and its inflated size in the trailer -- the one kind of gzip file whose pieces can be inflated
// independently, by as many threads as the host gives.
constexpr uint32_t BGZF_HEAD = 18, BGZF_TAIL = 8;
struct BgzfBlock {
  uint64_t off;      // of the block in the file
  uint32_t csize;    // whole block: header + deflate data + crc32 + isize
  uint32_t isize;    // inflated bytes
  uint64_t out_off;  // where they go in the chunk
  uint32_t head;     // bytes in front of the deflate data: 12 + XLEN (18 when 'BC' is the only extra subfield)
};
// the block header at `off`: 1 a BGZF block (csize set), 0 something else, < 0 read error / file ends inside it
inline int bgzf_block_at(int fd, uint64_t file_size, uint64_t off, BgzfBlock* b)
{
  // (the extra field may hold other subfields next to 'BC' -- RFC 1952 2.3.1.1; a block-0 probe that only knew XLEN == 6
  // chose the BGZF path for files whose later blocks then failed in the middle of the stream: ADVICE r04)
  uint8_t h[12 + 256];
  if (off + BGZF_HEAD + BGZF_TAIL > file_size) return -1;
  if (pread(fd, h, 12, (off_t)off) != 12) return -1;
  if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || (h[3] & 4) == 0 || (h[3] & ~4) != 0) return 0; // (FEXTRA alone, as bgzip writes)
  const uint32_t xlen = (uint32_t)h[10] | (uint32_t)h[11] << 8;
  if (xlen < 6 || xlen > 256 || off + 12 + xlen + BGZF_TAIL > file_size) return 0;
  if (pread(fd, h + 12, xlen, (off_t)(off + 12)) != (ssize_t)xlen) return -1;
  uint32_t csize = 0;
  for (uint32_t p = 0; p + 4 <= xlen;) {
    const uint8_t* f = h + 12 + p;
    const uint32_t flen = (uint32_t)f[2] | (uint32_t)f[3] << 8;
    if (p + 4 + flen > xlen) return 0;
    if (f[0] == 'B' && f[1] == 'C' && flen == 2) csize = ((uint32_t)f[4] | (uint32_t)f[5] << 8) + 1u;
    p += 4 + flen;
  }
  if (csize == 0) return 0;
  b->head = 12 + xlen;
  if (csize < b->head + BGZF_TAIL || off + csize > file_size) return -1;
  uint8_t t[4];
  if (pread(fd, t, 4, (off_t)(off + csize - 4)) != 4) return -1;
  b->off = off;
  b->csize = csize;
  b->isize = (uint32_t)t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
  return b->isize <= 65536u ? 1 : -1;
}
// blocks [lo, hi) of `bl` into dst (at their out_off); false: a block is corrupt
inline bool bgzf_inflate_blocks(int fd, const std::vector<BgzfBlock>& bl, size_t lo, size_t hi, uint8_t* dst)
{
  const ZLib& z = zlib_api();
  NtZStream zs;
  memset(&zs, 0, sizeof zs);
  if (z.inflateInit2_(&zs, -15, z.zlibVersion(), (int)sizeof zs) != 0) return false;
  std::vector<uint8_t> in(65536 + 64);
  bool ok = true;
  for (size_t i = lo; i < hi && ok; ++i) {
    const BgzfBlock& b = bl[i];
    uint64_t done = 0;
    while (done < b.csize) {
      const ssize_t r = pread(fd, in.data() + done, b.csize - done, (off_t)(b.off + done));
      if (r <= 0) { ok = false; break; }
      done += (uint64_t)r;
    }
    if (!ok) break;
    if (z.inflateReset(&zs) != 0) { ok = false; break; }
    zs.next_in = in.data() + b.head;
    zs.avail_in = b.csize - b.head - BGZF_TAIL;
    zs.next_out = dst + b.out_off;
    zs.avail_out = b.isize;
    const int rc = z.inflate(&zs, 4 /* Z_FINISH */);
    const uint8_t* t = in.data() + b.csize - BGZF_TAIL;
    const uint32_t crc = (uint32_t)t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
    ok = rc == 1 /* Z_STREAM_END */ && zs.avail_out == 0 && zs.avail_in == 0 &&
         (uint32_t)z.crc32(0, dst + b.out_off, b.isize) == crc;
  }
  (void)z.inflateEnd(&zs);
  return ok;
}

constexpr uint64_t FXS_HEAD = 16ull << 20; // room in front of a chunk for the carried-over tail of the previous one

struct FxReader {
  // the reader thread fills pinned[j & 1] with file bytes [j*S, (j+1)*S) for j = 0, 1, ... (or with the j-th of
  // `ranges`: record-aligned pieces chosen by the caller -- the multi-device driver)
  int fd = -1;
  uint64_t file_size = 0, chunk = 0, n_chunks = 0;
  const std::vector<FastxRange>* ranges = nullptr;
  uint8_t* pinned[2] = {nullptr, nullptr};
  uint64_t filled[2] = {0, 0};
  std::mutex mu;
  std::condition_variable cv;
  uint64_t next_ready = 0;  // chunks [0, next_ready) have been read
  uint64_t next_free = 2;   // chunks [0, next_free) may be read (their buffer is free)
  bool failed = false, stop = false;
  double read_seconds = 0;
  unsigned n_threads = 0;
  std::thread th;
  GzSource* gz = nullptr; // gzip input: n_chunks is not known until the stream ends (set, under mu, with the last chunk)
  std::string err;

  // chunks of the INFLATED stream; one byte is read ahead so that the chunk that ends the stream is known as the last
  void run_gz()
  {
    uint8_t ahead = 0;
    bool have_ahead = false;
    for (uint64_t j = 0;; ++j) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || j < next_free; });
        if (stop) return;
      }
      const auto t0 = std::chrono::steady_clock::now();
      uint8_t* dst = pinned[j & 1];
      uint64_t len = 0;
      bool ok = true, last = false;
      if (have_ahead) { dst[0] = ahead; len = 1; have_ahead = false; }
      const int64_t r = gz->read(dst + len, chunk - len);
      if (r < 0) ok = false;
      else {
        len += (uint64_t)r;
        if (len < chunk) last = true;
        else {
          const int64_t one = gz->read(&ahead, 1);
          if (one < 0) ok = false;
          else if (one == 0) last = true;
          else have_ahead = true;
        }
      }
      read_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      {
        std::lock_guard<std::mutex> lk(mu);
        filled[j & 1] = len;
        if (!ok) { failed = true; err = gz->err; }
        if (last) n_chunks = len ? j + 1 : j; // (len == 0 only for j == 0: an empty stream)
        next_ready = j + 1;
      }
      cv.notify_all();
      if (!ok || last) return;
    }
  }

  // BGZF: whole blocks up to `chunk` inflated bytes per chunk, the blocks of a chunk shared out over the reader threads
  bool bgzf = false;
  void run_bgzf()
  {
    const unsigned n_thr = n_threads ? n_threads : 16u;
    uint64_t pos = 0;
    std::vector<BgzfBlock> bl;
    for (uint64_t j = 0;; ++j) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || j < next_free; });
        if (stop) return;
      }
      const auto t0 = std::chrono::steady_clock::now();
      bl.clear();
      uint64_t len = 0;
      bool ok = true;
      while (pos < file_size) { // (blocks without bytes -- the end-of-file marker -- always fit)
        BgzfBlock b;
        const int kind = bgzf_block_at(fd, file_size, pos, &b);
        if (kind <= 0) {
          ok = false;
          err = kind == 0 ? "a member that is not a BGZF block follows BGZF blocks" : "the file ends inside a BGZF block (truncated)";
          break;
        }
        if (len + b.isize > chunk && !bl.empty()) break;
        b.out_off = len;
        len += b.isize;
        pos += b.csize;
        bl.push_back(b);
      }
      const bool last = pos >= file_size;
      if (ok) {
        std::atomic<bool> good{true};
        std::vector<std::thread> ws;
        const size_t per = (bl.size() + n_thr - 1) / n_thr;
        for (unsigned t = 0; t < n_thr && per; ++t) {
          const size_t lo = (size_t)t * per, hi = lo + per < bl.size() ? lo + per : bl.size();
          if (lo >= hi) break;
          ws.emplace_back([&, lo, hi] {
            if (!bgzf_inflate_blocks(fd, bl, lo, hi, pinned[j & 1])) good = false;
          });
        }
        for (auto& w : ws) w.join();
        if (!good) { ok = false; err = "corrupt BGZF block (inflate / crc32 / size mismatch)"; }
      }
      read_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      {
        std::lock_guard<std::mutex> lk(mu);
        filled[j & 1] = len;
        if (!ok) failed = true;
        if (last) n_chunks = len ? j + 1 : j;
        next_ready = j + 1;
      }
      cv.notify_all();
      if (!ok || last) return;
    }
  }

  void run()
  {
    if (bgzf) return run_bgzf();
    if (gz) return run_gz();
    const unsigned n_thr = n_threads ? n_threads : 8u; // (NTHIP_TUNE_READ_THREADS: A/B knob)
    for (uint64_t j = 0; j < n_chunks; ++j) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || j < next_free; });
        if (stop) return;
      }
      const uint64_t off = ranges ? (*ranges)[j].off : j * chunk;
      const uint64_t len = ranges ? (*ranges)[j].len : off + chunk <= file_size ? chunk : file_size - off;
      const auto t0 = std::chrono::steady_clock::now();
      // several preads at once: one thread does not reach the page-cache copy rate PCIe can take
      std::atomic<bool> ok{true};
      std::vector<std::thread> ws;
      const uint64_t part = (len + n_thr - 1) / n_thr;
      for (unsigned t = 0; t < n_thr; ++t) {
        const uint64_t a = (uint64_t)t * part, b = a + part < len ? a + part : len;
        if (a >= b) break;
        ws.emplace_back([&, a, b] {
          uint64_t done = a;
          while (done < b) {
            const ssize_t r = pread(fd, pinned[j & 1] + done, b - done, (off_t)(off + done));
            if (r <= 0) { ok = false; return; }
            done += (uint64_t)r;
          }
        });
      }
      for (auto& w : ws) w.join();
      read_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      {
        std::lock_guard<std::mutex> lk(mu);
        filled[j & 1] = len;
        if (!ok) failed = true;
        next_ready = j + 1;
      }
      cv.notify_all();
      if (!ok) return;
    }
  }
};

// multi-line FASTA (genomes): whole file -> HBM -> compact -> one hash call -> one callback
int fasta_multiline_file(nthip_ctx* c, const char* path, uint16_t k, uint8_t m, nthip_fastx_fn fn, void* user,
                         nthip_fastx_stats* stats)
{
  if (k == 0) return fail(NTHIP_ERR_ARG, "k must be greater than 0");
  if (k < 3 || m == 0) return fail(NTHIP_ERR_UNSUPPORTED, "k < 3 / m == 0 are undefined in the reference");
  HIPCHK(hipSetDevice(c->device));
  const auto t_begin = std::chrono::steady_clock::now();
  if (stats) memset(stats, 0, sizeof *stats);
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return fail(NTHIP_ERR_ARG, "cannot open %s", path);
  struct stat sb;
  if (fstat(fd, &sb) != 0) { close(fd); return fail(NTHIP_ERR_ARG, "cannot stat %s", path); }
  uint64_t size = (uint64_t)sb.st_size;
  if (stats) stats->file_bytes = size;
  if (size == 0) { close(fd); return NTHIP_OK; }
  const uint64_t piece = 64ull << 20;
  // gzip input: the whole file is one batch anyway -- inflated into host memory first (its size is not known before)
  std::vector<uint8_t> inflated;
  std::unique_ptr<uint8_t[]> inflated_raw;
  const uint8_t* inflated_ptr = nullptr;
  const int is_gz = fd_is_gzip(fd);
  if (is_gz < 0) { close(fd); return fail(NTHIP_ERR_ARG, "read error on %s", path); }
  BgzfBlock b0;
  const char* no_bgzf = getenv("NTHIP_TUNE_NO_BGZF");
  if (is_gz && zlib_api().ok_raw && !(no_bgzf && no_bgzf[0] == '1') && bgzf_block_at(fd, size, 0, &b0) == 1) {
    // BGZF (a bgzipped genome): the sizes are in the block headers -- one walk over them, then every block inflated in
    // its place by the reader threads
    std::vector<BgzfBlock> bl;
    uint64_t pos = 0, have = 0;
    while (pos < size) {
      BgzfBlock b;
      const int kind = bgzf_block_at(fd, size, pos, &b);
      if (kind <= 0) {
        close(fd);
        return fail(NTHIP_ERR_ARG, kind == 0 ? "%s: a member that is not a BGZF block follows BGZF blocks"
                                             : "%s: the file ends inside a BGZF block (truncated)", path);
      }
      b.out_off = have;
      have += b.isize;
      pos += b.csize;
      bl.push_back(b);
    }
    // (not a vector: its resize would touch every page from one thread before the first block is inflated)
    inflated_raw.reset(new uint8_t[have ? have : 1]);
    inflated_ptr = inflated_raw.get();
    const unsigned n_thr = c->tune.read_threads ? c->tune.read_threads : 16u;
    std::atomic<bool> good{true};
    std::vector<std::thread> ws;
    const size_t per = (bl.size() + n_thr - 1) / n_thr;
    for (unsigned t = 0; t < n_thr && per; ++t) {
      const size_t lo = (size_t)t * per, hi = lo + per < bl.size() ? lo + per : bl.size();
      if (lo >= hi) break;
      ws.emplace_back([&, lo, hi] {
        if (!bgzf_inflate_blocks(fd, bl, lo, hi, inflated_raw.get())) good = false;
      });
    }
    for (auto& w : ws) w.join();
    if (!good) { close(fd); return fail(NTHIP_ERR_ARG, "%s: corrupt BGZF block (inflate / crc32 / size mismatch)", path); }
    size = have;
    if (size == 0) { close(fd); return NTHIP_OK; }
  } else if (is_gz) {
    GzSource src;
    const int zrc = src.open_fd(fd);
    if (zrc != NTHIP_OK) {
      close(fd);
      return fail(zrc, zrc == NTHIP_ERR_UNSUPPORTED ? "%s is gzip-compressed and libz.so.1 could not be loaded" : "cannot open %s as a gzip stream", path);
    }
    uint64_t have = 0;
    for (;;) {
      inflated.resize(have + piece);
      const int64_t r = src.read(inflated.data() + have, piece);
      if (r < 0) { close(fd); return fail(NTHIP_ERR_ARG, "%s: %s", path, src.err.c_str()); }
      have += (uint64_t)r;
      if ((uint64_t)r < piece) break;
    }
    inflated.resize(have);
    inflated_ptr = inflated.data();
    size = have;
    if (size == 0) { close(fd); return NTHIP_OK; }
  }
  uint8_t* pinned[2] = {nullptr, nullptr};
  uint8_t *d_raw = nullptr, *d_seqs = nullptr;
  uint64_t *d_offsets = nullptr, *d_hashes = nullptr, *d_counts = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};
  int rc = NTHIP_OK;
  double read_s = 0, gpu_s = 0;
  auto cleanup = [&]() {
    (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < 2; ++i) {
      if (pinned[i]) (void)hipHostFree(pinned[i]);
      if (ev[i]) (void)hipEventDestroy(ev[i]);
    }
    for (void* p : {(void*)d_raw, (void*)d_seqs, (void*)d_offsets, (void*)d_hashes, (void*)d_counts})
      if (p) (void)hipFree(p);
    close(fd);
  };
#define FA_TRY(expr) \
  do { \
    hipError_t e_ = (expr); \
    if (e_ != hipSuccess) { rc = fail(NTHIP_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); cleanup(); return rc; } \
  } while (0)
  FA_TRY(hipMalloc((void**)&d_raw, size + 64));
  for (int i = 0; i < 2; ++i) {
    FA_TRY(hipHostMalloc((void**)&pinned[i], piece, hipHostMallocDefault));
    FA_TRY(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
  }
  uint64_t n_bytes = size;
  {
    const auto t0 = std::chrono::steady_clock::now();
    uint8_t last_byte = 0;
    uint64_t i = 0;
    for (uint64_t off = 0; off < size; off += piece, ++i) {
      const uint64_t len = off + piece <= size ? piece : size - off;
      if (i >= 2) FA_TRY(hipEventSynchronize(ev[i & 1])); // the upload that used this pinned buffer
      uint64_t done = is_gz ? len : 0;
      if (is_gz) memcpy(pinned[i & 1], inflated_ptr + off, len);
      while (done < len) {
        const ssize_t r = pread(fd, pinned[i & 1] + done, len - done, (off_t)(off + done));
        if (r <= 0) { rc = fail(NTHIP_ERR_ARG, "read error on %s", path); cleanup(); return rc; }
        done += (uint64_t)r;
      }
      last_byte = pinned[i & 1][len - 1];
      FA_TRY(hipMemcpyAsync(d_raw + off, pinned[i & 1], len, hipMemcpyHostToDevice, c->stream));
      FA_TRY(hipEventRecord(ev[i & 1], c->stream));
    }
    if (last_byte != '\n') { // close the last line
      const char nl = '\n';
      FA_TRY(hipMemcpyAsync(d_raw + size, &nl, 1, hipMemcpyHostToDevice, c->stream));
      n_bytes = size + 1;
    }
    FA_TRY(hipStreamSynchronize(c->stream));
    read_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  const auto t1 = std::chrono::steady_clock::now();
  // records: at most one per 2 bytes (">\n"); count them first with a zero-capacity probe
  uint64_t n_rec = 0, seq_bytes = 0;
  FA_TRY(hipMalloc((void**)&d_seqs, n_bytes + 64));
  FA_TRY(hipMalloc((void**)&d_offsets, 16));
  rc = nthip_fasta_compact(c, (const char*)d_raw, n_bytes, (char*)d_seqs, d_offsets, 0, &n_rec, &seq_bytes);
  if (rc != NTHIP_OK && rc != NTHIP_ERR_CAPACITY) { cleanup(); return rc; }
  (void)hipFree(d_offsets);
  d_offsets = nullptr;
  FA_TRY(hipMalloc((void**)&d_offsets, (n_rec + 1) * sizeof(uint64_t)));
  rc = nthip_fasta_compact(c, (const char*)d_raw, n_bytes, (char*)d_seqs, d_offsets, n_rec, &n_rec, &seq_bytes);
  if (rc != NTHIP_OK) { cleanup(); return rc; }
  (void)hipFree(d_raw);
  d_raw = nullptr;
  uint64_t n_kmers = 0;
  if (n_rec) {
    const uint64_t cap = seq_bytes ? seq_bytes : 1;
    FA_TRY(hipMalloc((void**)&d_hashes, cap * (uint64_t)m * sizeof(uint64_t)));
    FA_TRY(hipMalloc((void**)&d_counts, n_rec * sizeof(uint64_t)));
    nthip_reads rdx = {(const char*)d_seqs, d_offsets, n_rec, 0, 0};
    nthip_out out = {d_hashes, cap, d_counts, nullptr, nullptr, nullptr};
    rc = nthip_kmer_hash(c, &rdx, k, m, &out, &n_kmers, 0);
    if (rc != NTHIP_OK) { cleanup(); return rc; }
    if (fn) {
      nthip_fastx_batch b = {n_rec, n_kmers, d_hashes, d_counts, (const char*)d_seqs, d_offsets, d_offsets + 1, 0, c->device, 0};
      if (fn(user, &b) != 0) { rc = fail(NTHIP_ERR_ARG, "stopped by the callback"); cleanup(); return rc; }
    }
  }
#undef FA_TRY
  gpu_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
  cleanup();
  if (stats) {
    stats->reads = n_rec;
    stats->kmers = n_kmers;
    stats->batches = n_rec ? 1 : 0;
    stats->read_seconds = read_s;
    stats->gpu_seconds = gpu_s;
    stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
  }
  return NTHIP_OK;
}

} // namespace

namespace {
// seeds == nullptr: NtHash(k, m); else SeedNtHash(seeds, m = hashes per seed).
// ranges / deliver (the multi-device driver): hash exactly these record-aligned pieces of the file, in this order, and
// hand every batch -- also an empty one -- to deliver(j, batch) instead of fn
int fastx_stream_file(nthip_ctx* c, const char* path, uint32_t format, uint16_t k, uint8_t m, const nthip_seeds* seeds,
                      uint64_t chunk_bytes, nthip_fastx_fn fn, void* user, nthip_fastx_stats* stats,
                      const std::vector<FastxRange>* ranges = nullptr, const FastxDeliver* deliver = nullptr);
} // namespace

int ntamd::host::fastx_stream_ranges(nthip_ctx* c, const char* path, uint32_t format, uint16_t k, uint8_t m,
                                     const nthip_seeds* seeds, uint64_t chunk_bytes, const std::vector<FastxRange>& ranges,
                                     const FastxDeliver& deliver, nthip_fastx_stats* stats)
{
  return fastx_stream_file(c, path, format, k, m, seeds, chunk_bytes, nullptr, nullptr, stats, &ranges, &deliver);
}

int ntamd::host::fastx_file_is_gzip(int fd) { return fd_is_gzip(fd); }

// First record start at or after byte `pos` of a FASTQ / single-line FASTA file (pos > 0), found on the host from the
// line structure alone: FASTA -- a line that begins with '>'; FASTQ -- a line that begins with '@' whose third line
// begins with '+' and whose second and fourth lines are equally long ('@' and '+' are legal quality characters, so the
// first character alone does not say).  file_size when no record starts after pos.  < 0: read error.
int64_t ntamd::host::fastx_find_record_start(int fd, uint64_t file_size, uint64_t pos, uint32_t format)
{
  if (pos == 0) return 0;
  if (pos >= file_size) return (int64_t)file_size;
  std::vector<uint8_t> buf;
  for (uint64_t window = 1ull << 20; ; window *= 4) {
    const uint64_t from = pos - 1;
    const uint64_t want = from + window <= file_size ? window : file_size - from;
    buf.resize(want);
    uint64_t done = 0;
    while (done < want) {
      const ssize_t r = pread(fd, buf.data() + done, want - done, (off_t)(from + done));
      if (r <= 0) return -1;
      done += (uint64_t)r;
    }
    const bool to_eof = from + want == file_size;
    // line starts inside the window: after every '\n' at or behind `from`
    std::vector<uint64_t> ls;
    for (uint64_t i = 0; i < want; ++i)
      if (buf[i] == '\n' && i + 1 < want) ls.push_back(i + 1);
    auto line_len = [&](size_t li) -> int64_t { // without '\n' and a trailing '\r'; -1: the line's end is not in the window
      const uint64_t b = ls[li];
      uint64_t e;
      if (li + 1 < ls.size()) e = ls[li + 1] - 1;
      else if (to_eof) e = buf[want - 1] == '\n' ? want - 1 : want;
      else return -1;
      if (e > b && buf[e - 1] == '\r') --e;
      return (int64_t)(e - b);
    };
    bool undecided = false;
    for (size_t li = 0; li < ls.size(); ++li) {
      const uint8_t ch = buf[ls[li]];
      if (format == NTHIP_FASTA) {
        if (ch == '>') return (int64_t)(from + ls[li]);
        continue;
      }
      if (ch != '@') continue;
      if (li + 3 >= ls.size() && !to_eof) { undecided = true; break; }
      if (li + 3 >= ls.size()) continue; // fewer than four lines left in the file: not a record
      if (buf[ls[li + 2]] != '+') continue;
      const int64_t l1 = line_len(li + 1), l3 = line_len(li + 3);
      if (l3 < 0) { undecided = true; break; }
      if (l1 == l3) return (int64_t)(from + ls[li]);
    }
    if (to_eof) return (int64_t)file_size; // (decided or not: no record starts before the end of the file)
    if (window >= (1ull << 30)) return -2;  // a "record" of a gigabyte: no boundary near pos -- the caller tries further on
  }
}

extern "C" int nthip_fastx_kmer_hash_file(nthip_ctx* c, const char* path, uint32_t format, uint16_t k, uint8_t m,
                                          uint64_t chunk_bytes, nthip_fastx_fn fn, void* user,
                                          nthip_fastx_stats* stats)
{
  if (!c || !path) return fail(NTHIP_ERR_ARG, "ctx/path is NULL");
  if (format == NTHIP_FASTA_MULTILINE) return fasta_multiline_file(c, path, k, m, fn, user, stats);
  if (k == 0) return fail(NTHIP_ERR_ARG, "k must be greater than 0");
  if (k < 3 || m == 0) return fail(NTHIP_ERR_UNSUPPORTED, "k < 3 / m == 0 are undefined in the reference");
  return fastx_stream_file(c, path, format, k, m, nullptr, chunk_bytes, fn, user, stats);
}

extern "C" int nthip_fastx_seed_hash_file(nthip_ctx* c, const char* path, uint32_t format, const nthip_seeds* seeds,
                                          uint8_t m2, uint64_t chunk_bytes, nthip_fastx_fn fn, void* user,
                                          nthip_fastx_stats* stats)
{
  if (!c || !path || !seeds) return fail(NTHIP_ERR_ARG, "ctx/path/seeds is NULL");
  if (m2 == 0) return fail(NTHIP_ERR_UNSUPPORTED, "num_hashes_per_seed must be >= 1");
  return fastx_stream_file(c, path, format, (uint16_t)seeds->k, m2, seeds, chunk_bytes, fn, user, stats);
}

namespace {
int fastx_stream_file(nthip_ctx* c, const char* path, uint32_t format, uint16_t k, uint8_t m, const nthip_seeds* seeds,
                      uint64_t chunk_bytes, nthip_fastx_fn fn, void* user, nthip_fastx_stats* stats,
                      const std::vector<FastxRange>* ranges, const FastxDeliver* deliver)
{
  if (format != NTHIP_FASTQ && format != NTHIP_FASTA) return fail(NTHIP_ERR_ARG, "format must be NTHIP_FASTQ or NTHIP_FASTA");
  const uint64_t per = seeds ? (uint64_t)seeds->n_seeds * m : (uint64_t)m;
  HIPCHK(hipSetDevice(c->device));
  const auto t_begin = std::chrono::steady_clock::now();
  if (stats) memset(stats, 0, sizeof *stats);
  if (chunk_bytes == 0) chunk_bytes = 256ull << 20;
  if (chunk_bytes < (1ull << 16)) chunk_bytes = 1ull << 16;
  chunk_bytes = (chunk_bytes + 4095) & ~4095ull;

  FxReader rd;
  rd.n_threads = c->tune.read_threads;
  rd.fd = open(path, O_RDONLY);
  if (rd.fd < 0) return fail(NTHIP_ERR_ARG, "cannot open %s", path);
  struct stat sb;
  if (fstat(rd.fd, &sb) != 0) { close(rd.fd); return fail(NTHIP_ERR_ARG, "cannot stat %s", path); }
  rd.file_size = (uint64_t)sb.st_size;
  rd.chunk = chunk_bytes;
  rd.n_chunks = (rd.file_size + chunk_bytes - 1) / chunk_bytes;
  GzSource gz_src; // (declared before the reader starts: it outlives the reader thread)
  const int is_gz = rd.file_size ? fd_is_gzip(rd.fd) : 0;
  if (is_gz < 0) { close(rd.fd); return fail(NTHIP_ERR_ARG, "read error on %s", path); }
  if (is_gz) {
    // chunks of the inflated stream, as many as it turns out to hold
    if (ranges) { close(rd.fd); return fail(NTHIP_ERR_UNSUPPORTED, "%s: byte ranges of a gzip file cannot be read on their own", path); }
    BgzfBlock b0;
    // (NTHIP_TUNE_NO_BGZF=1, read at every call: the one-thread gzread path on a BGZF file too -- the A/B of tools/fastq_bench.py)
    const char* no_bgzf = getenv("NTHIP_TUNE_NO_BGZF");
    if (zlib_api().ok_raw && !(no_bgzf && no_bgzf[0] == '1') && bgzf_block_at(rd.fd, rd.file_size, 0, &b0) == 1) {
      rd.bgzf = true; // blocks inflated side by side
    } else {
      const int zrc = gz_src.open_fd(rd.fd);
      if (zrc != NTHIP_OK) {
        close(rd.fd);
        return fail(zrc, zrc == NTHIP_ERR_UNSUPPORTED ? "%s is gzip-compressed and libz.so.1 could not be loaded" : "cannot open %s as a gzip stream", path);
      }
      rd.gz = &gz_src;
    }
    rd.n_chunks = ~0ull >> 2;
  }
  if (ranges) { // record-aligned pieces: whole records, so a piece may be longer than a chunk by up to one record
    rd.ranges = ranges;
    rd.n_chunks = ranges->size();
    for (const FastxRange& r : *ranges) {
      if (r.off + r.len > rd.file_size) { close(rd.fd); return fail(NTHIP_ERR_ARG, "range outside %s", path); }
      if (r.len > chunk_bytes) chunk_bytes = (r.len + 4095) & ~4095ull;
    }
  }
  if (stats) stats->file_bytes = rd.file_size;
  if (rd.file_size == 0 || rd.n_chunks == 0) { close(rd.fd); return NTHIP_OK; }

  // worst cases inside one piece (head room + chunk + a final newline): an 8-byte record, half of the bytes bases
  const uint64_t piece_max = FXS_HEAD + chunk_bytes + 16;
  const uint64_t cap_reads = piece_max / 8 + 1;
  const uint64_t cap_kmers = format == NTHIP_FASTQ ? piece_max / 2 + 1 : piece_max; // FASTQ: as many quality bytes as bases
  // the buffers live in the context and are reused by later calls that fit them
  auto& fx = c->fx;
  int rc = NTHIP_OK;
  auto cleanup = [&]() {
    {
      std::lock_guard<std::mutex> lk(rd.mu);
      rd.stop = true;
    }
    rd.cv.notify_all();
    if (rd.th.joinable()) rd.th.join();
    (void)hipStreamSynchronize(c->stream);
    if (fx.copy_stream) (void)hipStreamSynchronize(fx.copy_stream);
    close(rd.fd);
  };
#define FX_TRY(expr) \
  do { \
    hipError_t e_ = (expr); \
    if (e_ != hipSuccess) { \
      rc = fail(NTHIP_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
      cleanup(); \
      fastx_buffers_release(c); \
      return rc; \
    } \
  } while (0)
  if (fx.pinned_bytes < chunk_bytes + 16 || fx.raw_bytes < piece_max + 64 || fx.reads_cap < cap_reads ||
      fx.hashes_cap < cap_kmers * per) {
    FX_TRY(hipStreamSynchronize(c->stream));
    fastx_buffers_release(c);
    FX_TRY(hipStreamCreateWithFlags(&fx.copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
      FX_TRY(hipEventCreateWithFlags(&fx.ev_h2d[i], hipEventDisableTiming));
      FX_TRY(hipHostMalloc((void**)&fx.pinned[i], chunk_bytes + 16, hipHostMallocDefault));
      FX_TRY(hipMalloc((void**)&fx.d_raw[i], piece_max + 64));
    }
    FX_TRY(hipMalloc((void**)&fx.d_starts, cap_reads * sizeof(uint64_t)));
    FX_TRY(hipMalloc((void**)&fx.d_ends, cap_reads * sizeof(uint64_t)));
    FX_TRY(hipMalloc((void**)&fx.d_counts, cap_reads * sizeof(uint64_t)));
    FX_TRY(hipMalloc((void**)&fx.d_hashes, cap_kmers * per * sizeof(uint64_t)));
    fx.pinned_bytes = chunk_bytes + 16;
    fx.raw_bytes = piece_max + 64;
    fx.reads_cap = cap_reads;
    fx.hashes_cap = cap_kmers * per;
  }
  rd.pinned[0] = fx.pinned[0];
  rd.pinned[1] = fx.pinned[1];
  uint8_t* const* d_raw = fx.d_raw;
  uint64_t *const d_starts = fx.d_starts, *const d_ends = fx.d_ends, *const d_hashes = fx.d_hashes, *const d_counts = fx.d_counts;
  hipStream_t const copy_stream = fx.copy_stream;
  hipEvent_t const* ev_h2d = fx.ev_h2d;
  rd.th = std::thread([&rd] { rd.run(); });

  uint64_t tail = 0;       // bytes of the piece before chunk j that belong to its first (incomplete) record
  uint64_t first_read = 0;
  double gpu_seconds = 0;
  // piece j = [tail of piece j-1][chunk j]; it lives at d_raw[j & 1] + FXS_HEAD - tail
  auto process = [&](uint64_t j, uint64_t len, bool last) -> int {
    uint8_t* piece = d_raw[j & 1] + FXS_HEAD - tail;
    uint64_t n_bytes = tail + len;
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t n_rec = 0, consumed = 0;
    int malformed = 0;
    NTCHK(nthip_fastx_index(c, (const char*)piece, n_bytes, format, d_starts, d_ends, cap_reads, &n_rec, &consumed,
                            &malformed));
    if (malformed) return fail(NTHIP_ERR_ARG, "%s: malformed record near byte %llu", path,
                               (unsigned long long)(j * chunk_bytes));
    uint64_t n_kmers = 0;
    if (n_rec) {
      nthip_out out = {d_hashes, cap_kmers, d_counts, nullptr, nullptr, nullptr};
      if (seeds) NTCHK(nthip_seed_hash_spans(c, (const char*)piece, n_bytes, d_starts, d_ends, n_rec, seeds, m, &out, &n_kmers, 0));
      else NTCHK(nthip_kmer_hash_spans(c, (const char*)piece, n_bytes, d_starts, d_ends, n_rec, k, m, &out, &n_kmers, 0));
    }
    if (deliver) {
      nthip_fastx_batch b = {n_rec, n_kmers, d_hashes, d_counts, (const char*)piece, d_starts, d_ends, first_read, c->device, 0};
      NTCHK((*deliver)(j, b));
    } else if (fn && n_rec) {
      nthip_fastx_batch b = {n_rec, n_kmers, d_hashes, d_counts, (const char*)piece, d_starts, d_ends, first_read, c->device, 0};
      if (fn(user, &b) != 0) return fail(NTHIP_ERR_ARG, "stopped by the callback");
    }
    if (stats) { stats->reads += n_rec; stats->kmers += n_kmers; stats->batches += 1; }
    first_read += n_rec;
    const uint64_t rest = n_bytes - consumed;
    if (ranges) { // (every piece ends where a record ends)
      if (rest != 0) return fail(NTHIP_ERR_ARG, "%s: a piece does not end on a record boundary near byte %llu", path,
                                 (unsigned long long)((*ranges)[j].off + (*ranges)[j].len));
      HIPCHK(hipStreamSynchronize(c->stream));
    } else if (last) {
      if (rest != 0) return fail(NTHIP_ERR_ARG, "%s: truncated record at the end of the file", path);
    } else {
      if (rest > FXS_HEAD) return fail(NTHIP_ERR_UNSUPPORTED, "%s: record longer than %llu bytes", path,
                                       (unsigned long long)FXS_HEAD);
      // carry the incomplete record in front of the next chunk (its upload writes from FXS_HEAD on)
      if (rest) HIPCHK(hipMemcpyAsync(d_raw[(j + 1) & 1] + FXS_HEAD - rest, piece + consumed, rest,
                                      hipMemcpyDeviceToDevice, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
    }
    tail = rest;
    gpu_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return NTHIP_OK;
  };

  uint64_t lens[2] = {0, 0};
  for (uint64_t j = 0; rc == NTHIP_OK; ++j) {
    uint64_t n_chunks = 0; // (a gzip stream: known once its last chunk is ready -- published together with it)
    {
      std::unique_lock<std::mutex> lk(rd.mu);
      rd.cv.wait(lk, [&] { return rd.failed || rd.next_ready > j || rd.n_chunks <= j; });
      if (rd.failed) {
        rc = rd.err.empty() ? fail(NTHIP_ERR_ARG, "read error on %s", path) : fail(NTHIP_ERR_ARG, "%s: %s", path, rd.err.c_str());
        break;
      }
      n_chunks = rd.n_chunks;
    }
    if (j > n_chunks) break;
    if (j < n_chunks) {
      uint64_t len = rd.filled[j & 1];
      const bool file_end = ranges ? (*ranges)[j].off + (*ranges)[j].len == rd.file_size : j + 1 == n_chunks;
      if (file_end && len && rd.pinned[j & 1][len - 1] != '\n') rd.pinned[j & 1][len++] = '\n';
      lens[j & 1] = len;
      // d_raw[j & 1] was last read by piece j-2, processed synchronously two iterations ago
      if (hipMemcpyAsync(d_raw[j & 1] + FXS_HEAD, rd.pinned[j & 1], len, hipMemcpyHostToDevice, copy_stream) != hipSuccess ||
          hipEventRecord(ev_h2d[j & 1], copy_stream) != hipSuccess) {
        rc = fail(NTHIP_ERR_HIP, "upload failed");
        break;
      }
    }
    if (j > 0) {
      // piece j-1: its upload was issued one iteration ago and ran under piece j-2's kernels
      if (hipEventSynchronize(ev_h2d[(j - 1) & 1]) != hipSuccess) { rc = fail(NTHIP_ERR_HIP, "upload failed"); break; }
      {
        // its pinned buffer is free again: the reader may fetch chunk j+1 into it
        std::lock_guard<std::mutex> lk(rd.mu);
        rd.next_free = j + 2;
      }
      rd.cv.notify_all();
      rc = process(j - 1, lens[(j - 1) & 1], j == n_chunks);
    }
  }
#undef FX_TRY
  const double read_s = rd.read_seconds;
  cleanup();
  if (stats) {
    stats->read_seconds = read_s;
    stats->gpu_seconds = gpu_seconds;
    stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
  }
  return rc;
}
} // namespace
