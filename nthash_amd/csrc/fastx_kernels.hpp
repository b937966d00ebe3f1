// fastx_kernels.hpp -- FASTQ / single-line FASTA record indexing ON THE DEVICE.
//
// The step before the hash path.  A host parser that finds
// record boundaries and copies sequence lines together runs at a few GB/s per core;
// PCIe moves raw file bytes at tens of GB/s and the GPU finds every newline of a
// chunk at HBM read rate.  So the raw chunk is uploaded as it is, indexed here, and
// the k-mer kernels read the sequence lines where they lie (kmer_ragged_kernel.hpp
// takes reads as spans of one buffer) -- no host parsing, no packing copy.
//
// A chunk starts at a record start (the streaming driver carries the incomplete tail
// of a chunk over to the next one), records have `lpr` lines (4: FASTQ, 2: FASTA with
// one sequence line), the sequence is line 1 of its record.  Two passes over the
// bytes: newlines per block -> exclusive scan -> every newline knows its ordinal, and
// with it its record and its line inside the record.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace ntamd {

constexpr int FX_THREADS = 256;
constexpr int FX_BYTES_PER_THREAD = 64;
constexpr int FX_BLOCK_BYTES = FX_THREADS * FX_BYTES_PER_THREAD; // 16 KiB per block

struct FastxIndexOut {
  unsigned long long consumed;  // one past the last byte of the last complete record
  uint32_t malformed;           // a record did not start with its marker / FASTQ line 2 not '+'
  uint32_t pad;
};

// bit i of the result: byte i of v (16 bytes) is '\n'
__device__ __forceinline__ uint32_t newline_mask16(const uint4 v)
{
  auto m4 = [](uint32_t w) -> uint32_t {
    const uint32_t x = w ^ 0x0A0A0A0Au;                                      // zero byte <=> '\n'
    const uint32_t nz = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u; // 0x80 per NON-zero byte
    const uint32_t z = ~nz & 0x80808080u;
    return __builtin_amdgcn_udot4(z >> 7, 0x08040201u, 0u, false);
  };
  return m4(v.x) | (m4(v.y) << 4) | (m4(v.z) << 8) | (m4(v.w) << 12);
}

// a thread's 64 bytes as a 64-bit newline mask (bytes past n_bytes never count)
__device__ __forceinline__ uint64_t thread_newlines(const uint8_t* __restrict__ buf, uint64_t n_bytes, uint64_t base)
{
  uint64_t mask = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint64_t off = base + 16u * q;
    uint32_t mq = 0;
    if (off + 16u <= n_bytes) {
      uint4 v;
      __builtin_memcpy(&v, buf + off, 16);
      mq = newline_mask16(v);
    } else if (off < n_bytes) {
      for (uint32_t b = 0; b < 16u && off + b < n_bytes; ++b)
        if (buf[off + b] == '\n') mq |= 1u << b;
    }
    mask |= (uint64_t)mq << (16 * q);
  }
  return mask;
}

static __global__ __launch_bounds__(FX_THREADS) void fastx_count_kernel(const uint8_t* __restrict__ buf, uint64_t n_bytes,
                                                                uint64_t* __restrict__ block_counts)
{
  __shared__ uint32_t ws[FX_THREADS / 64];
  const uint64_t base = (uint64_t)blockIdx.x * FX_BLOCK_BYTES + (uint64_t)threadIdx.x * FX_BYTES_PER_THREAD;
  uint32_t c = base < n_bytes ? (uint32_t)__builtin_popcountll(thread_newlines(buf, n_bytes, base)) : 0u;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d, 64);
  if ((threadIdx.x & 63u) == 0) ws[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = (uint64_t)ws[0] + ws[1] + ws[2] + ws[3];
}

// block_base: exclusive scan of block_counts; total_newlines: their sum (device memory).
// marker: '@' (FASTQ) or '>' (FASTA).  Records >= capacity are counted but not written.
static __global__ __launch_bounds__(FX_THREADS) void fastx_index_kernel(const uint8_t* __restrict__ buf, uint64_t n_bytes,
                                                                const uint64_t* __restrict__ block_base,
                                                                const uint64_t* __restrict__ total_newlines,
                                                                uint32_t lpr, uint8_t marker,
                                                                uint64_t* __restrict__ starts,
                                                                uint64_t* __restrict__ ends, uint64_t capacity,
                                                                FastxIndexOut* __restrict__ out)
{
  __shared__ uint32_t ws[FX_THREADS / 64];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint64_t base = (uint64_t)blockIdx.x * FX_BLOCK_BYTES + (uint64_t)tid * FX_BYTES_PER_THREAD;
  uint64_t mask = base < n_bytes ? thread_newlines(buf, n_bytes, base) : 0ull;
  const uint32_t c = (uint32_t)__builtin_popcountll(mask);
  uint32_t incl = c;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d, 64);
    if ((int)lane >= d) incl += o;
  }
  if (lane == 63) ws[wave] = incl;
  __syncthreads();
  uint64_t ord = block_base[blockIdx.x] + (incl - c); // ordinal of this thread's first newline
  for (uint32_t w = 0; w < wave; ++w) ord += ws[w];
  const uint64_t n_complete = *total_newlines / lpr;
  if (blockIdx.x == 0 && tid == 0 && n_bytes > 0 && buf[0] != marker) atomicOr(&out->malformed, 1u);
  unsigned long long consumed = 0;
  while (mask) {
    const uint32_t b = (uint32_t)__builtin_ctzll(mask);
    mask &= mask - 1;
    const uint64_t pos = base + b;
    const uint64_t rec = ord / lpr;
    const uint32_t li = (uint32_t)(ord - rec * lpr);
    ++ord;
    if (rec >= n_complete) break; // the incomplete record at the end of the chunk
    if (li == 0 && rec < capacity) starts[rec] = pos + 1;
    if (li == 1) {
      if (rec < capacity) ends[rec] = pos - ((pos > 0 && buf[pos - 1] == '\r') ? 1u : 0u);
      if (lpr == 4 && pos + 1 < n_bytes && buf[pos + 1] != '+') atomicOr(&out->malformed, 1u);
    }
    if (li == lpr - 1) {
      consumed = pos + 1;
      if (pos + 1 < n_bytes && buf[pos + 1] != marker) atomicOr(&out->malformed, 1u);
    }
  }
  if (consumed) atomicMax(&out->consumed, consumed);
}

} // namespace ntamd

// ==========================================================================
// multi-line FASTA: strip header lines and line ends, sequences back to back
// ==========================================================================
// Two kinds of events decide whether a byte belongs to a header line: a '>' at a line
// start (byte 0, or right after '\n') opens one, the next '\n' closes it.  Both are
// visible locally (one byte of look-behind), so "is byte i inside a header" is the kind
// of the LAST event at or before i: a scan with the operator "right operand unless it
// is empty".  Pass 1 gives every 16 KiB block its last event, one small kernel carries
// that across blocks, pass 2 counts kept bytes / headers per block (sum scans give the
// output position of every block), pass 3 writes the bytes and the record offsets.
namespace ntamd {

enum : uint32_t { FA_EV_NONE = 0, FA_EV_HEADER = 1, FA_EV_NEWLINE = 2 };

struct FaChunk {
  uint64_t nl, hs, keep_if_seq; // newline bits, header-start bits, bytes kept when no header is open
  uint32_t valid;               // bytes of this chunk inside the buffer
};

// masks of one thread's 64 bytes (nl from thread_newlines)
__device__ __forceinline__ FaChunk fasta_chunk(const uint8_t* __restrict__ buf, uint64_t n_bytes, uint64_t base)
{
  FaChunk ch;
  ch.nl = ch.hs = ch.keep_if_seq = 0;
  ch.valid = 0;
  if (base >= n_bytes) return ch;
  ch.valid = n_bytes - base < 64u ? (uint32_t)(n_bytes - base) : 64u;
  ch.nl = thread_newlines(buf, n_bytes, base);
  uint64_t gt = 0, cr = 0;
  for (uint32_t b = 0; b < ch.valid; ++b) {
    const uint8_t c = buf[base + b];
    if (c == '>') gt |= 1ull << b;
    if (c == '\r') cr |= 1ull << b;
  }
  const uint64_t prev_nl = base == 0 ? 1ull : (buf[base - 1] == '\n' ? 1ull : 0ull);
  const uint64_t line_start = (ch.nl << 1) | prev_nl;
  ch.hs = gt & line_start;
  const uint64_t in_range = ch.valid == 64u ? ~0ull : ((1ull << ch.valid) - 1ull);
  ch.keep_if_seq = in_range & ~ch.nl & ~cr;
  return ch;
}

// bit i set <=> byte i of the chunk lies in a header line (carry: a header is open at the chunk's start)
__device__ __forceinline__ uint64_t fasta_header_mask(const FaChunk& ch, bool carry)
{
  // walk the events in order; few per 64 bytes
  uint64_t mask = 0;
  uint64_t ev = ch.nl | ch.hs;
  bool open = carry;
  uint32_t from = 0;
  while (ev) {
    const uint32_t b = (uint32_t)__builtin_ctzll(ev);
    ev &= ev - 1;
    if (open && b > from) mask |= ((b >= 64u ? ~0ull : ((1ull << b) - 1ull)) & ~((1ull << from) - 1ull));
    if ((ch.hs >> b) & 1ull) { open = true; from = b; }
    else { if (open) mask |= 1ull << b; open = false; from = b + 1u; } // the closing '\n' is dropped anyway
  }
  if (open && from < 64u) mask |= ~((1ull << from) - 1ull);
  return mask;
}

__device__ __forceinline__ uint32_t fasta_last_event(const FaChunk& ch)
{
  const uint64_t ev = ch.nl | ch.hs;
  if (!ev) return FA_EV_NONE;
  const uint32_t b = 63u - (uint32_t)__builtin_clzll(ev);
  return ((ch.hs >> b) & 1ull) ? FA_EV_HEADER : FA_EV_NEWLINE;
}

// "right unless empty" inclusive scan over the 256 threads of a block; returns the EXCLUSIVE value
__device__ __forceinline__ uint32_t fasta_block_carry(uint32_t mine, uint32_t* ws /*[4]*/, uint32_t& block_last)
{
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  uint32_t incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d, 64);
    if ((int)lane >= d && incl == FA_EV_NONE) incl = o;
  }
  uint32_t excl = __shfl_up(incl, 1, 64);
  if (lane == 0) excl = FA_EV_NONE;
  if (lane == 63) ws[wave] = incl;
  __syncthreads();
  uint32_t before = FA_EV_NONE; // last event of the waves before this one
  for (uint32_t w = 0; w < wave; ++w)
    if (ws[w] != FA_EV_NONE) before = ws[w];
  if (excl == FA_EV_NONE) excl = before;
  block_last = FA_EV_NONE;
  for (uint32_t w = 0; w < FX_THREADS / 64; ++w)
    if (ws[w] != FA_EV_NONE) block_last = ws[w];
  return excl;
}

// pass 1: last event of every block
static __global__ __launch_bounds__(FX_THREADS) void fasta_events_kernel(const uint8_t* __restrict__ buf, uint64_t n_bytes,
                                                                 uint32_t* __restrict__ block_last)
{
  __shared__ uint32_t ws[FX_THREADS / 64];
  const uint64_t base = (uint64_t)blockIdx.x * FX_BLOCK_BYTES + (uint64_t)threadIdx.x * FX_BYTES_PER_THREAD;
  const FaChunk ch = fasta_chunk(buf, n_bytes, base);
  uint32_t bl;
  (void)fasta_block_carry(fasta_last_event(ch), ws, bl);
  if (threadIdx.x == 0) block_last[blockIdx.x] = bl;
}

// carry across blocks: block_carry[b] = last event of blocks [0, b) (one block, sequential over tiles of 1024)
static __global__ __launch_bounds__(1024) void fasta_carry_kernel(const uint32_t* __restrict__ block_last, uint64_t nb,
                                                          uint32_t* __restrict__ block_carry)
{
  __shared__ uint32_t ws[16];
  __shared__ uint32_t running;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tid == 0) running = FA_EV_NONE;
  __syncthreads();
  for (uint64_t b0 = 0; b0 < nb; b0 += 1024) {
    const uint64_t b = b0 + tid;
    const uint32_t mine = b < nb ? block_last[b] : FA_EV_NONE;
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(incl, d, 64);
      if ((int)lane >= d && incl == FA_EV_NONE) incl = o;
    }
    uint32_t excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = FA_EV_NONE;
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    uint32_t before = running;
    for (uint32_t w = 0; w < wave; ++w)
      if (ws[w] != FA_EV_NONE) before = ws[w];
    if (excl == FA_EV_NONE) excl = before;
    if (b < nb) block_carry[b] = excl;
    __syncthreads();
    if (tid == 1023) running = incl != FA_EV_NONE ? incl : before;
    __syncthreads();
  }
}

// pass 2: kept bytes and header starts per block
static __global__ __launch_bounds__(FX_THREADS) void fasta_count_kernel(const uint8_t* __restrict__ buf, uint64_t n_bytes,
                                                                const uint32_t* __restrict__ block_carry,
                                                                uint64_t* __restrict__ block_kept,
                                                                uint64_t* __restrict__ block_hdrs)
{
  __shared__ uint32_t ws[FX_THREADS / 64];
  __shared__ uint32_t sk[FX_THREADS / 64], sh[FX_THREADS / 64];
  const uint64_t base = (uint64_t)blockIdx.x * FX_BLOCK_BYTES + (uint64_t)threadIdx.x * FX_BYTES_PER_THREAD;
  const FaChunk ch = fasta_chunk(buf, n_bytes, base);
  uint32_t bl;
  uint32_t carry = fasta_block_carry(fasta_last_event(ch), ws, bl);
  if (carry == FA_EV_NONE) carry = block_carry[blockIdx.x];
  const uint64_t hdr = fasta_header_mask(ch, carry == FA_EV_HEADER);
  uint32_t kept = (uint32_t)__builtin_popcountll(ch.keep_if_seq & ~hdr);
  uint32_t hs = (uint32_t)__builtin_popcountll(ch.hs);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    kept += __shfl_xor(kept, d, 64);
    hs += __shfl_xor(hs, d, 64);
  }
  if ((threadIdx.x & 63u) == 0) { sk[threadIdx.x >> 6] = kept; sh[threadIdx.x >> 6] = hs; }
  __syncthreads();
  if (threadIdx.x == 0) {
    block_kept[blockIdx.x] = (uint64_t)sk[0] + sk[1] + sk[2] + sk[3];
    block_hdrs[blockIdx.x] = (uint64_t)sh[0] + sh[1] + sh[2] + sh[3];
  }
}

// pass 3: write the kept bytes at their output positions and the offset of every record
static __global__ __launch_bounds__(FX_THREADS) void fasta_scatter_kernel(const uint8_t* __restrict__ buf, uint64_t n_bytes,
                                                                  const uint32_t* __restrict__ block_carry,
                                                                  const uint64_t* __restrict__ kept_base,
                                                                  const uint64_t* __restrict__ hdr_base,
                                                                  uint8_t* __restrict__ seqs,
                                                                  uint64_t* __restrict__ offsets, uint64_t capacity)
{
  __shared__ uint32_t ws[FX_THREADS / 64];
  __shared__ uint32_t sk[FX_THREADS / 64], sh[FX_THREADS / 64];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint64_t base = (uint64_t)blockIdx.x * FX_BLOCK_BYTES + (uint64_t)tid * FX_BYTES_PER_THREAD;
  const FaChunk ch = fasta_chunk(buf, n_bytes, base);
  uint32_t bl;
  uint32_t carry = fasta_block_carry(fasta_last_event(ch), ws, bl);
  if (carry == FA_EV_NONE) carry = block_carry[blockIdx.x];
  const uint64_t hdr = fasta_header_mask(ch, carry == FA_EV_HEADER);
  uint64_t keep = ch.keep_if_seq & ~hdr;
  const uint32_t kept = (uint32_t)__builtin_popcountll(keep);
  const uint32_t hs = (uint32_t)__builtin_popcountll(ch.hs);
  uint32_t ik = kept, ih = hs;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t ok = __shfl_up(ik, d, 64), oh = __shfl_up(ih, d, 64);
    if ((int)lane >= d) { ik += ok; ih += oh; }
  }
  if (lane == 63) { sk[wave] = ik; sh[wave] = ih; }
  __syncthreads();
  uint64_t out = kept_base[blockIdx.x] + (ik - kept);
  uint64_t rec = hdr_base[blockIdx.x] + (ih - hs);
  for (uint32_t w = 0; w < wave; ++w) { out += sk[w]; rec += sh[w]; }
  // walk the chunk: kept bytes go out in order, a header start records the current output position
  uint64_t ev = keep | ch.hs;
  while (ev) {
    const uint32_t b = (uint32_t)__builtin_ctzll(ev);
    ev &= ev - 1;
    if ((ch.hs >> b) & 1ull) {
      if (rec < capacity) offsets[rec] = out;
      ++rec;
    } else {
      seqs[out++] = buf[base + b];
    }
  }
}

} // namespace ntamd
