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

__global__ __launch_bounds__(FX_THREADS) void fastx_count_kernel(const uint8_t* __restrict__ buf, uint64_t n_bytes,
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
__global__ __launch_bounds__(FX_THREADS) void fastx_index_kernel(const uint8_t* __restrict__ buf, uint64_t n_bytes,
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
