// util_kernels.hpp -- small device helpers around the hash kernels:
// exclusive scan of per-read counts (compact output offsets), dense fills,
// counter-based synthetic reads, stream checksums, and a copy yardstick.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "kmer_runs_gen_kernel.hpp" // horner_first_window, pack16
#include "nt_math.hpp"

namespace ntamd {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS; // 1024 values per block

__device__ __forceinline__ uint64_t wave_incl_scan(uint64_t v, uint32_t lane)
{
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t o = __shfl_up(v, d, 64);
    if ((int)lane >= d) v += o;
  }
  return v;
}

// per-tile exclusive scan; tile totals go to block_sums
static __global__ __launch_bounds__(SCAN_THREADS) void scan_tiles_kernel(const uint64_t* in,
                                                                 uint64_t* out,
                                                                 uint64_t* __restrict__ block_sums,
                                                                 uint64_t n)
{
  __shared__ uint64_t wave_tot[SCAN_THREADS / 64];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)tid * SCAN_ITEMS;
  uint64_t v[SCAN_ITEMS];
  uint64_t sum = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    v[i] = (base + i < n) ? in[base + i] : 0;
    sum += v[i];
  }
  const uint64_t incl = wave_incl_scan(sum, lane);
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint64_t off = 0;
  for (uint32_t w = 0; w < wave; ++w) off += wave_tot[w];
  uint64_t run = off + incl - sum;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    if (base + i < n) out[base + i] = run;
    run += v[i];
  }
  if (tid == SCAN_THREADS - 1) block_sums[blockIdx.x] = run;
}

// single block: exclusive scan of the tile totals in place, grand total out
static __global__ __launch_bounds__(SCAN_THREADS) void scan_sums_kernel(uint64_t* __restrict__ sums, uint64_t n,
                                                                uint64_t* __restrict__ total)
{
  __shared__ uint64_t wave_tot[SCAN_THREADS / 64];
  __shared__ uint64_t carry_s;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (uint64_t b0 = 0; b0 < n; b0 += SCAN_THREADS) {
    const uint64_t i = b0 + tid;
    const uint64_t v = i < n ? sums[i] : 0;
    const uint64_t incl = wave_incl_scan(v, lane);
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint64_t off = carry_s;
    for (uint32_t w = 0; w < wave; ++w) off += wave_tot[w];
    if (i < n) sums[i] = off + incl - v;
    __syncthreads();
    if (tid == SCAN_THREADS - 1) carry_s = off + incl;
    __syncthreads();
  }
  if (tid == 0) *total = carry_s;
}

static __global__ __launch_bounds__(SCAN_THREADS) void scan_add_kernel(uint64_t* __restrict__ out,
                                                               const uint64_t* __restrict__ sums, uint64_t n)
{
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
  const uint64_t add = sums[blockIdx.x];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i)
    if (base + i < n) out[base + i] += add;
}

// reads of one length at one stride as spans: [r * stride, r * stride + len)
static __global__ void fill_spans_kernel(uint64_t* __restrict__ starts, uint64_t* __restrict__ ends, uint64_t n, uint64_t stride,
                                         uint64_t len)
{
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x) {
    starts[r] = r * stride;
    ends[r] = r * stride + len;
  }
}

static __global__ void fill_u64_kernel(uint64_t* __restrict__ dst, uint64_t n, uint64_t value)
{
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x)
    dst[i] = value;
}

// pos[r*nwin + p] = p
static __global__ void fill_pos_kernel(uint32_t* __restrict__ dst, uint64_t n, uint32_t nwin)
{
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x)
    dst[i] = (uint32_t)(i % nwin);
}

__device__ __host__ inline uint64_t splitmix64(uint64_t x)
{
  uint64_t z = x + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

// Counter-based synthetic reads: one thread per (read, 32-base word)
static __global__ void synth_reads_kernel(uint8_t* __restrict__ dst, uint64_t first_read, uint64_t n_reads,
                                   uint32_t len, uint64_t seed)
{
  const uint64_t W = (len + 31u) / 32u;
  const uint64_t n = n_reads * W;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t r = i / W, w = i - r * W;
    const uint64_t x = splitmix64(seed + (first_read + r) * W + w);
    uint8_t* out = dst + r * (uint64_t)len + w * 32u;
    const uint32_t cnt = (uint32_t)((w * 32u + 32u <= len) ? 32u : (len - w * 32u));
    for (uint32_t j = 0; j < cnt; ++j) {
      const uint32_t c = (uint32_t)(x >> (2u * j)) & 3u;
      out[j] = (uint8_t)(c == 0 ? 'A' : c == 1 ? 'C' : c == 2 ? 'G' : 'T');
    }
  }
}

// wrapping sum and xor of a u64 stream; partial[2*block + {0,1}]
static __global__ __launch_bounds__(256) void checksum_kernel(const uint64_t* __restrict__ v, uint64_t n,
                                                      uint64_t* __restrict__ partial)
{
  __shared__ uint64_t ss[4], sx[4];
  uint64_t s = 0, x = 0;
  const uint64_t n2 = n / 2;
  const ulonglong2* v2 = (const ulonglong2*)v;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2;
       i += (uint64_t)gridDim.x * blockDim.x) {
    const ulonglong2 q = v2[i];
    s += q.x + q.y;
    x ^= q.x ^ q.y;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) { s += v[n - 1]; x ^= v[n - 1]; }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    s += __shfl_xor(s, d, 64);
    x ^= __shfl_xor(x, d, 64);
  }
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (lane == 0) { ss[wave] = s; sx[wave] = x; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = ss[0] + ss[1] + ss[2] + ss[3];
    partial[2 * blockIdx.x + 1] = sx[0] ^ sx[1] ^ sx[2] ^ sx[3];
  }
}

// Batched BlindNtHash::peek / peek_back (src/kmer.cpp:377-393): one lane per k-mer.
// The k-mer (bases only, not validated -- like BlindNtHash) is hashed directly, then the
// 4 successors / predecessors follow from one roll step each.
static __global__ __launch_bounds__(256) void kmer_extend_kernel(const uint8_t* __restrict__ kmers, uint64_t n, uint32_t k,
                                                         uint32_t m, uint64_t* __restrict__ self,
                                                         uint64_t* __restrict__ next, uint64_t* __restrict__ prev)
{
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* s = kmers + i * k;
  const uint64_t f = direct_fwd((const char*)s, k), r = direct_rev((const char*)s, k);
  const uint64_t base = (uint64_t)k * MULTISEED;
  auto emit = [&](uint64_t* dst, uint64_t ff, uint64_t rr) {
    const uint64_t h0 = ff + rr;
    dst[0] = h0;
    for (uint32_t j = 1; j < m; ++j) dst[j] = mix_hash(h0, (uint64_t)j ^ base);
  };
  if (self) emit(self + i * m, f, r);
  const uint8_t first = s[0], last = s[k - 1];
  const uint64_t out_f_next = srol_n(fwd_seed(first), k), out_r_next = rc_seed(first);
  const uint64_t out_f_prev = fwd_seed(last), out_r_prev = srol_n(rc_seed(last), k);
  const uint64_t sk_in_rc[4] = {srol_n(SEED_T, k), srol_n(SEED_G, k), srol_n(SEED_C, k), srol_n(SEED_A, k)};
  const uint64_t sk_in_fw[4] = {srol_n(SEED_A, k), srol_n(SEED_C, k), srol_n(SEED_G, k), srol_n(SEED_T, k)};
  const uint64_t in_fw[4] = {SEED_A, SEED_C, SEED_G, SEED_T}; // "ACGT"[b]
  const uint64_t in_rc[4] = {SEED_T, SEED_G, SEED_C, SEED_A};
#pragma unroll
  for (uint32_t b = 0; b < 4; ++b) {
    if (next) // next_forward_hash / next_reverse_hash, src/kmer.cpp:84-94,164-174
      emit(next + (i * 4 + b) * m, srol1(f) ^ in_fw[b] ^ out_f_next, sror1(r ^ sk_in_rc[b] ^ out_r_next));
    if (prev) // prev_forward_hash / prev_reverse_hash, src/kmer.cpp:104-114,184-194
      emit(prev + (i * 4 + b) * m, sror1(f ^ sk_in_fw[b] ^ out_f_prev), srol1(r) ^ in_rc[b] ^ out_r_prev);
  }
}

// The same for k <= 64 with the first-window byte tables in LDS (4 bases per lookup instead of 2k Horner
// steps) and, for m == 1, the four neighbours of the wave's 64 k-mers (2 KiB contiguous) exchanged through LDS
// so that each store instruction writes one contiguous KiB.  tab = build_byte_tables(k), zero-padded to ntab.
#ifndef KX_WIDE_THREADS
#define KX_WIDE_THREADS 512 // NW >= 3 (k > 32): 16 lookups in flight need more than the 128 VGPRs a 1024-thread block gets
#endif
template <int NW>
__global__ __launch_bounds__(NW >= 3 ? KX_WIDE_THREADS : 1024) void kmer_extend_tab_kernel(const uint8_t* __restrict__ kmers, uint64_t n, uint32_t k,
                                                              uint32_t m, const uint4* __restrict__ tab, uint32_t ntab,
                                                              uint64_t* __restrict__ self, uint64_t* __restrict__ next,
                                                              uint64_t* __restrict__ prev)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_ext[];
  uint4* itab = (uint4*)lds_ext;
  for (uint32_t i = threadIdx.x; i < ntab * 256u; i += blockDim.x) itab[i] = tab[i];
  __syncthreads();
  const uint64_t base = (uint64_t)k * MULTISEED;
  const uint64_t total_bytes = n * (uint64_t)k;
  // srol^k of the four seeds, forward and complement (code order A C T G of (c >> 1) & 3)
  uint64_t sk[4], skc[4];
#pragma unroll
  for (uint32_t c = 0; c < 4; ++c) {
    sk[c] = srol_n(seed_of_code(c), k);
    skc[c] = srol_n(seed_of_code(c ^ 2u), k);
  }
  // NW == 0 (k > 64): per-wave 2-bit stream of the wave's 64 k-mers after the exchange tiles
  const uint32_t wbits_dw = NW == 0 ? ((64u * k + 30u) >> 4) + 4u : 0u;
  uint32_t* wbits = (uint32_t*)(itab + ntab * 256u) + 16u * 512u + (threadIdx.x >> 6) * wbits_dw;
  // (the trip count is uniform inside a wave: the lanes past the end idle through the last iteration)
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i - (threadIdx.x & 63u) < n;
       i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t off = i * k;
    uint32_t f0 = 0, f1 = 0, r0 = 0, r1 = 0, c_first = 0, c_last = 0;
    if constexpr (NW == 0) {
      // stage the wave's k-mers (64 * k contiguous bytes) as a 2-bit stream, then Horner over 4-base table
      // entries per lane (the tables of kmer_runs_gen_kernel's NW = 0 path: a 4-mer's and a 1-mer's)
      const uint32_t lane0 = threadIdx.x & 63u;
      const uint64_t i_w = i - lane0;
      const uint64_t cnt = n - i_w < 64u ? n - i_w : 64u;
      const uint64_t addr = (uint64_t)(kmers + i_w * k);
      const uint32_t shift = (uint32_t)(addr & 15u);
      const uint32_t n_vec = (uint32_t)((shift + cnt * k + 15u) >> 4);
      // (aligned 16-byte blocks that hold at least one byte of the batch: never cross into another page)
      for (uint32_t v = lane0; v < n_vec; v += 64u) {
        uint32_t b = 0;
        wbits[v] = pack16(*(const uint4*)(addr - shift + ((uint64_t)v << 4)), b);
      }
      if (lane0 < 3u) wbits[n_vec + lane0] = 0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
      if (i < n) {
        const uint32_t b0 = shift + lane0 * k;
        any_k_first_window(wbits, itab, b0, k, f0, f1, r0, r1);
        c_first = (wbits[b0 >> 4] >> ((b0 & 15u) * 2u)) & 3u;
        const uint32_t bl = b0 + k - 1u;
        c_last = (wbits[bl >> 4] >> ((bl & 15u) * 2u)) & 3u;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
    }
    if (i >= n) continue;
    uint32_t w[NW ? NW : 1];
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      uint4 v = make_uint4(0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u);
      const uint64_t o = off + 16u * q;
      if (o + 16u <= total_bytes) {
        __builtin_memcpy(&v, kmers + o, 16);
      } else {
        uint32_t wv[4] = {0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u};
        for (uint32_t b = 0; o + b < total_bytes && b < 16u; ++b) {
          wv[b >> 2] &= ~(0xFFu << ((b & 3u) * 8u));
          wv[b >> 2] |= (uint32_t)kmers[o + b] << ((b & 3u) * 8u);
        }
        v = make_uint4(wv[0], wv[1], wv[2], wv[3]);
      }
      // 16 bases -> 16 two-bit codes ((c >> 1) & 3, no validation: like BlindNtHash)
      auto p4 = [](uint32_t x) { return __builtin_amdgcn_udot4((x >> 1) & 0x03030303u, 0x40100401u, 0u, false); };
      w[q] = p4(v.x) | (p4(v.y) << 8) | (p4(v.z) << 16) | (p4(v.w) << 24);
    }
#pragma unroll
    for (int jt = 0; jt < 4 * NW; ++jt) { // ntab == 4 * NW (zero tables past ceil(k/4))
      const uint4 e = itab[(uint32_t)jt * 256u + ((w[jt >> 2] >> ((jt & 3) * 8)) & 0xFFu)];
      f0 ^= e.x; f1 ^= e.y; r0 ^= e.z; r1 ^= e.w;
    }
    const uint64_t f = ((uint64_t)f1 << 32) | f0, r = ((uint64_t)r1 << 32) | r0;
    if constexpr (NW != 0) {
      c_first = w[0] & 3u;
      c_last = (w[(k - 1u) >> 4] >> (((k - 1u) & 15u) * 2u)) & 3u;
    }
    auto emit = [&](uint64_t* dst, uint64_t h0) {
      dst[0] = h0;
      for (uint32_t j = 1; j < m; ++j) dst[j] = mix_hash(h0, (uint64_t)j ^ base);
    };
    // base order "ACGT" = codes 0, 1, 3, 2
    uint64_t hn[4], hp[4];
#pragma unroll
    for (uint32_t b = 0; b < 4; ++b) {
      const uint32_t c = b == 2u ? 3u : b == 3u ? 2u : b;
      // next_forward_hash / next_reverse_hash, src/kmer.cpp:84-94,164-174
      hn[b] = (srol1(f) ^ seed_of_code(c) ^ sk[c_first]) + sror1(r ^ skc[c] ^ seed_of_code(c_first ^ 2u));
      // prev_forward_hash / prev_reverse_hash, src/kmer.cpp:104-114,184-194
      hp[b] = sror1(f ^ sk[c] ^ seed_of_code(c_last)) + (srol1(r) ^ seed_of_code(c ^ 2u) ^ skc[c_last]);
    }
    if (m == 1u) {
      if (self) self[i] = f + r;
      // a wave's 64 k-mers own 2 KiB of next[] (and of prev[]): through a wave-private LDS tile, so that each of
      // the two store instructions writes one contiguous KiB instead of 16-byte pieces 32 bytes apart
      const uint32_t lane = threadIdx.x & 63u;
      const uint64_t i_wave = i - lane;
      const bool whole_wave = i_wave + 64u <= n; // uniform: lanes of a wave hold consecutive k-mers
      ulonglong2* xt = (ulonglong2*)(itab + ntab * 256u) + (threadIdx.x >> 6) * 128u;
      auto out4 = [&](uint64_t* arr, const uint64_t* h) {
        if (!arr) return;
        if (whole_wave) {
          xt[2u * lane] = make_ulonglong2(h[0], h[1]);
          xt[2u * lane + 1u] = make_ulonglong2(h[2], h[3]);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
          ulonglong2* d = (ulonglong2*)(arr + i_wave * 4u);
          const ulonglong2 a0 = xt[lane], a1 = xt[64u + lane];
          d[lane] = a0;
          d[64u + lane] = a1;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
          __builtin_amdgcn_wave_barrier();
        } else {
          ulonglong2* d = (ulonglong2*)(arr + i * 4u);
          d[0] = make_ulonglong2(h[0], h[1]);
          d[1] = make_ulonglong2(h[2], h[3]);
        }
      };
      out4(next, hn);
      out4(prev, hp);
    } else {
      // m > 1: the wave's 64 * 4 * m values of next[] (and prev[]) are 2*m KiB contiguous.  The 256 base hashes go
      // to the exchange tile; then lane t produces values 2t, 2t+1 (+128, +256, ...) of the block -- value v is
      // hash v % m of base hash v / m -- and stores them as 16 bytes: every store instruction one contiguous KiB
      const uint32_t lane = threadIdx.x & 63u;
      const uint64_t i_wave = i - lane;
      const bool whole_wave = i_wave + 64u <= n;
      uint64_t* xt = (uint64_t*)(itab + ntab * 256u) + (threadIdx.x >> 6) * 256u;
      const uint32_t inv_m = 0xFFFFFFFFu / m + 1u; // v / m == umulhi(v, inv_m) for v < 2^29
      // the n_base base hashes in the exchange tile -> n_base * m values at dst, 16 bytes per lane and store
      auto expand = [&](uint64_t* dst, uint32_t n_base) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        for (uint32_t v2 = 2u * lane; v2 < n_base * m; v2 += 128u) {
          uint64_t o[2];
#pragma unroll
          for (uint32_t q = 0; q < 2; ++q) {
            const uint32_t v = v2 + q, e = __umulhi(v, inv_m), j = v - e * m;
            const uint64_t hb = xt[e];
            o[q] = j == 0u ? hb : mix_hash(hb, (uint64_t)j ^ base);
          }
          *(ulonglong2*)(dst + v2) = make_ulonglong2(o[0], o[1]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
      };
      auto out4m = [&](uint64_t* arr, const uint64_t* h) {
        if (!arr) return;
        if (!whole_wave) {
#pragma unroll
          for (uint32_t b = 0; b < 4; ++b) emit(arr + (i * 4u + b) * m, h[b]);
          return;
        }
        ((ulonglong2*)xt)[2u * lane] = make_ulonglong2(h[0], h[1]);
        ((ulonglong2*)xt)[2u * lane + 1u] = make_ulonglong2(h[2], h[3]);
        expand(arr + i_wave * 4u * m, 256u);
      };
      if (self) {
        if (whole_wave) {
          xt[lane] = f + r;
          expand(self + i_wave * m, 64u);
        } else {
          emit(self + i * m, f + r);
        }
      }
      out4m(next, hn);
      out4m(prev, hp);
    }
  }
}

// get_pos() of the k-mers the dense kernels write: a read of bases only emits every window, 0 .. nwin-1, at its
// place in the stream (read_off[r], or r * nwin for a clean batch); flagged reads are left to the N-aware kernels.
// One wave per read at a time: nwin * 4 contiguous bytes.
static __global__ __launch_bounds__(256) void fill_window_pos_kernel(uint32_t* __restrict__ pos, uint64_t n_reads, uint32_t nwin,
                                                           const uint64_t* __restrict__ read_dirty,
                                                           const uint64_t* __restrict__ read_off)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
  for (uint64_t r = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < n_reads; r += n_waves) {
    if (read_dirty && read_dirty[r]) continue;
    uint32_t* dst = pos + (read_off ? read_off[r] : r * nwin);
    for (uint32_t p = lane; p < nwin; p += 64u) dst[p] = p;
  }
}

// plain 16-byte/lane device copy: the achievable-bandwidth yardstick
static __global__ __launch_bounds__(256) void copy_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src,
                                                  uint64_t n16)
{
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16;
       i += (uint64_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}

// write-only yardstick, shaped like the headline kernel's copy-out (which writes 6.7-7.0 TB/s with its hashing and
// slab loads switched off): one block of 8 waves per CU streams through its own contiguous range, every wave writes
// whole tiles of 7.5 KiB -- one contiguous KiB per write-through store instruction -- and lets at most one tile's
// stores stay in flight while it issues the next (an unthrottled fill of 16 waves reaches only 5.7 TB/s)
static __global__ __launch_bounds__(512) void fill16_kernel(uint4* __restrict__ dst, uint64_t n16, uint32_t value)
{
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  const v4u v = {value, value ^ 0x55555555u, ~value, lane};
  constexpr uint64_t TILE = 480u; // uint4 per tile: 7.5 KiB
  const uint64_t n_tiles = n16 / TILE;
  const uint64_t per = (n_tiles + gridDim.x - 1) / gridDim.x;
  const uint64_t t0 = (uint64_t)blockIdx.x * per;
  const uint64_t t1 = t0 + per < n_tiles ? t0 + per : n_tiles;
  for (uint64_t t = t0 + wave; t < t1; t += n_waves) {
    uint4* p = dst + t * TILE + lane;
#pragma unroll
    for (uint32_t j = 0; j < 7u; ++j)
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p + j * 64u), "v"(v) : "memory");
    if (lane < 32u) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p + 448u), "v"(v) : "memory");
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#ifdef FILL_SLEEP
    __builtin_amdgcn_s_sleep(FILL_SLEEP);
#endif
  }
  // the tail that is not a whole tile
  for (uint64_t i = n_tiles * TILE + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16;
       i += (uint64_t)gridDim.x * blockDim.x)
    *((v4u*)dst + i) = v;
}

// One pass over n+1 offsets of back-to-back reads: are they in order and inside the buffer, do all reads have one
// length, how long is the longest.  res: [0] offsets[0] [1] offsets[1] [2] lengths differ [3] bad [4] max length.
// (Thousands of waves hammering one address serialise in L2: look first, a flag is set / the maximum reached early.)
static __global__ __launch_bounds__(256) void offsets_survey_kernel(const uint64_t* __restrict__ offsets, uint64_t n,
                                                                    uint64_t buf_bytes, unsigned long long* __restrict__ res)
{
  const uint64_t o0 = offsets[0], o1 = offsets[1];
  const uint64_t len0 = o1 - o0;
  uint32_t differs = 0, bad = 0;
  uint64_t mlen = 0;
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t a0 = offsets[r], a1 = offsets[r + 1];
    if (a1 < a0 || a1 > buf_bytes) { bad = 1u; continue; }
    if (a1 - a0 != len0) differs = 1u;
    if (a1 - a0 > mlen) mlen = a1 - a0;
  }
  for (int d = 32; d > 0; d >>= 1) {
    const uint64_t ol = ((uint64_t)(uint32_t)__shfl_down((int)(uint32_t)(mlen >> 32), d, 64) << 32) |
                        (uint32_t)__shfl_down((int)(uint32_t)mlen, d, 64);
    if (ol > mlen) mlen = ol;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { res[0] = o0; res[1] = o1; }
  // per block: one thread speaks for its four waves
  __shared__ uint64_t w_mlen[4];
  __shared__ uint32_t w_flags[4];
  const bool any_differs = __ballot(differs != 0) != 0, any_bad = __ballot(bad != 0) != 0;
  if ((threadIdx.x & 63u) == 0) {
    w_mlen[threadIdx.x >> 6] = mlen;
    w_flags[threadIdx.x >> 6] = (any_differs ? 1u : 0u) | (any_bad ? 2u : 0u);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint64_t bl = 0;
    uint32_t bf = 0;
    for (int w = 0; w < 4; ++w) {
      bl = w_mlen[w] > bl ? w_mlen[w] : bl;
      bf |= w_flags[w];
    }
    if ((bf & 1u) && __atomic_load_n(&res[2], __ATOMIC_RELAXED) == 0) atomicOr(&res[2], 1ull);
    if ((bf & 2u) && __atomic_load_n(&res[3], __ATOMIC_RELAXED) == 0) atomicOr(&res[3], 1ull);
    if (bl > __atomic_load_n(&res[4], __ATOMIC_RELAXED)) atomicMax(&res[4], (unsigned long long)bl);
  }
}

// spans / offsets sanity (the kernels trust them): every read must satisfy starts[r] <= ends[r] <= buf_bytes
static __global__ __launch_bounds__(256) void check_spans_kernel(const uint64_t* __restrict__ starts,
                                                                 const uint64_t* __restrict__ ends, uint64_t n,
                                                                 uint64_t buf_bytes, uint32_t* __restrict__ bad,
                                                                 unsigned long long* __restrict__ max_len)
{
  uint32_t b = 0;
  uint64_t longest = 0; // (of the well-formed spans; a malformed one fails the call anyway)
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t s0 = starts[r], e0 = ends[r];
    if (s0 > e0 || e0 > buf_bytes) b = 1u;
    else if (e0 - s0 > longest) longest = e0 - s0;
  }
  if (__ballot(b != 0) != 0 && (threadIdx.x & 63u) == 0) atomicOr(bad, 1u);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const uint64_t o = __shfl_xor(longest, d, 64);
    longest = o > longest ? o : longest;
  }
  if ((threadIdx.x & 63u) == 0 && longest) atomicMax(max_len, (unsigned long long)longest);
}

// the tiles of the N-aware count pass that lost a window: tile t holds min(64, n_runs - 64 t) runs of C windows when none
// is lost.  list[0 .. *n_list) in any order (*n_list zeroed by the host).  A block collects its tiles in LDS and takes
// its place in the list with ONE atomic per 1792 entries (20 000 atomics on one word took 0.19 ms).  256 threads.
static __global__ __launch_bounds__(256) void list_short_tiles_kernel(const uint64_t* tile_counts, uint64_t n_tiles, uint64_t n_runs,
                                                                      uint32_t C, uint64_t* list, unsigned long long* n_list)
{
  constexpr uint32_t CAP = 2048;
  __shared__ uint64_t buf[CAP];
  __shared__ uint32_t n_buf;
  __shared__ uint64_t base;
  if (threadIdx.x == 0) n_buf = 0;
  __syncthreads();
  auto flush = [&]() { // (block-wide)
    __syncthreads();
    const uint32_t n = n_buf;
    if (threadIdx.x == 0 && n) base = atomicAdd(n_list, (unsigned long long)n);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 256u) list[base + i] = buf[i];
    __syncthreads();
    if (threadIdx.x == 0) n_buf = 0;
    __syncthreads();
  };
  for (uint64_t t0 = (uint64_t)blockIdx.x * 256u; t0 < n_tiles; t0 += (uint64_t)gridDim.x * 256u) {
    const uint64_t t = t0 + threadIdx.x;
    if (t < n_tiles) {
      const uint64_t left = n_runs - t * 64u;
      const uint64_t full = (left < 64u ? left : 64u) * C;
      if (tile_counts[t] != full) buf[atomicAdd(&n_buf, 1u)] = t;
    }
    __syncthreads();
    const bool full = n_buf + 256u > CAP; // every thread reads the count behind ONE barrier ...
    __syncthreads();                      // ... and nobody adds to it before all have: the decision is the block's (ADVICE r04:
    if (full) flush();                    // a fast wave of the next iteration could push a slow one's reading over the limit)
  }
  flush();
}

} // namespace ntamd
