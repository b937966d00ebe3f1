// seed_extend_kernel.hpp -- nthip_seed_extend: the batched graph-extension query through spaced seeds.
//
// For each of n windows of k bases the n_seeds x m2 hash values of its 4 successors and 4 predecessors -- what
//     nthash::BlindSeedNtHash h(kmer, seeds, m2, k);  h.roll(c);  /  h.roll_back(c);      (src/seed.cpp:666-737)
// leave in h.hashes() for c = A, C, G, T, one object and one call at a time in the reference.
//   forward   NTMSM64 (src/seed.cpp:177-207) rolls every block and adds the monomers of the NEW window: the masked formula
//             of kmer[1..k) + c;
//   backward  the blocks are those of the new window c + kmer[0..k-1), the monomers are read at kmer_seq[pos + 1] of the
//             deque with c pushed in front -- the OLD window (src/seed.cpp:195-198 reused by ntmsm64l): reproduced,
//             prev = blocks(new window) ^ monomers(old window).
// One lane per window.  The 64 windows of a wave are packed to a 2-bit stream in LDS; a window is G = ceil(k / 16) words
// in registers, its neighbours are those words shifted by one base.  A masked strand pair comes from the k-independent
// 16-mer tables of first_window.hpp (the any-seed form of seed_kernels.hpp: four pre-rotated copies, one Horner step per
// four words) under one of three masks per seed -- every contributing position, those covered by blocks, the monomers.
#pragma once

#include <hip/hip_runtime.h>

#include "seed_kernels.hpp"

namespace ntamd {

struct SeedExtendArgs {
  const uint8_t* kmers;       // n * k bytes
  uint64_t n;
  const uint4* fw;            // the k-independent first-window tables (FW_ENTRIES)
  const uint32_t* mask;       // [3][n_seeds][G]: 0 every contributing position, 1 positions under blocks, 2 monomers
  const uint4* acorr;         // [3][n_seeds][G]: what the masked-out positions of a word contribute as code 0
  uint64_t* self;             // [n][n_seeds * m2] or NULL
  uint64_t* next;             // [n][4][n_seeds * m2] or NULL
  uint64_t* prev;             // likewise
  uint32_t k, n_seeds, m2, G, bits_dwords;
};

template <int GMAX>
__global__ __launch_bounds__(256) void seed_extend_kernel(const SeedExtendArgs a)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t k = a.k, G = a.G, S = a.n_seeds;
  const uint32_t n_grp = 3u * S * G;
  uint4* tabs = (uint4*)lds_dyn;
  uint4* g_acorr = tabs + SA_TAB_ENTRIES;
  uint32_t* g_mask = (uint32_t*)(g_acorr + n_grp);
  uint32_t* bits = g_mask + ((n_grp + 3u) & ~3u) + wave * a.bits_dwords;
  sa_load(tabs, a.fw, a.acorr, a.mask, n_grp, G, tid, blockDim.x);
  __syncthreads();
  const uint32_t k31 = k % 31u, k33 = k % 33u;
  const uint64_t kmul = (uint64_t)k * MULTISEED;
  const uint32_t per = S * a.m2;
  auto code_of = [](uint32_t b) { return b ^ (b >> 1); }; // A, C, G, T -> 0, 1, 3, 2 as pack4 codes them (code ^ 2: the complement)

  // the strand pair {F.lo, F.hi, R.lo, R.hi} of the window w[] under mask set `type` of seed s
  auto strands = [&](const uint32_t* w, uint32_t type, uint32_t s) -> uint4 {
    const uint32_t base = (type * S + s) * G;
    uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int c = (GMAX + 3) / 4 - 1; c >= 0; --c) {
      if (4u * (uint32_t)c >= G) continue; // (uniform)
      uint4 x = make_uint4(0, 0, 0, 0);
#pragma unroll
      for (uint32_t q = 0; q < 4; ++q) {
        const uint32_t g = 4u * (uint32_t)c + q;
        if (g < (uint32_t)GMAX && g < G) {
          const uint4 e = fw_word16(tabs + q * 1024u, w[g] & g_mask[base + g]), ac = g_acorr[base + g];
          x.x ^= e.x ^ ac.x; x.y ^= e.y ^ ac.y; x.z ^= e.z ^ ac.z; x.w ^= e.w ^ ac.w;
        }
      }
      sror_var(acc.x, acc.y, 2u, 31u); // 64 bases: 64 mod 31, 64 mod 33
      srol_var(acc.z, acc.w, 2u, 31u);
      acc.x ^= x.x; acc.y ^= x.y; acc.z ^= x.z; acc.w ^= x.w;
    }
    srol_var(acc.x, acc.y, k31, k33);
    return acc;
  };
  auto emit = [&](uint64_t* dst, const uint4 st) { // m2 values of one seed (src/internal.hpp:104-118)
    const uint64_t h0 = canon_pair(st.x, st.y, st.z, st.w);
    dst[0] = h0;
    for (uint32_t j = 1; j < a.m2; ++j) dst[j] = mix_hash(h0, (uint64_t)j ^ kmul);
  };

  const uint64_t n_groups = (a.n + 63u) / 64u;
  for (uint64_t grp = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wave; grp < n_groups; grp += (uint64_t)gridDim.x * (blockDim.x >> 6)) {
    const uint64_t i0 = grp * 64u;
    const uint32_t here = a.n - i0 < 64u ? (uint32_t)(a.n - i0) : 64u;
    // ---- the wave's windows as a 2-bit stream: byte j of the group = base j ----
    const uint32_t n_bytes = here * k;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t d = lane; d < a.bits_dwords; d += 64u) { // 16 bases per dword
      uint32_t word = 0;
#pragma unroll
      for (uint32_t q = 0; q < 4; ++q) {
        uint32_t four = 0;
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
          const uint32_t j = 16u * d + 4u * q + u;
          const uint32_t byte = j < n_bytes ? a.kmers[i0 * k + j] : (uint32_t)'A';
          four |= byte << (8u * u);
        }
        uint32_t bad = 0;
        word |= pack4(four, bad) << (8u * q);
      }
      bits[d] = word;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    if (lane >= here) continue;
    const uint64_t i = i0 + lane;
    const uint32_t b0 = lane * k, d0 = b0 >> 4, sh = (b0 & 15u) << 1;
    uint32_t X[GMAX + 1];
#pragma unroll
    for (uint32_t g = 0; g <= (uint32_t)GMAX; ++g) X[g] = g <= G ? funnel(bits[d0 + g + 1u], bits[d0 + g], sh) : 0u;
    // (X[G] and the bases past k in X[G - 1] belong to the next window: every mask is zero there)
    uint32_t Nw[GMAX], Pw[GMAX];
#pragma unroll
    for (uint32_t g = 0; g < (uint32_t)GMAX; ++g) {
      Nw[g] = funnel(X[g + 1], X[g], 2u);                       // the window moved on by one base
      Pw[g] = g == 0 ? X[0] << 2 : funnel(X[g], X[g - 1], 30u);  // ... and back by one
    }
    const uint32_t lw = (k - 1u) >> 4, ls = ((k - 1u) & 15u) << 1; // where a successor's new base goes
    for (uint32_t s = 0; s < S; ++s) {
      if (a.self) emit(a.self + i * per + s * a.m2, strands(X, 0u, s));
      if (a.next) {
        for (uint32_t b = 0; b < 4; ++b) {
          uint32_t w[GMAX];
#pragma unroll
          for (uint32_t g = 0; g < (uint32_t)GMAX; ++g) w[g] = g == lw ? (Nw[g] & ~(3u << ls)) | (code_of(b) << ls) : Nw[g];
          emit(a.next + (i * 4u + b) * per + s * a.m2, strands(w, 0u, s));
        }
      }
      if (a.prev) {
        const uint4 mono = strands(X, 2u, s); // (the monomers of the window it comes from)
        for (uint32_t b = 0; b < 4; ++b) {
          uint32_t w[GMAX];
#pragma unroll
          for (uint32_t g = 0; g < (uint32_t)GMAX; ++g) w[g] = g == 0 ? (Pw[0] | code_of(b)) : Pw[g];
          uint4 st = strands(w, 1u, s);
          st.x ^= mono.x; st.y ^= mono.y; st.z ^= mono.z; st.w ^= mono.w;
          emit(a.prev + (i * 4u + b) * per + s * a.m2, st);
        }
      }
    }
  }
}

} // namespace ntamd
