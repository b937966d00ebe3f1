// kmer_runs_na_kernel.hpp -- the N-aware sibling of kmer_runs_kernel: fixed-length
// reads that contain non-ACGTU bytes, hashed at (nearly) the clean-path rate with
// the reference's emission rule and a COMPACT output stream.
//
// NtHash emits exactly the windows whose k bytes are all bases (the net effect
// of init()/roll(), src/kmer.cpp:228-264).  That is a
// parallel predicate, so the run-split decomposition still works:
//   pass 1 (MODE_COUNT)  per wave tile (64 runs of up to C windows, geometry of
//                        kmer_runs_gen_kernel.hpp but with DISJOINT runs: the last
//                        run of a read is short): a validity bit per base, OR-ed
//                        over every window by doubling, popcount -> valid windows
//                        per tile (and per read, on request);
//   host                 exclusive scan of the tile counts -> tile offsets;
//   pass 2 (MODE_HASH)   the same tiles are hashed as in kmer_runs_kernel (a
//                        non-base keeps a garbage 2-bit code; it enters and
//                        leaves the rolled state with the same code, so every
//                        all-base window is exact), each lane drops its valid
//                        hashes at its compacted slot of the wave's LDS tile
//                        (wave-level exclusive scan of the per-lane counts), and
//                        the tile is copied out contiguously at its offset.
// Optional get_pos() stream through a second LDS tile.
#pragma once

#include <hip/hip_runtime.h>

#include "kmer_runs_kernel.hpp"

namespace ntamd {

enum : int { NA_MODE_COUNT = 1, NA_MODE_HASH = 2 };

struct KmerRunsNaArgs {
  const uint8_t* seqs;
  uint64_t* hashes;          // compact [emitted k-mer][m]
  uint32_t* pos;             // optional: position of every emitted k-mer in its read
  uint64_t* counts;          // optional (count pass, zeroed by the host): per-read emitted windows
  uint64_t* tile_counts;     // count pass out: valid windows per wave tile
  const uint64_t* tile_off;  // hash pass in: exclusive scan of tile_counts
  const uint4* init_tab;
  uint64_t n_reads, n_runs, n_wtiles;
  uint32_t len, stride, k, m;
  uint32_t nwin, C, rpr, ntab;   // rpr = ceil(nwin / C)
  uint32_t waves, bits_dwords, vbits_dwords, tile_u64;
  uint32_t inv_rpr, last_cnt;    // last_cnt = nwin - (rpr - 1) * C: windows of a read's last run
  uint64_t tab[16][2];
  uint64_t mult[KF_MAX_RUNTIME_M];
};

// 4 ASCII bytes -> 4 x 2-bit codes (one byte) and a 4-bit mask of the non-bases
__device__ __forceinline__ uint32_t pack4v(uint32_t w, uint32_t& inv4)
{
  const uint32_t t = (w >> 1) & 0x03030303u;
  uint32_t x = w | 0x20202020u;
  const uint32_t ubit = (x >> 4) & 0x01010101u;
  x = x & ~ubit;
  const uint32_t canon = __builtin_amdgcn_perm(0u, 0x67746361u, t);
  const uint32_t d = x ^ canon;                                   // byte != 0 <=> not a base
  const uint32_t nz = (((d & 0x7f7f7f7fu) + 0x7f7f7f7fu) | d) & 0x80808080u;
  inv4 = __builtin_amdgcn_udot4(nz >> 7, 0x08040201u, 0u, false); // gather the four flags
  return __builtin_amdgcn_udot4(t, 0x40100401u, 0u, false);
}

// Which of the (up to 32) windows starting at bases b0, b0+1, ... hold a non-base.
// vw: validity bit stream (1 = not a base); reads 128 bits from the dword of b0 on.
// OR over k <= 64 consecutive bits by doubling, on a 96-bit register pair.
__device__ __forceinline__ uint32_t windows_with_non_base(const uint32_t* vw, uint32_t b0, uint32_t k)
{
  const uint32_t dw = b0 >> 5, sh = b0 & 31u;
  const uint32_t x0 = vw[dw], x1 = vw[dw + 1], x2 = vw[dw + 2], x3 = vw[dw + 3];
  uint64_t lo = ((uint64_t)funnel(x2, x1, sh) << 32) | funnel(x1, x0, sh);
  uint64_t hi = funnel(x3, x2, sh); // bits 64..95
  auto fold = [&](uint32_t s) {     // bit j |= bit j + s, 1 <= s <= 32
    lo |= (lo >> s) | (hi << (64u - s));
    hi |= hi >> s;
  };
  uint32_t span = 1;
  while (2u * span <= k) {
    fold(span);
    span *= 2u;
  }
  if (k > span) fold(k - span);
  return (uint32_t)lo; // bit j exact for j + k - 1 <= 95
}

template <int MODE, int NW>
__global__ __launch_bounds__(KR_MAX_THREADS) void kmer_runs_na_kernel(const KmerRunsNaArgs a)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  const uint32_t k = a.k, m = a.m, C = a.C, rpr = a.rpr;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;

  // LDS: init tables | pair table | multipliers | per wave {hash tile, pos tile, bits, validity bits}
  uint4* itab = (uint4*)lds_dyn;
  uint4* ptab = itab + a.ntab * 256u;
  uint64_t* mults = (uint64_t*)(ptab + 16);
  const uint32_t per_wave = a.tile_u64 * 2u + a.tile_u64 + a.bits_dwords + a.vbits_dwords;
  uint32_t* wave_base = (uint32_t*)(mults + KF_MAX_RUNTIME_M) + wave * per_wave;
  uint64_t* tile = (uint64_t*)wave_base;
  uint32_t* ptile = wave_base + a.tile_u64 * 2u;
  uint32_t* bits = ptile + a.tile_u64;
  uint16_t* vbits = (uint16_t*)(bits + a.bits_dwords);

  if (MODE == NA_MODE_HASH) {
    for (uint32_t i = tid; i < a.ntab * 256u; i += blockDim.x) itab[i] = a.init_tab[i];
    if (tid < 16)
      ptab[tid] = make_uint4((uint32_t)a.tab[tid][0], (uint32_t)(a.tab[tid][0] >> 32),
                             (uint32_t)a.tab[tid][1], (uint32_t)(a.tab[tid][1] >> 32));
    if (tid < KF_MAX_RUNTIME_M) mults[tid] = a.mult[tid];
  }
  __syncthreads();

  auto lds_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  };

  // every block streams through its own contiguous range of tiles (as kmer_runs_kernel)
  const uint64_t per_block = (a.n_wtiles + gridDim.x - 1) / gridDim.x;
  const uint64_t t_begin = (uint64_t)blockIdx.x * per_block;
  const uint64_t t_end = t_begin + per_block < a.n_wtiles ? t_begin + per_block : a.n_wtiles;
  for (uint64_t wt = t_begin + wave; wt < t_end; wt += a.waves) {
    const uint64_t g0 = wt * 64u;
    const uint64_t r_first = g0 / rpr;
    const uint32_t rem0 = (uint32_t)(g0 - r_first * rpr);
    const uint64_t runs_left = a.n_runs - g0;
    const uint32_t runs_here = runs_left < 64u ? (uint32_t)runs_left : 64u;
    // gl = run index counted from run 0 of read r_first (gl < 64 + rpr) -> read, run in read
    auto split = [&](uint32_t gl, uint32_t& lr_, uint32_t& q_) {
      lr_ = rpr > 64u ? (gl >= rpr ? 1u : 0u) : (gl * a.inv_rpr) >> 16;
      q_ = gl - lr_ * rpr;
    };
    // the slab: first base of the first run .. last base of the last run's last window
    uint32_t lre, qe;
    split(rem0 + runs_here - 1u, lre, qe);
    const uint64_t off = r_first * a.stride + (uint64_t)rem0 * C;
    const uint32_t shift = (uint32_t)(((uint64_t)a.seqs + off) & 15u);
    const uint32_t slab_bytes = lre * a.stride + qe * C + (qe == rpr - 1u ? a.last_cnt : C) + k - 1u - rem0 * C;
    const uint32_t n_vec = (shift + slab_bytes + 15u) >> 4;
    lds_sync();
    // ---- stage: codes + one validity bit per base (bytes outside the slab are
    // never part of a window of this tile, so they need no special treatment)
    for (uint32_t i = lane; i < n_vec; i += 64u) {
      const uint4 v = *(const uint4*)(a.seqs + (off - shift) + ((uint64_t)i << 4));
      uint32_t i0, i1, i2, i3;
      const uint32_t c0 = pack4v(v.x, i0), c1 = pack4v(v.y, i1), c2 = pack4v(v.z, i2), c3 = pack4v(v.w, i3);
      bits[i] = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
      vbits[i] = (uint16_t)(i0 | (i1 << 4) | (i2 << 8) | (i3 << 12));
    }
    if (lane < (uint32_t)NW + 3u) bits[n_vec + lane] = 0;
    if (lane < 10u) vbits[n_vec + lane] = 0xFFFFu; // beyond the slab: not a base
    lds_sync();

    // ---- this lane's run and the validity of its C windows ----------------------
    const bool live = lane < runs_here;
    uint32_t lr, q;
    split(live ? rem0 + lane : rem0, lr, q);
    const uint32_t b0 = shift + lr * a.stride + q * C - rem0 * C;
    const uint32_t c_run = live ? (q == rpr - 1u ? a.last_cnt : C) : 0u; // C <= 16
    const uint32_t valid = ~windows_with_non_base((const uint32_t*)vbits, b0, k) & ((1u << c_run) - 1u);
    const uint32_t cnt = __builtin_popcount(valid);
    // wave exclusive scan of the per-lane counts
    uint32_t incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(incl, d, 64);
      if ((int)lane >= d) incl += o;
    }
    const uint32_t lane_off = incl - cnt;
    const uint32_t total = __shfl(incl, 63, 64);

    if (MODE == NA_MODE_COUNT) {
      if (lane == 0) a.tile_counts[wt] = total;
      if (a.counts && live && cnt) atomicAdd((unsigned long long*)&a.counts[r_first + lr], (unsigned long long)cnt);
      continue;
    }

    // ---- hash the run (as kmer_runs_kernel) and drop valid hashes at their slots --
    const uint32_t d0 = b0 >> 4, sh0 = (b0 & 15u) << 1;
    uint32_t w[NW];
    {
      uint32_t lo = bits[d0];
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        const uint32_t hi = bits[d0 + i + 1];
        w[i] = funnel(hi, lo, sh0);
        lo = hi;
      }
    }
    uint32_t f_lo = 0, f_hi = 0, r_lo = 0, r_hi = 0;
#pragma unroll
    for (int jt = 0; jt < 4 * NW; ++jt) {
      if ((uint32_t)jt < a.ntab) {
        const uint32_t byte = (w[jt >> 2] >> ((jt & 3) * 8)) & 0xFFu;
        const uint4 e = itab[(uint32_t)jt * 256u + byte];
        f_lo ^= e.x; f_hi ^= e.y; r_lo ^= e.z; r_hi ^= e.w;
      }
    }
    uint32_t slot = lane_off;
    const bool want_pos = a.pos != nullptr;
    auto emit = [&](uint32_t j) {
      if ((valid >> j) & 1u) {
        tile[slot] = (((uint64_t)f_hi << 32) | f_lo) + (((uint64_t)r_hi << 32) | r_lo);
        if (want_pos) ptile[slot] = q * C + j;
        ++slot;
      }
    };
    emit(0u);
    const uint32_t bi = b0 + k;
    const uint32_t di = bi >> 4, shi = (bi & 15u) << 1;
    for (uint32_t jw = 0; jw * 16u + 1u < C; ++jw) {
      const uint32_t w_in = funnel(bits[di + jw + 1], bits[di + jw], shi);
      const uint32_t w_out = funnel(bits[d0 + jw + 1], bits[d0 + jw], sh0);
      const uint32_t u = ((w_in & 0x33333333u) << 2) | (w_out & 0x33333333u);
      const uint32_t v = (w_in & 0xCCCCCCCCu) | ((w_out >> 2) & 0x33333333u);
      auto lookup = [&](uint32_t i) -> uint4 {
        const uint32_t src = (i & 1u) ? v : u;
        const uint32_t toff = ((src >> ((i >> 1) * 4u)) & 0xFu) << 4;
        return *(const uint4*)((const char*)ptab + toff);
      };
      auto roll = [&](const uint4 term) {
        srol_pair(f_lo, f_hi);
        f_lo ^= term.x;
        f_hi ^= term.y;
        r_lo ^= term.z;
        r_hi ^= term.w;
        sror_pair(r_lo, r_hi);
      };
      // table terms do not depend on the hash state: fetch a batch ahead of the dependent chain
      auto batch = [&](uint32_t i0, auto n_tag) {
        constexpr uint32_t N = decltype(n_tag)::value;
        uint4 terms[N];
#pragma unroll
        for (uint32_t i = 0; i < N; ++i) terms[i] = lookup(i0 + i);
#pragma unroll
        for (uint32_t i = 0; i < N; ++i) {
          roll(terms[i]);
          emit(jw * 16u + i0 + i + 1u);
        }
      };
      const uint32_t left = C - 1u - jw * 16u;
      const uint32_t ns = left < 16u ? left : 16u;
      uint32_t i0 = 0;
      for (; i0 + 8u <= ns; i0 += 8u) batch(i0, std::integral_constant<uint32_t, 8u>{});
      switch (ns - i0) {
        case 1: batch(i0, std::integral_constant<uint32_t, 1u>{}); break;
        case 2: batch(i0, std::integral_constant<uint32_t, 2u>{}); break;
        case 3: batch(i0, std::integral_constant<uint32_t, 3u>{}); break;
        case 4: batch(i0, std::integral_constant<uint32_t, 4u>{}); break;
        case 5: batch(i0, std::integral_constant<uint32_t, 5u>{}); break;
        case 6: batch(i0, std::integral_constant<uint32_t, 6u>{}); break;
        case 7: batch(i0, std::integral_constant<uint32_t, 7u>{}); break;
        default: break;
      }
    }
    lds_sync();
    // ---- copy out `total` k-mers (x m values) at this tile's offset -----------------
    const uint64_t o0 = a.tile_off[wt];
    uint64_t* out0 = a.hashes + o0 * m;
    if (m == 1) {
      // 16-byte pieces where the (arbitrary) tile offset allows: a possibly odd first
      // element, then pairs, then a possibly odd last one
      const uint32_t head = (uint32_t)(o0 & 1u) < total ? (uint32_t)(o0 & 1u) : total;
      if (lane == 0 && head) out0[0] = tile[0];
      const uint32_t n_pairs = (total - head) >> 1;
      for (uint32_t pi = lane; pi < n_pairs; pi += 64u) {
        const uint64_t x = tile[head + 2u * pi], y = tile[head + 2u * pi + 1u];
        *(uint4*)(out0 + head + 2u * pi) = make_uint4((uint32_t)x, (uint32_t)(x >> 32), (uint32_t)y, (uint32_t)(y >> 32));
      }
      if (lane == 0 && ((total - head) & 1u)) out0[total - 1u] = tile[total - 1u];
    } else {
      const uint32_t nv = total * m;
      for (uint32_t vi = lane; vi < nv; vi += 64u) {
        const uint32_t e = vi / m, jj = vi - e * m;
        const uint64_t h0 = tile[e];
        out0[vi] = jj == 0 ? h0 : mix_hash(h0, mults[jj & (KF_MAX_RUNTIME_M - 1)]);
      }
    }
    if (a.pos)
      for (uint32_t e = lane; e < total; e += 64u) a.pos[o0 + e] = ptile[e];
  }
}

} // namespace ntamd
