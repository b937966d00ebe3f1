// minimizer_fused_kernel.hpp -- (w, k)-minimizers of fixed-length reads in ONE pass over the bases: the canonical hashes
// never reach HBM (round 4).  Until round 3 the reads were hashed to a stream (8 B per k-mer written),
// the stream read back by minimizer_reg_kernel (one window POSITION per lane: ~96 lane-instructions per k-mer) and the
// picks gathered by a third kernel.
//
// Here the hashing geometry of the run-split kernels is kept -- a lane owns a run of C consecutive windows of one read,
// first window from the byte tables (src/kmer.cpp:43-73,123-152), the others rolled (src/kmer.cpp:84-94,164-174) -- and
// the sliding minimum is computed block-wise on that geometry (van Herk / Gil-Werman with the lane's run as the block,
// C <= w):
//   * forward, while rolling: the prefix arg-min of the run (leftmost on ties) after every window -> one byte per window
//     in LDS; the hashes themselves go to a wave-private LDS tile [lane][C];
//   * backward over the run: the suffix arg-min in registers; the window that starts at local i ends in block l + mm at
//     local j  (r = (w-1) mod C, m0 = (w-1) div C:  j = r + i, mm = m0, and j -= C, ++mm when j >= C), so its minimum is
//         min( suffix_l[i] , full blocks l+1 .. l+mm-1 , prefix_{l+mm}[j] )        (ties: the leftmost candidate)
//     -- two LDS lookups (the prefix arg-min byte, then its hash) and two 64-bit compares per window, whatever w is;
//     the middle blocks' minima (w > 2C) are two per-LANE values made once per tile;
//   * a window picks a new minimizer where its arg-min differs from its left neighbour's; the picks of a tile are counted,
//     parked in a small LDS stash and placed by the block-round look-back of block_rounds.hpp (one round later, nobody
//     waits), and written once: min_hashes, min_pos and min_offsets in their final places.  (Until the stash this kernel
//     looked back per TILE with the waves waiting: 14-20 ms per 20 M reads, of which the hashing and the sweeps were 9.)
// A tile is R = floor(64 / rpr) WHOLE reads (rpr = ceil(nwin / C) blocks each; the window slots past nwin hash whatever
// follows and no valid window covers them), so no window crosses a wave.  N-aware: see the staging of a tile.
#pragma once

#include <hip/hip_runtime.h>

#include "block_rounds.hpp"
#include "kmer_runs_gen_kernel.hpp"

namespace ntamd {

constexpr uint32_t MZF_ROWS = 65;     // 64 lanes + one slack row (what an invalid window may look at)
constexpr uint32_t MZF_FULL = 72;     // per-block minima: 64 + the furthest middle block
constexpr uint32_t MZF_MAX_MM = 7;    // rowdelta of a pick code: 3 bits above the 5-bit column
constexpr int MZF_MAX_THREADS = 768;  // 12 waves per block: 170 registers a lane
#ifndef MZF_ABL_NOWRITE
#define MZF_ABL_NOWRITE 0 // ablation: the picks are not written
#endif
constexpr unsigned long long MZF_FLAG_A = 1ull << 62; // a tile's own count is known
constexpr unsigned long long MZF_FLAG_P = 1ull << 63; // ... and the count of everything up to and including it
constexpr unsigned long long MZF_VALUE = (1ull << 62) - 1ull;

struct MinimizerFusedArgs {
  const uint8_t* seqs;
  const uint4* init_tab;        // [4 NW][256] {f.lo, f.hi, r.lo, r.hi}
  uint32_t* dirty;              // set when a non-base is seen
  unsigned long long* status;   // [(n_rounds + 1) * blocks] look-back words of the block-rounds, zeroed by the host
  uint32_t* abort;              // zeroed by the host; set by a wave that waited too long for a predecessor
  uint32_t timeout_us;          // 0: the 50 ms of block_rounds.hpp (tests shorten it)
  uint64_t* out_hashes;
  uint32_t* out_pos;            // may be NULL
  uint64_t* out_offsets;        // [n_reads + 1]
  uint64_t* total;              // device: the number of picks (also when capacity is smaller)
  uint64_t capacity;
  uint64_t n_reads, total_bytes;
  uint32_t n_tiles, n_rounds, stash_cap;
  uint32_t len, k, w, nwin, nwv; // nwv = nwin - w + 1 window starts
  uint32_t C, rpr, inv_rpr, R;   // run length (block), blocks per read, floor(65536 / rpr) + 1, reads per tile
  uint32_t m0, r;                // (w - 1) div C, (w - 1) mod C
  uint32_t extra;                // rpr * C - nwin: window slots past the read's last window
  uint32_t waves, bits_dwords, pitch_h, pitch_b, per_wave_dwords;
  uint64_t tab[16][2];
};

__device__ __forceinline__ uint32_t wave_incl_add32(uint32_t v)
{
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
  return v;
}

// NW: window words, k <= 16 NW; MID: w > C + 1 may put whole blocks between a window's first and last one
template <int NW, bool MID>
__global__ __launch_bounds__(MZF_MAX_THREADS) void minimizer_fused_kernel(const MinimizerFusedArgs a)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  const uint32_t k = a.k, C = a.C, rpr = a.rpr;
  constexpr uint32_t ntab = 4u * (uint32_t)NW;
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t pitch_h = a.pitch_h, pitch_b = a.pitch_b;
  const uint64_t seqs_addr = (uint64_t)a.seqs;

  // LDS: first-window tables | pair table | per wave { H tile, block minima, prefix arg-min bytes, pick codes, bit stream }
  uint4* itab = (uint4*)lds_dyn;
  uint4* ptab = itab + ntab * 256u;
  uint32_t* ctrl = (uint32_t*)(ptab + 16); // (block_rounds.hpp)
  uint32_t* wave_base = ctrl + BR_CTRL_DWORDS + wave * a.per_wave_dwords;
  uint64_t* H = (uint64_t*)wave_base;                       // [MZF_ROWS][pitch_h]
  uint64_t* fullh = H + MZF_ROWS * pitch_h;                 // [MZF_FULL]
  uint32_t* fullc = (uint32_t*)(fullh + MZF_FULL);          // [MZF_FULL]
  uint8_t* pmi = (uint8_t*)(fullc + MZF_FULL);              // [MZF_ROWS][pitch_b]
  uint8_t* pwt = pmi + MZF_ROWS * pitch_b;                  // [64][pitch_b]
  uint64_t* stash_h = (uint64_t*)(pmi + (((MZF_ROWS + 64u) * pitch_b + 7u) & ~7u)); // [stash_cap] a tile's picks, parked for a round
  uint16_t* stash_p = (uint16_t*)(stash_h + a.stash_cap);   // [stash_cap] (stash_cap is even)
  uint32_t* bits = (uint32_t*)(stash_p + a.stash_cap);
  uint16_t* vbits = (uint16_t*)(bits + a.bits_dwords);      // [bits_dwords + 8] validity bits of a slab that holds a non-base

  for (uint32_t i = tid; i < ntab * 256u; i += blockDim.x) itab[i] = a.init_tab[i];
  if (tid < 16)
    ptab[tid] = make_uint4((uint32_t)a.tab[tid][0], (uint32_t)(a.tab[tid][0] >> 32), (uint32_t)a.tab[tid][1],
                           (uint32_t)(a.tab[tid][1] >> 32));
  // the slack row of the arg-min bytes is never written by a tile: a window that looks there is not a valid one, but
  // the byte it finds indexes the H tile
  for (uint32_t i = lane; i < pitch_b; i += 64u) pmi[64u * pitch_b + i] = 0;
  for (uint32_t i = lane; i < MZF_FULL; i += 64u) {
    fullh[i] = ~0ull;
    fullc[i] = 0;
  }
  if (tid < BR_CTRL_DWORDS) ctrl[tid] = 0;
  __syncthreads(); // the only block-wide barrier
  BlockRounds rounds;
  rounds.init(ctrl, lane, wave, a.waves, a.n_rounds, a.status, a.abort, a.total, a.out_offsets + a.n_reads, a.timeout_us);

  auto lds_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  };

  // this lane's place in a tile: block q of read lr (relative to the tile's first read)
  const uint32_t lr_raw = (lane * a.inv_rpr) >> 16;
  const uint32_t q_raw = lane - lr_raw * rpr;
  const uint32_t m0 = a.m0, r = a.r;
  uint32_t bad = 0;

  const uint32_t n_waves_total = gridDim.x * a.waves;
  // what is parked from the round before
  bool have_prev = false;
  uint32_t prev_rd = 0, prev_total = 0, prev_rel = 0;
  uint64_t prev_read = 0;
  bool prev_first = false;
  auto flush_prev = [&]() {
    lds_sync();
    const uint64_t base = rounds.offset_of(prev_rd);
    for (uint32_t i = lane; i < prev_total && !MZF_ABL_NOWRITE; i += 64u) {
      const uint64_t o = base + i;
      if (o < a.capacity) {
        a.out_hashes[o] = stash_h[i];
        if (a.out_pos) a.out_pos[o] = stash_p[i];
      }
    }
    if (prev_first) a.out_offsets[prev_read] = base + prev_rel;
    have_prev = false;
  };
  uint32_t t = blockIdx.x * a.waves + wave; // tile of round rd: (rd * blocks + block) * waves + wave (n_tiles < 2^31)
  for (uint32_t rd = 0; rd <= a.n_rounds; ++rd, t += n_waves_total) { // (round n_rounds: nothing to hash, the round before is placed)
    if (!(rd < a.n_rounds && t < a.n_tiles)) {
      rounds.arrive_round(rd, 0u);
      if (have_prev) flush_prev();
      continue;
    }

    // ---- stage the tile's slab as a 2-bit stream ----------------------------------------------------------------
    const uint64_t rf = (uint64_t)t * a.R;
    const uint64_t left = a.n_reads - rf;
    const uint32_t reads_here = left < a.R ? (uint32_t)left : a.R;
    const uint64_t start = rf * a.len;
    uint64_t slab64 = (uint64_t)reads_here * a.len + a.extra;
    if (start + slab64 > a.total_bytes) slab64 = a.total_bytes - start;
    const uint32_t slab_bytes = (uint32_t)slab64;
    const uint32_t shift = (uint32_t)((seqs_addr + start) & 15u);
    const uint64_t byte0 = start - shift; // (wraps below 0 by < 16 for an unaligned buffer)
    const uint32_t n_vec = (shift + slab_bytes + 15u) >> 4;
    const bool edge = start < shift || byte0 + ((uint64_t)n_vec << 4) > a.total_bytes;
    lds_sync();
    for (uint32_t i = lane; i < n_vec; i += 64u) {
      const uint4 v = *(const uint4*)(a.seqs + byte0 + ((uint64_t)i << 4));
      uint32_t b = 0;
      const uint32_t p = pack16(v, b);
      if (edge) { // bytes outside the caller's buffer exist only in these slabs: they are not judged
        const int32_t lo_cut = (int32_t)shift - (int32_t)(i << 4);
        const int32_t hi_cut = (int32_t)(shift + slab_bytes) - (int32_t)(i << 4);
        if (lo_cut > 0 || hi_cut < 16) {
          uint32_t bx[4] = {0, 0, 0, 0};
          (void)pack4(v.x, bx[0]);
          (void)pack4(v.y, bx[1]);
          (void)pack4(v.z, bx[2]);
          (void)pack4(v.w, bx[3]);
          b = 0;
#pragma unroll
          for (int qq = 0; qq < 16; ++qq)
            if (qq >= lo_cut && qq < hi_cut) b |= (bx[qq >> 2] >> ((qq & 3) * 8)) & 0xFFu;
        }
      }
      bad |= b;
      bits[i] = p;
    }
    for (uint32_t i = n_vec + lane; i < a.bits_dwords; i += 64u) bits[i] = 0; // the rolls of a last block read ahead
    // N-aware (round 4, as minimizer_w_kernel.hpp): a k-mer that holds a non-base is no candidate (NtHash does not emit it:
    // src/kmer.cpp:228-264) -- its hash becomes the largest value, a window of such k-mers only picks nothing.  A tile
    // without a non-base (nearly all of them) pays one ballot.
    const bool tile_dirty = __ballot(bad != 0u) != 0ull;
    bad = 0;
    if (tile_dirty) {
      for (uint32_t i = lane; i < n_vec; i += 64u) { // (the slab comes from L2 this time)
        const uint4 v = *(const uint4*)(a.seqs + byte0 + ((uint64_t)i << 4));
        uint32_t i0, i1, i2, i3;
        (void)pack4v(v.x, i0);
        (void)pack4v(v.y, i1);
        (void)pack4v(v.z, i2);
        (void)pack4v(v.w, i3);
        vbits[i] = (uint16_t)(i0 | (i1 << 4) | (i2 << 8) | (i3 << 12));
      }
      for (uint32_t i = n_vec + lane; i < a.bits_dwords + 8u; i += 64u) vbits[i] = 0;
    }
    lds_sync();

    // ---- phase 1: hash the run, prefix arg-min on the way ---------------------------------------------------------
    const bool live = lr_raw < reads_here;
    const uint32_t lr = live ? lr_raw : 0u, q = live ? q_raw : 0u; // (idle lanes redo the tile's first run)
    const uint32_t b0 = shift + lr * a.len + q * C;
    // (window slots past the read's last window hash whatever follows: no valid window ever covers them)
    const uint32_t inval = tile_dirty ? windows_with_non_base((const uint32_t*)vbits, b0, k) : 0u; // bit j: k-mer j of the run holds a non-base
    uint64_t* const my_row = H + lane * pitch_h;
    uint8_t* const my_pm = pmi + lane * pitch_b;
    uint8_t* const my_pw = pwt + lane * pitch_b;
    uint64_t ph = ~0ull;
    uint32_t pidx = 0;

    const uint32_t d0 = b0 >> 4, sh0 = (b0 & 15u) << 1;
    uint32_t f_lo, f_hi, r_lo, r_hi;
    {
      uint32_t wv[NW];
      uint32_t lo = bits[d0];
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        const uint32_t hi = bits[d0 + i + 1];
        wv[i] = funnel(hi, lo, sh0);
        lo = hi;
      }
      uint4 e[4 * NW];
#pragma unroll
      for (int jt = 0; jt < 4 * NW; ++jt) e[jt] = itab[(uint32_t)jt * 256u + ((wv[jt >> 2] >> ((jt & 3) * 8)) & 0xFFu)];
      f_lo = e[0].x ^ e[1].x; f_hi = e[0].y ^ e[1].y; r_lo = e[0].z ^ e[1].z; r_hi = e[0].w ^ e[1].w;
#pragma unroll
      for (int jt = 2; jt < 4 * NW; jt += 2) {
        f_lo = __builtin_amdgcn_bitop3_b32(f_lo, e[jt].x, e[jt + 1].x, 0x96);
        f_hi = __builtin_amdgcn_bitop3_b32(f_hi, e[jt].y, e[jt + 1].y, 0x96);
        r_lo = __builtin_amdgcn_bitop3_b32(r_lo, e[jt].z, e[jt + 1].z, 0x96);
        r_hi = __builtin_amdgcn_bitop3_b32(r_hi, e[jt].w, e[jt + 1].w, 0x96);
      }
    }
    auto emit = [&](uint32_t j) {
      uint64_t h = canon_pair(f_lo, f_hi, r_lo, r_hi);
      h = ((inval >> j) & 1u) ? ~0ull : h;
      my_row[j] = h;
      const bool lt = h < ph; // strict: the leftmost of equal hashes stays
      ph = lt ? h : ph;
      pidx = lt ? j : pidx;
      my_pm[j] = (uint8_t)pidx;
    };
    emit(0u);
    {
      const uint32_t bi = b0 + k;
      const uint32_t di = bi >> 4, shi = (bi & 15u) << 1;
      // (C <= 16: one word of incoming / outgoing bases)
      const uint32_t w_in = funnel(bits[di + 1], bits[di], shi);
      const uint32_t w_out = funnel(bits[d0 + 1], bits[d0], sh0);
      const uint32_t u = ((w_in & 0x33333333u) << 2) | (w_out & 0x33333333u);
      const uint32_t v = (w_in & 0xCCCCCCCCu) | ((w_out >> 2) & 0x33333333u);
      auto lookup = [&](uint32_t i) -> uint4 {
        const uint32_t src = (i & 1u) ? v : u;
        const uint32_t off = ((src >> ((i >> 1) * 4u)) & 0xFu) << 4;
        return *(const uint4*)((const char*)ptab + off);
      };
      auto roll = [&](const uint4 term) {
        roll_step(f_lo, f_hi, r_lo, r_hi, term);
      };
      auto batch = [&](uint32_t i0, auto n_tag) {
        constexpr uint32_t N = decltype(n_tag)::value;
        uint4 terms[N];
#pragma unroll
        for (uint32_t i = 0; i < N; ++i) terms[i] = lookup(i0 + i);
#pragma unroll
        for (uint32_t i = 0; i < N; ++i) {
          roll(terms[i]);
          emit(i0 + i + 1u);
        }
      };
      const uint32_t ns = C - 1u;
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

    // ---- the blocks between a window's first and last one (w > C + 1) ----------------------------------------------
    uint64_t mid1_h = ~0ull, mid2_h = ~0ull;
    uint32_t mid1_c = 0, mid2_c = 0;
    if constexpr (MID) {
      fullh[lane] = ph;
      fullc[lane] = pidx;
      lds_sync();
      for (uint32_t tt = 1; tt <= m0; ++tt) {
        const uint32_t bi = lane + tt; // (< MZF_FULL: m0 < MZF_MAX_MM)
        const uint64_t fh = fullh[bi];
        const uint32_t fc = (tt << 5) | fullc[bi];
        if (tt < m0) {
          const bool lt = fh < mid1_h;
          mid1_h = lt ? fh : mid1_h;
          mid1_c = lt ? fc : mid1_c;
        }
        const bool lt2 = fh < mid2_h;
        mid2_h = lt2 ? fh : mid2_h;
        mid2_c = lt2 ? fc : mid2_c;
      }
    } else {
      lds_sync();
    }

    // ---- phase 2: backward over the run: every window's arg-min as a code (rowdelta << 5 | column) ----------------
    uint64_t sh = ~0ull;
    uint32_t sc = 0;          // suffix arg-min (column in this lane's row)
    uint32_t t_prev = ~0u;    // the code of the window to the right
    uint32_t t_last = 0;      // ... of this run's last window
    uint32_t chg = 0;         // bit i: window i's arg-min differs from window i - 1's (i >= 1)
    auto sweep = [&](uint32_t i_top, auto n_tag) {
      constexpr uint32_t N = decltype(n_tag)::value;
      uint64_t hv[N], hb[N];
      uint32_t pi[N], rowb[N], mmv[N];
#pragma unroll
      for (uint32_t uu = 0; uu < N; ++uu) {
        const uint32_t i = i_top - uu;
        uint32_t j = r + i, mm = m0;
        if (j >= C) {
          j -= C;
          ++mm;
        }
        mmv[uu] = mm;
        const uint32_t row = lane + mm < 64u ? lane + mm : 64u;
        rowb[uu] = row;
        pi[uu] = pmi[row * pitch_b + j];
        hv[uu] = my_row[i];
      }
#pragma unroll
      for (uint32_t uu = 0; uu < N; ++uu) hb[uu] = H[rowb[uu] * pitch_h + pi[uu]];
#pragma unroll
      for (uint32_t uu = 0; uu < N; ++uu) {
        const uint32_t i = i_top - uu;
        const bool le = hv[uu] <= sh; // the leftmost of equal hashes wins
        sh = le ? hv[uu] : sh;
        sc = le ? i : sc;
        uint64_t bh = sh;
        uint32_t bc = sc;
        if constexpr (MID) {
          const bool first = mmv[uu] == m0; // (uniform)
          const uint64_t mh = first ? mid1_h : mid2_h;
          const uint32_t mc = first ? mid1_c : mid2_c;
          const bool lt = mh < bh;
          bh = lt ? mh : bh;
          bc = lt ? mc : bc;
        }
        const bool ltb = hb[uu] < bh;
        bc = ltb ? ((mmv[uu] << 5) | pi[uu]) : bc;
        my_pw[i] = (uint8_t)bc;
        chg |= (bc != t_prev ? 1u : 0u) << (i + 1u);
        t_prev = bc;
        if (i == C - 1u) t_last = bc;
      }
    };
    {
      uint32_t i_top = C - 1u;
      uint32_t left_w = C;
      for (; left_w >= 8u; left_w -= 8u, i_top -= 8u) sweep(i_top, std::integral_constant<uint32_t, 8u>{});
      switch (left_w) {
        case 1: sweep(i_top, std::integral_constant<uint32_t, 1u>{}); break;
        case 2: sweep(i_top, std::integral_constant<uint32_t, 2u>{}); break;
        case 3: sweep(i_top, std::integral_constant<uint32_t, 3u>{}); break;
        case 4: sweep(i_top, std::integral_constant<uint32_t, 4u>{}); break;
        case 5: sweep(i_top, std::integral_constant<uint32_t, 5u>{}); break;
        case 6: sweep(i_top, std::integral_constant<uint32_t, 6u>{}); break;
        case 7: sweep(i_top, std::integral_constant<uint32_t, 7u>{}); break;
        default: break;
      }
    }
    // window 0 of the run against the last window of the run before (the lane below): that one's code is relative to
    // ITS row -- the same position when it lies one row up and says the same column
    const uint32_t below = (uint32_t)__shfl_up((int)t_last, 1, 64);
    const bool same0 = q != 0u && below >= 32u && below - 32u == t_prev; // (t_prev: the code of window 0 now)
    const uint32_t first_w = q * C; // the run's first window inside its read
    const uint32_t nv = live && a.nwv > first_w ? (a.nwv - first_w < C ? a.nwv - first_w : C) : 0u;
    uint32_t flags = ((chg & ~1u) | (same0 ? 0u : 1u)) & ((1u << nv) - 1u);

    if (tile_dirty) { // a window whose every k-mer holds a non-base picked one of them: that pick is none
      lds_sync();
      uint32_t f = flags;
      while (f != 0u) {
        const uint32_t i = (uint32_t)__builtin_ctz(f);
        f &= f - 1u;
        const uint32_t code = my_pw[i];
        if (H[(lane + (code >> 5)) * pitch_h + (code & 31u)] == ~0ull) flags &= ~(1u << i);
      }
    }
    // ---- phase 3: count the tile's picks, arrive, place the tile before, park this one's --------------------------
    const uint32_t cnt = (uint32_t)__builtin_popcount(flags);
    const uint32_t incl = wave_incl_add32(cnt);
    const uint32_t tile_total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    rounds.arrive_round(rd, tile_total);
    if (have_prev) flush_prev();
    const bool park = tile_total <= a.stash_cap;
    const bool first_block = live && q == 0u;
    uint64_t base = 0;
    if (!park) { // (reads of one repeated base: every window a new pick) this round's look-back now, the picks straight out
      rounds.try_lead(rd);
      base = rounds.offset_of(rd);
    }
    uint32_t o = incl - cnt;
    lds_sync();
    while (__ballot(flags != 0u) != 0ull) {
      if (flags != 0u) {
        const uint32_t i = (uint32_t)__builtin_ctz(flags);
        flags &= flags - 1u;
        const uint32_t code = my_pw[i];
        const uint32_t rdl = code >> 5, col = code & 31u;
        const uint64_t h = H[(lane + rdl) * pitch_h + col];
        const uint32_t pos = (q + rdl) * C + col;
        if (park) {
          stash_h[o] = h;
          stash_p[o] = (uint16_t)pos;
        } else if (base + o < a.capacity && !MZF_ABL_NOWRITE) {
          a.out_hashes[base + o] = h;
          if (a.out_pos) a.out_pos[base + o] = pos;
        }
        ++o;
      }
    }
    if (park) {
      have_prev = true;
      prev_rd = rd;
      prev_total = tile_total;
      prev_rel = incl - cnt;
      prev_read = rf + lr;
      prev_first = first_block;
    } else if (first_block) {
      a.out_offsets[rf + lr] = base + (incl - cnt);
    }
  }
}

} // namespace ntamd
