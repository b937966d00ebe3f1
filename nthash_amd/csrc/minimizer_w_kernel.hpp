// minimizer_w_kernel.hpp -- (w, k)-minimizers in one pass, the run length equal to the window: C = w (4 ... 16), k <= 32.
//
// minimizer_fused_kernel.hpp (any C <= w) spends ~27 VALU instructions per k-mer on the two sweeps of the block-wise
// sliding minimum: every window looks up the prefix arg-min of its last block in LDS and compares.  With the block AS LONG
// AS the window the picks follow from RECORDS alone.  Window i of block b is the suffix [i, C) of block b and the prefix
// [0, i) of block b + 1; as i grows the suffix minimum can only grow and the prefix minimum only shrink, so there is one
// crossover i* (the first i whose prefix minimum is strictly smaller): windows before it pick the suffix's leftmost
// minimum, windows from it on the prefix's.  Hence, per block and without looking at single windows:
//   * windows [0, i*) of block b pick the SUFFIX RECORDS of block b (columns c with h[c] <= everything to the right of c
//     in the block) up to and including the first record at a column >= i* - 1;
//   * windows [i*, nv) of block b - 1 pick the PREFIX RECORDS of block b (h[c] < everything to the left) from the record
//     that is current at column i* - 2 ... up to column nv - 2;
// both sets lie in block b, ascending, the prefix picks first (the arg-min never moves left); their only common element
// can be the block's leftmost minimum, and a union of two bit masks emits it once.  A lane (= block) needs its own two
// record masks -- one add-with-carry per window and sweep -- and the crossover: the neighbour lane's prefix minima come
// over the DPP network (wave_rol:1), nothing is looked up.  12 VALU instructions per k-mer for the sweeps instead of 27.
//
// The hashes never reach HBM (first window from the byte tables, src/kmer.cpp:43-73,123-152; the others rolled,
// src/kmer.cpp:84-94,164-174); a lane's C hashes and prefix minima stay in registers (C is a template parameter), a
// picked column's hash goes from its register to the stash below (a predicated store per column).
//
// Placing the picks (the CSR output is dense): rounds of tiles, tile = (round * blocks + block) * waves + wave.  The waves
// of a block add up their tiles' counts through LDS; the wave that arrives last publishes the block's count for the round
// in a.status.  The look-back over the earlier block-rounds (256 per hop: one hop reaches the round before, whose inclusive
// counts are published) is done a round LATER by the wave of the block that finishes its next tile first -- every count
// it needs was published a tile's time ago, and the wave that does it is the one with time to spare -- and leaves every
// wave's offset in LDS.  Nobody waits for it: a tile's picks are parked in a small LDS stash and written -- coalesced --
// one round later.  (With the last wave of a round looking back at once the look-back's latency sat on the block's
// critical path in every round: 7.8 ms against 5.4 ms without any look-back, 20 M x 150 bp, w = 10.)
// (A look-back per TILE with the waves waiting cost more than the hashing: minimizer_fused_kernel.hpp.)  The bounded-wait
// rule of that kernel holds here too: a leader that waits 50 ms sets a.abort and the caller takes the round-3 path.
#pragma once

#include <hip/hip_runtime.h>

#include "block_rounds.hpp"
#include "minimizer_fused_kernel.hpp"

namespace ntamd {

// waves per block: 16 while a lane's 4 C + ~60 registers fit 128, else 12
constexpr uint32_t mzw_max_waves(int C) { return C <= 12 ? 16u : 12u; }
#ifndef MZW_ABL_NOFLUSH
#define MZW_ABL_NOFLUSH 0 // ablation: the parked picks are not written
#endif
constexpr uint32_t MZW_CTRL_DWORDS = BR_CTRL_DWORDS;

struct MinimizerWArgs {
  const uint8_t* seqs;
  const uint4* init_tab;        // [src_tabs][256] {f.lo, f.hi, r.lo, r.hi}; the kernel pads to 8 tables with zeros
  uint32_t* abort;              // zeroed by the host; set by a leader that waited too long
  uint32_t timeout_us;          // 0: the 50 ms of block_rounds.hpp (tests shorten it)
  unsigned long long* status;   // [(n_rounds + 1) * blocks] look-back words of the block-rounds, zeroed by the host
  uint64_t* out_hashes;
  uint32_t* out_pos;            // may be NULL
  uint64_t* out_offsets;        // [n_reads + 1]
  uint64_t* total;              // device: the number of picks (also when capacity is smaller)
  uint64_t capacity;
  uint64_t n_reads, total_bytes;
  uint32_t n_tiles, n_rounds;
  uint32_t src_tabs;
  uint32_t len, k, nwin, nwv;   // nwv = nwin - w + 1 window starts
  uint32_t rpr, inv_rpr, R;     // blocks per read, floor(65536 / rpr) + 1, reads per tile
  uint32_t extra;               // rpr * C - nwin
  uint32_t waves, bits_dwords, stash_cap, per_wave_dwords;
  uint64_t tab[16][2];
};

template <int C>
__global__ __launch_bounds__(64 * mzw_max_waves(C)) void minimizer_w_kernel(const MinimizerWArgs a)
{
  static_assert(C >= 2 && C <= 16, "one word of incoming bases per run; 16-bit record masks");
  constexpr uint32_t NW = 2, ntab = 8;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  const uint32_t k = a.k, rpr = a.rpr;
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t waves = a.waves;
  const uint64_t seqs_addr = (uint64_t)a.seqs;

  // LDS: first-window tables | pair table | block control | per wave { stash hashes, stash positions, bit stream, validity bits }
  uint4* itab = (uint4*)lds_dyn;
  uint4* ptab = itab + ntab * 256u;
  uint32_t* ctrl = (uint32_t*)(ptab + 16);
  BlockRounds rounds; // (block_rounds.hpp: where a tile's picks go, without a pass before and without a wave that waits)
  rounds.init(ctrl, lane, wave, waves, a.n_rounds, a.status, a.abort, a.total, a.out_offsets + a.n_reads, a.timeout_us);
  uint32_t* wave_base = ctrl + MZW_CTRL_DWORDS + wave * a.per_wave_dwords;
  uint64_t* stash_h = (uint64_t*)wave_base;                  // [stash_cap]
  uint16_t* stash_p = (uint16_t*)(stash_h + a.stash_cap);    // [stash_cap]
  uint32_t* bits = (uint32_t*)(stash_p + a.stash_cap);       // (stash_cap is even) [bits_dwords]
  uint16_t* vbits = (uint16_t*)(bits + a.bits_dwords);       // [bits_dwords + 8] validity bits of a slab with a non-base

  for (uint32_t i = tid; i < ntab * 256u; i += blockDim.x) itab[i] = i < a.src_tabs * 256u ? a.init_tab[i] : make_uint4(0, 0, 0, 0);
  if (tid < 16)
    ptab[tid] = make_uint4((uint32_t)a.tab[tid][0], (uint32_t)(a.tab[tid][0] >> 32), (uint32_t)a.tab[tid][1],
                           (uint32_t)(a.tab[tid][1] >> 32));
  if (tid < MZW_CTRL_DWORDS) ctrl[tid] = 0;
  __syncthreads(); // the only block-wide barrier

  auto lds_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  };

  const uint32_t lr_raw = (lane * a.inv_rpr) >> 16;
  const uint32_t q_raw = lane - lr_raw * rpr;
  uint32_t tbad = 0; // non-bases seen while the current tile's slab was packed

  // what is parked from the round before
  bool have_prev = false;
  uint32_t prev_rd = 0, prev_total = 0, prev_rel = 0;
  uint64_t prev_read = 0;
  bool prev_first = false; // this lane holds the first block of a read of the parked tile

  // the picks of the parked tile, and its reads' offsets, to their final places
  auto flush_prev = [&]() {
    lds_sync(); // (the stash was written lane by lane)
    const uint64_t base = rounds.offset_of(prev_rd);
    for (uint32_t i = lane; i < prev_total && !MZW_ABL_NOFLUSH; i += 64u) {
      const uint64_t o = base + i;
      if (o < a.capacity) {
        a.out_hashes[o] = stash_h[i];
        if (a.out_pos) a.out_pos[o] = stash_p[i];
      }
    }
    if (prev_first) a.out_offsets[prev_read] = base + prev_rel;
    have_prev = false;
  };

  // geometry of a tile's slab
  struct Slab {
    uint64_t rf, byte0;
    uint32_t reads_here, shift, slab_bytes, n_vec;
    bool edge;
  };
  auto slab_of = [&](uint32_t t) {
    Slab sl;
    sl.rf = (uint64_t)t * a.R;
    const uint64_t left = a.n_reads - sl.rf;
    sl.reads_here = left < a.R ? (uint32_t)left : a.R;
    const uint64_t start = sl.rf * a.len;
    uint64_t slab64 = (uint64_t)sl.reads_here * a.len + a.extra;
    if (start + slab64 > a.total_bytes) slab64 = a.total_bytes - start;
    sl.slab_bytes = (uint32_t)slab64;
    sl.shift = (uint32_t)((seqs_addr + start) & 15u);
    sl.byte0 = start - sl.shift; // (wraps below 0 by < 16 for an unaligned buffer)
    sl.n_vec = (sl.shift + sl.slab_bytes + 15u) >> 4;
    sl.edge = start < sl.shift || sl.byte0 + ((uint64_t)sl.n_vec << 4) > a.total_bytes;
    return sl;
  };
  // vector i of a slab: judged (bytes outside the caller's buffer exist only in the slabs flagged `edge`) and packed
  auto pack_vec = [&](const Slab& sl, uint32_t i, const uint4 v) {
    uint32_t b = 0;
    const uint32_t p = pack16(v, b);
    if (sl.edge) {
      const int32_t lo_cut = (int32_t)sl.shift - (int32_t)(i << 4);
      const int32_t hi_cut = (int32_t)(sl.shift + sl.slab_bytes) - (int32_t)(i << 4);
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
    tbad |= b;
    bits[i] = p;
  };
  // a slab that holds a non-base (rare): one validity bit per base, 16 per vector (1 = not a base)
  auto validity_vec = [&](uint32_t i, const uint4 v) {
    uint32_t i0, i1, i2, i3;
    (void)pack4v(v.x, i0);
    (void)pack4v(v.y, i1);
    (void)pack4v(v.z, i2);
    (void)pack4v(v.w, i3);
    vbits[i] = (uint16_t)(i0 | (i1 << 4) | (i2 << 8) | (i3 << 12));
  };
  const uint32_t t_step = gridDim.x * waves;
  uint32_t t = blockIdx.x * waves + wave; // tile of round rd: (rd * blocks + block) * waves + wave (< 2^31 + the grid's waves)
  // the first two vectors per lane of the next tile's slab are loaded a round ahead
  uint4 pf0 = make_uint4(0, 0, 0, 0), pf1 = pf0;
  Slab cur;
  if (t < a.n_tiles) {
    cur = slab_of(t);
    if (lane < cur.n_vec) pf0 = *(const uint4*)(a.seqs + cur.byte0 + ((uint64_t)lane << 4));
    if (lane + 64u < cur.n_vec) pf1 = *(const uint4*)(a.seqs + cur.byte0 + ((uint64_t)(lane + 64u) << 4));
  }

  for (uint32_t rd = 0; rd <= a.n_rounds; ++rd, t += t_step) { // (round n_rounds: nothing to hash, the round before is placed)
    const bool has = rd < a.n_rounds && t < a.n_tiles;
    uint32_t pick = 0, cnt = 0, incl = 0, tile_total = 0, q = 0, lr = 0;
    uint64_t rf = 0;
    bool live = false;
    uint64_t h[C];
#pragma unroll
    for (int i = 0; i < C; ++i) h[i] = 0;
    if (has) {
      // ---- stage the tile's slab as a 2-bit stream; the next one's loads go out ---------------------------------
      const Slab sl = cur;
      rf = sl.rf;
      lds_sync();
      tbad = 0;
      if (lane < sl.n_vec) pack_vec(sl, lane, pf0);
      if (lane + 64u < sl.n_vec) pack_vec(sl, lane + 64u, pf1);
      for (uint32_t i = 128u + lane; i < sl.n_vec; i += 64u) pack_vec(sl, i, *(const uint4*)(a.seqs + sl.byte0 + ((uint64_t)i << 4)));
      for (uint32_t i = sl.n_vec + lane; i < a.bits_dwords; i += 64u) bits[i] = 0; // the rolls of a last block read ahead
      // N-aware: a k-mer that holds a non-base is no candidate (NtHash does not emit it: src/kmer.cpp:228-264) -- its hash
      // becomes the largest value and it is never picked; a tile without a non-base (nearly all) pays one ballot
      const bool tile_dirty = __ballot(tbad != 0u) != 0ull;
      if (tile_dirty) {
        if (lane < sl.n_vec) validity_vec(lane, pf0);
        if (lane + 64u < sl.n_vec) validity_vec(lane + 64u, pf1);
        for (uint32_t i = 128u + lane; i < sl.n_vec; i += 64u) validity_vec(i, *(const uint4*)(a.seqs + sl.byte0 + ((uint64_t)i << 4)));
        for (uint32_t i = sl.n_vec + lane; i < a.bits_dwords + 8u; i += 64u) vbits[i] = 0;
      }
      if (t + t_step < a.n_tiles) {
        cur = slab_of(t + t_step);
        if (lane < cur.n_vec) pf0 = *(const uint4*)(a.seqs + cur.byte0 + ((uint64_t)lane << 4));
        if (lane + 64u < cur.n_vec) pf1 = *(const uint4*)(a.seqs + cur.byte0 + ((uint64_t)(lane + 64u) << 4));
      }
      lds_sync();

      // ---- forward: hash the run; prefix minima and prefix records ------------------------------------------------
      // (window slots past the read's last window hash whatever follows: no valid window ever covers them)
      live = lr_raw < sl.reads_here;
      lr = live ? lr_raw : 0u;
      q = live ? q_raw : 0u; // (idle lanes redo the tile's first run)
      const uint32_t b0 = sl.shift + lr * a.len + q * (uint32_t)C;
      const uint32_t d0 = b0 >> 4, sh0 = (b0 & 15u) << 1;
      uint32_t f_lo, f_hi, r_lo, r_hi;
      {
        uint32_t wv[NW];
        uint32_t lo = bits[d0];
#pragma unroll
        for (uint32_t i = 0; i < NW; ++i) {
          const uint32_t hi = bits[d0 + i + 1];
          wv[i] = funnel(hi, lo, sh0);
          lo = hi;
        }
        uint4 e[4 * NW];
#pragma unroll
        for (uint32_t jt = 0; jt < 4 * NW; ++jt) e[jt] = itab[jt * 256u + ((wv[jt >> 2] >> ((jt & 3) * 8)) & 0xFFu)];
        f_lo = e[0].x ^ e[1].x; f_hi = e[0].y ^ e[1].y; r_lo = e[0].z ^ e[1].z; r_hi = e[0].w ^ e[1].w;
#pragma unroll
        for (uint32_t jt = 2; jt < 4 * NW; jt += 2) {
          f_lo = __builtin_amdgcn_bitop3_b32(f_lo, e[jt].x, e[jt + 1].x, 0x96);
          f_hi = __builtin_amdgcn_bitop3_b32(f_hi, e[jt].y, e[jt + 1].y, 0x96);
          r_lo = __builtin_amdgcn_bitop3_b32(r_lo, e[jt].z, e[jt + 1].z, 0x96);
          r_hi = __builtin_amdgcn_bitop3_b32(r_hi, e[jt].w, e[jt + 1].w, 0x96);
        }
      }
      h[0] = canon_pair(f_lo, f_hi, r_lo, r_hi);
      {
        const uint32_t bi = b0 + k;
        const uint32_t di = bi >> 4, shi = (bi & 15u) << 1;
        const uint32_t w_in = funnel(bits[di + 1], bits[di], shi);
        const uint32_t w_out = funnel(bits[d0 + 1], bits[d0], sh0);
        const uint32_t u = ((w_in & 0x33333333u) << 2) | (w_out & 0x33333333u);
        const uint32_t v = (w_in & 0xCCCCCCCCu) | ((w_out >> 2) & 0x33333333u);
        // (the table terms do not depend on the hash state: eight lookups ahead of the dependent chain)
#pragma unroll
        for (uint32_t j0 = 1; j0 < (uint32_t)C; j0 += 8u) {
          uint4 terms[8];
#pragma unroll
          for (uint32_t jj = 0; jj < 8u; ++jj) {
            const uint32_t i = j0 - 1u + jj;
            if (i + 1u < (uint32_t)C) {
              const uint32_t src = (i & 1u) ? v : u;
              const uint32_t off = ((src >> ((i >> 1) * 4u)) & 0xFu) << 4;
              terms[jj] = *(const uint4*)((const char*)ptab + off);
            }
          }
#pragma unroll
          for (uint32_t jj = 0; jj < 8u; ++jj) {
            const uint32_t j = j0 + jj;
            if (j < (uint32_t)C) {
              roll_step(f_lo, f_hi, r_lo, r_hi, terms[jj]);
              h[j] = canon_pair(f_lo, f_hi, r_lo, r_hi);
            }
          }
        }
      }
      uint32_t inval = 0; // bit j: the k-mer of column j holds a non-base
      if (tile_dirty) {
        lds_sync();
        inval = windows_with_non_base((const uint32_t*)vbits, b0, k) & ((1u << C) - 1u);
#pragma unroll
        for (int j = 0; j < C; ++j) h[j] = ((inval >> j) & 1u) ? ~0ull : h[j];
      }
      uint64_t pre[C];
      uint32_t prbits = 1; // bit (C - 1 - j): column j holds a hash smaller than every one to its left (column 0: always)
      pre[0] = h[0];
#pragma unroll
      for (int j = 1; j < C; ++j) {
        const bool lt = h[j] < pre[j - 1]; // strict: the leftmost of equal hashes stays
        pre[j] = lt ? h[j] : pre[j - 1];
        prbits = prbits + prbits + (lt ? 1u : 0u);
      }

      // ---- backward: suffix minima, suffix records, and where the next block's prefix takes over ------------------
      uint64_t sh = h[C - 1];
      uint32_t srbits = 1; // bit i: column i holds a hash <= every one to its right (column C - 1: always)
      uint32_t xbits = 0;  // bit i (>= 1): the prefix [0, i) of the next block is strictly smaller than the suffix [i, C)
      {
        const uint32_t nlo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)pre[C - 2], 0x134, 0xf, 0xf, false);
        const uint32_t nhi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(pre[C - 2] >> 32), 0x134, 0xf, 0xf, false);
        xbits = ((((uint64_t)nhi << 32) | nlo) < sh) ? 1u : 0u;
      }
#pragma unroll
      for (int i = C - 2; i >= 0; --i) {
        const bool le = !(sh < h[i]); // the leftmost of equal hashes wins
        sh = le ? h[i] : sh;
        srbits = srbits + srbits + (le ? 1u : 0u);
        bool px = false;
        if (i >= 1) { // lane l + 1's prefix minimum through column i - 1 (wave_rol:1: lane l reads lane l + 1)
          const uint32_t nlo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)pre[i - 1], 0x134, 0xf, 0xf, false);
          const uint32_t nhi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(pre[i - 1] >> 32), 0x134, 0xf, 0xf, false);
          px = ((((uint64_t)nhi << 32) | nlo) < sh);
        }
        xbits = xbits + xbits + (px ? 1u : 0u);
      }

      // ---- the block's picks: two ranges of records ---------------------------------------------------------------
      const uint32_t PR = __builtin_bitreverse32(prbits) >> (32u - (uint32_t)C); // bit j: prefix record at column j
      const uint32_t SR = srbits;
      const uint32_t first_w = q * (uint32_t)C;
      const uint32_t nv = live && a.nwv > first_w ? (a.nwv - first_w < (uint32_t)C ? a.nwv - first_w : (uint32_t)C) : 0u;
      const uint32_t xs = xbits & ((1u << nv) - 1u) & ~1u;
      const uint32_t istar = xs ? (uint32_t)__builtin_ctz(xs) : nv;
      const uint32_t p_istar = (uint32_t)__shfl_up((int)istar, 1, 64), p_nv = (uint32_t)__shfl_up((int)nv, 1, 64);
      uint32_t mask_a = 0, mask_b = 0;
      if (q != 0u && p_istar < p_nv) { // windows [p_istar, p_nv) of the block before take their minimum from this block's prefix
        const uint32_t jmin = p_istar - 1u, jmax = p_nv - 2u;
        const uint32_t low = PR & ((2u << jmin) - 1u);
        const uint32_t rec = 31u - (uint32_t)__builtin_clz(low);
        mask_a = PR & ((2u << jmax) - 1u) & ~((1u << rec) - 1u);
      }
      if (nv != 0u) { // windows [0, istar) of this block: the suffix records up to the first one at a column >= istar - 1
        const uint32_t cstop = (uint32_t)__builtin_ctz(SR & ~((1u << (istar - 1u)) - 1u));
        mask_b = SR & ((2u << cstop) - 1u);
      }
      pick = (mask_a | mask_b) & ~inval; // (a window whose every k-mer holds a non-base picks nothing)
      cnt = (uint32_t)__builtin_popcount(pick);
      incl = wave_incl_add32(cnt);
      tile_total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }

    // ---- arrive: the block's counts of this round; the round before is placed by the wave that gets here first -----
    rounds.arrive_round(rd, tile_total);
    if (have_prev) flush_prev();
    if (has) {
      // ---- park the tile's picks (hash, position) in the stash, straight from the registers; a tile with more than
      // fit (reads of one repeated base: every window a new pick) writes them itself once its offset is there ----------
      const bool park = tile_total <= a.stash_cap;
      const bool first_block = live && q == 0u;
      uint32_t o = incl - cnt;
      if (park) {
        lds_sync();
#pragma unroll
        for (uint32_t c = 0; c < (uint32_t)C; ++c) {
          if ((pick >> c) & 1u) {
            stash_h[o] = h[c];
            stash_p[o] = (uint16_t)(q * (uint32_t)C + c);
            ++o;
          }
        }
        have_prev = true;
        prev_rd = rd;
        prev_total = tile_total;
        prev_rel = incl - cnt;
        prev_read = rf + lr;
        prev_first = first_block;
      } else {
        rounds.try_lead(rd);
        const uint64_t base = rounds.offset_of(rd);
#pragma unroll
        for (uint32_t c = 0; c < (uint32_t)C; ++c) {
          if (((pick >> c) & 1u) != 0u) {
            if (base + o < a.capacity) {
              a.out_hashes[base + o] = h[c];
              if (a.out_pos) a.out_pos[base + o] = q * (uint32_t)C + c;
            }
            ++o;
          }
        }
        if (first_block) a.out_offsets[rf + lr] = base + (incl - cnt);
      }
    }
  }
  if (have_prev) flush_prev();
}

} // namespace ntamd
