// kmer_reads_kernel.hpp -- run-split k-mer hashing of VARIABLE-LENGTH short reads, tiles of WHOLE reads (round 2).
//
// Replaces, for a batch of reads of any lengths, the reference's per-read loop
//     NtHash h(seq, len, m, k); while (h.roll()) use(h.hashes());      (src/kmer.cpp:196-264)
// kmer_ragged_kernel (round 1) cuts the run sequence of the batch into tiles of exactly 64 runs, wherever in a read
// that falls: every tile needs a search structure (listed reads, first read of the tile, staged vectors scattered
// over up to 64 reads) and a count pass with the same machinery, and the hash pass carries N-awareness for every
// window although one read in hundreds has an N.  Here
//   * a wave's tile is R CONSECUTIVE WHOLE reads (R <= 64): its bytes are ONE contiguous slab (the reads of an
//     offsets batch lie back to back; the sequence lines of a FASTQ chunk have the header / quality lines between
//     them, staged and never looked at), its k-mers ONE contiguous piece of the stream;
//   * a MARK pass stages the slabs once and looks only for non-bases: a read without one emits every window
//     (count = len - k + 1), a read WITH one is put on a list and left to kmer_dirty_reads_kernel (one lane per
//     read, the reference's skipping rule, exact counts first);
//   * the HASH pass therefore rolls clean reads only: no validity stream, no predicated compaction, slots from the
//     geometry.  A lane is a run of C windows of one read, 64 consecutive runs per pass, the pass's hashes go
//     through a wave-private LDS tile to whole aligned pieces of the stream as in kmer_runs_gen_kernel.hpp.  The
//     last run of a read starts C windows before the read's last window (it overlaps the run before it and writes
//     the same values to the same slots), so every run has exactly C windows and nothing in the roll is predicated;
//     only a read with fewer than C windows has a short run (those passes take the predicated code).  The
//     k-mers of a listed read inside a pass are a hole in the tile (written as is, the dirty-read kernel runs
//     afterwards on the same stream and fills it).
#pragma once

#include <hip/hip_runtime.h>

#include "kmer_runs_gen_kernel.hpp"

namespace ntamd {

// RD_MODE_SLOTS (round 3, the NTHIP_OUT_READ_SLOTS contract): ONE pass over the bases.  Read r's k-mers go to the slot its
// LENGTH implies -- slot_off[r] = sum over the reads before it of max(len - k + 1, 0), a scan of the spans alone -- so
// nothing has to be counted before anything is written: no mark pass, no compaction.  The pass hashes every read as if
// clean and looks for non-bases while it packs (free); in the rare tile that has one the wave finds the reads concerned
// (validity bits of the slab, L2-hot), lists them and leaves their slots to kmer_dirty_reads_kernel, which writes the
// windows the reference emits at the front of the slot, the exact count into counts[r] and zeros behind.
enum : int { RD_MODE_MARK = 1, RD_MODE_HASH = 2, RD_MODE_SLOTS = 3 };
constexpr uint32_t RD_ALIGN_U64 = 16; // the output tile is aligned to a 128-byte line of the stream

struct KmerReadsArgs {
  const uint8_t* seqs;
  const uint64_t* starts;    // read r = bytes [starts[r], ends[r]) of seqs, starts non-decreasing, no overlap
  const uint64_t* ends;
  uint64_t n_reads, n_tiles;
  uint32_t R;                // reads per tile
  // MARK
  uint64_t* cnt;             // [n_reads] windows of a clean read; 0 for a listed one (kmer_dirty_reads_kernel counts it)
  uint8_t* flags;            // [n_reads] 1 = listed
  uint64_t* dirty_list;      // reads with a non-base and at least k bytes
  unsigned long long* dirty_count;
  uint64_t* tile_sum;        // [n_tiles] k-mers of the tile's clean reads (kmer_dirty_reads_kernel adds the listed ones)
  // HASH
  const uint64_t* tile_off;  // exclusive scan of tile_sum: the tile's first k-mer in the stream
  uint64_t* hashes;
  uint32_t* pos;
  const uint4* init_tab;
  uint32_t k, m, C, ntab;
  uint32_t waves, bits_dwords, tile_u64, ptile_dwords, rmap_dwords;
  uint32_t groups;           // tile_range(): groups of blocks sharing a range of tiles (0: one range per block)
  uint32_t value_sel;        // 0: canonical hash (+ mixes), 1: forward, 2: reverse strand (m == 1)
  uint64_t tab[16][2];
  uint64_t mult[KF_MAX_RUNTIME_M];
};

// POS: window positions wanted.  value_sel 1 / 2 (one strand's hash instead of the canonical one) needs no code in the
// roll: with the other strand's table terms zeroed its state stays 0 and forward + reverse IS the wanted strand.
// (the MARK pass waits for memory and nothing else: 8 waves per SIMD -- two blocks of 16 waves per CU -- instead of the
//  4 the register count of the common body would give it)
// (SLOTS with four window words needs a few registers more than the 128 that 16 waves leave: at most 12 waves there)
constexpr int rd_max_threads(int mode, int nw) { return mode == RD_MODE_SLOTS && nw >= 4 ? 768 : KR_MAX_THREADS; }
template <int MODE, int NW, bool POS = false>
__global__ __launch_bounds__(rd_max_threads(MODE, NW)) __attribute__((amdgpu_waves_per_eu(MODE == RD_MODE_MARK ? 8 : 1)))
void kmer_reads_kernel(const KmerReadsArgs a)
{
  extern __shared__ __attribute__((aligned(256))) uint32_t lds_dyn[];
  const uint32_t k = a.k, m = a.m, C = a.C;
  const uint32_t inv_m = 0xFFFFFFFFu / m + 1u;
  const uint64_t kmul = (uint64_t)k * MULTISEED;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // LDS: [HASH: first-window tables | pair table] | per wave {hash tile, pos tile, bit stream, read table 4 x 64}
  constexpr bool HASHING = MODE == RD_MODE_HASH || MODE == RD_MODE_SLOTS;
  uint4* itab = (uint4*)lds_dyn;
  uint4* ptab = itab + a.ntab * 256u;
  const uint32_t per_wave = a.tile_u64 * 2u + a.ptile_dwords + a.bits_dwords + 256u + a.rmap_dwords;
  uint32_t* wave_base = (HASHING ? (uint32_t*)(ptab + 16) : lds_dyn) + wave * per_wave;
  uint64_t* tile = (uint64_t*)wave_base;
  uint32_t* ptile = wave_base + a.tile_u64 * 2u;
  uint32_t* bits = ptile + a.ptile_dwords; // HASH: 2-bit codes, 16 per dword; MARK: 1 bit per byte, set = not a base
  uint32_t* rt_beg = bits + a.bits_dwords; // first run of read j in the tile (0xFFFFFFFF past the tile's reads)
  uint32_t* rt_sb = rt_beg + 64;           // stream index of the read's first base
  uint32_t* rt_nw = rt_sb + 64;            // windows (0: shorter than k, or listed)
  uint32_t* rt_out = rt_nw + 64;           // first k-mer of the read, relative to the tile's first
  uint8_t* rmap = (uint8_t*)(rt_out + 64); // run of the tile -> its read
  if (HASHING) {
    const uint32_t keep_f = a.value_sel == 2u ? 0u : ~0u, keep_r = a.value_sel == 1u ? 0u : ~0u;
    for (uint32_t i = tid; i < a.ntab * 256u; i += blockDim.x) {
      const uint4 e = a.init_tab[i];
      itab[i] = make_uint4(e.x & keep_f, e.y & keep_f, e.z & keep_r, e.w & keep_r);
    }
    if (tid < 16)
      ptab[tid] = make_uint4((uint32_t)a.tab[tid][0] & keep_f, (uint32_t)(a.tab[tid][0] >> 32) & keep_f,
                             (uint32_t)a.tab[tid][1] & keep_r, (uint32_t)(a.tab[tid][1] >> 32) & keep_r);
  }
  __syncthreads();

  const uint32_t ptab_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint4*)ptab;
  auto lds_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  };
  // inclusive prefix sum over the 64 lanes on the DPP network (no LDS round trips, unlike __shfl_up = ds_bpermute):
  // row_shr 1, 2, 4, 8 scan each row of 16 lanes, row_bcast15 / row_bcast31 carry the row totals across
  auto wave_incl_scan32 = [&](uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
  };
  auto bcast64 = [&](uint64_t v, uint32_t src) -> uint64_t {
    return ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(v >> 32), (int)src, 64) << 32) |
           (uint32_t)__shfl((int)(uint32_t)v, (int)src, 64);
  };

  const TileRange tr = tile_range(a.n_tiles, a.waves, wave, a.groups);
  const uint64_t t_end = tr.end, t_step = tr.step;

  // Software pipeline of the HASH pass (a wave is alone with its latencies): a tile's bytes sit behind two dependent
  // loads -- the reads' spans, then the slab they delimit.  The spans / flags / counts of tile t + 2 and the first
  // RD_PF_ROUNDS x 64 vectors of tile t + 1's slab are loaded while tile t is hashed.
#ifndef RD_PF_ROUNDS_N
#define RD_PF_ROUNDS_N 3 // 3 KiB of slab in registers (in-process A/B, 10 M reads: 100-150 bp +5 % against none, 5 rounds
                         // +7 %; nearly all reads 150 bp: -0.7 % / -4.7 %)
#endif
  // (the MARK pass does next to nothing per tile and has the registers: its whole next slab, up to 5 KiB, travels ahead)
#ifndef RD_PF_MARK_ROUNDS_N
#define RD_PF_MARK_ROUNDS_N 5
#endif
  constexpr uint32_t RD_PF_ROUNDS = MODE == RD_MODE_MARK ? RD_PF_MARK_ROUNDS_N : RD_PF_ROUNDS_N;
  struct Meta {
    uint64_t s, e;
    uint32_t cnt, listed;
  };
  auto load_meta = [&](uint64_t tt) -> Meta {
    const uint64_t rr0 = tt * a.R;
    const uint32_t nn = a.n_reads - rr0 < a.R ? (uint32_t)(a.n_reads - rr0) : a.R;
    const uint64_t rr = rr0 + (lane < nn ? lane : 0u);
    Meta mm;
    mm.s = a.starts[rr];
    mm.e = a.ends[rr];
    mm.cnt = MODE == RD_MODE_HASH ? (uint32_t)a.cnt[rr] : 0u;       // (SLOTS: from the length, below)
    mm.listed = MODE == RD_MODE_HASH ? (uint32_t)a.flags[rr] : 0u;  // (SLOTS: found while packing)
    return mm;
  };
  struct Geom {
    uint64_t slab0;
    const uint8_t* vbase;
    uint32_t shift, n_vec;
  };
  auto geom_of = [&](const Meta& mm, uint64_t tt) -> Geom {
    const uint64_t rr0 = tt * a.R;
    const uint32_t nn = a.n_reads - rr0 < a.R ? (uint32_t)(a.n_reads - rr0) : a.R;
    Geom g;
    g.slab0 = bcast64(mm.s, 0);
    const uint64_t slab_end = bcast64(mm.e, nn - 1u);
    g.shift = (uint32_t)(((uintptr_t)a.seqs + g.slab0) & 15u);
    g.vbase = a.seqs + g.slab0 - g.shift; // 16-byte aligned: a vector never crosses a page
    g.n_vec = slab_end > g.slab0 ? (uint32_t)((g.shift + (slab_end - g.slab0) + 15u) >> 4) : 0u;
    return g;
  };
  uint4 pv[RD_PF_ROUNDS ? RD_PF_ROUNDS : 1];
  auto issue_slab = [&](const Geom& g) {
#pragma unroll
    for (uint32_t rd = 0; rd < RD_PF_ROUNDS; ++rd) {
      const uint32_t i = rd * 64u + lane;
      pv[rd] = *(const uint4*)(g.vbase + ((uint64_t)(i < g.n_vec ? i : 0u) << 4)); // lanes past the slab re-read its start
    }
  };
  Meta m_cur, m_nxt;
  Geom g_cur;
  {
    const uint64_t t0 = tr.first;
    if (t0 >= t_end) return;
    m_cur = load_meta(t0);
    g_cur = geom_of(m_cur, t0);
    issue_slab(g_cur);
    m_nxt = t0 + t_step < t_end ? load_meta(t0 + t_step) : m_cur;
  }

  for (uint64_t t = tr.first; t < t_end; t += t_step) {
    const uint64_t r0 = t * a.R;
    const uint32_t nr = a.n_reads - r0 < a.R ? (uint32_t)(a.n_reads - r0) : a.R;
    // ---- this tile's reads: one per lane ----
    const bool has = lane < nr;
    const uint64_t rj = r0 + (has ? lane : 0u);
    const uint64_t s_j = m_cur.s, e_j = m_cur.e;
    const uint64_t len_j = has && e_j > s_j ? e_j - s_j : 0;
    bool listed = false;
    uint64_t ro_j = 0;
    uint64_t ro_0 = 0;
    const uint32_t nwin_raw = len_j >= k ? (uint32_t)(len_j - k + 1u) : 0u;
    if (HASHING) {
      listed = m_cur.listed != 0;
      // the read's first k-mer = the tile's + the k-mers of the reads before it in the tile (a tile has < 2^32);
      // SLOTS: + the WINDOWS of the reads before it -- a read's place does not depend on what its neighbours hold
      const uint32_t cnt_j = MODE == RD_MODE_SLOTS ? nwin_raw : has ? m_cur.cnt : 0u;
      ro_0 = a.tile_off[t];
      ro_j = ro_0 + (wave_incl_scan32(cnt_j) - cnt_j);
    }
    const uint64_t slab0 = g_cur.slab0;
    const uint32_t shift = g_cur.shift;
    const uint8_t* vbase = g_cur.vbase;
    const uint32_t n_vec = g_cur.n_vec;
    const uint32_t sb_j = shift + (uint32_t)(s_j - slab0);
    const bool have_next = t + t_step < t_end;

    if (MODE == RD_MODE_MARK) {
      // ---- stage: one validity bit per byte (the first vectors are in registers already) ----
      auto mark_vec = [&](const uint32_t i, const uint4 x) {
        uint32_t i0, i1, i2, i3;
        (void)pack4v(x.x, i0);
        (void)pack4v(x.y, i1);
        (void)pack4v(x.z, i2);
        (void)pack4v(x.w, i3);
        ((uint16_t*)bits)[i] = (uint16_t)(i0 | (i1 << 4) | (i2 << 8) | (i3 << 12));
      };
      // (round 5) a slab without any non-base -- all but a few per cent of the tiles of real reads, none of a FASTQ chunk's, whose
      // header and quality lines lie between the spans -- needs no bits: the validity masks of the vectors in registers are
      // OR-ed over the wave first, and only a slab that has one is staged and looked at read by read
      uint16_t vm[RD_PF_ROUNDS ? RD_PF_ROUNDS : 1];
      uint32_t vany = 0;
#pragma unroll
      for (uint32_t rd = 0; rd < RD_PF_ROUNDS; ++rd) {
        const uint32_t i = rd * 64u + lane;
        uint32_t i0, i1, i2, i3;
        (void)pack4v(pv[rd].x, i0);
        (void)pack4v(pv[rd].y, i1);
        (void)pack4v(pv[rd].z, i2);
        (void)pack4v(pv[rd].w, i3);
        vm[rd] = (uint16_t)(i0 | (i1 << 4) | (i2 << 8) | (i3 << 12));
        if (i < n_vec) vany |= vm[rd];
      }
      const bool clean_slab = n_vec <= RD_PF_ROUNDS * 64u && __ballot(vany != 0u) == 0ull;
      if (!clean_slab) {
#pragma unroll
        for (uint32_t rd = 0; rd < RD_PF_ROUNDS; ++rd) {
          const uint32_t i = rd * 64u + lane;
          if (i < n_vec) ((uint16_t*)bits)[i] = vm[rd];
        }
        for (uint32_t i = RD_PF_ROUNDS * 64u + lane; i < n_vec; i += 64u) mark_vec(i, *(const uint4*)(vbase + ((uint64_t)i << 4)));
        if (lane < 4u) ((uint16_t*)bits)[n_vec + lane] = 0;
        lds_sync();
      }
      // ---- the next tile's slab and the spans of the tile after it: in flight during the rest of this tile ----
      if (have_next) {
        m_cur = m_nxt;
        g_cur = geom_of(m_cur, t + t_step);
        issue_slab(g_cur);
        if (t + 2u * t_step < t_end) m_nxt = load_meta(t + 2u * t_step);
      }
      // ---- any non-base inside [sb_j, sb_j + len_j) ? ----
      uint32_t any = 0;
      if (len_j && !clean_slab) {
        const uint32_t b_end = sb_j + (uint32_t)len_j; // one past the last byte
        const uint32_t w_lo = sb_j >> 5, w_hi = (b_end - 1u) >> 5;
        for (uint32_t w = w_lo; w <= w_hi; ++w) {
          uint32_t word = bits[w];
          if (w == w_lo) word &= ~0u << (sb_j & 31u);
          if (w == w_hi && (b_end & 31u)) word &= ~0u >> (32u - (b_end & 31u));
          any |= word;
        }
      }
      const bool dirty = has && any != 0 && nwin_raw != 0;
      if (has) {
        a.cnt[rj] = dirty ? 0 : nwin_raw;
        a.flags[rj] = dirty ? 1 : 0;
        if (dirty) a.dirty_list[atomicAdd(a.dirty_count, 1ull)] = rj;
      }
      const uint32_t tsum = wave_incl_scan32(dirty || !has ? 0u : nwin_raw);
      if (lane == 63u) a.tile_sum[t] = tsum;
      if (!clean_slab) lds_sync(); // the bit stream is free again
      continue;
    }

    // ---- HASH: stage the slab as 2-bit codes (the first vectors are in registers already) ----
    uint32_t slab_bad = 0; // SLOTS: a byte of the slab (reads and whatever lies between them) is not a base
#pragma unroll
    for (uint32_t rd = 0; rd < RD_PF_ROUNDS; ++rd) {
      const uint32_t i = rd * 64u + lane;
      uint32_t bad = 0;
      if (i < n_vec) bits[i] = pack16(pv[rd], bad);
      slab_bad |= i < n_vec ? bad : 0u;
    }
    for (uint32_t i = RD_PF_ROUNDS * 64u + lane; i < n_vec; i += 64u) {
      const uint4 x = *(const uint4*)(vbase + ((uint64_t)i << 4));
      uint32_t bad = 0;
      bits[i] = pack16(x, bad);
      slab_bad |= bad;
    }
    if (lane < (uint32_t)NW + 3u) bits[n_vec + lane] = 0;
    if constexpr (MODE == RD_MODE_SLOTS) {
      // which reads hold a non-base?  Almost always none does -- but the bytes BETWEEN the reads of a FASTQ chunk
      // (headers, quality lines) are never bases, so the question is asked per read, from validity bits built in the
      // (still unused) hash tile: one bit per byte of the slab, as the MARK pass does
      if (__ballot(slab_bad != 0u) != 0ull) {
        uint16_t* vb = (uint16_t*)tile;
        for (uint32_t i = lane; i < n_vec; i += 64u) {
          const uint4 x = *(const uint4*)(vbase + ((uint64_t)i << 4));
          uint32_t i0, i1, i2, i3;
          (void)pack4v(x.x, i0);
          (void)pack4v(x.y, i1);
          (void)pack4v(x.z, i2);
          (void)pack4v(x.w, i3);
          vb[i] = (uint16_t)(i0 | (i1 << 4) | (i2 << 8) | (i3 << 12));
        }
        if (lane < 4u) vb[n_vec + lane] = 0;
        lds_sync();
        uint32_t any = 0;
        if (len_j) {
          const uint32_t* vw = (const uint32_t*)tile;
          const uint32_t b_end = sb_j + (uint32_t)len_j; // one past the last byte
          const uint32_t w_lo = sb_j >> 5, w_hi = (b_end - 1u) >> 5;
          for (uint32_t w = w_lo; w <= w_hi; ++w) {
            uint32_t word = vw[w];
            if (w == w_lo) word &= ~0u << (sb_j & 31u);
            if (w == w_hi && (b_end & 31u)) word &= ~0u >> (32u - (b_end & 31u));
            any |= word;
          }
        }
        listed = has && any != 0 && nwin_raw != 0;
        if (listed) a.dirty_list[atomicAdd(a.dirty_count, 1ull)] = rj;
        lds_sync(); // the tile area is the hash tile again
      }
      if (has && !listed) a.cnt[rj] = nwin_raw; // (a listed read's count comes from kmer_dirty_reads_kernel)
    }
    // ---- the next tile's slab and the spans of the tile after it: in flight during this tile's passes ----
    Meta m_n2 = m_nxt;
    Geom g_nxt = g_cur;
    if (have_next) {
      g_nxt = geom_of(m_nxt, t + t_step);
      issue_slab(g_nxt);
      if (t + 2u * t_step < t_end) m_n2 = load_meta(t + 2u * t_step);
    }
    // ---- read table ----
    const uint32_t nwin_j = listed ? 0u : nwin_raw;
    const uint32_t rc_j = (nwin_j + C - 1u) / C;
    const uint32_t rend = wave_incl_scan32(rc_j);
    rt_beg[lane] = has ? rend - rc_j : 0xFFFFFFFFu;
    rt_sb[lane] = sb_j;
    rt_nw[lane] = nwin_j;
    rt_out[lane] = (uint32_t)(ro_j - ro_0);
    const uint32_t total_runs = (uint32_t)__shfl((int)rend, 63, 64);
    for (uint32_t i = 0; i < rc_j; ++i) rmap[rend - rc_j + i] = (uint8_t)lane;
    const bool short_reads = __ballot(nwin_j != 0u && nwin_j < C) != 0; // a read with fewer than C windows in the tile
    lds_sync();

    // ---- passes of up to 64 consecutive runs ----
    for (uint32_t g0 = 0; g0 < total_runs;) {
      const uint32_t g = g0 + lane;
      const bool live = g < total_runs;
      const uint32_t j = rmap[live ? g : g0];
      const uint32_t q = (live ? g : g0) - rt_beg[j];
      const uint32_t nwin_r = rt_nw[j];
      // the read's last run is moved back so that it has C windows too (nwin_r >= C; else one short run)
      const uint32_t w_first = nwin_r >= C ? (q * C + C <= nwin_r ? q * C : nwin_r - C) : 0u;
      const uint32_t c_run = live ? (nwin_r < C ? nwin_r : C) : 0u;
      const uint32_t b0 = rt_sb[j] + w_first;
      const uint32_t oslot = rt_out[j] + w_first; // place in the tile's piece of the stream
      const uint32_t pass_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)oslot);
      const uint64_t o0 = ro_0 + pass_base; // the pass's first k-mer in the stream
      const uint32_t tpar = m == 1u ? (uint32_t)(o0 & (RD_ALIGN_U64 - 1u)) : 0u;
      const uint32_t slot = oslot - pass_base;
      // runs that fit the tile (a listed read inside the pass leaves a hole and may push later runs out)
      const bool fits = live && tpar + slot + c_run <= a.tile_u64;
      const uint32_t nl = (uint32_t)__builtin_popcountll(__ballot(fits));
      const bool act = lane < nl;
      const uint32_t span = (uint32_t)__shfl((int)(slot + c_run), (int)(nl - 1u), 64);

      auto hash_run = [&](auto pred_tag) {
        constexpr bool PRED = decltype(pred_tag)::value; // a read shorter than C windows somewhere in the tile
        const uint32_t d0 = b0 >> 4, sh0 = (b0 & 15u) << 1;
        uint32_t f_lo = 0, f_hi = 0, r_lo = 0, r_hi = 0;
        if constexpr (NW == 0) {
          any_k_first_window(bits, itab, b0, k, f_lo, f_hi, r_lo, r_hi);
        } else {
          uint32_t w[NW];
          uint32_t lo = bits[d0];
#pragma unroll
          for (int i = 0; i < NW; ++i) {
            const uint32_t hi = bits[d0 + i + 1];
            w[i] = funnel(hi, lo, sh0);
            lo = hi;
          }
          uint4 e[4 * NW];
#pragma unroll
          for (int jt = 0; jt < 4 * NW; ++jt) e[jt] = itab[(uint32_t)jt * 256u + ((w[jt >> 2] >> ((jt & 3) * 8)) & 0xFFu)];
          f_lo = e[0].x ^ e[1].x; f_hi = e[0].y ^ e[1].y; r_lo = e[0].z ^ e[1].z; r_hi = e[0].w ^ e[1].w;
#pragma unroll
          for (int jt = 2; jt < 4 * NW; jt += 2) { // a ^ b ^ c is one v_bitop3_b32
            f_lo = __builtin_amdgcn_bitop3_b32(f_lo, e[jt].x, e[jt + 1].x, 0x96);
            f_hi = __builtin_amdgcn_bitop3_b32(f_hi, e[jt].y, e[jt + 1].y, 0x96);
            r_lo = __builtin_amdgcn_bitop3_b32(r_lo, e[jt].z, e[jt + 1].z, 0x96);
            r_hi = __builtin_amdgcn_bitop3_b32(r_hi, e[jt].w, e[jt + 1].w, 0x96);
          }
        }
        uint64_t* const mine = tile + tpar + slot;
        uint32_t* const pmine = ptile + slot;
        auto emit = [&](uint32_t jw) {
          if (!PRED || jw < c_run) {
            mine[jw] = canon_pair(f_lo, f_hi, r_lo, r_hi);
            if (POS) pmine[jw] = w_first + jw;
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
          auto lookup = [&](uint32_t i) -> uint4 { // nibble i of the step stream -> its 16-byte entry (the table is 256-aligned)
            const uint32_t src = (i & 1u) ? v : u;
            const uint32_t sh = (i >> 1) * 4u;
            const uint32_t ad = ((sh == 0u ? src << 4 : src >> (sh - 4u)) & 0xF0u) | ptab_addr;
            const nt_v4u e = *(__attribute__((address_space(3))) const nt_v4u*)(uintptr_t)ad;
            return make_uint4(e.x, e.y, e.z, e.w);
          };
          auto roll = [&](const uint4 term) {
            roll_step<!(MODE == RD_MODE_HASH && NW >= 4)>(f_lo, f_hi, r_lo, r_hi, term);
          };
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
      };
#ifndef RD_ABL
#define RD_ABL 0 // measurement builds (WRONG results): 1: nothing is stored, 2: nothing is hashed, 3: neither
#endif
      if (act && !(RD_ABL & 2)) {
        if (short_reads) hash_run(std::true_type{});
        else hash_run(std::false_type{});
      }
      lds_sync();
      // ---- copy-out: the tile was built shifted by tpar = o0 mod 16: every 16-byte piece is aligned, a wave
      // instruction covers whole 128-byte lines of the stream ----
      if (RD_ABL & 1) {
      } else if (m == 1) {
        const uint32_t sp = tpar + span;
        const uint32_t pieces = (sp + 1u) >> 1;
        uint64_t* const base = a.hashes + (o0 - tpar);
        const uint32_t pf = (tpar + 1u) >> 1, pl = sp >> 1; // whole 16-byte pieces: [pf, pl)
        for (uint32_t pi = lane; pi < pl; pi += 64u) {
          if (pi >= pf) {
            const uint4 dv = *(const uint4*)(tile + 2u * pi);
#if defined(RD_ST_PLAIN)
            *(uint4*)(base + 2u * pi) = dv;
#elif defined(RD_ST_WT)
            stream_store16(base + 2u * pi, dv);
#else
            __builtin_nontemporal_store(*(const nt_v4u*)&dv, (nt_v4u*)(base + 2u * pi));
#endif
          }
        }
        if (lane == 0u && (tpar & 1u)) base[tpar] = tile[tpar];                   // head
        if (lane == 1u && (sp & 1u) && sp - 1u >= pf * 2u) base[sp - 1u] = tile[sp - 1u]; // tail
        (void)pieces;
      } else {
        (void)multi_hash_copy_out<false>(tile, a.hashes, o0, span, m, inv_m, kmul, lane);
      }
      if (POS)
        for (uint32_t e = lane; e < span; e += 64u) a.pos[o0 + e] = ptile[e];
      lds_sync(); // the tile is free again
      g0 += nl;
    }
    lds_sync(); // bit stream and read table are free again
    m_cur = m_nxt;
    m_nxt = m_n2;
    g_cur = g_nxt;
  }
}

// --------------------------------------------------------------------------------------------------------------
// The listed reads (a byte that is not ACGTU somewhere): one lane per read, the reference's rule -- a window is
// emitted iff its k bytes are all bases (NtHash::init / roll skipping, src/kmer.cpp:228-264) -- with the O(1)
// recurrences between consecutive valid windows and the direct formula after every skip.
// --------------------------------------------------------------------------------------------------------------
struct KmerDirtyReadsArgs {
  const uint8_t* seqs;
  const uint64_t* starts;
  const uint64_t* ends;
  const uint64_t* list;
  const unsigned long long* n_list;
  uint32_t k, m;
  uint64_t* cnt;             // count pass: exact windows of the read (written), hash pass: of every read (read)
  uint64_t* tile_sum;        // count pass: += the read's windows
  const uint64_t* tile_off;  // hash pass
  uint32_t R;                // reads per tile
  uint64_t* hashes;
  uint32_t* pos;
  uint64_t* fwd;
  uint64_t* rev;
  const uint4* horner_tab;   // hash pass: the k-independent fw tables (first_window.hpp; get_fw_tab), FW_ENTRIES entries
  uint64_t sk_fwd[4];        // srol^k(seed[code])
  uint64_t sk_rc[4];         // srol^k(seed[code ^ 2])
  // NTHIP_OUT_READ_SLOTS (hash pass only, no count pass before it): the read's k-mers go to the front of the slot its
  // length implies (tile_off = scan of the tiles' WINDOW counts), cnt[r] = how many, zeros behind them
  uint32_t slots, pad;
  // fixed-length reads (starts == NULL; NTHIP_OUT_READ_SLOTS): read r = bytes [r * fixed_stride, + fixed_len), slot r * windows
  uint32_t fixed_len, fixed_stride;
};

// NTHIP_OUT_READ_SLOTS on fixed-length reads: the dense pass set a bit per 16-byte vector that holds a non-base
// (KmerRunsArgs::vecmap).  A thread per word of that map: every read a marked vector touches goes on the list, once
// (readmap: a bit per read).  mis = address of the reads' first byte mod 16 (vector v = bytes [16 v - mis, 16 v - mis + 16)).
static __global__ __launch_bounds__(256) void slots_list_kernel(const uint32_t* __restrict__ vecmap, uint64_t n_words, uint32_t mis,
                                                                uint32_t stride, uint64_t n_reads, uint64_t total_bytes,
                                                                uint32_t* __restrict__ readmap, uint64_t* __restrict__ list,
                                                                unsigned long long* __restrict__ n_list)
{
  // (the reads found go through a block-local list: one atomic on the shared counter per block and flush, not one per
  //  read -- 20 000 atomics on one address took 0.27 ms)
  constexpr uint32_t CAP = 1024;
  __shared__ uint64_t found[CAP];
  __shared__ uint32_t n_found;
  __shared__ unsigned long long out_base;
  if (threadIdx.x == 0) n_found = 0;
  __syncthreads();
  const uint64_t per_round = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t w0 = (uint64_t)blockIdx.x * blockDim.x; w0 < n_words; w0 += per_round) { // (block-uniform trip count)
    const uint64_t w = w0 + threadIdx.x;
    uint32_t bitsw = w < n_words ? vecmap[w] : 0u;
    while (bitsw) {
      const uint32_t b = (uint32_t)__builtin_ctz(bitsw);
      bitsw &= bitsw - 1u;
      const uint64_t v = w * 32u + b;
      const uint64_t lo = (v << 4) >= mis ? (v << 4) - mis : 0u;
      uint64_t hi = (v << 4) + 15u - mis;
      if (hi >= total_bytes) hi = total_bytes - 1u;
      if (lo >= total_bytes) continue;
      for (uint64_t r = lo / stride; r <= hi / stride && r < n_reads; ++r) {
        const uint32_t bit = 1u << (r & 31u);
        if (!(atomicOr(&readmap[r >> 5], bit) & bit)) {
          const uint32_t at = atomicAdd(&n_found, 1u);
          if (at < CAP) found[at] = r;
          else list[atomicAdd(n_list, 1ull)] = r; // (a block with more than CAP reads in one round: straight out)
        }
      }
    }
    __syncthreads();
    const uint32_t nf = n_found < CAP ? n_found : CAP;
    if (nf >= CAP / 2 || w0 + per_round >= n_words) { // flush: half full, or the last round
      if (threadIdx.x == 0) out_base = nf ? atomicAdd(n_list, (unsigned long long)nf) : 0ull;
      __syncthreads();
      for (uint32_t x = threadIdx.x; x < nf; x += blockDim.x) list[out_base + x] = found[x];
      __syncthreads();
      if (threadIdx.x == 0) n_found = 0;
    }
    __syncthreads();
  }
}

constexpr uint32_t RD_MAX_LEN = 2048; // longest read this path takes (the host checks)

// One WAVE per listed read: the read's bytes in LDS, for every byte the place of the last non-base at or before
// it (a wave-wide running maximum), then 64 window starts at a time: a window is emitted iff no non-base lies in
// it, its place in the read's output is the running count of emitted windows (ballot + mbcnt), its hash the
// direct formula (Horner over its k bytes: F forward from the first base, R backward from the last).
template <bool COUNT_ONLY>
__global__ __launch_bounds__(256) void kmer_dirty_reads_kernel(const KmerDirtyReadsArgs a)
{
  __shared__ uint8_t raw_all[4][RD_MAX_LEN + 64];
  __shared__ uint16_t lb_all[4][RD_MAX_LEN + 64]; // 1 + index of the last non-base in [0, p], 0 if none
  // hash pass: the read as 2-bit codes too, and the k-independent tables of horner_first_window (4 bases per step:
  // a window is 2 * ceil(k / 4) table steps instead of 2k character steps)
  __shared__ uint32_t bits_all[COUNT_ONLY ? 1 : 4][COUNT_ONLY ? 1 : RD_MAX_LEN / 16 + 8];
  __shared__ uint4 htab[COUNT_ONLY ? 1 : FW_ENTRIES];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint8_t* raw = raw_all[wave];
  uint16_t* lb = lb_all[wave];
  uint32_t* bits = bits_all[COUNT_ONLY ? 0 : wave];
  if (!COUNT_ONLY) {
    for (uint32_t i = threadIdx.x; i < FW_ENTRIES; i += blockDim.x) htab[i] = a.horner_tab[i];
    __syncthreads();
  }
  const uint64_t n = *a.n_list;
  const uint32_t k = a.k, m = a.m;
  const uint64_t kmul = (uint64_t)k * MULTISEED;
  for (uint64_t i = (uint64_t)blockIdx.x * 4u + wave; i < n; i += (uint64_t)gridDim.x * 4u) {
    const uint64_t r = a.list[i];
    const bool fixed = a.starts == nullptr;
    const uint8_t* s = a.seqs + (fixed ? r * a.fixed_stride : a.starts[r]);
    const uint32_t len = fixed ? a.fixed_len : (uint32_t)(a.ends[r] - a.starts[r]);
    if (!COUNT_ONLY) {
      for (uint32_t j = lane; j < (len >> 4) + 8u; j += 64u) bits[j] = 0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
    }
    uint32_t carry = 0;
    for (uint32_t p0 = 0; p0 < len; p0 += 64u) {
      const uint32_t p = p0 + lane;
      const uint8_t c = p < len ? s[p] : (uint8_t)'A';
      if (!COUNT_ONLY && p < len && is_base(c)) atomicOr(&bits[p >> 4], code_of(c) << ((p & 15u) << 1));
      uint32_t v = (p < len && !is_base(c)) ? p + 1u : 0u;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)v, d, 64);
        if ((int)lane >= d && o > v) v = o;
      }
      if (carry > v) v = carry;
      raw[p] = c;
      lb[p] = (uint16_t)v;
      carry = (uint32_t)__shfl((int)v, 63, 64);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    const uint32_t nwin = len >= k ? len - k + 1u : 0u;
    uint64_t base = 0;
    if (!COUNT_ONLY && fixed) {
      base = r * (uint64_t)nwin; // (every slot has the fixed window count)
    } else if (!COUNT_ONLY) { // the tile's first k-mer + the k-mers of the reads before this one in its tile
      const uint64_t r0 = r / a.R * a.R;
      uint32_t before = 0;
      if (r0 + lane < r) {
        if (a.slots) { // windows, not emitted k-mers: a slot's place depends on lengths alone
          const uint64_t l2 = a.ends[r0 + lane] - a.starts[r0 + lane];
          before = l2 >= k ? (uint32_t)(l2 - k + 1u) : 0u;
        } else {
          before = (uint32_t)a.cnt[r0 + lane];
        }
      }
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) before += (uint32_t)__shfl_xor((int)before, d, 64);
      base = a.tile_off[r / a.R] + before;
    }
    uint32_t emitted = 0;
    for (uint32_t w0 = 0; w0 < nwin; w0 += 64u) {
      const uint32_t w = w0 + lane;
      const bool valid = w < nwin && lb[w + k - 1u] <= w; // no non-base in [w, w + k)
      const uint64_t mask = __ballot(valid);
      if (!COUNT_ONLY && valid) {
        const uint32_t slot = emitted + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        uint32_t f_lo, f_hi, r_lo, r_hi; // (a valid window holds bases only: its 2-bit codes are all it takes)
        any_k_first_window(bits, htab, w, k, f_lo, f_hi, r_lo, r_hi);
        const uint64_t fh = ((uint64_t)f_hi << 32) | f_lo, rh = ((uint64_t)r_hi << 32) | r_lo;
        const uint64_t o = base + slot;
        const uint64_t h0 = fh + rh;
        a.hashes[o * m] = h0;
        for (uint32_t jj = 1; jj < m; ++jj) a.hashes[o * m + jj] = mix_hash(h0, (uint64_t)jj ^ kmul);
        if (a.pos) a.pos[o] = w;
        if (a.fwd) a.fwd[o] = fh;
        if (a.rev) a.rev[o] = rh;
      }
      emitted += (uint32_t)__builtin_popcountll(mask);
    }
    if (COUNT_ONLY && lane == 0) {
      a.cnt[r] = emitted;
      atomicAdd((unsigned long long*)&a.tile_sum[r / a.R], (unsigned long long)emitted);
    }
    if (!COUNT_ONLY && a.slots) { // the rest of the slot: zeros (a whole-array checksum then equals the stream's), the count
      for (uint32_t w = emitted + lane; w < nwin; w += 64u) {
        for (uint32_t jj = 0; jj < m; ++jj) a.hashes[(base + w) * m + jj] = 0;
        if (a.pos) a.pos[base + w] = 0;
      }
      if (lane == 0) a.cnt[r] = emitted;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
  }
}

// NTHIP_OUT_READ_SLOTS: windows of every tile of R reads (one lane per tile): their scan places the tiles' slots
static __global__ __launch_bounds__(256) void reads_tile_windows_kernel(const uint64_t* __restrict__ starts,
                                                                        const uint64_t* __restrict__ ends, uint64_t n,
                                                                        uint32_t R, uint32_t k, uint64_t n_tiles,
                                                                        uint64_t* __restrict__ tile_sum)
{
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_tiles; t += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t r0 = t * R, r1 = r0 + R < n ? r0 + R : n;
    uint64_t sum = 0;
    for (uint64_t r = r0; r < r1; ++r) {
      const uint64_t l = ends[r] - starts[r];
      sum += l >= k ? l - k + 1u : 0u;
    }
    tile_sum[t] = sum;
  }
}

// max length, max distance between consecutive starts, order, total length: what the host needs to size the tiles
// tile_sum != nullptr (NTHIP_OUT_READ_SLOTS): also the windows of every tile of R reads -- (len - k + 1 where len >= k),
// summed per tile with a wave-level segmented scan and at most a few atomics per wave (tile_sum zeroed by the host)
// (launched with READS_PREP_THREADS threads and at most two blocks per CU: the sums and maxima end in same-address atomics, ~40 ns
// each one after the other -- 2048 blocks of 256 spent 0.16 of the kernel's 0.2 ms there)
constexpr uint32_t READS_PREP_THREADS = 1024;
static __global__ __launch_bounds__(READS_PREP_THREADS) void reads_prep_kernel(const uint64_t* __restrict__ starts,
                                                                const uint64_t* __restrict__ ends, uint64_t n,
                                                                uint64_t buf_bytes, unsigned long long* __restrict__ res,
                                                                uint32_t allow_overlap = 0, uint32_t R = 0, uint32_t k = 0,
                                                                unsigned long long* __restrict__ tile_sum = nullptr,
                                                                unsigned long long* __restrict__ win_sum = nullptr)
{
  // allow_overlap: consecutive reads may share bytes as long as starts and ends both go up (the pieces of a long read
  // overlap by k - 1: seed_rtile_kernel only needs a tile's reads inside one slab that ends with the last read)
  uint64_t mlen = 0, mpitch = 0, slen = 0, wsum = 0;
  uint32_t bad = 0;
  const uint32_t lane = threadIdx.x & 63u;
  // (whole waves iterate together: the segmented scan below needs every lane of a wave in the loop)
  // Round 5: the spans of the NEXT 64 reads are asked for before this wave's are looked at, and a read's successor comes from
  // the neighbouring lane (lane 63 loads it): one iteration's loads in flight per wave and four load instructions per read
  // made 0.19 ms of 320 MB of spans (1.7 TB/s).
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint64_t r0 = (uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u);
  uint64_t s_nx = 0, e_nx = 0, s_tl = 0, e_tl = 0; // this lane's read of the iteration; lane 63: the read behind the wave's
  auto fetch = [&](uint64_t base) {
    const uint64_t r = base + lane;
    s_nx = r < n ? starts[r] : 0;
    e_nx = r < n ? ends[r] : 0;
    if (lane == 63u && r + 1 < n) {
      s_tl = starts[r + 1];
      e_tl = ends[r + 1];
    }
  };
  if (r0 < n) fetch(r0);
  for (; r0 < n; r0 += stride) {
    const uint64_t r = r0 + lane;
    const uint64_t s0 = s_nx, e0 = e_nx, st = s_tl, et = e_tl;
    if (r0 + stride < n) fetch(r0 + stride);
    // the read behind this lane's: the next lane's, or what lane 63 loaded
    uint64_t s1 = ((uint64_t)(uint32_t)__shfl_down((int)(uint32_t)(s0 >> 32), 1, 64) << 32) | (uint32_t)__shfl_down((int)(uint32_t)s0, 1, 64);
    uint64_t e1 = ((uint64_t)(uint32_t)__shfl_down((int)(uint32_t)(e0 >> 32), 1, 64) << 32) | (uint32_t)__shfl_down((int)(uint32_t)e0, 1, 64);
    if (lane == 63u) {
      s1 = st;
      e1 = et;
    }
    uint64_t nwin = 0;
    if (r < n) {
      if (e0 < s0 || e0 > buf_bytes) {
        bad = 1; // (also what check_spans_kernel looks for)
      } else {
        if (e0 - s0 > mlen) mlen = e0 - s0;
        slen += e0 - s0;
        nwin = e0 - s0 >= k ? e0 - s0 - k + 1u : 0u;
        wsum += nwin;
        if (r + 1 < n) {
          if (s1 < e0 && !(allow_overlap && s1 >= s0 && e1 >= e0)) bad = 1; // not in order, or overlapping
          else if (s1 - s0 > mpitch) mpitch = s1 - s0;
        }
      }
    }
    if (tile_sum) {
      // inclusive prefix sum of the wave's window counts on the DPP network (a read has < 2^32 windows and so have 64
      // of them here: reads of this path are at most RD_MAX_LEN bytes -- longer ones make the sums meaningless, and the
      // caller drops them when it sees the maximum length); six LDS round trips of a 64-bit __shfl_up scan cost the
      // kernel two thirds of its time
      uint32_t incl = nwin < 0x3FFFFFFull ? (uint32_t)nwin : 0x3FFFFFFu;
      incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, true);
      incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, true);
      incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, true);
      incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, true);
      incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xa, 0xf, false);
      incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xc, 0xf, false);
      const uint32_t in_tile = (uint32_t)(r % R);                     // place of the read in its tile
      const uint32_t seg0 = in_tile <= lane ? lane - in_tile : 0u;     // first lane of the tile's part in this wave
      const uint32_t src = seg0 ? seg0 - 1u : 0u;
      const uint32_t before_all = (uint32_t)__shfl((int)incl, (int)src, 64);
      const uint32_t before = seg0 ? before_all : 0u;
      const bool last = r < n && (in_tile == R - 1u || lane == 63u || r == n - 1u);
      if (last && incl != before) atomicAdd(&tile_sum[r / R], (unsigned long long)(incl - before));
    }
  }
  // wave-level reduction, then one atomic per wave
  for (int d = 32; d > 0; d >>= 1) {
    const uint64_t ol = ((uint64_t)(uint32_t)__shfl_down((int)(uint32_t)(mlen >> 32), d, 64) << 32) |
                        (uint32_t)__shfl_down((int)(uint32_t)mlen, d, 64);
    const uint64_t op = ((uint64_t)(uint32_t)__shfl_down((int)(uint32_t)(mpitch >> 32), d, 64) << 32) |
                        (uint32_t)__shfl_down((int)(uint32_t)mpitch, d, 64);
    const uint64_t os = ((uint64_t)(uint32_t)__shfl_down((int)(uint32_t)(slen >> 32), d, 64) << 32) |
                        (uint32_t)__shfl_down((int)(uint32_t)slen, d, 64);
    const uint64_t ow = ((uint64_t)(uint32_t)__shfl_down((int)(uint32_t)(wsum >> 32), d, 64) << 32) |
                        (uint32_t)__shfl_down((int)(uint32_t)wsum, d, 64);
    if (ol > mlen) mlen = ol;
    if (op > mpitch) mpitch = op;
    slen += os;
    wsum += ow;
  }
  __shared__ unsigned long long block_sum, block_win; // total length / windows (k given) of the reads: one atomic per block
  if (threadIdx.x == 0) {
    block_sum = 0;
    block_win = 0;
  }
  __syncthreads();
  if ((threadIdx.x & 63u) == 0) {
    atomicAdd(&block_sum, (unsigned long long)slen);
    atomicAdd(&block_win, (unsigned long long)wsum);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&res[3], block_sum);
    if (win_sum) atomicAdd(win_sum, block_win); // (every window of every read: what the batch emits when no read holds a non-base)
  }
  // (thousands of waves hammering one address serialise in L2: look first, the maximum is reached early)
  if ((threadIdx.x & 63u) == 0) {
    if (mlen > __atomic_load_n(&res[0], __ATOMIC_RELAXED)) atomicMax(&res[0], (unsigned long long)mlen);
    if (mpitch > __atomic_load_n(&res[1], __ATOMIC_RELAXED)) atomicMax(&res[1], (unsigned long long)mpitch);
  }
  if (__ballot(bad != 0) != 0 && (threadIdx.x & 63u) == 0 && __atomic_load_n(&res[2], __ATOMIC_RELAXED) == 0)
    atomicOr(&res[2], 1ull);
}

} // namespace ntamd
