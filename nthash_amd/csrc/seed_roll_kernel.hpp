// seed_roll_kernel.hpp -- spaced seeds rolled run by run (round 4).
//
// The dense spaced-seed kernels (seed_kernels.hpp) hash every window from scratch: ceil(k / 4) table lookups of 16 bytes per
// seed and window, whatever the seed looks like -- a single seed of 128 bases runs at 0.07-0.11 of the HBM roofline, bound by
// LDS.  The reference rolls a seed (NTMSM64, src/seed.cpp:177-207): per care run [a, b) of the seed one base enters and one
// leaves,
//     F' = srol(F) ^ XOR_runs [ srol^(k - b)(S[in]) ^ srol^(k - a)(S[out]) ]
//     R' = sror( R ^ XOR_runs [ srol^b(S'[in]) ^ srol^a(S'[out]) ] )             (S' = the complement's seed value)
// with in = the base that becomes the run's last, out = the base in front of the run's first -- O(runs) per window instead
// of O(k).  The runs are the seed's care runs, monomers being runs of one (get_blocks' other description,
// src/seed.cpp:19-66 -- the whole window XOR its don't-care runs -- never has fewer terms: care and don't-care runs
// alternate).  One (in, out) pair table of 16 entries x 16 bytes per run: 256 bytes = one row of the LDS banks, entry e on
// banks 4e..4e+3, so that the 64 lanes' lookups never conflict whatever they read (lanes on the same entry share it).
//
// Who rolls what: the windows of every read are cut into SEGMENTS of SW consecutive windows (the last one of a read
// shorter); a lane takes one segment -- its first window from the k-independent tables of the any-seed form (sa_strands,
// seed_kernels.hpp: the cost of one window of the direct form, once per SW windows), the other SW - 1 by rolling, the
// lookups of the SW - 1 steps of a run in flight together -- for one seed after the other.  64 consecutive segments are a
// wave's tile: their values are one contiguous piece of the output stream, collected in a wave-private LDS stage laid out
// like the stream (swizzled per 16 values against bank conflicts) and written out 16 bytes per lane, 1 KiB per
// instruction, on whole 128-byte lines but for the tile's two ends.  (A first version gave every lane a whole read, as
// kmer_fixed_kernel does: 128-byte pieces of output at the pitch of a read's record, 214 G k-mers/s for one seed of 128
// bases in three runs whether the first window was rolled up to or looked up -- the stores, not the hashing.)
// The next tile's bases are loaded while this one is hashed.  A non-base sets a.dirty (SeedNtHash's position state machine,
// App. B Q3, is the other kernels' business).
#pragma once

#include <hip/hip_runtime.h>

#include "seed_kernels.hpp"

namespace ntamd {

constexpr uint32_t SR_MAX_RUNS = 64;    // all seeds together: 256 B of pair table each
constexpr uint32_t SR_MAX_WAVES = 8;
constexpr uint32_t SR_VEC_ROUNDS = 4;   // a tile's bases: at most 4 x 64 vectors of 16 bytes

struct SeedRollArgs {
  const uint8_t* seqs;
  uint64_t* hashes;        // dense [read][window][seed][m2]
  uint32_t* dirty;
  const uint4* pair_tabs;  // [n_runs][16]: entry (in << 2) | out -> {F term lo, hi, R term lo, hi}
  const uint4* fw_tabs;    // the k-independent tables of first_window.hpp
  const uint32_t* any_mask;  // [n_seeds][any_groups]  (nthip_seeds_create)
  const uint4* any_acorr;    // [n_seeds][any_groups]
  uint64_t n_items;        // segments of the batch: n_reads * segs
  uint64_t n_tiles;        // wave tiles of 64 segments
  uint32_t len, k, m2, n_seeds, nwin, any_groups;
  uint32_t segs;           // segments per read: ceil(nwin / SW)
  uint32_t inv_segs;       // floor(2^32 / segs) + 1  (only used while 1 < segs < 64)
  uint32_t n_runs, waves, bits_dwords, stage_vals;
  uint64_t total_bytes;    // of the batch: n_reads * len (whoever loads a byte of the batch judges it)
  uint64_t step_reads;     // a wave's next tile is grid * waves tiles on: (64 * grid * waves) / segs reads
  uint32_t step_segs;      // and (64 * grid * waves) % segs segments
  uint32_t seed_first[SR_MAX_RUNS + 1]; // the runs of seed s: [seed_first[s], seed_first[s + 1])  (seeds <= runs)
  // the roll from window p to p + 1 takes run j's incoming base from position p + run_end[j] of the read and its outgoing
  // one from p + run_first[j]
  uint32_t run_first[SR_MAX_RUNS], run_end[SR_MAX_RUNS];
  uint64_t mult[SF_MAX_RUNTIME_M];
};

// value index v of the stage -> where it lives: 16-value rows, the column XOR-ed with a function of the row so that lanes
// that write the same column of consecutive rows hit different banks; a row's aligned 16-byte pairs stay pairs (swapped
// when the function is odd)
__device__ __forceinline__ uint32_t sr_row_key(uint32_t row) { return ((row & 7u) << 1) | ((row >> 3) & 1u); }
__device__ __forceinline__ uint32_t sr_phys(uint32_t v) { return v ^ sr_row_key(v >> 4); }

template <int SW>
__global__ __launch_bounds__(SR_MAX_WAVES * 64) void seed_roll_kernel(const SeedRollArgs a)
{
  extern __shared__ __attribute__((aligned(256))) uint32_t lds_dyn[];
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // LDS: [any-seed tables | group constants | masks][pair tables][per wave: stage | bit stream x 2]
  uint4* const tabs = (uint4*)lds_dyn;
  const uint32_t G = a.any_groups, n_grp = a.n_seeds * G;
  const uint4* const g_acorr = tabs + SA_TAB_ENTRIES;
  const uint32_t* const g_mask = (const uint32_t*)(g_acorr + n_grp);
  const uint32_t n_entries = SA_TAB_ENTRIES + n_grp + ((n_grp + 3u) >> 2);
  uint4* const ptab = tabs + ((n_entries + 15u) & ~15u); // (256-byte rows)
  uint64_t* const stage = (uint64_t*)(ptab + a.n_runs * 16u) + (size_t)wave * (a.stage_vals + 2u * (a.bits_dwords >> 1));
  uint32_t* const bits_all = (uint32_t*)(stage + a.stage_vals);
  sa_load(tabs, a.fw_tabs, a.any_acorr, a.any_mask, n_grp, G, tid, a.waves * 64u);
  for (uint32_t i = tid; i < a.n_runs * 16u; i += a.waves * 64u) ptab[i] = a.pair_tabs[i];
  __syncthreads();

  const uint32_t k = a.k, m2 = a.m2, per = a.n_seeds * m2, nwin = a.nwin, segs = a.segs;
  const uint32_t k31 = k % 31u, k33 = k % 33u;
  uint32_t bad = 0;

  // the tile's place in the batch (all uniform)
  struct Tile {
    uint64_t read0;      // its first segment's read
    uint32_t seg0;       // and number in the read
    uint32_t n_here;     // segments (64 but for the batch's last tile)
    uint32_t shift;      // foreign bytes in front of its first vector
    uint32_t slab_bytes; // bytes of its segments' windows
    uint32_t n_vec;
    const uint4* vsrc;
  };
  auto lane_place = [&](uint32_t seg0, uint32_t l, uint32_t& rl, uint32_t& seg) { // segment l of a tile: read (relative), number
    const uint32_t local = seg0 + l;
    rl = segs >= 64u ? (local >= segs ? 1u : 0u) : segs == 1u ? local : __umulhi(local, a.inv_segs);
    seg = local - rl * segs;
  };
  auto finish = [&](Tile& T, uint64_t t) { // (T.read0, T.seg0 set)
    const uint64_t left = a.n_items - t * 64u;
    T.n_here = left < 64u ? (uint32_t)left : 64u;
    uint32_t rl_l, seg_l;
    lane_place(T.seg0, T.n_here - 1u, rl_l, seg_l);
    const uint32_t w_end = seg_l * SW + SW < nwin ? seg_l * SW + SW : nwin; // (exclusive) last window of the tile, in its read
    const uint64_t addr0 = (uint64_t)(a.seqs + T.read0 * a.len + (uint64_t)T.seg0 * SW);
    T.shift = (uint32_t)(addr0 & 15u);
    T.vsrc = (const uint4*)(addr0 - T.shift);
    T.slab_bytes = rl_l * a.len + w_end + k - 1u - T.seg0 * SW;
    T.n_vec = (T.shift + T.slab_bytes + 15u) >> 4;
  };
  auto place = [&](uint64_t t) -> Tile { // (a 64-bit division: the wave's first tile only)
    Tile T;
    const uint64_t q0 = t * 64u;
    T.read0 = q0 / segs;
    T.seg0 = (uint32_t)(q0 - T.read0 * segs);
    finish(T, t);
    return T;
  };
  auto place_next = [&](const Tile& T, uint64_t t_next) -> Tile {
    Tile N;
    N.read0 = T.read0 + a.step_reads;
    N.seg0 = T.seg0 + a.step_segs;
    if (N.seg0 >= segs) {
      N.seg0 -= segs;
      ++N.read0;
    }
    finish(N, t_next);
    return N;
  };
  uint4 nx[SR_VEC_ROUNDS];
  auto load = [&](const Tile& T) {
#pragma unroll
    for (uint32_t r = 0; r < SR_VEC_ROUNDS; ++r) {
      const uint32_t i = r * 64u + lane;
      nx[r] = i < T.n_vec ? T.vsrc[i] : make_uint4(0, 0, 0, 0);
    }
  };
  auto pack = [&](const Tile& T, uint32_t* bits) {
#pragma unroll
    for (uint32_t r = 0; r < SR_VEC_ROUNDS; ++r) {
      const uint32_t i = r * 64u + lane;
      if (i < T.n_vec + 2u) { // (the funnels read a word ahead: two zero words behind the slab)
        uint32_t b = 0;
        const uint32_t p = pack16(nx[r], b);
        // a byte of the batch is judged by whoever loads it (the flag is the batch's); only the vectors that hold the
        // batch's first and last bytes have somebody else's bytes in them
        const uint64_t va = (uint64_t)(T.vsrc + i);
        if (i >= T.n_vec) b = 0;
        else if (va < (uint64_t)a.seqs || va + 16u > (uint64_t)a.seqs + a.total_bytes) {
          const int64_t lo_cut = (int64_t)((uint64_t)a.seqs - va), hi_cut = (int64_t)((uint64_t)a.seqs + a.total_bytes - va);
          uint32_t bx[4] = {0, 0, 0, 0};
          (void)pack4(nx[r].x, bx[0]);
          (void)pack4(nx[r].y, bx[1]);
          (void)pack4(nx[r].z, bx[2]);
          (void)pack4(nx[r].w, bx[3]);
          b = 0;
          for (int q = 0; q < 16; ++q)
            if (q >= lo_cut && q < hi_cut) b |= (bx[q >> 2] >> ((q & 3) * 8)) & 0xFFu;
        }
        bad |= b;
        bits[i] = i < T.n_vec ? p : 0u;
      }
    }
  };

  const uint64_t t_step = (uint64_t)gridDim.x * a.waves;
  uint64_t t = (uint64_t)blockIdx.x * a.waves + wave;
  uint32_t cur = 0;
  Tile T;
  if (t < a.n_tiles) {
    T = place(t);
    load(T);
    pack(T, bits_all);
  }
  for (; t < a.n_tiles; t += t_step, cur ^= 1u) {
    uint32_t* const bits = bits_all + cur * a.bits_dwords;
    const bool more = t + t_step < a.n_tiles;
    Tile Tn;
    if (more) { // the next tile's bases: in flight while this one is hashed
      Tn = place_next(T, t + t_step);
      load(Tn);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // (the bit stream and the stage are this wave's alone)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // this lane's segment
    uint32_t rl, seg;
    lane_place(T.seg0, lane, rl, seg);
    const bool active = lane < T.n_here;
    const uint32_t w0 = seg * SW;
    const uint32_t n_w = !active ? 0u : (nwin - w0 < (uint32_t)SW ? nwin - w0 : (uint32_t)SW);
    const uint32_t rel = active ? rl * nwin + w0 - T.seg0 * SW : 0u; // its first window, counted from the tile's first
    const uint32_t P = active ? T.shift + rl * a.len + w0 - T.seg0 * SW : 0u; // and first base in the bit stream
    const uint64_t g0 = (T.read0 * nwin + (uint64_t)T.seg0 * SW) * per; // the tile's first value in the output stream
    const uint32_t off = (uint32_t)(g0 & 15u);
    const uint32_t v0 = off + rel * per;

    for (uint32_t s = 0; s < a.n_seeds; ++s) {
      uint64_t hw[SW]; // the segment's h[0] under this seed
      auto emit = [&](uint32_t w, uint32_t f_lo, uint32_t f_hi, uint32_t r_lo, uint32_t r_hi) { // window w of the segment
        hw[w] = canon_pair(f_lo, f_hi, r_lo, r_hi);
      };
      uint4 st = sa_strands(tabs, g_acorr, g_mask, bits, P >> 4, (P & 15u) << 1, s, G, k31, k33);
      emit(0, st.x, st.y, st.z, st.w);
      // the SW - 1 rolls: every run's (in, out) nibbles, XOR-ed up per step
      uint4 tacc[SW - 1];
#pragma unroll
      for (int i = 0; i < SW - 1; ++i) tacc[i] = make_uint4(0, 0, 0, 0);
      for (uint32_t b = a.seed_first[s]; b < a.seed_first[s + 1]; ++b) {
        const uint32_t pi = P + a.run_end[b], po = P + a.run_first[b];
        const uint32_t w_in = funnel(bits[(pi >> 4) + 1u], bits[pi >> 4], (pi & 15u) << 1);
        const uint32_t w_out = funnel(bits[(po >> 4) + 1u], bits[po >> 4], (po & 15u) << 1);
        // nibble streams: u = even steps, v = odd steps; nibble = (in << 2) | out
        const uint32_t u = ((w_in & 0x33333333u) << 2) | (w_out & 0x33333333u);
        const uint32_t v = (w_in & 0xCCCCCCCCu) | ((w_out >> 2) & 0x33333333u);
        const char* const tb = (const char*)(ptab + b * 16u);
        uint4 e[SW - 1]; // all the run's lookups in flight before the first is used
#pragma unroll
        for (int i = 0; i < SW - 1; ++i) {
          const uint32_t src = (i & 1) ? v : u;
          e[i] = *(const uint4*)(tb + (((src >> ((i >> 1) * 4)) & 0xFu) << 4));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < SW - 1; ++i) {
          tacc[i].x ^= e[i].x; tacc[i].y ^= e[i].y; tacc[i].z ^= e[i].z; tacc[i].w ^= e[i].w;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < SW - 1; ++i) {
        roll_step(st.x, st.y, st.z, st.w, tacc[i]);
        emit((uint32_t)i + 1u, st.x, st.y, st.z, st.w);
      }
      const uint32_t vs = v0 + s * m2;
      if (per == 1u) { // (uniform) one value per window: the segment covers at most two rows of the stage
        const uint32_t col0 = vs & 15u, key0 = sr_row_key(vs >> 4), key1 = sr_row_key((vs >> 4) + 1u);
#pragma unroll
        for (uint32_t w = 0; w < (uint32_t)SW; ++w)
          if (w < n_w) stage[(vs + w) ^ (col0 + w >= 16u ? key1 : key0)] = hw[w];
      } else if (m2 == 1u) {
#pragma unroll
        for (uint32_t w = 0; w < (uint32_t)SW; ++w)
          if (w < n_w) stage[sr_phys(vs + w * per)] = hw[w];
      } else {
#pragma unroll
        for (uint32_t w = 0; w < (uint32_t)SW; ++w)
          if (w < n_w) {
            stage[sr_phys(vs + w * per)] = hw[w];
            for (uint32_t jj = 1; jj < m2; ++jj)
              stage[sr_phys(vs + w * per + jj)] = mix_hash(hw[w], a.mult[jj & (SF_MAX_RUNTIME_M - 1)]);
          }
      }
    }

    if (more) pack(Tn, bits_all + (cur ^ 1u) * a.bits_dwords); // (before this tile's stores: a wait for the loads is a wait
                                                               //  for everything issued before them)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // the stage -> the stream: rows of 16 values = 128-byte lines of the stream, eight rows per instruction
    uint32_t rl_l, seg_l;
    lane_place(T.seg0, T.n_here - 1u, rl_l, seg_l);
    const uint32_t w_end = seg_l * SW + SW < nwin ? seg_l * SW + SW : nwin;
    const uint32_t v_end = off + (rl_l * nwin + w_end - T.seg0 * SW) * per; // (exclusive)
    const uint32_t n_rows = (v_end + 15u) >> 4;
    uint64_t* const dst0 = a.hashes + (g0 - off);
    const uint32_t c = lane & 7u;
    for (uint32_t r0 = 0; r0 < n_rows; r0 += 32u) {
      uint4 d[4];
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) {
        const uint32_t R = r0 + u * 8u + (lane >> 3), key = sr_row_key(R);
        const uint4 q = *(const uint4*)(stage + R * 16u + 2u * (c ^ (key >> 1)));
        d[u] = (key & 1u) ? make_uint4(q.z, q.w, q.x, q.y) : q;
      }
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) {
        const uint32_t R = r0 + u * 8u + (lane >> 3), v = R * 16u + 2u * c;
        const bool lo_ok = v >= off && v < v_end, hi_ok = v + 1u >= off && v + 1u < v_end;
        uint64_t* const dst = dst0 + v;
        if (lo_ok && hi_ok) *(uint4*)dst = d[u];
        else if (lo_ok) *(uint2*)dst = make_uint2(d[u].x, d[u].y);
        else if (hi_ok) *(uint2*)(dst + 1) = make_uint2(d[u].z, d[u].w);
      }
    }
    T = Tn;
  }
  if (__ballot(bad != 0) != 0 && lane == 0) atomicOr(a.dirty, 1u);
}

} // namespace ntamd
