// seed_ps_kernel.hpp -- spaced seeds as sparse sums over the prefix XOR of a read's terms, a lane per SEGMENT (round 6).
//
// seed_px_plan.hpp has the algebra: with the terms of a read in one frame -- here the read's own,
//     T(j) = srol^{-j}(S[c_j])      U(j) = srol^{j}(S[comp c_j])      (j: position in the read; one 16-byte entry {T, U})
// and P(j) = XOR_{j' < j} {T, U}(j'), the strand hashes of the window at p under a seed with care runs [a_i, b_i) are
//     F = srol^{p+k-1}( XOR_i P(p + b_i) ^ P(p + a_i) ).lo64      R = srol^{-p}( ... ).hi64
// -- two reads of 16 bytes per care run, what the reference's roll pays per block (src/seed.cpp:177-207), no table of k,
// no first window, no chain from window to window; a monomer is one read of the term array instead of two of the prefix.
//
// seed_px_kernel.hpp builds such arrays two positions per lane and step, with a wave-wide scan per step and a pair of
// per-lane split rotates per position: ~85 instructions per position, more than everything else together when k is half
// the read.  Here a lane owns a SEGMENT of W consecutive positions of one read:
//   * the frame is the read's, so the rotated term of (position, base) is the same for every read: one LDS table of
//     4 x (len + 1) entries per block, a conflict-free lookup instead of the rotates;
//   * the prefix inside a segment is sequential in the lane (4 XORs per position), only the segment totals are scanned
//     across the lanes of a read (once per W positions);
//   * the arrays are stored TRANSPOSED per read -- position j at slot (j % W) * NB + j / W -- so that the lanes of a store
//     (segment s, step i: position s W + i) and the lanes of a read instruction (window segments: window s W + i, read at
//     s W + i + e) both touch consecutive slots: no bank conflict in either direction whatever W, the seed and the bases.
// The windows of a read are cut into segments of the same W; NB is a power of two >= the segments of a read's positions,
// the window lanes per read a power of two >= the segments of its windows, 64 of them a wave's tile (1, 2 or 4 reads).
// A lane stores its values itself: consecutive windows of a segment, consecutive segments, consecutive reads are
// consecutive in the stream.  A non-base sets a.dirty (SeedNtHash's position state machine, App. B Q3, is the other
// kernels' business).
#pragma once

#include <hip/hip_runtime.h>

#include "seed_px_kernel.hpp"

namespace ntamd {

constexpr uint32_t PS_MAX_WAVES = 16;
constexpr uint32_t PS_MAX_W = 32;

struct SeedPsArgs {
  const uint8_t* seqs;
  uint64_t* hashes; // dense [read][window][seed][m2]
  uint32_t* dirty;
  const uint32_t* step_off; // [W][n_terms]: byte offset of term t's read at step i from the lane's own slot 0 of array 0
  uint64_t n_reads, n_tiles;
  uint64_t total_bytes; // n_reads * len
  uint32_t len, k, m2, n_seeds, nwin;
  uint32_t W;           // positions / windows per segment
  uint32_t nb_log;      // NB = 1 << nb_log slots per row of a read's arrays; entries per read = W << nb_log
  uint32_t lpr_log;     // window lanes per read = 1 << lpr_log
  uint32_t n_arrays;    // 1: the prefix; 2: the prefix, then the terms themselves
  uint32_t n_terms, waves;
  uint32_t segs_b;      // segments of a read's positions 0 .. len (the prefix has an entry behind the last base)
  uint32_t k31, k33;    // (k - 1) % 31, (k - 1) % 33
  uint32_t w31, w33;    // W % 31, W % 33
  uint32_t seed_first[PX_MAX_SEEDS + 1];
  uint64_t mult[SF_MAX_RUNTIME_M];
};

// exclusive XOR scan over aligned groups of 1 << width_log lanes (4, 5 or 6)
__device__ __forceinline__ uint32_t ps_xor_scan(uint32_t v, uint32_t width_log)
{
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
  if (width_log >= 5u) v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
  if (width_log >= 6u) v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
  return v;
}

// PERC: values per window known at compile time (1, 2) or 0 (any)
template <int PERC>
__global__ __launch_bounds__(PS_MAX_WAVES * 64) void seed_ps_kernel(const SeedPsArgs a)
{
  extern __shared__ __attribute__((aligned(256))) uint32_t lds_dyn[];
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t W = a.W, NB = 1u << a.nb_log, epr = W << a.nb_log; // entries per read and array
  const uint32_t rpw = 64u >> a.lpr_log;                           // reads per tile
  // LDS: [term table: 4 codes x epr entries][per wave: n_arrays x rpw x epr entries | 2-bit codes]
  uint4* const ttab = (uint4*)lds_dyn;
  const uint32_t tile_bytes_max = rpw * a.len + 16u;
  const uint32_t codes_dw = ((tile_bytes_max + 15u) >> 4) + 4u;
  const uint32_t arr_bytes = rpw * epr * 16u;
  const uint32_t wave_bytes = a.n_arrays * arr_bytes + ((codes_dw * 4u + 15u) & ~15u);
  char* const arrays = (char*)(ttab + 4u * epr) + (size_t)wave * wave_bytes;
  uint32_t* const codes = (uint32_t*)(arrays + a.n_arrays * arr_bytes);
  // the term table, in the arrays' own layout: {T, U}(j) of code c at entry c * epr + (j % W) * NB + j / W
  for (uint32_t e = tid; e < 4u * epr; e += a.waves * 64u) {
    const uint32_t c = e / epr, sl = e - c * epr, row = sl >> a.nb_log, col = sl & (NB - 1u);
    const uint32_t j = col * W + row;
    uint64_t f = seed_of_code(c), r = seed_of_code(c ^ 2u);
    uint32_t f_lo = (uint32_t)f, f_hi = (uint32_t)(f >> 32), r_lo = (uint32_t)r, r_hi = (uint32_t)(r >> 32);
    const uint32_t j31 = j % 31u, j33 = j % 33u;
    px_srol_var(f_lo, f_hi, j31 ? 31u - j31 : 0u, j33 ? 33u - j33 : 0u);
    px_srol_var(r_lo, r_hi, j31, j33);
    ttab[e] = make_uint4(f_lo, f_hi, r_lo, r_hi);
  }
  __syncthreads();

  const uint32_t m2 = a.m2, per = PERC ? (uint32_t)PERC : a.n_seeds * m2, nwin = a.nwin, len = a.len;
  uint32_t bad = 0;

  struct Tile {
    uint64_t read0;
    uint32_t n_r;    // reads (rpw but for the batch's last tile)
    uint32_t shift;  // foreign bytes in front of its first vector
    uint32_t n_vec;
    const uint4* vsrc;
  };
  auto place = [&](uint64_t t) -> Tile {
    Tile T;
    T.read0 = t * rpw;
    const uint64_t left = a.n_reads - T.read0;
    T.n_r = left < rpw ? (uint32_t)left : rpw;
    const uint64_t addr0 = (uint64_t)(a.seqs + T.read0 * len);
    T.shift = (uint32_t)(addr0 & 15u);
    T.vsrc = (const uint4*)(addr0 - T.shift);
    T.n_vec = (T.shift + T.n_r * len + 15u) >> 4;
    return T;
  };
  uint4 nx[PX_VEC_ROUNDS];
  auto load = [&](const Tile& T) {
#pragma unroll
    for (uint32_t r = 0; r < PX_VEC_ROUNDS; ++r) {
      const uint32_t i = r * 64u + lane;
      nx[r] = i < T.n_vec ? T.vsrc[i] : make_uint4(0, 0, 0, 0);
    }
  };
  auto pack = [&](const Tile& T) {
#pragma unroll
    for (uint32_t r = 0; r < PX_VEC_ROUNDS; ++r) {
      const uint32_t i = r * 64u + lane;
      if (i < codes_dw) {
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
        codes[i] = i < T.n_vec ? p : 0u;
      }
    }
  };
  auto fence = [&]() { // (everything a wave touches here is its own)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // the 2-bit codes of W consecutive positions from position n of the slab on, W <= 32: a 64-bit piece of the code stream
  auto codes_at = [&](uint32_t n) -> uint64_t {
    const uint32_t d = n >> 4, sh = (n & 15u) * 2u;
    const uint32_t w0 = codes[d], w1 = codes[d + 1u], w2 = codes[d + 2u];
    const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh), hi = __builtin_amdgcn_alignbit(w2, w1, sh);
    return ((uint64_t)hi << 32) | lo;
  };

  // build lanes: segment sb of read rb of a round; window lanes: segment sw of read rl of the tile
  const uint32_t sb = lane & (NB - 1u), rb_in_round = lane >> a.nb_log, reads_per_round = 64u >> a.nb_log;
  const uint32_t lpr = 1u << a.lpr_log, sw = lane & (lpr - 1u), rl = lane >> a.lpr_log;
  const bool sb_in = sb < a.segs_b;
  const uint32_t win0 = sw * W; // the lane's first window in its read
  const uint32_t n_steps_w = win0 < nwin ? (nwin - win0 < W ? nwin - win0 : W) : 0u;
  // the rotations of the lane's first window: F by (p + k - 1), R by -p
  const uint32_t p31_0 = win0 % 31u, p33_0 = win0 % 33u;

  const uint64_t t_step = (uint64_t)gridDim.x * a.waves;
  uint64_t t = (uint64_t)blockIdx.x * a.waves + wave;
  Tile T;
  if (t < a.n_tiles) {
    T = place(t);
    load(T);
  }
  for (; t < a.n_tiles; t += t_step) {
    pack(T); // (the codes of this tile; nx is free for the next one's bytes)
    const bool more = t + t_step < a.n_tiles;
    Tile Tn;
    if (more) {
      Tn = place(t + t_step);
      load(Tn);
    }
    fence();

    // ---- the arrays of the tile's reads, reads_per_round of them at a time ----
    for (uint32_t r0 = 0; r0 < T.n_r; r0 += reads_per_round) {
      const uint32_t rb = r0 + rb_in_round;
      const bool in = sb_in && rb < T.n_r;
      const uint64_t cs = in ? codes_at(T.shift + rb * len + sb * W) : 0ull;
      const char* const tsrc = (const char*)ttab + sb * 16u;
      // pass A: the segment's total
      uint4 tot = make_uint4(0, 0, 0, 0);
      for (uint32_t i = 0; i < W; ++i) {
        const uint32_t c = (uint32_t)(cs >> (2u * i)) & 3u;
        tot = tot ^ *(const uint4*)(tsrc + ((c * epr + (i << a.nb_log)) << 4));
      }
      if (!in) tot = make_uint4(0, 0, 0, 0);
      uint4 run = make_uint4(ps_xor_scan(tot.x, a.nb_log) ^ tot.x, ps_xor_scan(tot.y, a.nb_log) ^ tot.y,
                             ps_xor_scan(tot.z, a.nb_log) ^ tot.z, ps_xor_scan(tot.w, a.nb_log) ^ tot.w);
      // pass B: the prefix in front of every position (and the terms themselves)
      if (in) {
        char* const dst = arrays + (rb * epr + sb) * 16u;
        for (uint32_t i = 0; i < W; ++i) {
          const uint32_t c = (uint32_t)(cs >> (2u * i)) & 3u;
          const uint4 tv = *(const uint4*)(tsrc + ((c * epr + (i << a.nb_log)) << 4));
          *(uint4*)(dst + ((i << a.nb_log) << 4)) = run;
          if (a.n_arrays > 1u) *(uint4*)(dst + arr_bytes + ((i << a.nb_log) << 4)) = tv;
          run = run ^ tv;
        }
      }
    }
    fence();

    // ---- the windows: W steps, the lane's window p = win0 + i of read rl ----
    {
      const bool lane_in = rl < T.n_r;
      const uint32_t steps = lane_in ? n_steps_w : 0u;
      // (a lane without a window reads what its place implies all the same -- its values are dropped; sent to one address
      //  instead, the idle lanes shared a bank with a busy lane of their 16-lane group: 16 % of the LDS cycles.  Their reads
      //  may end up to 64 + k / W entries behind the wave's arrays: the launch leaves that room behind the last wave)
      const char* const ent = arrays + (rl * epr + sw) * 16u;
      uint64_t* out = a.hashes + ((T.read0 + rl) * nwin + win0) * per;
      uint32_t p31 = p31_0, p33 = p33_0;
      // (the constant address space: scalar loads whatever the kernel has stored before)
      const __attribute__((address_space(4))) uint32_t* so = (const __attribute__((address_space(4))) uint32_t*)a.step_off;
      for (uint32_t i = 0; i < W; ++i, so += a.n_terms, out += per) {
        const uint32_t f31 = px_wrap(p31 + a.k31, 31u), f33 = px_wrap(p33 + a.k33, 33u);
        const uint32_t b31 = p31 ? 31u - p31 : 0u, b33 = p33 ? 33u - p33 : 0u;
        uint64_t own[PERC ? PERC : 1];
        for (uint32_t s = 0; s < a.n_seeds; ++s) {
          uint4 acc = make_uint4(0, 0, 0, 0);
          uint32_t ti = a.seed_first[s];
          const uint32_t te = a.seed_first[s + 1];
          for (; ti + 4u <= te; ti += 4u) {
            const uint4 v0 = *(const uint4*)(ent + so[ti]), v1 = *(const uint4*)(ent + so[ti + 1u]);
            const uint4 v2 = *(const uint4*)(ent + so[ti + 2u]), v3 = *(const uint4*)(ent + so[ti + 3u]);
            acc.x = __builtin_amdgcn_bitop3_b32(acc.x, v0.x, v1.x, 0x96) ^ __builtin_amdgcn_bitop3_b32(v2.x, v3.x, 0u, 0x96);
            acc.y = __builtin_amdgcn_bitop3_b32(acc.y, v0.y, v1.y, 0x96) ^ __builtin_amdgcn_bitop3_b32(v2.y, v3.y, 0u, 0x96);
            acc.z = __builtin_amdgcn_bitop3_b32(acc.z, v0.z, v1.z, 0x96) ^ __builtin_amdgcn_bitop3_b32(v2.z, v3.z, 0u, 0x96);
            acc.w = __builtin_amdgcn_bitop3_b32(acc.w, v0.w, v1.w, 0x96) ^ __builtin_amdgcn_bitop3_b32(v2.w, v3.w, 0u, 0x96);
          }
          for (; ti < te; ++ti) acc = acc ^ *(const uint4*)(ent + so[ti]);
          px_srol_var(acc.x, acc.y, f31, f33);
          px_srol_var(acc.z, acc.w, b31, b33);
          const uint64_t h0 = canon_pair(acc.x, acc.y, acc.z, acc.w);
          if (PERC == 1) own[0] = h0;
          else if (PERC == 2) {
            if (m2 == 2u) {
              own[0] = h0;
              own[1] = mix_hash(h0, a.mult[1]);
            } else own[s & 1u] = h0;
          } else if (i < steps) {
            out[s * m2] = h0;
            for (uint32_t jj = 1; jj < m2; ++jj) out[s * m2 + jj] = mix_hash(h0, a.mult[jj & (SF_MAX_RUNTIME_M - 1)]);
          }
        }
        if (PERC == 1) {
          if (i < steps) out[0] = own[0];
        } else if (PERC == 2) {
          if (i < steps) *(uint4*)out = make_uint4((uint32_t)own[0], (uint32_t)(own[0] >> 32), (uint32_t)own[1], (uint32_t)(own[1] >> 32));
        }
        p31 = px_wrap(p31 + 1u, 31u);
        p33 = px_wrap(p33 + 1u, 33u);
      }
    }
    fence();
    T = Tn;
  }
  if (__ballot(bad != 0) != 0 && lane == 0) atomicOr(a.dirty, 1u);
}

} // namespace ntamd
