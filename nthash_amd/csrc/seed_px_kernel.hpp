// seed_px_kernel.hpp -- spaced seeds as sparse sums over scanned term arrays (round 6).
//
// The dense spaced-seed kernels (seed_kernels.hpp) hash a window from scratch, ceil(k / 4) random 16-byte LDS lookups per
// seed and window whatever the seed looks like; seed_roll_kernel.hpp rolls run by run but pays a first window per segment
// and a chain from window to window.  Here the cost follows the seed (seed_px_plan.hpp has the algebra): the per-position
// terms of a tile, rotated into ONE common frame,
//     T(q) = srol^{-q}(S[c_q])      U(q) = srol^{q}(S[comp c_q])      (q: position in the tile's slab; one 16-byte entry)
// are scanned once per tile into LDS arrays Y with T = B * Y (B = 1: the terms, 1 + x: the exclusive prefix XOR,
// (1 + x^d), (1 + x)(1 + x^d): stride-d scans), and a window's pair of strand hashes is
//     F = srol^{q+k-1}( XOR_{e in supp(C B)} Y(q + e) ).lo64      R = srol^{-q}( ... ).hi64
// -- for the prefix 2 reads per care run (what the reference's roll pays, src/seed.cpp:177-207), for a seed that repeats
// under a shift by d a handful -- with no table of k, no first window and no dependence between windows.  A lane is a
// window, the 64 lanes of a read instruction read 64 consecutive entries: conflict-free whatever the seed and the bases.
//
// A wave's tile is R whole reads (one contiguous slab of R * len bytes, no position built twice); its windows are taken
// in groups of 64 whose values are one contiguous piece of the output stream, collected in a wave-private stage and
// written 16 bytes per lane on whole 128-byte lines (the first group of a tile is cut so that every later one starts on a
// line).  The next tile's bytes are loaded while this one is hashed.  A non-base sets a.dirty (SeedNtHash's position
// state machine, App. B Q3, is the other kernels' business).
#pragma once

#include <hip/hip_runtime.h>

#include "seed_kernels.hpp"
#include "seed_px_plan.hpp"
#include "seed_roll_kernel.hpp" // (sr_phys: the stage's swizzle)

namespace ntamd {

constexpr uint32_t PX_MAX_WAVES = 16;
constexpr uint32_t PX_VEC_ROUNDS = 2; // a tile's bytes: at most 2 x 64 vectors of 16

struct SeedPxArgs {
  const uint8_t* seqs;
  uint64_t* hashes; // dense [read][window][seed][m2]
  uint32_t* dirty;
  uint64_t n_reads, n_tiles;
  uint64_t total_bytes; // n_reads * len
  uint32_t len, k, m2, n_seeds, nwin, inv_nwin;
  uint32_t R;          // reads per tile
  uint32_t n_entries;  // entries per array (a multiple of 128)
  uint32_t n_arrays;   // the first n_pre of them start from the exclusive prefix XOR, the others from the terms themselves
  uint32_t n_pre;
  uint32_t waves, stage_vals;
  uint32_t align_win; // 16 / gcd(values per window, 16): a group of windows that starts on a multiple of it starts on a line
  uint32_t reach;     // entries behind a slab's last position that a window may read (the arrays' scans), + 1
  uint32_t k31, k33;  // (k - 1) % 31, (k - 1) % 33
  uint32_t arr_d[PX_MAX_ARRAYS]; // the stride-d scan on top of an array's start (0: none)
  uint32_t seed_first[PX_MAX_SEEDS + 1];
  uint32_t term_off[PX_MAX_TERMS]; // byte offset of the read from the window's own entry of array 0: arr * n_entries * 16 + e * 16
  uint64_t mult[SF_MAX_RUNTIME_M];
};

// srol^d of a (lo, hi) pair, a = d % 31 and b = d % 33 given (any lane its own)
__device__ __forceinline__ void px_srol_var(uint32_t& lo, uint32_t& hi, uint32_t a, uint32_t b)
{
  const uint32_t x = hi >> 1; // bits 63..33: a 31-bit word
  const uint32_t xr = ((x << a) | (x >> (31u - a))) & 0x7FFFFFFFu; // (a == 0: x >> 31 == 0)
  const uint64_t y = ((uint64_t)(hi & 1u) << 32) | lo; // bits 32..0: a 33-bit word
  const uint64_t yr = ((y << b) | (y >> (33u - b))) & MASK33; // (b == 0: y >> 33 == 0)
  lo = (uint32_t)yr;
  hi = (xr << 1) | (uint32_t)(yr >> 32);
}

__device__ __forceinline__ uint32_t px_xor_scan(uint32_t v) // inclusive, over the wave
{
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
  return v;
}
__device__ __forceinline__ uint4 px_xor_scan4(uint4 v) { return make_uint4(px_xor_scan(v.x), px_xor_scan(v.y), px_xor_scan(v.z), px_xor_scan(v.w)); }
__device__ __forceinline__ uint4 px_last_lane(uint4 v)
{
  return make_uint4((uint32_t)__builtin_amdgcn_readlane((int)v.x, 63), (uint32_t)__builtin_amdgcn_readlane((int)v.y, 63),
                    (uint32_t)__builtin_amdgcn_readlane((int)v.z, 63), (uint32_t)__builtin_amdgcn_readlane((int)v.w, 63));
}
__device__ __forceinline__ uint4 operator^(uint4 a, uint4 b) { return make_uint4(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w); }
__device__ __forceinline__ uint32_t px_wrap(uint32_t r, uint32_t m) { return r >= m ? r - m : r; }

// PERC: values per window known at compile time (1, 2: a lane stores its window's values itself, 8 / 16 contiguous bytes
// per lane) or 0 (any: through the stage)
template <int PERC>
__global__ __launch_bounds__(PX_MAX_WAVES * 64) void seed_px_kernel(const SeedPxArgs a)
{
  extern __shared__ __attribute__((aligned(256))) uint32_t lds_dyn[];
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // LDS: [S table: 4 x 16 B][per wave: arrays | stage | 2-bit codes]
  uint4* const stab = (uint4*)lds_dyn;
  const uint32_t codes_dw = (a.n_entries >> 4) + 4u;
  const uint32_t wave_bytes = a.n_arrays * a.n_entries * 16u + a.stage_vals * 8u + codes_dw * 4u;
  char* const wbase = (char*)(stab + 16) + (size_t)wave * wave_bytes;
  char* const arrays = wbase;
  uint64_t* const stage = (uint64_t*)(wbase + a.n_arrays * a.n_entries * 16u);
  uint32_t* const codes = (uint32_t*)(stage + a.stage_vals);
  if (tid < 4u) {
    const uint64_t f = seed_of_code(tid), r = seed_of_code(tid ^ 2u);
    stab[tid] = make_uint4((uint32_t)f, (uint32_t)(f >> 32), (uint32_t)r, (uint32_t)(r >> 32));
  }
  __syncthreads();

  const uint32_t m2 = a.m2, per = PERC ? (uint32_t)PERC : a.n_seeds * m2, nwin = a.nwin, len = a.len;
  const uint32_t arr_bytes = a.n_entries * 16u;
  uint32_t bad = 0;

  struct Tile {
    uint64_t read0;
    uint32_t n_r;    // reads (R but for the batch's last tile)
    uint32_t shift;  // foreign bytes in front of its first vector
    uint32_t n_vec;
    const uint4* vsrc;
  };
  auto place = [&](uint64_t t) -> Tile {
    Tile T;
    T.read0 = t * a.R;
    const uint64_t left = a.n_reads - T.read0;
    T.n_r = left < a.R ? (uint32_t)left : a.R;
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

    // ---- the arrays: entries 0 .. NQ + reach, two consecutive ones per lane and step ----
    const uint32_t NQ = T.shift + T.n_r * len;
    const uint32_t n_it = (NQ + a.reach + 127u) >> 7;
    {
      uint32_t q0 = 2u * lane;
      uint32_t r31 = q0 % 31u, r33 = q0 % 33u;
      uint4 carry = make_uint4(0, 0, 0, 0);
      uint4* dst = (uint4*)arrays + q0;
      for (uint32_t it = 0; it < n_it; ++it, q0 += 128u, dst += 128) {
        const uint32_t w = codes[q0 >> 4], sh = (q0 & 15u) * 2u;
        uint4 t0 = stab[(w >> sh) & 3u], t1 = stab[(w >> (sh + 2u)) & 3u];
        const uint32_t r31b = px_wrap(r31 + 1u, 31u), r33b = px_wrap(r33 + 1u, 33u);
        px_srol_var(t0.x, t0.y, r31 ? 31u - r31 : 0u, r33 ? 33u - r33 : 0u); // T: srol^{-q}
        px_srol_var(t0.z, t0.w, r31, r33);                                   // U: srol^{q}
        px_srol_var(t1.x, t1.y, r31b ? 31u - r31b : 0u, r33b ? 33u - r33b : 0u);
        px_srol_var(t1.z, t1.w, r31b, r33b);
        uint32_t y = 0;
        if (a.n_pre != 0u) { // (uniform)
          const uint4 tot = t0 ^ t1, inc = px_xor_scan4(tot);
          const uint4 ex = inc ^ tot ^ carry, ex1 = ex ^ t0;
          carry = carry ^ px_last_lane(inc);
          for (; y < a.n_pre; ++y) {
            uint4* const d = (uint4*)((char*)dst + y * arr_bytes);
            d[0] = ex;
            d[1] = ex1;
          }
        }
        for (; y < a.n_arrays; ++y) {
          uint4* const d = (uint4*)((char*)dst + y * arr_bytes);
          d[0] = t0;
          d[1] = t1;
        }
        r31 = px_wrap(r31 + 4u, 31u);  // 128 % 31
        r33 = px_wrap(r33 + 29u, 33u); // 128 % 33
      }
    }
    // stride-d scans, in place: class by class (q = r, r + d, r + 2 d, ...), 64 of a class per step
    for (uint32_t y = 0; y < a.n_arrays; ++y) {
      const uint32_t d = a.arr_d[y];
      if (d > 1u) {
        fence();
        const uint32_t n_e = n_it * 128u;
        uint4* const arr = (uint4*)(arrays + y * arr_bytes);
        for (uint32_t r = 0; r < d; ++r) {
          uint4 cy = make_uint4(0, 0, 0, 0);
          for (uint32_t qb = r; qb < n_e; qb += 64u * d) {
            const uint32_t q = qb + lane * d;
            const bool in = q < n_e;
            const uint4 x = in ? arr[q] : make_uint4(0, 0, 0, 0);
            const uint4 inc = px_xor_scan4(x);
            if (in) arr[q] = inc ^ x ^ cy;
            cy = cy ^ px_last_lane(inc);
          }
        }
      }
    }
    fence();

    // ---- the windows, 64 at a time ----
    const uint32_t n_w = T.n_r * nwin;
    const uint64_t gw0 = T.read0 * nwin; // the tile's first window in the stream
    const uint32_t n_first = 64u - ((uint32_t)gw0 & (a.align_win - 1u)); // (align_win: a power of two <= 16)
    for (uint32_t w0 = 0; w0 < n_w;) {
      const uint32_t cnt_full = w0 == 0 ? n_first : 64u;
      const uint32_t cnt = n_w - w0 < cnt_full ? n_w - w0 : cnt_full;
      const bool active = lane < cnt;
      const uint32_t w = active ? w0 + lane : w0;
      const uint32_t rr = nwin == 1u ? w : __umulhi(w, a.inv_nwin);
      const uint32_t q = T.shift + rr * len + (w - rr * nwin);
      const char* const ent = arrays + q * 16u;
      const uint32_t q31 = q % 31u, q33 = q % 33u;
      const uint32_t f31 = px_wrap(q31 + a.k31, 31u), f33 = px_wrap(q33 + a.k33, 33u);
      const uint32_t b31 = q31 ? 31u - q31 : 0u, b33 = q33 ? 33u - q33 : 0u;
      const uint64_t v_first = (gw0 + w0) * per;       // the group's first value in the stream
      const uint32_t off = (uint32_t)(v_first & 15u);  // and its place in its 128-byte line
      uint64_t own[PERC ? PERC : 1];
      for (uint32_t s = 0; s < a.n_seeds; ++s) {
        uint4 acc = make_uint4(0, 0, 0, 0);
        uint32_t ti = a.seed_first[s];
        const uint32_t te = a.seed_first[s + 1];
        for (; ti + 4u <= te; ti += 4u) {
          const uint4 v0 = *(const uint4*)(ent + a.term_off[ti]), v1 = *(const uint4*)(ent + a.term_off[ti + 1u]);
          const uint4 v2 = *(const uint4*)(ent + a.term_off[ti + 2u]), v3 = *(const uint4*)(ent + a.term_off[ti + 3u]);
          acc = acc ^ v0 ^ v1 ^ v2 ^ v3;
        }
        for (; ti < te; ++ti) acc = acc ^ *(const uint4*)(ent + a.term_off[ti]);
        px_srol_var(acc.x, acc.y, f31, f33);
        px_srol_var(acc.z, acc.w, b31, b33);
        const uint64_t h0 = canon_pair(acc.x, acc.y, acc.z, acc.w);
        if (PERC == 1) own[0] = h0;
        else if (PERC == 2) {
          if (m2 == 2u) {
            own[0] = h0;
            own[1] = mix_hash(h0, a.mult[1]);
          } else own[s & 1u] = h0;
        } else if (active) {
          const uint32_t vs = off + lane * per + s * m2;
          stage[sr_phys(vs)] = h0;
          for (uint32_t jj = 1; jj < m2; ++jj) stage[sr_phys(vs + jj)] = mix_hash(h0, a.mult[jj & (SF_MAX_RUNTIME_M - 1)]);
        }
      }
      if (PERC == 1) {
        if (active) a.hashes[v_first + lane] = own[0];
      } else if (PERC == 2) {
        if (active) *(uint4*)(a.hashes + v_first + 2u * lane) =
            make_uint4((uint32_t)own[0], (uint32_t)(own[0] >> 32), (uint32_t)own[1], (uint32_t)(own[1] >> 32));
      } else {
        fence();
        // the stage -> the stream: rows of 16 values = 128-byte lines of the stream, eight rows per instruction
        const uint32_t v_end = off + cnt * per; // (exclusive)
        const uint32_t n_rows = (v_end + 15u) >> 4;
        uint64_t* const dst0 = a.hashes + (v_first - off);
        const uint32_t c = lane & 7u;
        for (uint32_t r0 = 0; r0 < n_rows; r0 += 8u) {
          const uint32_t Rw = r0 + (lane >> 3), key = sr_row_key(Rw);
          const uint4 qv = *(const uint4*)(stage + Rw * 16u + 2u * (c ^ (key >> 1)));
          const uint4 dv = (key & 1u) ? make_uint4(qv.z, qv.w, qv.x, qv.y) : qv;
          const uint32_t v = Rw * 16u + 2u * c;
          const bool lo_ok = v >= off && v < v_end, hi_ok = v + 1u >= off && v + 1u < v_end;
          uint64_t* const dst = dst0 + v;
          if (lo_ok && hi_ok) *(uint4*)dst = dv;
          else if (lo_ok) *(uint2*)dst = make_uint2(dv.x, dv.y);
          else if (hi_ok) *(uint2*)(dst + 1) = make_uint2(dv.z, dv.w);
        }
        fence();
      }
      w0 += cnt;
    }
    T = Tn;
  }
  if (__ballot(bad != 0) != 0 && lane == 0) atomicOr(a.dirty, 1u);
}

} // namespace ntamd
