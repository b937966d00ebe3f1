// first_window.hpp -- the first window of a run for ANY k, two ways, from ONE k-independent table set.
//
// The run-split kernels roll C-1 of a run's C windows (next_forward_hash / next_reverse_hash, src/kmer.cpp:84-94,
// 164-174) but must hash its FIRST window directly: what base_forward_hash / base_reverse_hash do, src/kmer.cpp:43-73,
// 123-152.  For k <= 64 that is ceil(k/4) lookups in position-specific byte tables (kmer_runs_kernel.hpp).  Beyond,
// the tables would outgrow LDS; round 2 walked the window 4 bases at a time (2 x k/4 dependent steps per run: k = 100
// ran at 0.35 of the roofline, k = 500 at 0.13).  Here (round 3):
//
//   grouped   16 bases per step: the window's j-th 16-base group is ONE funnel-shifted word of the 2-bit stream, its
//             16-mer hash four lookups (both strands in an entry), and the groups are chained with a CONSTANT split
//             rotate by 16 -- k/16 steps instead of 2 x k/4, every lookup of a step independent of the hash state.
//   scan      k-independent: a prefix over the wave's whole slab, then every window is a difference of two prefixes
//             (SURVEY.md App. A.3).  In the frame of absolute stream positions the prefix needs no rotation at all:
//                 U(n) = XOR_{j<n} sror^{j+1}(S[c_j])          F(p) = srol^{p+k}( U(p+k) ^ U(p) )
//                 V(n) = XOR_{j<n} srol^{j}  (S[~c_j])          R(p) = sror^{p}  ( V(p+k) ^ V(p) )
//             (substitute i = j - p: srol^{p+k-j-1} = srol^{k-1-i}, the rotation base i of a window has in
//             src/kmer.cpp:43-73; likewise srol^{j-p} = srol^{i} for the reverse strand).  U and V are plain XOR
//             prefix sums, so the wave-level scan is six DPP XORs per register; a lane's words enter the absolute
//             frame with one variable split rotate each, and a window costs six such rotates whatever k is.
//
// Tables ("fw tables", FW_ENTRIES uint4 {f.lo, f.hi, r.lo, r.hi}):
//   TW[q][byte]  q = 0..3: the byte's four bases at positions u = 4q..4q+3 of a 16-base word, in the word's own
//                absolute frame: f = XOR sror^{u+1}(S[c]), r = XOR srol^{u}(S[~c])
//   AC[r]        r = 0..15: what the positions u >= r of a word contribute when they hold code 0 ('A') -- XOR-ed onto
//                the lookups of a word MASKED to its first r bases it leaves exactly those r bases: a partial word
//                costs the same four unconditional lookups
//   MOD[y]       y = 0..1022: (y mod 31) | (y mod 33) << 8 (31 * 33 = 1023 and 1024 = 1 mod 1023: x mod 1023 is a fold)
//
// Everything below is plain integer C++ over {lo, hi} register pairs; it compiles for the host too (tests/host/
// first_window_host.cpp runs both forms against nt_math.hpp's direct hashes on the CPU, 64 "lanes" in a loop).
#pragma once

#include <cstdint>

#include "nt_math.hpp"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FW_HD __host__ __device__ __forceinline__
typedef uint4 fw_u4;
#else
#define FW_HD inline
struct fw_u4 {
  uint32_t x, y, z, w;
};
#endif

namespace ntamd {

constexpr uint32_t FW_AC = 1024;      // first AC entry
constexpr uint32_t FW_MOD = 1040;     // first MOD entry (1023 x u16 = 128 entries)
constexpr uint32_t FW_ENTRIES = 1280; // five 4 KiB tables' worth (the kernels copy whole tables)

FW_HD fw_u4 fw_make(uint32_t x, uint32_t y, uint32_t z, uint32_t w)
{
  fw_u4 v;
  v.x = x; v.y = y; v.z = z; v.w = w;
  return v;
}
FW_HD uint32_t fw_xor3(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
  return a ^ b ^ c;
#endif
}
FW_HD uint32_t fw_funnel(uint32_t hi, uint32_t lo, uint32_t sh) // ({hi,lo} >> (sh & 31))[31:0]
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
  return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (sh & 31u));
#endif
}

// split rotate left / right of the {lo, hi} pair by (a, b): a in [0, 31) for bits 63..33, b in [0, 33) for bits 32..0
// (reference: srol(x, d), src/internal.hpp:57-68, with a = d % 31, b = d % 33; any amounts, also per lane)
FW_HD void srol_var(uint32_t& lo, uint32_t& hi, uint32_t a, uint32_t b)
{
  const uint32_t H = hi >> 1;
  const uint64_t L = (uint64_t)lo | ((uint64_t)(hi & 1u) << 32);
  const uint32_t Hn = ((H << a) | (H >> (31u - a))) & 0x7FFFFFFFu;
  const uint64_t Ln = ((L << b) | (L >> (33u - b))) & MASK33;
  lo = (uint32_t)Ln;
  hi = (Hn << 1) | (uint32_t)(Ln >> 32);
}
FW_HD void sror_var(uint32_t& lo, uint32_t& hi, uint32_t a, uint32_t b)
{
  const uint32_t H = hi >> 1;
  const uint64_t L = (uint64_t)lo | ((uint64_t)(hi & 1u) << 32);
  const uint32_t Hn = ((H >> a) | (H << (31u - a))) & 0x7FFFFFFFu;
  const uint64_t Ln = ((L >> b) | (L << (33u - b))) & MASK33;
  lo = (uint32_t)Ln;
  hi = (Hn << 1) | (uint32_t)(Ln >> 32);
}

// x mod 1023 for x < 2^20 (1024 = 1 mod 1023: add the 10-bit digits)
FW_HD uint32_t fw_fold1023(uint32_t x)
{
  uint32_t y = (x & 1023u) + (x >> 10);
  y = y >= 1023u ? y - 1023u : y;
  return y;
}
// (x mod 31, x mod 33) of a position y < 1023 from the MOD table
FW_HD void fw_amounts(const fw_u4* tab, uint32_t y, uint32_t& a, uint32_t& b)
{
  const uint16_t e = ((const uint16_t*)(tab + FW_MOD))[y];
  a = e & 0xFFu;
  b = e >> 8;
}

// the 16 bases of a word, word-local absolute frame: {XOR_u sror^{u+1}(S[c_u]), XOR_u srol^{u}(S[~c_u])}
FW_HD fw_u4 fw_word16(const fw_u4* tab, uint32_t word)
{
  const fw_u4 e0 = tab[word & 0xFFu], e1 = tab[256u + ((word >> 8) & 0xFFu)], e2 = tab[512u + ((word >> 16) & 0xFFu)],
              e3 = tab[768u + (word >> 24)];
  return fw_make(fw_xor3(e0.x, e1.x, e2.x) ^ e3.x, fw_xor3(e0.y, e1.y, e2.y) ^ e3.y, fw_xor3(e0.z, e1.z, e2.z) ^ e3.z,
                 fw_xor3(e0.w, e1.w, e2.w) ^ e3.w);
}
// its first r bases only (r = 0..15): same four lookups on the masked word, plus what the masked positions contributed
FW_HD fw_u4 fw_partial(const fw_u4* tab, uint32_t word, uint32_t r)
{
  const fw_u4 p = fw_word16(tab, word & ((1u << (2u * r)) - 1u));
  const fw_u4 ac = tab[FW_AC + r];
  return fw_make(p.x ^ ac.x, p.y ^ ac.y, p.z ^ ac.z, p.w ^ ac.w);
}

// 16 bases of a 2-bit stream (16 per dword) from any position
FW_HD uint32_t fw_word_at(const uint32_t* bits, uint32_t pos)
{
  return fw_funnel(bits[(pos >> 4) + 1u], bits[pos >> 4], (pos & 15u) << 1);
}

// ---- grouped: F = srol^{k}( XOR_g sror^{16 g}(f_g) ), R = XOR_g srol^{16 g}(r_g), g-th group = bases [b0 + 16 g, ...) of
// the window, the last one masked to k % 16 bases; Horner from the last group down, constant rotates by 16.  The
// stream must be readable up to two dwords past the window's last base.
FW_HD void grouped_first_window(const uint32_t* bits, const fw_u4* tab, uint32_t b0, uint32_t k, uint32_t k31,
                                uint32_t k33, uint32_t& f_lo, uint32_t& f_hi, uint32_t& r_lo, uint32_t& r_hi)
{
  const uint32_t G = k >> 4, rem = k & 15u;
  fw_u4 acc = fw_partial(tab, fw_word_at(bits, b0 + 16u * G), rem);
  for (uint32_t g = G; g-- > 0;) {
    const fw_u4 e = fw_word16(tab, fw_word_at(bits, b0 + 16u * g));
    sror_var(acc.x, acc.y, 16u, 16u);
    srol_var(acc.z, acc.w, 16u, 16u);
    acc.x ^= e.x; acc.y ^= e.y; acc.z ^= e.z; acc.w ^= e.w;
  }
  srol_var(acc.x, acc.y, k31, k33);
  f_lo = acc.x; f_hi = acc.y; r_lo = acc.z; r_hi = acc.w;
}

// ---- scan, step 1: what word wi of the stream adds to the prefixes {U, V}, in the absolute frame (x = 16 wi) ----
FW_HD fw_u4 fw_scan_word(const fw_u4* tab, uint32_t word, uint32_t wi)
{
  fw_u4 e = fw_word16(tab, word);
  uint32_t a, b;
  fw_amounts(tab, fw_fold1023(16u * wi), a, b);
  sror_var(e.x, e.y, a, b); // sror^{x}: word-local u + 1 -> absolute x + u + 1
  srol_var(e.z, e.w, a, b);
  return e;
}
// ---- scan, step 2: uw[w] = {U(16 w), V(16 w)} for every word of the slab is in LDS; the window [p, p + k) ----
// k1023 = k % 1023, k31 = k % 31, k33 = k % 33
FW_HD void scan_first_window(const uint32_t* bits, const fw_u4* tab, const fw_u4* uw, uint32_t p, uint32_t k,
                             uint32_t k1023, uint32_t k31, uint32_t k33, uint32_t& f_lo, uint32_t& f_hi, uint32_t& r_lo,
                             uint32_t& r_hi)
{
  const uint32_t xa = p + k, wa = xa >> 4, ra = xa & 15u, wb = p >> 4, rb = p & 15u;
  const fw_u4 ua = uw[wa], ub = uw[wb];
  fw_u4 pa = fw_partial(tab, bits[wa], ra), pb = fw_partial(tab, bits[wb], rb);
  // F = srol^{p+k}(U_a ^ U_b) ^ srol^{ra}(pa.f) ^ srol^{k+rb}(pb.f)
  // R = sror^{p}  (V_a ^ V_b) ^ srol^{k-ra}(pa.r) ^ sror^{rb}(pb.r)
  uint32_t fl = ua.x ^ ub.x, fh = ua.y ^ ub.y, rl = ua.z ^ ub.z, rh = ua.w ^ ub.w;
  uint32_t a, b;
  const uint32_t yp = fw_fold1023(p);
  uint32_t ypk = yp + k1023;
  ypk = ypk >= 1023u ? ypk - 1023u : ypk;
  fw_amounts(tab, ypk, a, b);
  srol_var(fl, fh, a, b);
  fw_amounts(tab, yp, a, b);
  sror_var(rl, rh, a, b);
  srol_var(pa.x, pa.y, ra, ra);
  sror_var(pb.z, pb.w, rb, rb);
  uint32_t a2 = k31 + rb, b2 = k33 + rb;
  a2 = a2 >= 31u ? a2 - 31u : a2;
  b2 = b2 >= 33u ? b2 - 33u : b2;
  srol_var(pb.x, pb.y, a2, b2);
  uint32_t a3 = k31 + 31u - ra, b3 = k33 + 33u - ra;
  a3 = a3 >= 31u ? a3 - 31u : a3;
  b3 = b3 >= 33u ? b3 - 33u : b3;
  srol_var(pa.z, pa.w, a3, b3);
  f_lo = fw_xor3(fl, pa.x, pb.x);
  f_hi = fw_xor3(fh, pa.y, pb.y);
  r_lo = fw_xor3(rl, pa.z, pb.z);
  r_hi = fw_xor3(rh, pa.w, pb.w);
}

// host: the table block (FW_ENTRIES entries; zero past the MOD table)
inline void build_fw_tables(fw_u4* out)
{
  for (uint32_t i = 0; i < FW_ENTRIES; ++i) out[i] = fw_make(0, 0, 0, 0);
  auto phi = [](uint32_t u, uint32_t c) { return srol_n(seed_of_code(c), 1023u - (u + 1u)); }; // sror^{u+1}
  auto rho = [](uint32_t u, uint32_t c) { return srol_n(seed_of_code(c ^ 2u), u); };
  for (uint32_t q = 0; q < 4; ++q)
    for (uint32_t byte = 0; byte < 256; ++byte) {
      uint64_t f = 0, r = 0;
      for (uint32_t v = 0; v < 4; ++v) {
        const uint32_t c = (byte >> (2 * v)) & 3u;
        f ^= phi(4 * q + v, c);
        r ^= rho(4 * q + v, c);
      }
      out[q * 256 + byte] = fw_make((uint32_t)f, (uint32_t)(f >> 32), (uint32_t)r, (uint32_t)(r >> 32));
    }
  for (uint32_t rr = 0; rr < 16; ++rr) {
    uint64_t f = 0, r = 0;
    for (uint32_t u = rr; u < 16; ++u) {
      f ^= phi(u, 0);
      r ^= rho(u, 0);
    }
    out[FW_AC + rr] = fw_make((uint32_t)f, (uint32_t)(f >> 32), (uint32_t)r, (uint32_t)(r >> 32));
  }
  uint16_t* mod = (uint16_t*)(out + FW_MOD);
  for (uint32_t y = 0; y < 1023; ++y) mod[y] = (uint16_t)((y % 31u) | ((y % 33u) << 8));
}

} // namespace ntamd
