// nt_math.hpp -- the ntHash v2 arithmetic, shared by the HIP kernels and the
// host side of nthash_amd (C-ABI set-up code, host facade classes).
//
// Written from the mathematical specification of the hash,
// not from the reference's tables: the reference materialises srol^d(seed) in
// 33+31-entry lookup tables per base (src/internal.hpp:167-348); here it is the
// closed-form split rotate, so the device code needs no tables in memory beyond
// the 16-entry (in,out) pair table that each kernel builds in LDS.
//
// Base encoding used throughout nthash_amd: code = (ascii >> 1) & 3, i.e.
//   A/a = 0, C/c = 1, T/t/U/u = 2, G/g = 3,   complement(code) = code ^ 2.
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define NT_HD __host__ __device__ __forceinline__
#else
#define NT_HD inline
#endif

namespace ntamd {

// reference: src/internal.hpp:124-128 (the four 64-bit base seeds)
constexpr uint64_t SEED_A = 0x3c8bfbb395c60474ULL;
constexpr uint64_t SEED_C = 0x3193c18562a02b4cULL;
constexpr uint64_t SEED_G = 0x20323ed082572324ULL;
constexpr uint64_t SEED_T = 0x295549f54be24456ULL;
// reference: src/internal.hpp:91,94
constexpr unsigned MULTISHIFT = 27;
constexpr uint64_t MULTISEED = 0x90b45d39fb6da1faULL;

constexpr uint64_t MASK31 = 0x7FFFFFFFULL;
constexpr uint64_t MASK33 = 0x1FFFFFFFFULL;

// seed of a 2-bit code (A,C,T,G order of the (c>>1)&3 encoding)
NT_HD uint64_t seed_of_code(unsigned code)
{
  return code == 0 ? SEED_A : code == 1 ? SEED_C : code == 2 ? SEED_T : SEED_G;
}

// one split-rotate-left step: bits 63..33 rotate as a 31-bit word, bits 32..0
// as a 33-bit word (reference: srol, src/internal.hpp:41-47)
NT_HD uint64_t srol1(uint64_t x)
{
  const uint64_t carry = ((x >> 63) << 33) | ((x >> 32) & 1ULL);
  return ((x << 1) & ~(1ULL << 33)) | carry;
}

// inverse step (reference: sror, src/internal.hpp:83-88)
NT_HD uint64_t sror1(uint64_t x)
{
  const uint64_t carry = ((x & 1ULL) << 32) | (((x >> 33) & 1ULL) << 63);
  return ((x >> 1) & ~(1ULL << 32)) | carry;
}

// d-fold split rotate, any d (reference: srol_table, src/internal.hpp:343-348,
// which looks srol^d(seed) up with d % 31 and d % 33)
NT_HD uint64_t srol_n(uint64_t x, unsigned d)
{
  const unsigned a = d % 31u, b = d % 33u;
  uint64_t hi = x >> 33, lo = x & MASK33;
  if (a) hi = ((hi << a) | (hi >> (31u - a))) & MASK31;
  if (b) lo = ((lo << b) | (lo >> (33u - b))) & MASK33;
  return (hi << 33) | lo;
}

// ASCII classification (reference: SEED_TAB != SEED_N, src/internal.hpp:132-165).
// Valid bases are exactly ACGTU in either case.
NT_HD bool is_base(unsigned char c)
{
  const unsigned l = c | 0x20u;
  return l == 'a' || l == 'c' || l == 'g' || l == 't' || l == 'u';
}

NT_HD unsigned code_of(unsigned char c) { return (c >> 1) & 3u; }

// forward-strand seed of a raw byte: 0 for anything that is not a base
NT_HD uint64_t fwd_seed(unsigned char c) { return is_base(c) ? seed_of_code(code_of(c)) : 0; }

// reverse-strand seed of a raw byte.  The reference indexes its seed table
// with (c & 7) (src/internal.hpp:121, src/seed.cpp:156,163): for bases that is
// the complement's seed; for other bytes it is whatever sits in slots 0..7,
// which SeedNtHash does hash (it does not skip invalid bases inside a freshly
// initialised window -- SURVEY.md App. B Q3).  Slots: 1->T 3->G 4->A 5->A 7->C.
NT_HD uint64_t rc_seed(unsigned char c)
{
  switch (c & 7u) {
    case 1: return SEED_T;
    case 3: return SEED_G;
    case 4: return SEED_A;
    case 5: return SEED_A;
    case 7: return SEED_C;
    default: return 0;
  }
}

// multi-hash expansion (reference: extend_hashes, src/internal.hpp:104-118):
// h[0] = fwd + rev; h[i] = mix(h[0] * (i ^ k*MULTISEED))
NT_HD uint64_t mix_hash(uint64_t h0, uint64_t mult)
{
  uint64_t t = h0 * mult;
  return t ^ (t >> MULTISHIFT);
}
NT_HD uint64_t multiplier(unsigned k, unsigned i) { return (uint64_t)i ^ ((uint64_t)k * MULTISEED); }

// Direct (non-rolling) strand hashes of one window
//   F = XOR_i srol^{k-1-i}(S[s_i]),  R = XOR_i srol^{i}(S[comp s_i])
// (what base_forward_hash/base_reverse_hash compute, src/kmer.cpp:43-73,123-152)
NT_HD uint64_t direct_fwd(const char* s, unsigned k)
{
  uint64_t h = 0;
  for (unsigned i = 0; i < k; i++) h = srol1(h) ^ fwd_seed((unsigned char)s[i]);
  return h;
}
NT_HD uint64_t direct_rev(const char* s, unsigned k)
{
  uint64_t h = 0;
  for (unsigned i = k; i-- > 0;) h = srol1(h) ^ rc_seed((unsigned char)s[i]);
  return h;
}

} // namespace ntamd
