// bloom_math.hpp -- bit position of a hash value in a filter of n_bits bits (the Bloom consumers)
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace ntamd {

// h mod d for an invariant d (magic = floor((2^64 - 1) / d): the quotient estimate is at most 2 short)
__device__ __forceinline__ uint64_t mod_invariant(uint64_t h, uint64_t d, uint64_t magic)
{
  if (magic == 0) return h & (d - 1);
  uint64_t r = h - __umul64hi(h, magic) * d;
  if (r >= d) r -= d;
  if (r >= d) r -= d;
  return r;
}

} // namespace ntamd
