// seed_parse.hpp -- host-side spaced-seed parsing shared by the C-ABI
// (nthip_seeds_create) and the C++ facade (SeedNtHash / BlindSeedNtHash).
//
// Restates get_blocks() (reference src/seed.cpp:19-66): a seed is described
// either by its care runs or -- when that needs fewer table terms -- by its
// don't-care runs plus the whole-k-mer block, pushed LAST.  Runs of length one
// are "monomers".  The description matters beyond the mask it encodes because
// SeedNtHash::init scans exactly the block positions, in this order, for NUL
// bytes (src/seed.cpp:146-158).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace ntamd {

struct SeedShape {
  std::vector<uint32_t> block_pairs; // [start,end) pairs, in the reference's order
  std::vector<uint32_t> monomers;
  std::vector<uint8_t> care;         // care[p] = 1 iff position p contributes to the hash
  std::vector<uint8_t> blk_parity;   // parity of the block coverage of position p
  std::vector<uint8_t> is_mono;      // position p is a monomer (care = blk_parity ^ is_mono)
};

inline SeedShape parse_seed_shape(const std::string& s)
{
  SeedShape out;
  const uint32_t k = (uint32_t)s.size();
  std::vector<uint32_t> cb, ib, cm, im;
  // a care run ends at a literal '0', a don't-care run at a literal '1'; a
  // sentinel that differs from the last character closes the final run
  const char sentinel = s[k - 1] == '1' ? '0' : '1';
  bool care = s[0] == '1';
  uint32_t start = 0;
  for (uint32_t p = 0; p <= k; ++p) {
    const char ch = p < k ? s[p] : sentinel;
    if (care && ch == '0') {
      if (p - start == 1) cm.push_back(start);
      else { cb.push_back(start); cb.push_back(p); }
      start = p;
      care = false;
    } else if (!care && ch == '1') {
      if (p - start == 1) im.push_back(start);
      else { ib.push_back(start); ib.push_back(p); }
      start = p;
      care = true;
    }
  }
  const size_t cost_care = cb.size() + cm.size(); // 2 per block + 1 per monomer
  const size_t cost_ign = ib.size() + im.size() + 2;
  if (cost_ign < cost_care) {
    out.block_pairs = ib;
    out.block_pairs.push_back(0);
    out.block_pairs.push_back(k);
    out.monomers = im;
  } else {
    out.block_pairs = cb;
    out.monomers = cm;
  }
  // contributing positions = XOR-coverage of blocks and monomers (src/seed.cpp:149-164):
  // with the don't-care description those positions are covered twice and cancel
  out.blk_parity.assign(k, 0);
  out.is_mono.assign(k, 0);
  for (size_t b = 0; b + 1 < out.block_pairs.size(); b += 2)
    for (uint32_t p = out.block_pairs[b]; p < out.block_pairs[b + 1]; ++p) out.blk_parity[p] ^= 1;
  for (uint32_t p : out.monomers) out.is_mono[p] ^= 1;
  out.care.assign(k, 0);
  for (uint32_t p = 0; p < k; ++p) out.care[p] = out.blk_parity[p] ^ out.is_mono[p];
  return out;
}

inline bool seed_is_symmetric(const std::string& s)
{
  return std::equal(s.begin(), s.end(), s.rbegin());
}

} // namespace ntamd
