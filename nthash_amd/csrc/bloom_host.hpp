// bloom_host.hpp -- host-side helpers the binned Bloom / sketch consumers share (capi_sink_bloom.hip: insert;
// capi_sink_query.hip: query)
#pragma once

#include <cmath>

#include "capi_internal.hpp"
#include "bloom_binned_kernels.hpp"
#include "bloom_fused_kernels.hpp"

namespace ntamd {
namespace host {

inline uint64_t bloom_magic_of(uint64_t n_bits)
{
  return (n_bits & (n_bits - 1)) == 0 ? 0ull : ~0ull / n_bits;
}

struct BloomFusedSrc {
  const uint8_t* seqs = nullptr;
  uint64_t n_reads = 0;
  uint32_t len = 0, stride = 0, k = 0, m = 0;
};
constexpr uint32_t BF_COUNT_THREADS_BIG = 512; // filters of more than 2^34 slots' regions: 128 KiB of counters leave room for 512 reads
// LDS of a pass: the tile's bit stream + the counters / the sorted tile (the statics are the kernel's own)
inline size_t bloom_fused_lds(const BloomFusedSrc& s, uint32_t threads, uint32_t area_dwords)
{
  const uint32_t pad = (s.k + 15u) / 16u + 1u;
  const uint32_t bits = pad + (((threads - 1u) * s.stride + s.len + 30u) >> 4) + 2u;
  return ((size_t)((bits + 3u) & ~3u) + area_dwords) * 4;
}
inline bool bloom_fused_ok(const nthip_ctx* c, const BloomFusedSrc& s, uint32_t n_regions)
{
  if (c->tune.bloom_fused == 2 || s.m > (uint32_t)KF_MAX_RUNTIME_M || s.len < s.k || s.stride < s.len) return false;
  if (2u * (s.len - s.k + 1u) < s.len && c->tune.bloom_fused != 1) return false; // (more than two rolls per k-mer: the stream path)
  const size_t cap = lds_cap_of(c) - 4096; // (tab / hist / off / gbase are static)
  const uint32_t ct = n_regions > 16384u ? BF_COUNT_THREADS_BIG : 1024u;
  return bloom_fused_lds(s, ct, n_regions < 128u ? 128u : n_regions) <= cap && bloom_fused_lds(s, 1024u, 1024u * 16u) <= cap;
}
inline void bloom_fused_args(const BloomFusedSrc& s, uint32_t threads, uint64_t n_bits, uint64_t magic, BloomFusedArgs* a)
{
  memset(a, 0, sizeof *a);
  KmerFixedArgs consts;
  memset(&consts, 0, sizeof consts);
  fill_kmer_consts(s.k, s.m, consts);
  a->seqs = s.seqs;
  a->n_reads = s.n_reads;
  a->len = s.len;
  a->stride = s.stride;
  a->k = s.k;
  a->m = s.m;
  a->pad_dwords = (s.k + 15u) / 16u + 1u;
  a->n_tiles = (uint32_t)((s.n_reads + threads - 1) / threads);
  a->f_init = consts.f_init;
  a->r_init = consts.r_init;
  memcpy(a->tab, consts.tab, sizeof a->tab);
  memcpy(a->mult, consts.mult, sizeof a->mult);
  a->n_bits = n_bits;
  a->magic = magic;
}

// what a bucket of `bucket_slots` of the table's n_slots owns in a round of n values
inline uint64_t slot_cap(const nthip_ctx* c, uint64_t n, uint64_t bucket_slots, uint64_t n_slots)
{
  const double mean = (double)n * (double)(bucket_slots < n_slots ? bucket_slots : n_slots) / (double)n_slots;
  double cap = mean + 8.0 * std::sqrt(mean) + 256.0;
  if (c->tune.bloom_slot_tight == 1) cap = mean;       // (tests: a few values of every bucket take the overflow list)
  if (c->tune.bloom_slot_tight == 2) cap = mean * 0.5; // (tests: the overflow list overflows, the round fails)
  return ((uint64_t)cap + 64u) & ~(uint64_t)63u;
}


// ---- pieces mode (bloom_binned_kernels.hpp): the lists of a two-level round as block-private pieces ----------------------
constexpr uint32_t BB_PIECES_LDS_DWORDS = BB_MAX_BINS * 32u + 2u * BB_MAX_BINS; // level 1: the waiting entries + two counters per bucket
struct PiecesGeo {
  uint32_t g1 = 0, gx = 0;     // blocks of level 1 (a piece of every bin each); blocks per bin of level 2 (a piece of every region each)
  uint64_t cap1 = 0, cap2 = 0; // entries per piece
};
inline uint64_t piece_cap(const nthip_ctx* c, double mean)
{
  double cap = mean + 8.0 * std::sqrt(mean) + 256.0;
  if (c->tune.bloom_slot_tight == 1) cap = mean;       // (tests: the overflow list in use)
  if (c->tune.bloom_slot_tight == 2) cap = mean * 0.5; // (tests: the overflow list overflows, the round fails)
  return ((uint64_t)cap + 64u) & ~(uint64_t)63u;
}
// n_reads fixed-length reads of values_per_read values each into a table of n_slots slots, regions of 2^region_shift, bins of
// 128 regions; gx = blocks per bin the second level is launched with
inline void pieces_geo(const nthip_ctx* c, uint64_t n_reads, uint64_t values_per_read, uint64_t n_slots, uint32_t region_shift, uint32_t gx,
                       PiecesGeo* g)
{
  const uint64_t n_tiles = (n_reads + 1023) / 1024;
  g->g1 = (uint32_t)(n_tiles < (uint64_t)c->n_cu ? n_tiles : (uint64_t)c->n_cu);
  if (g->g1 == 0) g->g1 = 1;
  g->gx = gx;
  const uint64_t tiles_per_block = (n_tiles + g->g1 - 1) / g->g1;
  const double per_block = (double)tiles_per_block * 1024.0 * (double)values_per_read; // what a level-1 block may see
  const double bin_slots = (double)(1ull << (region_shift + 7u)), region_slots = (double)(1ull << region_shift);
  const double share1 = bin_slots < (double)n_slots ? bin_slots / (double)n_slots : 1.0;
  const double share2 = region_slots < (double)n_slots ? region_slots / (double)n_slots : 1.0;
  g->cap1 = piece_cap(c, per_block * share1);
  const uint32_t pieces_per_block = (g->g1 + gx - 1) / gx; // of a bin, per level-2 block
  g->cap2 = piece_cap(c, (double)pieces_per_block * per_block * share2);
}

} // namespace host
} // namespace ntamd
