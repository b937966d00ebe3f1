// capi_sink_seed.hip -- Bloom filter consumers of the SPACED-SEED hash stream: nthip_seed_bloom_insert / _query
// Part of libnthash_hip.so (include/nthash_hip.h); see capi_internal.hpp for the file map.
//
// SeedNtHash emits n_seeds x m2 hashes per window (seed-major; reference include/nthash/nthash.hpp:313-326, 460-479,
// src/seed.cpp:167-172) for exactly this use -- a spaced-seed Bloom filter takes every one of them.  The windows are the
// reference's (the position state machine of src/seed.cpp:493-544 on reads with non-bases: nthip_seed_hash).  Here the reads
// go round by round: a round's seed hashes to device scratch (nthip_seed_hash, whatever kernel the seed set and the read
// shape take), then the stream forms of the filter -- the binned insert (bloom_binned_kernels.hpp), the per-read query.
#include "capi_internal.hpp"

#include <algorithm>

#include "bloom_host.hpp"
#include "util_kernels.hpp" // (SCAN_TILE)

using namespace ntamd;
using namespace ntamd::host;

namespace {

int run_seed_bloom(nthip_ctx* c, const nthip_reads* rd, const nthip_seeds* sd, uint8_t m28, uint32_t* d_filter, uint64_t n_bits, uint64_t* hits,
                   uint64_t* total_out, uint64_t* total_hits, uint32_t flags, bool query)
{
  if (!c || !sd) return fail(NTHIP_ERR_ARG, "ctx/seeds is NULL");
  NTCHK(check_reads(rd));
  const uint32_t m2 = m28, k = sd->k, per = sd->n_seeds * m2;
  if (m2 == 0) return fail(NTHIP_ERR_UNSUPPORTED, "num_hashes_per_seed must be >= 1");
  if (per > 255) return fail(NTHIP_ERR_UNSUPPORTED, "more than 255 hashes per window");
  if (!d_filter || n_bits == 0) return fail(NTHIP_ERR_ARG, "filter is NULL / n_bits is 0");
  if ((uintptr_t)d_filter & 3u) return fail(NTHIP_ERR_ARG, "filter must be 4-byte aligned");
  HIPCHK(hipSetDevice(c->device));
  if (total_out) *total_out = 0;
  if (total_hits) *total_hits = 0;
  if (rd->n_reads == 0) return NTHIP_OK;
  const bool host_hits = query && hits && (flags & NTHIP_HOST_OUTPUT);
  uint64_t sum_windows = 0, sum_hits = 0;
  // one round: the reads of `part` (r0: its first read in the batch; bases: what it holds, an upper bound of its windows)
  auto one_round = [&](const nthip_reads* part, uint64_t r0, uint64_t bases) -> int {
    Staged keep;
    const uint64_t cap = bases ? bases : 1;
    const uint64_t nr = part->n_reads;
    uint64_t *d_h = nullptr, *d_counts = nullptr, n_windows = 0;
    NTCHK(kept_alloc(c, KEPT_STREAM, (size_t)cap * per * 8, (void**)&d_h));
    // The insert into a two-level filter asks the seed kernel for hashes()[0] of every seed only and lets the first partition level
    // make the other m2 - 1 (extend_hashes is a multiply and a shift of hashes()[0]): the stream written and read back is m2 times
    // shorter -- 17.6 GB each way instead of 53 for config 4's pair with 3 hashes per seed on 5 M reads.
    // (only when the binned insert will take the batch: one it declines would be hashed with one hash per seed, then in full)
    if (!query && m2 >= 2 && m2 <= 4 && n_bits > (1ull << 27) && c->tune.bloom_pieces != 2 && bloom_binned_applies(c, d_filter, n_bits, cap * per)) {
      nthip_out o1;
      memset(&o1, 0, sizeof o1);
      o1.hashes = d_h;
      o1.capacity = cap;
      uint64_t nw1 = 0;
      NTCHK(nthip_seed_hash(c, part, sd, 1, &o1, &nw1, flags & NTHIP_HOST_INPUT));
      bool done = false;
      if (nw1) NTCHK(stream_bloom_insert_expand(c, d_h, nw1 * sd->n_seeds, m2, (uint64_t)k * MULTISEED, d_filter, n_bits, &done));
      if (done || nw1 == 0) {
        sum_windows += nw1;
        return NTHIP_OK;
      } // (else: the full stream, below -- a bit set twice is set)
    }
    if (query) NTCHK(own_alloc(keep, (size_t)(nr + 1) * 8, (void**)&d_counts));
    // (the query of a two-level filter likewise: hashes()[0] of every seed, the other m2 - 1 values made on the way to the regions)
    const bool q_expand = query && m2 >= 2 && m2 <= 4 && n_bits > (1ull << 27) && c->tune.bloom_pieces != 2 && c->tune.bloom_query != 2 &&
                          (c->tune.bloom_query == 1 || (cap * per >= (1ull << 24) && (n_bits >> 3) >= (32ull << 20) && cap * per >= (n_bits >> 3) / 32));
    nthip_out out;
    memset(&out, 0, sizeof out);
    out.hashes = d_h;
    out.capacity = cap;
    out.counts = d_counts;
    NTCHK(nthip_seed_hash(c, part, sd, q_expand ? (uint8_t)1 : m28, &out, &n_windows, flags & NTHIP_HOST_INPUT));
    sum_windows += n_windows;
    if (!query) return n_windows ? nthip_stream_bloom_insert(c, d_h, n_windows * per, (uint8_t*)d_filter, n_bits) : NTHIP_OK;
    uint64_t* d_hits = hits ? hits + r0 : nullptr;
    if (host_hits) NTCHK(own_alloc(keep, (size_t)nr * 8, (void**)&d_hits));
    uint64_t *d_roff = nullptr, *d_sums = nullptr;
    NTCHK(own_alloc(keep, (size_t)(nr + 1) * 8, (void**)&d_roff));
    NTCHK(own_alloc(keep, (size_t)(nr / SCAN_TILE + 64) * 8, (void**)&d_sums));
    NTCHK(device_exclusive_scan(c, d_counts, d_roff, nr, d_sums, (uint64_t*)(c->d_small + 16)));
    HIPCHK(hipMemsetAsync(c->d_small + 24, 0, 8, c->stream));
    bool expanded = false;
    if (q_expand)
      NTCHK(stream_hits_per_read(c, d_h, d_roff, nr, n_windows, per, (const uint32_t*)d_filter, n_bits, d_hits, (unsigned long long*)(c->d_small + 24),
                                 "stream_bloom_query_kernel (spaced seeds)", m2, (uint64_t)k * MULTISEED, &expanded));
    if (!expanded) {
      if (q_expand) { // (the road was not taken after all -- skewed values, no memory: the full stream)
        uint64_t again = 0;
        out.counts = nullptr;
        NTCHK(nthip_seed_hash(c, part, sd, m28, &out, &again, flags & NTHIP_HOST_INPUT));
      }
      NTCHK(stream_hits_per_read(c, d_h, d_roff, nr, n_windows, per, (const uint32_t*)d_filter, n_bits, d_hits, (unsigned long long*)(c->d_small + 24),
                                 "stream_bloom_query_kernel (spaced seeds)"));
    }
    HIPCHK(hipMemcpyAsync(c->h_small + 24, c->d_small + 24, 8, hipMemcpyDeviceToHost, c->stream));
    if (host_hits) HIPCHK(hipMemcpyAsync(hits + r0, d_hits, nr * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    uint64_t h = 0;
    memcpy(&h, c->h_small + 24, 8);
    sum_hits += h;
    return NTHIP_OK;
  };
  if (rd->offsets) {
    NTCHK(offsets_in_rounds(c, rd, flags, (size_t)8 * per + (query ? (size_t)26 * per : 16), one_round));
  } else {
    const uint32_t len = rd->fixed_len, stride = rd->stride ? rd->stride : len;
    if (len < k) {
      if (query && hits) {
        if (host_hits) memset(hits, 0, rd->n_reads * sizeof(uint64_t));
        else HIPCHK(hipMemsetAsync(hits, 0, rd->n_reads * sizeof(uint64_t), c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
      }
      return NTHIP_OK;
    }
    const uint64_t nwin = len - k + 1;
    // rounds of reads whose seed hashes (+ the lists of the binned insert: 16 B per value) fit a third of the free memory
    const size_t free_b = round_memory(c, reusable_bytes(c), (size_t)4 << 30);
    // (the query through the regions: 8 B of hash + 1 B of answer + ~24 B of lists and records per value -- a round whose hashes
    //  alone took a third of the memory left the lists no room, and the call fell back to a filter line per value)
    // (the insert: 8 B of hash + ~9 B of lists per value)
    const uint64_t per_read = nwin * per * (query ? 34 : 20) + 48 + ((flags & NTHIP_HOST_INPUT) ? stride : 0);
    uint64_t reads_per_round = std::max<uint64_t>(1, (uint64_t)(free_b / 10 * 7) / per_read);
    if (c->tune.bloom_round) reads_per_round = std::max<uint64_t>(1, c->tune.bloom_round / (nwin * per));
    for (uint64_t r0 = 0; r0 < rd->n_reads; r0 += reads_per_round) {
      nthip_reads part = *rd;
      part.seqs = rd->seqs + r0 * stride;
      part.n_reads = std::min<uint64_t>(reads_per_round, rd->n_reads - r0);
      NTCHK(one_round(&part, r0, part.n_reads * nwin));
    }
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  if (total_out) *total_out = sum_windows;
  if (total_hits) *total_hits = sum_hits;
  return NTHIP_OK;
}

} // namespace

extern "C" int nthip_seed_bloom_insert(nthip_ctx* c, const nthip_reads* rd, const nthip_seeds* seeds, uint8_t m2, uint8_t* d_filter,
                                       uint64_t n_bits, uint64_t* total, uint32_t flags)
{
  return run_seed_bloom(c, rd, seeds, m2, (uint32_t*)d_filter, n_bits, nullptr, total, nullptr, flags, false);
}

extern "C" int nthip_seed_bloom_query(nthip_ctx* c, const nthip_reads* rd, const nthip_seeds* seeds, uint8_t m2, const uint8_t* d_filter,
                                      uint64_t n_bits, uint64_t* hits, uint64_t* total, uint64_t* total_hits, uint32_t flags)
{
  return run_seed_bloom(c, rd, seeds, m2, (uint32_t*)d_filter, n_bits, hits, total, total_hits, flags, true);
}
