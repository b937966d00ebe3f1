// capi_sink_bloom.hip -- fused consumers of the hash stream: Bloom filter insert / query
// Part of libnthash_hip.so (include/nthash_hip.h); see capi_internal.hpp for the file map.
#include "capi_internal.hpp"

#include <algorithm>

#include "bloom_binned_kernels.hpp"
#include "bloom_fused_kernels.hpp"
#include "bloom_host.hpp"
#include "bloom_query_kernels.hpp" // (BQ_BLOOM)
#include "util_kernels.hpp" // (SCAN_TILE)

using namespace ntamd;
using namespace ntamd::host;

// (not stage_alloc: the arena it hands out of starts over in every staged call, and nthip_kmer_hash below may be one)
int ntamd::host::own_alloc(Staged& keep, size_t bytes, void** p)
{
  HIPCHK(hipMalloc(p, bytes ? bytes : 16));
  keep.owned.push_back(*p);
  return NTHIP_OK;
}

int ntamd::host::offsets_in_rounds(nthip_ctx* c, const nthip_reads* rd, uint32_t flags, size_t scratch_per_base,
                                   const std::function<int(const nthip_reads*, uint64_t, uint64_t)>& fn)
{
  const uint64_t n = rd->n_reads;
  if (n == 0) return NTHIP_OK;
  const bool host = (flags & NTHIP_HOST_INPUT) != 0;
  // (the context's lists and kept buffers are reused by the round, not added to; the context's scratch limit caps the sum)
  const size_t free_b = round_memory(c, reusable_bytes(c), (size_t)4 << 30);
  uint64_t round_bases = (uint64_t)(free_b / 10 * 8) / (scratch_per_base + (host ? 1 : 0));
  if (c->tune.bloom_round) round_bases = c->tune.bloom_round; // (tests: several rounds on a small batch)
  const uint64_t reads_max = std::max<uint64_t>(1, (free_b / 10) / 48);
  uint64_t first = 0, last = 0;
  if (host) {
    first = rd->offsets[0];
    last = rd->offsets[n];
  } else {
    HIPCHK(hipMemcpy(&first, rd->offsets, 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&last, rd->offsets + n, 8, hipMemcpyDeviceToHost));
  }
  if (last < first) return fail(NTHIP_ERR_ARG, "offsets decrease");
  if (last - first <= round_bases && n <= reads_max) return fn(rd, 0, last - first);
  std::vector<uint64_t> ho(n + 1);
  if (host) memcpy(ho.data(), rd->offsets, (n + 1) * 8);
  else HIPCHK(hipMemcpy(ho.data(), rd->offsets, (n + 1) * 8, hipMemcpyDeviceToHost));
  for (uint64_t r = 0; r < n; ++r)
    if (ho[r + 1] < ho[r]) return fail(NTHIP_ERR_ARG, "offsets decrease at read %llu", (unsigned long long)r);
  std::vector<uint64_t> rebased;
  for (uint64_t r0 = 0; r0 < n;) {
    uint64_t r1 = (uint64_t)(std::upper_bound(ho.begin() + r0, ho.end(), ho[r0] + round_bases) - ho.begin()) - 1;
    if (r1 <= r0) r1 = r0 + 1; // (a read longer than a round: alone)
    if (r1 - r0 > reads_max) r1 = r0 + reads_max;
    if (r1 > n) r1 = n;
    nthip_reads part = *rd;
    part.n_reads = r1 - r0;
    if (host) {
      rebased.resize(r1 - r0 + 1);
      for (uint64_t i = 0; i <= r1 - r0; ++i) rebased[i] = ho[r0 + i] - ho[r0];
      part.seqs = rd->seqs + ho[r0];
      part.offsets = rebased.data();
    } else {
      part.offsets = rd->offsets + r0;
    }
    NTCHK(fn(&part, r0, ho[r1] - ho[r0]));
    r0 = r1;
  }
  return NTHIP_OK;
}

int ntamd::host::stream_of_offsets(nthip_ctx* c, const nthip_reads* rd, uint16_t k, uint8_t m, uint32_t flags, Staged& keep, uint64_t** d_h,
                      uint64_t** d_counts, uint64_t* n_kmers, uint64_t round_bases)
{
  uint64_t total_bytes = round_bases;
  if (round_bases == 0) NTCHK(reads_total_bytes(c, rd, flags, &total_bytes));
  const uint64_t cap = total_bytes > 0 ? total_bytes : 1;
  const size_t need = (size_t)cap * m * 8 + (d_counts ? (size_t)rd->n_reads * 16 + 4096 : 0);
  const size_t free_b = round_memory(c, c->kept_bytes[KEPT_STREAM], 0);
  if (need > free_b / 10 * 9)
    return fail(NTHIP_ERR_UNSUPPORTED, "reads given by offsets: the batch's hash stream (%llu MB) does not fit the device in one round; split the batch",
                (unsigned long long)(need >> 20));
  NTCHK(kept_alloc(c, KEPT_STREAM, (size_t)cap * m * 8, (void**)d_h));
  if (d_counts) NTCHK(own_alloc(keep, (size_t)(rd->n_reads + 1) * 8, (void**)d_counts));
  nthip_out out;
  memset(&out, 0, sizeof out);
  out.hashes = *d_h;
  out.capacity = cap;
  out.counts = d_counts ? *d_counts : nullptr;
  return nthip_kmer_hash(c, rd, k, m, &out, n_kmers, flags & NTHIP_HOST_INPUT);
}

namespace {

// ---- the binned insert (bloom_binned_kernels.hpp) -------------------------------------------------------------------
constexpr uint64_t BB_ROUND_MAX = 1ull << 31; // values per round: the lists are indexed with 32 bits

// Does a batch of n values into this filter go through the lists?  The filter must fit the histogram (2^35 bits) and
// sit on 16 bytes; the round's fixed costs -- one read-modify-write of the touched filter lines, five launches --
// must be small next to what the atomics would cost: ~37 ps per value against ~8 ps per value + ~0.4 ps per filter byte.
// (counters: the counting sketch -- one-byte slots, 2^15 per region -- instead of a filter's bits, 2^20 per region)
bool bloom_binned_ok(const nthip_ctx* c, const void* d_filter, uint64_t n_bits, uint64_t n_values, bool counters = false)
{
  if (c->tune.bloom_binned == 2) return false;
  const uint32_t region_shift = counters ? CS_REGION_SHIFT : BB_REGION_SHIFT;
  if (n_bits > ((uint64_t)BB_MAX_REGIONS << region_shift) || ((uintptr_t)d_filter & 15u)) return false;
  if (c->lds_max < (size_t)BB_REGION_DWORDS * 4 + 1024) return false;
  if (c->tune.bloom_binned == 1) return n_values != 0;
  const uint64_t filter_bytes = counters ? n_bits : (n_bits + 7) / 8;
  return n_values >= (1ull << 22) && n_values >= filter_bytes / 64;
}

struct BloomLists {
  uint32_t *counts = nullptr, *region_base = nullptr, *region_cursor = nullptr, *bin_cursor = nullptr;
  uint32_t *list1 = nullptr, *list2 = nullptr;
  uint64_t* hashes = nullptr; // the round's hash stream (the reads entry)
};

// lists for rounds of at most *round values (+ a hash stream of that many values when with_stream).  When the device does
// not have the memory, *round is halved until it has; returns 1 (not an error: the caller takes the atomic kernels) when
// not even a round of 2^20 values fits.
int bloom_lists(nthip_ctx* c, uint64_t* round, bool with_stream, BloomLists* t)
{
  const size_t head = ((size_t)BB_MAX_REGIONS * (2 + BB_CURSOR_STRIDE) + 1 + (size_t)BB_MAX_BINS * BB_CURSOR_STRIDE + 64) * sizeof(uint32_t);
  const size_t head_al = (head + 255) & ~(size_t)255;
  size_t list_bytes = 0;
  for (;;) {
    list_bytes = (((size_t)*round * 4) + 255) & ~(size_t)255;
    const size_t need = head_al + 2 * list_bytes + (with_stream ? (size_t)*round * 8 : 0);
    if (c->bloom_tmp_bytes >= need) break;
    if (c->bloom_tmp) HIPCHK(hipFree(c->bloom_tmp));
    c->bloom_tmp = nullptr;
    c->bloom_tmp_bytes = 0;
    if (hipMalloc((void**)&c->bloom_tmp, need) == hipSuccess) {
      c->bloom_tmp_bytes = need;
      break;
    }
    (void)hipGetLastError(); // (out of memory is an answer here, not a sticky error)
    c->bloom_tmp = nullptr;
    if (*round <= (1u << 20)) return 1;
    *round = *round / 2 < (1u << 20) ? (1u << 20) : *round / 2;
  }
  uint32_t* p = (uint32_t*)c->bloom_tmp;
  t->counts = p;
  t->region_base = p + BB_MAX_REGIONS;
  t->region_cursor = t->region_base + BB_MAX_REGIONS + 1;
  t->bin_cursor = t->region_cursor + (size_t)BB_MAX_REGIONS * BB_CURSOR_STRIDE;
  t->list1 = (uint32_t*)(c->bloom_tmp + head_al);
  t->list2 = (uint32_t*)(c->bloom_tmp + head_al + list_bytes);
  t->hashes = with_stream ? (uint64_t*)(c->bloom_tmp + head_al + 2 * list_bytes) : nullptr;
  return NTHIP_OK;
}

// values per round: what the free memory allows (16 B per value with the hash stream, 8 without), at most BB_ROUND_MAX
uint64_t bloom_round_values(const nthip_ctx* c, uint64_t n_values, bool with_stream)
{
  const size_t free_b = round_memory(const_cast<nthip_ctx*>(c), c->bloom_tmp_bytes, (size_t)8 << 30); // (what the context already holds is ours to reuse)
  const uint64_t per = with_stream ? 16 : 8;
  uint64_t round = (uint64_t)(free_b / 2) / per;
  if (round > BB_ROUND_MAX) round = BB_ROUND_MAX;
  if (round > n_values) round = n_values;
  if (round < (1u << 20)) round = 1u << 20;
  if (c->tune.bloom_round) round = c->tune.bloom_round;
  return round;
}

// ---- the round WITHOUT a hash stream (bloom_fused_kernels.hpp): device-resident fixed-length reads hashed twice ----------
// pass COUNT: t.counts := values per region (launch only)
int bloom_fused_count(nthip_ctx* c, const BloomFusedSrc& s, uint64_t n_bits, const BloomLists& t, bool counters)
{
  const uint32_t region_shift = counters ? CS_REGION_SHIFT : BB_REGION_SHIFT;
  const uint32_t n_regions = (uint32_t)((n_bits + (1ull << region_shift) - 1) >> region_shift);
  HIPCHK(hipMemsetAsync(t.counts, 0, (size_t)n_regions * sizeof(uint32_t), c->stream));
  const uint32_t threads = n_regions > 16384u ? BF_COUNT_THREADS_BIG : 1024u;
  BloomFusedArgs a;
  bloom_fused_args(s, threads, n_bits, bloom_magic_of(n_bits), &a);
  a.counts = t.counts;
  a.n_regions = n_regions;
  a.region_shift = region_shift;
  const size_t lds = bloom_fused_lds(s, threads, n_regions < 128u ? 128u : n_regions);
  const unsigned grid = (unsigned)std::min<uint64_t>(a.n_tiles, (uint64_t)c->n_cu);
  prof_begin(c, counters ? "count fused insert (count, scan, part, apply)" : "bloom fused insert (count, scan, part, apply)");
  if (threads == 1024u) {
    NTCHK(set_max_lds(c, bloom_fused_kernel<BF_COUNT, 1024>, lds));
    hipLaunchKernelGGL((bloom_fused_kernel<BF_COUNT, 1024>), dim3(grid), dim3(1024), lds, c->stream, a);
  } else {
    NTCHK(set_max_lds(c, bloom_fused_kernel<BF_COUNT, BF_COUNT_THREADS_BIG>, lds));
    hipLaunchKernelGGL((bloom_fused_kernel<BF_COUNT, BF_COUNT_THREADS_BIG>), dim3(grid), dim3(BF_COUNT_THREADS_BIG), lds, c->stream, a);
  }
  HIPCHK(hipGetLastError());
  return NTHIP_OK;
}
// the windows the fused pass PART of the exact lists did not emit: zeroed before the round, read after it (synchronises)
int bloom_fused_lost_reset(nthip_ctx* c) { HIPCHK(hipMemsetAsync(c->d_small + 48, 0, 8, c->stream)); return NTHIP_OK; }
int bloom_fused_lost_read(nthip_ctx* c, uint64_t* lost)
{
  HIPCHK(hipMemcpyAsync(c->h_small + 48, c->d_small + 48, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  memcpy(lost, c->h_small + 48, 8);
  return NTHIP_OK;
}

// one round: n <= BB_ROUND_MAX values of a device-resident stream into the filter (launches only, no synchronisation)
// (fused: the round's values come from the reads themselves -- t.counts already holds pass COUNT's histogram, the first
// partition level is pass PART)
int bloom_binned_round(nthip_ctx* c, const uint64_t* d_hashes, uint64_t n, uint32_t* d_filter, uint64_t n_bits,
                       const BloomLists& t, bool counters = false, const BloomFusedSrc* fused = nullptr)
{
  const uint32_t region_shift = counters ? CS_REGION_SHIFT : BB_REGION_SHIFT, bin_shift = region_shift + 7u;
  const uint32_t n_regions = (uint32_t)((n_bits + (1ull << region_shift) - 1) >> region_shift);
  const uint32_t n_bins = (n_regions + BB_REGIONS_PER_BIN - 1) / BB_REGIONS_PER_BIN;
  const uint64_t magic = bloom_magic_of(n_bits);
  const uint64_t filter_dwords = counters ? (n_bits + 3) / 4 : (n_bits + 31) / 32;
  if (!fused) {
    HIPCHK(hipMemsetAsync(t.counts, 0, (size_t)n_regions * sizeof(uint32_t), c->stream));
    const size_t hist_lds = (size_t)(n_regions < 128u ? 128u : n_regions) * sizeof(uint32_t);
    NTCHK(set_max_lds(c, bloom_hist_kernel, hist_lds));
    prof_begin(c, counters ? "count binned insert (hist, scan, part, apply)" : "bloom binned insert (hist, scan, part, apply)");
    hipLaunchKernelGGL(bloom_hist_kernel, dim3(c->n_cu), dim3(1024), hist_lds, c->stream, d_hashes, n, n_bits, magic, n_regions,
                       t.counts, region_shift);
  }
  // the first level from the reads (pass PART of bloom_fused_kernels.hpp) instead of from a stream
  auto fused_level1 = [&](uint32_t* out, uint32_t* cursor, uint32_t shift, uint32_t n_buckets) -> int {
    BloomFusedArgs fa;
    bloom_fused_args(*fused, 1024u, n_bits, magic, &fa);
    fa.lost = (unsigned long long*)(c->d_small + 48);
    fa.out = out;
    fa.cursor = cursor;
    fa.shift = shift;
    fa.mask = (1u << shift) - 1u;
    fa.n_buckets = n_buckets;
    const size_t lds = bloom_fused_lds(*fused, 1024u, 1024u * 16u);
    NTCHK(set_max_lds(c, bloom_fused_kernel<BF_PART, 1024>, lds));
    hipLaunchKernelGGL((bloom_fused_kernel<BF_PART, 1024>), dim3((unsigned)std::min<uint64_t>(fa.n_tiles, (uint64_t)c->n_cu)), dim3(1024), lds,
                       c->stream, fa);
    return NTHIP_OK;
  };
  hipLaunchKernelGGL(bloom_scan_kernel, dim3(1), dim3(1024), 0, c->stream, (const uint32_t*)t.counts, n_regions, t.region_base,
                     t.region_cursor, t.bin_cursor);
  auto part_lds = [](uint32_t threads) { return (size_t)threads * BB_PART_ITEMS * (sizeof(uint32_t) + (BB_COPY_SLOT ? 1 : 0)); };
  auto part_blocks = [&](uint32_t threads) { return (uint32_t)c->n_cu * (part_lds(threads) > 48 * 1024 ? 2u : 4u); };
  NTCHK(set_max_lds(c, bloom_part_kernel<true, BB_L1_THREADS>, part_lds(BB_L1_THREADS)));
  NTCHK(set_max_lds(c, bloom_part_kernel<false, BB_L2_THREADS>, part_lds(BB_L2_THREADS)));
  BloomPartArgs a;
  memset(&a, 0, sizeof a);
  a.n = n;
  a.n_bits = n_bits;
  a.magic = magic;
  a.n_regions = n_regions;
  a.seg_base = t.region_base;
  const uint32_t* entries;
  a.in = d_hashes;
  if (n_bins == 1) { // a filter of at most 2^27 bits: straight to the regions
    a.out = t.list2;
    a.cursor = t.region_cursor;
    a.shift = region_shift;
    a.mask = (1u << region_shift) - 1u;
    a.buckets_per_seg = n_regions;
    if (fused) NTCHK(fused_level1(t.list2, t.region_cursor, region_shift, n_regions));
    else
      hipLaunchKernelGGL((bloom_part_kernel<true, BB_L1_THREADS>), dim3(part_blocks(BB_L1_THREADS)), dim3(BB_L1_THREADS),
                         part_lds(BB_L1_THREADS), c->stream, a);
  } else {
    a.out = t.list1;
    a.cursor = t.bin_cursor;
    a.shift = bin_shift;
    a.mask = (1u << bin_shift) - 1u;
    a.buckets_per_seg = n_bins;
    if (fused) NTCHK(fused_level1(t.list1, t.bin_cursor, bin_shift, n_bins));
    else
      hipLaunchKernelGGL((bloom_part_kernel<true, BB_L1_THREADS>), dim3(part_blocks(BB_L1_THREADS)), dim3(BB_L1_THREADS),
                         part_lds(BB_L1_THREADS), c->stream, a);
    a.in = t.list1;
    a.out = t.list2;
    a.cursor = t.region_cursor;
    a.shift = region_shift;
    a.mask = (1u << region_shift) - 1u;
    a.buckets_per_seg = BB_REGIONS_PER_BIN;
    // blocks per bin: the grid is TWICE what the device holds at once (the kernel's registers decide that: two blocks per CU
    // since its loads run a tile ahead), so that no round of blocks runs part empty -- 5 per bin of a 4 GiB filter, 1280
    // blocks on 512 places, took 6.9 ms where 4 per bin take 6.2
    int l2_per_cu = 1;
    NTCHK(blocks_per_cu(c, bloom_part_kernel<false, BB_L2_THREADS>, (int)BB_L2_THREADS, part_lds(BB_L2_THREADS), &l2_per_cu));
    const uint32_t l2_grid = 2u * (uint32_t)c->n_cu * (uint32_t)l2_per_cu;
    const uint32_t per_bin = l2_grid / n_bins ? l2_grid / n_bins : 1u;
    hipLaunchKernelGGL((bloom_part_kernel<false, BB_L2_THREADS>), dim3(per_bin, n_bins), dim3(BB_L2_THREADS),
                       part_lds(BB_L2_THREADS), c->stream, a);
  }
  entries = t.list2;
  const size_t apply_lds = (size_t)BB_REGION_DWORDS * sizeof(uint32_t);
  const uint32_t grid = n_regions < (uint32_t)c->n_cu ? n_regions : (uint32_t)c->n_cu;
  if (counters) {
    NTCHK(set_max_lds(c, count_apply_kernel, apply_lds));
    hipLaunchKernelGGL(count_apply_kernel, dim3(grid), dim3(BB_APPLY_THREADS), apply_lds, c->stream, entries,
                       (const uint32_t*)t.region_base, n_regions, d_filter, filter_dwords, (uint64_t)0, (const uint32_t*)nullptr,
                       (const BloomStatus*)nullptr, (uint64_t)0);
  } else {
    NTCHK(set_max_lds(c, bloom_apply_kernel, apply_lds));
    hipLaunchKernelGGL(bloom_apply_kernel, dim3(grid), dim3(BB_APPLY_THREADS), apply_lds, c->stream, entries,
                       (const uint32_t*)t.region_base, n_regions, d_filter, filter_dwords, (uint64_t)0, (const uint32_t*)nullptr,
                       (const BloomStatus*)nullptr, (uint64_t)0);
  }
  prof_end(c);
  HIPCHK(hipGetLastError());
  return NTHIP_OK;
}

// ---- slots mode (bloom_binned_kernels.hpp): no histogram, every bucket owns mean + 8 sigma entries ----------------------
constexpr uint64_t BB_SLOTS_ROUND_MAX = 0xF0000000ull; // values per round: a bucket's cursor is 32 bits
constexpr uint32_t BB_SLOTS_BACKOFF = 16;              // calls that keep to the exact lists after a round failed

struct SlotLists {
  uint32_t *cur1 = nullptr, *cur2 = nullptr; // cursors of the bins / of the regions (filters of one bin: cur2 only)
  uint32_t *list1 = nullptr, *list2 = nullptr;
  uint64_t* ovf = nullptr;
  BloomStatus* status = nullptr;
  uint64_t cap1 = 0, cap2 = 0, ovf_cap = 0;
  size_t head_bytes = 0; // status + cursors: zeroed before every round
};
bool bloom_slots_ok(nthip_ctx* c)
{
  if (c->tune.bloom_slots == 2) return false;
  if (c->tune.bloom_slots == 1) return true;
  if (c->bloom_slots_backoff) {
    --c->bloom_slots_backoff;
    return false;
  }
  return true;
}
// values per round of the slots mode: what the free memory allows (about 8.7 B per value), at most BB_SLOTS_ROUND_MAX
uint64_t slots_round_values(const nthip_ctx* c, uint64_t n_values, uint64_t round_max = BB_SLOTS_ROUND_MAX)
{
  const size_t free_b = round_memory(const_cast<nthip_ctx*>(c), c->bloom_tmp_bytes, (size_t)8 << 30);
  uint64_t round = (uint64_t)(free_b / 2) / 9;
  if (round > round_max) round = round_max;
  if (round > n_values) round = n_values;
  if (round < (1u << 20)) round = 1u << 20;
  if (c->tune.bloom_round) round = c->tune.bloom_round;
  return round;
}
// the lists of a round of n values; 1: the device does not have the memory (the caller keeps to the exact lists' rounds)
int bloom_slot_lists(nthip_ctx* c, uint64_t n, uint64_t n_slots, bool counters, SlotLists* t)
{
  const uint32_t region_shift = counters ? CS_REGION_SHIFT : BB_REGION_SHIFT, bin_shift = region_shift + 7u;
  const uint32_t n_regions = (uint32_t)((n_slots + (1ull << region_shift) - 1) >> region_shift);
  const uint32_t n_bins = (n_regions + BB_REGIONS_PER_BIN - 1) / BB_REGIONS_PER_BIN;
  t->cap1 = n_bins > 1 ? slot_cap(c, n, 1ull << bin_shift, n_slots) : 0;
  t->cap2 = slot_cap(c, n, 1ull << region_shift, n_slots);
  t->ovf_cap = n / 64 < 65536 ? 65536 : n / 64;
  if (c->tune.bloom_slot_tight == 2) t->ovf_cap = 64;
  const size_t head = (sizeof(BloomStatus) + 255) / 256 * 256 + (size_t)(n_bins + n_regions) * BB_CURSOR_STRIDE * sizeof(uint32_t);
  const size_t head_al = (head + 255) & ~(size_t)255;
  const size_t l1 = (size_t)n_bins * t->cap1 * 4, l2 = (size_t)n_regions * t->cap2 * 4;
  const size_t need = head_al + l1 + l2 + (size_t)t->ovf_cap * 8;
  if (c->bloom_tmp_bytes < need) {
    if (c->bloom_tmp) HIPCHK(hipFree(c->bloom_tmp));
    c->bloom_tmp = nullptr;
    c->bloom_tmp_bytes = 0;
    if (hipMalloc((void**)&c->bloom_tmp, need) != hipSuccess) {
      (void)hipGetLastError();
      c->bloom_tmp = nullptr;
      return 1;
    }
    c->bloom_tmp_bytes = need;
  }
  t->status = (BloomStatus*)c->bloom_tmp;
  t->cur1 = (uint32_t*)(c->bloom_tmp + (sizeof(BloomStatus) + 255) / 256 * 256);
  t->cur2 = t->cur1 + (size_t)n_bins * BB_CURSOR_STRIDE;
  t->head_bytes = head;
  t->list1 = (uint32_t*)(c->bloom_tmp + head_al);
  t->list2 = (uint32_t*)(c->bloom_tmp + head_al + l1);
  t->ovf = (uint64_t*)(c->bloom_tmp + head_al + l1 + l2);
  return NTHIP_OK;
}

// One round in slots mode: n values -- of the reads `fused` describes (hashed once, bloom_fused_kernels.hpp pass PART; n = what
// they hold when no window is lost, *lost = the windows with a non-base) or of the stream d_hashes -- into the table.
// *outcome: 0 done; 2 the overflow list overflowed; 3 no memory for the lists.  After 2 / 3 the table is untouched: the
// caller redoes the round on the exact lists.
int bloom_slots_round(nthip_ctx* c, const BloomFusedSrc* fused, const uint64_t* d_hashes, uint64_t n, uint32_t* d_table, uint64_t n_slots,
                      bool counters, int* outcome, uint64_t* lost = nullptr)
{
  const uint32_t region_shift = counters ? CS_REGION_SHIFT : BB_REGION_SHIFT, bin_shift = region_shift + 7u;
  const uint32_t n_regions = (uint32_t)((n_slots + (1ull << region_shift) - 1) >> region_shift);
  const uint32_t n_bins = (n_regions + BB_REGIONS_PER_BIN - 1) / BB_REGIONS_PER_BIN;
  const uint64_t magic = bloom_magic_of(n_slots);
  const uint64_t table_dwords = counters ? (n_slots + 3) / 4 : (n_slots + 31) / 32;
  SlotLists t;
  {
    const int rc = bloom_slot_lists(c, n, n_slots, counters, &t);
    if (rc < 0) return rc;
    if (rc == 1) {
      *outcome = 3;
      return NTHIP_OK;
    }
  }
  HIPCHK(hipMemsetAsync(t.status, 0, t.head_bytes, c->stream));
  prof_begin(c, fused ? (counters ? "count fused insert, slots (part, part, apply)" : "bloom fused insert, slots (part, part, apply)")
                      : (counters ? "count binned insert, slots (part, part, apply)" : "bloom binned insert, slots (part, part, apply)"));
  // level 1: to the bins (to the regions when the table is one bin)
  const bool one = n_bins == 1;
  const uint32_t shift1 = one ? region_shift : bin_shift, buckets1 = one ? n_regions : n_bins;
  uint32_t* const out1 = one ? t.list2 : t.list1;
  uint32_t* const cur_l1 = one ? t.cur2 : t.cur1;
  const BloomSlots sl1 = {one ? t.cap2 : t.cap1, t.ovf, t.status, t.ovf_cap};
  auto part_lds = [](uint32_t threads) { return (size_t)threads * BB_PART_ITEMS * (sizeof(uint32_t) + (BB_COPY_SLOT ? 1 : 0)); };
  auto part_blocks = [&](uint32_t threads) { return (uint32_t)c->n_cu * (part_lds(threads) > 48 * 1024 ? 2u : 4u); };
  if (fused) {
    BloomFusedArgs fa;
    bloom_fused_args(*fused, 1024u, n_slots, magic, &fa);
    fa.lost = &t.status->lost;
    fa.out = out1;
    fa.cursor = cur_l1;
    fa.shift = shift1;
    fa.mask = (1u << shift1) - 1u;
    fa.n_buckets = buckets1;
    fa.sl = sl1;
    const size_t lds = bloom_fused_lds(*fused, 1024u, 1024u * 16u);
    NTCHK(set_max_lds(c, bloom_fused_kernel<BF_PART, 1024>, lds));
    hipLaunchKernelGGL((bloom_fused_kernel<BF_PART, 1024>), dim3((unsigned)std::min<uint64_t>(fa.n_tiles, (uint64_t)c->n_cu)), dim3(1024), lds,
                       c->stream, fa);
  }
  BloomPartArgs a;
  memset(&a, 0, sizeof a);
  a.n = n;
  a.n_bits = n_slots;
  a.magic = magic;
  a.n_regions = n_regions;
  if (!fused) {
    a.in = d_hashes;
    a.out = out1;
    a.cursor = cur_l1;
    a.shift = shift1;
    a.mask = (1u << shift1) - 1u;
    a.buckets_per_seg = buckets1;
    a.sl = sl1;
    NTCHK(set_max_lds(c, bloom_part_kernel<true, BB_L1_THREADS>, part_lds(BB_L1_THREADS)));
    hipLaunchKernelGGL((bloom_part_kernel<true, BB_L1_THREADS>), dim3(part_blocks(BB_L1_THREADS)), dim3(BB_L1_THREADS),
                       part_lds(BB_L1_THREADS), c->stream, a);
  }
  if (!one) { // level 2: every bin to its regions
    a.in = t.list1;
    a.out = t.list2;
    a.cursor = t.cur2;
    a.shift = region_shift;
    a.mask = (1u << region_shift) - 1u;
    a.buckets_per_seg = BB_REGIONS_PER_BIN;
    a.sl = {t.cap2, t.ovf, t.status, t.ovf_cap};
    a.cap_in = t.cap1;
    a.seg_fill = t.cur1;
    NTCHK(set_max_lds(c, bloom_part_kernel<false, BB_L2_THREADS>, part_lds(BB_L2_THREADS)));
    // blocks per bin: the grid is TWICE what the device holds at once (the kernel's registers decide that: two blocks per CU
    // since its loads run a tile ahead), so that no round of blocks runs part empty -- 5 per bin of a 4 GiB filter, 1280
    // blocks on 512 places, took 6.9 ms where 4 per bin take 6.2
    int l2_per_cu = 1;
    NTCHK(blocks_per_cu(c, bloom_part_kernel<false, BB_L2_THREADS>, (int)BB_L2_THREADS, part_lds(BB_L2_THREADS), &l2_per_cu));
    const uint32_t l2_grid = 2u * (uint32_t)c->n_cu * (uint32_t)l2_per_cu;
    const uint32_t per_bin = l2_grid / n_bins ? l2_grid / n_bins : 1u;
    hipLaunchKernelGGL((bloom_part_kernel<false, BB_L2_THREADS>), dim3(per_bin, n_bins), dim3(BB_L2_THREADS), part_lds(BB_L2_THREADS),
                       c->stream, a);
  }
  const size_t apply_lds = (size_t)BB_REGION_DWORDS * sizeof(uint32_t);
  const uint32_t grid = n_regions < (uint32_t)c->n_cu ? n_regions : (uint32_t)c->n_cu;
  if (counters) {
    NTCHK(set_max_lds(c, count_apply_kernel, apply_lds));
    hipLaunchKernelGGL(count_apply_kernel, dim3(grid), dim3(BB_APPLY_THREADS), apply_lds, c->stream, (const uint32_t*)t.list2,
                       (const uint32_t*)nullptr, n_regions, d_table, table_dwords, t.cap2, (const uint32_t*)t.cur2,
                       (const BloomStatus*)t.status, t.ovf_cap);
    hipLaunchKernelGGL(bloom_overflow_apply_kernel<true>, dim3(c->n_cu), dim3(256), 0, c->stream, (const uint64_t*)t.ovf,
                       (const BloomStatus*)t.status, t.ovf_cap, d_table);
  } else {
    NTCHK(set_max_lds(c, bloom_apply_kernel, apply_lds));
    hipLaunchKernelGGL(bloom_apply_kernel, dim3(grid), dim3(BB_APPLY_THREADS), apply_lds, c->stream, (const uint32_t*)t.list2,
                       (const uint32_t*)nullptr, n_regions, d_table, table_dwords, t.cap2, (const uint32_t*)t.cur2,
                       (const BloomStatus*)t.status, t.ovf_cap);
    hipLaunchKernelGGL(bloom_overflow_apply_kernel<false>, dim3(c->n_cu), dim3(256), 0, c->stream, (const uint64_t*)t.ovf,
                       (const BloomStatus*)t.status, t.ovf_cap, d_table);
  }
  prof_end(c);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(c->h_small + 64, t.status, sizeof(BloomStatus), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  BloomStatus st;
  memcpy(&st, c->h_small + 64, sizeof st);
  *outcome = st.ovf_n > t.ovf_cap ? 2 : 0;
  if (*outcome == 2) c->bloom_slots_backoff = BB_SLOTS_BACKOFF;
  if (lost) *lost = st.lost;
  return NTHIP_OK;
}

// ---- pieces mode (bloom_binned_kernels.hpp): a two-level round of device-resident fixed-length reads, the lists as
// block-private pieces written in whole lines -- no cursors, no atomics.  *outcome as bloom_slots_round, and 4: not a shape /
// table of this mode (nothing done)
// (stream != NULL: the n_stream values of a hash stream instead of reads -- level 1 is bloom_part_stream_pieces_kernel)
// (expand_m = 2 ... 4: the stream holds hashes()[0] only and level 1 makes the other expand_m - 1 values of every input -- kmul =
//  k * MULTISEED, n_stream counts the INPUTS; other expand_m: outcome 4)
int bloom_pieces_round(nthip_ctx* c, const BloomFusedSrc* srcp, const uint64_t* stream, uint64_t n_stream, uint32_t* d_table, uint64_t n_slots,
                       bool counters, int* outcome, uint64_t* lost, uint32_t expand_m = 1, uint64_t kmul = 0)
{
#ifndef BB_S1_THREADS
#define BB_S1_THREADS 1024
#endif
  constexpr uint32_t S1_THREADS = BB_S1_THREADS, S1_TILE = S1_THREADS * BB_PART_ITEMS;
  const BloomFusedSrc src = srcp ? *srcp : BloomFusedSrc{};
  const uint32_t region_shift = counters ? CS_REGION_SHIFT : BB_REGION_SHIFT, bin_shift = region_shift + 7u;
  const uint32_t n_regions = (uint32_t)((n_slots + (1ull << region_shift) - 1) >> region_shift);
  const uint32_t n_bins = (n_regions + BB_REGIONS_PER_BIN - 1) / BB_REGIONS_PER_BIN;
  *outcome = 4;
  if (n_bins < 2 || c->tune.bloom_pieces == 2) return NTHIP_OK;
  if (expand_m < 1 || expand_m > 4 || (expand_m > 1 && !stream)) return NTHIP_OK;
  const size_t lds1 = stream ? ((size_t)S1_TILE + (size_t)BB_MAX_BINS * 32u) * sizeof(uint32_t) : bloom_fused_lds(src, 1024u, 1024u * 16u + BB_PIECES_LDS_DWORDS);
  if (lds1 > lds_cap_of(c) - 4096) return NTHIP_OK;
  const uint64_t magic = bloom_magic_of(n_slots);
  const uint64_t table_dwords = counters ? (n_slots + 3) / 4 : (n_slots + 31) / 32;
  const uint32_t nwin = stream ? 0u : src.len - src.k + 1u;
  const uint64_t n = stream ? n_stream * expand_m : src.n_reads * (uint64_t)nwin * src.m; // values
  constexpr uint32_t L2_TILE = BB_L2_THREADS * BB_PART_ITEMS;
  const size_t lds2 = ((size_t)L2_TILE + (size_t)BB_REGIONS_PER_BIN * 32u) * sizeof(uint32_t);
  int l2_per_cu = 1;
  NTCHK(blocks_per_cu(c, bloom_part_pieces_kernel<BB_L2_THREADS, false>, (int)BB_L2_THREADS, lds2, &l2_per_cu));
  const uint32_t l2_grid = 2u * (uint32_t)c->n_cu * (uint32_t)l2_per_cu;
  const uint32_t gx = l2_grid / n_bins ? l2_grid / n_bins : 1u;
  PiecesGeo g;
  if (stream) { // a block of level 1 takes every g1-th tile of the stream
    int per1 = 1;
    NTCHK(blocks_per_cu(c, bloom_part_stream_pieces_kernel<S1_THREADS, false>, (int)S1_THREADS, lds1, &per1));
    const uint64_t tile_in = (uint64_t)S1_THREADS * (BB_PART_ITEMS / expand_m); // inputs per tile
    const uint64_t tiles1 = (n_stream + tile_in - 1) / tile_in;
    g.g1 = (uint32_t)std::min<uint64_t>(tiles1 ? tiles1 : 1, (uint64_t)c->n_cu * (uint64_t)per1);
    g.gx = gx;
    const double per_block = (double)((tiles1 + g.g1 - 1) / g.g1) * (double)(tile_in * expand_m);
    const double bin_slots = (double)(1ull << bin_shift), region_slots = (double)(1ull << region_shift);
    g.cap1 = piece_cap(c, per_block * (bin_slots < (double)n_slots ? bin_slots / (double)n_slots : 1.0));
    g.cap2 = piece_cap(c, (double)((g.g1 + gx - 1) / gx) * per_block * (region_slots < (double)n_slots ? region_slots / (double)n_slots : 1.0));
  } else {
    pieces_geo(c, src.n_reads, (uint64_t)nwin * src.m, n_slots, region_shift, gx, &g);
  }
  const uint64_t ovf_cap = c->tune.bloom_slot_tight == 2 ? 64 : (n / 64 < 65536 ? 65536 : n / 64);
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t o_fill1 = 256, o_fill2 = o_fill1 + al((size_t)g.g1 * n_bins * 4), o_l1 = o_fill2 + al((size_t)n_bins * gx * BB_REGIONS_PER_BIN * 4);
  const size_t o_l2 = o_l1 + al((size_t)n_bins * g.g1 * g.cap1 * 4), o_ovf = o_l2 + al((size_t)n_regions * gx * g.cap2 * 4);
  const size_t need = o_ovf + al((size_t)ovf_cap * 8);
  if (c->bloom_tmp_bytes < need) {
    if (c->bloom_tmp) HIPCHK(hipFree(c->bloom_tmp));
    c->bloom_tmp = nullptr;
    c->bloom_tmp_bytes = 0;
    if (hipMalloc((void**)&c->bloom_tmp, need) != hipSuccess) {
      (void)hipGetLastError();
      c->bloom_tmp = nullptr;
      *outcome = 3;
      return NTHIP_OK;
    }
    c->bloom_tmp_bytes = need;
  }
  BloomStatus* const status = (BloomStatus*)c->bloom_tmp;
  uint32_t* const fill1 = (uint32_t*)(c->bloom_tmp + o_fill1);
  uint32_t* const fill2 = (uint32_t*)(c->bloom_tmp + o_fill2);
  uint32_t* const list1 = (uint32_t*)(c->bloom_tmp + o_l1);
  uint32_t* const list2 = (uint32_t*)(c->bloom_tmp + o_l2);
  uint64_t* const ovf = (uint64_t*)(c->bloom_tmp + o_ovf);
  HIPCHK(hipMemsetAsync(status, 0, 256, c->stream));
  prof_begin(c, stream ? (counters ? "count binned insert, pieces (part, part, apply)" : "bloom binned insert, pieces (part, part, apply)")
                       : (counters ? "count fused insert, pieces (part, part, apply)" : "bloom fused insert, pieces (part, part, apply)"));
  if (stream) {
    BloomPartStreamPiecesArgs a;
    memset((void*)&a, 0, sizeof a);
    a.in = stream;
    a.n = n_stream;
    a.kmul = kmul;
    a.n_bits = n_slots;
    a.magic = magic;
    a.out = list1;
    a.fill_out = fill1;
    a.shift = bin_shift;
    a.mask = (1u << bin_shift) - 1u;
    a.n_buckets = n_bins;
    a.sl = {g.cap1, ovf, status, ovf_cap};
    switch (expand_m) {
      case 2:
        NTCHK(set_max_lds(c, bloom_part_stream_pieces_kernel<S1_THREADS, false, 2>, lds1));
        hipLaunchKernelGGL((bloom_part_stream_pieces_kernel<S1_THREADS, false, 2>), dim3(g.g1), dim3(S1_THREADS), lds1, c->stream, a);
        break;
      case 3:
        NTCHK(set_max_lds(c, bloom_part_stream_pieces_kernel<S1_THREADS, false, 3>, lds1));
        hipLaunchKernelGGL((bloom_part_stream_pieces_kernel<S1_THREADS, false, 3>), dim3(g.g1), dim3(S1_THREADS), lds1, c->stream, a);
        break;
      case 4:
        NTCHK(set_max_lds(c, bloom_part_stream_pieces_kernel<S1_THREADS, false, 4>, lds1));
        hipLaunchKernelGGL((bloom_part_stream_pieces_kernel<S1_THREADS, false, 4>), dim3(g.g1), dim3(S1_THREADS), lds1, c->stream, a);
        break;
      default:
        NTCHK(set_max_lds(c, bloom_part_stream_pieces_kernel<S1_THREADS, false>, lds1));
        hipLaunchKernelGGL((bloom_part_stream_pieces_kernel<S1_THREADS, false>), dim3(g.g1), dim3(S1_THREADS), lds1, c->stream, a);
    }
  } else {
    BloomFusedPiecesArgs fa;
    bloom_fused_args(src, 1024u, n_slots, magic, &fa);
    fa.lost = &status->lost;
    fa.out = list1;
    fa.cursor = nullptr;
    fa.shift = bin_shift;
    fa.mask = (1u << bin_shift) - 1u;
    fa.n_buckets = n_bins;
    fa.sl = {g.cap1, ovf, status, ovf_cap};
    fa.q_where = nullptr;
    fa.q_tab = nullptr;
    fa.q_tovf = nullptr;
    fa.q_steps = 0;
    fa.p_fill = fill1;
    NTCHK(set_max_lds(c, bloom_fused_kernel<BF_PART, 1024, false, true>, lds1));
    hipLaunchKernelGGL((bloom_fused_kernel<BF_PART, 1024, false, true>), dim3(g.g1), dim3(1024), lds1, c->stream, fa);
  }
  {
    BloomPartPiecesArgs a;
    memset(&a, 0, sizeof a);
    a.in = list1;
    a.out = list2;
    a.fill_in = fill1;
    a.fill_out = fill2;
    a.cap_in = g.cap1;
    a.n_pieces_in = g.g1;
    a.in_buckets = n_bins;
    a.n_regions = n_regions;
    a.shift = region_shift;
    a.mask = (1u << region_shift) - 1u;
    a.buckets_per_seg = BB_REGIONS_PER_BIN;
    a.sl = {g.cap2, ovf, status, ovf_cap};
    hipLaunchKernelGGL((bloom_part_pieces_kernel<BB_L2_THREADS, false>), dim3(gx, n_bins), dim3(BB_L2_THREADS), lds2, c->stream, a);
  }
  const size_t apply_lds = (size_t)BB_REGION_DWORDS * sizeof(uint32_t);
  const uint32_t grid = n_regions < (uint32_t)c->n_cu ? n_regions : (uint32_t)c->n_cu;
  if (counters) {
    NTCHK(set_max_lds(c, count_apply_kernel, apply_lds));
    hipLaunchKernelGGL(count_apply_kernel, dim3(grid), dim3(BB_APPLY_THREADS), apply_lds, c->stream, (const uint32_t*)list2, (const uint32_t*)nullptr,
                       n_regions, d_table, table_dwords, g.cap2, (const uint32_t*)fill2, (const BloomStatus*)status, ovf_cap, gx,
                       (uint32_t)BB_REGIONS_PER_BIN);
    hipLaunchKernelGGL(bloom_overflow_apply_kernel<true>, dim3(c->n_cu), dim3(256), 0, c->stream, (const uint64_t*)ovf, (const BloomStatus*)status,
                       ovf_cap, d_table);
  } else {
    NTCHK(set_max_lds(c, bloom_apply_kernel, apply_lds));
    hipLaunchKernelGGL(bloom_apply_kernel, dim3(grid), dim3(BB_APPLY_THREADS), apply_lds, c->stream, (const uint32_t*)list2, (const uint32_t*)nullptr,
                       n_regions, d_table, table_dwords, g.cap2, (const uint32_t*)fill2, (const BloomStatus*)status, ovf_cap, gx,
                       (uint32_t)BB_REGIONS_PER_BIN);
    hipLaunchKernelGGL(bloom_overflow_apply_kernel<false>, dim3(c->n_cu), dim3(256), 0, c->stream, (const uint64_t*)ovf, (const BloomStatus*)status,
                       ovf_cap, d_table);
  }
  prof_end(c);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(c->h_small + 64, status, sizeof(BloomStatus), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  BloomStatus st;
  memcpy(&st, c->h_small + 64, sizeof st);
  *outcome = st.ovf_n > ovf_cap ? 2 : 0;
  if (*outcome == 2) c->bloom_slots_backoff = BB_SLOTS_BACKOFF;
  if (lost) *lost = st.lost;
  return NTHIP_OK;
}

// Fixed-length reads on the device through slots-mode rounds: *r0 is advanced past every round that went through, *sum
// by its k-mers; stops at the first round that did not (skewed values, no memory) -- the caller takes the reads from *r0 on
// through the exact lists.
int fused_slots_rounds(nthip_ctx* c, const nthip_reads* rd, uint32_t k, uint32_t m, uint32_t* d_table, uint64_t n_slots, bool counters,
                       uint64_t* r0, uint64_t* sum)
{
  const uint32_t len = rd->fixed_len, stride = rd->stride ? rd->stride : len;
  const uint64_t per_read = (uint64_t)(len - k + 1) * m;
  const uint64_t round = slots_round_values(c, (rd->n_reads - *r0) * per_read);
  const uint64_t reads_per_round = round / per_read;
  if (reads_per_round == 0) return NTHIP_OK;
  // (pieces mode: rounds as long as the memory allows -- every round less is a read-modify-write of the whole table less)
  const uint64_t reads_per_round_p = std::max<uint64_t>(reads_per_round, slots_round_values(c, (rd->n_reads - *r0) * per_read, 1ull << 33) / per_read);
  while (*r0 < rd->n_reads) {
    uint64_t nr = rd->n_reads - *r0 < reads_per_round_p ? rd->n_reads - *r0 : reads_per_round_p;
    BloomFusedSrc src = {(const uint8_t*)rd->seqs + *r0 * stride, nr, len, stride, k, m};
    int outcome = 0;
    uint64_t lost = 0;
    NTCHK(bloom_pieces_round(c, &src, nullptr, 0, d_table, n_slots, counters, &outcome, &lost)); // (round 5: two-level tables)
    if (outcome == 4) {
      nr = rd->n_reads - *r0 < reads_per_round ? rd->n_reads - *r0 : reads_per_round;
      src.n_reads = nr;
      NTCHK(bloom_slots_round(c, &src, nullptr, nr * per_read, d_table, n_slots, counters, &outcome, &lost));
    }
    if (outcome) return NTHIP_OK;
    *sum += nr * (uint64_t)(len - k + 1) - lost;
    *r0 += nr;
  }
  return NTHIP_OK;
}

// A device-resident hash stream through slots-mode rounds (the stream must not live in the context's list buffer):
// *done = the values that went through; the caller takes the rest through the exact lists.
// (expand_m > 1: d_hashes holds hashes()[0] of n_values INPUTS, level 1 makes the other values -- pieces mode only: *done stays
//  short of n_values when the table / the device is not for it, and the caller hashes the full stream)
int stream_slots_rounds(nthip_ctx* c, const uint64_t* d_hashes, uint64_t n_values, uint32_t* d_table, uint64_t n_slots, bool counters,
                        uint64_t* done, uint32_t expand_m = 1, uint64_t kmul = 0)
{
  *done = 0;
  const uint8_t* const h0 = (const uint8_t*)d_hashes;
  if (c->bloom_tmp && h0 < c->bloom_tmp + c->bloom_tmp_bytes && h0 + n_values * 8 > c->bloom_tmp) return NTHIP_OK;
  if (!bloom_slots_ok(c)) return NTHIP_OK;
  const uint64_t round = slots_round_values(c, n_values);
  // pieces mode counts per piece (32 bits each) and places in 64 bits: its rounds are as long as the memory allows (at most 2^33
  // values) -- every round less is one read-modify-write of the whole table less (config 4's seed pair on 5 M reads: 6.6 G values)
  const uint64_t round_p = slots_round_values(c, n_values * expand_m, 1ull << 33) / expand_m;
  while (*done < n_values) {
    uint64_t nn = n_values - *done < round_p ? n_values - *done : round_p;
    int outcome = 4;
    NTCHK(bloom_pieces_round(c, nullptr, d_hashes + *done, nn, d_table, n_slots, counters, &outcome, nullptr, expand_m, kmul)); // (round 5: two-level tables)
    if (outcome == 4 && expand_m > 1) break;
    if (outcome == 4) {
      nn = n_values - *done < round ? n_values - *done : round;
      NTCHK(bloom_slots_round(c, nullptr, d_hashes + *done, nn, d_table, n_slots, counters, &outcome));
    }
    if (outcome) break;
    *done += nn;
  }
  return NTHIP_OK;
}

// the reads entry on the binned insert: rounds of reads hashed to a stream in the context's lists, every round binned
// Reads of any lengths (offsets != NULL) reach the consumers through their compact hash stream: the whole batch hashed in
// ONE round into a scratch buffer that `keep` owns (a k-mer starts at a base: at most total_bytes k-mers), then the
// stream forms of the consumers.  NTHIP_ERR_UNSUPPORTED when the stream does not fit the device: split the batch.
int run_kmer_bloom_binned(nthip_ctx* c, const nthip_reads* rd, uint16_t k, uint8_t m, uint32_t* d_filter, uint64_t n_bits,
                          uint64_t* total_out, uint32_t flags)
{
  const uint32_t len = rd->fixed_len, stride = rd->stride ? rd->stride : len;
  const uint64_t per_read = (uint64_t)(len - k + 1) * m;
  uint64_t sum = 0, first = 0;
  const uint32_t n_regions_all = (uint32_t)((n_bits + (1ull << BB_REGION_SHIFT) - 1) >> BB_REGION_SHIFT);
  // round 4, later: slots mode -- the reads hashed once, no histogram (bloom_binned_kernels.hpp)
  if (!(flags & NTHIP_HOST_INPUT) && bloom_fused_ok(c, {(const uint8_t*)rd->seqs, rd->n_reads, len, stride, k, m}, n_regions_all) &&
      bloom_slots_ok(c))
    NTCHK(fused_slots_rounds(c, rd, k, m, d_filter, n_bits, false, &first, &sum));
  if (first == rd->n_reads) {
    if (total_out) *total_out = sum;
    return NTHIP_OK;
  }
  uint64_t round = bloom_round_values(c, (rd->n_reads - first) * per_read, true);
  // (1: the caller takes the fused / atomic kernel for the WHOLE batch -- also when some rounds have gone through already:
  // setting a bit twice is setting it once, so a call that only ran short of memory half way finishes there -- ADVICE r04)
  if (round < per_read) return 1;
  round = round / per_read * per_read;
  BloomLists t;
  {
    const int rc = bloom_lists(c, &round, true, &t);
    if (rc) return rc; // (1: no memory for the lists -- the caller takes the fused kernel)
  }
  const uint64_t reads_per_round = round / per_read;
  if (reads_per_round == 0) return 1;
  for (uint64_t r0 = first; r0 < rd->n_reads; r0 += reads_per_round) {
    const uint64_t nr = rd->n_reads - r0 < reads_per_round ? rd->n_reads - r0 : reads_per_round;
    nthip_reads part = *rd;
    part.seqs = rd->seqs + r0 * stride;
    part.n_reads = nr;
    // round 4: no hash stream at all when the reads are on the device and hold bases only -- hashed twice, counted in LDS,
    // partitioned from the registers (bloom_fused_kernels.hpp: 16 B of list traffic per value instead of 40)
    const BloomFusedSrc src = {(const uint8_t*)part.seqs, nr, len, stride, k, m};
    const uint32_t n_regions = (uint32_t)((n_bits + (1ull << BB_REGION_SHIFT) - 1) >> BB_REGION_SHIFT);
    if (!(flags & NTHIP_HOST_INPUT) && bloom_fused_ok(c, src, n_regions)) {
      uint64_t lost = 0;
      NTCHK(bloom_fused_lost_reset(c));
      NTCHK(bloom_fused_count(c, src, n_bits, t, false));
      const uint64_t total = nr * (uint64_t)(len - k + 1);
      NTCHK(bloom_binned_round(c, nullptr, total * m, d_filter, n_bits, t, false, &src));
      NTCHK(bloom_fused_lost_read(c, &lost));
      sum += total - lost;
      continue;
    }
    nthip_out out;
    memset(&out, 0, sizeof out);
    out.hashes = t.hashes;
    out.capacity = nr * (uint64_t)(len - k + 1);
    uint64_t total = 0;
    NTCHK(nthip_kmer_hash(c, &part, k, m, &out, &total, flags & NTHIP_HOST_INPUT));
    sum += total;
    if (total) NTCHK(bloom_binned_round(c, t.hashes, total * m, d_filter, n_bits, t));
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  if (total_out) *total_out = sum;
  return NTHIP_OK;
}

// shared body: SINK_BLOOM_INSERT or SINK_BLOOM_QUERY over fixed-length reads
int run_kmer_bloom(nthip_ctx* c, const nthip_reads* rd, uint16_t k16, uint8_t m8, uint32_t* d_filter, uint64_t n_bits,
                   uint64_t* hits, uint64_t* total_out, uint64_t* total_hits, uint32_t flags, bool query)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  NTCHK(check_reads(rd));
  const uint32_t k = k16, m = m8;
  if (k == 0) return fail(NTHIP_ERR_ARG, "k must be greater than 0");
  if (k < 3) return fail(NTHIP_ERR_UNSUPPORTED, "k < 3 is undefined in the reference (src/kmer.cpp:47)");
  if (m == 0) return fail(NTHIP_ERR_UNSUPPORTED, "num_hashes must be >= 1");
  if (!d_filter || n_bits == 0) return fail(NTHIP_ERR_ARG, "filter is NULL / n_bits is 0");
  if ((uintptr_t)d_filter & 3u) return fail(NTHIP_ERR_ARG, "filter must be 4-byte aligned");
  HIPCHK(hipSetDevice(c->device));
  if (total_out) *total_out = 0;
  if (total_hits) *total_hits = 0;
  if (rd->offsets) { // reads of any lengths: the compact stream of a round of them, then the stream forms
    if (rd->n_reads == 0) return NTHIP_OK;
    uint64_t sum_kmers = 0, sum_hits = 0;
    NTCHK(offsets_in_rounds(c, rd, flags, (size_t)8 * m + (query ? (size_t)26 * m : 16), [&](const nthip_reads* part, uint64_t r0, uint64_t bases) -> int {
      Staged keep;
      uint64_t *d_h = nullptr, *d_counts = nullptr, n_kmers = 0;
      // several hashes per k-mer into / against a two-level filter: the round's stream holds hashes()[0] only, the first partition
      // level makes the others (stream_bloom_insert_expand / stream_hits_per_read: an m times shorter stream written and read back)
      // (only when the road that consumes such a stream will take the batch -- the same predicates as stream_bloom_insert_expand /
      //  stream_query_binned: a batch they decline would be hashed with one hash per k-mer, then again in full)
      const uint64_t est_values = (bases ? bases : 1) * (uint64_t)m;
      const bool expand = m >= 2 && m <= 4 && n_bits > (1ull << 27) && c->tune.bloom_pieces != 2 &&
                          (query ? c->tune.bloom_query != 2 && (c->tune.bloom_query == 1 || (est_values >= (1ull << 24) && (n_bits >> 3) >= (32ull << 20) &&
                                                                                             est_values >= (n_bits >> 3) / 32))
                                 : bloom_binned_ok(c, d_filter, n_bits, est_values));
      NTCHK(stream_of_offsets(c, part, k16, expand ? (uint8_t)1 : m8, flags, keep, &d_h, query ? &d_counts : nullptr, &n_kmers, bases ? bases : 1));
      sum_kmers += n_kmers;
      if (!query) {
        bool done = false;
        if (expand && n_kmers) NTCHK(stream_bloom_insert_expand(c, d_h, n_kmers, m, (uint64_t)k * MULTISEED, d_filter, n_bits, &done));
        if (done || n_kmers == 0) return NTHIP_OK;
        if (expand) { // (not a batch / a device for it: the full stream)
          uint64_t again = 0;
          NTCHK(stream_of_offsets(c, part, k16, m8, flags, keep, &d_h, nullptr, &again, bases ? bases : 1));
        }
        return nthip_stream_bloom_insert(c, d_h, n_kmers * m, (uint8_t*)d_filter, n_bits);
      }
      const uint64_t nr = part->n_reads;
      const bool host = hits && (flags & NTHIP_HOST_OUTPUT);
      uint64_t* d_hits = hits ? hits + r0 : nullptr;
      if (host) NTCHK(own_alloc(keep, (size_t)nr * 8, (void**)&d_hits));
      uint64_t *d_roff = nullptr, *d_sums = nullptr;
      NTCHK(own_alloc(keep, (size_t)(nr + 1) * 8, (void**)&d_roff));
      NTCHK(own_alloc(keep, (size_t)(nr / SCAN_TILE + 64) * 8, (void**)&d_sums));
      NTCHK(device_exclusive_scan(c, d_counts, d_roff, nr, d_sums, (uint64_t*)(c->d_small + 16)));
      HIPCHK(hipMemsetAsync(c->d_small + 24, 0, 8, c->stream));
      bool expanded = false;
      if (expand)
        NTCHK(stream_hits_per_read(c, d_h, d_roff, nr, n_kmers, m, d_filter, n_bits, d_hits, (unsigned long long*)(c->d_small + 24),
                                   "stream_bloom_query_kernel", m, (uint64_t)k * MULTISEED, &expanded));
      if (!expanded) {
        if (expand) {
          uint64_t again = 0;
          NTCHK(stream_of_offsets(c, part, k16, m8, flags, keep, &d_h, nullptr, &again, bases ? bases : 1));
        }
        NTCHK(stream_hits_per_read(c, d_h, d_roff, nr, n_kmers, m, d_filter, n_bits, d_hits, (unsigned long long*)(c->d_small + 24),
                                   "stream_bloom_query_kernel"));
      }
      HIPCHK(hipMemcpyAsync(c->h_small + 24, c->d_small + 24, 8, hipMemcpyDeviceToHost, c->stream));
      if (host) HIPCHK(hipMemcpyAsync(hits + r0, d_hits, nr * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      uint64_t h = 0;
      memcpy(&h, c->h_small + 24, 8);
      sum_hits += h;
      return NTHIP_OK;
    }));
    if (total_out) *total_out = sum_kmers;
    if (total_hits) *total_hits = sum_hits;
    return NTHIP_OK;
  }
  const uint32_t len = rd->fixed_len, stride = rd->stride ? rd->stride : len;
  const bool host_hits = query && hits && (flags & NTHIP_HOST_OUTPUT);
  if (rd->n_reads == 0) return NTHIP_OK;
  if (len < k) {
    if (query && hits) {
      if (host_hits) memset(hits, 0, rd->n_reads * sizeof(uint64_t));
      else HIPCHK(hipMemsetAsync(hits, 0, rd->n_reads * sizeof(uint64_t), c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
    }
    return NTHIP_OK;
  }
  // a batch that is large next to the filter: hash stream + binned insert, no device atomics
  if (!query && stride >= len && bloom_binned_ok(c, d_filter, n_bits, rd->n_reads * (uint64_t)(len - k + 1) * m)) {
    const int rc = run_kmer_bloom_binned(c, rd, k16, m8, d_filter, n_bits, total_out, flags);
    if (rc != 1) return rc; // (1: the lists do not fit the device right now)
  }
  // round 5: the query of a large batch against a large filter goes region by region as the insert does, the answers
  // finding their way back to the reads (bloom_query_kernels.hpp); what that does not take -- small batches, filters that
  // sit in the caches, host input, a round of skewed values -- keeps the kernel below, one filter load per k-mer
  uint64_t first = 0, q_kmers = 0, q_hits = 0;
  uint64_t* d_hits = hits;
  Staged st;
  if (host_hits) {
    HIPCHK(hipMalloc((void**)&d_hits, rd->n_reads * sizeof(uint64_t)));
    st.owned.push_back(d_hits);
  }
  NaPlan plan;
  const bool plan_ok = kmer_na_plan(c, len, stride, k, m, /*want_pos (the k-mer's read)*/ query, &plan);
  // (a shape the direct kernels below do not take is refused BEFORE the binned query writes anything: it may stop early --
  //  skewed values, memory -- and would leave the rest of the reads without a kernel, the hits half written)
  if (query && !plan_ok) return fail(NTHIP_ERR_UNSUPPORTED, "shape outside the fused consumer kernels (k <= 64, m <= 8, stride >= windows)");
  if (query && !(flags & NTHIP_HOST_INPUT)) {
    NTCHK(bloom_query_binned(c, rd, k, m, d_filter, n_bits, BQ_BLOOM, d_hits, nullptr, &first, &q_kmers, &q_hits));
    if (first == rd->n_reads) {
      if (host_hits) HIPCHK(hipMemcpyAsync(hits, d_hits, rd->n_reads * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      if (total_out) *total_out = q_kmers;
      if (total_hits) *total_hits = q_hits;
      return NTHIP_OK;
    }
  }
  nthip_reads part = *rd;
  part.seqs = rd->seqs + first * stride;
  part.n_reads = rd->n_reads - first;
  if (!plan_ok) return fail(NTHIP_ERR_UNSUPPORTED, "shape outside the fused consumer kernels (k <= 64, m <= 8, stride >= windows)");
  uint64_t total_bytes = 0;
  NTCHK(reads_total_bytes(c, &part, flags, &total_bytes));
  NTCHK(stage_inputs(c, &part, flags, total_bytes, st));
  if (query && d_hits) HIPCHK(hipMemsetAsync(d_hits + first, 0, part.n_reads * sizeof(uint64_t), c->stream));
  KmerFixedArgs consts;
  memset(&consts, 0, sizeof consts);
  fill_kmer_consts(k, m, consts);
  KmerRunsGenArgs a;
  fill_gen_args(a, c, st, &part, k, m, plan.g, consts);
  NTCHK(get_kmer_tab(c, k, &a.init_tab));
  a.hashes = nullptr;
  a.vbits_dwords = plan.vbits_dwords;
  a.ptile_dwords = plan.ptile_dwords;
  a.tile_u64 = plan.tile_u64;
  a.waves = plan.waves;
  a.bloom = d_filter;
  a.n_bits = n_bits;
  a.bloom_magic = bloom_magic_of(n_bits);
  a.hits = query && d_hits ? d_hits + first : nullptr;
  a.sink_totals = (uint64_t*)(c->d_small + 16);
  HIPCHK(hipMemsetAsync(c->d_small + 16, 0, 16, c->stream));
  if (query) NTCHK((launch_kmer_runs_gen_nw<true, SINK_BLOOM_QUERY>(c, a, plan.lds, plan.g.nw, plan.g.dword_tail != 0)));
  else NTCHK((launch_kmer_runs_gen_nw<true, SINK_BLOOM_INSERT>(c, a, plan.lds, plan.g.nw, plan.g.dword_tail != 0)));
  HIPCHK(hipMemcpyAsync(c->h_small + 16, c->d_small + 16, 16, hipMemcpyDeviceToHost, c->stream));
  if (host_hits) HIPCHK(hipMemcpyAsync(hits, d_hits, rd->n_reads * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  uint64_t tot[2];
  memcpy(tot, c->h_small + 16, 16);
  if (total_out) *total_out = tot[0] + q_kmers;
  if (total_hits) *total_hits = tot[1] + q_hits;
  return NTHIP_OK;
}

} // namespace

extern "C" int nthip_kmer_bloom_insert(nthip_ctx* c, const nthip_reads* rd, uint16_t k, uint8_t m, uint8_t* d_filter,
                                       uint64_t n_bits, uint64_t* total, uint32_t flags)
{
  return run_kmer_bloom(c, rd, k, m, (uint32_t*)d_filter, n_bits, nullptr, total, nullptr, flags, false);
}

extern "C" int nthip_kmer_bloom_query(nthip_ctx* c, const nthip_reads* rd, uint16_t k, uint8_t m,
                                      const uint8_t* d_filter, uint64_t n_bits, uint64_t* hits, uint64_t* total,
                                      uint64_t* total_hits, uint32_t flags)
{
  return run_kmer_bloom(c, rd, k, m, (uint32_t*)d_filter, n_bits, hits, total, total_hits, flags, true);
}

// The stream insert for a stream that holds hashes()[0] only (n_inputs of them; the filter gets expand_m values of each: level 1
// makes the others).  *done = false (filter untouched or touched in part -- setting a bit twice is harmless): not a table / a
// device for it, the caller hashes the full stream and takes nthip_stream_bloom_insert.  Waits for the stream.
bool ntamd::host::bloom_binned_applies(const nthip_ctx* c, const void* d_filter, uint64_t n_bits, uint64_t n_values)
{
  return bloom_binned_ok(c, d_filter, n_bits, n_values);
}
int ntamd::host::stream_bloom_insert_expand(nthip_ctx* c, const uint64_t* d_h0, uint64_t n_inputs, uint32_t expand_m, uint64_t kmul, uint32_t* d_filter,
                                            uint64_t n_bits, bool* done)
{
  *done = false;
  if (expand_m < 2 || expand_m > 4 || n_inputs == 0) return NTHIP_OK;
  if (!bloom_binned_ok(c, d_filter, n_bits, n_inputs * expand_m)) return NTHIP_OK;
  uint64_t got = 0;
  NTCHK(stream_slots_rounds(c, d_h0, n_inputs, d_filter, n_bits, false, &got, expand_m, kmul));
  HIPCHK(hipStreamSynchronize(c->stream));
  *done = got == n_inputs;
  return NTHIP_OK;
}

extern "C" int nthip_stream_bloom_insert(nthip_ctx* c, const uint64_t* d_hashes, uint64_t n_values, uint8_t* d_filter,
                                         uint64_t n_bits)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  if (!d_filter || n_bits == 0) return fail(NTHIP_ERR_ARG, "filter is NULL / n_bits is 0");
  if ((uintptr_t)d_filter & 3u) return fail(NTHIP_ERR_ARG, "filter must be 4-byte aligned");
  if (n_values && !d_hashes) return fail(NTHIP_ERR_ARG, "hashes is NULL");
  HIPCHK(hipSetDevice(c->device));
  if (n_values == 0) return NTHIP_OK;
  if (bloom_binned_ok(c, d_filter, n_bits, n_values)) {
    uint64_t done = 0;
    NTCHK(stream_slots_rounds(c, d_hashes, n_values, (uint32_t*)d_filter, n_bits, false, &done));
    if (done == n_values) return NTHIP_OK;
    d_hashes += done;
    n_values -= done;
    uint64_t round = bloom_round_values(c, n_values, false);
    BloomLists t;
    const int lrc = bloom_lists(c, &round, false, &t);
    if (lrc < 0) return lrc;
    if (lrc == 0) {
      for (uint64_t v0 = 0; v0 < n_values; v0 += round)
        NTCHK(bloom_binned_round(c, d_hashes + v0, n_values - v0 < round ? n_values - v0 : round, (uint32_t*)d_filter, n_bits, t));
      HIPCHK(hipStreamSynchronize(c->stream));
      return NTHIP_OK;
    } // (1: no memory for the lists: the atomic kernel below)
  }
  prof_begin(c, "stream_bloom_insert_kernel");
  hipLaunchKernelGGL(stream_bloom_insert_kernel, dim3(c->n_cu * 8), dim3(256), 0, c->stream, d_hashes, n_values,
                     (uint32_t*)d_filter, n_bits, bloom_magic_of(n_bits));
  prof_end(c);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  return NTHIP_OK;
}

// ---- k-mer counting sketch ("k-mer counting sketch") -------------------------------------------------
// A count-min sketch of one-byte counters: every hash value adds one to counter `h mod n_counters`, saturating at 255; the
// estimate of a k-mer is the smallest of its m counters.  Plain count-min (not the order-dependent "increment the
// minimum" variant): the result does not depend on the order of the values, so it is compared byte for byte with the
// table built on the CPU from the oracle's stream.  Large batches go through the lists of the binned Bloom insert
// (regions of 2^15 counters, 32-bit tallies in LDS, one saturating read-modify-write of the touched dwords); small ones
// and sketches of more than 2^30 counters take a compare-and-swap per value.
namespace {
int count_insert_stream(nthip_ctx* c, const uint64_t* d_hashes, uint64_t n_values, uint32_t* d_counters, uint64_t n_counters,
                        const BloomLists* have)
{
  if (n_values == 0) return NTHIP_OK;
  if (bloom_binned_ok(c, d_counters, n_counters, n_values, true)) {
    if (!have) {
      uint64_t done = 0;
      NTCHK(stream_slots_rounds(c, d_hashes, n_values, d_counters, n_counters, true, &done));
      if (done == n_values) return NTHIP_OK;
      d_hashes += done;
      n_values -= done;
    }
    BloomLists t;
    uint64_t round = n_values;
    int lrc = 0;
    if (have) {
      t = *have; // (the reads entry: its round already fits the lists)
    } else {
      round = bloom_round_values(c, n_values, false);
      lrc = bloom_lists(c, &round, false, &t);
      if (lrc < 0) return lrc;
    }
    if (lrc == 0) {
      for (uint64_t v0 = 0; v0 < n_values; v0 += round)
        NTCHK(bloom_binned_round(c, d_hashes + v0, n_values - v0 < round ? n_values - v0 : round, d_counters, n_counters, t, true));
      return NTHIP_OK;
    } // (1: no memory for the lists: one compare-and-swap per value below)
  }
  prof_begin(c, "count_atomic_kernel");
  hipLaunchKernelGGL(count_atomic_kernel, dim3(c->n_cu * 8), dim3(256), 0, c->stream, d_hashes, n_values, d_counters, n_counters,
                     bloom_magic_of(n_counters));
  prof_end(c);
  HIPCHK(hipGetLastError());
  return NTHIP_OK;
}
int check_sketch(const void* d_counters, uint64_t n_counters)
{
  if (!d_counters || n_counters == 0) return fail(NTHIP_ERR_ARG, "sketch is NULL / n_counters is 0");
  if ((uintptr_t)d_counters & 3u) return fail(NTHIP_ERR_ARG, "sketch must be 4-byte aligned");
  if (n_counters & 3u) return fail(NTHIP_ERR_ARG, "n_counters must be a multiple of 4 (the counters are updated a dword at a time)");
  return NTHIP_OK;
}
} // namespace

extern "C" int nthip_stream_bloom_query(nthip_ctx* c, const uint64_t* d_hashes, uint64_t n_kmers, uint8_t m, const uint8_t* d_filter,
                                        uint64_t n_bits, uint8_t* d_flags, uint64_t* found)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  if (!d_filter || n_bits == 0) return fail(NTHIP_ERR_ARG, "filter is NULL / n_bits is 0");
  if ((uintptr_t)d_filter & 3u) return fail(NTHIP_ERR_ARG, "filter must be 4-byte aligned");
  if (m == 0) return fail(NTHIP_ERR_UNSUPPORTED, "num_hashes must be >= 1");
  if (n_kmers && (!d_hashes || !d_flags)) return fail(NTHIP_ERR_ARG, "hashes / flags is NULL");
  HIPCHK(hipSetDevice(c->device));
  if (found) *found = 0;
  if (n_kmers == 0) return NTHIP_OK;
  HIPCHK(hipMemsetAsync(c->d_small + 24, 0, 8, c->stream));
  { // a large stream against a large filter: region by region (capi_sink_query.hip); m == 1: the answers ARE the flags
    const uint64_t n_values = n_kmers * m;
    bool done = false;
    uint8_t* d_ans = d_flags;
    if (!stream_query_applies(c, d_filter, n_bits, 0, n_values) || (m > 1 && kept_alloc(c, KEPT_ANSWERS, n_values + 8, (void**)&d_ans) != NTHIP_OK)) {
      d_ans = nullptr;
    }
    if (d_ans) {
      int rc = stream_query_binned(c, d_hashes, n_values, (const uint32_t*)d_filter, n_bits, 0, d_ans, &done);
      if (rc == NTHIP_OK && done) {
        prof_begin(c, "answers_per_kmer_kernel");
        rc = answers_per_kmer(c, d_ans, n_kmers, m, 0, d_flags, (unsigned long long*)(c->d_small + 24));
        prof_end(c);
      }
      if (rc == NTHIP_OK && done) HIPCHK(hipMemcpyAsync(c->h_small + 24, c->d_small + 24, 8, hipMemcpyDeviceToHost, c->stream));
      (void)hipStreamSynchronize(c->stream);
      NTCHK(rc);
      if (done) {
        if (found) memcpy(found, c->h_small + 24, 8);
        return NTHIP_OK;
      }
    }
  }
  prof_begin(c, "stream_bloom_flags_kernel");
  hipLaunchKernelGGL(stream_bloom_flags_kernel, dim3(c->n_cu * 8), dim3(256), 0, c->stream, d_hashes, n_kmers, (uint32_t)m,
                     (const uint32_t*)d_filter, n_bits, bloom_magic_of(n_bits), d_flags, (unsigned long long*)(c->d_small + 24));
  prof_end(c);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(c->h_small + 24, c->d_small + 24, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (found) memcpy(found, c->h_small + 24, 8);
  return NTHIP_OK;
}

extern "C" int nthip_stream_count_insert(nthip_ctx* c, const uint64_t* d_hashes, uint64_t n_values, uint8_t* d_counters,
                                         uint64_t n_counters)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  NTCHK(check_sketch(d_counters, n_counters));
  if (n_values && !d_hashes) return fail(NTHIP_ERR_ARG, "hashes is NULL");
  HIPCHK(hipSetDevice(c->device));
  NTCHK(count_insert_stream(c, d_hashes, n_values, (uint32_t*)d_counters, n_counters, nullptr));
  HIPCHK(hipStreamSynchronize(c->stream));
  return NTHIP_OK;
}

extern "C" int nthip_kmer_count_insert(nthip_ctx* c, const nthip_reads* rd, uint16_t k, uint8_t m, uint8_t* d_counters,
                                       uint64_t n_counters, uint64_t* total_out, uint32_t flags)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  NTCHK(check_reads(rd));
  NTCHK(check_sketch(d_counters, n_counters));
  if (k == 0) return fail(NTHIP_ERR_ARG, "k must be greater than 0");
  if (k < 3) return fail(NTHIP_ERR_UNSUPPORTED, "k < 3 is undefined in the reference (src/kmer.cpp:47)");
  if (m == 0) return fail(NTHIP_ERR_UNSUPPORTED, "num_hashes must be >= 1");
  HIPCHK(hipSetDevice(c->device));
  if (total_out) *total_out = 0;
  if (rd->offsets) { // reads of any lengths: the compact stream of a round of them, then the stream form
    if (rd->n_reads == 0) return NTHIP_OK;
    uint64_t sum_kmers = 0;
    NTCHK(offsets_in_rounds(c, rd, flags, (size_t)8 * m + 16, [&](const nthip_reads* part, uint64_t, uint64_t bases) -> int {
      Staged keep;
      uint64_t *d_h = nullptr, n_kmers = 0;
      NTCHK(stream_of_offsets(c, part, k, m, flags, keep, &d_h, nullptr, &n_kmers, bases ? bases : 1));
      sum_kmers += n_kmers;
      return nthip_stream_count_insert(c, d_h, n_kmers * m, d_counters, n_counters);
    }));
    if (total_out) *total_out = sum_kmers;
    return NTHIP_OK;
  }
  const uint32_t len = rd->fixed_len, stride = rd->stride ? rd->stride : len;
  if (rd->n_reads == 0 || len < k) return NTHIP_OK;
  const uint64_t per_read = (uint64_t)(len - k + 1) * m;
  uint64_t sum = 0, first = 0;
  {
    // slots mode first (bloom_binned_kernels.hpp): the reads hashed once, no histogram
    const uint32_t n_regions_all = (uint32_t)((n_counters + (1ull << CS_REGION_SHIFT) - 1) >> CS_REGION_SHIFT);
    if (!(flags & NTHIP_HOST_INPUT) && stride >= len && bloom_binned_ok(c, d_counters, n_counters, rd->n_reads * per_read, true) &&
        bloom_fused_ok(c, {(const uint8_t*)rd->seqs, rd->n_reads, len, stride, k, m}, n_regions_all) && bloom_slots_ok(c))
      NTCHK(fused_slots_rounds(c, rd, k, m, (uint32_t*)d_counters, n_counters, true, &first, &sum));
    if (first == rd->n_reads) {
      if (total_out) *total_out = sum;
      return NTHIP_OK;
    }
  }
  uint64_t round = bloom_round_values(c, (rd->n_reads - first) * per_read, true);
  if (round < per_read) round = per_read;
  round = round / per_read * per_read;
  BloomLists t;
  {
    const int rc = bloom_lists(c, &round, true, &t);
    if (rc < 0) return rc;
    if (rc == 1) return fail(NTHIP_ERR_HIP, "no device memory for a round of the counting sketch (%llu values)", (unsigned long long)round);
  }
  const uint64_t reads_per_round = round / per_read;
  if (reads_per_round == 0) return fail(NTHIP_ERR_UNSUPPORTED, "reads too long for the counting sketch's rounds");
  for (uint64_t r0 = first; r0 < rd->n_reads; r0 += reads_per_round) {
    const uint64_t nr = rd->n_reads - r0 < reads_per_round ? rd->n_reads - r0 : reads_per_round;
    nthip_reads part = *rd;
    part.seqs = rd->seqs + r0 * stride;
    part.n_reads = nr;
    // (as the Bloom insert: device-resident reads of bases only are hashed twice and never streamed)
    const BloomFusedSrc src = {(const uint8_t*)part.seqs, nr, len, stride, k, m};
    const uint32_t n_regions = (uint32_t)((n_counters + (1ull << CS_REGION_SHIFT) - 1) >> CS_REGION_SHIFT);
    const uint64_t dense = nr * (uint64_t)(len - k + 1);
    if (!(flags & NTHIP_HOST_INPUT) && stride >= len && bloom_binned_ok(c, d_counters, n_counters, dense * m, true) &&
        bloom_fused_ok(c, src, n_regions)) {
      uint64_t lost = 0;
      NTCHK(bloom_fused_lost_reset(c));
      NTCHK(bloom_fused_count(c, src, n_counters, t, true));
      NTCHK(bloom_binned_round(c, nullptr, dense * m, (uint32_t*)d_counters, n_counters, t, true, &src));
      NTCHK(bloom_fused_lost_read(c, &lost));
      sum += dense - lost;
      continue;
    }
    nthip_out out;
    memset(&out, 0, sizeof out);
    out.hashes = t.hashes;
    out.capacity = nr * (uint64_t)(len - k + 1);
    uint64_t total = 0;
    NTCHK(nthip_kmer_hash(c, &part, k, m, &out, &total, flags & NTHIP_HOST_INPUT));
    sum += total;
    NTCHK(count_insert_stream(c, t.hashes, total * m, (uint32_t*)d_counters, n_counters, &t));
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  if (total_out) *total_out = sum;
  return NTHIP_OK;
}

extern "C" int nthip_stream_count_query(nthip_ctx* c, const uint64_t* d_hashes, uint64_t n_kmers, uint8_t m,
                                        const uint8_t* d_counters, uint64_t n_counters, uint8_t* d_estimates)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  NTCHK(check_sketch(d_counters, n_counters));
  if (m == 0) return fail(NTHIP_ERR_UNSUPPORTED, "num_hashes must be >= 1");
  if (n_kmers && (!d_hashes || !d_estimates)) return fail(NTHIP_ERR_ARG, "hashes / estimates is NULL");
  HIPCHK(hipSetDevice(c->device));
  if (n_kmers == 0) return NTHIP_OK;
  { // a large stream against a large sketch: region by region; m == 1: the answers ARE the estimates
    const uint64_t n_values = n_kmers * m;
    bool done = false;
    uint8_t* d_ans = d_estimates;
    if (!stream_query_applies(c, d_counters, n_counters, 1, n_values) || (m > 1 && kept_alloc(c, KEPT_ANSWERS, n_values + 8, (void**)&d_ans) != NTHIP_OK)) {
      d_ans = nullptr;
    }
    if (d_ans) {
      int rc = stream_query_binned(c, d_hashes, n_values, (const uint32_t*)d_counters, n_counters, 1, d_ans, &done);
      if (rc == NTHIP_OK && done && m > 1) {
        prof_begin(c, "answers_per_kmer_kernel");
        rc = answers_per_kmer(c, d_ans, n_kmers, m, 1, d_estimates, nullptr);
        prof_end(c);
      }
      (void)hipStreamSynchronize(c->stream);
      NTCHK(rc);
      if (done) return NTHIP_OK;
    }
  }
  prof_begin(c, "count_query_kernel");
  hipLaunchKernelGGL(count_query_kernel, dim3(c->n_cu * 8), dim3(256), 0, c->stream, d_hashes, n_kmers, (uint32_t)m, d_counters,
                     n_counters, bloom_magic_of(n_counters), d_estimates);
  prof_end(c);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  return NTHIP_OK;
}
