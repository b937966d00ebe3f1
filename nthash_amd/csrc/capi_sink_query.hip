// capi_sink_query.hip -- the binned read side of the Bloom filter / counting sketch (bloom_query_kernels.hpp), and
// nthip_kmer_count_query
// Part of libnthash_hip.so (include/nthash_hip.h); see capi_internal.hpp for the file map.
#include "capi_internal.hpp"

#include <algorithm>

#include "bloom_host.hpp"
#include "bloom_query_kernels.hpp"
#include "util_kernels.hpp" // (SCAN_TILE)

using namespace ntamd;
using namespace ntamd::host;

namespace {

constexpr uint64_t BQ_PIECES_ROUND_MAX = 1ull << 33; // pieces mode: counts per piece (32 bits each), every place in 64 bits -- as long as the memory allows
constexpr uint64_t BQ_ROUND_MAX = 0xC0000000ull; // values per round (list positions are 64-bit; piece counts, overflow indices and tile rows fit 32 bits)
constexpr uint32_t BQ_L2_THREADS = BB_L2_THREADS, BQ_L2_TILE = BQ_L2_THREADS * BB_PART_ITEMS;

struct QueryGeo {
  uint32_t region_shift = 0, bin_shift = 0, n_regions = 0, n_bins = 0;
  bool one = false; // a table of one bin: level 1 goes straight to the regions
};
bool query_geo(uint64_t n_slots, int kind, QueryGeo* g)
{
  g->region_shift = kind == BQ_BLOOM ? BB_REGION_SHIFT : BQ_COUNT_REGION_SHIFT;
  g->bin_shift = g->region_shift + 7u;
  const uint64_t nr = (n_slots + (1ull << g->region_shift) - 1) >> g->region_shift;
  if (nr > BB_MAX_REGIONS) return false;
  g->n_regions = (uint32_t)nr;
  g->n_bins = (g->n_regions + BB_REGIONS_PER_BIN - 1) / BB_REGIONS_PER_BIN;
  g->one = g->n_bins == 1;
  return true;
}

struct QueryScratch {
  BloomStatus* status = nullptr;
  unsigned long long* total_hits = nullptr;
  uint32_t *cur1 = nullptr, *cur2 = nullptr, *list1 = nullptr, *list2 = nullptr;
  uint64_t* ovf = nullptr;
  uint16_t *where1 = nullptr, *where2 = nullptr; // places in the tiles' sorted order (2 B per value and level)
  uint32_t *tovf1 = nullptr, *tovf2 = nullptr;
  uint2 *tab1 = nullptr, *tab2 = nullptr;
  uint8_t *pay1 = nullptr, *pay2 = nullptr, *ovf_pay = nullptr;
  uint16_t* surv = nullptr; // m > 1 in passes: per tile, emitting word and thread, the windows whose hashes so far all hit
  uint64_t cap1 = 0, cap2 = 0, ovf_cap = 0;
  uint32_t tiles_per_seg = 0; // slots mode: tile rows per bin; pieces mode: per piece
  size_t head_bytes = 0;
  // pieces mode (two-level tables; bloom_binned_kernels.hpp): cur1 / cur2 are the fill arrays of the pieces
  bool pieces = false;
  PiecesGeo pg;
};
inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

// the scratch of a round of nr reads (n values) carved out of the context's list buffer; *need (always) = its size;
// returns false when the buffer is smaller (nothing carved)
// pieces mode? (a two-level table, the level-1 tile + its waiting lines fit the LDS); *gx = blocks per bin of level 2
bool query_pieces_ok(nthip_ctx* c, const QueryGeo& g, const BloomFusedSrc& shape, uint32_t* gx)
{
  if (g.one || c->tune.bloom_pieces == 2) return false;
  if (bloom_fused_lds(shape, 1024u, 1024u * 16u + BB_PIECES_LDS_DWORDS) > lds_cap_of(c) - 4096) return false;
  const size_t lds2 = ((size_t)BQ_L2_TILE + (size_t)BB_REGIONS_PER_BIN * 32u) * sizeof(uint32_t);
  int per_cu = 1;
  if (blocks_per_cu(c, bloom_part_pieces_kernel<BQ_L2_THREADS, true>, (int)BQ_L2_THREADS, lds2, &per_cu) != NTHIP_OK) return false;
  const uint32_t grid = 2u * (uint32_t)c->n_cu * (uint32_t)per_cu;
  *gx = grid / g.n_bins ? grid / g.n_bins : 1u;
  return true;
}

bool query_scratch(nthip_ctx* c, const QueryGeo& g, uint64_t nr, uint64_t n, uint64_t n_slots, uint32_t steps, uint32_t m, QueryScratch* q,
                   size_t* need, bool pieces, uint32_t gx, uint64_t values_per_read)
{
  q->pieces = pieces;
  size_t slots1, slots2, fills1, fills2;
  uint64_t rows2;
  if (pieces) {
    pieces_geo(c, nr, values_per_read, n_slots, g.region_shift, gx, &q->pg);
    q->cap1 = q->pg.cap1;
    q->cap2 = q->pg.cap2;
    q->tiles_per_seg = (uint32_t)((q->cap1 + BQ_L2_TILE - 1) / BQ_L2_TILE);
    slots1 = (size_t)g.n_bins * q->pg.g1 * q->cap1;
    slots2 = (size_t)g.n_regions * gx * q->cap2;
    fills1 = (size_t)q->pg.g1 * g.n_bins;
    fills2 = (size_t)g.n_bins * gx * BB_REGIONS_PER_BIN;
    rows2 = (uint64_t)g.n_bins * q->pg.g1 * q->tiles_per_seg;
  } else {
    q->cap1 = g.one ? 0 : slot_cap(c, n, 1ull << g.bin_shift, n_slots);
    q->cap2 = slot_cap(c, n, 1ull << g.region_shift, n_slots);
    q->tiles_per_seg = (uint32_t)((q->cap1 + BQ_L2_TILE - 1) / BQ_L2_TILE);
    slots1 = (size_t)g.n_bins * q->cap1;
    slots2 = (size_t)g.n_regions * q->cap2;
    fills1 = (size_t)g.n_bins * BB_CURSOR_STRIDE;
    fills2 = (size_t)g.n_regions * BB_CURSOR_STRIDE;
    rows2 = g.one ? 0 : (uint64_t)g.n_bins * q->tiles_per_seg;
  }
  q->ovf_cap = n / 64 < 65536 ? 65536 : n / 64;
  if (c->tune.bloom_slot_tight == 2) q->ovf_cap = 64;
  const uint64_t n_tiles1 = (nr + 1023) / 1024;
  const uint64_t rows1 = n_tiles1 * steps * m;
  const uint32_t buckets1 = g.one ? g.n_regions : g.n_bins;
  const size_t head = 256 + (fills1 + fills2) * sizeof(uint32_t);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t at = off;
    off += al256(bytes);
    return at;
  };
  const size_t o_head = take(head), o_l1 = take(slots1 * 4), o_l2 = take(slots2 * 4), o_ovf = take((size_t)q->ovf_cap * 8);
  const size_t o_w1 = take((size_t)rows1 * 16 * 1024 * 2), o_w2 = take(slots1 * 2);
  const size_t o_t1 = take((size_t)rows1 * buckets1 * 8), o_v1 = take((size_t)rows1 * buckets1 * 4);
  const size_t o_t2 = take((size_t)rows2 * BB_REGIONS_PER_BIN * 8), o_v2 = take((size_t)rows2 * BB_REGIONS_PER_BIN * 4);
  const size_t o_p1 = take(slots1), o_p2 = take(slots2), o_po = take((size_t)q->ovf_cap);
  const size_t o_sv = take((size_t)n_tiles1 * steps * 1024 * 2);
  *need = off;
  if (c->bloom_tmp_bytes < off) return false;
  uint8_t* const b = c->bloom_tmp;
  q->status = (BloomStatus*)(b + o_head);
  q->total_hits = (unsigned long long*)(b + o_head + 128);
  q->cur1 = (uint32_t*)(b + o_head + 256);
  q->cur2 = q->cur1 + fills1;
  q->head_bytes = pieces ? 256 : head; // (the pieces' fill arrays are written whole by the kernels)
  q->list1 = (uint32_t*)(b + o_l1);
  q->list2 = (uint32_t*)(b + o_l2);
  q->ovf = (uint64_t*)(b + o_ovf);
  q->where1 = (uint16_t*)(b + o_w1);
  q->where2 = (uint16_t*)(b + o_w2);
  q->tab1 = (uint2*)(b + o_t1);
  q->tovf1 = (uint32_t*)(b + o_v1);
  q->tab2 = (uint2*)(b + o_t2);
  q->tovf2 = (uint32_t*)(b + o_v2);
  q->pay1 = b + o_p1;
  q->pay2 = b + o_p2;
  q->ovf_pay = b + o_po;
  q->surv = (uint16_t*)(b + o_sv);
  return true;
}

// ---- what the rounds of both entry points (reads: query_round; hash streams: stream_query_binned) share: level 2 forward, the
// lookup region by region, level 2 back -- and *b ready for the caller's level-1 back kernel (where1 / tab1 / tovf1, the answers
// next to the list level 1 wrote, g1).  q.pieces: block-private pieces at both levels (q.pg.g1 blocks at level 1, q.pg.gx per bin
// at level 2), else shared cursors.  Launches only.
template <int KIND>
int query_middle(nthip_ctx* c, const QueryGeo& g, const QueryScratch& q, const uint32_t* d_table, uint64_t n_slots, BloomBackArgs* bp)
{
  const uint64_t magic = bloom_magic_of(n_slots);
  const uint64_t table_dwords = KIND == BQ_BLOOM ? (n_slots + 31) / 32 : (n_slots + 3) / 4;
  const uint32_t gx = q.pg.gx;
  // ---- forward, level 2: every bin to its regions ----
  if (q.pieces) {
    BloomPartPiecesArgs a;
    memset(&a, 0, sizeof a);
    a.in = q.list1;
    a.out = q.list2;
    a.fill_in = q.cur1;
    a.fill_out = q.cur2;
    a.cap_in = q.cap1;
    a.n_pieces_in = q.pg.g1;
    a.in_buckets = g.n_bins;
    a.n_regions = g.n_regions;
    a.shift = g.region_shift;
    a.mask = (1u << g.region_shift) - 1u;
    a.buckets_per_seg = BB_REGIONS_PER_BIN;
    a.sl = {q.cap2, q.ovf, q.status, q.ovf_cap};
    a.q_where = q.where2;
    a.q_tab = q.tab2;
    a.q_tovf = q.tovf2;
    a.q_tiles_per_piece = q.tiles_per_seg;
    const size_t lds = ((size_t)BQ_L2_TILE + (size_t)BB_REGIONS_PER_BIN * 32u) * sizeof(uint32_t);
    NTCHK(set_max_lds(c, bloom_part_pieces_kernel<BQ_L2_THREADS, true>, lds));
    hipLaunchKernelGGL((bloom_part_pieces_kernel<BQ_L2_THREADS, true>), dim3(gx, g.n_bins), dim3(BQ_L2_THREADS), lds, c->stream, a);
  } else if (!g.one) {
    BloomPartQueryArgs a;
    memset((void*)&a, 0, sizeof a);
    a.n_bits = n_slots;
    a.magic = magic;
    a.n_regions = g.n_regions;
    a.in = q.list1;
    a.out = q.list2;
    a.cursor = q.cur2;
    a.shift = g.region_shift;
    a.mask = (1u << g.region_shift) - 1u;
    a.buckets_per_seg = BB_REGIONS_PER_BIN;
    a.sl = {q.cap2, q.ovf, q.status, q.ovf_cap};
    a.cap_in = q.cap1;
    a.seg_fill = q.cur1;
    a.q_where = q.where2;
    a.q_tab = q.tab2;
    a.q_tovf = q.tovf2;
    a.q_tiles_per_seg = q.tiles_per_seg;
    const size_t lds = (size_t)BQ_L2_TILE * sizeof(uint32_t);
    int l2_per_cu = 1;
    NTCHK(blocks_per_cu(c, bloom_part_kernel<false, BQ_L2_THREADS, true>, (int)BQ_L2_THREADS, lds, &l2_per_cu));
    const uint32_t l2_grid = 2u * (uint32_t)c->n_cu * (uint32_t)l2_per_cu;
    const uint32_t per_bin = l2_grid / g.n_bins ? l2_grid / g.n_bins : 1u;
    hipLaunchKernelGGL((bloom_part_kernel<false, BQ_L2_THREADS, true>), dim3(per_bin, g.n_bins), dim3(BQ_L2_THREADS), lds, c->stream, a);
  }
  // ---- lookup: region by region, and the overflow list ----
  {
    const size_t lds = (size_t)BB_REGION_DWORDS * sizeof(uint32_t);
    NTCHK(set_max_lds(c, bloom_lookup_kernel<KIND>, lds));
    const uint32_t grid = g.n_regions < (uint32_t)c->n_cu ? g.n_regions : (uint32_t)c->n_cu;
    hipLaunchKernelGGL(bloom_lookup_kernel<KIND>, dim3(grid), dim3(BQ_LOOKUP_THREADS), lds, c->stream, (const uint32_t*)q.list2,
                       (const uint32_t*)q.cur2, q.cap2, g.n_regions, d_table, table_dwords, q.pay2, q.pieces ? gx : 0u,
                       (uint32_t)BB_REGIONS_PER_BIN);
    hipLaunchKernelGGL(bloom_ovf_lookup_kernel<KIND>, dim3(c->n_cu), dim3(256), 0, c->stream, (const uint64_t*)q.ovf, (const BloomStatus*)q.status,
                       q.ovf_cap, d_table, q.ovf_pay);
  }
  // ---- back, level 2 ----
  BloomBackArgs& b = *bp;
  memset(&b, 0, sizeof b);
  b.ovf_pay = q.ovf_pay;
  b.status = q.status;
  b.ovf_cap = q.ovf_cap;
  if (!g.one) {
    b.where = q.where2;
    b.tab = q.tab2;
    b.tovf = q.tovf2;
    b.pay_in = q.pay2;
    b.cap = q.cap2;
    b.pay_out = q.pay1;
    b.seg_fill = q.cur1;
    b.cap_in = q.cap1;
    b.n_regions = g.n_regions;
    b.buckets_per_seg = BB_REGIONS_PER_BIN;
    b.tiles_per_seg = q.tiles_per_seg;
    int per_cu = 1;
    if (q.pieces) {
      b.fill_in = q.cur1;
      b.n_pieces_in = q.pg.g1;
      b.in_buckets = g.n_bins;
      b.gx = gx;
      NTCHK(blocks_per_cu(c, bloom_back2_pieces_kernel<BQ_L2_THREADS>, (int)BQ_L2_THREADS, 0, &per_cu));
    } else {
      NTCHK(blocks_per_cu(c, bloom_back2_kernel<BQ_L2_THREADS>, (int)BQ_L2_THREADS, 0, &per_cu));
    }
    const uint32_t grid2 = 2u * (uint32_t)c->n_cu * (uint32_t)per_cu;
    const uint32_t pb = grid2 / g.n_bins ? grid2 / g.n_bins : 1u;
    if (q.pieces) hipLaunchKernelGGL(bloom_back2_pieces_kernel<BQ_L2_THREADS>, dim3(pb, g.n_bins), dim3(BQ_L2_THREADS), 0, c->stream, b);
    else hipLaunchKernelGGL(bloom_back2_kernel<BQ_L2_THREADS>, dim3(pb, g.n_bins), dim3(BQ_L2_THREADS), 0, c->stream, b);
  }
  // ---- what level 1's way back reads ----
  b.where = q.where1;
  b.tab = q.tab1;
  b.tovf = q.tovf1;
  b.pay_in = g.one ? q.pay2 : q.pay1;
  b.cap = g.one ? q.cap2 : q.cap1;
  b.n_buckets = g.one ? g.n_regions : g.n_bins;
  b.g1 = q.pieces ? q.pg.g1 : 0u;
  return NTHIP_OK;
}

template <int KIND>
int query_round(nthip_ctx* c, const QueryGeo& g, const QueryScratch& q, const BloomFusedSrc& src, const uint32_t* d_table, uint64_t n_slots,
                uint32_t steps, uint64_t* d_hits, uint8_t* d_est, bool* failed, uint64_t* lost, uint64_t* hits, uint32_t jj0 = 0,
                const uint16_t* surv_in = nullptr, uint16_t* surv_out = nullptr)
{
  // (src.m: the hashes of THIS pass, hashes()[jj0 ... jj0 + src.m); surv_in / surv_out: see BloomFusedQueryArgs / BloomBackArgs)
  const uint64_t magic = bloom_magic_of(n_slots);
  const uint32_t shift1 = g.one ? g.region_shift : g.bin_shift, buckets1 = g.one ? g.n_regions : g.n_bins;
  const uint64_t capL1 = g.one ? q.cap2 : q.cap1;
  HIPCHK(hipMemsetAsync(q.status, 0, q.head_bytes, c->stream));
  prof_begin(c, KIND == BQ_BLOOM ? "bloom binned query (part, part, lookup, back, back)" : "count binned query (part, part, lookup, back, back)");
  // ---- forward, level 1: from the reads (bloom_fused_kernels.hpp pass PART, QUERY) ----
  BloomFusedPiecesArgs fa;
  bloom_fused_args(src, 1024u, n_slots, magic, &fa);
  fa.lost = &q.status->lost;
  fa.out = g.one ? q.list2 : q.list1;
  fa.cursor = g.one ? q.cur2 : q.cur1;
  fa.shift = shift1;
  fa.mask = (1u << shift1) - 1u;
  fa.n_buckets = buckets1;
  fa.sl = {capL1, q.ovf, q.status, q.ovf_cap};
  fa.q_where = q.where1;
  fa.q_tab = q.tab1;
  fa.q_tovf = q.tovf1;
  fa.q_steps = steps;
  fa.q_jj0 = jj0;
  fa.q_surv = surv_in;
  fa.p_fill = q.cur1;
  if (q.pieces) {
    const size_t lds = bloom_fused_lds(src, 1024u, 1024u * 16u + BB_PIECES_LDS_DWORDS);
    NTCHK(set_max_lds(c, bloom_fused_kernel<BF_PART, 1024, true, true>, lds));
    hipLaunchKernelGGL((bloom_fused_kernel<BF_PART, 1024, true, true>), dim3(q.pg.g1), dim3(1024), lds, c->stream, fa);
  } else {
    const BloomFusedQueryArgs& fq = fa;
    const size_t lds = bloom_fused_lds(src, 1024u, 1024u * 16u);
    NTCHK(set_max_lds(c, bloom_fused_kernel<BF_PART, 1024, true>, lds));
    hipLaunchKernelGGL((bloom_fused_kernel<BF_PART, 1024, true>), dim3((unsigned)std::min<uint64_t>(fa.n_tiles, (uint64_t)c->n_cu)), dim3(1024), lds,
                       c->stream, fq);
  }
  // ---- level 2 forward, lookup, level 2 back ----
  BloomBackArgs b;
  NTCHK(query_middle<KIND>(c, g, q, d_table, n_slots, &b));
  b.n_reads = src.n_reads;
  b.len = src.len;
  b.k = src.k;
  b.m = src.m;
  b.n_tiles = fa.n_tiles;
  b.steps = steps;
  b.surv_in = surv_in;
  b.surv_out = surv_out;
  b.hits = d_hits;
  b.total_hits = q.total_hits;
  b.estimates = d_est;
  {
    // (the sketch: a tile's estimates collected in LDS when 1024 x windows bytes fit beside the stage)
    size_t lds1 = 0;
    if (KIND == BQ_COUNT && d_est) {
      const size_t want = (((size_t)1024 * (src.len - src.k + 1u)) + 255) & ~(size_t)255;
      if (want + 24 * 1024 <= lds_cap_of(c)) lds1 = want;
    }
    b.est_lds = lds1 ? 1u : 0u;
    int per_cu = 1;
    NTCHK(blocks_per_cu(c, bloom_back1_kernel<KIND, 1024>, 1024, lds1, &per_cu));
    const uint64_t grid = std::min<uint64_t>(fa.n_tiles, (uint64_t)c->n_cu * (uint64_t)per_cu);
    hipLaunchKernelGGL((bloom_back1_kernel<KIND, 1024>), dim3((unsigned)grid), dim3(1024), lds1, c->stream, b);
  }
  prof_end(c);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(c->h_small + 64, q.status, 136, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  BloomStatus st;
  memcpy(&st, c->h_small + 64, sizeof st);
  *failed = st.ovf_n > q.ovf_cap;
  *lost = st.lost;
  memcpy(hits, c->h_small + 64 + 128, 8);
  return NTHIP_OK;
}

// ---- nthip_kmer_count_query on the direct road: the compact stream of a round, the stream query, the estimates to their windows ----
static __global__ __launch_bounds__(256) void window_counts_kernel(const uint64_t* __restrict__ offsets, uint64_t n_reads, uint32_t k,
                                                                   uint64_t* __restrict__ wins)
{
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t len = offsets[r + 1] - offsets[r];
    wins[r] = len >= k ? len - k + 1 : 0;
  }
}
// a wave per read: the estimates of its emitted k-mers (compact, at roff[r]) go to est[slot[r] + pos]
static __global__ __launch_bounds__(256) void place_estimates_kernel(const uint8_t* __restrict__ compact, const uint32_t* __restrict__ pos,
                                                                     const uint64_t* __restrict__ roff, uint64_t n_kmers, const uint64_t* __restrict__ slot,
                                                                     uint64_t fixed_wins, uint64_t n_reads, uint8_t* __restrict__ est)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (uint64_t r = wave; r < n_reads; r += n_waves) {
    const uint64_t i0 = roff[r], i1 = r + 1 < n_reads ? roff[r + 1] : n_kmers;
    const uint64_t s = slot ? slot[r] : r * fixed_wins;
    for (uint64_t i = i0 + lane; i < i1; i += 64u) est[s + pos[i]] = compact[i];
  }
}

} // namespace

int ntamd::host::bloom_query_binned(nthip_ctx* c, const nthip_reads* rd, uint32_t k, uint32_t m, const uint32_t* d_table, uint64_t n_slots, int kind,
                                    uint64_t* d_hits, uint8_t* d_est, uint64_t* first, uint64_t* kmers, uint64_t* hits_sum)
{
  if (c->tune.bloom_query == 2 || rd->offsets || *first >= rd->n_reads) return NTHIP_OK;
  const uint32_t len = rd->fixed_len, stride = rd->stride ? rd->stride : len;
  if (len < k || stride < len || ((uintptr_t)d_table & 15u)) return NTHIP_OK;
  QueryGeo g;
  if (!query_geo(n_slots, kind, &g)) return NTHIP_OK;
  if (c->lds_max < (size_t)BB_REGION_DWORDS * 4 + 1024) return NTHIP_OK;
  const BloomFusedSrc shape = {(const uint8_t*)rd->seqs, rd->n_reads, len, stride, k, m};
  if (!bloom_fused_ok(c, shape, 128u)) return NTHIP_OK;
  const uint32_t nwin = len - k + 1u;
  // m > 1 against a filter: the later hashes are asked only for the k-mers whose earlier ones hit (a k-mer that is not in the
  // filter is out after one lookup, not m) -- pass 0 takes hashes()[0] of every k-mer and leaves the survivors' windows,
  // the second pass the other m - 1 hashes of the survivors.  m_s: the most hashes a pass takes.
  const bool passes = kind == BQ_BLOOM && m > 1 && c->tune.bloom_query_passes != 2;
  const uint32_t m_s = passes ? (m - 1u > 1u ? m - 1u : 1u) : m;
  const uint64_t per_read = (uint64_t)nwin * m_s;
  const uint64_t left_values = (rd->n_reads - *first) * (uint64_t)nwin * m;
  const uint64_t table_bytes = kind == BQ_BLOOM ? (n_slots + 7) / 8 : n_slots;
  // worth it?  The direct kernels pay a 128-byte line per value (~20 ps) unless the table sits in the L2s (~8 ps); the lists
  // ~11 ps per value, and a pass over the table (0.2 ps per byte)
  if (c->tune.bloom_query != 1 && (left_values < (1ull << 24) || table_bytes < (32ull << 20) || left_values < table_bytes / 32)) return NTHIP_OK;
  const uint32_t steps = ((len + 15u) >> 4) - ((k - 1u) >> 4);
  // reads per round: what the free memory allows (~21 B per value + the tiles' tables)
  const size_t free_b = round_memory(c, c->bloom_tmp_bytes, (size_t)8 << 30);
  QueryScratch q;
  uint32_t gx = 1;
  const bool pieces = query_pieces_ok(c, g, shape, &gx);
  uint64_t round = (uint64_t)(free_b / 10 * 8) / 26;
  const uint64_t round_max = pieces ? BQ_PIECES_ROUND_MAX : BQ_ROUND_MAX;
  if (round > round_max) round = round_max;
  if (c->tune.bloom_round) round = c->tune.bloom_round;
  uint64_t reads_per_round = round / per_read;
  if (reads_per_round == 0) return NTHIP_OK;
  if (reads_per_round > rd->n_reads - *first) reads_per_round = rd->n_reads - *first;
  for (;;) { // the scratch of the largest round
    size_t need = 0;
    if (query_scratch(c, g, reads_per_round, reads_per_round * per_read, n_slots, steps, m_s, &q, &need, pieces, gx, per_read)) break;
    if (c->bloom_tmp) HIPCHK(hipFree(c->bloom_tmp));
    c->bloom_tmp = nullptr;
    c->bloom_tmp_bytes = 0;
    if (hipMalloc((void**)&c->bloom_tmp, need) == hipSuccess) {
      c->bloom_tmp_bytes = need;
      continue;
    }
    (void)hipGetLastError();
    c->bloom_tmp = nullptr;
    if (reads_per_round * per_read <= (1u << 22)) return NTHIP_OK; // (no memory for the lists: the direct kernels)
    reads_per_round /= 2;
  }
  while (*first < rd->n_reads) {
    const uint64_t nr = std::min<uint64_t>(rd->n_reads - *first, reads_per_round);
    size_t need = 0;
    if (!query_scratch(c, g, nr, nr * per_read, n_slots, steps, m_s, &q, &need, pieces, gx, per_read)) return NTHIP_OK; // (cannot happen: a smaller round needs less)
    BloomFusedSrc src = {(const uint8_t*)rd->seqs + *first * stride, nr, len, stride, k, m};
    bool failed = false;
    uint64_t lost = 0, hits = 0;
    uint64_t* const dh = d_hits ? d_hits + *first : nullptr;
    uint8_t* const de = d_est ? d_est + *first * nwin : nullptr;
    if (kind != BQ_BLOOM) {
      NTCHK((query_round<BQ_COUNT>(c, g, q, src, d_table, n_slots, steps, dh, de, &failed, &lost, &hits)));
    } else if (!passes) {
      NTCHK((query_round<BQ_BLOOM>(c, g, q, src, d_table, n_slots, steps, dh, de, &failed, &lost, &hits)));
    } else {
      uint64_t lost_p = 0, alive = 0;
      src.m = 1;
      NTCHK((query_round<BQ_BLOOM>(c, g, q, src, d_table, n_slots, steps, nullptr, nullptr, &failed, &lost, &alive, 0u, nullptr, q.surv)));
      const uint64_t emitted = nr * (uint64_t)nwin - lost;
      // the rest in ONE more pass (a pass has fixed costs -- the reads are hashed, every window's place is written and read
      // back -- that a pass per hash pays m - 1 times: 4 GiB filter, m = 3, half the k-mers in it: 71 ms one by one, ~55 ms
      // so, 74 ms with all three in one pass)
      (void)emitted;
      const bool at_once = c->tune.bloom_query_passes != 1;
      for (uint32_t jj0 = 1; jj0 < m && !failed;) {
        const uint32_t mp = at_once ? m - jj0 : 1u;
        const bool last = jj0 + mp == m;
        src.m = mp;
        NTCHK((query_round<BQ_BLOOM>(c, g, q, src, d_table, n_slots, steps, last ? dh : nullptr, nullptr, &failed, &lost_p, last ? &hits : &alive, jj0,
                                     q.surv, last ? nullptr : q.surv)));
        jj0 += mp;
      }
    }
    if (failed) return NTHIP_OK; // (skewed values: the caller's direct kernels take it from here)
    *first += nr;
    *kmers += nr * (uint64_t)nwin - lost;
    *hits_sum += hits;
  }
  return NTHIP_OK;
}

// ---- the binned query of a hash stream ---------------------------------------------------------------------------------------------
int ntamd::host::answers_per_kmer(nthip_ctx* c, const uint8_t* d_ans, uint64_t n_kmers, uint32_t m, int kind, uint8_t* d_out,
                                  unsigned long long* d_found)
{
  if (n_kmers == 0) return NTHIP_OK;
  const unsigned grid = (unsigned)std::min<uint64_t>((n_kmers + 255) / 256, (uint64_t)c->n_cu * 16);
  if (kind == BQ_BLOOM) hipLaunchKernelGGL(answers_per_kmer_kernel<BQ_BLOOM>, dim3(grid), dim3(256), 0, c->stream, d_ans, n_kmers, m, d_out, d_found);
  else hipLaunchKernelGGL(answers_per_kmer_kernel<BQ_COUNT>, dim3(grid), dim3(256), 0, c->stream, d_ans, n_kmers, m, d_out, d_found);
  HIPCHK(hipGetLastError());
  return NTHIP_OK;
}
int ntamd::host::answers_hits_per_read(nthip_ctx* c, const uint8_t* d_ans, const uint64_t* d_roff, uint64_t n_reads, uint64_t n_kmers, uint32_t m,
                                       uint64_t* d_hits, unsigned long long* d_total_hits)
{
  if (n_reads == 0) return NTHIP_OK;
  hipLaunchKernelGGL(answers_per_read_kernel, dim3((unsigned)(c->n_cu * 8)), dim3(256), 0, c->stream, d_ans, d_roff, n_reads, n_kmers, m, d_hits,
                     d_total_hits);
  HIPCHK(hipGetLastError());
  return NTHIP_OK;
}

#ifndef SQ_L1_THREADS
#define SQ_L1_THREADS 512 // slots mode (1024 threads with the query's records: 28 B of scratch; 512 held to 128 VGPRs: 20 B)
#endif
#ifndef SQ_PIECES_THREADS
#define SQ_PIECES_THREADS 1024 // pieces mode
#endif
// the tests of stream_query_binned that need nothing but the shapes: a caller asks BEFORE it takes the kept answers buffer (a
// grow-only buffer of a byte per value has no business in a context whose batches the binned road then declines)
bool ntamd::host::stream_query_applies(const nthip_ctx* c, const void* d_table, uint64_t n_slots, int kind, uint64_t n_values)
{
  if (c->tune.bloom_query == 2 || n_values == 0 || ((uintptr_t)d_table & 15u)) return false;
  QueryGeo g;
  if (!query_geo(n_slots, kind, &g)) return false;
  if (c->lds_max < (size_t)BB_REGION_DWORDS * 4 + 1024) return false;
  const uint64_t table_bytes = kind == BQ_BLOOM ? (n_slots + 7) / 8 : n_slots;
  return c->tune.bloom_query == 1 || !(n_values < (1ull << 24) || table_bytes < (32ull << 20) || n_values < table_bytes / 32);
}

int ntamd::host::stream_query_binned(nthip_ctx* c, const uint64_t* d_hashes, uint64_t n_in, const uint32_t* d_table, uint64_t n_slots, int kind,
                                     uint8_t* d_ans, bool* done, uint32_t M, uint64_t kmul)
{
  // (M = 2 ... 4: the stream holds hashes()[0] of n_in inputs, level 1 makes the other M - 1 values of each -- pieces mode only;
  //  d_ans[input * M + j]; n_values below: what is looked up)
  *done = false;
  if (M < 1 || M > 4) return NTHIP_OK;
  const uint64_t n_values = n_in * M;
  if (!stream_query_applies(c, d_table, n_slots, kind, n_values)) return NTHIP_OK;
  QueryGeo g;
  if (!query_geo(n_slots, kind, &g)) return NTHIP_OK;
  {
    const uint8_t* const h0 = (const uint8_t*)d_hashes; // (the stream must not live in the buffer the lists are carved from)
    if (c->bloom_tmp && h0 < c->bloom_tmp + c->bloom_tmp_bytes && h0 + n_in * 8 > c->bloom_tmp) return NTHIP_OK;
  }
  constexpr uint32_t L1_THREADS = SQ_L1_THREADS, L1_TILE = L1_THREADS * BB_PART_ITEMS;
  constexpr uint32_t P1_THREADS = SQ_PIECES_THREADS, P1_SORTED = P1_THREADS * BB_PART_ITEMS;
  const uint64_t P1_TILE = (uint64_t)P1_THREADS * (BB_PART_ITEMS / M); // inputs per tile of the pieces level
  const uint64_t magic = bloom_magic_of(n_slots);
  const uint32_t shift1 = g.one ? g.region_shift : g.bin_shift, buckets1 = g.one ? g.n_regions : g.n_bins;
  // pieces mode (a two-level table): block-private pieces at both levels, whole lines only (bloom_binned_kernels.hpp)
  bool pieces = !g.one && c->tune.bloom_pieces != 2;
  const size_t p1_lds = ((size_t)P1_SORTED + (size_t)BB_MAX_BINS * 32u) * sizeof(uint32_t);
  const size_t p2_lds = ((size_t)BQ_L2_TILE + (size_t)BB_REGIONS_PER_BIN * 32u) * sizeof(uint32_t);
  uint32_t g1_max = 1, gx = 1;
  if (pieces) {
    int per1 = 1, per2 = 1;
    if (p1_lds + 4096 > lds_cap_of(c) || blocks_per_cu(c, bloom_part_stream_pieces_kernel<P1_THREADS, true>, (int)P1_THREADS, p1_lds, &per1) != NTHIP_OK ||
        blocks_per_cu(c, bloom_part_pieces_kernel<BQ_L2_THREADS, true>, (int)BQ_L2_THREADS, p2_lds, &per2) != NTHIP_OK) {
      (void)hipGetLastError();
      pieces = false;
    } else {
      g1_max = (uint32_t)c->n_cu * (uint32_t)per1;
      const uint32_t grid2 = 2u * (uint32_t)c->n_cu * (uint32_t)per2;
      gx = grid2 / g.n_bins ? grid2 / g.n_bins : 1u;
    }
  }
  if (M > 1 && !pieces) return NTHIP_OK;
  // values per round: ~22 B of scratch per value
  const size_t free_b = round_memory(c, c->bloom_tmp_bytes, (size_t)8 << 30);
  uint64_t round = (uint64_t)(free_b / 10 * 8) / 24;
  if (round > (pieces ? BQ_PIECES_ROUND_MAX : BQ_ROUND_MAX)) round = pieces ? BQ_PIECES_ROUND_MAX : BQ_ROUND_MAX;
  if (c->tune.bloom_round) round = c->tune.bloom_round;
  if (round > n_values) round = n_values;
  if (round < (1u << 16)) round = 1u << 16;
  round = (round + M - 1) / M; // inputs per round from here on
  QueryScratch q;
  auto carve = [&](uint64_t n, size_t* need) -> bool {
    size_t slots1, slots2, fills1, fills2;
    uint64_t tiles1, rows2;
    if (pieces) {
      tiles1 = (n + P1_TILE - 1) / P1_TILE;
      q.pg.g1 = (uint32_t)std::min<uint64_t>(tiles1, g1_max);
      q.pg.gx = gx;
      const double per_block = (double)((tiles1 + q.pg.g1 - 1) / q.pg.g1) * (double)(P1_TILE * M); // the values a level-1 block may see
      const double bin_slots = (double)(1ull << g.bin_shift), region_slots = (double)(1ull << g.region_shift);
      q.cap1 = piece_cap(c, per_block * (bin_slots < (double)n_slots ? bin_slots / (double)n_slots : 1.0));
      q.cap2 = piece_cap(c, (double)((q.pg.g1 + gx - 1) / gx) * per_block * (region_slots < (double)n_slots ? region_slots / (double)n_slots : 1.0));
      q.tiles_per_seg = (uint32_t)((q.cap1 + BQ_L2_TILE - 1) / BQ_L2_TILE); // (tile rows per PIECE)
      slots1 = (size_t)g.n_bins * q.pg.g1 * q.cap1;
      slots2 = (size_t)g.n_regions * gx * q.cap2;
      fills1 = (size_t)q.pg.g1 * g.n_bins;
      fills2 = (size_t)g.n_bins * gx * BB_REGIONS_PER_BIN;
      rows2 = (uint64_t)g.n_bins * q.pg.g1 * q.tiles_per_seg;
    } else {
      tiles1 = (n + L1_TILE - 1) / L1_TILE;
      q.pg.g1 = 0;
      q.pg.gx = 0;
      q.cap1 = g.one ? 0 : slot_cap(c, n, 1ull << g.bin_shift, n_slots); // (M == 1 here)
      q.cap2 = slot_cap(c, n, 1ull << g.region_shift, n_slots);
      q.tiles_per_seg = (uint32_t)((q.cap1 + BQ_L2_TILE - 1) / BQ_L2_TILE);
      slots1 = (size_t)g.n_bins * q.cap1;
      slots2 = (size_t)g.n_regions * q.cap2;
      fills1 = (size_t)g.n_bins * BB_CURSOR_STRIDE;
      fills2 = (size_t)g.n_regions * BB_CURSOR_STRIDE;
      rows2 = g.one ? 0 : (uint64_t)g.n_bins * q.tiles_per_seg;
    }
    q.ovf_cap = n * M / 64 < 65536 ? 65536 : n * M / 64;
    if (c->tune.bloom_slot_tight == 2) q.ovf_cap = 64;
    q.pieces = pieces;
    const size_t head = 256 + (fills1 + fills2) * sizeof(uint32_t);
    q.head_bytes = pieces ? 256 : head; // (the pieces' fill arrays are written whole by the kernels)
    size_t off = 0;
    auto take = [&](size_t bytes) {
      const size_t at = off;
      off += al256(bytes);
      return at;
    };
    const size_t o_head = take(head), o_l1 = take(slots1 * 4), o_l2 = take(slots2 * 4), o_ovf = take((size_t)q.ovf_cap * 8);
    const size_t o_w1 = take((size_t)n * M * 2), o_w2 = take(slots1 * 2);
    const size_t o_t1 = take((size_t)tiles1 * buckets1 * 8), o_v1 = take((size_t)tiles1 * buckets1 * 4);
    const size_t o_t2 = take((size_t)rows2 * BB_REGIONS_PER_BIN * 8), o_v2 = take((size_t)rows2 * BB_REGIONS_PER_BIN * 4);
    const size_t o_p1 = take(slots1), o_p2 = take(slots2), o_po = take((size_t)q.ovf_cap);
    *need = off;
    if (c->bloom_tmp_bytes < off) return false;
    uint8_t* const b = c->bloom_tmp;
    q.status = (BloomStatus*)(b + o_head);
    q.cur1 = (uint32_t*)(b + o_head + 256);
    q.cur2 = q.cur1 + fills1;
    q.list1 = (uint32_t*)(b + o_l1);
    q.list2 = (uint32_t*)(b + o_l2);
    q.ovf = (uint64_t*)(b + o_ovf);
    q.where1 = (uint16_t*)(b + o_w1);
    q.where2 = (uint16_t*)(b + o_w2);
    q.tab1 = (uint2*)(b + o_t1);
    q.tovf1 = (uint32_t*)(b + o_v1);
    q.tab2 = (uint2*)(b + o_t2);
    q.tovf2 = (uint32_t*)(b + o_v2);
    q.pay1 = b + o_p1;
    q.pay2 = b + o_p2;
    q.ovf_pay = b + o_po;
    return true;
  };
  for (;;) {
    size_t need = 0;
    if (carve(round, &need)) break;
    if (c->bloom_tmp) HIPCHK(hipFree(c->bloom_tmp));
    c->bloom_tmp = nullptr;
    c->bloom_tmp_bytes = 0;
    if (hipMalloc((void**)&c->bloom_tmp, need) == hipSuccess) {
      c->bloom_tmp_bytes = need;
      continue;
    }
    (void)hipGetLastError();
    c->bloom_tmp = nullptr;
    if (round <= (1u << 22)) return NTHIP_OK;
    round /= 2;
  }
  for (uint64_t v0 = 0; v0 < n_in; v0 += round) {
    const uint64_t n = std::min<uint64_t>(round, n_in - v0);
    size_t need = 0;
    if (!carve(n, &need)) return NTHIP_OK; // (a sizing slip: *done stays false, the caller's direct kernel answers everything)
    const uint64_t capL1 = g.one ? q.cap2 : q.cap1;
    HIPCHK(hipMemsetAsync(q.status, 0, q.head_bytes, c->stream));
    prof_begin(c, kind == BQ_BLOOM ? "bloom binned stream query (part, part, lookup, back, back)" : "count binned stream query (part, part, lookup, back, back)");
    if (pieces) { // forward, level 1: the stream's values into the blocks' pieces of the bins
      BloomPartStreamPiecesArgs a;
      memset((void*)&a, 0, sizeof a);
      a.in = d_hashes + v0;
      a.n = n;
      a.kmul = kmul;
      a.n_bits = n_slots;
      a.magic = magic;
      a.out = q.list1;
      a.fill_out = q.cur1;
      a.shift = g.bin_shift;
      a.mask = (1u << g.bin_shift) - 1u;
      a.n_buckets = g.n_bins;
      a.sl = {q.cap1, q.ovf, q.status, q.ovf_cap};
      a.q_where = q.where1;
      a.q_tab = q.tab1;
      a.q_tovf = q.tovf1;
      switch (M) {
        case 2:
          NTCHK(set_max_lds(c, bloom_part_stream_pieces_kernel<P1_THREADS, true, 2>, p1_lds));
          hipLaunchKernelGGL((bloom_part_stream_pieces_kernel<P1_THREADS, true, 2>), dim3(q.pg.g1), dim3(P1_THREADS), p1_lds, c->stream, a);
          break;
        case 3:
          NTCHK(set_max_lds(c, bloom_part_stream_pieces_kernel<P1_THREADS, true, 3>, p1_lds));
          hipLaunchKernelGGL((bloom_part_stream_pieces_kernel<P1_THREADS, true, 3>), dim3(q.pg.g1), dim3(P1_THREADS), p1_lds, c->stream, a);
          break;
        case 4:
          NTCHK(set_max_lds(c, bloom_part_stream_pieces_kernel<P1_THREADS, true, 4>, p1_lds));
          hipLaunchKernelGGL((bloom_part_stream_pieces_kernel<P1_THREADS, true, 4>), dim3(q.pg.g1), dim3(P1_THREADS), p1_lds, c->stream, a);
          break;
        default:
          NTCHK(set_max_lds(c, bloom_part_stream_pieces_kernel<P1_THREADS, true>, p1_lds));
          hipLaunchKernelGGL((bloom_part_stream_pieces_kernel<P1_THREADS, true>), dim3(q.pg.g1), dim3(P1_THREADS), p1_lds, c->stream, a);
      }
    } else {
      { // forward, level 1: the stream's values behind shared cursors
        BloomPartQueryArgs a;
        memset((void*)&a, 0, sizeof a);
        a.in = d_hashes + v0;
        a.n = n;
        a.n_bits = n_slots;
        a.magic = magic;
        a.n_regions = g.n_regions;
        a.out = g.one ? q.list2 : q.list1;
        a.cursor = g.one ? q.cur2 : q.cur1;
        a.shift = shift1;
        a.mask = (1u << shift1) - 1u;
        a.buckets_per_seg = buckets1;
        a.sl = {capL1, q.ovf, q.status, q.ovf_cap};
        a.q_where = q.where1;
        a.q_tab = q.tab1;
        a.q_tovf = q.tovf1;
        a.q_tiles_per_seg = (uint32_t)((n + L1_TILE - 1) / L1_TILE);
        const size_t lds = (size_t)L1_TILE * sizeof(uint32_t);
        NTCHK(set_max_lds(c, bloom_part_kernel<true, L1_THREADS, true>, lds));
        hipLaunchKernelGGL((bloom_part_kernel<true, L1_THREADS, true>), dim3((unsigned)c->n_cu * (2048u / L1_THREADS)), dim3(L1_THREADS), lds, c->stream, a);
      }
    }
    // level 2 forward, lookup, level 2 back (what the reads' rounds do); then every value's answer
    BloomBackArgs b;
    if (kind == BQ_BLOOM) NTCHK(query_middle<BQ_BLOOM>(c, g, q, d_table, n_slots, &b));
    else NTCHK(query_middle<BQ_COUNT>(c, g, q, d_table, n_slots, &b));
    {
      int per_cu = 1;
      if (pieces) {
        NTCHK(blocks_per_cu(c, bloom_back1_stream_kernel<P1_THREADS>, (int)P1_THREADS, 0, &per_cu));
        const uint64_t tiles = (n + P1_TILE - 1) / P1_TILE;
        const unsigned grid = (unsigned)std::min<uint64_t>(tiles, (uint64_t)c->n_cu * (uint64_t)per_cu);
        uint8_t* const ans = d_ans + v0 * M;
        switch (M) {
          case 2: hipLaunchKernelGGL((bloom_back1_stream_kernel<P1_THREADS, 2>), dim3(grid), dim3(P1_THREADS), 0, c->stream, b, n, ans); break;
          case 3: hipLaunchKernelGGL((bloom_back1_stream_kernel<P1_THREADS, 3>), dim3(grid), dim3(P1_THREADS), 0, c->stream, b, n, ans); break;
          case 4: hipLaunchKernelGGL((bloom_back1_stream_kernel<P1_THREADS, 4>), dim3(grid), dim3(P1_THREADS), 0, c->stream, b, n, ans); break;
          default: hipLaunchKernelGGL(bloom_back1_stream_kernel<P1_THREADS>, dim3(grid), dim3(P1_THREADS), 0, c->stream, b, n, ans);
        }
      } else {
        NTCHK(blocks_per_cu(c, bloom_back1_stream_kernel<L1_THREADS>, (int)L1_THREADS, 0, &per_cu));
        const uint64_t tiles = (n + L1_TILE - 1) / L1_TILE;
        const uint64_t grid = std::min<uint64_t>(tiles, (uint64_t)c->n_cu * (uint64_t)per_cu);
        hipLaunchKernelGGL(bloom_back1_stream_kernel<L1_THREADS>, dim3((unsigned)grid), dim3(L1_THREADS), 0, c->stream, b, n, d_ans + v0);
      }
    }
    prof_end(c);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->h_small + 64, q.status, sizeof(BloomStatus), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    BloomStatus st;
    memcpy(&st, c->h_small + 64, sizeof st);
    if (st.ovf_n > q.ovf_cap) return NTHIP_OK; // (skewed values: *done stays false, the caller's direct kernel answers everything)
  }
  *done = true;
  return NTHIP_OK;
}

int ntamd::host::stream_hits_per_read(nthip_ctx* c, const uint64_t* d_h, const uint64_t* d_roff, uint64_t n_reads, uint64_t n_kmers, uint32_t m,
                                      const uint32_t* d_filter, uint64_t n_bits, uint64_t* d_hits, unsigned long long* d_total,
                                      const char* direct_label, uint32_t expand_m, uint64_t kmul, bool* expanded)
{
  // (expand_m > 1: d_h holds hashes()[0] only -- m / expand_m values per k-mer -- and the binned road makes the others; *expanded
  //  = false, nothing launched, when that road is not taken: the caller hashes the full stream and calls again)
  if (expanded) *expanded = false;
  if (n_reads == 0) {
    if (expanded) *expanded = true;
    return NTHIP_OK;
  }
  const uint64_t n_values = n_kmers * m;
  bool done = false;
  uint8_t* d_ans = nullptr;
  if (stream_query_applies(c, d_filter, n_bits, BQ_BLOOM, n_values) &&
      kept_alloc(c, KEPT_ANSWERS, n_values + 8, (void**)&d_ans) != NTHIP_OK) // (+ 8: answers_per_read_kernel loads 8 bytes at a k-mer's first)
    d_ans = nullptr;
  if (d_ans) {
    NTCHK(stream_query_binned(c, d_h, n_values / expand_m, d_filter, n_bits, BQ_BLOOM, d_ans, &done, expand_m, kmul));
    if (done) {
      prof_begin(c, "answers_per_read_kernel");
      const int rc2 = answers_hits_per_read(c, d_ans, d_roff, n_reads, n_kmers, m, d_hits, d_total);
      prof_end(c);
      if (expanded) *expanded = true;
      return rc2;
    }
  }
  if (expand_m > 1) return NTHIP_OK;
  prof_begin(c, direct_label);
  hipLaunchKernelGGL(stream_bloom_query_kernel, dim3((unsigned)(c->n_cu * 8)), dim3(256), 0, c->stream, d_h, d_roff, n_reads, n_kmers, m, d_filter,
                     n_bits, bloom_magic_of(n_bits), d_hits, d_total);
  prof_end(c);
  HIPCHK(hipGetLastError());
  return NTHIP_OK;
}

// ---- nthip_kmer_count_query ---------------------------------------------------------------------------------------------------
extern "C" int nthip_kmer_count_query(nthip_ctx* c, const nthip_reads* rd, uint16_t k16, uint8_t m8, const uint8_t* d_counters, uint64_t n_counters,
                                      uint8_t* estimates, uint64_t* total, uint32_t flags)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  NTCHK(check_reads(rd));
  const uint32_t k = k16, m = m8;
  if (!d_counters || n_counters == 0) return fail(NTHIP_ERR_ARG, "sketch is NULL / n_counters is 0");
  if ((uintptr_t)d_counters & 3u) return fail(NTHIP_ERR_ARG, "sketch must be 4-byte aligned");
  if (n_counters & 3u) return fail(NTHIP_ERR_ARG, "n_counters must be a multiple of 4");
  if (k == 0) return fail(NTHIP_ERR_ARG, "k must be greater than 0");
  if (k < 3) return fail(NTHIP_ERR_UNSUPPORTED, "k < 3 is undefined in the reference (src/kmer.cpp:47)");
  if (m == 0) return fail(NTHIP_ERR_UNSUPPORTED, "num_hashes must be >= 1");
  if (rd->n_reads && !estimates) return fail(NTHIP_ERR_ARG, "estimates is NULL");
  HIPCHK(hipSetDevice(c->device));
  if (total) *total = 0;
  if (rd->n_reads == 0) return NTHIP_OK;
  const bool host_in = (flags & NTHIP_HOST_INPUT) != 0, host_out = (flags & NTHIP_HOST_OUTPUT) != 0;
  Staged keep;
  uint64_t sum_kmers = 0;
  // the slot of read r: the windows of the reads before it
  uint64_t n_slots_out = 0;
  uint64_t* d_slot = nullptr; // (offsets only)
  const uint32_t len = rd->fixed_len, stride = rd->stride ? rd->stride : len;
  const uint64_t fixed_wins = (!rd->offsets && len >= k) ? len - k + 1 : 0;
  const uint64_t* d_offsets = rd->offsets;
  if (rd->offsets) {
    if (host_in) {
      uint64_t* p = nullptr;
      NTCHK(own_alloc(keep, (size_t)(rd->n_reads + 1) * 8, (void**)&p));
      HIPCHK(hipMemcpyAsync(p, rd->offsets, (size_t)(rd->n_reads + 1) * 8, hipMemcpyHostToDevice, c->stream));
      d_offsets = p;
    }
    uint64_t* d_sums = nullptr;
    NTCHK(own_alloc(keep, (size_t)(rd->n_reads + 1) * 8, (void**)&d_slot));
    NTCHK(own_alloc(keep, (size_t)(rd->n_reads / SCAN_TILE + 64) * 8, (void**)&d_sums));
    hipLaunchKernelGGL(window_counts_kernel, dim3(c->n_cu * 4), dim3(256), 0, c->stream, d_offsets, rd->n_reads, k, d_slot);
    HIPCHK(hipGetLastError());
    NTCHK(device_exclusive_scan(c, d_slot, d_slot, rd->n_reads, d_sums, (uint64_t*)(c->d_small + 16)));
    HIPCHK(hipMemcpyAsync(c->h_small + 16, c->d_small + 16, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(&n_slots_out, c->h_small + 16, 8);
  } else {
    n_slots_out = rd->n_reads * fixed_wins;
  }
  if (n_slots_out == 0) return NTHIP_OK;
  uint8_t* d_est = estimates;
  if (host_out) NTCHK(own_alloc(keep, (size_t)n_slots_out, (void**)&d_est));
  uint64_t first = 0, hits_dummy = 0;
  if (!rd->offsets && !host_in)
    NTCHK(bloom_query_binned(c, rd, k, m, (const uint32_t*)d_counters, n_counters, BQ_COUNT, nullptr, d_est, &first, &sum_kmers, &hits_dummy));
  if (first < rd->n_reads) {
    // the direct road for what is left: rounds of reads hashed to their compact stream (positions and counts with it), the
    // stream query, every estimate to its window; windows that emit nothing stay 0
    const uint64_t done_slots = rd->offsets ? 0 : first * fixed_wins;
    HIPCHK(hipMemsetAsync(d_est + done_slots, 0, (size_t)(n_slots_out - done_slots), c->stream));
    auto one_round = [&](const nthip_reads* part, uint64_t r0, uint64_t bases) -> int {
      Staged rk;
      const uint64_t cap = bases ? bases : 1;
      uint64_t *d_h = nullptr, *d_counts = nullptr, *d_roff = nullptr, *d_sums = nullptr, n_kmers = 0;
      uint32_t* d_pos = nullptr;
      uint8_t* d_cmp = nullptr;
      NTCHK(own_alloc(rk, (size_t)cap * m * 8, (void**)&d_h));
      NTCHK(own_alloc(rk, (size_t)cap * 4, (void**)&d_pos));
      NTCHK(own_alloc(rk, (size_t)cap, (void**)&d_cmp));
      NTCHK(own_alloc(rk, (size_t)(part->n_reads + 1) * 8, (void**)&d_counts));
      NTCHK(own_alloc(rk, (size_t)(part->n_reads + 1) * 8, (void**)&d_roff));
      NTCHK(own_alloc(rk, (size_t)(part->n_reads / SCAN_TILE + 64) * 8, (void**)&d_sums));
      nthip_out out;
      memset(&out, 0, sizeof out);
      out.hashes = d_h;
      out.capacity = cap;
      out.counts = d_counts;
      out.pos = d_pos;
      NTCHK(nthip_kmer_hash(c, part, k16, m8, &out, &n_kmers, flags & NTHIP_HOST_INPUT));
      sum_kmers += n_kmers;
      if (n_kmers == 0) return NTHIP_OK;
      NTCHK(nthip_stream_count_query(c, d_h, n_kmers, m8, d_counters, n_counters, d_cmp));
      NTCHK(device_exclusive_scan(c, d_counts, d_roff, part->n_reads, d_sums, (uint64_t*)(c->d_small + 16)));
      hipLaunchKernelGGL(place_estimates_kernel, dim3(c->n_cu * 8), dim3(256), 0, c->stream, (const uint8_t*)d_cmp, (const uint32_t*)d_pos,
                         (const uint64_t*)d_roff, n_kmers, d_slot ? (const uint64_t*)(d_slot + r0) : (const uint64_t*)nullptr, fixed_wins,
                         part->n_reads, d_slot ? d_est : d_est + r0 * fixed_wins);
      HIPCHK(hipGetLastError());
      HIPCHK(hipStreamSynchronize(c->stream));
      return NTHIP_OK;
    };
    if (rd->offsets) {
      NTCHK(offsets_in_rounds(c, rd, flags, (size_t)8 * m + 8, one_round));
    } else {
      // fixed-length reads: rounds of reads whose stream fits a fifth of the free memory
      const size_t free_b = round_memory(c, reusable_bytes(c), (size_t)4 << 30);
      uint64_t reads_per_round = std::max<uint64_t>(1, (uint64_t)(free_b / 5) / ((uint64_t)fixed_wins * (8 * m + 5) + 32 + (host_in ? stride : 0)));
      if (c->tune.bloom_round) reads_per_round = std::max<uint64_t>(1, c->tune.bloom_round / (fixed_wins * m));
      for (uint64_t r0 = first; r0 < rd->n_reads; r0 += reads_per_round) {
        nthip_reads part = *rd;
        part.seqs = rd->seqs + r0 * stride;
        part.n_reads = std::min<uint64_t>(reads_per_round, rd->n_reads - r0);
        NTCHK(one_round(&part, r0, part.n_reads * fixed_wins));
      }
    }
  }
  if (host_out) HIPCHK(hipMemcpyAsync(estimates, d_est, (size_t)n_slots_out, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (total) *total = sum_kmers;
  return NTHIP_OK;
}
