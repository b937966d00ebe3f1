// bloom_query_kernels.hpp -- the READ side of the binned Bloom filter / counting sketch (round 5).
//
// What the reference's callers do with hashes() as often as they insert (include/nthash/nthash.hpp:14-17, 56-57 of the
// reference: btllib's contains()): test bit `h mod n_bits` of every value.  One dependent 4-byte load per k-mer from a
// table of gigabytes moves one 128-byte line per k-mer: 49 G k-mers/s at the fabric's line rate, whatever the table is
// cached in (profiles/r04_notes.md 9.2).  Only locality goes under that, so the query takes the insert's road -- the
// values brought to the table region by region (bloom_binned_kernels.hpp, slots mode; bloom_fused_kernels.hpp: the first
// level straight from the hashing registers) -- and then the ANSWERS have to find their way back to the reads:
//
//   forward  level 1 (+ level 2): as the insert, and every tile of a level also leaves
//              where   per value: bucket << 16 | rank inside the tile's bucket      (4 B, written where the value was read)
//              tab     per tile and bucket: {entries, where the run went in the bucket's slots}
//   lookup   a workgroup per region: the region's 128 KiB of the table into LDS, then pay[slot] = the answer (a byte: the
//            bit of a filter, the counter of a sketch) of every entry of the region's list, written next to the list
//   back     level 2, then level 1: a tile loads the runs it once wrote -- now of answers, one byte each -- back into LDS
//            in its sorted order (whole runs, coalesced), every value picks its own with `where`; level 2 writes them next
//            to the bin lists, level 1 -- a thread per read again -- ANDs a k-mer's m answers and sums the read's hits
//            (sketch: the smallest of the m counters, one byte per window)
//
// No atomics on the way back, no 8-byte entries: 20 B of extra traffic per value, all of it in whole runs.  Values that
// did not fit their bucket's slots (a k-mer repeated a million times) went to the overflow list as full positions; a small
// kernel looks those up directly and the way back finds them through tovf (bloom_copy_out).  A round whose overflow list
// overflows is reported to the host, which redoes it with the direct kernel.
#pragma once

#include <hip/hip_runtime.h>

#include "bloom_binned_kernels.hpp"
#include "bloom_fused_kernels.hpp"

namespace ntamd {

enum : int { BQ_BLOOM = 0, BQ_COUNT = 1 };
constexpr uint32_t BQ_NONE = 0xFFFFu;         // `where` of a window that emitted nothing (a place in a tile's sorted order is < 16 Ki)
constexpr uint32_t BQ_COUNT_REGION_SHIFT = 17; // a sketch's region for the query: 2^17 one-byte counters = 128 KiB of LDS
constexpr uint32_t BQ_LOOKUP_THREADS = 1024;
constexpr uint32_t BQ_LOOKUP_BATCH = 4;        // 16-byte loads of entries a thread has in flight

template <int KIND>
__device__ __forceinline__ uint32_t bq_answer_lds(const uint32_t* lds, uint32_t e)
{
  if constexpr (KIND == BQ_BLOOM) return (lds[(e >> 5) & (BB_REGION_DWORDS - 1u)] >> (e & 31u)) & 1u;
  else return ((const uint8_t*)lds)[e & ((1u << BQ_COUNT_REGION_SHIFT) - 1u)];
}

// ---- lookup: a workgroup per region --------------------------------------------------------------------------------------
// dynamic LDS: 128 KiB.  Region r's entries: list[r * cap ... + min(fill[r], cap)), offsets inside the region; pay[slot]
// (one byte per slot of the list, same index) = the answer.  table: 16-byte aligned.
template <int KIND>
static __global__ __launch_bounds__(BQ_LOOKUP_THREADS) void bloom_lookup_kernel(const uint32_t* __restrict__ list, const uint32_t* __restrict__ fill,
                                                                                uint64_t cap, uint32_t n_regions,
                                                                                const uint32_t* __restrict__ table, uint64_t table_dwords,
                                                                                uint8_t* __restrict__ pay, uint32_t n_pieces, uint32_t bps)
{
  // (pieces mode, n_pieces != 0: region r is n_pieces pieces -- piece x at slot (r * n_pieces + x) * cap, its entries
  // fill[((r / bps) * n_pieces + x) * bps + r % bps]; bloom_part_pieces_kernel)
  extern __shared__ __attribute__((aligned(16))) uint32_t bq_lds[];
  uint4* const l4 = (uint4*)bq_lds;
  const uint32_t tid = threadIdx.x;
  const uint32_t n_runs = n_pieces ? n_pieces : 1u;
  for (uint32_t r = blockIdx.x; r < n_regions; r += gridDim.x) {
    bool loaded = false;
    for (uint32_t x = 0; x < n_runs; ++x) {
      const uint64_t f64 = n_pieces ? fill[((size_t)(r / bps) * n_pieces + x) * bps + r % bps] : fill[(size_t)r * BB_CURSOR_STRIDE];
      const uint32_t f = (uint32_t)(f64 < cap ? f64 : cap);
      if (f == 0) continue; // (uniform over the block)
      if (!loaded) {
        __syncthreads(); // the region before is answered
        const uint64_t d0 = (uint64_t)r * BB_REGION_DWORDS;
        const uint64_t left = table_dwords - d0;
        const uint32_t here = left < BB_REGION_DWORDS ? (uint32_t)left : BB_REGION_DWORDS;
        const uint4* const t4 = (const uint4*)(table + d0);
        for (uint32_t i = tid; i < BB_REGION_DWORDS / 4u; i += BQ_LOOKUP_THREADS) {
          uint4 v = make_uint4(0, 0, 0, 0);
          if (i * 4u + 4u <= here) v = t4[i];
          else if (i * 4u < here) { // (a table that does not end on 16 bytes)
            v.x = table[d0 + i * 4u];
            if (i * 4u + 1u < here) v.y = table[d0 + i * 4u + 1u];
            if (i * 4u + 2u < here) v.z = table[d0 + i * 4u + 2u];
          }
          l4[i] = v;
        }
        __syncthreads();
        loaded = true;
      }
      const size_t slot0 = ((size_t)r * n_runs + x) * cap; // (cap is a multiple of 64: 16-byte aligned, whole vectors)
      const bb_v4u* const e4 = (const bb_v4u*)(list + slot0);
      uint32_t* const p4 = (uint32_t*)(pay + slot0);
      const uint32_t nv = (f + 3u) >> 2;
      for (uint32_t i0 = tid; i0 < nv; i0 += BQ_LOOKUP_BATCH * BQ_LOOKUP_THREADS) {
        bb_v4u q[BQ_LOOKUP_BATCH];
#pragma unroll
        for (uint32_t u = 0; u < BQ_LOOKUP_BATCH; ++u) {
          const uint32_t i = i0 + u * BQ_LOOKUP_THREADS;
          q[u] = __builtin_nontemporal_load(e4 + (i < nv ? i : i0));
        }
#pragma unroll
        for (uint32_t u = 0; u < BQ_LOOKUP_BATCH; ++u) {
          const uint32_t i = i0 + u * BQ_LOOKUP_THREADS;
          if (i < nv)
            p4[i] = bq_answer_lds<KIND>(bq_lds, q[u].x) | (bq_answer_lds<KIND>(bq_lds, q[u].y) << 8) |
                    (bq_answer_lds<KIND>(bq_lds, q[u].z) << 16) | (bq_answer_lds<KIND>(bq_lds, q[u].w) << 24);
        }
      }
    }
  }
}

// the values that did not fit their bucket: full positions, answered straight from the table
template <int KIND>
static __global__ __launch_bounds__(256) void bloom_ovf_lookup_kernel(const uint64_t* __restrict__ ovf, const BloomStatus* status, uint64_t ovf_cap,
                                                                      const uint32_t* __restrict__ table, uint8_t* __restrict__ ovf_pay)
{
  uint64_t n = __builtin_nontemporal_load(&status->ovf_n);
  if (n > ovf_cap) n = ovf_cap; // (the round failed: the host redoes it; nothing here is used)
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t p = ovf[i];
    if constexpr (KIND == BQ_BLOOM) ovf_pay[i] = (uint8_t)((table[p >> 5] >> ((uint32_t)p & 31u)) & 1u);
    else ovf_pay[i] = ((const uint8_t*)table)[p];
  }
}

// ---- the way back of one tile ------------------------------------------------------------------------------------------------
// A tile of a partition level wrote, for every bucket b < n_buckets, a run of tab[b].x entries; the first `fit` of them at
// slots (piece0 + b * piece_step) * cap + tab[b].y ... of the level's list (slots mode: piece0 = the first bucket, step 1;
// pieces mode: the writing block's piece of every bucket), the rest to the overflow list.  bq_stage_runs loads the
// answers of those runs (pay: one byte per slot) into `stage` in the tile's sorted order -- stage[off[b] + rank] -- and
// leaves offfit[b] = off[b] | fit[b] << 16.  Whole block; ends with a barrier.
// (offfit has BB_MAX_BINS + 1 words: the last one is set when a run of the tile did not fit its bucket -- bq_pick's slow road)
template <uint32_t NW>
__device__ __forceinline__ void bq_stage_runs(const uint2* __restrict__ tab, uint32_t n_buckets, const uint8_t* __restrict__ pay, uint64_t piece0,
                                              uint64_t piece_step, uint64_t cap, uint8_t* stage, uint32_t* cnt, uint32_t* gat, uint32_t* offfit, uint32_t tid,
                                              uint32_t lane, uint32_t wave_v)
{
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_v); // (uniform, and known to be: scalar address arithmetic)
  if (tid < BB_MAX_BINS) {
    uint2 e = make_uint2(0, 0);
    if (tid < n_buckets) e = tab[tid];
    cnt[tid] = e.x;
    gat[tid] = e.y;
  }
  if (tid == 0) offfit[BB_MAX_BINS] = 0;
  __syncthreads();
  if (wave == 0) { // the tile's exclusive scan, as the forward pass made it: 4 buckets per lane
    uint32_t c[4], s = 0;
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
      c[i] = cnt[lane * 4u + i];
      s += c[i];
    }
    uint32_t incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(incl, d, 64);
      if ((int)lane >= d) incl += o;
    }
    uint32_t run = incl - s;
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
      const uint32_t at = gat[lane * 4u + i];
      const uint32_t fit = at >= cap ? 0u : (cap - at < c[i] ? (uint32_t)(cap - at) : c[i]);
      offfit[lane * 4u + i] = run | (fit << 16);
      if (fit < c[i]) offfit[BB_MAX_BINS] = 1u; // (rare: part of the run is in the overflow list)
      run += c[i];
    }
  }
  __syncthreads();
  // sixteen lanes per bucket, four buckets per instruction (as bloom_copy_out_lines: what a bucket needs is vector arithmetic
  // done once per four); the answers come as aligned dwords -- 64 entries of a bucket per load -- and go to LDS byte by byte
  const uint32_t g = lane >> 4, l = lane & 15u;
  const uint32_t n_mine = n_buckets > wave ? (n_buckets - wave + NW - 1u) / NW : 0u;
  for (uint32_t i0 = 0; i0 < n_mine; i0 += 4u) {
    const bool mine = i0 + g < n_mine;
    const uint32_t b = wave + (mine ? i0 + g : i0) * NW;
    const uint32_t of = offfit[b], at = gat[b];
    const uint32_t o = of & 0xFFFFu, fit = mine ? of >> 16 : 0u;
    const uint32_t sh = at & 3u, span = fit + sh; // the run as dwords from the aligned byte before it
    const uint32_t* const src = (const uint32_t*)(pay + (piece0 + (uint64_t)b * piece_step) * cap + (at - sh));
    uint8_t* const st = stage + o - sh; // byte e of the span goes to st[e] (sh <= e < span)
    auto put = [&](uint32_t k, uint32_t w) {
#pragma unroll
      for (uint32_t t = 0; t < 4u; ++t) {
        const uint32_t e = 4u * k + t;
        if (e >= sh && e < span) st[e] = (uint8_t)(w >> (8u * t));
      }
    };
    const uint32_t k0 = l, k1 = l + 16u;
    const uint32_t w0 = 4u * k0 < span ? __builtin_nontemporal_load(src + k0) : 0u;
    const uint32_t w1 = 4u * k1 < span ? __builtin_nontemporal_load(src + k1) : 0u;
    if (4u * k0 < span) put(k0, w0);
    if (4u * k1 < span) put(k1, w1);
    for (uint32_t k = l + 32u; __ballot(4u * k < span) != 0ull; k += 16u) // (rare: more than ~124 of the tile's values in one bucket)
      if (4u * k < span) put(k, src[k]);
  }
  __syncthreads();
}
// the answer of the value at place `w` of the tile's sorted order (after bq_stage_runs): the stage holds the tile's answers in
// that order.  Only a tile with a run that did not fit its bucket asks which bucket the place belongs to (the last one whose
// offset is <= w: empty buckets share their successor's offset) and, past the bucket's fit, goes to the overflow list.
__device__ __forceinline__ uint32_t bq_pick(uint32_t w, const uint8_t* stage, const uint32_t* offfit, const uint32_t* __restrict__ tovf,
                                            const uint8_t* __restrict__ ovf_pay)
{
  if (offfit[BB_MAX_BINS] == 0u) return stage[w];
  uint32_t lo = 0, hi = BB_MAX_BINS;
  while (hi - lo > 1u) {
    const uint32_t mid = (lo + hi) >> 1;
    if ((offfit[mid] & 0xFFFFu) <= w) lo = mid;
    else hi = mid;
  }
  const uint32_t of = offfit[lo];
  const uint32_t rank = w - (of & 0xFFFFu), fit = of >> 16;
  if (rank < fit) return stage[w];
  return ovf_pay[(uint64_t)tovf[lo] + (rank - fit)];
}

// ---- back, level 2: the answers of the regions' lists to the slots of the bins' lists -------------------------------------
struct BloomBackArgs {
  const uint16_t* where;  // level 2: per slot of the bins' lists; level 1: per tile row, step and thread (places in the tiles' sorted order)
  const uint2* tab;
  const uint32_t* tovf;
  const uint8_t* pay_in;  // answers next to the list this level WROTE
  const uint8_t* ovf_pay;
  const BloomStatus* status; // a round whose overflow list overflowed is left alone (the host redoes it)
  uint64_t ovf_cap;
  uint64_t cap;           // slots per bucket of that list
  // level 2
  uint8_t* pay_out;       // answers next to the bins' lists (what level 1 reads)
  const uint32_t* seg_fill;
  uint64_t cap_in;
  uint32_t n_regions, buckets_per_seg, tiles_per_seg;
  // pieces mode (bloom_part_pieces_kernel): level 2 read n_pieces_in pieces per segment (fill_in, in_buckets as there) with gx
  // blocks per segment, tiles_per_seg = tile rows per PIECE; level 1 ran with g1 blocks (0: slots mode)
  const uint32_t* fill_in;
  uint32_t n_pieces_in, in_buckets, gx, g1;
  // level 1 (a thread per read, the tiles of bloom_fused_kernel)
  uint64_t n_reads;
  uint32_t len, k, m, n_tiles, steps, n_buckets;
  const uint16_t* surv_in;         // BQ_BLOOM, a pass over some of the hashes: the windows still in the race (NULL: all) ...
  uint16_t* surv_out;              // ... and those that still are after it (NULL: the last pass -- the hits are summed)
  uint64_t* hits;                  // BQ_BLOOM: per read (may be NULL)
  unsigned long long* total_hits;  // BQ_BLOOM: += the sum
  uint8_t* estimates;              // BQ_COUNT: [read][window], 0 for a window that emitted nothing
  uint32_t est_lds;                // BQ_COUNT: != 0: a tile's estimates (THREADS x windows bytes) are collected in dynamic LDS and go out
                                   // in whole vectors (16 bytes per thread and word at a stride of `windows` bytes: 17.9 ms of the call's 36)
};

template <uint32_t THREADS>
static __global__ __launch_bounds__(THREADS) void bloom_back2_kernel(const BloomBackArgs a)
{
  constexpr uint32_t TILE = THREADS * BB_PART_ITEMS;
  __shared__ uint32_t cnt[BB_MAX_BINS], gat[BB_MAX_BINS], offfit[BB_MAX_BINS + 1];
  __shared__ __attribute__((aligned(16))) uint8_t stage[TILE];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (bloom_round_failed(a.status, a.ovf_cap)) return;
  const uint32_t seg = blockIdx.y;
  const uint32_t r0 = seg * a.buckets_per_seg;
  const uint32_t r1 = r0 + a.buckets_per_seg < a.n_regions ? r0 + a.buckets_per_seg : a.n_regions;
  const uint32_t n_buckets = r1 - r0;
  const uint64_t fill = a.seg_fill[(size_t)seg * BB_CURSOR_STRIDE];
  const uint64_t s0 = (uint64_t)seg * a.cap_in, s1 = s0 + (fill < a.cap_in ? fill : a.cap_in);
  const uint64_t n_tiles = (s1 - s0 + TILE - 1) / TILE;
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t t0 = s0 + tile * TILE;
    uint32_t w[BB_PART_ITEMS];
#pragma unroll
    for (uint32_t j = 0; j < BB_PART_ITEMS; ++j) { // (asked for first: they arrive while the runs are staged)
      const uint64_t idx = t0 + (uint64_t)j * THREADS + tid;
      w[j] = idx < s1 ? (uint32_t)__builtin_nontemporal_load(a.where + idx) : BQ_NONE;
    }
    const uint64_t row = ((uint64_t)seg * a.tiles_per_seg + tile) * a.buckets_per_seg;
    bq_stage_runs<THREADS / 64u>(a.tab + row, n_buckets, a.pay_in, (uint64_t)r0, 1ull, a.cap, stage, cnt, gat, offfit, tid, lane, wave);
#pragma unroll
    for (uint32_t j = 0; j < BB_PART_ITEMS; ++j) {
      const uint64_t idx = t0 + (uint64_t)j * THREADS + tid;
      if (w[j] != BQ_NONE) a.pay_out[idx] = (uint8_t)bq_pick(w[j], stage, offfit, a.tovf + row, a.ovf_pay);
    }
    __syncthreads(); // (stage / offfit are the next tile's)
  }
}

// back, level 2, pieces mode: the tiles of bloom_part_pieces_kernel (piece p of segment s was sorted by block p % gx of the segment)
template <uint32_t THREADS>
static __global__ __launch_bounds__(THREADS) void bloom_back2_pieces_kernel(const BloomBackArgs a)
{
  constexpr uint32_t TILE = THREADS * BB_PART_ITEMS;
  __shared__ uint32_t cnt[BB_MAX_BINS], gat[BB_MAX_BINS], offfit[BB_MAX_BINS + 1];
  __shared__ __attribute__((aligned(16))) uint8_t stage[TILE];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (bloom_round_failed(a.status, a.ovf_cap)) return;
  const uint32_t seg = blockIdx.y;
  const uint32_t r0 = seg * a.buckets_per_seg;
  const uint32_t r1 = r0 + a.buckets_per_seg < a.n_regions ? r0 + a.buckets_per_seg : a.n_regions;
  const uint32_t n_buckets = r1 - r0;
  for (uint32_t p = blockIdx.x; p < a.n_pieces_in; p += gridDim.x) {
    const uint64_t f = a.fill_in[(size_t)p * a.in_buckets + seg];
    const uint32_t n_here = (uint32_t)(f < a.cap_in ? f : a.cap_in);
    const uint64_t base = ((uint64_t)seg * a.n_pieces_in + p) * a.cap_in;
    const uint64_t piece0 = (uint64_t)r0 * a.gx + p % a.gx;
    for (uint32_t kk = 0; kk * TILE < n_here; ++kk) {
      const uint32_t t0 = kk * TILE;
      uint32_t w[BB_PART_ITEMS];
#pragma unroll
      for (uint32_t j = 0; j < BB_PART_ITEMS; ++j) {
        const uint32_t idx = t0 + j * THREADS + tid;
        w[j] = idx < n_here ? (uint32_t)__builtin_nontemporal_load(a.where + base + idx) : BQ_NONE;
      }
      const uint64_t row = (((uint64_t)seg * a.n_pieces_in + p) * a.tiles_per_seg + kk) * a.buckets_per_seg;
      bq_stage_runs<THREADS / 64u>(a.tab + row, n_buckets, a.pay_in, piece0, (uint64_t)a.gx, a.cap, stage, cnt, gat, offfit, tid, lane, wave);
#pragma unroll
      for (uint32_t j = 0; j < BB_PART_ITEMS; ++j) {
        const uint32_t idx = t0 + j * THREADS + tid;
        if (w[j] != BQ_NONE) a.pay_out[base + idx] = (uint8_t)bq_pick(w[j], stage, offfit, a.tovf + row, a.ovf_pay);
      }
      __syncthreads();
    }
  }
}

// ---- back, level 1: a thread per read, the tiles of bloom_fused_kernel<BF_PART, THREADS, true> -----------------------------
template <int KIND, uint32_t THREADS>
static __global__ __launch_bounds__(THREADS) void bloom_back1_kernel(const BloomBackArgs a)
{
  __shared__ uint32_t cnt[BB_MAX_BINS], gat[BB_MAX_BINS], offfit[BB_MAX_BINS + 1];
  __shared__ __attribute__((aligned(16))) uint8_t stage[THREADS * 16u];
  extern __shared__ __attribute__((aligned(16))) uint8_t bq_est_tile[]; // BQ_COUNT with est_lds: [thread][window]
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t k = a.k, m = a.m;
  const uint32_t kmod = (k - 1u) & 15u, jb = (k - 1u) >> 4;
  const uint32_t nwin = a.len - k + 1u;
  if (bloom_round_failed(a.status, a.ovf_cap)) return;
  unsigned long long mine = 0;
  for (uint32_t t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
    const uint64_t run0 = (uint64_t)t * THREADS;
    const uint64_t left = a.n_reads - run0;
    const bool live = tid < left;
    uint32_t hits = 0;
    for (uint32_t s = 0; s < a.steps; ++s) {
      const uint32_t j = jb + s, s0 = j << 4;
      const uint32_t lo = j == jb ? kmod : 0u;
      const uint32_t hi = a.len - s0 < 16u ? a.len - s0 : 16u;
      uint32_t ok = 0, all = 0xFFFFu; // windows of this word that emitted / whose m answers are all 1
      uint32_t est[16];
      if constexpr (KIND == BQ_COUNT) {
#pragma unroll
        for (uint32_t i = 0; i < 16; ++i) est[i] = 255u;
      }
      for (uint32_t jj = 0; jj < m; ++jj) {
        const uint64_t ts = ((uint64_t)t * a.steps + s) * m + jj;
        uint32_t w[16];
#pragma unroll
        for (uint32_t i = 0; i < 16; ++i)
          w[i] = (i >= lo && i < hi) ? (uint32_t)__builtin_nontemporal_load(a.where + (ts * 16u + i) * THREADS + tid) : BQ_NONE;
        bq_stage_runs<THREADS / 64u>(a.tab + ts * a.n_buckets, a.n_buckets, a.pay_in, a.g1 ? (uint64_t)(t % a.g1) : 0ull, a.g1 ? (uint64_t)a.g1 : 1ull,
                                     a.cap, stage, cnt, gat, offfit, tid, lane, wave);
#pragma unroll
        for (uint32_t i = 0; i < 16; ++i)
          if (w[i] != BQ_NONE) {
            const uint32_t p = bq_pick(w[i], stage, offfit, a.tovf + ts * a.n_buckets, a.ovf_pay);
            ok |= 1u << i;
            if constexpr (KIND == BQ_BLOOM) {
              if (!p) all &= ~(1u << i);
            } else {
              est[i] = p < est[i] ? p : est[i];
            }
          }
        __syncthreads(); // (stage / offfit are the next row's)
      }
      if constexpr (KIND == BQ_BLOOM) {
        // (a later pass emitted the survivors only: `ok` is inside surv_in already)
        if (a.surv_out) a.surv_out[((uint64_t)t * a.steps + s) * THREADS + tid] = (uint16_t)(ok & all);
        hits += (uint32_t)__builtin_popcount(ok & all); // (a pass that is not the last: the k-mers still in the race)
      }
      else if (live && a.estimates) {
        uint8_t* const dst = a.est_lds ? bq_est_tile + (size_t)tid * nwin : a.estimates + (run0 + tid) * nwin;
#pragma unroll
        for (uint32_t i = 0; i < 16; ++i)
          if (i >= lo && i < hi) dst[s0 + i - (k - 1u)] = (uint8_t)(((ok >> i) & 1u) ? est[i] : 0u);
      }
    }
    if constexpr (KIND == BQ_COUNT) {
      if (a.est_lds && a.estimates) { // the tile's estimates: one contiguous piece of the output, in whole vectors where they are aligned
        __syncthreads();
        const uint32_t n_here = (uint32_t)((left < THREADS ? left : THREADS) * nwin);
        uint8_t* const out = a.estimates + run0 * nwin;
        const uint32_t head = (uint32_t)((16u - ((uintptr_t)out & 15u)) & 15u) < n_here ? (uint32_t)((16u - ((uintptr_t)out & 15u)) & 15u) : n_here;
        if (tid < head) out[tid] = bq_est_tile[tid];
        const uint32_t nv = (n_here - head) >> 4;
        for (uint32_t i = tid; i < nv; i += THREADS) { // (LDS side: head is not a multiple of 16 in general -- bytes, four dwords at a time)
          const uint8_t* const sp = bq_est_tile + head + (size_t)i * 16u;
          uint4 v;
          __builtin_memcpy(&v, sp, 16);
          *(uint4*)(out + head + (size_t)i * 16u) = v;
        }
        const uint32_t done = head + nv * 16u;
        if (tid < n_here - done) out[done + tid] = bq_est_tile[done + tid];
        __syncthreads();
      }
    }
    if constexpr (KIND == BQ_BLOOM) {
      if (live) {
        if (a.hits && !a.surv_out) a.hits[run0 + tid] = hits;
        mine += hits;
      }
    }
  }
  if constexpr (KIND == BQ_BLOOM) {
    for (int d = 32; d > 0; d >>= 1) mine += (unsigned long long)__shfl_xor((long long)mine, d, 64);
    if (lane == 0 && mine && a.total_hits) atomicAdd(a.total_hits, mine);
  }
}

// ---- the binned query of a hash STREAM (reads of any lengths, spaced seeds, nthip_stream_*_query) -------------------------------
// Level 1 is bloom_part_kernel<true, THREADS, true> on the stream (tiles of THREADS x 16 values, `where` per value, `tab` per
// tile); the way back of its tiles gives one answer byte per VALUE, next to the stream: ans[i].
template <uint32_t THREADS, uint32_t M = 1>
static __global__ __launch_bounds__(THREADS) void bloom_back1_stream_kernel(const BloomBackArgs a, uint64_t n_inputs, uint8_t* __restrict__ ans)
{
  // (M > 1: the tiles of bloom_part_stream_pieces_kernel<.., M> -- 16 / M inputs per thread, value v = input * M + j)
  constexpr uint32_t IN = BB_PART_ITEMS / M, TILE = THREADS * IN;
  __shared__ uint32_t cnt[BB_MAX_BINS], gat[BB_MAX_BINS], offfit[BB_MAX_BINS + 1];
  __shared__ __attribute__((aligned(16))) uint8_t stage[THREADS * BB_PART_ITEMS];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (bloom_round_failed(a.status, a.ovf_cap)) return;
  const uint64_t n_tiles = (n_inputs + TILE - 1) / TILE;
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t t0 = tile * TILE;
    uint32_t w[BB_PART_ITEMS];
#pragma unroll
    for (uint32_t u = 0; u < IN; ++u)
#pragma unroll
      for (uint32_t j = 0; j < M; ++j) {
        const uint64_t idx = t0 + (uint64_t)u * THREADS + tid;
        w[u * M + j] = idx < n_inputs ? (uint32_t)__builtin_nontemporal_load(a.where + idx * M + j) : BQ_NONE;
      }
    const uint64_t row = tile * a.n_buckets;
    // (pieces mode, g1 blocks at level 1: the tile was written by block tile % g1 into ITS piece of every bucket)
    bq_stage_runs<THREADS / 64u>(a.tab + row, a.n_buckets, a.pay_in, a.g1 ? tile % a.g1 : 0ull, a.g1 ? (uint64_t)a.g1 : 1ull, a.cap, stage, cnt, gat,
                                 offfit, tid, lane, wave);
#pragma unroll
    for (uint32_t u = 0; u < IN; ++u)
#pragma unroll
      for (uint32_t j = 0; j < M; ++j) {
        const uint64_t idx = t0 + (uint64_t)u * THREADS + tid;
        if (w[u * M + j] != BQ_NONE) ans[idx * M + j] = (uint8_t)bq_pick(w[u * M + j], stage, offfit, a.tovf + row, a.ovf_pay);
      }
    __syncthreads();
  }
}

// a k-mer's m answers (consecutive in the stream) to ONE byte: filter: 1 when all are set (*found += the ones); sketch: the smallest
// (out may be ans when m == 1)
template <int KIND>
static __global__ __launch_bounds__(256) void answers_per_kmer_kernel(const uint8_t* ans, uint64_t n_kmers, uint32_t m,
                                                                      uint8_t* out, unsigned long long* __restrict__ found)
{
  uint32_t mine = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_kmers; i += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t v = KIND == BQ_BLOOM ? 1u : 255u;
    if (m <= 8u && out != ans) { // the m answer bytes in one unaligned 8-byte load (m > 1: `ans` is scratch with 8 bytes of slack behind it)
      uint64_t w;
      __builtin_memcpy(&w, ans + i * m, 8);
      for (uint32_t j = 0; j < m; ++j) {
        const uint32_t x = (uint32_t)(w >> (8u * j)) & 0xFFu;
        v = KIND == BQ_BLOOM ? (v & (x != 0u ? 1u : 0u)) : (x < v ? x : v);
      }
    } else {
      for (uint32_t j = 0; j < m; ++j) {
        const uint32_t x = ans[i * m + j];
        v = KIND == BQ_BLOOM ? (v & (x != 0u ? 1u : 0u)) : (x < v ? x : v);
      }
    }
    out[i] = (uint8_t)v;
    mine += KIND == BQ_BLOOM ? v : 0u;
  }
  if constexpr (KIND == BQ_BLOOM) {
    for (int d = 32; d > 0; d >>= 1) mine += (uint32_t)__shfl_xor((int)mine, d, 64);
    if ((threadIdx.x & 63u) == 0 && mine && found) atomicAdd(found, (unsigned long long)mine);
  }
}

// hits per READ from the answers of its k-mers' values (read r: k-mers roff[r] ... roff[r + 1], m values each): a wave per read
static __global__ __launch_bounds__(256) void answers_per_read_kernel(const uint8_t* __restrict__ ans, const uint64_t* __restrict__ roff,
                                                                      uint64_t n_reads, uint64_t n_kmers, uint32_t m, uint64_t* __restrict__ hits,
                                                                      unsigned long long* __restrict__ total_hits)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  uint64_t mine = 0;
  for (uint64_t r = wave; r < n_reads; r += n_waves) {
    const uint64_t i0 = roff[r], i1 = r + 1 < n_reads ? roff[r + 1] : n_kmers;
    uint32_t found = 0;
    if (m <= 8u) { // the k-mer's m answer bytes (0 / 1) in ONE unaligned 8-byte load (the array has 8 bytes of slack behind it)
      const uint64_t want = m == 8u ? 0x0101010101010101ull : (0x0101010101010101ull & ((1ull << (8u * m)) - 1ull));
      for (uint64_t i = i0 + lane; i < i1; i += 64u) {
        uint64_t v;
        __builtin_memcpy(&v, ans + i * m, 8);
        found += (v & want) == want ? 1u : 0u;
      }
    } else {
      for (uint64_t i = i0 + lane; i < i1; i += 64u) {
        uint32_t v = 1u;
        for (uint32_t j = 0; j < m; ++j) v &= ans[i * m + j] != 0u ? 1u : 0u;
        found += v;
      }
    }
    for (int d = 32; d > 0; d >>= 1) found += (uint32_t)__shfl_xor((int)found, d, 64);
    if (lane == 0) {
      if (hits) hits[r] = found;
      mine += found;
    }
  }
  if (lane == 0 && mine) atomicAdd(total_hits, (unsigned long long)mine);
}

} // namespace ntamd
