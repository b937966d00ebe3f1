// bloom_binned_kernels.hpp -- Bloom-filter insert that is not bound by device atomics.
//
// What the reference's callers do with hashes() (btllib Bloom filters, include/nthash/nthash.hpp:14-17, 56-57 of the
// reference): every value sets bit `h mod n_bits`.  Scattered 4-byte atomics retire at ~27 G/s on MI355X whatever the
// scope or the filter size (tools/bench_micro/atomics.hip) -- 4 % of the rate the hashes are produced at.  Here the
// values of a batch are brought to the filter region by region instead:
//
//   region   2^20 bits of the filter = 128 KiB: what one workgroup keeps in LDS
//   bin      128 consecutive regions = 2^27 bits
//
//   hist     one pass over the values: how many fall in every region (LDS counters, one flush per block)
//   scan     region bases (exclusive scan), bin bases = every 128th of them; cursors start there
//   part 1   values -> 27-bit offsets, bin by bin          (tiles of 8192 values sorted by bin in LDS, whole runs out)
//   part 2   bin by bin: offsets -> 20-bit offsets, region by region   (the same kernel on 4-byte input)
//   apply    a workgroup per region: its offsets into a zeroed 128 KiB of LDS (ds_or), then the lines of the filter
//            that got a bit are read, OR-ed and written back -- plain loads and stores, the region has one owner
//
// Every byte moves in whole runs: 8 + (8 + 4) + (4 + 4) + 4 B per value plus one read-modify-write of the touched
// filter lines per batch.  These EXACT lists (round 3) guess no capacity and have no overflow path: repeated values
// (low-complexity sequence) only make one region's list longer.  Round 4 put SLOTS MODE in front of them (below: no hist,
// no scan, buckets of mean + 8 sigma entries and an overflow list); the exact lists are what a round falls back to when
// its values are too skewed for that.  Filters of at most 2^35 bits (32768 regions: the histogram's LDS); larger ones and
// batches too small to amortise the filter pass keep the atomic kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "bloom_math.hpp"
#include "nt_math.hpp" // (mix_hash: the values of a stream that holds hashes()[0] only)

namespace ntamd {

typedef uint32_t bb_v4u __attribute__((ext_vector_type(4)));
typedef uint64_t bb_v2ul __attribute__((ext_vector_type(2)));

constexpr uint32_t BB_REGION_SHIFT = 20;            // bits of the filter per region: 2^20 = 128 KiB
constexpr uint32_t BB_BIN_SHIFT = 27;               // ... per bin: 128 regions
constexpr uint32_t BB_REGIONS_PER_BIN = 1u << (BB_BIN_SHIFT - BB_REGION_SHIFT);
constexpr uint32_t BB_MAX_REGIONS = 32768;          // 2^35 bits
constexpr uint32_t BB_MAX_BINS = 256;
#ifndef BB_ABL
#define BB_ABL 0
#endif
#ifndef BB_TIMING
#define BB_TIMING 0 // 1: one block of the second level prints the phase times of its tiles (-DBB_TIMING=1 build, tools/ab_build.sh)
#endif
#ifndef BB_COPY_SLOT
#define BB_COPY_SLOT 0
#endif
#ifndef BB_PART_ITEMS_N
#define BB_PART_ITEMS_N 16
#endif
constexpr uint32_t BB_PART_ITEMS = BB_PART_ITEMS_N; // values per thread and tile
// threads per block of the two partition levels: a tile is 16 values per thread.  From the values to the bins, tiles of
// 16 Ki (runs of 64 offsets per bin: 256-byte pieces) -- in-process rocprofv3, 2^31 values: 11.0 -> 7.1 ms against tiles
// of 8 Ki; from a bin to its 128 regions the smaller tile already makes such runs and leaves four blocks per CU (8.0 ms;
// 16 Ki: 8.2).  Without its stores the first level takes 4.9 ms: what is left is what HBM makes of 4-byte-granular runs.
#ifndef BB_L1_THREADS
#define BB_L1_THREADS 1024
#endif
#ifndef BB_L2_THREADS
#define BB_L2_THREADS 512
#endif
constexpr uint32_t BB_APPLY_THREADS = 1024;
constexpr uint32_t BB_APPLY_BATCH = 6; // 16-byte loads of the region's entries a thread has in flight
constexpr uint32_t BB_REGION_DWORDS = 1u << (BB_REGION_SHIFT - 5);
#ifndef BB_CURSOR_STRIDE
#define BB_CURSOR_STRIDE 32u // dwords between two cursors: one cache line each (tiles of every block hit the same few
#endif                       // hundred cursors; neighbours in one line serialise in L2 -- profiles/r03_notes.md)

// ---- slots mode (round 4): no histogram at all ------------------------------------------------------------------------
// The exact lists need every region's count BEFORE the first value moves: a pass over the values (hist, 8 B per value
// read) or, without a hash stream, a whole extra hashing of the reads (bloom_fused_kernels.hpp pass COUNT: 2.4 of a
// round's 13.4 ms).  Hash values are uniform, so a bucket's share of a round is known to a few standard deviations
// without looking: in slots mode every bucket OWNS `cap` = mean + 8 sqrt(mean) + 256 entries (rounded up to 64) of its
// level's list, its cursor counts from 0, and what does not fit -- a k-mer repeated a million times in real data --
// goes, as a full 64-bit position, to an overflow list that a test-then-atomic kernel applies after the regions
// (a bit already set / a counter already at 255 costs a load, not an atomic).  Only when that list overflows too
// (more than 1/64 of the round) does the round fail: apply and overflow kernels see it in BloomStatus and leave the
// table untouched, the host redoes the round on the exact lists and keeps to them for that table.
struct BloomStatus {
  unsigned long long ovf_n; // values sent to the overflow list (those past ovf_cap were dropped: the round failed)
  unsigned long long lost;  // fused pass: windows that hold a non-base (not emitted)
};
struct BloomSlots {
  uint64_t cap;             // entries per bucket of the level written (0: exact lists, absolute cursors)
  uint64_t* ovf;
  BloomStatus* status;
  uint64_t ovf_cap;
};
__device__ __forceinline__ bool bloom_round_failed(const BloomStatus* st, uint64_t ovf_cap)
{
  return st && __builtin_nontemporal_load(&st->ovf_n) > ovf_cap;
}
// a wave appends the entries [fit, cnt) of a tile's bucket run to the overflow list as full positions
// (returns the run's first index in the overflow list -- the binned QUERY finds its answers there, bloom_query_kernels.hpp)
__device__ __forceinline__ unsigned long long bloom_overflow_run(const BloomSlots& sl, const uint32_t* run, uint32_t fit, uint32_t cnt,
                                                                 uint64_t bucket_pos, uint32_t lane)
{
  unsigned long long base = 0;
  if (lane == 0) base = atomicAdd(&sl.status->ovf_n, (unsigned long long)(cnt - fit));
  base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32) |
         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
  for (uint32_t q = fit + lane; q < cnt; q += 64u) {
    const unsigned long long idx = base + (q - fit);
    if (idx < sl.ovf_cap) sl.ovf[idx] = bucket_pos | run[q];
  }
  return base;
}

// The copy-out of a sorted tile: wave w takes the buckets w, w + NW, w + 2 NW, ...  A lane holds the count, the place in the
// tile and the place in the list of ONE of them (three LDS reads per wave instead of three per bucket), the loop takes
// them back with v_readlane, four buckets at a time -- their first 128 entries read from LDS before the first store is
// issued.  (One bucket after the other, every LDS read waited for: 4.8 of a tile's 10 us on the second level.)
// exact lists: gbase = the run's place in `out`; slots mode: relative to bucket (bucket0 + b)'s own cap entries.
template <uint32_t NW, bool QUERY = false>
__device__ __forceinline__ void bloom_copy_out(const uint32_t* sorted, const uint32_t* hist, const uint32_t* off, const uint32_t* gbase,
                                               uint32_t n_buckets, uint32_t wave_v, uint32_t lane, uint32_t* out, uint64_t bucket0,
                                               const BloomSlots& sl, uint32_t shift, uint32_t* tovf = nullptr)
{
  // (tovf, the binned query: tovf[b] = where the tile's overflowing entries of bucket b start in the overflow list)
  static_assert(NW >= 4, "at most 64 buckets per wave");
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_v); // (uniform, and known to be: scalar address arithmetic)
  const uint32_t myb = wave + lane * NW;
  uint32_t mc = 0, mo = 0, mg = 0;
  if (myb < n_buckets) {
    mc = hist[myb];
    mo = off[myb];
    mg = gbase[myb];
  }
  const uint32_t n_mine = n_buckets > wave ? (n_buckets - wave + NW - 1u) / NW : 0u;
  const uint64_t cap = sl.cap;
  for (uint32_t i0 = 0; i0 < n_mine; i0 += 4u) {
    uint32_t c[4], o[4], fit[4], v0[4], v1[4];
    uint32_t* dst[4];
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      const uint32_t i = i0 + u < n_mine ? i0 + u : i0;
      c[u] = i0 + u < n_mine ? (uint32_t)__builtin_amdgcn_readlane((int)mc, (int)i) : 0u;
      o[u] = (uint32_t)__builtin_amdgcn_readlane((int)mo, (int)i);
      const uint32_t at = (uint32_t)__builtin_amdgcn_readlane((int)mg, (int)i);
      if (cap == 0) {
        fit[u] = c[u];
        dst[u] = out + at;
      } else {
        fit[u] = at >= cap ? 0u : (cap - at < c[u] ? (uint32_t)(cap - at) : c[u]);
        dst[u] = out + (bucket0 + wave + (uint64_t)i * NW) * cap + at;
      }
      v0[u] = lane < fit[u] ? sorted[o[u] + lane] : 0u;
      v1[u] = lane + 64u < fit[u] ? sorted[o[u] + 64u + lane] : 0u;
    }
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      if (lane < fit[u]) dst[u][lane] = v0[u];
      if (lane + 64u < fit[u]) dst[u][lane + 64u] = v1[u];
    }
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) { // (rare: a bucket with more than 128 of the tile's values, a bucket that is full)
      for (uint32_t j = lane + 128u; j < fit[u]; j += 64u) dst[u][j] = sorted[o[u] + j];
      if constexpr (QUERY) {
        if (fit[u] < c[u]) {
          const unsigned long long ob = bloom_overflow_run(sl, sorted + o[u], fit[u], c[u], (bucket0 + wave + (uint64_t)(i0 + u) * NW) << shift, lane);
          if (lane == 0) tovf[wave + (i0 + u) * NW] = (uint32_t)(ob < 0xFFFFFFFFull ? ob : 0xFFFFFFFFull);
        }
      } else {
        if (fit[u] < c[u]) (void)bloom_overflow_run(sl, sorted + o[u], fit[u], c[u], (bucket0 + wave + (uint64_t)(i0 + u) * NW) << shift, lane);
      }
    }
  }
}

// bloom_copy_out for tiles of MANY buckets (runs of ~64 entries): sixteen lanes per bucket, four buckets per instruction -- what a
// bucket needs is vector arithmetic done once per four (round 5: one bucket per wave at a time spends ~100 mostly scalar
// instructions per run, and a CU issues one scalar instruction per cycle for all its waves; see bloom_copy_out_lines).
// The same runs to the same places as bloom_copy_out.
template <uint32_t NW, bool QUERY = false>
__device__ __forceinline__ void bloom_copy_out_groups(const uint32_t* sorted, const uint32_t* hist, const uint32_t* off, const uint32_t* gbase,
                                                      uint32_t n_buckets, uint32_t wave_v, uint32_t lane, uint32_t* out, uint64_t bucket0,
                                                      const BloomSlots& sl, uint32_t shift, uint32_t* tovf = nullptr)
{
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_v);
  const uint64_t cap = sl.cap;
  const uint32_t g = lane >> 4, l = lane & 15u;
  const uint32_t n_mine = n_buckets > wave ? (n_buckets - wave + NW - 1u) / NW : 0u;
  for (uint32_t i0 = 0; i0 < n_mine; i0 += 4u) {
    const bool mine = i0 + g < n_mine;
    const uint32_t b = wave + (mine ? i0 + g : i0) * NW;
    const uint32_t c = mine ? hist[b] : 0u, o = off[b], at = gbase[b];
    uint32_t fit = c;
    uint32_t* dst = out + at; // exact lists: the run's place in `out`
    if (cap != 0) {           // slots mode: relative to the bucket's own cap entries
      fit = at >= cap ? 0u : (cap - at < c ? (uint32_t)(cap - at) : c);
      dst = out + (bucket0 + b) * cap + at;
    }
    const uint32_t* const sb = sorted + o;
    {
      const uint32_t q0 = l, q1 = l + 16u, q2 = l + 32u, q3 = l + 48u;
      const uint32_t v0 = q0 < fit ? sb[q0] : 0u, v1 = q1 < fit ? sb[q1] : 0u, v2 = q2 < fit ? sb[q2] : 0u, v3 = q3 < fit ? sb[q3] : 0u;
      if (q0 < fit) dst[q0] = v0;
      if (q1 < fit) dst[q1] = v1;
      if (q2 < fit) dst[q2] = v2;
      if (q3 < fit) dst[q3] = v3;
    }
    for (uint32_t q = l + 64u; __ballot(q < fit) != 0ull; q += 32u) {
      const uint32_t v0 = q < fit ? sb[q] : 0u, v1 = q + 16u < fit ? sb[q + 16u] : 0u;
      if (q < fit) dst[q] = v0;
      if (q + 16u < fit) dst[q + 16u] = v1;
    }
    if (__ballot(fit < c) != 0ull) { // (rare) a bucket that is full: the overflow list, a bucket at a time
      for (uint32_t gg = 0; gg < 4u; ++gg) {
        const uint32_t cc = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)(gg * 16u)), ff = (uint32_t)__builtin_amdgcn_readlane((int)fit, (int)(gg * 16u));
        const uint32_t oo = (uint32_t)__builtin_amdgcn_readlane((int)o, (int)(gg * 16u)), bb = (uint32_t)__builtin_amdgcn_readlane((int)b, (int)(gg * 16u));
        if (ff < cc) {
          const unsigned long long ob = bloom_overflow_run(sl, sorted + oo, ff, cc, (bucket0 + bb) << shift, lane);
          if constexpr (QUERY) {
            if (lane == 0) tovf[bb] = (uint32_t)(ob < 0xFFFFFFFFull ? ob : 0xFFFFFFFFull);
          }
        }
      }
    }
  }
}

// ---- pieces mode (round 5): whole lines into block-private pieces ------------------------------------------------------
// What the appended runs cost is what HBM makes of them (tools/bench_micro/runs_append.hip, 8 GiB of runs of 48..80 entries
// to 256 lists): behind a cursor every block shares -- every run starts where another CU's ended, two of its three lines
// are written in part -- 4.0-4.8 ms; every block appending to its OWN piece of every list, whole 128-byte lines only:
// 2.1 ms while writing a quarter more bytes.  So in pieces mode bucket b's list is not one run of slots but one PIECE per
// writing block ((piece0 + b * piece_step) * cap entries into `out`), the block keeps the entries of a bucket that do not
// fill a line (fewer than 32) in LDS until the next tile brings the rest (`left`, `lcnt`), and its cursor is its own count
// (no global atomic, no round trip per tile): everything that reaches memory is a whole aligned line behind the block's
// last one.  An entry's place in its piece is still `count before the tile + rank in the tile`: the binned query's way back
// (bloom_query_kernels.hpp) does not care when it was written.  Entries past the piece's cap go to the overflow list as
// before.  The last lines of the pieces are written when the block is done (bloom_flush_lines).
// Sixteen lanes per bucket, four buckets per instruction: what a bucket's run needs -- its counts, its place, the 64-bit
// address -- is vector arithmetic done once per four buckets.  (One bucket per wave at a time with everything about it in
// scalar registers -- the shape of bloom_copy_out -- took 6-9 of a tile's 11-14 us WITHOUT its stores: a CU issues one scalar
// instruction per cycle for all its waves, and 256 bucket runs of ~100 instructions each are 25 K cycles.)
template <uint32_t NW, bool QUERY = false>
__device__ __forceinline__ void bloom_copy_out_lines(const uint32_t* sorted, const uint32_t* hist, const uint32_t* off, const uint32_t* gbase,
                                                     uint32_t* left, uint32_t* lcnt, uint32_t n_buckets, uint32_t wave_v, uint32_t lane,
                                                     uint32_t* out, uint64_t piece0, uint64_t piece_step, uint64_t bucket0, const BloomSlots& sl,
                                                     uint32_t shift, uint32_t* tovf = nullptr)
{
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_v);
  const uint32_t cap = (uint32_t)sl.cap; // (a multiple of 32)
  const uint32_t g = lane >> 4, l = lane & 15u;
  const uint32_t n_mine = n_buckets > wave ? (n_buckets - wave + NW - 1u) / NW : 0u;
  for (uint32_t i0 = 0; i0 < n_mine; i0 += 4u) {
    const bool mine = i0 + g < n_mine;
    const uint32_t b = wave + (mine ? i0 + g : i0) * NW;
    const uint32_t c = mine ? hist[b] : 0u, o = off[b], at = gbase[b], L = mine ? lcnt[b] : 0u;
    const uint32_t fit = at >= cap ? 0u : (cap - at < c ? cap - at : c); // what the piece still takes of the tile's run
    const uint32_t tot = L + fit, n = tot & ~31u, rem = tot - n;          // the whole lines that can go; what waits
    uint32_t* const dst = out + ((piece0 + (uint64_t)b * piece_step) * cap + ((at < cap ? at : cap) - L));
    uint32_t* const lb = left + b * 32u;
    const uint32_t* const sb = sorted + o - L; // the waiting entries, then the run: entry q is lb[q] (q < L) or sb[q]
    {
      // the first 64 entries of the four buckets: read, then written
      const uint32_t q0 = l, q1 = l + 16u, q2 = l + 32u, q3 = l + 48u;
      const uint32_t v0 = q0 < n ? (q0 < L ? lb[q0] : sb[q0]) : 0u;
      const uint32_t v1 = q1 < n ? (q1 < L ? lb[q1] : sb[q1]) : 0u;
      const uint32_t v2 = q2 < n ? sb[q2] : 0u;
      const uint32_t v3 = q3 < n ? sb[q3] : 0u;
      if (q0 < n) dst[q0] = v0;
      if (q1 < n) dst[q1] = v1;
      if (q2 < n) dst[q2] = v2;
      if (q3 < n) dst[q3] = v3;
    }
    for (uint32_t q = l + 64u; __ballot(q < n) != 0ull; q += 32u) { // (longer runs: 32 entries of each at a time)
      const uint32_t v0 = q < n ? sb[q] : 0u;
      const uint32_t v1 = q + 16u < n ? sb[q + 16u] : 0u;
      if (q < n) dst[q] = v0;
      if (q + 16u < n) dst[q + 16u] = v1;
    }
    // what does not fill a line waits in LDS for the next tile: behind the old ones when no line went, else the tail
    if (n) {
      if (l < rem) lb[l] = sb[n + l];
      if (l + 16u < rem) lb[l + 16u] = sb[n + l + 16u];
    } else {
      if (l < fit) lb[L + l] = sb[L + l];
      if (l + 16u < fit) lb[L + l + 16u] = sb[L + l + 16u];
    }
    if (mine && l == 0) lcnt[b] = rem;
    if (__ballot(fit < c) != 0ull) { // (rare) the entries past a piece's cap: the overflow list, a bucket at a time
      for (uint32_t gg = 0; gg < 4u; ++gg) {
        const uint32_t cc = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)(gg * 16u)), ff = (uint32_t)__builtin_amdgcn_readlane((int)fit, (int)(gg * 16u));
        const uint32_t oo = (uint32_t)__builtin_amdgcn_readlane((int)o, (int)(gg * 16u)), bb = (uint32_t)__builtin_amdgcn_readlane((int)b, (int)(gg * 16u));
        if (ff < cc) {
          const unsigned long long ob = bloom_overflow_run(sl, sorted + oo, ff, cc, (bucket0 + bb) << shift, lane);
          if constexpr (QUERY) {
            if (lane == 0) tovf[bb] = (uint32_t)(ob < 0xFFFFFFFFull ? ob : 0xFFFFFFFFull);
          }
        }
      }
    }
  }
}
// the block is done: the lines its pieces end with (in part), and how many entries every piece got (fill[b]; may be past the cap)
template <uint32_t NW>
__device__ __forceinline__ void bloom_flush_lines(const uint32_t* left, const uint32_t* lcnt, const uint32_t* pcur, uint32_t n_buckets, uint32_t wave,
                                                  uint32_t lane, uint32_t tid, uint32_t* out, uint64_t piece0, uint64_t piece_step, uint64_t cap,
                                                  uint32_t* fill)
{
  wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
  for (uint32_t b = wave; b < n_buckets; b += NW) {
    const uint32_t L = lcnt[b];
    const uint64_t cur = pcur[b] < cap ? pcur[b] : cap;
    if (lane < L) out[(piece0 + (uint64_t)b * piece_step) * cap + (cur - L) + lane] = left[b * 32u + lane];
  }
  if (tid < n_buckets) fill[tid] = pcur[tid];
}

// ---- hist: values per region ---------------------------------------------------------------------------------------
// dynamic LDS: n_regions counters.  counts[r] += ...; every block flushes the counters it touched.
// (region_shift: log2 of the table slots per region -- 20 for a filter's bits, 15 for a sketch's counters)
static __global__ __launch_bounds__(1024) void bloom_hist_kernel(const uint64_t* __restrict__ hashes, uint64_t n, uint64_t n_bits,
                                                                 uint64_t magic, uint32_t n_regions,
                                                                 uint32_t* __restrict__ counts, uint32_t region_shift)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t bb_lds[];
  for (uint32_t i = threadIdx.x; i < n_regions; i += blockDim.x) bb_lds[i] = 0;
  __syncthreads();
  // two values per load, from the first 16-byte boundary on
  const uint64_t head = (((uintptr_t)hashes & 8u) && n) ? 1u : 0u;
  const uint64_t n2 = (n - head) >> 1;
  const bb_v2ul* h2 = (const bb_v2ul*)(hashes + head);
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (uint64_t)gridDim.x * blockDim.x) {
    const bb_v2ul v = __builtin_nontemporal_load(h2 + i);
    atomicAdd(&bb_lds[(uint32_t)(mod_invariant(v.x, n_bits, magic) >> region_shift)], 1u);
    atomicAdd(&bb_lds[(uint32_t)(mod_invariant(v.y, n_bits, magic) >> region_shift)], 1u);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (head) atomicAdd(&bb_lds[(uint32_t)(mod_invariant(hashes[0], n_bits, magic) >> region_shift)], 1u);
    if ((n - head) & 1u) atomicAdd(&bb_lds[(uint32_t)(mod_invariant(hashes[n - 1], n_bits, magic) >> region_shift)], 1u);
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n_regions; i += blockDim.x) {
    const uint32_t v = bb_lds[i];
    if (v) atomicAdd(&counts[i], v);
  }
}

// ---- scan: one block.  region_base[0 .. n_regions] = exclusive scan of counts; cursors = copies of the bases ----------
static __global__ __launch_bounds__(1024) void bloom_scan_kernel(const uint32_t* __restrict__ counts, uint32_t n_regions,
                                                                 uint32_t* __restrict__ region_base,
                                                                 uint32_t* __restrict__ region_cursor,
                                                                 uint32_t* __restrict__ bin_cursor)
{
  __shared__ uint32_t part[1024];
  const uint32_t per = (n_regions + 1023u) / 1024u; // <= 32
  const uint32_t r0 = threadIdx.x * per;
  uint32_t sum = 0;
  for (uint32_t i = 0; i < per; ++i)
    if (r0 + i < n_regions) sum += counts[r0 + i];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) { // Hillis-Steele over the 1024 partial sums
    const uint32_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = part[threadIdx.x] - sum;
  for (uint32_t i = 0; i < per; ++i) {
    const uint32_t r = r0 + i;
    if (r < n_regions) {
      region_base[r] = run;
      region_cursor[(size_t)r * BB_CURSOR_STRIDE] = run;
      if ((r & (BB_REGIONS_PER_BIN - 1u)) == 0) bin_cursor[(size_t)(r >> (BB_BIN_SHIFT - BB_REGION_SHIFT)) * BB_CURSOR_STRIDE] = run;
      run += counts[r];
    }
  }
  if (threadIdx.x == 1023) region_base[n_regions] = part[1023];
}

// ---- part: one level of the partition --------------------------------------------------------------------------------
// IN64: the values themselves (bit position = h mod n_bits), one segment = the whole input; otherwise 4-byte offsets,
// segment s = bin s of the level before: [seg_base[s * seg_step], seg_base[(s + 1) * seg_step]).
// A value's bucket is (position >> shift) inside its segment, what is written is position & mask, at the bucket's cursor.
struct BloomPartArgs {
  const void* in;
  uint32_t* out;
  const uint32_t* seg_base; // !IN64: region_base (bin s = regions [128 s, 128 s + 128))
  uint32_t* cursor;         // per bucket, all segments: bucket b of segment s = cursor[s * buckets_per_seg + b]
  uint64_t n;               // IN64: values
  uint64_t n_bits, magic;
  uint32_t n_regions;
  uint32_t shift, mask;
  uint32_t buckets_per_seg; // IN64: the number of buckets; !IN64: 128 (the last segment may own fewer regions)
  // slots mode (sl.cap != 0): bucket g = seg * buckets_per_seg + b owns out[g * sl.cap ...), cursor[g] counts from 0;
  // !IN64: segment s is in[s * cap_in ... + min(seg_fill[s], cap_in))
  BloomSlots sl;
  uint64_t cap_in;
  const uint32_t* seg_fill;
};
// the binned query (QUERY instantiation, slots mode; bloom_query_kernels.hpp): what the way back needs of every tile --
// q_where[slot of the input list] = the entry's place in its tile's SORTED order (16 bits: a tile has at most 16 Ki entries); per tile (segment s, its tile
// t: row s * q_tiles_per_seg + t) and bucket b, q_tab[row * buckets_per_seg + b] = {entries of the tile, where the run
// went in the bucket's slots}, q_tovf[...] = where its overflowing entries start in the overflow list (set when any)
// (a struct of its own: the insert's instantiations keep the argument block -- and the register allocation -- they had)
struct BloomPartQueryArgs : BloomPartArgs {
  uint16_t* q_where;
  uint2* q_tab;
  uint32_t* q_tovf;
  uint32_t q_tiles_per_seg;
};

template <bool IN64, uint32_t BB_PART_THREADS, bool QUERY = false>
static __global__ __launch_bounds__(BB_PART_THREADS) void bloom_part_kernel(
    const typename std::conditional<QUERY, BloomPartQueryArgs, BloomPartArgs>::type a)
{
  constexpr uint32_t BB_TILE = BB_PART_THREADS * BB_PART_ITEMS;
  __shared__ uint32_t hist[BB_MAX_BINS];
  __shared__ uint32_t off[BB_MAX_BINS];
  __shared__ uint32_t gbase[BB_MAX_BINS];
  extern __shared__ __attribute__((aligned(16))) uint32_t bb_lds[];
  uint32_t* const sorted = bb_lds; // BB_TILE entries (dynamic: tiles of 16 Ki values and more pass the static limit)
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t seg = blockIdx.y;
  uint64_t s0, s1;
  uint32_t n_buckets;
  if constexpr (IN64) {
    s0 = 0;
    s1 = a.n;
    n_buckets = a.buckets_per_seg;
  } else {
    const uint32_t r0 = seg * a.buckets_per_seg;
    const uint32_t r1 = r0 + a.buckets_per_seg < a.n_regions ? r0 + a.buckets_per_seg : a.n_regions;
    if (a.sl.cap) {
      const uint64_t fill = a.seg_fill[(size_t)seg * BB_CURSOR_STRIDE];
      s0 = (uint64_t)seg * a.cap_in;
      s1 = s0 + (fill < a.cap_in ? fill : a.cap_in);
    } else {
      s0 = a.seg_base[r0];
      s1 = a.seg_base[r1];
    }
    n_buckets = r1 - r0;
  }
  uint32_t* const cursor = a.cursor + (size_t)seg * a.buckets_per_seg * BB_CURSOR_STRIDE;
  const uint64_t n_tiles = (s1 - s0 + BB_TILE - 1) / BB_TILE;
  // The values of the NEXT tile are asked for as soon as this tile's are ranked and arrive while it is scanned, sorted and
  // copied out (the barriers wait for LDS only): a block used to sit 12 of a tile's 19 us in front of its loads -- the
  // queueing delay of a memory system that every block had just filled at once (per-phase clocks of one block,
  // profiles/r04_notes.md 2c).
  typedef typename std::conditional<IN64, uint64_t, uint32_t>::type in_t;
  in_t pre[BB_PART_ITEMS];
  auto fetch = [&](uint64_t tile) {
    const uint64_t t0 = s0 + tile * BB_TILE;
#pragma unroll
    for (uint32_t j = 0; j < BB_PART_ITEMS; ++j) {
      const uint64_t idx = t0 + (uint64_t)j * BB_PART_THREADS + tid;
      pre[j] = 0;
      if (idx < s1) pre[j] = __builtin_nontemporal_load((const in_t*)a.in + idx);
    }
  };
  if (blockIdx.x < n_tiles) fetch(blockIdx.x);
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
#if BB_TIMING
    const uint64_t tk0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (tid < BB_MAX_BINS) hist[tid] = 0;
    __syncthreads();
    const uint64_t t0 = s0 + tile * BB_TILE;
    uint32_t val[BB_PART_ITEMS], where[BB_PART_ITEMS]; // where = bucket << 16 | rank inside the tile's bucket
    if (t0 + BB_TILE <= s1) {
      // a whole tile (all but a segment's last): no branch per value, so the thread's 16 LDS atomics are in flight together
      // (behind `if (idx < s1)` each one was waited for: an s_waitcnt lgkmcnt(0) per value in the ISA)
#pragma unroll
      for (uint32_t j = 0; j < BB_PART_ITEMS; ++j) {
        uint64_t p;
        if constexpr (IN64) p = mod_invariant(pre[j], a.n_bits, a.magic);
        else p = pre[j];
        const uint32_t b = (uint32_t)(p >> a.shift);
        val[j] = (uint32_t)p & a.mask;
        where[j] = (b << 16) | atomicAdd(&hist[b], 1u); // (a tile has at most 16 Ki values: the rank fits 16 bits)
      }
    } else {
#pragma unroll
      for (uint32_t j = 0; j < BB_PART_ITEMS; ++j) {
        const uint64_t idx = t0 + (uint64_t)j * BB_PART_THREADS + tid;
        where[j] = ~0u;
        val[j] = 0;
        if (idx < s1) {
          uint64_t p;
          if constexpr (IN64) p = mod_invariant(pre[j], a.n_bits, a.magic);
          else p = pre[j];
          const uint32_t b = (uint32_t)(p >> a.shift);
          val[j] = (uint32_t)p & a.mask;
          where[j] = (b << 16) | atomicAdd(&hist[b], 1u);
        }
      }
    }
    if (tile + gridDim.x < n_tiles) fetch(tile + gridDim.x);
    __syncthreads();
#if BB_TIMING
    const uint64_t tk1 = __builtin_amdgcn_s_memrealtime();
#endif
    uint32_t my_base = 0;
    uint64_t q_row = 0;
    if constexpr (QUERY) q_row = ((uint64_t)seg * a.q_tiles_per_seg + tile) * a.buckets_per_seg;
    if (tid < n_buckets) {
      const uint32_t c = hist[tid];
      // (the answer is wanted by the copy-out only: it travels while the tile is sorted)
      my_base = c ? atomicAdd(&cursor[(size_t)tid * BB_CURSOR_STRIDE], c) : 0u;
      if constexpr (QUERY) a.q_tab[q_row + tid] = make_uint2(c, my_base);
    }
    if (wave == 0) { // exclusive scan of the (at most 256) bucket counts: 4 per lane
      uint32_t c[4], s = 0;
#pragma unroll
      for (uint32_t i = 0; i < 4; ++i) {
        c[i] = hist[lane * 4u + i];
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
        off[lane * 4u + i] = run;
        run += c[i];
      }
    }
    __syncthreads();
#if BB_TIMING
    const uint64_t tk2 = __builtin_amdgcn_s_memrealtime();
#endif
#if BB_COPY_SLOT
    // copy-out a slot per lane: slot i of the sorted tile goes to its bucket's run (the bucket of a slot: one byte each)
    uint8_t* const sbin = (uint8_t*)(sorted + BB_TILE);
#pragma unroll
    for (uint32_t j = 0; j < BB_PART_ITEMS; ++j)
      if (where[j] != ~0u) {
        const uint32_t slot = off[where[j] >> 16] + (where[j] & 0xFFFFu);
        sorted[slot] = val[j];
        sbin[slot] = (uint8_t)(where[j] >> 16);
        if constexpr (QUERY) a.q_where[t0 + (uint64_t)j * BB_PART_THREADS + tid] = (uint16_t)slot;
      }
    if (tid < n_buckets) gbase[tid] = my_base;
    __syncthreads();
    const uint64_t left = s1 - t0;
    const uint32_t n_here = left < BB_TILE ? (uint32_t)left : BB_TILE;
    for (uint32_t i = tid; i < n_here; i += BB_PART_THREADS) {
      const uint32_t b = sbin[i];
      a.out[gbase[b] + (i - off[b])] = sorted[i];
    }
    __syncthreads();
#else
#pragma unroll
    for (uint32_t j = 0; j < BB_PART_ITEMS; ++j)
      if (where[j] != ~0u) {
        const uint32_t slot = off[where[j] >> 16] + (where[j] & 0xFFFFu);
        sorted[slot] = val[j];
        // the way back: every slot of the input list remembers its place in the tile's sorted order
        if constexpr (QUERY) a.q_where[t0 + (uint64_t)j * BB_PART_THREADS + tid] = (uint16_t)slot;
      }
    if (tid < n_buckets) gbase[tid] = my_base;
    __syncthreads();
#if BB_TIMING
    const uint64_t tk3 = __builtin_amdgcn_s_memrealtime();
#endif
#if BB_ABL == 1 // ablation (WRONG results): everything but the stores
    for (uint32_t b = wave; b < n_buckets; b += BB_PART_THREADS / 64u) {
      const uint32_t c = hist[b], o = off[b];
      uint32_t* const dst = a.out + gbase[b];
      for (uint32_t j = lane; j < c; j += 64u) asm volatile("" ::"v"(sorted[o + j]), "v"(dst));
    }
#else
    if (n_buckets * 192u >= BB_TILE) { // runs of at most ~192 entries on average: four buckets per instruction
      if constexpr (QUERY)
        bloom_copy_out_groups<BB_PART_THREADS / 64u, true>(sorted, hist, off, gbase, n_buckets, wave, lane, a.out, (uint64_t)seg * a.buckets_per_seg,
                                                           a.sl, a.shift, a.q_tovf + q_row);
      else
        bloom_copy_out_groups<BB_PART_THREADS / 64u>(sorted, hist, off, gbase, n_buckets, wave, lane, a.out, (uint64_t)seg * a.buckets_per_seg, a.sl,
                                                     a.shift);
    } else if constexpr (QUERY)
      bloom_copy_out<BB_PART_THREADS / 64u, true>(sorted, hist, off, gbase, n_buckets, wave, lane, a.out, (uint64_t)seg * a.buckets_per_seg, a.sl,
                                                  a.shift, a.q_tovf + q_row);
    else
      bloom_copy_out<BB_PART_THREADS / 64u>(sorted, hist, off, gbase, n_buckets, wave, lane, a.out, (uint64_t)seg * a.buckets_per_seg, a.sl, a.shift);
#endif
    __syncthreads();
#if BB_TIMING
    if (!IN64 && tid == 0 && blockIdx.x == 2u && blockIdx.y == 3u && tile < 100) {
      const uint64_t tk4 = __builtin_amdgcn_s_memrealtime();
      printf("tile %u: wait + rank %u  scan + cursors %u  sort %u  copy-out %u  (10 ns)\n", (unsigned)tile, (unsigned)(tk1 - tk0),
             (unsigned)(tk2 - tk1), (unsigned)(tk3 - tk2), (unsigned)(tk4 - tk3));
    }
#endif
#endif
  }
}

// ---- part, pieces mode: the second level between two lists of pieces -----------------------------------------------------
// Segment s (a bin) of the level before is n_pieces_in pieces -- piece p at in[(s * n_pieces_in + p) * cap_in ...), its
// entries min(fill_in[p * in_buckets + s], cap_in) -- and block x of the segment's gridDim.x blocks takes the pieces x, x +
// gridDim.x, ... tile by tile; bucket b of the segment (region s * buckets_per_seg + b) gets a piece per block:
// out[((s * buckets_per_seg + b) * gridDim.x + x) * sl.cap ...), fill_out[(s * gridDim.x + x) * buckets_per_seg + b] entries.
struct BloomPartPiecesArgs {
  const uint32_t* in;
  uint32_t* out;
  const uint32_t* fill_in;
  uint32_t* fill_out;
  uint64_t cap_in;
  uint32_t n_pieces_in, in_buckets;
  uint32_t n_regions, shift, mask, buckets_per_seg;
  BloomSlots sl; // cap: entries per piece written
  // the binned query: q_where per slot of `in`; tile row ((s * n_pieces_in + p) * q_tiles_per_piece + tile of the piece)
  uint16_t* q_where;
  uint2* q_tab;
  uint32_t* q_tovf;
  uint32_t q_tiles_per_piece;
};

template <uint32_t BB_PART_THREADS, bool QUERY = false>
static __global__ __launch_bounds__(BB_PART_THREADS) void bloom_part_pieces_kernel(const BloomPartPiecesArgs a)
{
  constexpr uint32_t BB_TILE = BB_PART_THREADS * BB_PART_ITEMS;
  __shared__ uint32_t hist[BB_MAX_BINS];
  __shared__ uint32_t off[BB_MAX_BINS];
  __shared__ uint32_t gbase[BB_MAX_BINS];
  __shared__ uint32_t lcnt[BB_MAX_BINS], pcur[BB_MAX_BINS];
  extern __shared__ __attribute__((aligned(16))) uint32_t bb_lds[];
  uint32_t* const sorted = bb_lds;       // BB_TILE entries
  uint32_t* const left = bb_lds + BB_TILE; // [bucket][32]
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t seg = blockIdx.y, gx = gridDim.x;
  const uint32_t r0 = seg * a.buckets_per_seg;
  const uint32_t r1 = r0 + a.buckets_per_seg < a.n_regions ? r0 + a.buckets_per_seg : a.n_regions;
  const uint32_t n_buckets = r1 - r0;
  if (tid < BB_MAX_BINS) {
    lcnt[tid] = 0;
    pcur[tid] = 0;
  }
  auto count_of = [&](uint32_t p) -> uint32_t { // (uniform)
    if (p >= a.n_pieces_in) return 0u;
    const uint64_t f = a.fill_in[(size_t)p * a.in_buckets + seg];
    return (uint32_t)(f < a.cap_in ? f : a.cap_in);
  };
  // the tile at hand: piece p, its tile kk; cnt = the piece's entries; cnt_next = those of piece p + gx (asked for a piece ahead)
  uint32_t p = blockIdx.x, kk = 0, cnt = count_of(p), cnt_next = count_of(p + gx);
  while (p < a.n_pieces_in && cnt == 0) {
    p += gx;
    cnt = cnt_next;
    cnt_next = count_of(p + gx);
  }
  uint32_t pre[BB_PART_ITEMS];
  auto fetch = [&](uint32_t pp, uint32_t k2, uint32_t c2) {
    const uint64_t base = ((uint64_t)seg * a.n_pieces_in + pp) * a.cap_in;
    const uint32_t t0 = k2 * BB_TILE;
#pragma unroll
    for (uint32_t j = 0; j < BB_PART_ITEMS; ++j) {
      const uint32_t idx = t0 + j * BB_PART_THREADS + tid;
      pre[j] = 0;
      if (idx < c2) pre[j] = __builtin_nontemporal_load(a.in + base + idx);
    }
  };
  if (p < a.n_pieces_in) fetch(p, 0, cnt);
  while (p < a.n_pieces_in) {
    if (tid < BB_MAX_BINS) hist[tid] = 0;
    __syncthreads();
    const uint64_t base = ((uint64_t)seg * a.n_pieces_in + p) * a.cap_in;
    const uint32_t t0 = kk * BB_TILE;
    uint32_t val[BB_PART_ITEMS], where[BB_PART_ITEMS]; // where = bucket << 16 | rank inside the tile's bucket
    if (t0 + BB_TILE <= cnt) {
#pragma unroll
      for (uint32_t j = 0; j < BB_PART_ITEMS; ++j) {
        const uint32_t b = pre[j] >> a.shift;
        val[j] = pre[j] & a.mask;
        where[j] = (b << 16) | atomicAdd(&hist[b], 1u);
      }
    } else {
#pragma unroll
      for (uint32_t j = 0; j < BB_PART_ITEMS; ++j) {
        const uint32_t idx = t0 + j * BB_PART_THREADS + tid;
        where[j] = ~0u;
        val[j] = 0;
        if (idx < cnt) {
          const uint32_t b = pre[j] >> a.shift;
          val[j] = pre[j] & a.mask;
          where[j] = (b << 16) | atomicAdd(&hist[b], 1u);
        }
      }
    }
    // the next tile: of this piece, or the first of the block's next piece that holds anything
    uint32_t np = p, nk = kk + 1u, ncnt = cnt;
    if (nk * BB_TILE >= cnt) {
      nk = 0;
      do {
        np += gx;
        ncnt = cnt_next;
        cnt_next = count_of(np + gx);
      } while (np < a.n_pieces_in && ncnt == 0);
    }
    if (np < a.n_pieces_in) fetch(np, nk, ncnt);
    __syncthreads();
    uint32_t my_base = 0;
    uint64_t q_row = 0;
    if constexpr (QUERY) q_row = (((uint64_t)seg * a.n_pieces_in + p) * a.q_tiles_per_piece + kk) * a.buckets_per_seg;
    if (tid < n_buckets) {
      const uint32_t c = hist[tid];
      my_base = pcur[tid];
      pcur[tid] = my_base + c;
      gbase[tid] = my_base;
      if constexpr (QUERY) a.q_tab[q_row + tid] = make_uint2(c, my_base);
    }
    if (wave == 0) { // exclusive scan of the (at most 256) bucket counts: 4 per lane
      uint32_t c[4], s = 0;
#pragma unroll
      for (uint32_t i = 0; i < 4; ++i) {
        c[i] = hist[lane * 4u + i];
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
        off[lane * 4u + i] = run;
        run += c[i];
      }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < BB_PART_ITEMS; ++j)
      if (where[j] != ~0u) {
        const uint32_t slot = off[where[j] >> 16] + (where[j] & 0xFFFFu);
        sorted[slot] = val[j];
        if constexpr (QUERY) a.q_where[base + t0 + j * BB_PART_THREADS + tid] = (uint16_t)slot;
      }
    __syncthreads();
    bloom_copy_out_lines<BB_PART_THREADS / 64u, QUERY>(sorted, hist, off, gbase, left, lcnt, n_buckets, wave, lane, a.out,
                                                       (uint64_t)r0 * gx + blockIdx.x, (uint64_t)gx, (uint64_t)r0, a.sl, a.shift,
                                                       QUERY ? a.q_tovf + q_row : nullptr);
    __syncthreads();
    p = np;
    kk = nk;
    cnt = ncnt;
  }
  __syncthreads();
  bloom_flush_lines<BB_PART_THREADS / 64u>(left, lcnt, pcur, n_buckets, wave, lane, tid, a.out, (uint64_t)r0 * gx + blockIdx.x, (uint64_t)gx, a.sl.cap,
                                           a.fill_out + ((size_t)seg * gx + blockIdx.x) * a.buckets_per_seg);
}

// ---- part, pieces mode, level 1 of a hash STREAM ------------------------------------------------------------------------
// Block x of the gridDim.x blocks takes the tiles x, x + gridDim.x, ... of the stream (BB_PART_THREADS * 16 values each) and
// owns piece x of every bucket (a bin of a two-level table): out[(b * gridDim.x + x) * sl.cap ...), fill_out[x * n_buckets + b]
// entries -- what bloom_part_pieces_kernel reads as n_pieces_in = gridDim.x, in_buckets = n_buckets.  What
// bloom_fused_kernel<..., PIECES> is to the reads (bloom_fused_kernels.hpp), for values that are in memory already (spaced-seed
// hashes, the compact stream of reads given by offsets, a caller's own stream).  dynamic LDS: the tile + 32 waiting entries
// per bucket.  QUERY: q_where per value, q_tab / q_tovf rows per tile (tile t: row t), as bloom_part_kernel.
struct BloomPartStreamPiecesArgs {
  const uint64_t* in;
  uint64_t n, n_bits, magic;
  uint32_t* out;
  uint32_t* fill_out;
  uint32_t shift, mask, n_buckets;
  BloomSlots sl; // cap: entries per piece
  uint16_t* q_where;
  uint2* q_tab;
  uint32_t* q_tovf;
  uint64_t kmul; // M > 1: k * MULTISEED
};

// M > 1: the stream holds ONE value per M -- hashes()[0] of a k-mer / of a seed's window -- and the kernel makes the other M - 1
// itself (extend_hashes, src/internal.hpp:104-118: h[j] = mix(h[0] * (j ^ k * MULTISEED))): a.n counts the INPUTS, a thread takes
// 16 / M of them per tile, value v of the full stream is h[v % M] of input v / M (q_where is indexed by v).  The stream that is
// written and read back shrinks M-fold: config 4's seed pair with 3 hashes per seed moves 17.6 GB each way instead of 53.
template <uint32_t BB_PART_THREADS, bool QUERY = false, uint32_t M = 1>
static __global__ __launch_bounds__(BB_PART_THREADS) void bloom_part_stream_pieces_kernel(const BloomPartStreamPiecesArgs a)
{
  constexpr uint32_t IN = BB_PART_ITEMS / M;            // inputs per thread and tile
  constexpr uint32_t BB_TILE = BB_PART_THREADS * IN;    // inputs per tile (M == 1: BB_PART_THREADS * 16)
  constexpr uint32_t BB_SORTED = BB_PART_THREADS * BB_PART_ITEMS; // room of the sorted tile (its values: BB_TILE * M)
  __shared__ uint32_t hist[BB_MAX_BINS];
  __shared__ uint32_t off[BB_MAX_BINS];
  __shared__ uint32_t gbase[BB_MAX_BINS];
  __shared__ uint32_t lcnt[BB_MAX_BINS], pcur[BB_MAX_BINS];
  extern __shared__ __attribute__((aligned(16))) uint32_t bb_lds[];
  uint32_t* const sorted = bb_lds;           // BB_SORTED entries
  uint32_t* const left = bb_lds + BB_SORTED; // [bucket][32]
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t n_buckets = a.n_buckets;
  if (tid < BB_MAX_BINS) {
    lcnt[tid] = 0;
    pcur[tid] = 0;
  }
  const uint64_t n_tiles = (a.n + BB_TILE - 1) / BB_TILE;
  uint64_t pre[IN];
  auto fetch = [&](uint64_t tile) {
    const uint64_t t0 = tile * BB_TILE;
#pragma unroll
    for (uint32_t j = 0; j < IN; ++j) {
      const uint64_t idx = t0 + (uint64_t)j * BB_PART_THREADS + tid;
      pre[j] = 0;
      if (idx < a.n) pre[j] = __builtin_nontemporal_load(a.in + idx);
    }
  };
  if (blockIdx.x < n_tiles) fetch(blockIdx.x);
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    if (tid < BB_MAX_BINS) hist[tid] = 0;
    __syncthreads();
    const uint64_t t0 = tile * BB_TILE;
    uint32_t val[BB_PART_ITEMS], where[BB_PART_ITEMS]; // where = bucket << 16 | rank inside the tile's bucket
    const bool whole = t0 + BB_TILE <= a.n; // (no test per value then: the thread's rank atomics are in flight together)
#pragma unroll
    for (uint32_t q = 0; q < BB_PART_ITEMS; ++q) {
      where[q] = ~0u;
      val[q] = 0;
    }
#pragma unroll
    for (uint32_t u = 0; u < IN; ++u) {
      const uint64_t idx = t0 + (uint64_t)u * BB_PART_THREADS + tid;
      if (whole || idx < a.n) {
#pragma unroll
        for (uint32_t j = 0; j < M; ++j) {
          const uint64_t h = j == 0 ? pre[u] : mix_hash(pre[u], (uint64_t)j ^ a.kmul);
          const uint64_t p = mod_invariant(h, a.n_bits, a.magic);
          const uint32_t b = (uint32_t)(p >> a.shift);
          val[u * M + j] = (uint32_t)p & a.mask;
          where[u * M + j] = (b << 16) | atomicAdd(&hist[b], 1u);
        }
      }
    }
    if (tile + gridDim.x < n_tiles) fetch(tile + gridDim.x);
    __syncthreads();
    const uint64_t q_row = tile * n_buckets;
    if (tid < n_buckets) {
      const uint32_t c = hist[tid];
      const uint32_t my_base = pcur[tid];
      pcur[tid] = my_base + c;
      gbase[tid] = my_base;
      if constexpr (QUERY) a.q_tab[q_row + tid] = make_uint2(c, my_base);
    }
    if (wave == 0) { // exclusive scan of the (at most 256) bucket counts: 4 per lane
      uint32_t c[4], s = 0;
#pragma unroll
      for (uint32_t i = 0; i < 4; ++i) {
        c[i] = hist[lane * 4u + i];
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
        off[lane * 4u + i] = run;
        run += c[i];
      }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < IN; ++u)
#pragma unroll
      for (uint32_t j = 0; j < M; ++j)
        if (where[u * M + j] != ~0u) {
          const uint32_t slot = off[where[u * M + j] >> 16] + (where[u * M + j] & 0xFFFFu);
          sorted[slot] = val[u * M + j];
          if constexpr (QUERY) a.q_where[(t0 + (uint64_t)u * BB_PART_THREADS + tid) * M + j] = (uint16_t)slot;
        }
    __syncthreads();
    bloom_copy_out_lines<BB_PART_THREADS / 64u, QUERY>(sorted, hist, off, gbase, left, lcnt, n_buckets, wave, lane, a.out, (uint64_t)blockIdx.x,
                                                       (uint64_t)gridDim.x, 0ull, a.sl, a.shift, QUERY ? a.q_tovf + q_row : nullptr);
    __syncthreads();
  }
  __syncthreads();
  bloom_flush_lines<BB_PART_THREADS / 64u>(left, lcnt, pcur, n_buckets, wave, lane, tid, a.out, (uint64_t)blockIdx.x, (uint64_t)gridDim.x, a.sl.cap,
                                           a.fill_out + (size_t)blockIdx.x * n_buckets);
}

// ---- apply: a workgroup per region -----------------------------------------------------------------------------------
// dynamic LDS: 128 KiB.  entries = 20-bit offsets of the region's values.
// (slots mode, cap != 0: region r's entries are all_entries[r * cap ... + min(fill[r], cap)); a failed round is left alone)
static __global__ __launch_bounds__(BB_APPLY_THREADS) void bloom_apply_kernel(const uint32_t* __restrict__ all_entries,
                                                                              const uint32_t* __restrict__ region_base,
                                                                              uint32_t n_regions, uint32_t* __restrict__ filter,
                                                                              uint64_t filter_dwords, uint64_t cap,
                                                                              const uint32_t* __restrict__ fill,
                                                                              const BloomStatus* status, uint64_t ovf_cap,
                                                                              uint32_t n_pieces = 0, uint32_t bps = 0)
{
  // (pieces mode, n_pieces != 0: region r is n_pieces pieces of cap entries -- piece x at (r * n_pieces + x) * cap, its entries
  // fill[((r / bps) * n_pieces + x) * bps + r % bps] -- see bloom_part_pieces_kernel)
  extern __shared__ __attribute__((aligned(16))) uint32_t bb_lds[];
  uint4* const l4 = (uint4*)bb_lds;
  if (bloom_round_failed(status, ovf_cap)) return;
  for (uint32_t r = blockIdx.x; r < n_regions; r += gridDim.x) {
   const uint32_t n_runs = n_pieces ? n_pieces : 1u;
   bool zeroed = false;
   for (uint32_t x = 0; x < n_runs; ++x) {
    uint32_t e0, e1;
    const uint32_t* entries = all_entries;
    if (n_pieces) {
      const uint64_t f = fill[((size_t)(r / bps) * n_pieces + x) * bps + r % bps];
      entries += ((size_t)r * n_pieces + x) * cap;
      e0 = 0;
      e1 = (uint32_t)(f < cap ? f : cap);
    } else if (cap) {
      const uint64_t f = fill[(size_t)r * BB_CURSOR_STRIDE];
      entries += (size_t)r * cap;
      e0 = 0;
      e1 = (uint32_t)(f < cap ? f : cap);
    } else {
      e0 = region_base[r];
      e1 = region_base[r + 1];
    }
    if (e0 == e1) continue; // (uniform over the block)
    if (!zeroed) {
      for (uint32_t i = threadIdx.x; i < BB_REGION_DWORDS / 4u; i += BB_APPLY_THREADS) l4[i] = make_uint4(0, 0, 0, 0);
      __syncthreads();
      zeroed = true;
    }
    // entries: vectors of 4 where aligned, singles at the two ends
    const uint32_t up = (e0 + 3u) & ~3u;
    const uint32_t a0 = up < e1 ? up : e1, a1 = e1 & ~3u;
    auto put = [&](uint32_t e) { atomicOr(&bb_lds[e >> 5], 1u << (e & 31u)); };
    if (threadIdx.x < a0 - e0) put(entries[e0 + threadIdx.x]);
    if (a1 > a0) {
      const bb_v4u* v = (const bb_v4u*)(entries + a0);
      const uint32_t nv = (a1 - a0) >> 2;
      // BB_APPLY_BATCH loads in flight per thread (one block per CU and nobody to wait behind: one load at a time was a
      // round trip to HBM per 16 bytes -- 18 in a row per region)
      for (uint32_t i0 = threadIdx.x; i0 < nv; i0 += BB_APPLY_BATCH * BB_APPLY_THREADS) {
        bb_v4u q[BB_APPLY_BATCH];
#pragma unroll
        for (uint32_t u = 0; u < BB_APPLY_BATCH; ++u) {
          const uint32_t i = i0 + u * BB_APPLY_THREADS;
          q[u] = __builtin_nontemporal_load(v + (i < nv ? i : i0));
        }
#pragma unroll
        for (uint32_t u = 0; u < BB_APPLY_BATCH; ++u)
          if (i0 + u * BB_APPLY_THREADS < nv) {
            put(q[u].x);
            put(q[u].y);
            put(q[u].z);
            put(q[u].w);
          }
      }
    }
    if (a1 >= a0 && threadIdx.x < e1 - a1) put(entries[a1 + threadIdx.x]);
   }
   if (!zeroed) continue; // (nothing for this region: uniform)
    __syncthreads();
    // the filter lines that got a bit
    const uint64_t d0 = (uint64_t)r * BB_REGION_DWORDS;
    const uint64_t left = filter_dwords - d0;
    const uint32_t here = left < BB_REGION_DWORDS ? (uint32_t)left : BB_REGION_DWORDS;
    uint4* const f4 = (uint4*)(filter + d0); // (regions start on 128 KiB of a 16-byte aligned filter -- see the host side)
    for (uint32_t i0 = threadIdx.x; i0 < here / 4u; i0 += 4u * BB_APPLY_THREADS) { // (four read-modify-writes in flight)
      uint4 v[4], g[4];
      bool every = true;
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) {
        const uint32_t i = i0 + u * BB_APPLY_THREADS;
        v[u] = i < here / 4u ? l4[i] : make_uint4(0, 0, 0, 0);
        every = every && (v[u].x | v[u].y | v[u].z | v[u].w) != 0u;
      }
      if (__ballot(!every) == 0ull) { // every vector of the wave's four rows got a bit (a batch that is dense in the filter):
#pragma unroll                      // no branch between the loads and the stores, so no wait between the stores either
        for (uint32_t u = 0; u < 4; ++u) g[u] = f4[i0 + u * BB_APPLY_THREADS];
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
          g[u].x |= v[u].x;
          g[u].y |= v[u].y;
          g[u].z |= v[u].z;
          g[u].w |= v[u].w;
          f4[i0 + u * BB_APPLY_THREADS] = g[u];
        }
        continue;
      }
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u)
        if (v[u].x | v[u].y | v[u].z | v[u].w) g[u] = f4[i0 + u * BB_APPLY_THREADS];
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u)
        if (v[u].x | v[u].y | v[u].z | v[u].w) {
          g[u].x |= v[u].x;
          g[u].y |= v[u].y;
          g[u].z |= v[u].z;
          g[u].w |= v[u].w;
          f4[i0 + u * BB_APPLY_THREADS] = g[u];
        }
    }
    for (uint32_t i = (here & ~3u) + threadIdx.x; i < here; i += BB_APPLY_THREADS) // (a filter that does not end on 16 bytes)
      if (bb_lds[i]) filter[d0 + i] |= bb_lds[i];
    __syncthreads();
  }
}

// ---- the counting sketch's apply: a workgroup per region of 2^15 one-byte counters -------------------------------------
// (count-min sketch: every value adds one to counter `h mod n_counters`, saturating at 255.)  dynamic LDS: 128 KiB =
// 32-bit tallies of the region's counters; entries = 15-bit offsets.  Then every 4 counters that got something are read
// (one dword of the sketch), added to byte by byte with saturation, and written back.
constexpr uint32_t CS_REGION_SHIFT = 15;
constexpr uint32_t CS_BIN_SHIFT = CS_REGION_SHIFT + 7; // 128 regions per bin, as the filter's
__device__ __forceinline__ uint32_t sat_add_bytes(uint32_t word, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3)
{
  const uint32_t b0 = (word & 0xFFu) + a0, b1 = ((word >> 8) & 0xFFu) + a1, b2 = ((word >> 16) & 0xFFu) + a2, b3 = (word >> 24) + a3;
  // (a tally can be anything up to 2^31: compare before the sum can wrap)
  const uint32_t c0 = a0 > 255u || b0 > 255u ? 255u : b0, c1 = a1 > 255u || b1 > 255u ? 255u : b1;
  const uint32_t c2 = a2 > 255u || b2 > 255u ? 255u : b2, c3 = a3 > 255u || b3 > 255u ? 255u : b3;
  return c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
}
static __global__ __launch_bounds__(BB_APPLY_THREADS) void count_apply_kernel(const uint32_t* __restrict__ all_entries,
                                                                              const uint32_t* __restrict__ region_base,
                                                                              uint32_t n_regions, uint32_t* __restrict__ sketch,
                                                                              uint64_t sketch_dwords, uint64_t cap,
                                                                              const uint32_t* __restrict__ fill,
                                                                              const BloomStatus* status, uint64_t ovf_cap,
                                                                              uint32_t n_pieces = 0, uint32_t bps = 0)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t bb_lds[];
  uint4* const l4 = (uint4*)bb_lds;
  constexpr uint32_t SLOTS = 1u << CS_REGION_SHIFT; // counters per region = LDS tallies
  if (bloom_round_failed(status, ovf_cap)) return;
  for (uint32_t r = blockIdx.x; r < n_regions; r += gridDim.x) {
   const uint32_t n_runs = n_pieces ? n_pieces : 1u; // (pieces mode: as bloom_apply_kernel)
   bool zeroed = false;
   for (uint32_t x = 0; x < n_runs; ++x) {
    uint32_t e0, e1;
    const uint32_t* entries = all_entries;
    if (n_pieces) {
      const uint64_t f = fill[((size_t)(r / bps) * n_pieces + x) * bps + r % bps];
      entries += ((size_t)r * n_pieces + x) * cap;
      e0 = 0;
      e1 = (uint32_t)(f < cap ? f : cap);
    } else if (cap) {
      const uint64_t f = fill[(size_t)r * BB_CURSOR_STRIDE];
      entries += (size_t)r * cap;
      e0 = 0;
      e1 = (uint32_t)(f < cap ? f : cap);
    } else {
      e0 = region_base[r];
      e1 = region_base[r + 1];
    }
    if (e0 == e1) continue; // (uniform over the block)
    if (!zeroed) {
      for (uint32_t i = threadIdx.x; i < SLOTS / 4u; i += BB_APPLY_THREADS) l4[i] = make_uint4(0, 0, 0, 0);
      __syncthreads();
      zeroed = true;
    }
    const uint32_t up = (e0 + 3u) & ~3u;
    const uint32_t a0 = up < e1 ? up : e1, a1 = e1 & ~3u;
    auto put = [&](uint32_t e) { atomicAdd(&bb_lds[e], 1u); };
    if (threadIdx.x < a0 - e0) put(entries[e0 + threadIdx.x]);
    if (a1 > a0) {
      const bb_v4u* v = (const bb_v4u*)(entries + a0);
      const uint32_t nv = (a1 - a0) >> 2;
      // BB_APPLY_BATCH loads in flight per thread (one block per CU and nobody to wait behind: one load at a time was a
      // round trip to HBM per 16 bytes -- 18 in a row per region)
      for (uint32_t i0 = threadIdx.x; i0 < nv; i0 += BB_APPLY_BATCH * BB_APPLY_THREADS) {
        bb_v4u q[BB_APPLY_BATCH];
#pragma unroll
        for (uint32_t u = 0; u < BB_APPLY_BATCH; ++u) {
          const uint32_t i = i0 + u * BB_APPLY_THREADS;
          q[u] = __builtin_nontemporal_load(v + (i < nv ? i : i0));
        }
#pragma unroll
        for (uint32_t u = 0; u < BB_APPLY_BATCH; ++u)
          if (i0 + u * BB_APPLY_THREADS < nv) {
            put(q[u].x);
            put(q[u].y);
            put(q[u].z);
            put(q[u].w);
          }
      }
    }
    if (a1 >= a0 && threadIdx.x < e1 - a1) put(entries[a1 + threadIdx.x]);
   }
   if (!zeroed) continue; // (nothing for this region: uniform)
    __syncthreads();
    const uint64_t d0 = (uint64_t)r * (SLOTS / 4u); // the region's first dword of the sketch
    const uint64_t left = sketch_dwords - d0;
    const uint32_t here = left < SLOTS / 4u ? (uint32_t)left : SLOTS / 4u;
    for (uint32_t i0 = threadIdx.x; i0 < here; i0 += 4u * BB_APPLY_THREADS) { // (four read-modify-writes in flight)
      uint4 t[4];
      uint32_t w[4];
      bool every = true;
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u) {
        const uint32_t i = i0 + u * BB_APPLY_THREADS;
        t[u] = i < here ? l4[i] : make_uint4(0, 0, 0, 0);
        every = every && (t[u].x | t[u].y | t[u].z | t[u].w) != 0u;
      }
      if (__ballot(!every) == 0ull) { // (every dword of the wave's four rows got a count: no branch, no wait between the stores)
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) w[u] = sketch[d0 + i0 + u * BB_APPLY_THREADS];
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) sketch[d0 + i0 + u * BB_APPLY_THREADS] = sat_add_bytes(w[u], t[u].x, t[u].y, t[u].z, t[u].w);
        continue;
      }
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u)
        if (t[u].x | t[u].y | t[u].z | t[u].w) w[u] = sketch[d0 + i0 + u * BB_APPLY_THREADS];
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u)
        if (t[u].x | t[u].y | t[u].z | t[u].w) sketch[d0 + i0 + u * BB_APPLY_THREADS] = sat_add_bytes(w[u], t[u].x, t[u].y, t[u].z, t[u].w);
    }
    __syncthreads();
  }
}

// slots mode: the values that did not fit their bucket, as full positions.  Test, then touch: a heavy hitter's bit is set
// (its counter saturated) after the first few, the rest cost a load.
template <bool COUNTERS>
static __global__ __launch_bounds__(256) void bloom_overflow_apply_kernel(const uint64_t* __restrict__ ovf, const BloomStatus* status,
                                                                          uint64_t ovf_cap, uint32_t* __restrict__ table)
{
  if (bloom_round_failed(status, ovf_cap)) return;
  const uint64_t n = __builtin_nontemporal_load(&status->ovf_n);
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t p = ovf[i];
    if constexpr (COUNTERS) {
      uint32_t* const w = table + (p >> 2);
      const uint32_t sh = ((uint32_t)p & 3u) * 8u;
      uint32_t old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (((old >> sh) & 0xFFu) != 0xFFu) {
        const uint32_t seen = atomicCAS(w, old, old + (1u << sh));
        if (seen == old) break;
        old = seen;
      }
    } else {
      uint32_t* const w = table + (p >> 5);
      const uint32_t bit = 1u << ((uint32_t)p & 31u);
      if (!(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) atomicOr(w, bit);
    }
  }
}

// small batches / sketches beyond the lists' reach: one compare-and-swap loop per value
static __global__ __launch_bounds__(256) void count_atomic_kernel(const uint64_t* __restrict__ hashes, uint64_t n,
                                                                  uint32_t* __restrict__ sketch, uint64_t n_counters, uint64_t magic)
{
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t p = mod_invariant(hashes[i], n_counters, magic);
    uint32_t* const w = sketch + (p >> 2);
    const uint32_t sh = ((uint32_t)p & 3u) * 8u;
    uint32_t old = *w;
    while (((old >> sh) & 0xFFu) != 0xFFu) {
      const uint32_t seen = atomicCAS(w, old, old + (1u << sh));
      if (seen == old) break;
      old = seen;
    }
  }
}

// estimate of a k-mer: the smallest of its m counters (values: m per k-mer, as nthip_kmer_hash writes them)
static __global__ __launch_bounds__(256) void count_query_kernel(const uint64_t* __restrict__ hashes, uint64_t n_kmers, uint32_t m,
                                                                 const uint8_t* __restrict__ sketch, uint64_t n_counters,
                                                                 uint64_t magic, uint8_t* __restrict__ out)
{
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_kmers; i += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t lo = 255u;
    for (uint32_t j = 0; j < m; ++j) {
      const uint32_t v = sketch[mod_invariant(hashes[i * m + j], n_counters, magic)];
      lo = v < lo ? v : lo;
    }
    out[i] = (uint8_t)lo;
  }
}


// Bloom membership of every k-mer of a hash stream (m values per k-mer, as nthip_kmer_hash writes them): out[i] = 1 when all
// m bits of k-mer i are set, else 0; *found += the number of ones
static __global__ __launch_bounds__(256) void stream_bloom_flags_kernel(const uint64_t* __restrict__ hashes, uint64_t n_kmers, uint32_t m,
                                                                        const uint32_t* __restrict__ bloom, uint64_t n_bits, uint64_t magic,
                                                                        uint8_t* __restrict__ out, unsigned long long* __restrict__ found)
{
  uint32_t mine = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_kmers; i += (uint64_t)gridDim.x * blockDim.x) {
    bool hit = true;
    for (uint32_t j = 0; j < m; ++j) {
      const uint64_t p = mod_invariant(hashes[i * m + j], n_bits, magic);
      hit = hit && ((bloom[p >> 5] >> ((uint32_t)p & 31u)) & 1u);
    }
    out[i] = hit ? 1 : 0;
    mine += hit ? 1u : 0u;
  }
  for (int d = 32; d > 0; d >>= 1) mine += (uint32_t)__shfl_xor((int)mine, d, 64);
  if ((threadIdx.x & 63u) == 0 && mine) atomicAdd(found, (unsigned long long)mine);
}

// Bloom membership of the k-mers of a hash stream, per READ (reads of any lengths: the stream's read r is k-mers
// roff[r] ... roff[r + 1], m values each): hits[r] = number of its k-mers whose m bits are all set.  One wave per read
// at a time; the loads of the filter are what it waits for (four k-mers per lane in flight).
static __global__ __launch_bounds__(256) void stream_bloom_query_kernel(const uint64_t* __restrict__ hashes, const uint64_t* __restrict__ roff,
                                                                        uint64_t n_reads, uint64_t n_kmers, uint32_t m,
                                                                        const uint32_t* __restrict__ bloom, uint64_t n_bits, uint64_t magic,
                                                                        uint64_t* __restrict__ hits, unsigned long long* __restrict__ total_hits)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  uint64_t mine = 0;
  for (uint64_t r = wave; r < n_reads; r += n_waves) {
    const uint64_t i0 = roff[r], i1 = r + 1 < n_reads ? roff[r + 1] : n_kmers;
    uint32_t found = 0;
    constexpr uint32_t U = 4;
    for (uint64_t c0 = i0; c0 < i1; c0 += 64u * U) {
      bool hit[U];
#pragma unroll
      for (uint32_t u = 0; u < U; ++u) hit[u] = c0 + u * 64u + lane < i1;
      for (uint32_t j = 0; j < m; ++j) {
        uint64_t p[U];
        uint32_t word[U];
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
          const uint64_t i = c0 + u * 64u + lane;
          p[u] = mod_invariant(hashes[(i < i1 ? i : i0) * m + j], n_bits, magic);
          word[u] = bloom[p[u] >> 5];
        }
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) hit[u] = hit[u] && ((word[u] >> ((uint32_t)p[u] & 31u)) & 1u);
      }
#pragma unroll
      for (uint32_t u = 0; u < U; ++u) found += (uint32_t)__builtin_popcountll(__ballot(hit[u]));
    }
    if (lane == 0) {
      if (hits) hits[r] = found;
      mine += found;
    }
  }
  if (lane == 0 && mine) atomicAdd(total_hits, (unsigned long long)mine);
}

} // namespace ntamd
