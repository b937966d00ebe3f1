// bloom_fused_kernels.hpp -- the binned Bloom / counting-sketch insert WITHOUT a hash stream (round 4).
//
// Until round 3 a round of reads was hashed to a stream (8 B per value written), the stream read by the histogram (8 B)
// and read again by the first partition level (8 B): 24 of the insert's 40 bytes per value, and the three kernels that
// move them -- hash 4.7, hist 3.9, part 8.3 ms per 2.4 G values -- more than half of its time.  Hashing is cheaper than
// that traffic, so here the reads are hashed TWICE and the values never leave the CU:
//   pass COUNT  every value's region counted in LDS (32-bit counters for every region of the filter, flushed once per
//               block -- what bloom_hist_kernel did from the stream);
//   pass PART   every value goes straight from the registers into the first partition level: a tile is the 16 windows a
//               thread has just rolled x the block's reads, sorted by bucket in LDS and appended run by run at the
//               buckets' cursors -- bloom_part_kernel<IN64>'s tile, fed by the hash instead of by a load.
// Slots mode (bloom_binned_kernels.hpp) needs no counts: pass PART alone, the reads hashed ONCE.
// The second level and the apply kernels are bloom_binned_kernels.hpp's, unchanged: 16 B per value instead of 40.
//
// The counters (up to 128 KiB for a filter of 2^35 bits) and the sort buffer leave no room for first-window tables, so
// the hashing is the table-free form of kmer_fixed_kernel: one read per thread, its first window reached by k rolls from
// a window of virtual 'A's (f_init / r_init), every step one 16-entry pair-table lookup (next_forward_hash /
// next_reverse_hash, src/kmer.cpp:84-94,164-174; the first window is what base_forward_hash / base_reverse_hash give,
// src/kmer.cpp:43-73,123-152, reached by the roll).  len / (len - k + 1) rolls per k-mer: 1.25 for 150 bp reads.
// Non-bases (round 4, later): the staging loop marks the READS that hold one (a bit per read of the tile); the thread of
// such a read -- one in a thousand on real data -- fetches its own bytes again, keeps the number of bases since the last
// non-base and emits a window only when that reaches k (NtHash::roll, src/kmer.cpp:228-264: windows with a non-base are
// skipped); the garbage code of a non-base enters and leaves the rolled state identically, so every emitted value is
// exact.  The windows not emitted are counted in *a.lost (pass PART).
#pragma once

#include <hip/hip_runtime.h>

#include "bloom_binned_kernels.hpp"
#include "kmer_kernels.hpp"

namespace ntamd {

enum : int { BF_COUNT = 0, BF_PART = 1 };
#ifndef BF_TIMING
#define BF_TIMING 0 // 1: one block prints the phase times of its first tiles (tools/ab_build.sh)
#endif
#ifndef BF_ABL
#define BF_ABL 0 // ablations of pass PART (WRONG results; tools/ab_build.sh): 1 hashing only, 2 + ranking, 3 + scan and sort (no copy-out)
#endif

struct BloomFusedArgs {
  const uint8_t* seqs;      // fixed-length reads, stride bytes apart
  unsigned long long* lost; // pass PART: += windows that hold a non-base (not emitted)
  uint64_t n_reads;
  uint32_t len, stride, k, m;
  uint32_t pad_dwords;      // front pad of the LDS bit stream: ceil(k / 16) + 1 words of virtual 'A'
  uint32_t n_tiles;         // ceil(n_reads / threads)
  uint64_t f_init, r_init;  // strand hashes of k virtual 'A's
  uint64_t tab[16][2];      // [(in << 2) | out] -> {forward term, reverse term}
  uint64_t mult[KF_MAX_RUNTIME_M];
  uint64_t n_bits, magic;   // value -> slot: h mod n_bits (filter bits / sketch counters)
  // pass COUNT
  uint32_t* counts;         // [n_regions] += values per region
  uint32_t n_regions, region_shift;
  // pass PART: the bucket of a slot is slot >> shift, what is written is slot & mask, at the bucket's cursor
  uint32_t* out;
  uint32_t* cursor;
  uint32_t shift, mask, n_buckets;
  BloomSlots sl;            // slots mode (sl.cap != 0; bloom_binned_kernels.hpp): bucket b owns out[b * cap ...), cursors count from 0
};
// the binned QUERY (bloom_query_kernels.hpp; QUERY instantiation of pass PART, slots mode): the way back of every tile.
// Tile row ts = ((tile * q_steps) + (word - first emitting word)) * m + jj;  q_where[(ts * 16 + i) * THREADS + thread] =
// the place of that window's value in the tile's sorted order (0xFFFF: nothing emitted), for the steps i that can emit;
// q_tab[ts * n_buckets + b] = {entries, place of the run in bucket b's slots}; q_tovf[...]: see bloom_copy_out
// (a struct of its own: the insert's instantiations keep the argument block they had)
struct BloomFusedQueryArgs : BloomFusedArgs {
  uint16_t* q_where;
  uint2* q_tab;
  uint32_t* q_tovf;
  uint32_t q_steps;
  // a pass over SOME of a k-mer's hashes (the query of m > 1 asks the later ones only for the k-mers whose earlier ones hit):
  // the pass's values are hashes()[q_jj0 ... q_jj0 + m) (m: the pass's count); q_surv (may be NULL: every window) holds, per
  // tile, emitting word and thread, the 16 windows of the word that are still in the race
  uint32_t q_jj0;
  const uint16_t* q_surv;
};

// pieces mode (bloom_binned_kernels.hpp, bloom_copy_out_lines): bucket b's entries of block x go to piece (b * gridDim.x + x)
// of `out` (sl.cap entries each), whole lines only; p_fill[x * n_buckets + b] = how many it got.  No cursors.
struct BloomFusedPiecesArgs : BloomFusedQueryArgs {
  uint32_t* p_fill;
};
template <bool QUERY, bool PIECES>
struct BloomFusedArgsOf {
  typedef typename std::conditional<PIECES, BloomFusedPiecesArgs, typename std::conditional<QUERY, BloomFusedQueryArgs, BloomFusedArgs>::type>::type type;
};

// THREADS reads per tile; dynamic LDS: bit stream | COUNT: n_regions counters / PART: THREADS * 16 sorted slots
// (PIECES: + 256 x 32 entries that wait for their line, + 2 x 256 counters)
template <int PASS, uint32_t THREADS, bool QUERY = false, bool PIECES = false>
__global__ __launch_bounds__(THREADS) void bloom_fused_kernel(const typename BloomFusedArgsOf<QUERY, PIECES>::type a)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  __shared__ __attribute__((aligned(16))) uint4 tab[16];
  __shared__ uint32_t hist[BB_MAX_BINS], off[BB_MAX_BINS], gbase[BB_MAX_BINS];
  __shared__ uint32_t rflag[THREADS / 32u]; // reads of the tile that hold a non-base
  const uint32_t k = a.k, m = a.m;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  // the bit stream of a tile: pad + slab (THREADS reads) + two spare words
  const uint32_t bits_dwords = a.pad_dwords + (((THREADS - 1u) * a.stride + a.len + 15u + 15u) >> 4) + 2u;
  uint32_t* const bits = lds_dyn;
  uint32_t* const area = lds_dyn + ((bits_dwords + 3u) & ~3u); // COUNT: the counters; PART: the sorted tile
  uint32_t* const lwait = area + THREADS * 16u;                 // PIECES: [bucket][32] entries waiting for their line,
  uint32_t* const lcnt = lwait + BB_MAX_BINS * 32u;              //   how many of them,
  uint32_t* const pcur = lcnt + BB_MAX_BINS;                    //   entries of the block's piece of every bucket so far
  if constexpr (PIECES) {
    if (tid < BB_MAX_BINS) {
      lcnt[tid] = 0;
      pcur[tid] = 0;
    }
  }

  if (tid < 16)
    tab[tid] = make_uint4((uint32_t)a.tab[tid][0], (uint32_t)(a.tab[tid][0] >> 32), (uint32_t)a.tab[tid][1],
                          (uint32_t)(a.tab[tid][1] >> 32));
  for (uint32_t i = tid; i < a.pad_dwords; i += THREADS) bits[i] = 0; // virtual 'A's (code 0)
  if constexpr (PASS == BF_COUNT)
    for (uint32_t i = tid; i < a.n_regions; i += THREADS) area[i] = 0;

  const uint32_t kmod = (k - 1u) & 15u;
  const uint32_t jb = (k - 1u) >> 4;          // the word that holds the first emitting step
  const uint32_t n_words = (a.len + 15u) >> 4;
  const uint32_t nwin = a.len - k + 1u;
  uint32_t lost = 0;

  for (uint32_t t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
    const uint64_t run0 = (uint64_t)t * THREADS;
    const uint64_t left = a.n_reads - run0;
    const uint32_t runs_here = left < THREADS ? (uint32_t)left : THREADS;
    // ---- the slab of ASCII as a 2-bit stream (16-byte vectors aligned in memory; the bytes of a first / last vector that
    // lie outside the slab are somebody else's: not judged) ----
    const uint64_t addr0 = (uint64_t)(a.seqs + run0 * a.stride);
    const uint32_t shift = (uint32_t)(addr0 & 15u);
    const uint4* vsrc = (const uint4*)(addr0 - shift);
    const uint32_t slab_bytes = (runs_here - 1u) * a.stride + a.len;
    const uint32_t n_vec = (shift + slab_bytes + 15u) >> 4;
    __syncthreads(); // the tile before is consumed; pad / tab / counters are there on the first pass
    if (tid < THREADS / 32u) rflag[tid] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n_vec; i += THREADS) {
      const uint4 v = vsrc[i];
      uint32_t b = 0;
      const uint32_t p = pack16(v, b);
      if (b) { // (rare) whose non-bases?  Bytes outside the slab and between the reads are nobody's
        uint32_t bx[4] = {0, 0, 0, 0};
        (void)pack4(v.x, bx[0]);
        (void)pack4(v.y, bx[1]);
        (void)pack4(v.z, bx[2]);
        (void)pack4(v.w, bx[3]);
        for (int q = 0; q < 16; ++q)
          if ((bx[q >> 2] >> ((q & 3) * 8)) & 0xFFu) {
            const int32_t at = (int32_t)(i << 4) + q - (int32_t)shift;
            if (at >= 0 && at < (int32_t)slab_bytes) {
              const uint32_t r = (uint32_t)at / a.stride;
              if ((uint32_t)at - r * a.stride < a.len) atomicOr(&rflag[r >> 5], 1u << (r & 31u));
            }
          }
      }
      bits[a.pad_dwords + i] = p;
    }
    if (tid < 2) bits[a.pad_dwords + n_vec + tid] = 0; // funnels read one word ahead
    __syncthreads();

    // ---- every thread rolls its read, 16 steps at a time ----
    const bool live = tid < runs_here;
    const uint32_t lrun = live ? tid : 0u; // (idle threads of the last tile redo its first read and emit nothing)
    const uint32_t bl = a.pad_dwords * 16u + shift + lrun * a.stride;
    const uint32_t in_d = bl >> 4, in_sh = (bl & 15u) << 1;
    const uint32_t ob = bl - k; // (never negative thanks to the pad)
    const uint32_t out_d = ob >> 4, out_sh = (ob & 15u) << 1;
    uint32_t f_lo = (uint32_t)a.f_init, f_hi = (uint32_t)(a.f_init >> 32);
    uint32_t r_lo = (uint32_t)a.r_init, r_hi = (uint32_t)(a.r_init >> 32);
    uint32_t in_lo = bits[in_d], out_lo = bits[out_d];
    const bool dirty_me = live && ((rflag[lrun >> 5] >> (lrun & 31u)) & 1u);
    const uint8_t* const my_read = a.seqs + (run0 + lrun) * a.stride;
    uint32_t since = 0;                   // bases since the read's last non-base (dirty_me only)
    if (dirty_me) lost += nwin;           // (what it does emit is taken off below)

    for (uint32_t j = 0; j < n_words; ++j) {
#if BF_TIMING
      const uint64_t tk0 = __builtin_amdgcn_s_memrealtime();
#endif
      const uint32_t in_hi = bits[in_d + j + 1], out_hi = bits[out_d + j + 1];
      const uint32_t w_in = funnel(in_hi, in_lo, in_sh);
      uint32_t w_out = funnel(out_hi, out_lo, out_sh);
      in_lo = in_hi;
      out_lo = out_hi;
      const uint32_t s0 = j << 4;
      // steps whose outgoing base lies before the read's start see a virtual 'A'
      if (s0 + 16u <= k) w_out = 0;
      else if (s0 < k) w_out &= ~0u << ((k - s0) << 1);
      const uint32_t u = ((w_in & 0x33333333u) << 2) | (w_out & 0x33333333u);
      const uint32_t v = (w_in & 0xCCCCCCCCu) | ((w_out >> 2) & 0x33333333u);
      uint4 terms[16];
#pragma unroll
      for (uint32_t i = 0; i < 16; ++i) { // (the table terms do not depend on the hash state: all in flight at once)
        const uint32_t src = (i & 1u) ? v : u;
        terms[i] = *(const uint4*)((const char*)tab + (((src >> ((i >> 1) * 4u)) & 0xFu) << 4));
      }
      // steps [lo, hi) of this word end a window of the read (uniform over the block)
      const uint32_t lo = j < jb ? 16u : (j == jb ? kmod : 0u);
      const uint32_t hi = a.len - s0 < 16u ? a.len - s0 : 16u;
      uint64_t h[16];
#pragma unroll
      for (uint32_t i = 0; i < 16; ++i) {
        roll_step(f_lo, f_hi, r_lo, r_hi, terms[i]);
        h[i] = canon_pair(f_lo, f_hi, r_lo, r_hi);
      }
      // the steps of this word that emit: all of [lo, hi) for a read of bases only
      uint32_t emask = live ? ((0xFFFFu >> (16u - hi)) & (0xFFFFu << lo)) & 0xFFFFu : 0u;
      if (dirty_me) {
        uint32_t ok = 0;
        const uint32_t here = a.len - s0 < 16u ? a.len - s0 : 16u;
        for (uint32_t i = 0; i < here; ++i) {
          uint32_t bb = 0;
          (void)pack4((uint32_t)my_read[s0 + i] * 0x01010101u, bb);
          since = bb ? 0u : since + 1u;
          if (since >= k) ok |= 1u << i;
        }
        emask &= ok;
        if (lo < hi) lost -= (uint32_t)__builtin_popcount(emask);
      }
      if (lo >= hi) continue; // (no window ends in this word: uniform)
      uint32_t jj0 = 0;
      if constexpr (QUERY) {
        jj0 = a.q_jj0;
        if (a.q_surv) emask &= a.q_surv[((uint64_t)t * a.q_steps + (j - jb)) * THREADS + tid]; // (behind the `lost` count: that is pass 0's)
      }
      for (uint32_t jj = 0; jj < m; ++jj) { // the k-mers' m values (extend_hashes, src/internal.hpp:104-118), one set at a time
        if constexpr (PASS == BF_COUNT) {
#pragma unroll
          for (uint32_t i = 0; i < 16; ++i)
            if ((emask >> i) & 1u) {
              const uint64_t val = jj == 0 ? h[i] : mix_hash(h[i], a.mult[jj & (KF_MAX_RUNTIME_M - 1)]);
              atomicAdd(&area[(uint32_t)(mod_invariant(val, a.n_bits, a.magic) >> a.region_shift)], 1u);
            }
        } else {
          // ---- a tile of the first partition level: THREADS x 16 slots, sorted by bucket, appended at the cursors ----
#if BF_ABL == 1
#pragma unroll
          for (uint32_t i = 0; i < 16; ++i)
            if ((emask >> i) & 1u) lost += (uint32_t)h[i] == 0x12345u;
          continue;
#endif
          if (tid < BB_MAX_BINS) hist[tid] = 0;
          __syncthreads();
#if BF_TIMING
          const uint64_t tk1 = __builtin_amdgcn_s_memrealtime();
#endif
          uint32_t val[16], where[16]; // where = bucket << 16 | rank inside the tile's bucket
#pragma unroll
          for (uint32_t i = 0; i < 16; ++i) {
            where[i] = ~0u;
            val[i] = 0;
            if ((emask >> i) & 1u) {
              const uint64_t hv = jj + jj0 == 0 ? h[i] : mix_hash(h[i], a.mult[(jj + jj0) & (KF_MAX_RUNTIME_M - 1)]);
              const uint64_t p = mod_invariant(hv, a.n_bits, a.magic);
              const uint32_t b = (uint32_t)(p >> a.shift);
              val[i] = (uint32_t)p & a.mask;
              where[i] = (b << 16) | atomicAdd(&hist[b], 1u);
            }
          }
          uint64_t q_ts = 0;
          if constexpr (QUERY) q_ts = ((uint64_t)t * a.q_steps + (j - jb)) * m + jj;
          __syncthreads();
#if BF_TIMING
          const uint64_t tk2 = __builtin_amdgcn_s_memrealtime();
#endif
#if BF_ABL == 2
#pragma unroll
          for (uint32_t i = 0; i < 16; ++i) lost += where[i] == 0x12345u && val[i] == 77u;
          continue;
#endif
          uint32_t my_base = 0; // (the cursor's answer is wanted by the copy-out only: it travels while the tile is sorted)
          if (tid < a.n_buckets) {
            const uint32_t cnt = hist[tid];
            if constexpr (PIECES) {
              my_base = pcur[tid];
              pcur[tid] = my_base + cnt;
            } else {
              my_base = cnt ? atomicAdd(&a.cursor[(size_t)tid * BB_CURSOR_STRIDE], cnt) : 0u;
            }
            if constexpr (QUERY) a.q_tab[q_ts * a.n_buckets + tid] = make_uint2(cnt, my_base);
          }
          if (wave == 0) { // exclusive scan of the (at most 256) bucket counts: 4 per lane
            uint32_t cc[4], s = 0;
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
              cc[i] = hist[lane * 4u + i];
              s += cc[i];
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
              run += cc[i];
            }
          }
          __syncthreads();
#if BF_TIMING
          const uint64_t tk3 = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
          for (uint32_t i = 0; i < 16; ++i) {
            uint32_t slot = 0xFFFFu;
            if (where[i] != ~0u) {
              slot = off[where[i] >> 16] + (where[i] & 0xFFFFu);
              area[slot] = val[i];
            }
            if constexpr (QUERY) {
              if (i >= lo && i < hi) a.q_where[(q_ts * 16u + i) * THREADS + tid] = (uint16_t)slot;
            }
          }
          if (tid < a.n_buckets) gbase[tid] = my_base;
          __syncthreads();
#if BF_TIMING
          const uint64_t tk4 = __builtin_amdgcn_s_memrealtime();
#endif
#if BF_ABL == 3
          continue;
#endif
          if constexpr (PIECES)
            bloom_copy_out_lines<THREADS / 64u, QUERY>(area, hist, off, gbase, lwait, lcnt, a.n_buckets, wave, lane, a.out, (uint64_t)blockIdx.x,
                                                       (uint64_t)gridDim.x, 0ull, a.sl, a.shift, QUERY ? a.q_tovf + q_ts * a.n_buckets : nullptr);
          else if constexpr (QUERY)
            bloom_copy_out<THREADS / 64u, true>(area, hist, off, gbase, a.n_buckets, wave, lane, a.out, 0ull, a.sl, a.shift,
                                                a.q_tovf + q_ts * a.n_buckets);
          else bloom_copy_out<THREADS / 64u>(area, hist, off, gbase, a.n_buckets, wave, lane, a.out, 0ull, a.sl, a.shift);
          __syncthreads();
#if BF_TIMING
          if (tid == 0 && blockIdx.x == 3u && t < gridDim.x * 3u) { const uint64_t tk5 = __builtin_amdgcn_s_memrealtime(); printf("tile %u word %u: hash %u  rank %u  scan %u  sort %u  copy-out %u (10 ns)\n", t, j, (unsigned)(tk1 - tk0), (unsigned)(tk2 - tk1), (unsigned)(tk3 - tk2), (unsigned)(tk4 - tk3), (unsigned)(tk5 - tk4)); }
#endif
        }
      }
    }
  }
  if constexpr (PASS == BF_COUNT) {
    __syncthreads();
    for (uint32_t i = tid; i < a.n_regions; i += THREADS) {
      const uint32_t v = area[i];
      if (v) atomicAdd(&a.counts[i], v);
    }
  }
  if constexpr (PIECES) {
    __syncthreads();
    bloom_flush_lines<THREADS / 64u>(lwait, lcnt, pcur, a.n_buckets, wave, lane, tid, a.out, (uint64_t)blockIdx.x, (uint64_t)gridDim.x, a.sl.cap,
                                     a.p_fill + (size_t)blockIdx.x * a.n_buckets);
  }
  if constexpr (PASS == BF_PART) {
    if (__ballot(lost != 0) != 0) {
      for (int d = 32; d > 0; d >>= 1) lost += (uint32_t)__shfl_xor((int)lost, d, 64);
      if (lane == 0 && a.lost) atomicAdd(a.lost, (unsigned long long)lost);
    }
  }
}

} // namespace ntamd
