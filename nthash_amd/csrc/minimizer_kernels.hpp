// minimizer_kernels.hpp -- per-read (w, k)-minimizers of the canonical ntHash values (a consumer of the hash stream).
//
// What sketching tools built on ntHash keep of a read: of every w consecutive k-mers the one with the smallest hash
// (here: the canonical hashes()[0] of NtHash, reference src/kmer.cpp:246-264 for which windows are emitted; ties go to
// the leftmost).  Works on the emitted stream of a batch -- per read its k-mers in order with their window positions --
// so a k-mer that holds a non-base simply is not there: a window of w positions picks among the k-mers emitted inside
// it, and picks nothing when there is none.  Reads with fewer than w windows count as one window.
//
// k-mer j (position P, hash h) of a read with `nwin` window positions is picked by the window starting at s iff
//   s <= P <= s + w - 1,   0 <= s <= nwin - w,
//   no emitted k-mer at a position in [s, P) has a hash <= h   (an equal one to the left wins the tie),
//   no emitted k-mer at a position in (P, s + w) has a hash < h.
// With PL = position of the nearest k-mer on the left with hash <= h (none: -1) and PR = of the nearest on the right with
// hash < h (none: +inf), such an s exists iff  max(PL + 1, P - w + 1, 0) <= min(P, PR - w, nwin - w).
// That walk (at most w - 1 positions to either side) is what reads of more than MZ_LDS_POS window positions take.  A
// shorter read -- every short read -- is laid out by position in the wave's LDS (a k-mer that is not there: the largest
// value) and every window asks a sparse table for its leftmost minimum: M_j[i] = argmin of [i, i + 2^j) by doubling
// (log2 w uniform passes over the read, no lane waits for another's walk), window [s, s + w) = the better of
// M_J[s] and M_J[s + w - 2^J].  The picked positions are bits of a per-read mask (position-indexed).
#pragma once

#include <hip/hip_runtime.h>

#include "nt_math.hpp"

#include <cstdint>
#include <type_traits>

namespace ntamd {

struct MinimizerArgs {
  const uint64_t* hashes; // the batch's emitted k-mers, read by read (one value per k-mer)
  const uint32_t* pos;    // their window positions inside the read; NULL: every read emits every window (position = index)
  const uint64_t* roff;   // [n_reads]: first k-mer of read r in the stream (exclusive scan of the counts); NULL with pos == NULL
  uint64_t n_reads, n_kmers;
  uint32_t nwin;          // window positions of a read (fixed-length reads: len - k + 1; with offsets: of the longest read)
  uint32_t w;
  const uint64_t* offsets; // reads of any lengths (read r = [offsets[r], offsets[r + 1])): the window positions come from here; else NULL
  const uint64_t* ends;    // with offsets: read r = [offsets[r], ends[r]) (spans of a raw buffer); NULL: offsets[r + 1]
  uint32_t k, pad1;
  uint64_t* masks;        // [n_reads * chunks] bit l of word (r, c): the k-mer at window position 64 c + l of read r is a minimizer
  uint32_t chunks;        // ceil(nwin / 64)
  // reads given by offsets (not spans): read r's mask words start at ((offsets[r] - offsets[0]) >> 6) + r instead of r * chunks -- the
  // reads lie apart in the buffer, so the rows do too, and the masks take total_bytes / 8 + 8 n bytes whatever the
  // longest read is (100 k contigs with one of 10 Mbp asked for 125 GB of r * chunks rows)
  uint32_t mask_by_start;
  uint64_t* picked;       // [n_reads] minimizers of the read
  // second pass
  const uint64_t* out_off; // [n_reads] exclusive scan of picked
  uint64_t base, capacity;
  uint64_t* out_hashes;
  uint32_t* out_pos;       // may be NULL
  uint64_t* out_offsets;   // [n_reads] = base + out_off
};

// pass 1 -- one wave per read at a time: the mask of picked positions, their number.  masks must be zero on entry for the
// reads that take the walk (the host clears the array).
// (POSN: window positions a wave's LDS holds -- 256 for short reads: 2.6 KiB per wave, the CU's 32 wave slots fill and
// the chain of LDS round trips a read is hides behind the other waves; 1024: 10 KiB per wave, 12 waves per CU)
constexpr uint32_t MZ_LDS_POS = 1024;
constexpr uint32_t MZ_WAVES = 4; // per block
template <bool DENSE, uint32_t POSN = MZ_LDS_POS>
static __global__ __launch_bounds__(64 * MZ_WAVES) void minimizer_flag_kernel(const MinimizerArgs a)
{
  static_assert(POSN >= 128 && POSN <= MZ_LDS_POS, "a wave's window positions");
  __shared__ uint64_t lds_h[MZ_WAVES][POSN];
  __shared__ uint16_t lds_m[MZ_WAVES][POSN];
  __shared__ uint32_t lds_k[MZ_WAVES][POSN / 32];
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  uint64_t* const A = lds_h[wv];
  uint16_t* const M = lds_m[wv];
  uint32_t* const K = lds_k[wv];
  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  };
  // (a wave is alone with its latencies: the first 128 hashes of its NEXT read travel in registers while this one is done)
  uint64_t pf0 = 0, pf1 = 0;
  auto prefetch = [&](uint64_t rr) {
    if (DENSE && rr < a.n_reads) {
      const uint64_t b = rr * a.nwin;
      pf0 = lane < a.nwin ? a.hashes[b + lane] : 0;
      pf1 = lane + 64u < a.nwin ? a.hashes[b + 64u + lane] : 0;
    }
  };
  prefetch(wave);
  for (uint64_t r = wave; r < a.n_reads; r += n_waves) {
    uint32_t nwin = a.nwin;
    if (!DENSE && a.offsets != nullptr) { // reads of any lengths
      const uint64_t l = (a.ends != nullptr ? a.ends[r] : a.offsets[r + 1]) - a.offsets[r];
      nwin = l >= a.k ? (l - a.k + 1u < 0xFFFFFFFFull ? (uint32_t)(l - a.k + 1u) : 0xFFFFFFFFu) : 0u;
    }
    if (nwin == 0u) { // (shorter than k: no k-mer, nothing picked)
      if (lane == 0) a.picked[r] = 0;
      continue;
    }
    const uint32_t w = a.w < nwin ? a.w : nwin; // (a read with fewer windows than w: one window)
    const uint32_t n_starts = nwin - w + 1;
    uint32_t J = 0;
    while ((2u << J) <= w) ++J; // 2^J <= w < 2^(J+1)
    const uint32_t chunks_r = (nwin + 63u) >> 6; // mask words of this read (<= a.chunks)
    const uint64_t i0 = DENSE ? r * nwin : a.roff[r];
    const uint64_t i1 = DENSE ? i0 + nwin : (r + 1 < a.n_reads ? a.roff[r + 1] : a.n_kmers);
    uint64_t* const mrow = a.masks + (a.mask_by_start ? ((a.offsets[r] - a.offsets[0]) >> 6) + r : r * a.chunks);
    uint32_t n_picked = 0;
    if (nwin <= POSN) {
      // ---- the read by position ----
      if (DENSE) {
        if (lane < nwin) A[lane] = pf0;
        if (lane + 64u < nwin) A[lane + 64u] = pf1;
        for (uint32_t p = 128u + lane; p < nwin; p += 64u) A[p] = a.hashes[i0 + p];
        prefetch(r + n_waves);
      }
      for (uint32_t p = lane; p < nwin; p += 64u) {
        if (!DENSE) A[p] = ~0ull;
        M[p] = (uint16_t)p;
      }
      if (lane < 2u * chunks_r) K[lane] = 0; // (chunks_r <= POSN / 64 <= 16)
      if (!DENSE) {
        wave_sync();
        for (uint64_t i = i0 + lane; i < i1; i += 64u) A[a.pos[i]] = a.hashes[i];
      }
      wave_sync();
      // ---- A[i], M[i] := the smallest value of [i, i + 2^j) and the leftmost place it stands at; in place, ascending (a
      // step reads only what has not been written yet), 128 positions per step: one LDS round trip of 8 independent reads ----
      for (uint32_t j = 0; j < J; ++j) {
        const uint32_t off = 1u << j;
        for (uint32_t p0 = 0; p0 < nwin; p0 += 128u) {
          uint64_t hv[2];
          uint32_t mv[2];
#pragma unroll
          for (uint32_t u = 0; u < 2; ++u) {
            const uint32_t p = p0 + u * 64u + lane;
            const uint32_t pc = p < nwin ? p : nwin - 1u;
            const uint32_t q = pc + off < nwin ? pc + off : nwin - 1u;
            const uint64_t hp = A[pc], hq = A[q];
            const uint32_t mp = M[pc], mq = M[q];
            hv[u] = hq < hp ? hq : hp; // (an equal value on the left stays)
            mv[u] = hq < hp ? mq : mp;
          }
          wave_sync();
#pragma unroll
          for (uint32_t u = 0; u < 2; ++u) {
            const uint32_t p = p0 + u * 64u + lane;
            if (p < nwin) {
              A[p] = hv[u];
              M[p] = (uint16_t)mv[u];
            }
          }
          wave_sync();
        }
      }
      // ---- every window picks ----
      const uint32_t tail = w - (1u << J);
      for (uint32_t s0 = 0; s0 < n_starts; s0 += 64u) {
        const uint32_t st = s0 + lane;
        if (st < n_starts) {
          const uint64_t hx = A[st], hy = A[st + tail];
          const uint32_t x = M[st], y = M[st + tail];
          const uint32_t best = hy < hx ? y : hx < hy ? x : (x < y ? x : y);
          if ((hy < hx ? hy : hx) != ~0ull) atomicOr(&K[best >> 5], 1u << (best & 31u)); // (a window without a k-mer: nothing)
        }
      }
      wave_sync();
      if (lane < chunks_r) {
        const uint64_t m = (uint64_t)K[2u * lane] | ((uint64_t)K[2u * lane + 1u] << 32);
        mrow[lane] = m;
        n_picked = (uint32_t)__builtin_popcountll(m);
      }
      for (int d = 32; d > 0; d >>= 1) n_picked += (uint32_t)__shfl_xor((int)n_picked, d, 64);
      wave_sync(); // (the next read's staging waits for this one's reads)
    } else {
      // ---- a long read: the walks, in global memory ----
      const int64_t wl = w, last_s = (int64_t)nwin - wl;
      for (uint64_t c0 = i0; c0 < i1; c0 += 64u) {
        const uint64_t i = c0 + lane;
        bool pick = false;
        int64_t P = 0;
        if (i < i1) {
          const uint64_t h = a.hashes[i];
          P = DENSE ? (int64_t)(i - i0) : (int64_t)a.pos[i];
          int64_t lo = P - wl + 1 > 0 ? P - wl + 1 : 0; // smallest window start not yet excluded
          for (uint64_t j = i; j > i0;) {
            --j;
            const int64_t q = DENSE ? (int64_t)(j - i0) : (int64_t)a.pos[j];
            if (q < lo) break;
            if (a.hashes[j] <= h) { lo = q + 1; break; }
          }
          int64_t hi = P < last_s ? P : last_s;       // largest window start
          for (uint64_t j = i + 1; j < i1; ++j) {
            const int64_t q = DENSE ? (int64_t)(j - i0) : (int64_t)a.pos[j];
            if (q > hi + wl - 1) break;
            if (a.hashes[j] < h) { hi = q - wl < hi ? q - wl : hi; break; }
          }
          pick = lo <= hi;
        }
        if (pick) atomicOr((unsigned long long*)&mrow[P >> 6], 1ull << (P & 63));
        n_picked += (uint32_t)__builtin_popcountll(__ballot(pick));
      }
    }
    if (lane == 0) a.picked[r] = n_picked;
  }
}

// pass 2 -- the picked k-mers of read r to [base + out_off[r], ...), left to right; the read's offset
template <bool DENSE>
static __global__ __launch_bounds__(256) void minimizer_write_kernel(const MinimizerArgs a)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (uint64_t r = wave; r < a.n_reads; r += n_waves) {
    const uint64_t i0 = DENSE ? r * a.nwin : a.roff[r];
    const uint64_t i1 = DENSE ? i0 + a.nwin : (r + 1 < a.n_reads ? a.roff[r + 1] : a.n_kmers);
    uint64_t o = a.base + a.out_off[r];
    if (lane == 0) a.out_offsets[r] = o;
    const uint64_t* const mrow = a.masks + (a.mask_by_start ? ((a.offsets[r] - a.offsets[0]) >> 6) + r : r * a.chunks);
    for (uint64_t c0 = i0; c0 < i1; c0 += 64u) {
      const uint64_t i = c0 + lane;
      bool pick = false;
      if (i < i1) {
        const uint32_t P = DENSE ? (uint32_t)(i - i0) : a.pos[i];
        pick = (mrow[P >> 6] >> (P & 63u)) & 1ull;
      }
      const uint64_t m = __ballot(pick);
      if (pick) {
        const uint64_t at = o + (uint64_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (at < a.capacity) {
          a.out_hashes[at] = a.hashes[i];
          if (a.out_pos) a.out_pos[at] = DENSE ? (uint32_t)(i - i0) : a.pos[i];
        }
      }
      o += (uint64_t)__builtin_popcountll(m);
    }
  }
}


// ---- clean fixed-length short reads (every read emits every window, at most MZ_REG_POS of them): the table in REGISTERS ----
// The usual batch (150 bp reads without a non-base).  Position i of the read lives in lane i (and lane i - 64 of a second
// register set); a doubling step M[i] = better(M[i], M[i + d]) is a rotation of the wave by d lanes (ds_bpermute: the LDS
// crossbar, no LDS memory) and a compare, the window query better(M_J[s], M_J[s + w - 2^J]) one more such step.  The hash
// travels with its position, so when the windows' picks p(s) are known -- non-decreasing in s: a new minimizer wherever
// p(s) != p(s - 1) -- the wave has (hash, position) of every pick in registers and writes them out at once.  No masks,
// no second pass over the stream: a wave takes CHUNKS of consecutive reads and compacts each chunk's picks IN PLACE to the
// front of the chunk's own piece of the stream (a read's picks never pass the end of the read; the next read's hashes are
// in registers before this read's picks are stored), a scan over the chunks' totals, and minimizer_gather_kernel moves the
// compacted runs to their place in the output -- 17 % of the stream (w = 10) instead of all of it a second time.
// (LDS-table version above, 20 M x 150 bp: flag pass 13.5 ms -- its LDS pipe saturated, ~60 LDS operations a read -- and
//  8.5 ms for the write pass that reads the whole stream again.)
constexpr uint32_t MZ_REG_POS = 128;  // minimizer_reg_kernel
constexpr uint32_t MZ_REGN_POS = 256; // minimizer_regn_kernel<.., 4>
struct MinimizerDenseArgs {
  uint64_t* hashes;    // in: the stream; out, in place: the picks of chunk c from the chunk's first k-mer on
  uint32_t* tpos;      // the picks' window positions, same indexing (SPARSE: in as well -- the k-mers' positions)
  uint64_t n_reads;
  uint32_t nwin, w, rb, pad0; // nwin: of a fixed-length read (with offsets: unused); rb: reads per chunk (rb * 128 < 2^31)
  // SPARSE -- reads with non-bases and / or reads given by offsets: the emitted k-mers of read r are roff[r] ... roff[r + 1]
  const uint64_t* roff;    // [n_reads] (exclusive scan of the reads' counts); NULL with counts
  const uint64_t* counts;  // the read-slots form (fixed-length reads): read r's counts[r] k-mers stand at r * nwin; else NULL
  uint64_t n_kmers;
  const uint64_t* offsets; // reads of any lengths: read r = [offsets[r], offsets[r + 1]); NULL: fixed-length reads
  const uint64_t* ends;    // with offsets: read r = [offsets[r], ends[r]) (spans of a raw buffer); NULL: offsets[r + 1]
  uint32_t k, pad1;
  uint64_t* lpre;      // [n_reads] picks of the chunk's reads before read r
  uint64_t* ctot;      // [n_chunks] picks of the chunk
  // gather
  const uint64_t* coff; // exclusive scan of ctot
  uint64_t base, capacity;
  uint64_t* out_hashes;
  uint32_t* out_pos;     // may be NULL
  uint64_t* out_offsets; // [n_reads]
};

#ifndef MZ_DPP_ROT
#define MZ_DPP_ROT 1
#endif
constexpr uint32_t MZ_BUF = 256; // picks a wave collects in LDS before it writes them out (>= 2 x 128: a read always fits)
// SPARSE (reads with a non-base somewhere, reads of any lengths -- at most MZ_REG_POS windows each): the read's k-mers
// are laid out by position through 1 KiB of the wave's LDS first (a k-mer that is not there: the largest value; a window
// without a k-mer picks nothing), the window count may differ from read to read; everything after that is the same.
// ONE: no read of the batch has more than 64 windows -- the second register set and everything done to it drop out
template <bool SPARSE, bool ONE = false>
static __global__ __launch_bounds__(256) void minimizer_reg_kernel(const MinimizerDenseArgs a)
{
  // per wave: the picks of the last few reads (hash, position) and those reads' offsets -- written out together, whole
  // 512-byte pieces at a time, not after every read
  __shared__ uint64_t lds_h[4][MZ_BUF];
  __shared__ uint64_t lds_a[SPARSE ? 4 : 1][SPARSE ? MZ_REG_POS : 1];
  __shared__ uint32_t lds_l[4][64];
  __shared__ uint8_t lds_p[4][MZ_BUF];
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
  uint64_t* const hb = lds_h[wv];
  uint64_t* const A = lds_a[SPARSE ? wv : 0];
  uint32_t* const lb = lds_l[wv];
  uint8_t* const pb = lds_p[wv];
  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  };
  // fixed-length reads: one geometry for all
  uint32_t nwin = a.nwin;
  uint32_t w = a.w < nwin ? a.w : nwin; // (a read with fewer windows than w: one window)
  uint32_t n_starts = nwin - w + 1u;
  uint32_t J = 0;
  while ((2u << J) <= w) ++J; // 2^J <= w < 2^(J+1)
  uint32_t q = w - (1u << J);
  const uint64_t n_chunks = (a.n_reads + a.rb - 1u) / a.rb;
  bool v0 = lane < n_starts, v1 = lane + 64u < n_starts;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  for (uint64_t c = wave; c < n_chunks; c += n_waves) {
    const uint64_t r0 = c * a.rb;
    const uint32_t nr = (uint32_t)(r0 + a.rb < a.n_reads ? a.rb : a.n_reads - r0);
    const uint64_t kb = SPARSE && a.counts == nullptr ? a.roff[r0] : r0 * nwin; // the chunk's first k-mer
    uint64_t* const hc = a.hashes + kb; // the chunk's piece of the stream; its picks, compacted, from the front
    uint32_t* const pc = a.tpos + kb;
    uint64_t* const lc = a.lpre + r0;
    uint32_t done = 0;  // picks of the chunk already written out
    uint32_t nbuf = 0;  // picks in the buffer
    uint32_t rbuf = 0;  // reads in the buffer, the first of them read rfirst of the chunk
    uint32_t rfirst = 0;
    auto flush = [&]() {
      wave_sync();
      for (uint32_t i = lane; i < nbuf; i += 64u) {
        hc[done + i] = hb[i];
        pc[done + i] = pb[i];
      }
      if (lane < rbuf) lc[rfirst + lane] = lb[lane];
      done += nbuf;
      rfirst += rbuf;
      nbuf = 0;
      rbuf = 0;
      wave_sync();
    };
    // dense: the hashes of reads r + 1 and r + 2 are in flight while read r is worked on (one read ahead: 960 bytes a
    // wave in flight, 3.2 TB/s); unconditional loads -- a lane past the read's last window loads that window again, past
    // the chunk's last read that read again -- so that the waits can be counted
    const uint32_t l0 = lane < nwin ? lane : nwin - 1u, l1 = lane + 64u < nwin ? lane + 64u : nwin - 1u;
    uint64_t pf0 = 0, pf1 = 0, pg0 = 0, pg1 = 0;
    // sparse: the k-mers (hash, position) of read r + 1 are in flight; s_* / c_*: first k-mer (relative to the chunk's)
    // and number of k-mers of read r (cur) / r + 1 (nxt)
    uint32_t pp0 = 0, pp1 = 0, s_cur = 0, c_cur = 0, s_nxt = 0, c_nxt = 0;
    // the geometry of 64 consecutive reads of the chunk at a time, a read per lane (a scalar value that comes through a
    // vector load is a wait for everything in flight: once per 64 reads instead of once per read), handed out by readlane
    uint32_t g_st = 0, g_cn = 0, g_len = 0, g_blk = ~0u, l_blk = ~0u;
    auto geom = [&](const uint32_t rr, uint32_t& st, uint32_t& cn) { // read rr of the chunk: first k-mer (relative), k-mers
      const uint32_t blk = rr >> 6;
      if (blk != g_blk) { // (wave-uniform)
        const uint32_t rl = blk * 64u + lane;
        const uint64_t g = r0 + rl;
        g_st = 0;
        g_cn = 0;
        if (g < a.n_reads) {
          if (a.counts != nullptr) {
            g_st = rl * a.nwin;
            g_cn = (uint32_t)a.counts[g];
          } else {
            const uint64_t b = a.roff[g], e = g + 1u < a.n_reads ? a.roff[g + 1u] : a.n_kmers;
            g_st = (uint32_t)(b - kb);
            g_cn = (uint32_t)(e - b);
          }
        }
        g_blk = blk;
      }
      st = (uint32_t)__builtin_amdgcn_readlane((int)g_st, (int)(rr & 63u));
      cn = (uint32_t)__builtin_amdgcn_readlane((int)g_cn, (int)(rr & 63u));
    };
    auto read_len = [&](const uint32_t rr) -> uint32_t { // bases of read rr of the chunk (reads given by offsets)
      const uint32_t blk = rr >> 6;
      if (blk != l_blk) {
        const uint64_t g = r0 + blk * 64u + lane;
        uint64_t l = 0;
        if (g < a.n_reads) l = (a.ends != nullptr ? a.ends[g] : a.offsets[g + 1u]) - a.offsets[g];
        g_len = l < 0xFFFFFFFFull ? (uint32_t)l : 0xFFFFFFFFu;
        l_blk = blk;
      }
      return (uint32_t)__builtin_amdgcn_readlane((int)g_len, (int)(rr & 63u));
    };
    // (read-slots form: a read that fills its slot -- counts[r] == nwin, no non-base -- has its k-mers at their window
    //  indices; its positions are neither stored nor read, and it needs no layout through LDS)
    auto load_sparse = [&](const uint32_t i0, const uint32_t cn) {
      const uint32_t i1 = i0 + cn;
      if (cn != 0u) { // (wave-uniform)
        const uint32_t e0 = i0 + lane < i1 ? i0 + lane : i1 - 1u, e1 = i0 + 64u + lane < i1 ? i0 + 64u + lane : i1 - 1u;
        pf0 = hc[e0];
        pp0 = pc[e0]; // (of a read that fills its slot: not written, not used -- loaded all the same: no branch among the loads)
        if constexpr (!ONE) {
          pf1 = hc[e1];
          pp1 = pc[e1];
        }
      }
    };
    if constexpr (SPARSE) {
      geom(0u, s_cur, c_cur);
      geom(1u, s_nxt, c_nxt);
      load_sparse(s_cur, c_cur);
    } else {
      pf0 = hc[l0];
      if constexpr (!ONE) pf1 = hc[l1];
      const uint32_t bn = (nr > 1u ? 1u : 0u) * nwin;
      pg0 = hc[bn + l0];
      if constexpr (!ONE) pg1 = hc[bn + l1];
    }
    for (uint32_t r = 0; r < nr; ++r) {
      uint32_t h0l, h0h, h1l, h1h;
      uint32_t p0 = lane, p1 = lane + 64u;
      if constexpr (SPARSE) {
        const uint32_t cnt = c_cur; // k-mers of the read (<= its windows)
        if (a.offsets != nullptr) {         // this read's windows
          const uint32_t l = read_len(r);
          nwin = l >= a.k ? l - a.k + 1u : 0u;
          w = a.w < nwin ? a.w : nwin;
          n_starts = nwin - w + 1u;
          J = 0;
          while ((2u << J) <= w) ++J;
          q = w - (1u << J);
          v0 = lane < n_starts;
          v1 = lane + 64u < n_starts;
        }
        const bool full = cnt == nwin && cnt != 0u; // every window of the read is there: the k-mers stand at their window indices
        if (cnt != 0u && !full) {
          A[lane] = ~0ull;
          if constexpr (!ONE) A[lane + 64u] = ~0ull;
          wave_sync();
          if (lane < cnt) A[pp0] = pf0;
          if constexpr (!ONE)
            if (lane + 64u < cnt) A[pp1] = pf1;
          wave_sync();
        }
        const uint64_t x0 = full ? pf0 : cnt != 0u ? A[lane] : ~0ull;
        const uint64_t x1 = ONE ? ~0ull : full ? pf1 : cnt != 0u ? A[lane + 64u] : ~0ull;
        h0l = (uint32_t)x0; h0h = (uint32_t)(x0 >> 32); h1l = (uint32_t)x1; h1h = (uint32_t)(x1 >> 32);
        s_cur = s_nxt;
        c_cur = c_nxt;
        geom(r + 2u, s_nxt, c_nxt);
        if (r + 1u < nr) load_sparse(s_cur, c_cur);
        if (cnt == 0u) { // no k-mer, no pick (a read shorter than k, a read of non-bases)
          if (rbuf == 64u) flush();
          if (lane == 0u) lb[rbuf] = done + nbuf;
          ++rbuf;
          continue;
        }
      } else {
        h0l = (uint32_t)pf0; h0h = (uint32_t)(pf0 >> 32); h1l = (uint32_t)pf1; h1h = (uint32_t)(pf1 >> 32);
        pf0 = pg0;
        pf1 = pg1;
        const uint32_t b = (r + 2u < nr ? r + 2u : nr - 1u) * nwin;
        pg0 = hc[b + l0];
        if constexpr (!ONE) pg1 = hc[b + l1];
      }
      // (the buffered picks belong to reads before r: their place in the stream is below read r's hashes)
      if (nbuf + n_starts > MZ_BUF || rbuf == 64u) flush();
      // M[i] = better(M[i], M[i + d]): the left one wins a tie.  (Positions past the read hold anything: no window of the
      // read looks at a range that reaches them.)
      // The rotation by d lanes: ds_bpermute (the LDS crossbar: six of them a step are what this kernel waits for) -- or,
      // for d = 1 and 2, the whole-wave rotate of the DPP network (wave_rol:1, lane i reads lane i + 1 mod 64: VALU moves).
      auto better = [&](const uint32_t d, const uint32_t a0l, const uint32_t a0h, const uint32_t a0p, const uint32_t a1l,
                        const uint32_t a1h, const uint32_t a1p) {
        const bool low = ONE || lane + d < 64u; // position lane + d is in set 0
        const uint32_t n0l = low ? a0l : a1l, n0h = low ? a0h : a1h, n0p = low ? a0p : a1p;
        const bool t0 = (((uint64_t)n0h << 32) | n0l) < (((uint64_t)h0h << 32) | h0l);
        h0l = t0 ? n0l : h0l; h0h = t0 ? n0h : h0h; p0 = t0 ? n0p : p0;
        if constexpr (!ONE) {
          const bool t1 = (((uint64_t)a1h << 32) | a1l) < (((uint64_t)h1h << 32) | h1l);
          h1l = t1 ? a1l : h1l; h1h = t1 ? a1h : h1h; p1 = t1 ? a1p : p1;
        }
      };
      auto step = [&](const uint32_t d) {
        const int idx = (int)(((lane + d) & 63u) << 2);
        const uint32_t a0l = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)h0l), a0h = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)h0h);
        const uint32_t a0p = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)p0);
        uint32_t a1l = 0, a1h = 0, a1p = 0;
        if constexpr (!ONE) {
          a1l = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)h1l);
          a1h = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)h1h);
          a1p = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)p1);
        }
        better(d, a0l, a0h, a0p, a1l, a1h, a1p);
      };
      auto rol1 = [](const uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x134, 0xf, 0xf, false); };
      auto step_rol = [&](auto d_tag) {
        constexpr uint32_t D = decltype(d_tag)::value; // 1 or 2
        uint32_t a0l = rol1(h0l), a0h = rol1(h0h), a0p = rol1(p0), a1l = 0, a1h = 0, a1p = 0;
        if constexpr (!ONE) {
          a1l = rol1(h1l); a1h = rol1(h1h); a1p = rol1(p1);
        }
        if constexpr (D == 2u) {
          a0l = rol1(a0l); a0h = rol1(a0h); a0p = rol1(a0p);
          if constexpr (!ONE) {
            a1l = rol1(a1l); a1h = rol1(a1h); a1p = rol1(a1p);
          }
        }
        better(D, a0l, a0h, a0p, a1l, a1h, a1p);
      };
      auto step_any = [&](const uint32_t d) {
#if MZ_DPP_ROT
        if (d == 1u) step_rol(std::integral_constant<uint32_t, 1u>{});
        else if (d == 2u) step_rol(std::integral_constant<uint32_t, 2u>{});
        else
#endif
          step(d);
      };
      for (uint32_t j = 0; j < J; ++j) step_any(1u << j);
      if (q != 0u) step_any(q);
      // p(s) is non-decreasing: a new minimizer wherever it moves
      // (wave_ror:1: lane i reads lane i - 1 mod 64 -- lane 0 of the second set wants lane 63 of the first)
      const uint32_t pr0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p0, 0x13C, 0xf, 0xf, false);
      uint32_t pr1 = ONE ? 0u : (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p1, 0x13C, 0xf, 0xf, false);
      pr1 = lane == 0u ? pr0 : pr1;
      bool new0 = v0 && (lane == 0u || p0 != pr0);
      bool new1 = !ONE && v1 && p1 != pr1;
      if constexpr (SPARSE) { // a window without a k-mer picks nothing
        new0 = new0 && (h0l & h0h) != ~0u;
        new1 = new1 && (h1l & h1h) != ~0u;
      }
      const uint64_t b0 = __ballot(new0), b1 = __ballot(new1);
      const uint32_t c0 = (uint32_t)__builtin_popcountll(b0), c1 = (uint32_t)__builtin_popcountll(b1);
      if (new0) {
        const uint32_t at = nbuf + (uint32_t)__builtin_popcountll(b0 & lt_mask);
        hb[at] = ((uint64_t)h0h << 32) | h0l;
        pb[at] = (uint8_t)p0;
      }
      if (new1) {
        const uint32_t at = nbuf + c0 + (uint32_t)__builtin_popcountll(b1 & lt_mask);
        hb[at] = ((uint64_t)h1h << 32) | h1l;
        pb[at] = (uint8_t)p1;
      }
      if (lane == 0u) lb[rbuf] = done + nbuf;
      nbuf += c0 + c1;
      ++rbuf;
    }
    flush();
    if (lane == 0u) a.ctot[c] = done;
  }
}

// The same for reads of up to 64 NS windows (NS register sets; instantiated for NS = 4: 250 bp reads).  Written over arrays
// of NS sets; a step by d = 64 ds + dl lanes is a rotation by dl (none for the powers of two from 64 on) and a choice of
// sets.  One read of hashes in flight.
template <bool SPARSE, uint32_t NS>
static __global__ __launch_bounds__(256) void minimizer_regn_kernel(const MinimizerDenseArgs a)
{
  constexpr uint32_t BUF = 128u * NS; // (a read's picks always fit behind what the buffer holds: n_starts <= 64 NS)
  __shared__ uint64_t lds_h[4][BUF];
  __shared__ uint64_t lds_a[SPARSE ? 4 : 1][SPARSE ? 64u * NS : 1];
  __shared__ uint32_t lds_l[4][64];
  __shared__ uint8_t lds_p[4][BUF];
  static_assert(64u * NS <= 256u, "positions are bytes in the buffer");
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
  uint64_t* const hb = lds_h[wv];
  uint64_t* const A = lds_a[SPARSE ? wv : 0];
  uint32_t* const lb = lds_l[wv];
  uint8_t* const pb = lds_p[wv];
  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  };
  uint32_t nwin = a.nwin;
  uint32_t w = a.w < nwin ? a.w : nwin;
  uint32_t n_starts = nwin - w + 1u;
  uint32_t J = 0;
  while ((2u << J) <= w) ++J;
  uint32_t q = w - (1u << J);
  const uint64_t n_chunks = (a.n_reads + a.rb - 1u) / a.rb;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  for (uint64_t c = wave; c < n_chunks; c += n_waves) {
    const uint64_t r0 = c * a.rb;
    const uint32_t nr = (uint32_t)(r0 + a.rb < a.n_reads ? a.rb : a.n_reads - r0);
    const uint64_t kb = SPARSE && a.counts == nullptr ? a.roff[r0] : r0 * nwin;
    uint64_t* const hc = a.hashes + kb;
    uint32_t* const pc = a.tpos + kb;
    uint64_t* const lc = a.lpre + r0;
    uint32_t done = 0, nbuf = 0, rbuf = 0, rfirst = 0;
    auto flush = [&]() {
      wave_sync();
      for (uint32_t i = lane; i < nbuf; i += 64u) {
        hc[done + i] = hb[i];
        pc[done + i] = pb[i];
      }
      if (lane < rbuf) lc[rfirst + lane] = lb[lane];
      done += nbuf;
      rfirst += rbuf;
      nbuf = 0;
      rbuf = 0;
      wave_sync();
    };
    uint64_t pf[NS];
    uint32_t pp[NS];
    uint32_t s_cur = 0, c_cur = 0, s_nxt = 0, c_nxt = 0;
    // the geometry of 64 consecutive reads of the chunk at a time, a read per lane (a scalar value that comes through a
    // vector load is a wait for everything in flight: once per 64 reads instead of once per read), handed out by readlane
    uint32_t g_st = 0, g_cn = 0, g_len = 0, g_blk = ~0u, l_blk = ~0u;
    auto geom = [&](const uint32_t rr, uint32_t& st, uint32_t& cn) { // read rr of the chunk: first k-mer (relative), k-mers
      const uint32_t blk = rr >> 6;
      if (blk != g_blk) { // (wave-uniform)
        const uint32_t rl = blk * 64u + lane;
        const uint64_t g = r0 + rl;
        g_st = 0;
        g_cn = 0;
        if (g < a.n_reads) {
          if (a.counts != nullptr) {
            g_st = rl * a.nwin;
            g_cn = (uint32_t)a.counts[g];
          } else {
            const uint64_t b = a.roff[g], e = g + 1u < a.n_reads ? a.roff[g + 1u] : a.n_kmers;
            g_st = (uint32_t)(b - kb);
            g_cn = (uint32_t)(e - b);
          }
        }
        g_blk = blk;
      }
      st = (uint32_t)__builtin_amdgcn_readlane((int)g_st, (int)(rr & 63u));
      cn = (uint32_t)__builtin_amdgcn_readlane((int)g_cn, (int)(rr & 63u));
    };
    auto read_len = [&](const uint32_t rr) -> uint32_t { // bases of read rr of the chunk (reads given by offsets)
      const uint32_t blk = rr >> 6;
      if (blk != l_blk) {
        const uint64_t g = r0 + blk * 64u + lane;
        uint64_t l = 0;
        if (g < a.n_reads) l = (a.ends != nullptr ? a.ends[g] : a.offsets[g + 1u]) - a.offsets[g];
        g_len = l < 0xFFFFFFFFull ? (uint32_t)l : 0xFFFFFFFFu;
        l_blk = blk;
      }
      return (uint32_t)__builtin_amdgcn_readlane((int)g_len, (int)(rr & 63u));
    };
    auto load_read = [&](const uint32_t i0, const uint32_t cn) { // the read's k-mers (dense: its windows), a lane past the last one loads it again
      if (cn != 0u) {
#pragma unroll
        for (uint32_t s_ = 0; s_ < NS; ++s_) {
          const uint32_t e = i0 + (s_ * 64u + lane < cn ? s_ * 64u + lane : cn - 1u);
          pf[s_] = hc[e];
          if constexpr (SPARSE) pp[s_] = pc[e]; // (of a read that fills its slot: not written, not used)
        }
      }
    };
#pragma unroll
    for (uint32_t s_ = 0; s_ < NS; ++s_) {
      pf[s_] = 0;
      pp[s_] = 0;
    }
    if constexpr (SPARSE) {
      geom(0u, s_cur, c_cur);
      geom(1u, s_nxt, c_nxt);
    } else {
      s_cur = 0;
      c_cur = nwin;
      s_nxt = nwin;
      c_nxt = nwin;
    }
    load_read(s_cur, c_cur);
    for (uint32_t r = 0; r < nr; ++r) {
      uint32_t hl[NS], hh[NS], p[NS];
      const uint32_t cnt = c_cur;
      if constexpr (SPARSE) {
        if (a.offsets != nullptr) {
          const uint32_t l = read_len(r);
          nwin = l >= a.k ? l - a.k + 1u : 0u;
          w = a.w < nwin ? a.w : nwin;
          n_starts = nwin - w + 1u;
          J = 0;
          while ((2u << J) <= w) ++J;
          q = w - (1u << J);
        }
        const bool full = cnt == nwin && cnt != 0u; // every window of the read is there: the k-mers stand at their window indices
        if (cnt != 0u && !full) {
#pragma unroll
          for (uint32_t s_ = 0; s_ < NS; ++s_) A[s_ * 64u + lane] = ~0ull;
          wave_sync();
#pragma unroll
          for (uint32_t s_ = 0; s_ < NS; ++s_)
            if (s_ * 64u + lane < cnt) A[pp[s_]] = pf[s_];
          wave_sync();
        }
#pragma unroll
        for (uint32_t s_ = 0; s_ < NS; ++s_) {
          const uint64_t x = full ? pf[s_] : cnt != 0u ? A[s_ * 64u + lane] : ~0ull;
          hl[s_] = (uint32_t)x;
          hh[s_] = (uint32_t)(x >> 32);
        }
      } else {
#pragma unroll
        for (uint32_t s_ = 0; s_ < NS; ++s_) {
          hl[s_] = (uint32_t)pf[s_];
          hh[s_] = (uint32_t)(pf[s_] >> 32);
        }
      }
#pragma unroll
      for (uint32_t s_ = 0; s_ < NS; ++s_) p[s_] = s_ * 64u + lane;
      // the next read's k-mers on their way
      s_cur = s_nxt;
      c_cur = c_nxt;
      if constexpr (SPARSE) geom(r + 2u, s_nxt, c_nxt);
      else s_nxt += nwin;
      if (r + 1u < nr) load_read(s_cur, c_cur);
      if (SPARSE && cnt == 0u) {
        if (rbuf == 64u) flush();
        if (lane == 0u) lb[rbuf] = done + nbuf;
        ++rbuf;
        continue;
      }
      if (nbuf + n_starts > BUF || rbuf == 64u) flush();
      // M[i] = better(M[i], M[i + d]), d = 64 ds + dl
      auto step = [&](const uint32_t d) {
        const uint32_t dl = d & 63u, ds = d >> 6;
        uint32_t al[NS], ah[NS], ap[NS];
        if (dl == 0u) {
#pragma unroll
          for (uint32_t s_ = 0; s_ < NS; ++s_) {
            al[s_] = hl[s_];
            ah[s_] = hh[s_];
            ap[s_] = p[s_];
          }
        } else {
          const int idx = (int)(((lane + dl) & 63u) << 2);
#pragma unroll
          for (uint32_t s_ = 0; s_ < NS; ++s_) {
            al[s_] = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)hl[s_]);
            ah[s_] = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)hh[s_]);
            ap[s_] = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)p[s_]);
          }
        }
        const bool low = lane + dl < 64u;
        auto apply = [&](auto ds_tag) {
          constexpr uint32_t DS = decltype(ds_tag)::value;
#pragma unroll
          for (uint32_t s_ = 0; s_ < NS; ++s_) {
            const uint32_t i0 = s_ + DS, i1 = s_ + DS + 1u;
            if (i0 < NS) { // (else: no such position)
              const uint32_t j1 = i1 < NS ? i1 : i0;
              const uint32_t nl = low ? al[i0] : al[j1], nh = low ? ah[i0] : ah[j1], np = low ? ap[i0] : ap[j1];
              const bool t = (((uint64_t)nh << 32) | nl) < (((uint64_t)hh[s_] << 32) | hl[s_]);
              hl[s_] = t ? nl : hl[s_];
              hh[s_] = t ? nh : hh[s_];
              p[s_] = t ? np : p[s_];
            }
          }
        };
        switch (ds) {
          case 0: apply(std::integral_constant<uint32_t, 0u>{}); break;
          case 1: apply(std::integral_constant<uint32_t, 1u>{}); break;
          case 2: apply(std::integral_constant<uint32_t, 2u>{}); break;
          default: apply(std::integral_constant<uint32_t, 3u>{}); break;
        }
      };
      for (uint32_t j = 0; j < J; ++j) step(1u << j);
      if (q != 0u) step(q);
      // p(s) is non-decreasing: a new minimizer wherever it moves
      uint32_t off = nbuf;
      uint32_t prev_ror = 0;
#pragma unroll
      for (uint32_t s_ = 0; s_ < NS; ++s_) {
        const uint32_t ror = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p[s_], 0x13C, 0xf, 0xf, false); // lane i reads lane i - 1 mod 64
        const uint32_t pr = lane == 0u ? prev_ror : ror; // (lane 0: lane 63 of the set before)
        prev_ror = ror;
        bool nw_ = s_ * 64u + lane < n_starts && ((s_ == 0u && lane == 0u) || p[s_] != pr);
        if constexpr (SPARSE) nw_ = nw_ && (hl[s_] & hh[s_]) != ~0u;
        const uint64_t b = __ballot(nw_);
        if (nw_) {
          const uint32_t at = off + (uint32_t)__builtin_popcountll(b & lt_mask);
          hb[at] = ((uint64_t)hh[s_] << 32) | hl[s_];
          pb[at] = (uint8_t)p[s_];
        }
        off += (uint32_t)__builtin_popcountll(b);
      }
      if (lane == 0u) lb[rbuf] = done + nbuf;
      nbuf = off;
      ++rbuf;
    }
    flush();
    if (lane == 0u) a.ctot[c] = done;
  }
}

// the chunks' compacted picks to [base + coff[c], ...) of the output; every read's offset
static __global__ __launch_bounds__(256) void minimizer_gather_kernel(const MinimizerDenseArgs a)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  const uint64_t n_chunks = (a.n_reads + a.rb - 1u) / a.rb;
  for (uint64_t c = wave; c < n_chunks; c += n_waves) {
    const uint64_t r0 = c * a.rb;
    const uint64_t r1 = r0 + a.rb < a.n_reads ? r0 + a.rb : a.n_reads;
    const uint64_t k0 = a.roff ? a.roff[r0] : r0 * a.nwin;
    const uint64_t o0 = a.base + a.coff[c];
    const uint32_t n = (uint32_t)a.ctot[c];
    for (uint32_t i = lane; i < n; i += 64u) {
      const uint64_t at = o0 + i;
      if (at < a.capacity) {
        a.out_hashes[at] = a.hashes[k0 + i];
        if (a.out_pos) a.out_pos[at] = a.tpos[k0 + i];
      }
    }
    for (uint64_t r = r0 + lane; r < r1; r += 64u) a.out_offsets[r] = o0 + a.lpre[r];
  }
}

// per-read MinHash signatures from a hash stream (reads of any lengths: read r = k-mers roff[r] ... roff[r + 1], ONE value
// -- h[0] -- per k-mer): sig[r][j] = min over the read's k-mers of h[j] (extend_hashes, reference src/internal.hpp:104-118,
// recomputed here from h[0]); all ones for a read without a k-mer.  One wave per read at a time.
static __global__ __launch_bounds__(256) void stream_minhash_kernel(const uint64_t* __restrict__ hashes, const uint64_t* __restrict__ roff,
                                                                    uint64_t n_reads, uint64_t n_kmers, uint32_t m, uint64_t kmul,
                                                                    uint64_t* __restrict__ sig)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (uint64_t r = wave; r < n_reads; r += n_waves) {
    const uint64_t i0 = roff[r], i1 = r + 1 < n_reads ? roff[r + 1] : n_kmers;
    for (uint32_t j0 = 0; j0 < m; j0 += 4u) {
      uint64_t mn[4] = {~0ull, ~0ull, ~0ull, ~0ull};
      for (uint64_t e = i0 + lane; e < i1; e += 64u) {
        const uint64_t h0 = hashes[e];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
          const uint32_t j = j0 + u;
          const uint64_t v = j == 0u ? h0 : mix_hash(h0, (uint64_t)j ^ kmul);
          mn[u] = v < mn[u] ? v : mn[u];
        }
      }
#pragma unroll
      for (uint32_t u = 0; u < 4u; ++u) {
        uint64_t v = mn[u];
        for (int d = 32; d > 0; d >>= 1) {
          const uint64_t o = ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), d, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)v, d, 64);
          v = o < v ? o : v;
        }
        if (lane == 0 && j0 + u < m) sig[r * m + j0 + u] = v;
      }
    }
  }
}

} // namespace ntamd
