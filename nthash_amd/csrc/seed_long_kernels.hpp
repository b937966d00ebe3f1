// seed_long_kernels.hpp -- SeedNtHash on LONG reads (chromosomes, contigs): cutting a read into independent pieces.
// Part of libnthash_hip.so (include/nthash_hip.h); see capi_internal.hpp for the file map.
//
// seed_wave_kernel walks a read with one wave, segment after segment, because the reference's walk has a state
// (src/seed.cpp:493-544: after a non-base came IN at q the hasher restarts at the window that starts AT q, and the next
// non-base only counts from q + k on).  That state is forgotten after k clean characters: if the 2k characters
// [c - k, c + k) around a position c are all bases, then
//   * the windows before c only depend on characters before c + k - 1, none of which is a non-base from c - k on;
//   * the first non-base after c lies at q >= c + k, further than k from any earlier restart, so it restarts the walk
//     whatever happened before c -- exactly as in a read that BEGINS at c (there the first k characters cannot restart
//     anything, here they are bases), whose first window needs no NUL check either.
// So [0, c + k - 1) and [c, len) are two independent reads whose k-mer streams concatenate to the read's.  The kernels
// here find such cuts near every S-th window (none inside a long run of non-bases: that piece simply stays long), and
// the pieces go through the span path (one wave per piece instead of one per read); counts and positions are folded
// back per read afterwards.
#pragma once
#include "nt_math.hpp"

namespace ntamd {

struct SeedLongArgs {
  const uint8_t* seqs;
  const uint64_t* offsets;  // nullptr: fixed length
  const uint64_t* ends;     // optional with offsets (spans)
  uint64_t n_reads;
  uint64_t len, stride;     // fixed length
  uint32_t k;
  uint32_t S;               // nominal windows per piece
  const uint64_t* pbase;    // [n_reads + 1] first nominal piece of a read (offsets input); fixed: pieces_per_read
  uint64_t pieces_per_read; // fixed length
  uint64_t n_pieces;        // nominal pieces of the batch
};

__device__ __forceinline__ void seed_long_read_of(const SeedLongArgs& a, uint64_t r, uint64_t& start, uint64_t& len)
{
  if (a.offsets) {
    start = a.offsets[r];
    const uint64_t e = a.ends ? a.ends[r] : a.offsets[r + 1];
    len = e > start ? e - start : 0;
  } else {
    start = r * a.stride;
    len = a.len;
  }
}
__device__ __forceinline__ uint64_t seed_long_pieces_of(uint64_t len, uint32_t k, uint32_t S)
{
  if (len < k) return 1;
  const uint64_t nwin = len - k + 1;
  return (nwin + S - 1) / S;
}

// nominal pieces per read (offsets input)
static __global__ __launch_bounds__(256) void seed_long_count_kernel(const SeedLongArgs a, uint64_t* __restrict__ pieces)
{
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.n_reads; r += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t s, l;
    seed_long_read_of(a, r, s, l);
    pieces[r] = seed_long_pieces_of(l, a.k, a.S);
  }
}

// nominal piece p -> (read, index inside the read)
__device__ __forceinline__ void seed_long_piece(const SeedLongArgs& a, uint64_t p, uint64_t& r, uint64_t& j)
{
  if (!a.offsets) {
    r = p / a.pieces_per_read;
    j = p - r * a.pieces_per_read;
    return;
  }
  uint64_t lo = 0, hi = a.n_reads; // last r with pbase[r] <= p
  while (hi - lo > 1) {
    const uint64_t mid = (lo + hi) >> 1;
    if (a.pbase[mid] <= p) lo = mid; else hi = mid;
  }
  r = lo;
  j = p - a.pbase[lo];
}

// One lane per nominal piece: the last character of its S bytes [j S, (j + 1) S) that is not the letter N / n
// (~0: there is none) -- what the cut kernel needs to find where a long run of N began.
static __global__ __launch_bounds__(256) void seed_long_lastbase_kernel(const SeedLongArgs a, uint64_t* __restrict__ last_not_n)
{
  for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < a.n_pieces; p += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t r, j;
    seed_long_piece(a, p, r, j);
    uint64_t start, len;
    seed_long_read_of(a, r, start, len);
    const uint64_t lo = j * (uint64_t)a.S;
    uint64_t hi = lo + a.S;
    if (hi > len) hi = len;
    const uint8_t* s = a.seqs + start;
    uint64_t found = ~0ull;
    for (uint64_t x = hi; x-- > lo;)
      if ((s[x] | 0x20u) != 'n') {
        found = x;
        break;
      }
    last_not_n[p] = found;
  }
}

// One lane per nominal piece: the cut that starts it.  Piece 0 of a read starts at 0; piece j > 0 at the first
// c in [j S, j S + S / 2) with [c - k, c + k) all bases, if there is one (valid = 0 otherwise: the piece before it grows).
//
// Inside a long run of the letter N there is no such c, but there is another kind of exact cut: a position where the
// walk RESTARTS.  If the run begins at r0 >= k behind k bases, its first N restarts the walk (no earlier restart lies
// within k), and so does every k-th N after it; a run that begins before position k restarts at k, 2k, ...
// (src/seed.cpp:518-544).  A piece that starts AT a restart c is an independent read: the reference calls init() at c
// there too (NUL check included), the characters c + 1 .. c + k - 1 cannot restart anything in either walk, and the
// windows before c only look at characters before c + k - 1.
static __global__ __launch_bounds__(256) void seed_long_cut_kernel(const SeedLongArgs a, const uint64_t* __restrict__ last_not_n,
                                                                   uint64_t* __restrict__ valid, uint64_t* __restrict__ cut)
{
  for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < a.n_pieces; p += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t r, j;
    seed_long_piece(a, p, r, j);
    if (j == 0) {
      valid[p] = 1;
      cut[p] = 0;
      continue;
    }
    uint64_t start, len;
    seed_long_read_of(a, r, start, len);
    const uint64_t nwin = len - a.k + 1; // (j > 0: the read has windows)
    const uint64_t c0 = j * (uint64_t)a.S;
    uint64_t c_hi = c0 + a.S / 2;        // cuts searched in [c0, c_hi)
    if (c_hi > nwin) c_hi = nwin;
    const uint8_t* s = a.seqs + start;
    uint64_t run = 0, found = ~0ull;
    // x runs over the characters from c0 - k on; a cut at c needs run >= 2k at x = c + k - 1
    for (uint64_t x = c0 - a.k; x < c_hi + a.k - 1 && x < len; ++x) {
      run = is_base(s[x]) ? run + 1 : 0;
      if (run >= 2ull * a.k) {
        found = x + 1 - a.k; // >= c0 because x >= c0 + k - 1 when run reaches 2k from c0 - k on
        break;
      }
    }
    bool ok = found != ~0ull && found < c_hi;
    if (!ok && c0 < len && (s[c0] | 0x20u) == 'n' && (s[c0 - 1] | 0x20u) == 'n') {
      // ---- a run of N around c0: where did it begin? (the pieces before this one know their last other character)
      uint64_t r0 = 0; // first N of the run
      for (uint64_t q = p; q-- > p - j;) {
        const uint64_t l = last_not_n[q];
        if (l != ~0ull) {
          r0 = l + 1;
          break;
        }
      }
      uint64_t q0 = ~0ull; // the run's first restart
      if (r0 < a.k) {
        // (c0 >= S >= 4k: position k is inside the run.  The characters before the run must be bases: a NUL among them
        //  would move the read's first init(), src/seed.cpp:493-516)
        bool bases = true;
        for (uint64_t x = 0; x < r0 && bases; ++x) bases = is_base(s[x]);
        if (bases) q0 = a.k;
      } else {
        bool bases = true;
        for (uint64_t x = r0 - a.k; x < r0 && bases; ++x) bases = is_base(s[x]);
        if (bases) q0 = r0;
      }
      if (q0 != ~0ull) {
        const uint64_t c = q0 + (c0 - q0 + a.k - 1) / a.k * a.k; // the first restart at or after c0
        bool run = c < c_hi;
        for (uint64_t x = c0; x <= c && run; ++x) run = (s[x] | 0x20u) == 'n';
        if (run) {
          ok = true;
          found = c;
        }
      }
    }
    valid[p] = ok ? 1 : 0;
    cut[p] = found;
  }
}

// Compaction: the valid cuts become the pieces' spans [start + cut, next cut of the same read + k - 1 | end of the read)
static __global__ __launch_bounds__(256) void seed_long_spans_kernel(const SeedLongArgs a, const uint64_t* __restrict__ valid,
                                                                     const uint64_t* __restrict__ idx,
                                                                     const uint64_t* __restrict__ cut,
                                                                     uint64_t* __restrict__ sub_start,
                                                                     uint64_t* __restrict__ sub_end,
                                                                     uint64_t* __restrict__ sub_read,
                                                                     uint64_t* __restrict__ sub_rel)
{
  for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < a.n_pieces; p += (uint64_t)gridDim.x * blockDim.x) {
    if (!valid[p]) continue;
    uint64_t r, j;
    seed_long_piece(a, p, r, j);
    uint64_t start, len;
    seed_long_read_of(a, r, start, len);
    const uint64_t np = a.offsets ? a.pbase[r + 1] - a.pbase[r] : a.pieces_per_read;
    uint64_t end = len; // the read's last piece
    for (uint64_t q = p + 1; q < p + (np - j); ++q) // the next valid cut of the same read (usually the next piece)
      if (valid[q]) {
        end = cut[q] + a.k - 1;
        break;
      }
    const uint64_t i = idx[p];
    sub_start[i] = start + cut[p];
    sub_end[i] = start + end;
    sub_read[i] = r;
    sub_rel[i] = cut[p];
  }
}

// After the span path: per-read counts = sum of the pieces' counts; get_pos() of a piece's k-mers += the piece's cut
static __global__ __launch_bounds__(256) void seed_long_fold_counts_kernel(const uint64_t* __restrict__ sub_cnt,
                                                                           const uint64_t* __restrict__ sub_read, uint64_t m,
                                                                           uint64_t* __restrict__ counts)
{
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (uint64_t)gridDim.x * blockDim.x)
    if (sub_cnt[i]) atomicAdd((unsigned long long*)&counts[sub_read[i]], (unsigned long long)sub_cnt[i]);
}
static __global__ __launch_bounds__(256) void seed_long_fold_pos_kernel(const uint64_t* __restrict__ sub_cnt,
                                                                        const uint64_t* __restrict__ sub_off,
                                                                        const uint64_t* __restrict__ sub_rel, uint64_t m,
                                                                        uint32_t* __restrict__ pos)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (uint64_t i = wave; i < m; i += n_waves) {
    const uint64_t rel = sub_rel[i];
    if (rel == 0) continue;
    const uint64_t off = sub_off[i], cnt = sub_cnt[i];
    for (uint64_t t = lane; t < cnt; t += 64u) pos[off + t] += (uint32_t)rel;
  }
}

} // namespace ntamd
