// kmer_runs_gen_kernel.hpp -- the run-split kernels for ANY fixed read shape:
//   kmer_runs_gen_kernel<NW, DT, false>  dense stream (optimistic pass: every byte a base)
//   kmer_runs_gen_kernel<NW, DT, true>   N-aware hash pass: compact stream at scanned tile offsets
//   kmer_runs_count_kernel               N-aware count pass: valid windows per tile / per read
//   kmer_runs_gen_kernel<NW, DT, true, SINK_BLOOM_INSERT / SINK_BLOOM_QUERY / SINK_MINHASH>
//                                        fused consumers: the hashes of a tile go from LDS straight
//                                        into a Bloom filter (set / test bits), or from registers
//                                        into per-read MinHash signatures -- never to HBM
//
// kmer_runs_kernel.hpp needs the run length C to divide the window count and
// stages whole reads; that leaves cliffs (a prime window count, reads of 10 kb).
// These kernels keep its structure -- 64 lanes own 64 consecutive runs, wave-private
// 2-bit slab + output tile, prefetch of the next slab with a counted s_waitcnt --
// and generalise the geometry:
//   * runs per read rpr = ceil(nwin / C); every run has C windows, and the LAST run
//     of a read starts at window nwin - C, i.e. it overlaps its predecessor and
//     recomputes a few windows (same values to the same addresses) instead of being
//     short -- no per-lane trip counts, no predicated stores.  Runs stay in stream
//     order, so the 64 runs of a wave still cover one contiguous piece of the hash
//     stream; only its length and its alignment now vary per tile.  The tile is built
//     in LDS shifted by its position inside a 1 KiB block of the output and leaves
//     as aligned 16-byte stores with an 8-byte head / tail.
//   * the slab is the byte range the 64 runs really touch (first base of the first
//     run .. last base of the last run), not whole reads.
//
// N-aware (NtHash emits exactly the windows whose k bytes are all bases: the net
// effect of init()/roll(), src/kmer.cpp:228-264).  That is a
// parallel predicate: one validity bit per base, OR-ed over every window by
// doubling.  A run's recomputed windows are masked out, so every window counts
// once.  Count pass -> device scan of the tile counts -> hash pass; a non-base keeps
// a garbage 2-bit code that enters and leaves the rolled state with the same code,
// so every all-base window is exact.  Tiles without any non-base (almost all of
// them in real data) take the same unpredicated code path as the dense kernel.
//
// Hash arithmetic as in kmer_runs_kernel.hpp (first window from the byte tables,
// src/kmer.cpp:43-73,123-152; the rest rolled, src/kmer.cpp:84-94,164-174).
#pragma once
#include "bloom_math.hpp" // (mod_invariant)

#include <hip/hip_runtime.h>

#include "first_window.hpp"
#include "kmer_runs_kernel.hpp"

namespace ntamd {

// Several hashes per k-mer (extend_hashes, reference src/internal.hpp:104-118), fused into the copy-out of a wave's tile:
// the tile holds h[0] of n_emit consecutive k-mers, the first of them k-mer o0 of the stream; stream value v is
// h[v % m] of k-mer v / m.  A lane writes one aligned 16-byte piece (two values) per instruction, the wave whole lines.
// No division in the loop: a lane's values advance by 128 per iteration -- k-mer += 128 / m, hash index += 128 % m with
// a carry -- and the first value of the piece follows from the second (round 3: the loop before took a multiply-high and
// a multiply per value for v / m and v % m, with the three of the 64-bit product five quarter-rate instructions a value).
// The half pieces at either end (the tile starts / ends on an odd value) are single 8-byte stores of two lanes.  The
// lane -> piece map starts at a LINE of the stream, not at the tile's first value: every store instruction of the wave
// covers whole aligned 128-byte lines (the lanes before the tile's first value sit out the first one).
// NT: streaming stores.  -> the number of store instructions every lane of the wave surely issued (counted waits).
#ifndef MH_ALIGN_U64
#define MH_ALIGN_U64 16 // values (a 128-byte line of the stream): lane 0's piece of every store starts a line
#endif
template <bool NT>
__device__ __forceinline__ uint32_t multi_hash_copy_out(const uint64_t* __restrict__ tile, uint64_t* __restrict__ hashes, uint64_t o0,
                                                        uint32_t n_emit, uint32_t m, uint32_t inv_m, uint64_t kmul, uint32_t lane)
{
  const uint64_t v0 = o0 * m;
  const uint32_t vpar = (uint32_t)(v0 & (uint64_t)(MH_ALIGN_U64 - 1));
  const uint32_t n_vals = n_emit * m;
  const uint32_t span = vpar + n_vals;
  const uint32_t p_first = (vpar + 1u) >> 1, p_end = span >> 1; // pieces [p_first, p_end) hold two values
  uint64_t* const base = hashes + (v0 - vpar);
  const uint32_t step_e = 128u / m, step_j = 128u - step_e * m; // (uniform: scalar unit)
  // the piece's second value, sv1 = 2 lane + 1 - vpar = e1 m + j1 (floor division; negative in the lanes before the
  // tile's first value, whose first iteration is idle): divide sv1 + 128 and take one step back
  const uint32_t sv1p = 2u * lane + 129u - vpar;
  uint32_t e1 = __umulhi(sv1p, inv_m), j1 = sv1p - e1 * m;
  j1 += m - step_j;
  e1 -= step_e + 1u;
  if (j1 >= m) {
    j1 -= m;
    ++e1;
  }
  for (uint32_t pi = lane; pi < p_end; pi += 64u) {
    const bool wrap = j1 == 0u; // the first value is the last hash of the k-mer before
    const uint32_t j0 = wrap ? m - 1u : j1 - 1u;
    const uint32_t e0 = wrap ? e1 - 1u : e1;
    if (pi >= p_first) {
      const uint64_t ha = tile[e0], hb = tile[e1];
      const uint64_t ma = mix_hash(ha, (uint64_t)j0 ^ kmul), mb = mix_hash(hb, (uint64_t)j1 ^ kmul);
      const uint64_t oa = j0 == 0u ? ha : ma, ob = wrap ? hb : mb;
      const nt_v4u ov = {(uint32_t)oa, (uint32_t)(oa >> 32), (uint32_t)ob, (uint32_t)(ob >> 32)};
      if (NT) __builtin_nontemporal_store(ov, (nt_v4u*)(base + 2u * pi));
      else *(nt_v4u*)(base + 2u * pi) = ov;
    }
    j1 += step_j;
    e1 += step_e;
    if (j1 >= m) {
      j1 -= m;
      ++e1;
    }
  }
  if (n_vals != 0u) {
    if ((vpar & 1u) != 0u && lane == 0u) base[vpar] = tile[0]; // value 0 = h[0] of the first k-mer (the second half of its piece)
    if ((span & 1u) != 0u && lane == 1u) base[span - 1u] = mix_hash(tile[n_emit - 1u], (uint64_t)(m - 1u) ^ kmul);
  }
  return p_end >> 6;
}

// the reads are streamed once: non-temporal loads keep them from displacing the output lines being
// assembled in L2 (+1.9 % on the headline kernel, in-process A/B)
#ifndef KRG_LOAD_NT
#define KRG_LOAD_NT " nt"
#endif
#ifndef KRG_XCD_GROUPS
#define KRG_XCD_GROUPS 0 // experiment of round 4 (tile_range below): no effect, off
#endif
constexpr uint32_t KRG_ALIGN_U64 = 128; // the output tile is aligned to 1 KiB of the stream
constexpr uint32_t KRG_SLACK_U64 = 32;  // N-aware: room below the tile for a first run's recomputed windows (< C <= 31)
enum : int { SINK_NONE = 0, SINK_BLOOM_INSERT = 1, SINK_BLOOM_QUERY = 2, SINK_MINHASH = 3, SINK_MINHASH1 = 4 };
// (SINK_MINHASH1: the signature is the minimum canonical hash alone -- one register pair, no multiplies)
constexpr uint32_t KRG_SIG_MAX = 8;     // MinHash: signature entries one launch keeps in registers
constexpr uint32_t KRG_TILE_READS = 66; // reads a tile of 64 runs can touch, rounded up

struct KmerRunsGenArgs {
  const uint8_t* seqs;
  uint64_t* hashes;          // dense [read][window][m]  /  N-aware: compact [emitted k-mer][m]
  uint32_t* dirty;           // dense: set when a non-base is seen
  uint32_t* vecmap;          // dense, NTHIP_OUT_READ_SLOTS on fixed-length reads: see KmerRunsArgs::vecmap (NULL: the plain pass)
  const uint4* init_tab;     // global [ntab][256] {f.lo,f.hi,r.lo,r.hi}
  uint32_t* pos;             // N-aware, optional: position of every emitted k-mer in its read
  uint64_t* counts;          // count pass, optional (zeroed by the host): per-read emitted windows
  uint64_t* tile_counts;     // count pass out: valid windows per wave tile
  const uint64_t* tile_off;  // N-aware hash pass in: exclusive scan of tile_counts
  // N-aware hash pass over SOME tiles (round 4): the n_list tiles listed here, in any order -- the tiles that lost a
  // window, when kmer_runs_kernel has written all the others of the compact stream.  NULL: every tile
  const uint64_t* tile_list;
  uint64_t n_list;
  const unsigned long long* n_list_dev; // != NULL: the number of listed tiles is read HERE (the host launched without waiting for it)
  uint64_t n_reads;
  uint64_t n_runs;       // n_reads * rpr
  uint64_t n_wtiles;     // ceil(n_runs / 64)
  uint64_t total_bytes;  // (n_reads - 1) * stride + len
  uint32_t len, stride, k, m;
  uint32_t nwin;
  uint32_t C;            // windows per run
  uint32_t rpr;          // runs per read = ceil(nwin / C)
  uint32_t last_start;   // first window of a read's last run = nwin - C
  uint32_t last_dup;     // windows the last run recomputes = rpr * C - nwin
  uint32_t ntab;         // byte tables in LDS: 4 * NW (zero past ceil(k/4)); 2 for the Horner path
  uint32_t waves;        // waves per block
  uint32_t bits_dwords;  // per-wave bit-stream capacity
  uint32_t vbits_dwords; // per-wave validity-bit capacity (N-aware)
  uint32_t ptile_dwords; // per-wave position tile (N-aware with pos), else 0
  uint32_t tile_u64;     // per-wave tile capacity
  uint32_t inv_rpr;      // floor(65536 / rpr) + 1 (used when rpr <= 64)
  uint32_t tile_map;     // wave groups of the tile -> wave mapping (as in kmer_runs_kernel)
  uint64_t tab[16][2];
  uint64_t mult[KF_MAX_RUNTIME_M];
  // fused consumers (SINK != 0)
  uint32_t* bloom;           // filter as 32-bit words: bit p = bit (p & 31) of word p >> 5
  uint64_t n_bits;
  uint64_t bloom_magic;      // floor((2^64 - 1) / n_bits), 0 when n_bits is a power of two
  uint64_t* hits;            // query, optional: per-read k-mers found
  uint64_t* sink_totals;     // [0] += k-mers consumed, [1] += k-mers found (query)
  // MinHash: sig[r * m + i] = min over read r's k-mers of hashes()[i], for sig_first <= i < sig_first + sig_n
  // (sig_n <= KRG_SIG_MAX; preset to all ones by the host)
  uint64_t* sig;
  uint32_t sig_first, sig_n;
  // N-aware pass: what a k-mer's value is -- 0: the canonical hash (+ m-1 mixes), 1: the forward-strand hash,
  // 2: the reverse-strand hash (NtHash::get_forward_hash / get_reverse_hash; one value per k-mer, m must be 1)
  uint32_t value_sel;
  // NW == 0 (k beyond the position tables): 0 = the grouped first window, 1 = the prefix-scan form (first_window.hpp);
  // the scan keeps {U, V} at every word boundary of the slab in uw_dwords dwords of LDS per wave (4 per word)
  uint32_t fw_scan;
  uint32_t uw_dwords;
  // FH instantiations: the position tables hold the forward halves only (8 bytes per entry: half the LDS); the reverse
  // strand's terms are the forward terms of the window's reverse complement
  uint32_t fh;
  // packed input (PK instantiations; NTHIP_PACKED_INPUT): seqs is the 2-bit code stream of nthip_pack_reads, positions
  // are bases of it; invalid = its companion stream, one bit per base (1 = not a base), read by the N-aware passes only
  const uint16_t* invalid;
};

// s_waitcnt needs an immediate: wait until at most n (0..15) vector-memory operations are in flight.  A ladder, not a
// switch: the weakest wait is unconditional and every smaller count adds a stronger one, so EVERY path through this
// code executes an inline wait -- which is what the ISA lint can check (rule R3; hipcc lowers a switch to a tree of
// skips in which "no case taken" is a path of the flow graph).  The rungs above n are satisfied when they issue.
#define KRG_RUNG(c) if (n < c + 1u) asm volatile("s_waitcnt vmcnt(" #c ")" ::: "memory");
__device__ __forceinline__ void wait_vmcnt_upto15(uint32_t n)
{
  asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
  KRG_RUNG(14) KRG_RUNG(13) KRG_RUNG(12) KRG_RUNG(11) KRG_RUNG(10) KRG_RUNG(9) KRG_RUNG(8) KRG_RUNG(7)
  KRG_RUNG(6) KRG_RUNG(5) KRG_RUNG(4) KRG_RUNG(3) KRG_RUNG(2) KRG_RUNG(1) KRG_RUNG(0)
}
#undef KRG_RUNG

// 4 ASCII bytes: a byte of the result is non-zero <=> that byte is not a base (the test of pack4 alone)
__device__ __forceinline__ uint32_t non_base4(uint32_t w)
{
  const uint32_t t = (w >> 1) & 0x03030303u;
  uint32_t x = w | 0x20202020u;
  const uint32_t ubit = (x >> 4) & 0x01010101u;
  x = x & ~ubit;
  return x ^ __builtin_amdgcn_perm(0u, 0x67746361u, t);
}

// 4 ASCII bytes -> 4 x 2-bit codes (one byte) and a 4-bit mask of the non-bases
__device__ __forceinline__ uint32_t pack4v(uint32_t w, uint32_t& inv4)
{
  const uint32_t t = (w >> 1) & 0x03030303u;
  uint32_t x = w | 0x20202020u;
  const uint32_t ubit = (x >> 4) & 0x01010101u;
  x = x & ~ubit;
  const uint32_t canon = __builtin_amdgcn_perm(0u, 0x67746361u, t);
  const uint32_t d = x ^ canon;                                   // byte != 0 <=> not a base
  const uint32_t nz = (((d & 0x7f7f7f7fu) + 0x7f7f7f7fu) | d) & 0x80808080u;
  inv4 = __builtin_amdgcn_udot4(nz >> 7, 0x08040201u, 0u, false); // gather the four flags
  return __builtin_amdgcn_udot4(t, 0x40100401u, 0u, false);
}

// Which of the (up to 32) windows starting at bases b0, b0+1, ... hold a non-base.
// vw: validity bit stream (1 = not a base); reads 128 bits from the dword of b0 on.
// OR over k <= 64 consecutive bits by doubling, on a 96-bit register pair.
__device__ __forceinline__ uint32_t windows_with_non_base(const uint32_t* vw, uint32_t b0, uint32_t k)
{
  const uint32_t dw = b0 >> 5, sh = b0 & 31u;
  const uint32_t x0 = vw[dw], x1 = vw[dw + 1], x2 = vw[dw + 2], x3 = vw[dw + 3];
  uint64_t lo = ((uint64_t)funnel(x2, x1, sh) << 32) | funnel(x1, x0, sh);
  uint64_t hi = funnel(x3, x2, sh); // bits 64..95
  auto fold = [&](uint32_t s) {     // bit j |= bit j + s, 1 <= s <= 32
    lo |= (lo >> s) | (hi << (64u - s));
    hi |= hi >> s;
  };
  uint32_t span = 1;
  while (2u * span <= k) {
    fold(span);
    span *= 2u;
  }
  if (k > span) fold(k - span);
  return (uint32_t)lo; // bit j exact for j + k - 1 <= 95
}

// The same for ANY k: walk the non-bases of [b0, b0 + C + k - 1) and knock out the windows they hit.
__device__ __forceinline__ uint32_t windows_with_non_base_long(const uint32_t* vw, uint32_t b0, uint32_t k,
                                                               uint32_t C)
{
  uint32_t inv = 0;
  const uint32_t end = b0 + C + k - 1u; // one past the last base a window of the run touches
  uint32_t w = b0 >> 5;
  uint32_t word = vw[w] & (~0u << (b0 & 31u));
  for (;;) {
    while (word) {
      const uint32_t pos = (w << 5) + (uint32_t)__builtin_ctz(word);
      word &= word - 1u;
      if (pos >= end) return inv;
      // windows j with b0 + j <= pos < b0 + j + k
      const int32_t hi = (int32_t)(pos - b0), lo = hi - (int32_t)k + 1;
      const int32_t lo_c = lo > 0 ? lo : 0, hi_c = hi < (int32_t)C - 1 ? hi : (int32_t)C - 1;
      if (lo_c <= hi_c) inv |= ((2u << hi_c) - 1u) & ~((1u << lo_c) - 1u);
    }
    ++w;
    if ((w << 5) >= end) break;
    word = vw[w];
  }
  return inv;
}

// First window of a run for ANY k (the NW == 0 instantiations and every other kernel's k > 64 path): the grouped form of
// first_window.hpp -- 16 bases per step from the k-independent fw tables (get_fw_tab; kmer_ntab(k) = 5 tables' worth of
// LDS), constant rotates by 16 between the steps.  Same values as base_forward_hash / base_reverse_hash,
// src/kmer.cpp:43-73,123-152.  (Round 2 walked the window 4 bases at a time, both strands apart: 2 x k/4 dependent steps.)
__device__ __forceinline__ void any_k_first_window(const uint32_t* bits, const uint4* tab, uint32_t b0, uint32_t k,
                                                   uint32_t& f_lo, uint32_t& f_hi, uint32_t& r_lo, uint32_t& r_hi)
{
  grouped_first_window(bits, tab, b0, k, k % 31u, k % 33u, f_lo, f_hi, r_lo, r_hi);
}

// inclusive XOR scan over the 64 lanes on the DPP network: row_shr 1, 2, 4, 8 inside each row of 16 lanes, row_bcast15 /
// row_bcast31 carry the row totals across (the prefix sums of kmer_reads_kernel.hpp with ^ for +)
__device__ __forceinline__ uint32_t wave_incl_xor32(uint32_t v)
{
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
  v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
  return v;
}

// ---- geometry shared by the three kernels (wave-uniform integers only) ---------------
struct RunShape {
  uint32_t C, rpr, inv_rpr, last_start, last_dup, stride, nwin, k;
};
// gl = run index counted from run 0 of a tile's first read, gl < 64 + rpr:
// read (relative), first window of the run inside it, and whether it is a read's last run
__device__ __forceinline__ void run_split(const RunShape& s, uint32_t gl, uint32_t& lr, uint32_t& w0, bool& last)
{
  lr = s.rpr > 64u ? (gl >= s.rpr ? 1u : 0u) : (gl * s.inv_rpr) >> 16;
  const uint32_t q = gl - lr * s.rpr;
  last = q == s.rpr - 1u;
  w0 = last ? s.last_start : q * s.C;
}
struct TileGeo {
  uint64_t byte0;   // offset of the first 16-byte vector from seqs (wraps below 0 by < 16)
  uint64_t out0;    // dense stream index of the tile's first k-mer
  uint32_t shift, slab_bytes, n_vec, runs_here;
  uint32_t n_kmers; // k-mers of the dense tile
  uint32_t w_first; // first window (inside its read) of the tile's first run
  uint32_t edge;    // the vectors of the slab reach outside the caller's buffer
};
// the tile whose first run is run rm of read rf (g0 = its global run index)
__device__ __forceinline__ TileGeo tile_geo(const RunShape& s, uint64_t seqs_addr, uint64_t n_runs,
                                            uint64_t total_bytes, uint64_t g0, uint64_t rf, uint32_t rm)
{
  TileGeo g;
  const uint64_t runs_left = n_runs - g0;
  g.runs_here = runs_left < 64u ? (uint32_t)runs_left : 64u;
  uint32_t lre, we;
  bool laste;
  run_split(s, rm + g.runs_here - 1u, lre, we, laste);
  const uint32_t w_first = rm == s.rpr - 1u ? s.last_start : rm * s.C;
  const uint64_t start = rf * s.stride + w_first;
  g.w_first = w_first;
  g.slab_bytes = lre * s.stride + we + s.C + s.k - 1u - w_first;
  g.n_kmers = lre * s.nwin + we + s.C - w_first;
  g.out0 = rf * s.nwin + w_first;
  g.shift = (uint32_t)((seqs_addr + start) & 15u);
  g.byte0 = start - g.shift;
  g.n_vec = (g.shift + g.slab_bytes + 15u) >> 4;
  g.edge = (start < g.shift || g.byte0 + ((uint64_t)g.n_vec << 4) > total_bytes) ? 1u : 0u;
  return g;
}
// the waves of the grid are split into tile_map groups of consecutive waves; every group owns one
// contiguous range of tiles and its waves interleave inside it (see kmer_runs_kernel.hpp)
template <bool XCD = true>
__device__ __forceinline__ void tile_range(uint32_t tile_map, uint32_t waves, uint32_t wave, uint64_t n_wtiles,
                                           uint64_t& wt, uint64_t& wstride, uint64_t& wt_end)
{
  const uint64_t n_waves_total = (uint64_t)gridDim.x * waves;
  // Round 4, -DKRG_XCD_GROUPS=1 (negative result, off): the blocks of a group on ONE XCD.  Block b runs on XCD b % 8
  // (observed, MI355X_MICROARCH.md) and every XCD has its own L2; a tile that does not start on a line of the stream (every
  // tile of the N-aware pass behind the first skipped k-mer, every shape whose reads' windows are no multiple of 16) shares
  // its first and last line with its neighbours -- the next waves of its GROUP.  The idea: on one XCD the two halves of a
  // line would meet in one L2.  In-process A/B twice over (tools/xcd_ab.sh): N-aware pass 5.89 / 5.85 against 5.79 / 5.90 ms,
  // 151 / 101 / 250 bp dense within 0.5 % either way -- whatever the partial lines cost, it is not the L2 they go through.
  uint32_t vb = blockIdx.x;
#if KRG_XCD_GROUPS
  if (XCD && (gridDim.x & 7u) == 0u) vb = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
#endif
  const uint64_t gw = (uint64_t)vb * waves + wave;
  uint64_t groups = tile_map ? tile_map : 1u;
  if (groups > n_waves_total) groups = n_waves_total;
  const uint64_t wpg = n_waves_total / groups;
  uint64_t g = gw / wpg;
  if (g >= groups) g = groups - 1;
  const uint64_t w_in_g = gw - g * wpg;
  const uint64_t g_waves = g == groups - 1 ? n_waves_total - g * wpg : wpg;
  const uint64_t per = (n_wtiles + groups - 1) / groups;
  const uint64_t t0 = g * per;
  wt = t0 + w_in_g;
  wstride = g_waves;
  wt_end = t0 + per < n_wtiles ? t0 + per : n_wtiles;
}

// NW: window words, k <= 16*NW (0: any k, the k-independent first window); DT: every slab is <= 1280 bytes (tail = one
// dword per lane);
// NA: N-aware hash pass (compact output at a.tile_off) instead of the dense optimistic pass;
// SINK (needs NA): consume the tile's hashes instead of writing them out;
// PK: packed input -- a.seqs is the 2-bit code stream (16 bases per dword, the format of the LDS bit stream: a slab is
//     staged with one dword load per 16 bases), a.invalid the validity stream the N-aware pass reads instead of judging bytes
template <int NW, bool DT, bool NA, int SINK = SINK_NONE, bool PK = false, bool FH = false>
__global__ __launch_bounds__(KR_MAX_THREADS) void kmer_runs_gen_kernel(const KmerRunsGenArgs a)
{
  static_assert(!PK || SINK == SINK_NONE, "packed input: the hash-stream passes");
  // FH (round 3; k = 49 ... 64, dense pass): R(window) = F(reverse complement of the window), so the first window needs
  // the forward halves of the position tables only -- 32 KiB instead of 64 at k = 64, i.e. 12 waves per CU instead of 8
  // for the reference's own benchmark shape -- at the price of 2 x 16 lookups of 8 bytes instead of 16 of 16 and ~20
  // instructions that reverse-complement the window's four words
  static_assert(!FH || (NW >= 1 && !NA && !PK), "forward-half tables: the dense pass, k within the position tables");
  const uint64_t seqs_addr = PK ? 0ull : (uint64_t)a.seqs; // (packed: positions are bases of a 16-byte aligned stream)
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
#ifdef KRG_FORCE_C // experiment: what a compile-time run length is worth
  const uint32_t k = a.k, m = a.m, C = KRG_FORCE_C, ntab = a.ntab, rpr = a.rpr;
#else
  const uint32_t k = a.k, m = a.m, C = a.C, ntab = a.ntab, rpr = a.rpr;
#endif
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
#ifndef KRG_UNIFORM_WAVE
#define KRG_UNIFORM_WAVE 1
#endif
#if KRG_UNIFORM_WAVE
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6); // uniform: tile geometry runs on the scalar unit
#else
  const uint32_t wave = tid >> 6;
#endif
  const RunShape shape = {C, rpr, a.inv_rpr, a.last_start, a.last_dup, a.stride, a.nwin, k};
  const uint32_t inv_m = 0xFFFFFFFFu / m + 1u; // v / m == umulhi(v, inv_m) for v < 2^29
  const uint64_t kmul = (uint64_t)k * MULTISEED; // h[i] = mix(h[0] * (i ^ k*MULTISEED)), any m (src/internal.hpp:104-118)

  // LDS: init tables | pair table | multipliers | per wave {tile, [pos tile], bits, [validity bits]}
  uint4* itab = (uint4*)lds_dyn;
  uint4* ptab = FH ? (uint4*)((uint2*)lds_dyn + ntab * 256u) : itab + ntab * 256u;
  uint64_t* mults = (uint64_t*)(ptab + 16);
  const uint32_t per_wave = a.tile_u64 * 2u + a.ptile_dwords + a.bits_dwords + a.vbits_dwords + a.uw_dwords;
  uint32_t* wave_base = (uint32_t*)(mults + KF_MAX_RUNTIME_M) + wave * per_wave;
  uint64_t* tile = (uint64_t*)wave_base + (NA ? KRG_SLACK_U64 : 0u);
  uint32_t* ptile = wave_base + a.tile_u64 * 2u + KRG_SLACK_U64;
  uint32_t* bits = wave_base + a.tile_u64 * 2u + a.ptile_dwords;
  uint16_t* vbits = (uint16_t*)(bits + a.bits_dwords);
  uint4* uw = (uint4*)(bits + a.bits_dwords + a.vbits_dwords); // scan form: {U, V} at the slab's word boundaries

  if constexpr (FH) {
    for (uint32_t i = tid; i < ntab * 256u; i += blockDim.x) {
      const uint4 e = a.init_tab[i];
      ((uint2*)lds_dyn)[i] = make_uint2(e.x, e.y);
    }
  } else {
    for (uint32_t i = tid; i < ntab * 256u; i += blockDim.x) itab[i] = a.init_tab[i];
  }
  if (tid < 16)
    ptab[tid] = make_uint4((uint32_t)a.tab[tid][0], (uint32_t)(a.tab[tid][0] >> 32),
                           (uint32_t)a.tab[tid][1], (uint32_t)(a.tab[tid][1] >> 32));
  if (tid < KF_MAX_RUNTIME_M) mults[tid] = a.mult[tid];
  __syncthreads(); // the only block-wide barrier

  uint32_t bad = 0;
  // (listed: wt counts the entries of a.tile_list, the tile itself is tile_at(wt))
  const bool listed = NA && SINK == SINK_NONE && !PK && a.tile_list != nullptr;
  auto tile_at = [&](uint64_t i) -> uint64_t { return listed ? a.tile_list[i] : i; };
  uint64_t wt, wstride, wt_end;
  const uint64_t n_listed = listed ? (a.n_list_dev ? (uint64_t)*a.n_list_dev : a.n_list) : 0;
  tile_range<SINK == SINK_NONE>(a.tile_map, a.waves, wave, listed ? n_listed : a.n_wtiles, wt, wstride, wt_end); // (the consumers write no stream)
  uint64_t cur_tile = wt < wt_end ? tile_at(wt) : 0u;
  uint64_t r_first = (cur_tile * 64u) / rpr;
  uint32_t rem0 = (uint32_t)(cur_tile * 64u - r_first * rpr);
  const uint64_t step_q = (wstride * 64u) / rpr;
  const uint32_t step_r = (uint32_t)(wstride * 64u - step_q * rpr);

  // Dense pass: bytes of the batch next to the slab inside its first / last vector are
  // judged too (a non-base there makes the batch dirty anyway); bytes outside the
  // caller's buffer are not: they exist only in the slabs flagged `edge`.
  // N-aware pass: no judging, a validity bit per base instead (bits outside the slab
  // never reach an emitted window).
  auto pack_vec = [&](const TileGeo& sl, uint32_t i, const uint4 v) {
    if constexpr (NA) {
      // validity bits are only built for the slabs that hold a non-base (validity_bits() below): `bad`
      // collects the test for the tile being staged (bytes next to the slab included: harmless)
      uint32_t b = 0;
      bits[i] = pack16(v, b);
      bad |= b;
    } else {
      uint32_t b = 0;
      const uint32_t p = pack16(v, b);
      if (sl.edge) {
        const int32_t lo_cut = (int32_t)sl.shift - (int32_t)(i << 4);
        const int32_t hi_cut = (int32_t)(sl.shift + sl.slab_bytes) - (int32_t)(i << 4);
        if (lo_cut > 0 || hi_cut < 16) {
          uint32_t bx[4] = {0, 0, 0, 0};
          (void)pack4(v.x, bx[0]);
          (void)pack4(v.y, bx[1]);
          (void)pack4(v.z, bx[2]);
          (void)pack4(v.w, bx[3]);
          b = 0;
#pragma unroll
          for (int q = 0; q < 16; ++q)
            if (q >= lo_cut && q < hi_cut) b |= (bx[q >> 2] >> ((q & 3) * 8)) & 0xFFu;
        }
      }
      if (b != 0u && a.vecmap != nullptr) { // (rare) read slots: remember the vector, keep going
        const uint64_t av = ((seqs_addr + sl.byte0) >> 4) - (seqs_addr >> 4) + i;
        atomicOr(&a.vecmap[av >> 5], 1u << (av & 31u));
        b = 0;
      }
      bad |= b;
      bits[i] = p;
    }
  };
  // the tail of a short slab: dword 256 + lane (4 bases -> one byte of the stream), all lanes call
  auto pack_tail = [&](const TileGeo& sl, const uint32_t wv) {
    const uint32_t n_dw = (sl.shift + sl.slab_bytes + 3u) >> 2;
    const bool mine = 256u + lane < n_dw;
    if constexpr (NA) {
      if (mine) {
        uint32_t b = 0;
        ((uint8_t*)bits)[256u + lane] = (uint8_t)pack4(wv, b);
        bad |= b;
      }
    } else if (mine) {
      uint32_t b = 0;
      const uint32_t p = pack4(wv, b);
      if (sl.edge) {
        const int32_t hi_cut = (int32_t)(sl.shift + sl.slab_bytes) - (int32_t)(1024u + (lane << 2));
        uint32_t keep = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q < hi_cut) keep |= 0xFFu << (q * 8);
        b &= keep;
      }
      if (b != 0u && a.vecmap != nullptr) {
        const uint64_t av = ((seqs_addr + sl.byte0) >> 4) - (seqs_addr >> 4) + 64u + (lane >> 2);
        atomicOr(&a.vecmap[av >> 5], 1u << (av & 31u));
        b = 0;
      }
      bad |= b;
      ((uint8_t*)bits)[256u + lane] = (uint8_t)p;
    }
  };
  auto stage = [&](const TileGeo& sl, uint32_t first) {
    if constexpr (PK) {
      const uint32_t* codes = (const uint32_t*)a.seqs + (sl.byte0 >> 4); // dword i = bases [byte0 + 16 i, ...)
      for (uint32_t i = first + lane; i < sl.n_vec; i += 64u) bits[i] = codes[i];
    } else {
      for (uint32_t i = first + lane; i < sl.n_vec; i += 64u)
        pack_vec(sl, i, *(const uint4*)(a.seqs + sl.byte0 + ((uint64_t)i << 4)));
    }
    if (lane < (uint32_t)NW + 5u) bits[sl.n_vec + lane] = 0; // funnels read a little ahead
  };
  // packed N-aware pass: the validity bits of the slab come from the companion stream (16 per vector, as vbits holds
  // them); `bad` says whether the tile holds a non-base at all
  auto packed_validity = [&](const TileGeo& sl) {
    const uint16_t* inv = a.invalid + (sl.byte0 >> 4);
    for (uint32_t i = lane; i < sl.n_vec; i += 64u) {
      const uint16_t w = inv[i];
      vbits[i] = w;
      bad |= w;
    }
  };
  // N-aware pass, slab with a non-base (rare): one validity bit per base of the slab, 16 per vector, from
  // the bytes themselves (L2-hot: they were staged a moment ago)
  auto validity_bits = [&](const TileGeo& sl) {
    if constexpr (PK) return; // (packed_validity has already written them)
    for (uint32_t i = lane; i < sl.n_vec; i += 64u) {
      const uint4 v = *(const uint4*)(a.seqs + sl.byte0 + ((uint64_t)i << 4));
      uint32_t i0, i1, i2, i3;
      (void)pack4v(v.x, i0);
      (void)pack4v(v.y, i1);
      (void)pack4v(v.z, i2);
      (void)pack4v(v.w, i3);
      vbits[i] = (uint16_t)(i0 | (i1 << 4) | (i2 << 8) | (i3 << 12));
    }
  };
  auto lds_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  };

  TileGeo cur;
  cur.byte0 = cur.out0 = 0;
  cur.shift = cur.slab_bytes = cur.n_vec = cur.runs_here = cur.n_kmers = cur.w_first = 0;
  cur.edge = 1u;
  static_assert(SINK == SINK_NONE || NA, "consumers run on the N-aware pass");
  constexpr bool MINH = SINK == SINK_MINHASH || SINK == SINK_MINHASH1;
  constexpr uint32_t N_MN = SINK == SINK_MINHASH ? KRG_SIG_MAX : 1u;
  uint64_t cur_off = 0; // N-aware: compact stream index of the tile's first emitted k-mer
  bool cur_dirty = false; // N-aware: the staged slab holds a non-base (validity bits are built)
  uint64_t sum_emit = 0, sum_hits = 0; // consumers: k-mers consumed / found by this wave
  bool test_first = true;              // Bloom insert: look at the bit before the atomic (see below)
  uint32_t probe_wait = 0;
  if (wt < wt_end) {
    cur = tile_geo(shape, seqs_addr, a.n_runs, a.total_bytes, cur_tile * 64u, r_first, rem0);
    if constexpr (NA && SINK == SINK_NONE) cur_off = a.tile_off[cur_tile];
    stage(cur, 0u);
    if constexpr (NA && PK) packed_validity(cur);
    if constexpr (NA) {
      cur_dirty = __ballot(bad != 0u) != 0ull;
      if (cur_dirty) validity_bits(cur);
    }
  }
  for (; wt < wt_end; wt += wstride) {
    lds_sync();
    const uint32_t shift = cur.shift, runs_here = cur.runs_here;
    const uint32_t my_rem0 = rem0;
    const uint64_t my_rf = r_first;
    // ---- issue the loads of the NEXT tile's slab ---------------------------------
    const uint64_t nwt = wt + wstride;
    const bool have_next = nwt < wt_end;
    const uint64_t nxt_tile = listed ? (have_next ? a.tile_list[nwt] : cur_tile) : nwt;
    if (listed) { // (a handful of tiles: a division each)
      r_first = (nxt_tile * 64u) / rpr;
      rem0 = (uint32_t)(nxt_tile * 64u - r_first * rpr);
    } else {
      r_first += step_q;
      rem0 += step_r;
      if (rem0 >= rpr) { rem0 -= rpr; r_first += 1; }
    }
    TileGeo nxt = cur;
    uint64_t nxt_off = cur_off;
    if (have_next) {
      nxt = tile_geo(shape, seqs_addr, a.n_runs, a.total_bytes, nxt_tile * 64u, r_first, rem0);
      if constexpr (NA && SINK == SINK_NONE) nxt_off = a.tile_off[nxt_tile];
    }
    v4u pv0, pv1;
    uint32_t pw;
    uint32_t dirty_seen;
    {
      const uint32_t i0 = lane < nxt.n_vec ? lane : 0u;
      const uint8_t* p0 = a.seqs + nxt.byte0 + ((uint64_t)i0 << 4);
      if constexpr (PK) {
        // dwords lane and 64 + lane of the slab's code stream (the rest of a longer slab: stage(cur, 128) below)
        const uint32_t* codes = (const uint32_t*)a.seqs + (nxt.byte0 >> 4);
        const uint32_t* q0 = codes + i0;
        const uint32_t* q1 = codes + (64u + lane < nxt.n_vec ? 64u + lane : 0u);
        asm volatile("global_load_dword %0, %2, off nt\n\tglobal_load_dword %1, %3, off nt"
                     : "=&v"(pw), "=&v"(dirty_seen)
                     : "v"(q0), "v"(q1)
                     : "memory");
        (void)p0;
      } else if constexpr (DT) {
        const uint32_t n_dw = (nxt.shift + nxt.slab_bytes + 3u) >> 2;
        const uint32_t j = 256u + lane < n_dw ? 256u + lane : 0u;
        const uint8_t* p1 = a.seqs + nxt.byte0 + ((uint64_t)j << 2);
        asm volatile("global_load_dword %2, %5, off sc1\n\tglobal_load_dwordx4 %0, %3, off" KRG_LOAD_NT "\n\t"
                     "global_load_dword %1, %4, off" KRG_LOAD_NT
                     : "=&v"(pv0), "=&v"(pw), "=&v"(dirty_seen)
                     : "v"(p0), "v"(p1), "v"(a.dirty)
                     : "memory");
      } else {
        const uint32_t i1 = lane + 64u < nxt.n_vec ? lane + 64u : 0u;
        const uint8_t* p1 = a.seqs + nxt.byte0 + ((uint64_t)i1 << 4);
        asm volatile("global_load_dword %2, %5, off sc1\n\tglobal_load_dwordx4 %0, %3, off" KRG_LOAD_NT "\n\t"
                     "global_load_dwordx4 %1, %4, off" KRG_LOAD_NT
                     : "=&v"(pv0), "=&v"(pv1), "=&v"(dirty_seen)
                     : "v"(p0), "v"(p1), "v"(a.dirty)
                     : "memory");
      }
    }

    // ---- this lane's run ----------------------------------------------------------
    // lanes past the end of the last tile redo the tile's first run (same values, same addresses)
    const bool live = lane < runs_here;
    uint32_t lr, w0;
    bool last_run;
    run_split(shape, live ? my_rem0 + lane : my_rem0, lr, w0, last_run);
    const uint32_t b0 = shift + lr * a.stride + w0 - cur.w_first; // first base of the first window

    // N-aware: which of the C windows are emitted, and where in the tile they go
    uint32_t valid = 0, lane_off = 0, n_emit = cur.n_kmers;
    bool all_valid = true;
    if constexpr (NA) {
      const uint32_t dup = live && last_run ? a.last_dup : 0u; // windows the run before already covers
      const uint32_t run_mask = live ? ((1u << C) - 1u) & ~((1u << dup) - 1u) : 0u;
      uint32_t with_non_base = 0; // a slab of bases only: every window is valid
      if (cur_dirty)
        with_non_base = k <= 64u ? windows_with_non_base((const uint32_t*)vbits, b0, k)
                                 : windows_with_non_base_long((const uint32_t*)vbits, b0, k, C);
      valid = ~with_non_base & run_mask;
      // (consumers always compact: slots 0 .. n_emit-1 of the tile, no alignment shift)
      // (MinHash folds every window of a clean tile, the recomputed ones too: those must be valid as well --
      // the run they repeat may sit in another wave's tile)
      all_valid = MINH ? __ballot(live && (with_non_base & ((1u << C) - 1u)) != 0u) == 0
                       : SINK == SINK_NONE && __ballot(valid != run_mask) == 0;
      if (all_valid) {
        // clean tile (almost all of them): the dense geometry, no scan.  Window j of a run goes to its dense
        // slot; a recomputed window lands on the slot the run before gives the same value -- only the tile's
        // FIRST run has its recomputed windows in another tile: they fall below slot 0 (KRG_SLACK_U64)
        const uint32_t d0 = my_rem0 == rpr - 1u ? a.last_dup : 0u;
        lane_off = lr * a.nwin + w0 - cur.w_first - d0;
        n_emit = cur.n_kmers - d0;
      } else {
        const uint32_t cnt = __builtin_popcount(valid);
        uint32_t incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t o = __shfl_up(incl, d, 64);
          if ((int)lane >= d) incl += o;
        }
        lane_off = incl - cnt;
        n_emit = __shfl(incl, 63, 64);
      }
    }
    // the tile is built shifted by the position of its first stream element inside a
    // 1 KiB block of the output (m == 1): every store instruction of the copy-out then
    // covers one aligned KiB, instead of every wave splitting cache lines with its neighbours
    const uint64_t out0 = NA ? cur_off : cur.out0;
    const uint32_t tpar = (m == 1u && SINK == SINK_NONE) ? (uint32_t)(out0 & (KRG_ALIGN_U64 - 1u)) : 0u;
    // (signed: a clean tile's first run may start up to KRG_SLACK_U64 slots below the tile)
    uint64_t* my_row = tile + (int32_t)(tpar + (NA ? lane_off : lr * a.nwin + w0 - cur.w_first));
    uint32_t* my_pos = ptile + (int32_t)lane_off;
    const bool want_pos = NA && (SINK == SINK_BLOOM_QUERY || a.pos != nullptr); // query: the k-mer's read
    uint32_t slot = 0; // N-aware, tile with non-bases: next free slot of this lane
    uint64_t mn[N_MN]; // MinHash: this run's minima
#pragma unroll
    for (uint32_t i = 0; i < N_MN; ++i) mn[i] = ~0ull;

    const uint32_t d0 = b0 >> 4, sh0 = (b0 & 15u) << 1;
    uint32_t f_lo = 0, f_hi = 0, r_lo = 0, r_hi = 0;
    if constexpr (NW == 0) {
      if (a.fw_scan) {
        // Prefix over the whole slab, k-independent (first_window.hpp): lane l takes Wl consecutive words of the 2-bit
        // stream, leaves the running XOR of their absolute-frame terms at each word's boundary, the lane totals are
        // XOR-scanned over the wave and folded back in; then a window is a difference of two prefixes.
        const uint32_t n_words = cur.n_vec + 1u; // boundaries 0 .. n_vec (the word past the slab is zeroed slack)
        const uint32_t Wl = (n_words + 63u) >> 6;
        uint4 run = make_uint4(0, 0, 0, 0);
        for (uint32_t w = 0; w < Wl; ++w) {
          const uint32_t wi = lane * Wl + w;
          if (wi < n_words) {
            uw[wi] = run;
            const uint4 e = fw_scan_word(itab, bits[wi], wi);
            run.x ^= e.x; run.y ^= e.y; run.z ^= e.z; run.w ^= e.w;
          }
        }
        const uint4 excl = make_uint4(wave_incl_xor32(run.x) ^ run.x, wave_incl_xor32(run.y) ^ run.y,
                                      wave_incl_xor32(run.z) ^ run.z, wave_incl_xor32(run.w) ^ run.w);
        for (uint32_t w = 0; w < Wl; ++w) {
          const uint32_t wi = lane * Wl + w;
          if (wi < n_words) {
            uint4 u = uw[wi];
            u.x ^= excl.x; u.y ^= excl.y; u.z ^= excl.z; u.w ^= excl.w;
            uw[wi] = u;
          }
        }
        lds_sync();
        scan_first_window(bits, itab, uw, b0, k, k % 1023u, k % 31u, k % 33u, f_lo, f_hi, r_lo, r_hi);
      } else {
        any_k_first_window(bits, itab, b0, k, f_lo, f_hi, r_lo, r_hi);
      }
    } else if constexpr (FH) {
      const uint2* const ftab = (const uint2*)lds_dyn;
      uint32_t w[NW];
      uint32_t lo = bits[d0];
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        const uint32_t hi = bits[d0 + i + 1];
        w[i] = funnel(hi, lo, sh0);
        lo = hi;
      }
      // the window's reverse complement: words in reverse order, each with its 16 bases reversed (bit reverse, then the two
      // bits of every base swapped back) and complemented (code ^ 2: A <-> T, C <-> G), the 16 NW - k bases of padding
      // shifted out (fewer than 16: k > 16 (NW - 1))
      uint32_t rw[NW];
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        const uint32_t r = __builtin_bitreverse32(w[NW - 1 - i]);
        rw[i] = (((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1)) ^ 0xAAAAAAAAu;
      }
      const uint32_t sft = 2u * (16u * (uint32_t)NW - k);
      uint32_t rs[NW];
#pragma unroll
      for (int i = 0; i + 1 < NW; ++i) rs[i] = funnel(rw[i + 1], rw[i], sft);
      rs[NW - 1] = rw[NW - 1] >> sft;
      uint2 ef[4 * NW], er[4 * NW];
#pragma unroll
      for (int jt = 0; jt < 4 * NW; ++jt) {
        ef[jt] = ftab[(uint32_t)jt * 256u + ((w[jt >> 2] >> ((jt & 3) * 8)) & 0xFFu)];
        er[jt] = ftab[(uint32_t)jt * 256u + ((rs[jt >> 2] >> ((jt & 3) * 8)) & 0xFFu)];
      }
      f_lo = ef[0].x ^ ef[1].x; f_hi = ef[0].y ^ ef[1].y; r_lo = er[0].x ^ er[1].x; r_hi = er[0].y ^ er[1].y;
#pragma unroll
      for (int jt = 2; jt < 4 * NW; jt += 2) {
        f_lo = __builtin_amdgcn_bitop3_b32(f_lo, ef[jt].x, ef[jt + 1].x, 0x96);
        f_hi = __builtin_amdgcn_bitop3_b32(f_hi, ef[jt].y, ef[jt + 1].y, 0x96);
        r_lo = __builtin_amdgcn_bitop3_b32(r_lo, er[jt].x, er[jt + 1].x, 0x96);
        r_hi = __builtin_amdgcn_bitop3_b32(r_hi, er[jt].y, er[jt + 1].y, 0x96);
      }
    } else {
      uint32_t w[NW];
      uint32_t lo = bits[d0];
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        const uint32_t hi = bits[d0 + i + 1];
        w[i] = funnel(hi, lo, sh0);
        lo = hi;
      }
      // ntab == 4 * NW: the host pads the tables with zero ones past ceil(k/4), so no lookup sits behind a
      // (uniform) branch and the compiler keeps them all in flight -- with the branch it waits for each
      // ds_read_b128 before issuing the next
      uint4 e[4 * NW];
#pragma unroll
      for (int jt = 0; jt < 4 * NW; ++jt) e[jt] = itab[(uint32_t)jt * 256u + ((w[jt >> 2] >> ((jt & 3) * 8)) & 0xFFu)];
      f_lo = e[0].x ^ e[1].x; f_hi = e[0].y ^ e[1].y; r_lo = e[0].z ^ e[1].z; r_hi = e[0].w ^ e[1].w;
#pragma unroll
      for (int jt = 2; jt < 4 * NW; jt += 2) { // a ^ b ^ c: one v_bitop3_b32
        f_lo = __builtin_amdgcn_bitop3_b32(f_lo, e[jt].x, e[jt + 1].x, 0x96);
        f_hi = __builtin_amdgcn_bitop3_b32(f_hi, e[jt].y, e[jt + 1].y, 0x96);
        r_lo = __builtin_amdgcn_bitop3_b32(r_lo, e[jt].z, e[jt + 1].z, 0x96);
        r_hi = __builtin_amdgcn_bitop3_b32(r_hi, e[jt].w, e[jt + 1].w, 0x96);
      }
    }
    // window j of the run has just been hashed
    auto emit = [&](uint32_t j, auto clean_tag) {
      uint64_t h = canon_pair(f_lo, f_hi, r_lo, r_hi);
      if constexpr (MINH) {
        // the consumer lives in registers: nothing of the k-mer is written anywhere.  (Clean tile: a
        // recomputed window repeats a value the minimum has already seen.)
        const bool on = decltype(clean_tag)::value || ((valid >> j) & 1u);
        if constexpr (SINK == SINK_MINHASH1) {
          mn[0] = on && h < mn[0] ? h : mn[0];
        } else {
#pragma unroll
          for (uint32_t i = 0; i < N_MN; ++i)
            if (i < a.sig_n) {
              const uint32_t hi_idx = a.sig_first + i;
              const uint64_t hv = hi_idx == 0u ? h : mix_hash(h, ((uint64_t)hi_idx ^ kmul));
              mn[i] = on && hv < mn[i] ? hv : mn[i];
            }
        }
        return;
      }
      if (NA && SINK == SINK_NONE && a.value_sel != 0u)
        h = a.value_sel == 1u ? (((uint64_t)f_hi << 32) | f_lo) : (((uint64_t)r_hi << 32) | r_lo);
      if constexpr (decltype(clean_tag)::value) {
        my_row[j] = h;
        if (want_pos) my_pos[j] = w0 + j;
      } else if ((valid >> j) & 1u) {
        my_row[slot] = h;
        if (want_pos) my_pos[slot] = SINK == SINK_BLOOM_QUERY ? lr : w0 + j;
        ++slot;
      }
    };
    // remaining C-1 windows: roll.  step t: in = base b0+k-1+t, out = base b0+t-1
    auto hash_run = [&](auto clean_tag) {
      emit(0u, clean_tag);
      const uint32_t bi = b0 + k;
      const uint32_t di = bi >> 4, shi = (bi & 15u) << 1;
      for (uint32_t jw = 0; jw * 16u + 1u < C; ++jw) {
        const uint32_t w_in = funnel(bits[di + jw + 1], bits[di + jw], shi);
        const uint32_t w_out = funnel(bits[d0 + jw + 1], bits[d0 + jw], sh0);
        const uint32_t u = ((w_in & 0x33333333u) << 2) | (w_out & 0x33333333u);
        const uint32_t v = (w_in & 0xCCCCCCCCu) | ((w_out >> 2) & 0x33333333u);
        auto lookup = [&](uint32_t i) -> uint4 {
          const uint32_t src = (i & 1u) ? v : u;
          const uint32_t off = ((src >> ((i >> 1) * 4u)) & 0xFu) << 4;
          return *(const uint4*)((const char*)ptab + off);
        };
        auto roll = [&](const uint4 term) {
          roll_step<!(NA && SINK == SINK_NONE && NW >= 3)>(f_lo, f_hi, r_lo, r_hi, term);
        };
        // table terms do not depend on the hash state: fetch a batch of them ahead of the
        // dependent chain so that their LDS latencies overlap
        auto batch = [&](uint32_t i0, auto n_tag) {
          constexpr uint32_t N = decltype(n_tag)::value;
          uint4 terms[N];
#pragma unroll
          for (uint32_t i = 0; i < N; ++i) terms[i] = lookup(i0 + i);
#pragma unroll
          for (uint32_t i = 0; i < N; ++i) {
            roll(terms[i]);
            emit(jw * 16u + i0 + i + 1u, clean_tag);
          }
        };
        const uint32_t left = C - 1u - jw * 16u;
        const uint32_t ns = left < 16u ? left : 16u;
        uint32_t i0 = 0;
        for (; i0 + 8u <= ns; i0 += 8u) batch(i0, std::integral_constant<uint32_t, 8u>{});
        switch (ns - i0) {
          case 1: batch(i0, std::integral_constant<uint32_t, 1u>{}); break;
          case 2: batch(i0, std::integral_constant<uint32_t, 2u>{}); break;
          case 3: batch(i0, std::integral_constant<uint32_t, 3u>{}); break;
          case 4: batch(i0, std::integral_constant<uint32_t, 4u>{}); break;
          case 5: batch(i0, std::integral_constant<uint32_t, 5u>{}); break;
          case 6: batch(i0, std::integral_constant<uint32_t, 6u>{}); break;
          case 7: batch(i0, std::integral_constant<uint32_t, 7u>{}); break;
          default: break;
        }
      }
    };
    if (!NA || all_valid) hash_run(std::true_type{});
    else hash_run(std::false_type{});

    // ---- copy the tile out: n_emit * m consecutive values of the hash stream -------
    lds_sync();
    NT_LINT_SELFTEST_TOUCH(dirty_seen);
    uint32_t n_counted; // store instructions surely issued after the prefetch loads
    if constexpr (SINK == SINK_BLOOM_INSERT) {
      // Every hash of the tile sets its bit.  Device-scope atomics retire at ~27 G/s on MI355X whatever
      // the scope or the filter size (tools/bench_micro/atomics.hip), random loads at 50-115 G/s: when
      // most bits are already set (real data: every k-mer arrives once per unit of coverage) it pays
      // to look before setting.  Each wave probes one tile in 16 that way and keeps doing it while at
      // least half of the bits it looks at are set.  (A stale "not set" only costs a redundant atomic.)
      if (test_first) {
        uint32_t seen = 0;
        constexpr uint32_t U = 4; // k-mers per lane in flight: the loop is latency-bound on the filter loads
        for (uint32_t e0 = 0; e0 < n_emit; e0 += 64u * U) {
          uint64_t h0[U];
          bool ok[U];
#pragma unroll
          for (uint32_t u = 0; u < U; ++u) {
            const uint32_t e = e0 + u * 64u + lane;
            ok[u] = e < n_emit;
            h0[u] = tile[ok[u] ? e : 0u];
          }
          for (uint32_t i = 0; i < m; ++i) {
            uint64_t p[U];
            uint32_t word[U];
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
              const uint64_t h = i == 0 ? h0[u] : mix_hash(h0[u], ((uint64_t)i ^ kmul));
              p[u] = mod_invariant(h, a.n_bits, a.bloom_magic);
              word[u] = a.bloom[p[u] >> 5];
            }
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
              const uint32_t bit = 1u << ((uint32_t)p[u] & 31u);
              if (ok[u]) {
                if (word[u] & bit) ++seen;
                else atomicOr(&a.bloom[p[u] >> 5], bit);
              }
            }
          }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) seen += __shfl_xor(seen, d, 64);
        test_first = 2u * seen >= n_emit * m;
        probe_wait = 15u;
        n_counted = 0;
      } else {
        for (uint32_t e = lane; e < n_emit; e += 64u) {
          const uint64_t h0 = tile[e];
          for (uint32_t i = 0; i < m; ++i) {
            const uint64_t h = i == 0 ? h0 : mix_hash(h0, ((uint64_t)i ^ kmul));
            const uint64_t p = mod_invariant(h, a.n_bits, a.bloom_magic);
            atomicOr(&a.bloom[p >> 5], 1u << ((uint32_t)p & 31u)); // no return value: counted like a store
          }
        }
        n_counted = (n_emit >> 6) * m;
        if (--probe_wait == 0u) test_first = true;
      }
      sum_emit += n_emit;
    } else if constexpr (SINK == SINK_BLOOM_QUERY) {
      // per-read hit counters of this tile (<= 65 reads) in the unused top of the tile
      uint32_t* rhits = (uint32_t*)(tile + 64u * C);
      rhits[lane] = 0;
      if (lane < 2u) rhits[64u + lane] = 0;
      lds_sync();
      constexpr uint32_t U = 4; // k-mers per lane in flight: the loop is latency-bound on the filter loads
      for (uint32_t e0 = 0; e0 < n_emit; e0 += 64u * U) {
        uint64_t h0[U];
        uint32_t rel[U];
        bool hit[U];
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
          const uint32_t e = e0 + u * 64u + lane;
          hit[u] = e < n_emit;
          h0[u] = tile[hit[u] ? e : 0u];
          rel[u] = ptile[hit[u] ? e : 0u];
        }
        for (uint32_t i = 0; i < m; ++i) {
          uint64_t p[U];
          uint32_t word[U];
#pragma unroll
          for (uint32_t u = 0; u < U; ++u) {
            const uint64_t h = i == 0 ? h0[u] : mix_hash(h0[u], ((uint64_t)i ^ kmul));
            p[u] = mod_invariant(h, a.n_bits, a.bloom_magic);
            word[u] = a.bloom[p[u] >> 5];
          }
#pragma unroll
          for (uint32_t u = 0; u < U; ++u) hit[u] = hit[u] && ((word[u] >> ((uint32_t)p[u] & 31u)) & 1u);
        }
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
          const uint64_t hb = __ballot(hit[u]);
          if (hb == 0) continue;
          const bool in_tile = e0 + u * 64u + lane < n_emit;
          const uint32_t rel0 = __builtin_amdgcn_readfirstlane(rel[u]);
          if (__ballot(in_tile && rel[u] != rel0) == 0) { // one read (the usual case): one add
            if (lane == 0) rhits[rel0] += (uint32_t)__builtin_popcountll(hb);
          } else if (hit[u]) {
            atomicAdd(&rhits[rel[u]], 1u);
          }
        }
      }
      lds_sync();
      uint32_t mine = rhits[lane] + (lane == 0 ? rhits[64] : 0u);
      if (a.hits) {
        if (rhits[lane]) atomicAdd((unsigned long long*)&a.hits[my_rf + lane], (unsigned long long)rhits[lane]);
        if (lane == 0 && rhits[64]) atomicAdd((unsigned long long*)&a.hits[my_rf + 64u], (unsigned long long)rhits[64]);
      }
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d, 64);
      sum_hits += mine;
      sum_emit += n_emit;
      n_counted = 0;
    } else if constexpr (MINH) {
      // runs of one read sit in neighbouring lanes: fold them in LDS (the tile is not used by this
      // consumer), then one global atomic per read and signature entry the tile touched
      uint64_t* rmin = tile;
      const uint32_t n_slots = KRG_TILE_READS * a.sig_n;
      for (uint32_t i = lane; i < n_slots; i += 64u) rmin[i] = ~0ull;
      lds_sync();
#pragma unroll
      for (uint32_t i = 0; i < N_MN; ++i)
        if (i < a.sig_n && mn[i] != ~0ull) atomicMin((unsigned long long*)&rmin[lr * a.sig_n + i], (unsigned long long)mn[i]);
      lds_sync();
      for (uint32_t i = lane; i < n_slots; i += 64u) {
        const uint64_t v = rmin[i];
        if (v != ~0ull) {
          const uint32_t rr = i / a.sig_n, ii = i - rr * a.sig_n;
          atomicMin((unsigned long long*)&a.sig[(my_rf + rr) * m + a.sig_first + ii], (unsigned long long)v);
        }
      }
      sum_emit += n_emit;
      n_counted = 0;
    } else if (m == 1u) {
      const uint32_t span = tpar + n_emit;
      const uint32_t pieces = (span + 1u) >> 1;
      const uint32_t first = tpar >> 1;               // first piece that holds a value
      uint64_t* const base = a.hashes + (out0 - tpar); // 1 KiB aligned
      for (uint32_t pi = lane; pi < pieces; pi += 64u) {
        const uint4 dv = *(const uint4*)(tile + 2u * pi);
        const bool lo_ok = 2u * pi >= tpar && 2u * pi < span;
        const bool hi_ok = 2u * pi + 1u >= tpar && 2u * pi + 1u < span;
        // whole 16-byte pieces: streaming stores (in-process A/B: 151 bp +3 %, 250 bp +5 %, 100 bp +0.4 %; the
        // write-through policy of the headline kernel loses 9 % on 151 bp, where pieces do not fill their lines)
#if defined(KRG_ST_PLAIN)
        if (lo_ok && hi_ok) *(uint4*)(base + 2u * pi) = dv;
#else
        if (lo_ok && hi_ok) __builtin_nontemporal_store(*(const nt_v4u*)&dv, (nt_v4u*)(base + 2u * pi));
#endif
        else if (lo_ok) *(uint2*)(base + 2u * pi) = make_uint2(dv.x, dv.y);
        else if (hi_ok) *(uint2*)(base + 2u * pi + 1u) = make_uint2(dv.z, dv.w);
      }
      // iterations in which some lane surely stores a whole piece
      const uint32_t it_lo = (first + 64u) >> 6, it_hi = pieces >> 6;
      n_counted = it_hi > it_lo ? it_hi - it_lo : 0u;
    } else {
      // multi-hash expansion (extend_hashes, src/internal.hpp:104-118) fused into the
      // copy-out: stream value v is h[v % m] of k-mer v / m.  (Round 3, negative: one K-MER per lane -- m - 1 multiplies,
      // no division, the values through a wave-private staging area back into stream order -- lost 15 % on 100 bp /
      // k = 64 / m = 3 and 7 % at m = 2: three dependent LDS round trips per 64 k-mers against independent iterations here.)
#if defined(KRG_MH_PLAIN)
      n_counted = multi_hash_copy_out<false>(tile, a.hashes, out0, n_emit, m, inv_m, kmul, lane);
#else // streaming stores for the whole pieces, as in the m = 1 copy-out
      n_counted = multi_hash_copy_out<true>(tile, a.hashes, out0, n_emit, m, inv_m, kmul, lane);
#endif
    }
    if (SINK == SINK_NONE && want_pos)
      for (uint32_t e = lane; e < n_emit; e += 64u) a.pos[out0 + e] = ptile[e];
    lds_sync(); // tile and bits are free again

    // ---- consume the prefetched slab ------------------------------------------------
    // vmcnt retires in order: once at most n_counted operations are in flight, the
    // three loads issued before those stores have landed (never count a store that
    // might not have been issued: an iteration with all 64 lanes active always is)
    wait_vmcnt_upto15(n_counted < 15u ? n_counted : 15u);
    if constexpr (PK) asm volatile("; NTLINT_CONSUME %0 %1" : "+v"(pw), "+v"(dirty_seen)::"memory");
    else if constexpr (DT) asm volatile("; NTLINT_CONSUME %0 %1 %2" : "+v"(pv0), "+v"(pw), "+v"(dirty_seen)::"memory");
    else asm volatile("; NTLINT_CONSUME %0 %1 %2" : "+v"(pv0), "+v"(pv1), "+v"(dirty_seen)::"memory");
    if constexpr (PK) { // (dirty_seen holds the slab's second dword here: a packed batch is not judged while it is staged)
      if (have_next) {
        cur = nxt;
        cur_off = nxt_off;
        if (lane < cur.n_vec) bits[lane] = pw;
        if (64u + lane < cur.n_vec) bits[64u + lane] = dirty_seen;
        stage(cur, 128u);
        if constexpr (NA) {
          bad = 0;
          packed_validity(cur);
          cur_dirty = __ballot(bad != 0u) != 0ull;
        }
      }
      continue;
    }
    // dense pass: some wave already found a non-base byte -- the caller will redo the batch
    // on the N-aware path, so stop producing a dense stream nobody will read
    if (!NA && __builtin_amdgcn_readfirstlane(dirty_seen) != 0u) break;
    if (have_next) {
      cur = nxt;
      cur_off = nxt_off;
      cur_tile = nxt_tile;
      if constexpr (NA) bad = 0;
      if (lane < cur.n_vec) pack_vec(cur, lane, make_uint4(pv0.x, pv0.y, pv0.z, pv0.w));
      if constexpr (DT) {
        pack_tail(cur, pw);
        if (lane < (uint32_t)NW + 5u) bits[cur.n_vec + lane] = 0;
      } else {
        if (lane + 64u < cur.n_vec) pack_vec(cur, lane + 64u, make_uint4(pv1.x, pv1.y, pv1.z, pv1.w));
        stage(cur, 128u);
      }
      if constexpr (NA) {
        cur_dirty = __ballot(bad != 0u) != 0ull;
        if (cur_dirty) validity_bits(cur);
      }
      if (!NA && __ballot(bad != 0) != 0) { // publish at once so that every wave can stop early
        if (lane == 0) atomicOr(a.dirty, 1u);
        break;
      }
    }
  }
  if (!NA && __ballot(bad != 0) != 0 && lane == 0) atomicOr(a.dirty, 1u);
  if (SINK != SINK_NONE && lane == 0) {
    if (sum_emit) atomicAdd((unsigned long long*)&a.sink_totals[0], (unsigned long long)sum_emit);
    if (sum_hits) atomicAdd((unsigned long long*)&a.sink_totals[1], (unsigned long long)sum_hits);
  }
}

// the unfused consumer: set the bits of an already materialised hash stream
static __global__ __launch_bounds__(256) void stream_bloom_insert_kernel(const uint64_t* __restrict__ hashes, uint64_t n,
                                                                 uint32_t* __restrict__ bloom, uint64_t n_bits,
                                                                 uint64_t magic)
{
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t p = mod_invariant(hashes[i], n_bits, magic);
    atomicOr(&bloom[p >> 5], 1u << ((uint32_t)p & 31u));
  }
}

// Count pass of the N-aware path: validity bits only, no tables, no tile -- a few
// hundred bytes of LDS per wave, so the CU runs at full occupancy and the pass
// streams the reads at close to HBM read rate.  Same geometry and the same window
// masks as the hash pass.
// one tile of the count pass: a.tile_counts[wt] = the windows of tile wt without a non-base (and the reads' counts)
template <bool PK>
__device__ __forceinline__ void count_one_tile(const KmerRunsGenArgs& a, const RunShape& shape, uint64_t wt, uint16_t* vbits, uint32_t lane)
{
  const uint32_t k = a.k, C = a.C, rpr = a.rpr;
  const uint64_t g0 = wt * 64u;
  const uint64_t rf = g0 / rpr;
  const uint32_t rm = (uint32_t)(g0 - rf * rpr);
  const TileGeo g = tile_geo(shape, PK ? 0ull : (uint64_t)a.seqs, a.n_runs, a.total_bytes, g0, rf, rm);
  const bool live = lane < g.runs_here;
  uint32_t lr, w0;
  bool last_run;
  run_split(shape, live ? rm + lane : rm, lr, w0, last_run);
  const uint32_t dup = last_run ? a.last_dup : 0u;
  const uint32_t run_mask = live ? ((1u << C) - 1u) & ~((1u << dup) - 1u) : 0u;
  // almost every tile holds bases only: look for a non-base first (no validity bits, no LDS) ...
  uint32_t any_bad = 0;
  if constexpr (PK) {
    for (uint32_t i = lane; i < g.n_vec; i += 64u) any_bad |= a.invalid[(g.byte0 >> 4) + i];
  } else {
    for (uint32_t i = lane; i < g.n_vec; i += 64u) {
      const uint4 v = *(const uint4*)(a.seqs + g.byte0 + ((uint64_t)i << 4));
      any_bad |= non_base4(v.x) | non_base4(v.y) | non_base4(v.z) | non_base4(v.w);
    }
  }
  uint32_t valid = run_mask;
  if (__ballot(any_bad != 0u) != 0ull) {
    // ... and only then build the validity bits (the slab comes from L2 this time) and test every window
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = lane; i < g.n_vec; i += 64u) {
      if constexpr (PK) {
        vbits[i] = a.invalid[(g.byte0 >> 4) + i];
      } else {
        const uint4 v = *(const uint4*)(a.seqs + g.byte0 + ((uint64_t)i << 4));
        uint32_t i0, i1, i2, i3;
        (void)pack4v(v.x, i0);
        (void)pack4v(v.y, i1);
        (void)pack4v(v.z, i2);
        (void)pack4v(v.w, i3);
        vbits[i] = (uint16_t)(i0 | (i1 << 4) | (i2 << 8) | (i3 << 12));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    const uint32_t b0 = g.shift + lr * a.stride + w0 - g.w_first;
    valid = ~(k <= 64u ? windows_with_non_base((const uint32_t*)vbits, b0, k)
                       : windows_with_non_base_long((const uint32_t*)vbits, b0, k, C)) & run_mask;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
  }
  const uint32_t cnt = __builtin_popcount(valid);
  uint32_t sum = cnt;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d, 64);
  if (lane == 0) a.tile_counts[wt] = sum;
  if (a.counts) {
    if (rpr > 64u) {
      // long reads: a tile touches at most two of them -- one add each instead of one per lane
      uint32_t s0 = lr == 0u ? cnt : 0u;
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) s0 += __shfl_xor(s0, d, 64);
      if (lane == 0) {
        if (s0) atomicAdd((unsigned long long*)&a.counts[rf], (unsigned long long)s0);
        if (sum - s0) atomicAdd((unsigned long long*)&a.counts[rf + 1u], (unsigned long long)(sum - s0));
      }
    } else if (cnt) {
      atomicAdd((unsigned long long*)&a.counts[rf + lr], (unsigned long long)cnt);
    }
  }
}

template <bool PK = false> // PK: packed input -- the validity stream is read as it is, the bases not at all
static __global__ __launch_bounds__(KR_MAX_THREADS) void kmer_runs_count_kernel(const KmerRunsGenArgs a)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const RunShape shape = {a.C, a.rpr, a.inv_rpr, a.last_start, a.last_dup, a.stride, a.nwin, a.k};
  uint16_t* vbits = (uint16_t*)(lds_dyn + wave * a.vbits_dwords);
  const uint64_t n_waves_total = (uint64_t)gridDim.x * a.waves;
  for (uint64_t wt = (uint64_t)blockIdx.x * a.waves + wave; wt < a.n_wtiles; wt += n_waves_total) count_one_tile<PK>(a, shape, wt, vbits, lane);
}

// The count pass in two steps, for tiles that are whole reads lying back to back (stride == len, C | windows, runs per
// read | 64: the shapes of run_kmer_na_special).  Looking at every base of a tile on its own -- a wave, its 1.2 KB, a wait --
// runs at 3.7 TB/s (0.81 ms per 20 M x 150 bp); a plain grid-stride scan of the batch's 16-byte vectors at 5.1-5.9
// (tools/bench_micro/scan_rate.hip: 0.51-0.59 ms per 3 GB).  So: tiles_flag_kernel marks the tiles that hold a non-base (a
// bit per tile; a vector on the border of two tiles marks both), tiles_count_flagged_kernel gives the others their full
// count and counts the marked ones -- one in a hundred on real data -- exactly (count_one_tile).
static __global__ __launch_bounds__(256) void tiles_flag_kernel(const uint8_t* __restrict__ seqs, uint64_t total_bytes, uint32_t tile_bytes,
                                                                uint32_t* __restrict__ flags)
{
  const uint64_t base = (uint64_t)seqs;
  const uint32_t sh = (uint32_t)(base & 15u);
  const uint4* const v = (const uint4*)(base - sh);
  const uint64_t n_vec = (sh + total_bytes + 15u) >> 4;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  auto mark = [&](uint64_t at) {
    const uint64_t t = at / tile_bytes;
    atomicOr(&flags[t >> 5], 1u << (t & 31u));
  };
  constexpr int U = 2;
  for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n_vec; i0 += stride * U) {
    uint4 x[U];
    bool inside[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t i = i0 + (uint64_t)u * stride;
      inside[u] = i < n_vec && (i << 4) >= sh && (i << 4) + 16u <= sh + total_bytes; // (all 16 bytes are the buffer's)
      x[u] = inside[u] ? v[i] : make_uint4(0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t i = i0 + (uint64_t)u * stride;
      if (i >= n_vec) continue;
      if (!inside[u]) { // the buffer's first / last vector: only its own bytes are read
        for (int q = 0; q < 16; ++q) {
          const int64_t at = (int64_t)(i << 4) + q - (int64_t)sh;
          if (at >= 0 && at < (int64_t)total_bytes && non_base4((uint32_t)seqs[at] * 0x01010101u)) mark((uint64_t)at);
        }
        continue;
      }
      const uint32_t b0 = non_base4(x[u].x), b1 = non_base4(x[u].y), b2 = non_base4(x[u].z), b3 = non_base4(x[u].w);
      if ((b0 | b1 | b2 | b3) == 0u) continue;
      // (rare) a vector lies in one tile or on the border of two: its first and its last non-base say which
      const uint64_t first = (i << 4) - sh;
      const uint32_t lo = b0 ? 0u : b1 ? 4u : b2 ? 8u : 12u, hi = b3 ? 15u : b2 ? 11u : b1 ? 7u : 3u; // (to four bytes: a tile too many at worst)
      for (uint64_t t = (first + lo) / tile_bytes; t <= (first + hi) / tile_bytes; ++t) atomicOr(&flags[t >> 5], 1u << (t & 31u));
    }
  }
}
static __global__ __launch_bounds__(KR_MAX_THREADS) void tiles_count_flagged_kernel(const KmerRunsGenArgs a, const uint32_t* __restrict__ flags)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const RunShape shape = {a.C, a.rpr, a.inv_rpr, a.last_start, a.last_dup, a.stride, a.nwin, a.k};
  uint16_t* vbits = (uint16_t*)(lds_dyn + wave * a.vbits_dwords);
  const uint64_t n_waves_total = (uint64_t)gridDim.x * a.waves;
  const uint32_t reads_per_tile = 64u / a.rpr;
  const uint64_t n_groups = (a.n_wtiles + 63u) >> 6; // 64 tiles per wave and round: a tile per lane
  for (uint64_t gi = (uint64_t)blockIdx.x * a.waves + wave; gi < n_groups; gi += n_waves_total) {
    const uint64_t t = gi * 64u + lane;
    const bool live = t < a.n_wtiles;
    const bool marked = live && ((flags[t >> 5] >> (t & 31u)) & 1u);
    if (live && !marked) {
      const uint64_t runs_left = a.n_runs - t * 64u;
      a.tile_counts[t] = (runs_left < 64u ? (uint32_t)runs_left : 64u) * a.C;
      if (a.counts) {
        const uint64_t r0 = t * reads_per_tile;
        for (uint32_t r = 0; r < reads_per_tile; ++r)
          if ((r0 + r) * a.rpr < a.n_runs) a.counts[r0 + r] = a.nwin;
      }
    }
    uint64_t todo = __ballot(marked);
    while (todo) { // (rare) the marked tiles of the 64, one after the other, the whole wave on each
      const uint32_t l = (uint32_t)__builtin_ctzll(todo);
      todo &= todo - 1ull;
      count_one_tile<false>(a, shape, gi * 64u + l, vbits, lane);
    }
  }
}

} // namespace ntamd
