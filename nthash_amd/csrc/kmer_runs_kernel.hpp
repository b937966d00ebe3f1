// kmer_runs_kernel.hpp -- the headline kernel: contiguous k-mer hashing of
// fixed-length reads with a fully contiguous hash-stream write-out.
//
// Why this shape (measured on MI355X, profiles/r01_notes.md): HBM write
// bandwidth depends on how much contiguous memory one wave writes at a time.
// 128-byte pieces at a 960-byte stride (one row per read, the first design,
// kept as kmer_fixed_kernel) reach ~2.9 TB/s; a wave that writes >= 1 KiB
// contiguous per store instruction reaches ~6 TB/s.  So the unit of work here
// is a RUN of C consecutive windows inside one read, C | (len-k+1): the 64
// lanes of a wave own 64 consecutive runs, i.e. 64*C consecutive k-mers of the
// output stream (7.5 KiB for C=15), staged in a wave-private LDS tile and
// copied out as one contiguous block.
//
// Per lane: the first window of its run is hashed directly with byte-indexed
// LDS tables (4 bases per lookup, both strands in one 16-byte entry -- the
// same tables the spaced-seed kernel uses with an all-care mask; this replaces
// base_forward_hash/base_reverse_hash, src/kmer.cpp:43-73,123-152); the other
// C-1 windows are rolled (next_forward_hash/next_reverse_hash,
// src/kmer.cpp:84-94,164-174) with one 16-entry (in,out) pair-table lookup per
// step.  Waves never synchronise with each other after the tables are loaded:
// each wave stages its own ~1.2 KB slab of ASCII as a private 2-bit stream.
//
// Round 2 (profiles/r02_notes.md): the kernel sits on the rate HBM gives a MIX of its two streams (13 % reads,
// 87 % writes); apart, the streams run 10 % faster -- hence the phased path at the end of the kernel.
#pragma once

#include <hip/hip_runtime.h>

#include "kmer_kernels.hpp"

namespace ntamd {

constexpr int KR_MAX_THREADS = 1024;
// KR_CHUNKED=1 compiles the windowed path in (the end of the kernel; profiles/r02_notes.md): every wave of the chip loads
// the slabs of its next 16 tiles inside a common window of the constant 100 MHz clock and hashes / writes outside it.
// It is the only structure in which HBM served the two streams apart -- 17.1-17.9 ms against 18.5-18.9 ms per 100 M
// reads with the hash switched off -- but with the hash on a period holds 5 us of loads + 6 us of packing + 34 us of
// hashing at the 8 waves per CU its 80 slab registers allow, which is the period HBM needs: 19.9-20.6 ms against
// 19.3-19.6 for the static loop at 16 waves on the same boxes.  Off; parity-tested (tests run it through NTHASH_AMD_LIB).
#ifndef KR_CHUNKED
#define KR_CHUNKED 0
#endif
constexpr int KR_M1_THREADS = KR_CHUNKED ? 512 : 1024;
// a slab starts at the read's first byte rounded down to this many bytes (16: one vector; 64 / 128: whole lines)
#ifndef KR_SLAB_ALIGN
#define KR_SLAB_ALIGN 16
#endif
// Ablation builds (tools/ab_build.sh, WRONG results by design): where does the time of the headline kernel go?
//   KR_ABL_NOHASH   no first window, no rolls: staging + LDS copy-out + the memory streams only
//   KR_ABL_NOSTORE  the hash stream is not written (everything else as usual)
//   KR_ABL_NOLOAD   the next slab is not read (the first one is hashed again and again)
#ifndef KR_ABL_NOHASH
#define KR_ABL_NOHASH 0
#endif
#ifndef KR_ABL_NOSTORE
#define KR_ABL_NOSTORE 0
#endif
#ifndef KR_ABL_NOLOAD
#define KR_ABL_NOLOAD 0
#endif

// KR_DEBUG_TIMES=1 (measurement builds): every wave of the phased path adds the ticks it spent waiting for its own
// stores, waiting for the read window, loading + packing, waiting for the write window, and hashing + writing to five
// counters behind the dirty flag (bytes 64..104 of the context's scratch), printed by the launcher
#ifndef KR_DEBUG_TIMES
#define KR_DEBUG_TIMES 0
#endif
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

// counted wait: at most N vector-memory operations of this wave stay in flight
template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct KmerRunsArgs {
  const uint8_t* seqs;
  uint64_t* hashes;      // dense [read][window][m]
  uint32_t* dirty;
  // NTHIP_OUT_READ_SLOTS on fixed-length reads (round 3): a bit per 16-byte vector of the batch (vector index relative to
  // a.seqs rounded down to 16) -- set where a vector holds a non-base INSTEAD of flagging the batch dirty: the pass goes on
  // as if the batch were clean, the reads those vectors touch are redone in their slots afterwards.  NULL: the plain pass.
  uint32_t* vecmap;
  // the COMPACT stream of a batch with non-bases (round 4; burst path, m = 1): tile t's k-mers start at tile_off[t]
  // (the exclusive scan of the count pass's tile_counts); a tile that lost a window (tile_counts[t] < its windows) is left
  // out here -- the N-aware kernel writes those few, from a list.  Nothing is judged: the count pass has.  NULL: dense.
  const uint64_t* tile_off;
  const uint64_t* tile_counts;
  const uint4* init_tab; // global [ntab][256] {f.lo,f.hi,r.lo,r.hi}, all-care mask
  uint64_t n_reads;
  uint64_t n_runs;       // n_reads * rpr
  uint64_t n_wtiles;     // ceil(n_runs / 64)
  uint32_t len, stride, k, m;
  uint32_t nwin;
  uint32_t C;            // windows per run, C | nwin
  uint32_t rpr;          // runs per read = nwin / C
  uint32_t ntab;         // ceil(k/4)
  uint32_t waves;        // waves per block
  uint32_t bits_dwords;  // per-wave bit-stream capacity
  uint32_t tile_u64;     // per-wave tile capacity (64*C; 16 more for the compact stream: a tile starts anywhere in a line)
  uint32_t inv_rpr;      // floor(65536 / rpr) + 1
  uint32_t dword_tail;   // every slab is <= 1280 bytes: tail staged as one dword per lane
  uint32_t tile_map;     // number of wave groups of the tile -> wave mapping (see the kernel)
  // windowed path (KR_CHUNKED builds): tiles per period (0 = the static loop; every wave has that many bit streams in
  // LDS), period and read window in ticks of the constant 100 MHz clock (period 0 = groups without pacing)
  uint32_t ph_tiles, ph_period, ph_read;
  uint64_t tab[16][2];
  uint64_t mult[KF_MAX_RUNTIME_M];
};

// C_T: compile-time run length (0 = runtime); NW: window words, k <= 16*NW;
// DT: every slab is <= 1280 bytes, so its tail is staged as one dword per lane;
// PK: a.seqs is the 2-bit code stream nthip_pack_reads made (NTHIP_PACKED_INPUT: base i of the buffer at bits 2i, 2i + 1;
//     16 bases per dword -- the very format of the LDS bit stream, so a slab is staged with one dword load per 16 bases
//     and nothing is packed or judged; a.stride / a.len / offsets stay in bases).  DT shapes only (<= 128 dwords per slab).
template <int K_T, int M_T, int C_T, int NW, bool DT, bool PK = false>
__global__ __launch_bounds__(M_T == 1 ? KR_M1_THREADS : KR_MAX_THREADS) void kmer_runs_kernel(const KmerRunsArgs a)
{
  static_assert(!PK || (DT && !KR_CHUNKED), "packed input: the dword-tail shapes of the static loop");
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  const uint32_t k = K_T ? (uint32_t)K_T : a.k;
  const uint32_t m = M_T ? (uint32_t)M_T : a.m;
  const uint32_t C = C_T ? (uint32_t)C_T : a.C;
  // K_T == 0 (k at run time): the host pads the first-window tables to 4 per window word with zero tables
  // (kmer_ntab / get_kmer_tab), so that the lookups below stay unconditional
  const uint32_t ntab = K_T ? (uint32_t)((K_T + 3) / 4) : 4u * (uint32_t)NW;
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
#ifndef KR_UNIFORM_WAVE
#define KR_UNIFORM_WAVE 1 // (+1.5 % on C2: the tile geometry moves to the scalar unit)
#endif
#if KR_UNIFORM_WAVE
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6); // tile bookkeeping on the scalar unit
#else
  const uint32_t wave = tid >> 6;
#endif

  // LDS layout: init tables | pair table | multipliers | per-wave {tile, bits}
  uint4* itab = (uint4*)lds_dyn;
  uint4* ptab = itab + ntab * 256u;
  uint64_t* mults = (uint64_t*)(ptab + 16);
  uint32_t* wave_base =
      (uint32_t*)(mults + KF_MAX_RUNTIME_M) + wave * (a.tile_u64 * 2u + a.bits_dwords * (a.ph_tiles ? a.ph_tiles : 1u));
  uint64_t* tile = (uint64_t*)wave_base;
  uint32_t* bits = wave_base + a.tile_u64 * 2u; // the bit stream being packed / hashed (ph_tiles of them when phased)

  for (uint32_t i = tid; i < ntab * 256u; i += blockDim.x) itab[i] = a.init_tab[i];
  if (tid < 16)
    ptab[tid] = make_uint4((uint32_t)a.tab[tid][0], (uint32_t)(a.tab[tid][0] >> 32),
                           (uint32_t)a.tab[tid][1], (uint32_t)(a.tab[tid][1] >> 32));
  if (tid < KF_MAX_RUNTIME_M) mults[tid] = a.mult[tid];
  __syncthreads(); // the only block-wide barrier

  const uint32_t vals_per_run = C * m;
  const uint32_t inv_m = 0xFFFFFFFFu / m + 1u; // v / m == umulhi(v, inv_m) for v < 2^29 (runtime m)
  uint32_t bad = 0;
  // Tile bookkeeping without a 64-bit division per tile: (r_first, rem0) =
  // divmod(64*wt, rpr) is advanced by the constant divmod(64*wstride, rpr).
  // tile -> wave mapping: the waves of the grid are split into a.tile_map groups
  // of consecutive waves; every group owns one contiguous range of tiles and its
  // waves interleave inside it (0/1 = one group = plain grid stride; one group
  // per block measured best: each CU then streams through its own region)
  uint64_t wt, wstride, wt_end;
  uint64_t grp_t0 = 0, grp_w = 0; // first tile of the wave's group and the wave's index in it (KR_CONSEC)
  {
    const uint64_t n_waves_total = (uint64_t)gridDim.x * a.waves;
    const uint64_t gw = (uint64_t)blockIdx.x * a.waves + wave;
    uint64_t groups = a.tile_map ? a.tile_map : 1u;
    if (groups > n_waves_total) groups = n_waves_total;
    const uint64_t wpg = n_waves_total / groups;           // waves per group (last group may be larger)
    uint64_t g = gw / wpg;
    if (g >= groups) g = groups - 1;
    const uint64_t w_in_g = gw - g * wpg;
    const uint64_t g_waves = g == groups - 1 ? n_waves_total - g * wpg : wpg;
    const uint64_t per = (a.n_wtiles + groups - 1) / groups; // tiles per group
    const uint64_t t0 = g * per;
    wt = t0 + w_in_g;
    wstride = g_waves;
    wt_end = t0 + per < a.n_wtiles ? t0 + per : a.n_wtiles;
    grp_t0 = t0;
    grp_w = w_in_g;
  }
  uint64_t r_first = (wt * 64u) / a.rpr;
  uint32_t rem0 = (uint32_t)(wt * 64u - r_first * a.rpr);
  const uint64_t step_q = (wstride * 64u) / a.rpr;
  const uint32_t step_r = (uint32_t)(wstride * 64u - step_q * a.rpr);

  // slab geometry of the tile whose first run lives in read rf (run rm of it)
  struct Slab {
    uint64_t byte0; // offset of the first 16-byte vector from a.seqs (wraps below 0 by < 16)
    uint32_t shift, slab_bytes, n_vec, runs_here;
    uint32_t edge; // slab touches the first or the last byte of the caller's buffer
  };
  auto slab_of = [&](uint64_t g0, uint64_t rf, uint32_t rm) -> Slab {
    Slab sl;
    const uint64_t runs_left = a.n_runs - g0;
    sl.runs_here = runs_left < 64u ? (uint32_t)runs_left : 64u;
    const uint32_t n_slab_reads = (rm + sl.runs_here - 1u) / a.rpr + 1u;
    const uint64_t off = rf * a.stride;
    // (packed: positions are bases of the code stream, which starts on a 16-byte boundary; byte0 counts bases as well)
    sl.shift = PK ? (uint32_t)(off & 15u) : (uint32_t)(((uint64_t)a.seqs + off) & (uint32_t)(KR_SLAB_ALIGN - 1));
    sl.byte0 = off - sl.shift;
    sl.slab_bytes = (n_slab_reads - 1u) * a.stride + a.len;
    sl.n_vec = (sl.shift + sl.slab_bytes + 15u) >> 4;
    sl.edge = (rf == 0 || rf + n_slab_reads >= a.n_reads) ? 1u : 0u;
    return sl;
  };
  // Pack one 16-byte vector of the slab into the bit stream and judge its bytes.
  // Bytes of neighbouring reads inside an edge vector are judged too: they are
  // bytes of this batch, so a non-base there makes the batch dirty anyway.  Only
  // bytes outside the caller's buffer (before the first read / after the last)
  // must not be judged; that can only happen in the first and last slab.
  auto pack_vec = [&](const Slab& sl, uint32_t i, const uint4 v) {
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
#if !KR_NO_VECMAP // (A/B: the pass without the read-slots bookkeeping)
    if (b != 0u && a.vecmap != nullptr) { // (rare) read slots: remember the vector, keep going
      const uint64_t av = (((uint64_t)a.seqs + sl.byte0) >> 4) - ((uint64_t)a.seqs >> 4) + i;
      atomicOr(&a.vecmap[av >> 5], 1u << (av & 31u));
      b = 0;
    }
#endif
    bad |= b;
    bits[i] = p;
  };
  // the same for one dword (4 bases -> one byte of the stream): the tail of a slab
  auto pack_dword = [&](const Slab& sl, uint32_t j, const uint32_t wv) {
    uint32_t b = 0;
    const uint32_t p = pack4(wv, b);
    if (sl.edge) {
      const int32_t hi_cut = (int32_t)(sl.shift + sl.slab_bytes) - (int32_t)(1024u + (j << 2));
      uint32_t keep = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (q < hi_cut) keep |= 0xFFu << (q * 8);
      b &= keep;
    }
#if !KR_NO_VECMAP
    if (b != 0u && a.vecmap != nullptr) {
      const uint64_t av = (((uint64_t)a.seqs + sl.byte0) >> 4) - ((uint64_t)a.seqs >> 4) + 64u + (j >> 2);
      atomicOr(&a.vecmap[av >> 5], 1u << (av & 31u));
      b = 0;
    }
#endif
    bad |= b;
    ((uint8_t*)bits)[256u + j] = (uint8_t)p;
  };
  // stage vectors [first, n_vec) of a slab with ordinary loads
  auto stage = [&](const Slab& sl, uint32_t first) {
    if constexpr (PK) {
      const uint32_t* codes = (const uint32_t*)a.seqs + (sl.byte0 >> 4); // dword i = bases [byte0 + 16 i, ...)
      for (uint32_t i = first + lane; i < sl.n_vec; i += 64u) bits[i] = codes[i];
    } else {
      for (uint32_t i = first + lane; i < sl.n_vec; i += 64u)
        pack_vec(sl, i, *(const uint4*)(a.seqs + sl.byte0 + ((uint64_t)i << 4)));
    }
    if (lane < (uint32_t)NW + 3u) bits[sl.n_vec + lane] = 0; // funnels read a little ahead
  };
  auto lds_sync = [&]() { // LDS is in-order per wave: only the compiler must not reorder
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  };

  constexpr uint32_t NP = 32u * (uint32_t)(C_T ? C_T : 1); // 16-byte pieces of a full m=1 tile
  constexpr uint32_t NFULL = NP / 64u, REM = NP % 64u;
  constexpr uint32_t NST = NFULL + (REM ? 1u : 0u);        // store instructions of a full tile

  // ---- one tile: every lane hashes its run into the wave's LDS tile -----------------------------------
  auto hash_tile = [&](const uint32_t shift, const uint32_t runs_here, const uint32_t my_rem0, const uint32_t tpar = 0u) {
      const bool live = lane < runs_here;
      const uint32_t gl = live ? my_rem0 + lane : my_rem0; // run index relative to r_first's run 0
      const uint32_t lr = (gl * a.inv_rpr) >> 16;           // gl / rpr (gl < 64 + rpr: exact)
      const uint32_t q = gl - lr * a.rpr;                   // run inside the read
      const uint32_t b0 = shift + lr * a.stride + q * C;    // first base of the first window
      const uint32_t d0 = b0 >> 4, sh0 = (b0 & 15u) << 1;
#if KR_ABL_NOHASH
      (void)d0; (void)sh0; (void)tpar;
      tile[lane] = (uint64_t)b0; // (keeps the geometry alive)
#else
      uint32_t w[NW];
      {
        uint32_t lo = bits[d0];
#pragma unroll
        for (int i = 0; i < NW; ++i) {
          const uint32_t hi = bits[d0 + i + 1];
          w[i] = funnel(hi, lo, sh0);
          lo = hi;
        }
      }
      // first window: XOR of per-byte table entries (4 bases per lookup)
      uint32_t f_lo = 0, f_hi = 0, r_lo = 0, r_hi = 0;
#ifndef KR_BATCH_INIT
#define KR_BATCH_INIT 1
#endif
#if KR_BATCH_INIT
      {
        // all lookups in flight before the first XOR (the wave is one of two on its SIMD: registers are plentiful)
        uint4 e[4 * NW];
#pragma unroll
        for (int jt = 0; jt < 4 * NW; ++jt) {
          const uint32_t byte = (w[jt >> 2] >> ((jt & 3) * 8)) & 0xFFu;
          e[jt] = (uint32_t)jt < ntab ? itab[(uint32_t)jt * 256u + byte] : make_uint4(0, 0, 0, 0);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int jt = 0; jt < 4 * NW; ++jt) { f_lo ^= e[jt].x; f_hi ^= e[jt].y; r_lo ^= e[jt].z; r_hi ^= e[jt].w; }
      }
#else
#pragma unroll
      for (int jt = 0; jt < 4 * NW; ++jt) {
        if ((uint32_t)jt < ntab) {
          const uint32_t byte = (w[jt >> 2] >> ((jt & 3) * 8)) & 0xFFu;
          const uint4 e = itab[(uint32_t)jt * 256u + byte];
          f_lo ^= e.x; f_hi ^= e.y; r_lo ^= e.z; r_hi ^= e.w;
        }
      }
#endif
      uint64_t* my_row = tile + tpar + lane * C; // (compact stream: the tile starts tpar values into a line of the stream)
      my_row[0] = canon_pair(f_lo, f_hi, r_lo, r_hi);

      // remaining C-1 windows: roll.  step t (1..C-1): in = base b0+k-1+t, out = base b0+t-1
      const uint32_t bi = b0 + k;
      const uint32_t di = bi >> 4, shi = (bi & 15u) << 1;
      auto roll_word = [&](uint32_t jw, auto n_tag) {
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
          roll_step(f_lo, f_hi, r_lo, r_hi, term);
        };
        constexpr uint32_t NS = decltype(n_tag)::value; // 0 = runtime count
        if constexpr (NS != 0) {
          // table terms do not depend on the hash state: fetch them in batches
          // ahead of the dependent chain so their LDS latencies overlap
          constexpr uint32_t B = 8;
#pragma unroll
          for (uint32_t i0 = 0; i0 < NS; i0 += B) {
            uint4 terms[B];
#pragma unroll
            for (uint32_t i = 0; i < B; ++i)
              if (i0 + i < NS) terms[i] = lookup(i0 + i);
#pragma unroll
            for (uint32_t i = 0; i < B; ++i) {
              if (i0 + i < NS) {
                roll(terms[i]);
                my_row[jw * 16u + i0 + i + 1u] =
                    canon_pair(f_lo, f_hi, r_lo, r_hi);
              }
            }
          }
        } else {
          // runtime step count: full batches of 8 (terms prefetched), then the remainder
          const uint32_t left = C - 1u - jw * 16u;
          const uint32_t ns = left < 16u ? left : 16u;
          uint32_t i0 = 0;
          for (; i0 + 8u <= ns; i0 += 8u) {
            uint4 terms[8];
#pragma unroll
            for (uint32_t i = 0; i < 8; ++i) terms[i] = lookup(i0 + i);
#pragma unroll
            for (uint32_t i = 0; i < 8; ++i) {
              roll(terms[i]);
              my_row[jw * 16u + i0 + i + 1u] =
                  canon_pair(f_lo, f_hi, r_lo, r_hi);
            }
          }
#pragma unroll 1
          for (uint32_t i = i0; i < ns; ++i) {
            roll(lookup(i));
            my_row[jw * 16u + i + 1u] =
                canon_pair(f_lo, f_hi, r_lo, r_hi);
          }
        }
      };
      if constexpr (C_T != 0 && C_T <= 33) {
        constexpr uint32_t NROLL = (uint32_t)(C_T - 1);
        if constexpr (NROLL > 0) roll_word(0u, std::integral_constant<uint32_t, (NROLL < 16u ? NROLL : 16u)>{});
        if constexpr (NROLL > 16) roll_word(1u, std::integral_constant<uint32_t, NROLL - 16u>{});
      } else {
        for (uint32_t jw = 0; jw * 16u + 1u < C; ++jw) roll_word(jw, std::integral_constant<uint32_t, 0u>{});
      }

#endif // KR_ABL_NOHASH

  };

  // ---- copy the tile out: 64*C*m consecutive values of the hash stream; true = a full tile left as exactly
  // NST store instructions (what the counted waits below rely on) ---------------------------------------
  // cmp_off != ~0: the tile's first value in the compact stream (m = 1): the tile was built tpar = cmp_off & 15 slots up
  auto copy_out = [&](const uint64_t g0, const uint32_t runs_here, const uint64_t cmp_off = ~0ull) -> bool {
    lds_sync();
      const bool cmp = cmp_off != ~0ull;
      const uint32_t tpar = cmp ? (uint32_t)(cmp_off & 15u) : 0u;
      uint64_t* const out0 = cmp ? a.hashes + (cmp_off - tpar) : a.hashes + g0 * vals_per_run;
      const uint32_t n_vals = runs_here * vals_per_run;
      const uint32_t n_pairs = (n_vals + 1u) >> 1;
      bool counted = false;
      if (m == 1) {
        if (cmp) {
          // the tile was built tpar slots up, so that piece 0 starts a 128-byte line of the stream: every store instruction
          // covers whole lines.  Lines that lie inside the tile leave write-through, as the dense tiles do; the two lines it
          // shares with its neighbours in plain stores (write-through on pieces that do not fill their lines: 4.4 -> 9.1 ms
          // per 2.4 G k-mers; plain or streaming stores from 16-byte boundaries anywhere: 4.4-4.5).  A full tile still
          // leaves as at least 8 store instructions (the burst path's wait)
          const uint32_t span = tpar + n_vals, pieces = (span + 1u) >> 1;
          const uint32_t line_lo = (tpar + 15u) >> 4, line_hi = span >> 4; // whole lines: [line_lo, line_hi)
#ifndef KR_CMP_STORE
#define KR_CMP_STORE 1 // whole lines: 0 plain, 1 streaming (nt), 2 write-through (sc0 sc1)
#endif
          auto put_line = [&](const uint32_t pi, const uint4 dv) { // a piece of a whole line
#if KR_CMP_STORE == 2
            stream_store16(out0 + 2u * pi, dv);
#elif KR_CMP_STORE == 1
            __builtin_nontemporal_store(*(const v4u*)&dv, (v4u*)(out0 + 2u * pi));
#else
            *(uint4*)(out0 + 2u * pi) = dv;
#endif
          };
          auto put_edge = [&](const uint32_t pi) { // a piece of a line the tile shares with a neighbour
            const uint4 dv = *(const uint4*)(tile + 2u * pi);
            const bool lo_ok = 2u * pi >= tpar && 2u * pi < span, hi_ok = 2u * pi + 1u >= tpar && 2u * pi + 1u < span;
            if (lo_ok && hi_ok) *(uint4*)(out0 + 2u * pi) = dv;
            else if (lo_ok) *(uint2*)(out0 + 2u * pi) = make_uint2(dv.x, dv.y);
            else if (hi_ok) *(uint2*)(out0 + 2u * pi + 1u) = make_uint2(dv.z, dv.w);
          };
          // whole lines: groups of four LDS reads, then their stores (the slab registers of the burst path leave no room
          // for the dense copy-out's eight); a full tile is 60 or 61 whole lines = 8 store instructions
          const uint32_t p_lo = line_lo << 3, p_hi = line_hi << 3; // their pieces: [p_lo, p_hi)
          for (uint32_t p0 = p_lo; p0 < p_hi; p0 += 256u) {
            const uint4* src = (const uint4*)tile + p0 + lane;
            const uint4 d0 = src[0], d1 = src[64], d2 = src[128], d3 = src[192]; // (past p_hi: the wave's own LDS, not stored)
            if (p0 + lane < p_hi) put_line(p0 + lane, d0);
            if (p0 + 64u + lane < p_hi) put_line(p0 + 64u + lane, d1);
            if (p0 + 128u + lane < p_hi) put_line(p0 + 128u + lane, d2);
            if (p0 + 192u + lane < p_hi) put_line(p0 + 192u + lane, d3);
          }
          if (lane < 8u) { if (line_lo != 0u) put_edge(lane); }                       // the tile's first line
          else if (lane < 16u) { if (pieces > p_hi) put_edge(p_hi + lane - 8u); }     // and its last
        } else if (C_T != 0 && runs_here == 64u) {
          // full tile, compile-time shape: LDS reads first, then the stores, in groups of 8
          // (named registers, not an array: hipcc sends a partially predicated
          // uint4 array to scratch, whose traffic would also break the counted wait)
          static_assert(NST <= 16, "flush is written for at most 16 store instructions");
          const uint4* src = (const uint4*)tile + lane;
          uint4* dst = (uint4*)out0 + lane;
          uint4 d0, d1, d2, d3, d4, d5, d6, d7;
#define KR_LD(n, var) \
          if constexpr ((n) < NFULL) var = src[(n) * 64u]; \
          else if constexpr ((n) == NFULL && REM != 0) { if (lane < REM) var = src[(n) * 64u]; }
#if KR_ABL_NOSTORE
#define KR_ST1(p, var) asm volatile("" ::"v"((p)), "v"((var).x), "v"((var).y), "v"((var).z), "v"((var).w))
#else
#define KR_ST1(p, var) stream_store16((p), var)
#endif
#define KR_ST(n, var) \
          if constexpr ((n) < NFULL) KR_ST1(dst + (n) * 64u, var); \
          else if constexpr ((n) == NFULL && REM != 0) { if (lane < REM) KR_ST1(dst + (n) * 64u, var); }
#define KR_GROUP(b) \
          KR_LD(b + 0, d0) KR_LD(b + 1, d1) KR_LD(b + 2, d2) KR_LD(b + 3, d3) \
          KR_LD(b + 4, d4) KR_LD(b + 5, d5) KR_LD(b + 6, d6) KR_LD(b + 7, d7) \
          KR_ST(b + 0, d0) KR_ST(b + 1, d1) KR_ST(b + 2, d2) KR_ST(b + 3, d3) \
          KR_ST(b + 4, d4) KR_ST(b + 5, d5) KR_ST(b + 6, d6) KR_ST(b + 7, d7)
          KR_GROUP(0)
          if constexpr (NST > 8) { KR_GROUP(8) }
#undef KR_GROUP
#undef KR_LD
#undef KR_ST
#undef KR_ST1
          counted = !KR_ABL_NOSTORE;
        } else {
          for (uint32_t pi = lane; pi < n_pairs; pi += 64u) {
            const uint4 dv = *(const uint4*)(tile + 2u * pi);
            if (2u * pi + 1u < n_vals) *(uint4*)(out0 + 2u * pi) = dv;
            else *(uint2*)(out0 + 2u * pi) = make_uint2(dv.x, dv.y);
          }
        }
      } else {
        // multi-hash expansion (extend_hashes, src/internal.hpp:104-118) fused into the
        // copy-out: the tile holds h[0] only; value v of the stream is h[v % m] of k-mer v / m
        for (uint32_t pi = lane; pi < n_pairs; pi += 64u) {
          uint64_t o[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const uint32_t vi = 2u * pi + (uint32_t)h;
            const uint32_t e = M_T ? vi / m : __umulhi(vi, inv_m), jj = vi - e * m;
            const uint64_t h0 = tile[e < runs_here * C ? e : 0];
            o[h] = jj == 0 ? h0 : mix_hash(h0, mults[jj & (KF_MAX_RUNTIME_M - 1)]);
          }
#if KR_ABL_NOSTORE
          asm volatile("" ::"v"((uint32_t)o[0]), "v"((uint32_t)(o[0] >> 32)), "v"((uint32_t)o[1]), "v"((uint32_t)(o[1] >> 32)));
#else
#ifndef KR_MH_STORE
#define KR_MH_STORE 1 // 1: write-through (sc0 sc1) as the m = 1 copy-out (a tile's values are whole, aligned lines), 0: plain
#endif
          if (2u * pi + 1u < n_vals) {
            const uint4 ov = make_uint4((uint32_t)o[0], (uint32_t)(o[0] >> 32), (uint32_t)o[1], (uint32_t)(o[1] >> 32));
#if KR_MH_STORE
            stream_store16(out0 + 2u * pi, ov);
#else
            *(uint4*)(out0 + 2u * pi) = ov;
#endif
          } else {
            *(uint2*)(out0 + 2u * pi) = make_uint2((uint32_t)o[0], (uint32_t)(o[0] >> 32));
          }
#endif
        }
      }
    return counted;
  };

  // (the windowed path exists in the m = 1 instantiations of a KR_CHUNKED build only: 512 threads, 256 VGPRs)
  constexpr bool KR_WINDOWED = KR_CHUNKED && DT && M_T == 1;
  // ---- burst path (round 3; DT shapes whose tiles are whole reads: 64 % rpr == 0, stride == len; a.ph_tiles == KR_BURST) --
  // What the reads of this kernel cost HBM is how OFTEN they come, not how many bytes they are: with the hash switched
  // off the static loop below takes 18.6 ms per 100 M reads whether it reads ASCII or 2-bit packed input (a quarter of
  // the bytes), the write stream alone 13.7.  Read in bursts -- the slabs of 16 tiles per wave at a time -- the same
  // memory-only loop takes 17.0-17.3 ms (profiles/r03_notes.md).  So a wave takes PIECES of KR_BURST consecutive tiles:
  // one contiguous read of KR_BURST slabs (9.4 KiB), packed into KR_BURST bit streams in LDS, then KR_BURST tiles hashed
  // and written as one contiguous 60 KiB of the stream.  The next piece's loads are issued before the first tile is
  // hashed and consumed after the last tile's stores: 8 x 8 = 64 younger stores, and an in-order counter that holds at
  // most 63 operations proves they have landed (the seed kernel's argument) -- no pacing, no block barrier, any number
  // of waves.  The waves of a group take the pieces of its tile range in turn.
#ifndef KR_NO_VECMAP
#define KR_NO_VECMAP 0
#endif
#ifndef KR_BURST_PLAIN_LOADS
#define KR_BURST_PLAIN_LOADS 0
#endif
#ifndef KR_BURST
#define KR_BURST 8
#endif
  constexpr bool KR_BURST_OK = !KR_CHUNKED && DT && C_T != 0 && NST == 8u;
  if constexpr (KR_BURST_OK) {
    if (a.ph_tiles == (uint32_t)KR_BURST) {
      constexpr uint32_t P = KR_BURST;
      static_assert(P * NST >= 64u, "the counted wait of the burst path needs 64 stores per piece");
      if constexpr (PK) {
        // packed input: the piece's P slabs are ONE contiguous run of the code stream (P x 1200 bases = 601 dwords for
        // config 2), loaded with PKR dword loads per lane and kept as ONE bit stream in LDS -- tile q of the piece
        // starts q tiles' worth of bases further into it; nothing is packed or judged
        constexpr uint32_t PKR = 12; // dwords per lane: 768 x 16 bases >= a piece of 8 dword-tail slabs (8 x 1280 + 15 bases)
        const uint32_t reads_per_tile = 64u / a.rpr;
        const uint64_t tile_bases = (uint64_t)reads_per_tile * a.stride;
        uint64_t sgrp = grp_w;
        uint64_t pt = grp_t0 + sgrp * P;
        uint32_t pk[PKR];
        auto piece_dwords = [&](const uint64_t t0) -> uint32_t { // dwords of the piece's stream (bases of its tiles' reads)
          const uint64_t t_last = t0 + P <= wt_end ? t0 + P - 1u : wt_end - 1u;
          const uint64_t reads_end = (t_last + 1u) * reads_per_tile < a.n_reads ? (t_last + 1u) * reads_per_tile : a.n_reads;
          const uint64_t first_base = t0 * tile_bases, end_base = (reads_end - 1u) * a.stride + a.len;
          return (uint32_t)(((first_base & 15u) + (end_base - first_base) + 15u) >> 4);
        };
        auto issue_piece = [&](const uint64_t t0) {
          const uint64_t first_base = t0 * tile_bases;
          const uint32_t* codes = (const uint32_t*)a.seqs + (first_base >> 4);
          const uint32_t n_dw = piece_dwords(t0);
#pragma unroll
          for (uint32_t r = 0; r < PKR; ++r) {
            const uint32_t i = r * 64u + lane;
            const uint32_t* q = codes + (i < n_dw ? i : 0u);
            asm volatile("global_load_dword %0, %1, off nt" : "=&v"(pk[r]) : "v"(q) : "memory");
          }
        };
        auto store_piece = [&](const uint64_t t0) {
          const uint32_t n_dw = piece_dwords(t0);
#pragma unroll
          for (uint32_t r = 0; r < PKR; ++r) {
            const uint32_t i = r * 64u + lane;
            if (i < n_dw) bits[i] = pk[r];
          }
          if (lane < (uint32_t)NW + 3u) bits[n_dw + lane] = 0;
        };
#define KR_PK_MARK() \
        asm volatile("; NTLINT_CONSUME %0 %1 %2 %3 %4 %5 %6 %7 %8 %9 %10 %11" \
                     : "+v"(pk[0]), "+v"(pk[1]), "+v"(pk[2]), "+v"(pk[3]), "+v"(pk[4]), "+v"(pk[5]), "+v"(pk[6]), "+v"(pk[7]), \
                       "+v"(pk[8]), "+v"(pk[9]), "+v"(pk[10]), "+v"(pk[11])::"memory")
        static_assert(PKR == 12u, "the marker above lists 12 registers");
        if (pt < wt_end) {
          issue_piece(pt);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          KR_PK_MARK();
          store_piece(pt);
        }
        while (pt < wt_end) {
          const uint64_t left = wt_end - pt;
          const uint32_t n_here = left < P ? (uint32_t)left : P;
          const uint64_t npt = grp_t0 + (sgrp + wstride) * P;
          const bool have_next = npt < wt_end;
          issue_piece(have_next ? npt : pt);
          bool all_full = n_here == P;
          const uint32_t shift0 = (uint32_t)((pt * tile_bases) & 15u);
          for (uint32_t q = 0; q < n_here; ++q) {
            const uint64_t t = pt + q;
            const uint64_t g0 = t * 64u;
            const uint64_t runs_left = a.n_runs - g0;
            const uint32_t runs_here = runs_left < 64u ? (uint32_t)runs_left : 64u;
            lds_sync();
            hash_tile(shift0 + q * (uint32_t)tile_bases, runs_here, 0u);
            NT_LINT_SELFTEST_TOUCH(pk[0]);
            (void)copy_out(g0, runs_here);
            all_full = all_full && runs_here == 64u;
            lds_sync(); // the tile is free again
          }
          asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
          if (!all_full) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          KR_PK_MARK();
          sgrp += wstride;
          pt = npt;
          if (have_next) store_piece(pt);
        }
#undef KR_PK_MARK
        return;
      }
      uint32_t* const bits0 = bits;
      const bool compact = M_T == 1 && a.tile_off != nullptr;
      const uint32_t reads_per_tile = 64u / a.rpr;
      const uint64_t tile_bytes = (uint64_t)reads_per_tile * a.stride;
      const uint64_t base_addr = (uint64_t)a.seqs;
      // bytes of tile t's slab (the batch's last tile may hold fewer reads: nothing past the buffer is read or judged)
      auto slab_bytes_of = [&](const uint64_t t) -> uint32_t {
        const uint64_t reads_left = a.n_reads - t * reads_per_tile;
        const uint32_t nr = reads_left < reads_per_tile ? (uint32_t)reads_left : reads_per_tile;
        return (nr - 1u) * a.stride + a.len;
      };
      uint64_t sgrp = grp_w;
      uint64_t pt = grp_t0 + sgrp * P; // first tile of the wave's current piece
      v4u v[P];
      uint32_t w4[P];
      uint32_t dirty_seen = 0;
      // loads of the P slabs of the piece that starts at tile t0 (tiles past the range reload the range's last slab)
      auto issue_piece = [&](const uint64_t t0) {
#pragma unroll
        for (uint32_t i = 0; i < P; ++i) {
          uint64_t t = t0 + i;
          if (t >= wt_end) t = wt_end - 1u;
          const uint64_t off = t * tile_bytes;
          const uint32_t shift = (uint32_t)((base_addr + off) & 15u);
          const uint64_t b0 = off - shift;
          const uint32_t slab_bytes = slab_bytes_of(t);
          const uint32_t n_vec = (shift + slab_bytes + 15u) >> 4;
          const uint32_t n_dw = (shift + slab_bytes + 3u) >> 2;
          const uint32_t i0 = lane < n_vec ? lane : 0u;
          const uint32_t jd = 256u + lane < n_dw ? 256u + lane : 0u;
          const uint64_t sb = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b0) |
                              ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b0 >> 32)) << 32);
          const uint8_t* sbase = a.seqs + sb; // scalar base + 32-bit lane offset
#if KR_BURST_PLAIN_LOADS // (experiment: default cache policy on the slab loads, for the input-prefetch experiment)
#define KR_BURST_POLICY ""
#else
#define KR_BURST_POLICY " sc1 nt"
#endif
          asm volatile("global_load_dwordx4 %0, %2, %4" KR_BURST_POLICY "\n\tglobal_load_dword %1, %3, %4" KR_BURST_POLICY
                       : "=&v"(v[i]), "=&v"(w4[i])
                       : "v"(i0 << 4), "v"(jd << 2), "s"(sbase)
                       : "memory");
        }
        asm volatile("global_load_dword %0, %1, off sc1" : "=&v"(dirty_seen) : "v"(a.dirty) : "memory");
      };
      auto pack_piece = [&](const uint64_t t0) {
#pragma unroll
        for (uint32_t i = 0; i < P; ++i) {
          const uint64_t t = t0 + i;
          if (t < wt_end) {
            Slab sl;
            const uint32_t slab_bytes = slab_bytes_of(t);
            sl.shift = (uint32_t)((base_addr + t * tile_bytes) & 15u);
            sl.slab_bytes = slab_bytes;
            sl.n_vec = (sl.shift + slab_bytes + 15u) >> 4;
            sl.byte0 = t * tile_bytes - sl.shift; // (only the read-slots bookkeeping of pack_vec looks at it here)
            sl.runs_here = 64u;
            sl.edge = (t == 0u || (t + 1u) * reads_per_tile >= a.n_reads) ? 1u : 0u; // first / last slab of the buffer
            bits = bits0 + i * a.bits_dwords;
            if (lane < sl.n_vec) pack_vec(sl, lane, make_uint4(v[i].x, v[i].y, v[i].z, v[i].w));
            const uint32_t n_dw = (sl.shift + slab_bytes + 3u) >> 2;
            if (256u + lane < n_dw) pack_dword(sl, lane, w4[i]);
            if (lane < (uint32_t)NW + 3u) bits[sl.n_vec + lane] = 0;
          }
        }
      };
#define KR_BURST_MARK() \
      asm volatile("; NTLINT_CONSUME %0 %1 %2 %3 %4 %5 %6 %7 %8 %9 %10 %11 %12 %13 %14 %15 %16" \
                   : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), \
                     "+v"(w4[0]), "+v"(w4[1]), "+v"(w4[2]), "+v"(w4[3]), "+v"(w4[4]), "+v"(w4[5]), "+v"(w4[6]), "+v"(w4[7]), \
                     "+v"(dirty_seen)::"memory")
      static_assert(P == 8u, "the marker above lists 8 slabs");
      uint64_t nx_cnt = 0, nx_off = 0; // compact stream: count and offset of the tile that comes next
      if (pt < wt_end) {
        issue_piece(pt);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        KR_BURST_MARK();
        pack_piece(pt);
        if (compact) {
          nx_cnt = a.tile_counts[pt];
          nx_off = a.tile_off[pt];
        }
      }
      while (pt < wt_end) {
        const uint64_t left = wt_end - pt;
        const uint32_t n_here = left < P ? (uint32_t)left : P;
        const uint64_t npt = grp_t0 + (sgrp + wstride) * P;
        const bool have_next = npt < wt_end;
        // a non-base in the piece just packed (or anywhere in the batch, as of the last look): the caller redoes the
        // batch on the N-aware path, so stop producing a dense stream nobody will read
        if (compact) bad = 0; // (the count pass has judged the bytes; a.dirty is a word that stays 0)
        if (__ballot(bad != 0) != 0) {
          if (lane == 0) atomicOr(a.dirty, 1u);
          break;
        }
        if (__builtin_amdgcn_readfirstlane(dirty_seen) != 0u) break;
        issue_piece(have_next ? npt : pt);
        bool all_full = n_here == P; // (a full tile leaves as >= 8 store instructions, whatever m is)
        for (uint32_t q = 0; q < n_here; ++q) {
          const uint64_t t = pt + q;
          const uint64_t g0 = t * 64u;
          const uint64_t runs_left = a.n_runs - g0;
          const uint32_t runs_here = runs_left < 64u ? (uint32_t)runs_left : 64u;
          const uint32_t shift = (uint32_t)((base_addr + t * tile_bytes) & 15u);
          uint64_t cmp_off = ~0ull;
          if (compact) {
            // (the next tile's count and offset are asked for now, a tile ahead: scalar loads whose latency would
            //  otherwise stand in front of every tile)
            const uint64_t t_cnt = nx_cnt;
            cmp_off = nx_off;
            const uint64_t tn = q + 1u < n_here ? t + 1u : (have_next ? npt : t);
            nx_cnt = a.tile_counts[tn];
            nx_off = a.tile_off[tn];
            if (t_cnt != (uint64_t)runs_here * C) { // a window lost: the N-aware kernel's tile
              all_full = false;
              continue;
            }
          }
          bits = bits0 + q * a.bits_dwords;
          lds_sync();
          hash_tile(shift, runs_here, 0u, compact ? (uint32_t)(cmp_off & 15u) : 0u);
          NT_LINT_SELFTEST_TOUCH(dirty_seen);
          (void)copy_out(g0, runs_here, cmp_off);
          all_full = all_full && runs_here == 64u;
          lds_sync(); // the tile is free again
        }
        // P full tiles = at least 64 store instructions issued behind the 17 loads: "at most 63 in flight" says they have landed
        asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
        if (!all_full) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        KR_BURST_MARK();
        sgrp += wstride;
        pt = npt;
        if (have_next) pack_piece(pt);
      }
#undef KR_BURST_MARK
      if (__ballot(bad != 0) != 0 && lane == 0) atomicOr(a.dirty, 1u);
      return;
    }
  }
  if (!KR_WINDOWED || a.ph_tiles == 0u) {
    Slab cur;
    cur.byte0 = 0;
    cur.shift = cur.slab_bytes = cur.n_vec = cur.runs_here = 0;
    cur.edge = 1u;
    if (wt < wt_end) {
      cur = slab_of(wt * 64u, r_first, rem0);
      stage(cur, 0u);
    }
    for (; wt < wt_end; wt += wstride) {
      lds_sync();
      const uint64_t g0 = wt * 64u;
      const uint32_t shift = cur.shift, runs_here = cur.runs_here;
      const uint32_t my_rem0 = rem0;
      // ---- issue the loads of the NEXT tile's slab (two vectors per lane) --------
      r_first += step_q;
      rem0 += step_r;
      if (rem0 >= a.rpr) { rem0 -= a.rpr; r_first += 1; }
      const uint64_t nwt = wt + wstride;
      const bool have_next = nwt < wt_end;
      Slab nxt = cur;
      if (have_next) nxt = slab_of(nwt * 64u, r_first, rem0);
      // vector `lane` of the slab, plus its tail: slabs of at most 1280 bytes
      // (a.dword_tail, the common case) finish with ONE DWORD per lane -- a second
      // 16-byte round would leave most lanes idle --, longer slabs with a second vector
      v4u pv0, pv1;
      uint32_t pw;
      uint32_t dirty_seen; // the batch-wide dirty flag, read along with the next slab
      {
        // lanes without an item re-read item 0 so that every lane issues both loads
        const uint32_t i0 = lane < nxt.n_vec ? lane : 0u;
        const uint8_t* p0 = a.seqs + nxt.byte0 + ((uint64_t)i0 << 4);
#if KR_ABL_NOLOAD
        (void)p0;
        pv0 = v4u{0x41414141u, 0x43434343u, 0x47474747u, 0x54545454u};
        pv1 = pv0;
        pw = 0x41434754u;
        dirty_seen = 0;
        asm volatile("" : "+v"(pv0), "+v"(pv1), "+v"(pw), "+v"(dirty_seen));
#else
        if constexpr (PK) {
          // the slab's dwords lane and 64 + lane of the code stream (a DT slab has at most 80)
          const uint32_t* codes = (const uint32_t*)a.seqs + (nxt.byte0 >> 4);
          const uint32_t* q0 = codes + (lane < nxt.n_vec ? lane : 0u);
          const uint32_t* q1 = codes + (64u + lane < nxt.n_vec ? 64u + lane : 0u);
          asm volatile("global_load_dword %0, %2, off nt\n\tglobal_load_dword %1, %3, off nt"
                       : "=&v"(pw), "=&v"(dirty_seen)
                       : "v"(q0), "v"(q1)
                       : "memory");
          (void)p0;
        } else if constexpr (DT) {
          const uint32_t n_dw = (nxt.shift + nxt.slab_bytes + 3u) >> 2; // dwords in the slab
          const uint32_t j = 256u + lane < n_dw ? 256u + lane : 0u;
          const uint8_t* p1 = a.seqs + nxt.byte0 + ((uint64_t)j << 2);
#ifndef KR_LOAD_NT
#define KR_LOAD_NT " sc1 nt"
#endif
#ifndef KR_FLAG_SC
#define KR_FLAG_SC " sc1"
#endif
          asm volatile("global_load_dword %2, %5, off" KR_FLAG_SC "\n\tglobal_load_dwordx4 %0, %3, off" KR_LOAD_NT "\n\t"
                       "global_load_dword %1, %4, off" KR_LOAD_NT
                       : "=&v"(pv0), "=&v"(pw), "=&v"(dirty_seen)
                       : "v"(p0), "v"(p1), "v"(a.dirty)
                       : "memory");
        } else {
          const uint32_t i1 = lane + 64u < nxt.n_vec ? lane + 64u : 0u;
          const uint8_t* p1 = a.seqs + nxt.byte0 + ((uint64_t)i1 << 4);
          asm volatile("global_load_dword %2, %5, off sc1\n\tglobal_load_dwordx4 %0, %3, off\n\t"
                       "global_load_dwordx4 %1, %4, off"
                       : "=&v"(pv0), "=&v"(pv1), "=&v"(dirty_seen)
                       : "v"(p0), "v"(p1), "v"(a.dirty)
                       : "memory");
        }
#endif
      }

      hash_tile(shift, runs_here, my_rem0);
      NT_LINT_SELFTEST_TOUCH(dirty_seen);
      const bool counted = copy_out(g0, runs_here);
      lds_sync(); // tile and bits are free again

      // ---- consume the prefetched slab ---------------------------------------------
      // After a counted full tile the only VMEM operations younger than the two
      // loads are its NST stores: wait until at most NST operations are in flight.
      // (NST is 8 for C=15 and 15 for C=30)
      // (the waits carry no operands and the marker is ONE statement behind them: with the registers tied to each of
      //  three alternative wait statements hipcc allocated them differently per branch and placed the copies of the
      //  vmcnt(0) branch BEFORE its wait -- found by the ISA lint, round 3.  The counted wait is unconditional -- it is
      //  the weaker one -- so that every path into the marker passes an inline wait: lint rule R3.)
      if constexpr (NST == 8u) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if constexpr (NST == 15u) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!counted) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if constexpr (PK) asm volatile("; NTLINT_CONSUME %0 %1" : "+v"(pw), "+v"(dirty_seen)::"memory");
      else if constexpr (DT) asm volatile("; NTLINT_CONSUME %0 %1 %2" : "+v"(pv0), "+v"(pw), "+v"(dirty_seen)::"memory");
      else asm volatile("; NTLINT_CONSUME %0 %1 %2" : "+v"(pv0), "+v"(pv1), "+v"(dirty_seen)::"memory");
      if constexpr (PK) { // (dirty_seen holds the slab's second dword here: a packed batch has no non-bases to find)
        if (have_next) {
          cur = nxt;
          if (lane < cur.n_vec) bits[lane] = pw;
          if (64u + lane < cur.n_vec) bits[64u + lane] = dirty_seen;
          if (lane < (uint32_t)NW + 3u) bits[cur.n_vec + lane] = 0;
        }
        continue;
      }
      // some wave already found a non-base byte: the caller will redo the batch on the
      // N-aware path, so stop producing a dense stream nobody will read
      if (__builtin_amdgcn_readfirstlane(dirty_seen) != 0u) break;
      if (have_next) {
        cur = nxt;
        if (lane < cur.n_vec) pack_vec(cur, lane, make_uint4(pv0.x, pv0.y, pv0.z, pv0.w));
        if constexpr (DT) {
          const uint32_t n_dw = (cur.shift + cur.slab_bytes + 3u) >> 2;
          if (256u + lane < n_dw) pack_dword(cur, lane, pw);
          if (lane < (uint32_t)NW + 3u) bits[cur.n_vec + lane] = 0;
        } else {
          if (lane + 64u < cur.n_vec) pack_vec(cur, lane + 64u, make_uint4(pv1.x, pv1.y, pv1.z, pv1.w));
          stage(cur, 128u); // slabs longer than 128 vectors: the rest with ordinary loads
        }
        if (__ballot(bad != 0) != 0) { // publish at once so that every wave can stop early
          if (lane == 0) atomicOr(a.dirty, 1u);
          break;
        }
      }
    }
  }
#if KR_CHUNKED
  else if constexpr (KR_WINDOWED) {
    // ---- windowed path (DT shapes whose tiles are whole reads: 64 % rpr == 0, stride == len) --------------------------
    // HBM serves this kernel's two streams far better apart than mixed (profiles/r02_notes.md: with the hash switched
    // off, 18.6 ms per 100 M reads mixed, 17.3 ms or less in phases).  So every wave of the chip READS in the same time
    // window and writes outside it.  The windows are defined on the constant 100 MHz clock every CU sees
    // (s_memrealtime): period n starts at tick n * T on all of them, nothing is exchanged to keep the waves in step.
    //   [0, ~R)   the slabs of the wave's next P tiles: 2 P loads in flight in 5 P registers, packed into P bit streams in LDS
    //   [~R, T)   the P tiles hashed and written, one after the other
    // A wave that falls more than a period behind stops waiting (the streams then simply mix, as in the static loop).
    // T comes from the host (bytes per period over the rates measured apart; NTHIP_TUNE_PH_PERIOD), 0 = no pacing.
    const uint32_t P = a.ph_tiles;
    uint32_t* const bits0 = bits;
    const uint32_t reads_per_tile = 64u / a.rpr;
    const uint64_t tile_bytes = (uint64_t)reads_per_tile * a.stride; // consecutive tiles: this many bytes further
    const uint32_t slab_bytes = (reads_per_tile - 1u) * a.stride + a.len;
    auto spin_until = [&](const uint64_t tick) {
      while (__builtin_amdgcn_s_memrealtime() < tick) __builtin_amdgcn_s_sleep(1);
    };
    const uint64_t T = a.ph_period, R = a.ph_read;
    uint64_t t_read = 0;
    if (T) t_read = (__builtin_amdgcn_s_memrealtime() / T + 1u) * T;
    constexpr uint32_t RND = 16; // slabs a lane has in flight (5 registers each); P <= RND
#if KR_DEBUG_TIMES
    uint64_t dbg[5] = {0, 0, 0, 0, 0}, dbg_t = __builtin_amdgcn_s_memrealtime();
    const uint64_t dbg_start = dbg_t;
#define KR_DBG(i) do { const uint64_t n_ = __builtin_amdgcn_s_memrealtime(); dbg[i] += n_ - dbg_t; dbg_t = n_; } while (0)
#else
#define KR_DBG(i) do { } while (0)
#endif
#ifndef KR_LOAD_NT
#define KR_LOAD_NT " sc1 nt"
#endif
    // KR_CONSEC=1 (round 3 experiment): the P tiles of a group are CONSECUTIVE in memory -- the wave reads one contiguous
    // piece of P slabs and writes one contiguous piece of P tiles, waves of a group take such pieces in turn -- instead
    // of P tiles wstride apart.  What the reads cost HBM is their number of bursts, not their bytes (2-bit packed input,
    // a quarter of the bytes, has the same memory-only floor): fewer, longer read bursts per channel.
#ifndef KR_CONSEC
#define KR_CONSEC 0
#endif
#if KR_CONSEC
    uint64_t sgrp = grp_w; // index of the wave's current piece of P tiles inside its group
    wt = grp_t0 + sgrp * P;
    auto tile_of = [&](uint32_t i) -> uint64_t { return wt + i; };
#else
    auto tile_of = [&](uint32_t i) -> uint64_t { return wt + (uint64_t)i * wstride; };
#endif
    while (wt < wt_end) {
#if KR_CONSEC
      const uint64_t left = wt_end - wt;
#else
      const uint64_t left = (wt_end - wt + wstride - 1u) / wstride;
#endif
      const uint32_t n_here = left < P ? (uint32_t)left : P;
      KR_DBG(4);
      if (T) {
        const uint64_t now = __builtin_amdgcn_s_memrealtime();
        if (now >= t_read + T) t_read = (now / T) * T; // more than a period late: rejoin the grid, no waiting
        spin_until(t_read);
      }
      KR_DBG(1);
      // ---- read window: every load of the group back to back (no branch in between: behind one hipcc waits for the
      // previous load before it issues the next), one wait, then the packing ----
      v4u v[RND];
      uint32_t w4[RND];
      uint32_t shifts[RND];
      const uint64_t base_addr = (uint64_t)a.seqs;
#pragma unroll
      for (uint32_t i = 0; i < RND; ++i) {
        const uint64_t t = tile_of(i < n_here ? i : n_here - 1u); // (a short group loads its last slab again)
        const uint64_t off = t * tile_bytes;
        const uint32_t shift = (uint32_t)((base_addr + off) & 15u);
        shifts[i] = shift;
        const uint64_t b0 = off - shift;
        const uint32_t n_vec = (shift + slab_bytes + 15u) >> 4;
        const uint32_t n_dw = (shift + slab_bytes + 3u) >> 2;
        const uint32_t i0 = lane < n_vec ? lane : 0u;
        const uint32_t jd = 256u + lane < n_dw ? 256u + lane : 0u;
#if KR_ABL_NOLOAD
        v[i] = v4u{0x41414141u, 0x43434343u, 0x47474747u, 0x54545454u};
        w4[i] = 0x41434754u;
        (void)i0; (void)jd; (void)b0;
#else
        const uint64_t sb = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b0) |
                            ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b0 >> 32)) << 32);
        const uint8_t* sbase = a.seqs + sb; // scalar base + 32-bit lane offset
        asm volatile("global_load_dwordx4 %0, %2, %4" KR_LOAD_NT "\n\tglobal_load_dword %1, %3, %4" KR_LOAD_NT
                     : "=&v"(v[i]), "=&v"(w4[i])
                     : "v"(i0 << 4), "v"(jd << 2), "s"(sbase)
                     : "memory");
#endif
      }
      // everything this wave has in flight: the loads above and, older, the stores of its previous group
      asm volatile("s_waitcnt vmcnt(0)\n\t; NTLINT_CONSUME %0 %1 %2 %3 %4 %5 %6 %7 %8 %9 %10 %11 %12 %13 %14 %15"
                   : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                     "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
                   :: "memory");
      asm volatile("; NTLINT_CONSUME %0 %1 %2 %3 %4 %5 %6 %7 %8 %9 %10 %11 %12 %13 %14 %15" : "+v"(w4[0]), "+v"(w4[1]), "+v"(w4[2]), "+v"(w4[3]), "+v"(w4[4]), "+v"(w4[5]), "+v"(w4[6]), "+v"(w4[7]),
                        "+v"(w4[8]), "+v"(w4[9]), "+v"(w4[10]), "+v"(w4[11]), "+v"(w4[12]), "+v"(w4[13]), "+v"(w4[14]), "+v"(w4[15])::"memory");
      KR_DBG(0);
#pragma unroll
      for (uint32_t i = 0; i < RND; ++i) {
        if (i < n_here) {
          const uint64_t t = tile_of(i);
          Slab sl;
          sl.shift = shifts[i];
          sl.slab_bytes = slab_bytes;
          sl.n_vec = (shifts[i] + slab_bytes + 15u) >> 4;
          sl.byte0 = 0;
          sl.runs_here = 64u;
          sl.edge = (t == 0u || (t + 1u) * reads_per_tile >= a.n_reads) ? 1u : 0u; // first / last slab of the buffer
          bits = bits0 + i * a.bits_dwords;
          if (lane < sl.n_vec) pack_vec(sl, lane, make_uint4(v[i].x, v[i].y, v[i].z, v[i].w));
          const uint32_t n_dw = (sl.shift + slab_bytes + 3u) >> 2;
          if (256u + lane < n_dw) pack_dword(sl, lane, w4[i]);
          if (lane < (uint32_t)NW + 3u) bits[sl.n_vec + lane] = 0;
        }
      }
      KR_DBG(2);
      // a non-base anywhere in the batch (found here or by another wave): the caller redoes the batch on the
      // N-aware path, so stop producing a dense stream nobody will read
      if (__ballot(bad != 0) != 0) {
        if (lane == 0) atomicOr(a.dirty, 1u);
        break;
      }
      if (__hip_atomic_load(a.dirty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
      // ---- write window ----
      if (T && R) spin_until(t_read + R);
      KR_DBG(3);
      for (uint32_t q = 0; q < n_here; ++q) {
        const uint64_t t = tile_of(q);
        const uint64_t g0 = t * 64u;
        const uint64_t runs_left = a.n_runs - g0;
        const uint32_t runs_here = runs_left < 64u ? (uint32_t)runs_left : 64u;
        const uint32_t shift = (uint32_t)((base_addr + t * tile_bytes) & 15u);
        bits = bits0 + q * a.bits_dwords;
        lds_sync();
        hash_tile(shift, runs_here, 0u);
        (void)copy_out(g0, runs_here);
        lds_sync(); // the tile is free again
      }
#if KR_CONSEC
      sgrp += wstride;
      wt = grp_t0 + sgrp * P;
#else
      wt += (uint64_t)n_here * wstride;
#endif
      t_read += T;
    }
#if KR_DEBUG_TIMES
    KR_DBG(4);
    if (lane == 0) {
      for (int i = 0; i < 5; ++i) atomicAdd((unsigned long long*)(a.dirty + 16) + i, (unsigned long long)dbg[i]);
      const unsigned long long el = __builtin_amdgcn_s_memrealtime() - dbg_start;
      atomicMax((unsigned long long*)(a.dirty + 16) + 5, el);
      atomicMin((unsigned long long*)(a.dirty + 16) + 6, el);
      atomicAdd((unsigned long long*)(a.dirty + 16) + 7, el);
    }
#endif
#undef KR_DBG
  }
#endif // KR_CHUNKED

  if (__ballot(bad != 0) != 0 && lane == 0) atomicOr(a.dirty, 1u);
}

} // namespace ntamd
