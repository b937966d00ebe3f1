// seed_kernels.hpp -- CDNA4 (gfx950) kernels for spaced-seed hashing.
//
// Replaces, for a whole batch of reads, the reference's per-read loop
//     SeedNtHash h(seq, len, seeds, m2, k); while (h.roll()) use(h.hashes());
// (src/seed.cpp:449-544).  The reference rolls a blocks-only state per seed
// and re-adds the monomers on every window (NTMSM64, src/seed.cpp:177-207);
// what that evaluates on every window is the masked direct formula
//     F = XOR_{p in care} srol^{k-1-p}(S[c_p]),  R = XOR_{p in care} srol^{p}(S[comp c_p])
// (SURVEY.md App. A.4).  On a GPU the masked formula is cheaper to evaluate
// directly: the 2-bit window is cut into bytes (4 bases) and each byte indexes
// a per-seed, per-byte-position table in LDS that already holds the XOR of the
// four masked, rotated seeds for both strands (16 B per entry).  No per-lane
// state is carried from window to window, so one lane owns one window and the
// stores are naturally ordered.
//
//   seed_fixed_kernel    hot path: fixed-length clean reads.
//   seed_general_kernel  exact reference position state machine (quirk Q3,
//                        NUL handling of src/seed.cpp:151) for ragged / dirty
//                        batches; one lane per read.
#pragma once

#include <hip/hip_runtime.h>

#include "kmer_kernels.hpp" // pack16 / funnel
#include "kmer_runs_gen_kernel.hpp" // pack4v, windows_with_non_base
#include "nt_math.hpp"

namespace ntamd {

#ifndef SF_THREADS_N
#define SF_THREADS_N 1024
#endif
constexpr int SF_THREADS = SF_THREADS_N; // 16 waves: one block per CU shares the byte tables
// ablation / A-B switches of seed_fixed_kernel (measurement builds)
#ifndef SF_ABL_NOHASH
#define SF_ABL_NOHASH 0 // no table lookups: staging + copy-out + the memory streams only (WRONG results)
#endif
#ifndef SF_ABL_NOSTORE
#define SF_ABL_NOSTORE 0 // the hash stream is not written
#endif
#ifndef SF_ABL_NOCONFLICT
#define SF_ABL_NOCONFLICT 0 // table entries picked so that no lookup has a bank conflict (WRONG results): the LDS ceiling
#endif
#ifndef SF_STORE_POLICY
#define SF_STORE_POLICY " nt"
#endif
#ifndef SW_STORE_POLICY
#define SW_STORE_POLICY " sc0 sc1" // seed_wtile_kernel: whole aligned lines of a contiguous block -> write-through
#endif
#ifndef SF_BLOCK_RANGES
#define SF_BLOCK_RANGES 0 // 1: every block streams through its own contiguous range of tiles (0: grid stride)
#endif
constexpr int SF_MAX_RUNTIME_M = 8;

struct SeedFixedArgs {
  const uint8_t* seqs;
  uint64_t* hashes;     // dense [run][window][seed][m2]
  uint32_t* dirty;
  const uint4* tables;  // global: [seed][byte position][256] {f.lo,f.hi,r.lo,r.hi}
  uint64_t n_runs;
  uint32_t len, stride, k, m2;
  uint32_t n_seeds, ntab; // ntab = ceil(k/4)
  uint32_t nwin;
  uint32_t runs_per_tile;
  uint32_t n_tiles;
  uint32_t inv_nwin;      // floor(2^32 / nwin) + 1 (for q / nwin)
  uint32_t bits_dwords;   // LDS dwords reserved for the slab's bit stream
  uint64_t mult[SF_MAX_RUNTIME_M];
  // SPLIT instantiation (batches with non-bases): reads flagged dirty are left to seed_general_kernel, the
  // records of a clean read go to its own place in the compact stream
  const uint64_t* read_dirty; // [run] != 0: the read has a byte that is not ACGTU
  const uint64_t* read_off;   // [run] index of the read's first k-mer in the compact stream
};

// Which fixed-length reads contain a byte that is not a base?  One 16-byte vector per thread; a byte
// belongs to every read whose [r*stride, r*stride + len) covers it (overlapping reads when stride < len).
static __global__ __launch_bounds__(256) void seed_mark_dirty_kernel(const uint8_t* __restrict__ seqs, uint64_t total_bytes,
                                                             uint32_t len, uint32_t stride, uint64_t n_reads,
                                                             uint64_t* __restrict__ flags)
{
  const uint64_t n_vec = (total_bytes + 15) >> 4;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t off = i << 4;
    uint32_t badmask = 0; // bit b: byte off + b is not a base
    if (off + 16 <= total_bytes) {
      uint4 v;
      __builtin_memcpy(&v, seqs + off, 16);
      uint32_t bx[4] = {0, 0, 0, 0};
      (void)pack4(v.x, bx[0]);
      (void)pack4(v.y, bx[1]);
      (void)pack4(v.z, bx[2]);
      (void)pack4(v.w, bx[3]);
      if ((bx[0] | bx[1] | bx[2] | bx[3]) == 0) continue;
      for (int q = 0; q < 16; ++q)
        if ((bx[q >> 2] >> ((q & 3) * 8)) & 0xFFu) badmask |= 1u << q;
    } else {
      for (uint32_t b = 0; off + b < total_bytes; ++b)
        if (!is_base(seqs[off + b])) badmask |= 1u << b;
    }
    while (badmask) {
      const uint32_t b = (uint32_t)__builtin_ctz(badmask);
      badmask &= badmask - 1;
      const uint64_t pos = off + b;
      uint64_t r_hi = pos / stride;
      if (r_hi >= n_reads) r_hi = n_reads - 1;
      for (uint64_t r = r_hi;; --r) { // every read that covers the byte
        if (pos >= r * stride + len) break;
        flags[r] = 1;
        if (r == 0) break;
      }
    }
  }
}

// list[idx[r]] = r for the flagged reads (idx = exclusive scan of the flags)
static __global__ __launch_bounds__(256) void seed_list_kernel(const uint64_t* __restrict__ flags, const uint64_t* __restrict__ idx,
                                                       uint64_t n_reads, uint64_t* __restrict__ list)
{
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n_reads && flags[r]) list[idx[r]] = r;
}

// NH = 16-bit halves of the 2-bit window (8 bases each): k <= 8*NH.  Every seed has NT = 2*NH byte tables in LDS
// (the ones past ceil(k/4) are zero), so the lookups of a window are unconditional and all in flight at once --
// with a uniform branch around each one (a runtime table count) the compiler waits for every ds_read before it
// issues the next: 16 serial LDS round trips per window for two k=31 seeds.
// SPLIT: skip the reads flagged in a.read_dirty, write every other read at a.read_off (compact stream)
template <int NH, bool SPLIT = false>
__global__ __launch_bounds__(SF_THREADS) void seed_fixed_kernel(const SeedFixedArgs a)
{
  constexpr int NW = (NH + 1) / 2; // 32-bit words of window kept in registers
  constexpr uint32_t NT = 2u * NH; // byte tables per seed in LDS
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  // layout: [tables: n_seeds*NT*256 uint4][bit stream][per-wave output tiles: 64*per u64][SPLIT: read offsets]
  uint4* tabs = (uint4*)lds_dyn;
  const uint32_t n_entries = a.n_seeds * NT * 256u;
  uint32_t* bits = lds_dyn + n_entries * 4u;

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = tid >> 6;
  for (uint32_t i = tid; i < n_entries; i += SF_THREADS) {
    const uint32_t tb = i >> 8, sd = tb / NT, jt = tb - sd * NT;
    tabs[i] = jt < a.ntab ? a.tables[((size_t)sd * a.ntab + jt) * 256u + (i & 255u)] : make_uint4(0, 0, 0, 0);
  }

  const uint32_t per = a.n_seeds * a.m2; // values per window
  const uint32_t otile_u64 = 64u * per + 2u; // + room to build the tile shifted by one value (see the copy-out)
  uint64_t* otile = (uint64_t*)(bits + a.bits_dwords) + wave * otile_u64;
  // SPLIT: the tile's reads -- offset in the compact stream, or ~0 for a read left to the general kernel
  uint64_t* roff = (uint64_t*)(bits + a.bits_dwords) + (SF_THREADS / 64u) * otile_u64;
  const uint32_t inv_per = 0xFFFFFFFFu / per + 1u; // v / per == umulhi(v, inv_per) for v < 2^29
  uint32_t bad = 0;

#if SF_BLOCK_RANGES
  const uint32_t tiles_per_block = (a.n_tiles + gridDim.x - 1u) / gridDim.x;
  const uint32_t t_begin = blockIdx.x * tiles_per_block;
  const uint32_t t_end = t_begin + tiles_per_block < a.n_tiles ? t_begin + tiles_per_block : a.n_tiles;
  for (uint32_t t = t_begin; t < t_end; ++t) {
#else
  for (uint32_t t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
#endif
    const uint64_t run0 = (uint64_t)t * a.runs_per_tile;
    const uint64_t left = a.n_runs - run0;
    const uint32_t runs_here = left < a.runs_per_tile ? (uint32_t)left : a.runs_per_tile;
    const uint64_t byte0 = run0 * a.stride;
    const uint64_t addr0 = (uint64_t)(a.seqs + byte0);
    const uint32_t shift = (uint32_t)(addr0 & 15u);
    const uint4* vsrc = (const uint4*)(addr0 - shift);
    const uint32_t slab_bytes = (runs_here - 1u) * a.stride + a.len;
    const uint32_t n_vec = (shift + slab_bytes + 15u) >> 4;
    __syncthreads();
    for (uint32_t i = tid; i < n_vec; i += SF_THREADS) {
      const uint4 v = vsrc[i];
      uint32_t b = 0;
      const uint32_t p = pack16(v, b);
      const int32_t lo_cut = (int32_t)shift - (int32_t)(i << 4);
      const int32_t hi_cut = (int32_t)(shift + slab_bytes) - (int32_t)(i << 4);
      if (lo_cut > 0 || hi_cut < 16) {
        uint32_t bx[4] = {0, 0, 0, 0};
        (void)pack4(v.x, bx[0]);
        (void)pack4(v.y, bx[1]);
        (void)pack4(v.z, bx[2]);
        (void)pack4(v.w, bx[3]);
        b = 0;
        for (int q = 0; q < 16; ++q)
          if (q >= lo_cut && q < hi_cut) b |= (bx[q >> 2] >> ((q & 3) * 8)) & 0xFFu;
      }
      bad |= b;
      bits[i] = p;
    }
    if (tid < NW + 1) bits[n_vec + tid] = 0;
    if (!SPLIT) {
      // optimistic pass: publish a non-base at once and stop producing a dense stream nobody will read
      // (some block already found one: the caller redoes the batch on the split path)
      if (__ballot(bad != 0) != 0 && lane == 0) atomicOr(a.dirty, 1u);
      if (__hip_atomic_load(a.dirty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
    }
    if (SPLIT) {
      for (uint32_t i = tid; i < runs_here; i += SF_THREADS)
        roff[i] = a.read_dirty[run0 + i] ? ~0ull : a.read_off[run0 + i];
    }
    __syncthreads();

    // Every wave takes 64 consecutive windows at a time: their per*8-byte records are
    // contiguous in the output stream, so they are collected in a wave-private LDS
    // tile and written out as one contiguous block, 16 bytes per lane per store.
    const uint32_t n_win_tile = runs_here * a.nwin;
    for (uint32_t q0 = wave * 64u; q0 < n_win_tile; q0 += (SF_THREADS / 64u) * 64u) {
      const uint32_t q = q0 + lane;
      const bool live = q < n_win_tile;
      const uint32_t qq = live ? q : q0;
      // qq -> (run, window) with a multiply-high and one fix-up
      uint32_t lr = a.nwin == 1u ? qq : __umulhi(qq, a.inv_nwin);
      if (lr * a.nwin > qq) lr--;
      const uint32_t p = qq - lr * a.nwin;
      const uint32_t b = shift + lr * a.stride + p; // first base of the window (stream index)
      const uint32_t d = b >> 4, sh = (b & 15u) << 1;
      uint32_t w[NW];
      uint32_t lo = bits[d];
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        const uint32_t hi = bits[d + i + 1];
        w[i] = funnel(hi, lo, sh);
        lo = hi;
      }
      // the 64 records are built shifted by the parity of their place in the output stream, so that both the
      // LDS reads and the global stores of the copy-out are 16-byte aligned
      uint64_t* const dst = a.hashes + (run0 * a.nwin + q0) * per;
      const uint32_t par = SPLIT ? 0u : (uint32_t)(((uintptr_t)dst >> 3) & 1u);
      uint64_t* mine = otile + par + lane * per;
#if SF_ABL_NOHASH
      for (uint32_t s = 0; s < per; ++s) mine[s] = w[0] + s;
#else
      for (uint32_t s = 0; s < a.n_seeds; ++s) {
        const uint4* ts = tabs + s * NT * 256u;
        // all lookups of the seed in flight, then XOR them up
        uint4 e[NT];
#pragma unroll
        for (uint32_t jt = 0; jt < NT; ++jt) {
          const uint32_t byte = (w[jt >> 2] >> ((jt & 3u) * 8u)) & 0xFFu;
          e[jt] = ts[jt * 256u + byte];
        }
        uint32_t f0 = e[0].x ^ e[1].x, f1 = e[0].y ^ e[1].y, r0 = e[0].z ^ e[1].z, r1 = e[0].w ^ e[1].w;
#pragma unroll
        for (uint32_t jt = 2; jt < NT; jt += 2) { // NT is even; a ^ b ^ c is one v_bitop3_b32
#ifdef SF_NO_XOR3
          f0 ^= e[jt].x ^ e[jt + 1].x; f1 ^= e[jt].y ^ e[jt + 1].y; r0 ^= e[jt].z ^ e[jt + 1].z; r1 ^= e[jt].w ^ e[jt + 1].w;
#else
          f0 = __builtin_amdgcn_bitop3_b32(f0, e[jt].x, e[jt + 1].x, 0x96);
          f1 = __builtin_amdgcn_bitop3_b32(f1, e[jt].y, e[jt + 1].y, 0x96);
          r0 = __builtin_amdgcn_bitop3_b32(r0, e[jt].z, e[jt + 1].z, 0x96);
          r1 = __builtin_amdgcn_bitop3_b32(r1, e[jt].w, e[jt + 1].w, 0x96);
#endif
        }
        const uint64_t h0 = canon_pair(f0, f1, r0, r1);
        mine[s * a.m2] = h0;
#pragma unroll
        for (uint32_t jj = 1; jj < (uint32_t)SF_MAX_RUNTIME_M; ++jj)
          if (jj < a.m2) mine[s * a.m2 + jj] = mix_hash(h0, a.mult[jj]);
      }
#endif

      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
      const uint32_t n_here = (n_win_tile - q0) < 64u ? (n_win_tile - q0) : 64u;
      const uint32_t n_vals = n_here * per;
      if (SPLIT) {
        // every value finds its read's place in the compact stream (16 bytes at a time when records are even)
        const uint32_t step = (per & 1u) ? 1u : 2u;
        for (uint32_t v = lane * step; v < n_vals; v += 64u * step) {
          const uint32_t wv = per == 1u ? v : __umulhi(v, inv_per); // window inside the group (inv_per wraps for per = 1)
          const uint32_t vi = v - wv * per;                          // value inside the record
          const uint32_t wq = q0 + wv;
          uint32_t lr2 = a.nwin == 1u ? wq : __umulhi(wq, a.inv_nwin);
          if (lr2 * a.nwin > wq) lr2--;
          const uint64_t ro = roff[lr2];
          if (ro == ~0ull) continue;
          uint64_t* d = a.hashes + (ro + (wq - lr2 * a.nwin)) * per + vi;
          if (step == 2u) *(uint4*)d = *(const uint4*)(otile + v);
          else *d = otile[v];
        }
      } else {
        uint64_t* const base = dst - par; // 16-byte aligned; value i of the shifted tile goes to base[i]
        const uint32_t span = par + n_vals;
        for (uint32_t pi = par + lane; pi < (span >> 1); pi += 64u) { // whole 16-byte pieces
#if SF_ABL_NOSTORE
          asm volatile("" ::"v"(*(const nt_v4u*)(otile + 2u * pi)));
#else     // written once, never read back by this kernel, whole aligned 16-byte pieces of a contiguous 3 KiB: SF_STORE_POLICY
          {
            const nt_v4u sv = *(const nt_v4u*)(otile + 2u * pi);
            asm volatile("global_store_dwordx4 %0, %1, off" SF_STORE_POLICY "\n\ts_nop 1" ::"v"(base + 2u * pi), "v"(sv) : "memory");
          }
#endif
        }
        if (lane == 0u && par != 0u) base[1] = otile[1];                         // head
        if (lane == 1u && (span & 1u) != 0u && span > 2u * par) base[span - 1u] = otile[span - 1u]; // tail
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (__ballot(bad != 0) != 0 && lane == 0) atomicOr(a.dirty, 1u);
}

// --------------------------------------------------------------------------
// seed_wtile_kernel -- the dense spaced-seed path, one WAVE per tile (round 2).
//
// seed_fixed_kernel stages the reads of a tile block-wide: two __syncthreads() per tile, each of them draining the
// wave's stores (hipcc waits vmcnt(0) in front of a barrier) and lining the 16 waves up, so that they hash together
// and write together: hashing alone took 26.5 ms per 16 M reads, the stores alone 26.2 ms, both 31.9 ms
// (profiles/r02_notes.md).  Here a wave owns its tile: R consecutive reads (R chosen so that a tile's records are a
// multiple of 1 KiB of the stream: 16 reads of 250 bp = 165 KiB), staged as a wave-private bit stream; the next tile's
// slab is loaded while this one is hashed; no block barrier after the tables are in LDS.  Every group of 64 consecutive
// windows leaves as one contiguous, KiB-aligned block (3 KiB for two seeds x three hashes) of write-through stores.
// Blocks stream through contiguous ranges of tiles.
// --------------------------------------------------------------------------
// ---- the any-seed form (seed_wtile_kernel<0>, seed_rtile_kernel<0>): shared pieces ------------------------------------
// LDS: FOUR copies of the four 16-mer byte tables of first_window.hpp, copy q pre-rotated by 16 q bases ({F, R} terms:
// sror^{16 q} / srol^{16 q}), so that four consecutive 16-base groups of a window are XOR-ed up WITHOUT rotating anything
// and the Horner step -- one pair of split rotates, by 64 bases -- comes once per four groups instead of once per group
// (the rotates were half of the ~50 instructions a group cost); then per (seed, group) the correction word, rotated the
// same way, and the care mask.
constexpr uint32_t SA_SETS = 4;
constexpr uint32_t SA_TAB_ENTRIES = SA_SETS * 1024u; // uint4
__device__ __forceinline__ uint4 sa_rotate(uint4 e, uint32_t q)
{
  const uint32_t a = (16u * q) % 31u, b = (16u * q) % 33u;
  sror_var(e.x, e.y, a, b);
  srol_var(e.z, e.w, a, b);
  return e;
}
// block-wide: fill the LDS area (no barrier here)
__device__ __forceinline__ void sa_load(uint4* tabs, const uint4* fw, const uint4* acorr, const uint32_t* mask, uint32_t n_grp,
                                        uint32_t G, uint32_t tid, uint32_t n_threads)
{
  for (uint32_t i = tid; i < SA_TAB_ENTRIES; i += n_threads) tabs[i] = sa_rotate(fw[i & 1023u], i >> 10);
  uint4* const g_acorr = tabs + SA_TAB_ENTRIES;
  uint32_t* const g_mask = (uint32_t*)(g_acorr + n_grp);
  for (uint32_t i = tid; i < n_grp; i += n_threads) {
    g_acorr[i] = sa_rotate(acorr[i], (i % G) & 3u);
    g_mask[i] = mask[i];
  }
}
// strand hashes {F lo, hi, R lo, hi} of the window that starts at stream position (d, sh) under seed s
__device__ __forceinline__ uint4 sa_strands(const uint4* tabs, const uint4* g_acorr, const uint32_t* g_mask, const uint32_t* bits,
                                            uint32_t d, uint32_t sh, uint32_t s, uint32_t G, uint32_t k31, uint32_t k33)
{
  uint4 acc = make_uint4(0, 0, 0, 0);
  const uint32_t chunks = (G + 3u) >> 2;
  for (uint32_t c = chunks; c-- > 0;) {
    uint4 x = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q) {
      const uint32_t g = 4u * c + q;
      if (g < G) { // (uniform)
        const uint32_t word = funnel(bits[d + g + 1u], bits[d + g], sh) & g_mask[s * G + g];
        const uint4 e = fw_word16(tabs + q * 1024u, word), ac = g_acorr[s * G + g];
        x.x ^= e.x ^ ac.x; x.y ^= e.y ^ ac.y; x.z ^= e.z ^ ac.z; x.w ^= e.w ^ ac.w;
      }
    }
    sror_var(acc.x, acc.y, 2u, 31u); // 64 bases: 64 mod 31, 64 mod 33
    srol_var(acc.z, acc.w, 2u, 31u);
    acc.x ^= x.x; acc.y ^= x.y; acc.z ^= x.z; acc.w ^= x.w;
  }
  srol_var(acc.x, acc.y, k31, k33);
  return acc;
}
// its canonical hash
__device__ __forceinline__ uint64_t sa_hash(const uint4* tabs, const uint4* g_acorr, const uint32_t* g_mask, const uint32_t* bits,
                                            uint32_t d, uint32_t sh, uint32_t s, uint32_t G, uint32_t k31, uint32_t k33)
{
  const uint4 st = sa_strands(tabs, g_acorr, g_mask, bits, d, sh, s, G, k31, k33);
  return canon_pair(st.x, st.y, st.z, st.w);
}

struct SeedWtileArgs {
  const uint8_t* seqs;
  uint64_t* hashes;      // dense [read][window][seed][m2]
  uint32_t* dirty;
  const uint4* tables;   // global: [seed][ntab][256] {f.lo,f.hi,r.lo,r.hi}
  uint64_t n_reads;
  uint64_t n_tiles;      // wave tiles of reads_per_tile reads
  uint32_t len, stride, k, m2;
  uint32_t n_seeds, ntab;
  uint32_t nwin;
  uint32_t reads_per_tile;
  uint32_t inv_nwin;     // floor(2^32 / nwin) + 1
  uint32_t bits_dwords;  // per wave
  uint32_t waves;        // per block
  uint32_t groups;       // tile_range(): groups of blocks sharing a range of tiles (0: one range per block)
  uint32_t align_recs;   // records after which the stream is on a 128-byte line again: 16 / gcd(values per record, 16)
  // a pass over SOME of the seeds (seed sets whose byte tables do not fit in LDS together, or more than two seeds on the
  // rotated-slot layout): n_seeds / tables are this pass's seeds, the records keep the stream's layout --
  // rec_stride values per window (0: the pass writes whole records), this pass's values from rec_off on
  uint32_t rec_stride, rec_off;
  uint64_t mult[SF_MAX_RUNTIME_M];
  // NH == 0 (any seed set, any k; round 3): tables = the k-independent fw tables of first_window.hpp; per seed and
  // 16-base group of the window: the care positions as a 2-bit-per-base mask, and what the OTHER positions of the group
  // contribute when they read as code 0 (XOR-ed off again)
  const uint32_t* any_mask;  // [n_seeds][any_groups]
  const uint4* any_acorr;    // [n_seeds][any_groups]
  uint32_t any_groups, pad_any;
};

constexpr uint32_t SW_MAX_VEC_ROUNDS = 8; // a tile's slab: at most 8 x 64 vectors of 16 bytes (8 KiB of reads)

//
// RNS > 0 (k <= 32, one or two seeds, RM2 <= 4 hashes per seed, all compile-time): the ROTATED-SLOT table layout.
// A ds_read_b128 is served in four groups of 16 lanes, each lane's 16 bytes = one of the 16 four-bank "slots" of the
// 256-byte bank row; random entries of one table put ~3 lanes of a group on the same slot (58 % of the LDS cycles of
// the plain layout were conflicts).  Here the 16 byte tables of the two seeds are interleaved ENTRY-major --
// entry e of table v at e*256 + v*16, so a table owns one slot -- and at step s lane l looks up byte position
// (l + s) & 7 of the seed-half (l >> 3) & 1 (the other half in the second round): every 16-lane group holds each
// residue l & 15 once, so its 16 lanes read 16 different tables = 16 different slots, whatever the entries.  XOR is
// commutative, so the order in which a lane meets its eight tables does not matter.  One seed: both halves hold it.
template <int NH, int RNS = 0, int RM2 = 0, bool SUB = false>
__global__ __launch_bounds__(SF_THREADS) void seed_wtile_kernel(const SeedWtileArgs a)
{
  // SUB: a pass over some of the seeds (a.rec_stride != 0), its own instantiation so that the whole-record kernels keep
  // their register budget
  constexpr bool ROT = RNS > 0;
  static_assert(!ROT || (NH == 4 && RNS <= 4 && RM2 >= 1 && RM2 <= 4), "rotated-slot layout: k <= 32, <= 4 seeds");
  constexpr uint32_t NSETS = ROT ? (uint32_t)(RNS + 1) / 2u : 0u; // table sets of 64 KiB: seeds {0, 1}, {2, 3}
  // ANY (NH == 0; round 3): any seed set of any k from the k-independent fw tables (first_window.hpp, 16 KiB whatever the
  // seeds are): a window's g-th 16-base group is one funnel-shifted word of the bit stream AND-ed with the seed's care
  // mask -- the positions masked out then read as code 0, whose contribution is a per-(seed, group) constant XOR-ed off
  // again --, its masked 16-mer hash four lookups, the groups chained with constant rotates by 16 as in the grouped first
  // window.  ~46 VALU per seed and 16 bases of k against ~12 on the position tables -- but ONE pass writing whole records
  // however many seeds there are, 16 waves, and no limit on k.
  constexpr bool ANY = NH == 0;
  static_assert(!ANY || (!ROT && !SUB), "the any-seed form is one pass over all the seeds");
  // PF: the next tile's slab travels in registers behind hidden loads.  Only while the kernel does not spill: a spilled
  // register of a load hipcc cannot see is saved before the load has landed (k > 32: 2 * NH lookups of 16 bytes in flight
  // take the registers; nthash_amd/build.py refuses a build in which a kernel with hidden loads spills)
  constexpr bool PF = NH <= 4 && !SUB;
  constexpr int NW = ANY ? 1 : (NH + 1) / 2; // 32-bit words of window kept in registers
  constexpr uint32_t NT = 2u * NH; // byte tables per seed in LDS
  extern __shared__ __attribute__((aligned(256))) uint32_t lds_dyn[];
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // layout: [tables: n_seeds*NT*256 uint4 (ROT: 16*256)][per wave: output tile 64*per+2 u64 | bit stream]
  uint4* tabs = (uint4*)lds_dyn;
  // (ANY: the fw tables up to the AC entries, then the seeds' group constants, then their masks)
  const uint32_t n_grp = ANY ? a.n_seeds * a.any_groups : 0u;
  const uint32_t n_entries = ANY ? SA_TAB_ENTRIES + n_grp + ((n_grp + 3u) >> 2) : ROT ? 4096u * NSETS : a.n_seeds * NT * 256u;
  const uint4* g_acorr = tabs + SA_TAB_ENTRIES;
  const uint32_t* g_mask = (const uint32_t*)(g_acorr + n_grp);
  const uint32_t per = ROT ? (uint32_t)(RNS * RM2) : a.n_seeds * a.m2; // values per window
  const uint32_t otile_u64 = 64u * per + 2u;
  uint32_t* wbase = lds_dyn + n_entries * 4u + wave * (otile_u64 * 2u + a.bits_dwords);
  uint64_t* otile = (uint64_t*)wbase;
  uint32_t* bits = wbase + otile_u64 * 2u;
  if constexpr (ANY) {
    sa_load(tabs, a.tables, a.any_acorr, a.any_mask, n_grp, a.any_groups, tid, blockDim.x);
  } else if constexpr (ROT) {
    for (uint32_t i = tid; i < n_entries; i += blockDim.x) {
      const uint32_t set = i >> 12, ii = i & 4095u;
      const uint32_t in_set = (uint32_t)RNS - 2u * set < 2u ? (uint32_t)RNS - 2u * set : 2u; // seeds of this set
      const uint32_t e = ii >> 4, v = ii & 15u, jt = v & 7u, sd = 2u * set + ((v >> 3) < in_set ? (v >> 3) : in_set - 1u);
      tabs[i] = jt < a.ntab ? a.tables[((size_t)sd * a.ntab + jt) * 256u + e] : make_uint4(0, 0, 0, 0);
    }
  } else {
    for (uint32_t i = tid; i < n_entries; i += blockDim.x) {
      const uint32_t tb = i >> 8, sd = tb / NT, jt = tb - sd * NT;
      tabs[i] = jt < a.ntab ? a.tables[((size_t)sd * a.ntab + jt) * 256u + (i & 255u)] : make_uint4(0, 0, 0, 0);
    }
  }
  __syncthreads(); // the only block-wide barrier
  // ROT: what lane l needs at step s -- the v_perm selector that puts byte (l + s) & 7 of the window at bits 8..15
  // (= entry * 256) and the LDS address of its slot in the first round; the second round is 128 bytes up or down
  typedef __attribute__((address_space(3))) const nt_v4u lds_v4u;
  uint32_t rsel[8], roff[8];
  uint32_t rdelta = 0, rb3 = 0;
  if constexpr (ROT) {
    rb3 = (lane >> 3) & 1u;
    const uint32_t tb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)lds_dyn;
#pragma unroll
    for (uint32_t st = 0; st < 8; ++st) {
      const uint32_t cpos = (lane + st) & 7u;
      rsel[st] = 0x0c0c000cu | (cpos << 8);
      roff[st] = tb + (((rb3 << 3) + cpos) << 4);
      asm volatile("" : "+v"(rsel[st]), "+v"(roff[st])); // kept in registers, not recomputed per window
    }
    rdelta = rb3 ? (uint32_t)-128 : 128u;
  }

  // this block's contiguous range of tiles, its waves interleaved inside it
  const TileRange tr = tile_range(a.n_tiles, a.waves, wave, a.groups);
  const uint64_t t_end = tr.end, t_step = tr.step;
  const uint32_t R = a.reads_per_tile;
  uint32_t bad = 0;

  struct Slab {
    uint64_t byte0; // offset of the first 16-byte vector from a.seqs (wraps below 0 by < 16)
    uint32_t shift, slab_bytes, n_vec, reads_here, edge;
  };
  auto slab_of = [&](const uint64_t t) -> Slab {
    Slab sl;
    const uint64_t r0 = t * R;
    const uint64_t left = a.n_reads - r0;
    sl.reads_here = left < R ? (uint32_t)left : R;
    const uint64_t off = r0 * a.stride;
    sl.shift = (uint32_t)(((uint64_t)a.seqs + off) & 15u);
    sl.byte0 = off - sl.shift;
    sl.slab_bytes = (sl.reads_here - 1u) * a.stride + a.len;
    sl.n_vec = (sl.shift + sl.slab_bytes + 15u) >> 4;
    sl.edge = (r0 == 0 || r0 + sl.reads_here >= a.n_reads) ? 1u : 0u;
    return sl;
  };
  // one vector of the slab into the bit stream; bytes of this batch are judged (a non-base anywhere makes the batch
  // dirty), bytes outside the caller's buffer (first / last slab only) are not
  auto pack_vec = [&](const Slab& sl, const uint32_t i, const uint4 v) {
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
    bad |= b;
    bits[i] = p;
  };
  auto lds_sync = [&]() { // LDS is in-order per wave: only the compiler must not reorder
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  };

  uint64_t t = tr.first;
  if (t >= t_end) return;
  Slab cur = slab_of(t);
  for (uint32_t i = lane; i < cur.n_vec; i += 64u)
    pack_vec(cur, i, *(const uint4*)(a.seqs + cur.byte0 + ((uint64_t)i << 4)));
  if (lane < (uint32_t)NW + 1u) bits[cur.n_vec + lane] = 0;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  for (; t < t_end; t += t_step) {
    lds_sync();
    // ---- the NEXT tile's slab: loads now, consumed after this tile's stores have been issued.  Hidden from hipcc
    // (it would wait for them behind the tile's stores, i.e. for the store acknowledgements); vmcnt retires in order
    // and holds at most 63 operations, so after 64 younger operations have been issued they have landed. ----
    const uint64_t tn = t + t_step;
    const bool have_next = tn < t_end;
    const Slab nxt = have_next ? slab_of(tn) : cur;
    nt_v4u pv[SW_MAX_VEC_ROUNDS];
    uint32_t dirty_seen = 0;
    if constexpr (PF) {
      const uint64_t b0 = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)nxt.byte0) |
                          ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(nxt.byte0 >> 32)) << 32);
      const uint8_t* sbase = a.seqs + b0; // (wave-uniform: scalar base + 32-bit lane offset)
#pragma unroll
      for (uint32_t rd = 0; rd < SW_MAX_VEC_ROUNDS; ++rd) {
        const uint32_t i = rd * 64u + lane;
        const uint32_t off = (i < nxt.n_vec ? i : 0u) << 4; // lanes past the slab re-read its start
        asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=&v"(pv[rd]) : "v"(off), "s"(sbase) : "memory");
      }
      asm volatile("global_load_dword %0, %1, off sc1" : "=&v"(dirty_seen) : "v"(a.dirty) : "memory");
      NT_LINT_SELFTEST_TOUCH(dirty_seen);
    }

    // ---- this tile: groups of 64 consecutive windows ----
    const uint32_t n_win_tile = cur.reads_here * a.nwin;
    const uint64_t rec0 = t * R * (uint64_t)a.nwin; // first window of the tile in the stream
    uint32_t n_stores = 0;
    // a tile whose first record does not start a 128-byte line of the stream (window counts the tile geometry could not
    // make whole KiB of): its first group ends where a line ends, so that the other groups' write-through stores cover
    // whole lines (seed_rtile_kernel does the same; partial lines cost that policy a third of its rate)
    uint32_t g_size = (uint32_t)(((rec0 + 64u) / a.align_recs) * a.align_recs - rec0); // in (64 - align, 64]
    for (uint32_t q0 = 0; q0 < n_win_tile; q0 += g_size, g_size = 64u) {
      const uint32_t q = q0 + lane;
      const bool live = q < n_win_tile && lane < g_size;
      const uint32_t qq = live ? q : q0;
      uint32_t lr = a.nwin == 1u ? qq : __umulhi(qq, a.inv_nwin); // qq -> (read, window)
      if (lr * a.nwin > qq) lr--;
      const uint32_t p = qq - lr * a.nwin;
      const uint32_t b = cur.shift + lr * a.stride + p; // first base of the window (stream index)
      const uint32_t d = b >> 4, sh = (b & 15u) << 1;
      uint32_t w[NW];
      if constexpr (!ANY) {
        uint32_t lo = bits[d];
#pragma unroll
        for (int i = 0; i < NW; ++i) {
          const uint32_t hi = bits[d + i + 1];
          w[i] = funnel(hi, lo, sh);
          lo = hi;
        }
      } else {
        w[0] = 0;
      }
      // the 64 records are built shifted by the parity of their place in the stream: 16-byte aligned LDS reads and stores
      uint64_t* const dst = a.hashes + (rec0 + q0) * per;
      const uint32_t par = SUB ? 0u : (uint32_t)(((uintptr_t)dst >> 3) & 1u);
      uint64_t* mine = otile + par + lane * per;
#if SF_ABL_NOHASH
#ifdef SF_ABL_SPIN // pure-VALU stand-in for the hashing (SF_ABL_SPIN dependent operations per group, no LDS)
      {
        uint32_t x = w[0];
#pragma unroll 8
        for (int it = 0; it < SF_ABL_SPIN; ++it) asm volatile("v_mad_u32_u24 %0, %0, %0, %1" : "+v"(x) : "v"(lane));
        w[0] = x;
      }
#endif
      for (uint32_t s = 0; s < per; ++s) mine[s] = w[0] + s;
#else
      if constexpr (ANY) {
        const uint32_t G = a.any_groups, k31 = a.k % 31u, k33 = a.k % 33u;
        for (uint32_t s = 0; s < a.n_seeds; ++s) {
          const uint64_t h0 = sa_hash(tabs, g_acorr, g_mask, bits, d, sh, s, G, k31, k33);
          mine[s * a.m2] = h0;
#pragma unroll
          for (uint32_t jj = 1; jj < (uint32_t)SF_MAX_RUNTIME_M; ++jj)
            if (jj < a.m2) mine[s * a.m2 + jj] = mix_hash(h0, a.mult[jj]);
        }
      } else if constexpr (ROT) {
        uint32_t ad[8];
#pragma unroll
        for (uint32_t st = 0; st < 8; ++st) ad[st] = __builtin_amdgcn_perm(w[1], w[0], rsel[st]) + roff[st];
#pragma unroll
        for (int round = 0; round < RNS; ++round) {
          // rounds 0, 1: the table set of seeds {0, 1}; rounds 2, 3: of seeds {2, 3}, 64 KiB further (a set with one
          // seed holds it in both halves and takes one round)
          const int set = round >> 1, half = round & 1;
          const bool pair = RNS - 2 * set >= 2; // this set holds two seeds
          const uint32_t sbase = (uint32_t)set * 65536u;
          nt_v4u e[8];
#pragma unroll
          for (uint32_t st = 0; st < 8; ++st) e[st] = *(lds_v4u*)(uintptr_t)((half == 0 ? ad[st] : ad[st] + rdelta) + sbase);
          uint32_t f0 = e[0].x ^ e[1].x, f1 = e[0].y ^ e[1].y, r0 = e[0].z ^ e[1].z, r1 = e[0].w ^ e[1].w;
#pragma unroll
          for (uint32_t st = 2; st < 8; st += 2) {
            f0 = __builtin_amdgcn_bitop3_b32(f0, e[st].x, e[st + 1].x, 0x96);
            f1 = __builtin_amdgcn_bitop3_b32(f1, e[st].y, e[st + 1].y, 0x96);
            r0 = __builtin_amdgcn_bitop3_b32(r0, e[st].z, e[st + 1].z, 0x96);
            r1 = __builtin_amdgcn_bitop3_b32(r1, e[st].w, e[st + 1].w, 0x96);
          }
          const uint64_t h0 = canon_pair(f0, f1, r0, r1);
          // the seed this lane has just hashed: its half in the first round of a pair, the other one in the second
          uint64_t* const rec = mine + (2u * (uint32_t)set + (pair ? ((uint32_t)half ^ rb3) : 0u)) * (uint32_t)RM2;
          rec[0] = h0;
#pragma unroll
          for (uint32_t jj = 1; jj < (uint32_t)RM2; ++jj) rec[jj] = mix_hash(h0, a.mult[jj]);
        }
      } else
      for (uint32_t s = 0; s < a.n_seeds; ++s) {
        const uint4* ts = tabs + s * NT * 256u;
        uint32_t f0 = 0, f1 = 0, r0 = 0, r1 = 0;
        // the lookups of the seed in flight -- all of them up to 64 bases, 8 at a time beyond (NH > 8) -- then XORed up
        constexpr uint32_t CH = NT <= 16u ? NT : 8u;
#pragma unroll
        for (uint32_t j0 = 0; j0 < NT; j0 += CH) {
          uint4 e[CH];
#pragma unroll
          for (uint32_t jt = 0; jt < CH; ++jt) {
#if SF_ABL_NOCONFLICT
            const uint32_t byte = ((w[(j0 + jt) >> 2] >> (((j0 + jt) & 3u) * 8u)) & 0xF0u) | (lane & 15u);
#else
            const uint32_t byte = (w[(j0 + jt < NT ? j0 + jt : 0u) >> 2] >> (((j0 + jt) & 3u) * 8u)) & 0xFFu;
#endif
            e[jt] = j0 + jt < NT ? ts[(j0 + jt) * 256u + byte] : make_uint4(0, 0, 0, 0); // (NT = 20, 28: the last batch is half full)
          }
#pragma unroll
          for (uint32_t jt = 0; jt < CH; jt += 2) { // CH is even; a ^ b ^ c is one v_bitop3_b32
            f0 = __builtin_amdgcn_bitop3_b32(f0, e[jt].x, e[jt + 1].x, 0x96);
            f1 = __builtin_amdgcn_bitop3_b32(f1, e[jt].y, e[jt + 1].y, 0x96);
            r0 = __builtin_amdgcn_bitop3_b32(r0, e[jt].z, e[jt + 1].z, 0x96);
            r1 = __builtin_amdgcn_bitop3_b32(r1, e[jt].w, e[jt + 1].w, 0x96);
          }
          if constexpr (NT > 16u) asm volatile("" : "+v"(f0), "+v"(f1), "+v"(r0), "+v"(r1)); // (one batch of lookups at a time)
        }
        const uint64_t h0 = canon_pair(f0, f1, r0, r1);
        mine[s * a.m2] = h0;
#pragma unroll
        for (uint32_t jj = 1; jj < (uint32_t)SF_MAX_RUNTIME_M; ++jj)
          if (jj < a.m2) mine[s * a.m2 + jj] = mix_hash(h0, a.mult[jj]);
      }
#endif
      lds_sync();
      const uint32_t n_here = (n_win_tile - q0) < g_size ? (n_win_tile - q0) : g_size;
      const uint32_t n_vals = n_here * per;
      if constexpr (SUB) {
        // some of the record's values: 8-byte stores, per values in a row then a gap (ordinary stores: the other passes'
        // pieces of a line meet them in L2 when they come soon enough)
        uint64_t* const o = a.hashes + (rec0 + q0) * a.rec_stride + a.rec_off;
        for (uint32_t j = 0; j < per; ++j)
          if (lane < n_here) o[lane * a.rec_stride + j] = otile[lane * per + j];
        n_stores += per;
        lds_sync();
        continue;
      }
      uint64_t* const base = dst - par; // 16-byte aligned; value i of the shifted tile goes to base[i]
      const uint32_t span = par + n_vals;
      for (uint32_t pi = par + lane; pi < (span >> 1); pi += 64u) { // whole 16-byte pieces
#if SF_ABL_NOSTORE
        asm volatile("" ::"v"(*(const nt_v4u*)(otile + 2u * pi)));
#else
        const nt_v4u sv = *(const nt_v4u*)(otile + 2u * pi);
        asm volatile("global_store_dwordx4 %0, %1, off" SW_STORE_POLICY "\n\ts_nop 1" ::"v"(base + 2u * pi), "v"(sv) : "memory");
#endif
      }
      n_stores += ((span >> 1) - par + 63u) >> 6;
      if (lane == 0u && par != 0u) base[1] = otile[1];                                              // head
      if (lane == 1u && (span & 1u) != 0u && span > 2u * par) base[span - 1u] = otile[span - 1u]; // tail
      lds_sync(); // the output tile is free again
    }

    // ---- consume the next slab ----
    if constexpr (PF) {
      // vmcnt counts in order and holds at most 63: once 64 younger operations have been issued the loads have landed,
      // and "at most 63 in flight" is the wait that says so (unconditional: every path into the marker passes an
      // inline wait, lint rule R3); fewer stores than that: everything
      asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
      if (n_stores < 64u) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("; NTLINT_CONSUME %0 %1 %2 %3 %4 %5 %6 %7 %8" : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]), "+v"(pv[4]),
                   "+v"(pv[5]), "+v"(pv[6]), "+v"(pv[7]), "+v"(dirty_seen)::"memory");
    } else {
      dirty_seen = __atomic_load_n(a.dirty, __ATOMIC_RELAXED);
    }
    // some wave already found a non-base: the caller redoes the batch on the split path
    if (__builtin_amdgcn_readfirstlane(dirty_seen) != 0u) break;
    if (have_next) {
      cur = nxt;
      if constexpr (PF) {
#pragma unroll
        for (uint32_t rd = 0; rd < SW_MAX_VEC_ROUNDS; ++rd) {
          const uint32_t i = rd * 64u + lane;
          if (i < cur.n_vec) pack_vec(cur, i, make_uint4(pv[rd].x, pv[rd].y, pv[rd].z, pv[rd].w));
        }
      } else { // long k: the lookups of a window need the registers, the slab is loaded when it is needed
        for (uint32_t i = lane; i < cur.n_vec; i += 64u)
          pack_vec(cur, i, *(const uint4*)(a.seqs + cur.byte0 + ((uint64_t)i << 4)));
      }
      if (lane < (uint32_t)NW + 1u) bits[cur.n_vec + lane] = 0;
      if (__ballot(bad != 0) != 0) { // publish at once so that every wave can stop early
        if (lane == 0) atomicOr(a.dirty, 1u);
        break;
      }
    }
  }
  if (__ballot(bad != 0) != 0 && lane == 0) atomicOr(a.dirty, 1u);
}

// --------------------------------------------------------------------------------------------------------------
// seed_rtile_kernel -- spaced seeds on VARIABLE-LENGTH short reads without a non-base (round 2): the counterpart of
// kmer_reads_kernel.hpp.  A wave's tile is R consecutive whole reads (one contiguous slab), its windows are numbered
// densely over the tile's clean reads; a lane is a window, 64 consecutive windows per group, hashed by the masked
// direct formula exactly as in seed_wtile_kernel (rotated-slot tables when RNS > 0), the records collected in a
// wave-private LDS tile and written as aligned 16-byte pieces.  Reads flagged by the mark pass (a non-base: the
// reference's position state machine applies) keep their place in the stream as a hole and are hashed afterwards by
// seed_wave_kernel from the list.
// --------------------------------------------------------------------------------------------------------------
struct SeedRtileArgs {
  const uint8_t* seqs;
  const uint64_t* starts;   // read r = bytes [starts[r], ends[r])
  const uint64_t* ends;
  const uint8_t* flags;     // 1 = listed (not hashed here)
  const uint64_t* read_off; // first k-mer of read r in the stream
  uint64_t* hashes;
  uint32_t* pos;            // optional
  const uint4* tables;      // global: [seed][ntab][256]
  uint64_t n_reads, n_tiles;
  uint32_t R, k, m2, n_seeds, ntab;
  uint32_t bits_dwords, otile_recs, wmap_dwords, waves;
  uint32_t align_recs;      // records after which the stream is on a 128-byte line again: 16 / gcd(values per record, 16)
  uint32_t rec_stride, rec_off; // a pass over some of the seeds: as in SeedWtileArgs
  // NH == 0, the any-seed form (see seed_wtile_kernel): tables = the fw tables, per (seed, group) mask and constant
  const uint32_t* any_mask;
  const uint4* any_acorr;
  uint32_t any_groups, pad0;
  uint64_t mult[SF_MAX_RUNTIME_M];
};

template <int NH, int RNS = 0, int RM2 = 0>
__global__ __launch_bounds__(SF_THREADS) void seed_rtile_kernel(const SeedRtileArgs a)
{
  constexpr bool ROT = RNS > 0;
  static_assert(!ROT || (NH == 4 && RNS <= 4 && RM2 >= 1 && RM2 <= 4), "rotated-slot layout: k <= 32, <= 4 seeds");
  constexpr uint32_t NSETS = ROT ? (uint32_t)(RNS + 1) / 2u : 0u; // table sets of 64 KiB: seeds {0, 1}, {2, 3}
  // ANY (NH == 0, round 3): any seed set of any k in ONE pass from the k-independent fw tables, exactly as in
  // seed_wtile_kernel -- seeds of more than 64 bases and seed sets of several table passes on variable-length reads
  constexpr bool ANY = NH == 0;
  static_assert(!ANY || !ROT, "the any-seed form has its own tables");
  constexpr int NW = ANY ? 1 : (NH + 1) / 2;
  constexpr uint32_t NT = 2u * NH;
  extern __shared__ __attribute__((aligned(256))) uint32_t lds_dyn[];
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint4* tabs = (uint4*)lds_dyn;
  const uint32_t n_grp = ANY ? a.n_seeds * a.any_groups : 0u;
  const uint32_t n_entries = ANY ? SA_TAB_ENTRIES + n_grp + ((n_grp + 3u) >> 2) : ROT ? 4096u * NSETS : a.n_seeds * NT * 256u;
  const uint4* g_acorr = tabs + SA_TAB_ENTRIES;
  const uint32_t* g_mask = (const uint32_t*)(g_acorr + n_grp);
  const uint32_t per = ROT ? (uint32_t)(RNS * RM2) : a.n_seeds * a.m2;
  const uint32_t otile_u64 = a.otile_recs * per + 2u;
  // per wave: record tile | bit stream | read table (first window, last window + 1, first base, first record) | window map
  const uint32_t RT = a.R <= 16u ? 16u : a.R <= 32u ? 32u : 64u; // entries of the read table
  const uint32_t per_wave = otile_u64 * 2u + a.bits_dwords + 4u * RT + a.wmap_dwords;
  uint32_t* wbase = lds_dyn + n_entries * 4u + wave * per_wave;
  uint64_t* otile = (uint64_t*)wbase;
  uint32_t* bits = wbase + otile_u64 * 2u;
  // read table: one 16-byte entry per read {first window, last window + 1, first base, first record} -- one LDS read
  uint4* rt = (uint4*)(bits + a.bits_dwords);
  uint8_t* wmap = (uint8_t*)(rt + RT); // windows [16c, 16c + 16) of the tile: the read that holds window 16c
  if constexpr (ANY) {
    sa_load(tabs, a.tables, a.any_acorr, a.any_mask, n_grp, a.any_groups, tid, blockDim.x);
  } else if constexpr (ROT) {
    for (uint32_t i = tid; i < n_entries; i += blockDim.x) {
      const uint32_t set = i >> 12, ii = i & 4095u;
      const uint32_t in_set = (uint32_t)RNS - 2u * set < 2u ? (uint32_t)RNS - 2u * set : 2u;
      const uint32_t e = ii >> 4, v = ii & 15u, jt = v & 7u, sd = 2u * set + ((v >> 3) < in_set ? (v >> 3) : in_set - 1u);
      tabs[i] = jt < a.ntab ? a.tables[((size_t)sd * a.ntab + jt) * 256u + e] : make_uint4(0, 0, 0, 0);
    }
  } else {
    for (uint32_t i = tid; i < n_entries; i += blockDim.x) {
      const uint32_t tb = i >> 8, sd = tb / NT, jt = tb - sd * NT;
      tabs[i] = jt < a.ntab ? a.tables[((size_t)sd * a.ntab + jt) * 256u + (i & 255u)] : make_uint4(0, 0, 0, 0);
    }
  }
  __syncthreads();
  typedef __attribute__((address_space(3))) const nt_v4u lds_v4u;
  uint32_t rsel[8], roff[8];
  uint32_t rdelta = 0, rb3 = 0;
  if constexpr (ROT) {
    rb3 = (lane >> 3) & 1u;
    const uint32_t tb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)lds_dyn;
#pragma unroll
    for (uint32_t st = 0; st < 8; ++st) {
      const uint32_t cpos = (lane + st) & 7u;
      rsel[st] = 0x0c0c000cu | (cpos << 8);
      roff[st] = tb + (((rb3 << 3) + cpos) << 4);
      asm volatile("" : "+v"(rsel[st]), "+v"(roff[st]));
    }
    rdelta = rb3 ? (uint32_t)-128 : 128u;
  }
  auto lds_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  };
  auto wave_incl_scan32 = [&](uint32_t v) { // DPP prefix sum (see kmer_reads_kernel.hpp)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
  };
  auto bcast64 = [&](uint64_t v, uint32_t src) -> uint64_t {
    return ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(v >> 32), (int)src, 64) << 32) |
           (uint32_t)__shfl((int)(uint32_t)v, (int)src, 64);
  };
  const uint32_t k = a.k;
  const uint64_t per_block = (a.n_tiles + gridDim.x - 1) / gridDim.x;
  const uint64_t t_begin = (uint64_t)blockIdx.x * per_block;
  const uint64_t t_end = t_begin + per_block < a.n_tiles ? t_begin + per_block : a.n_tiles;

  for (uint64_t t = t_begin + wave; t < t_end; t += a.waves) {
    const uint64_t r0 = t * a.R;
    const uint32_t nr = a.n_reads - r0 < a.R ? (uint32_t)(a.n_reads - r0) : a.R;
    const bool has = lane < nr;
    const uint64_t rj = r0 + (has ? lane : 0u);
    const uint64_t s_j = a.starts[rj], e_j = a.ends[rj];
    const uint64_t len_j = has && e_j > s_j ? e_j - s_j : 0;
    const bool listed = a.flags[rj] != 0;
    const uint64_t ro_j = a.read_off[rj];
    const uint64_t slab0 = bcast64(s_j, 0), slab_end = bcast64(e_j, nr - 1u), ro_0 = bcast64(ro_j, 0);
    const uint32_t shift = (uint32_t)(((uintptr_t)a.seqs + slab0) & 15u);
    const uint8_t* vbase = a.seqs + slab0 - shift;
    const uint32_t n_vec = slab_end > slab0 ? (uint32_t)((shift + (slab_end - slab0) + 15u) >> 4) : 0u;
    for (uint32_t i = lane; i < n_vec; i += 64u) {
      const uint4 x = *(const uint4*)(vbase + ((uint64_t)i << 4));
      uint32_t bad = 0;
      bits[i] = pack16(x, bad);
    }
    if (lane < (uint32_t)NW + 3u) bits[n_vec + lane] = 0;
    const uint32_t nwin_j = (!listed && len_j >= k) ? (uint32_t)(len_j - k + 1u) : 0u;
    const uint32_t wend = wave_incl_scan32(nwin_j), wbeg = wend - nwin_j;
    if (lane < RT) {
      // (y = ~0 past the tile's reads: the walk below stops there at the latest)
      rt[lane] = make_uint4(wbeg, has ? wend : 0xFFFFFFFFu, shift + (uint32_t)(s_j - slab0), (uint32_t)(ro_j - ro_0));
    }
    const uint32_t W = (uint32_t)__shfl((int)wend, 63, 64);
    for (uint32_t c = (wbeg + 15u) >> 4; (c << 4) < wend; ++c) wmap[c] = (uint8_t)lane; // (nothing for a read without windows)
    lds_sync();

    for (uint32_t q0 = 0; q0 < W;) {
      const uint32_t q = q0 + lane;
      const bool live = q < W;
      const uint32_t qq = live ? q : q0;
      uint32_t j = wmap[qq >> 4];
      // the chunk's first read or, usually at most, the one after it: both entries in flight at once
      uint4 ent = rt[j];
      {
        const uint4 ent1 = rt[j + 1u < RT ? j + 1u : j];
        if (qq >= ent.y) { ent = ent1; ++j; }
      }
      while (qq >= ent.y) ent = rt[++j]; // (several short reads inside one chunk of 16 windows)
      const uint32_t p = qq - ent.x;
      const uint32_t b = ent.z + p;
      const uint32_t oslot = ent.w + p;
      const uint32_t gbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)oslot);
      const uint32_t slot = oslot - gbase;
      // (a listed read between two clean ones leaves a hole; windows past the tile's capacity wait for the next group)
      // A group ends where a 128-byte line of the stream ends (every `align_recs` records: 8 for 48-byte records), so that
      // the groups after the tile's first start on a line: their write-through stores then cover whole lines only
      const uint64_t g_first = ro_0 + gbase;
      const uint32_t cut = (uint32_t)(((g_first + 64u) / a.align_recs) * a.align_recs - g_first); // in (64 - align, 64]
      const bool fits = live && slot < a.otile_recs && slot < cut;
      const uint32_t nl = (uint32_t)__builtin_popcountll(__ballot(fits));
      const bool act = lane < nl;
      const uint32_t span = (uint32_t)__shfl((int)slot, (int)(nl - 1u), 64) + 1u;
      uint64_t* const dst = a.hashes + (ro_0 + gbase) * per;
      const uint32_t par = a.rec_stride ? 0u : (uint32_t)(((uintptr_t)dst >> 3) & 1u);
      if (act) {
        const uint32_t d = b >> 4, sh = (b & 15u) << 1;
        uint32_t w[NW];
        if constexpr (!ANY) {
          uint32_t lo = bits[d];
#pragma unroll
          for (int i = 0; i < NW; ++i) {
            const uint32_t hi = bits[d + i + 1];
            w[i] = funnel(hi, lo, sh);
            lo = hi;
          }
        } else {
          w[0] = 0;
        }
        uint64_t* mine = otile + par + slot * per;
        if constexpr (ANY) {
          const uint32_t G = a.any_groups, k31 = a.k % 31u, k33 = a.k % 33u;
          for (uint32_t s = 0; s < a.n_seeds; ++s) {
            const uint64_t h0 = sa_hash(tabs, g_acorr, g_mask, bits, d, sh, s, G, k31, k33);
            mine[s * a.m2] = h0;
#pragma unroll
            for (uint32_t jj = 1; jj < (uint32_t)SF_MAX_RUNTIME_M; ++jj)
              if (jj < a.m2) mine[s * a.m2 + jj] = mix_hash(h0, a.mult[jj]);
          }
        } else if constexpr (ROT) {
          uint32_t ad[8];
#pragma unroll
          for (uint32_t st = 0; st < 8; ++st) ad[st] = __builtin_amdgcn_perm(w[1], w[0], rsel[st]) + roff[st];
#pragma unroll
          for (int round = 0; round < RNS; ++round) { // (rounds and table sets as in seed_wtile_kernel)
            const int set = round >> 1, half = round & 1;
            const bool pair = RNS - 2 * set >= 2;
            const uint32_t sbase = (uint32_t)set * 65536u;
            nt_v4u e[8];
#pragma unroll
            for (uint32_t st = 0; st < 8; ++st) e[st] = *(lds_v4u*)(uintptr_t)((half == 0 ? ad[st] : ad[st] + rdelta) + sbase);
            uint32_t f0 = e[0].x ^ e[1].x, f1 = e[0].y ^ e[1].y, r0 = e[0].z ^ e[1].z, r1 = e[0].w ^ e[1].w;
#pragma unroll
            for (uint32_t st = 2; st < 8; st += 2) {
              f0 = __builtin_amdgcn_bitop3_b32(f0, e[st].x, e[st + 1].x, 0x96);
              f1 = __builtin_amdgcn_bitop3_b32(f1, e[st].y, e[st + 1].y, 0x96);
              r0 = __builtin_amdgcn_bitop3_b32(r0, e[st].z, e[st + 1].z, 0x96);
              r1 = __builtin_amdgcn_bitop3_b32(r1, e[st].w, e[st + 1].w, 0x96);
            }
            const uint64_t h0 = canon_pair(f0, f1, r0, r1);
            uint64_t* const rec = mine + (2u * (uint32_t)set + (pair ? ((uint32_t)half ^ rb3) : 0u)) * (uint32_t)RM2;
            rec[0] = h0;
#pragma unroll
            for (uint32_t jj = 1; jj < (uint32_t)RM2; ++jj) rec[jj] = mix_hash(h0, a.mult[jj]);
          }
        } else {
          for (uint32_t s = 0; s < a.n_seeds; ++s) {
            const uint4* ts = tabs + s * NT * 256u;
            uint4 e[NT];
#pragma unroll
            for (uint32_t jt = 0; jt < NT; ++jt) e[jt] = ts[jt * 256u + ((w[jt >> 2] >> ((jt & 3u) * 8u)) & 0xFFu)];
            uint32_t f0 = e[0].x ^ e[1].x, f1 = e[0].y ^ e[1].y, r0 = e[0].z ^ e[1].z, r1 = e[0].w ^ e[1].w;
#pragma unroll
            for (uint32_t jt = 2; jt < NT; jt += 2) {
              f0 = __builtin_amdgcn_bitop3_b32(f0, e[jt].x, e[jt + 1].x, 0x96);
              f1 = __builtin_amdgcn_bitop3_b32(f1, e[jt].y, e[jt + 1].y, 0x96);
              r0 = __builtin_amdgcn_bitop3_b32(r0, e[jt].z, e[jt + 1].z, 0x96);
              r1 = __builtin_amdgcn_bitop3_b32(r1, e[jt].w, e[jt + 1].w, 0x96);
            }
            const uint64_t h0 = canon_pair(f0, f1, r0, r1);
            mine[s * a.m2] = h0;
#pragma unroll
            for (uint32_t jj = 1; jj < (uint32_t)SF_MAX_RUNTIME_M; ++jj)
              if (jj < a.m2) mine[s * a.m2 + jj] = mix_hash(h0, a.mult[jj]);
          }
        }
        if (a.pos) a.pos[ro_0 + gbase + slot] = p; // SeedNtHash::get_pos() of a clean read's window
      }
      lds_sync();
      // ---- the group's records (holes included) as aligned 16-byte pieces ----
      const uint32_t n_vals = span * per;
      if (a.rec_stride) { // this pass's part of each record: 8-byte pieces
        uint64_t* const o = a.hashes + (ro_0 + gbase) * a.rec_stride + a.rec_off;
        for (uint32_t wi = lane; wi < span; wi += 64u)
          for (uint32_t jv = 0; jv < per; ++jv) o[wi * a.rec_stride + jv] = otile[wi * per + jv];
        lds_sync();
        q0 += nl;
        continue;
      }
      uint64_t* const base = dst - par;
      const uint32_t sp = par + n_vals;
      const uint32_t pf = par, pl = sp >> 1; // whole pieces [pf, pl) (par is 0 or 1: piece 0 is whole iff par == 0)
      for (uint32_t pi = lane; pi < pl; pi += 64u) {
        if (pi >= pf) {
          const nt_v4u sv = *(const nt_v4u*)(otile + 2u * pi);
          asm volatile("global_store_dwordx4 %0, %1, off" SW_STORE_POLICY "\n\ts_nop 1" ::"v"(base + 2u * pi), "v"(sv) : "memory");
        }
      }
      if (lane == 0u && par != 0u) base[1] = otile[1];                              // head
      if (lane == 1u && (sp & 1u) != 0u && sp > 2u * par) base[sp - 1u] = otile[sp - 1u]; // tail
      lds_sync();
      q0 += nl;
    }
    lds_sync(); // bit stream, read table and window map are free again
  }
}

// --------------------------------------------------------------------------
// General path: the reference's position state machine, one lane per read.
// --------------------------------------------------------------------------
struct SeedGeneralArgs {
  const uint8_t* seqs;
  const uint64_t* read_list;  // optional: only these reads (n_reads = length of the list)
  const uint64_t* offsets;    // n_reads+1 offsets, or (with ends) the first byte of every read
  const uint64_t* ends;       // optional: read r = [offsets[r], ends[r]) -- spans of one buffer
  uint64_t n_reads;
  uint32_t len, stride;
  uint32_t k, m2, n_seeds;
  uint32_t care_words;        // ceil(k/32)
  const uint32_t* care_bits;  // [seed][care_words]: bit p = position p contributes
  const uint32_t* blk_start;  // [seed] index into blk_pairs
  const uint32_t* blk_count;  // [seed]
  const uint32_t* blk_pairs;  // [start,end) pairs in get_blocks order (src/seed.cpp:19-66)
  const uint4* tables;        // optional: [seed][ntab][256] byte tables (windows made of bases only)
  uint32_t ntab, pad1;
  uint32_t wave_lmax, wave_waves; // seed_wave_kernel: longest read it stages, waves per block
  const uint64_t* read_off;
  uint64_t* counts;
  uint64_t* hashes;
  uint32_t* pos;
  uint64_t* fwd;              // optional: [k-mer][seed] forward-strand hashes
  uint64_t* rev;              // optional: [k-mer][seed] reverse-strand hashes
  uint64_t capacity;
  uint64_t mult[256];
};

// src/seed.cpp:146-158: walking seeds -> blocks -> positions, the first NUL byte
__device__ inline bool seed_first_nul(const SeedGeneralArgs& a, const uint8_t* win, uint32_t* where)
{
  for (uint32_t s = 0; s < a.n_seeds; ++s) {
    const uint32_t b0 = a.blk_start[s];
    for (uint32_t b = 0; b < a.blk_count[s]; ++b) {
      const uint32_t lo = a.blk_pairs[2 * (b0 + b)], hi = a.blk_pairs[2 * (b0 + b) + 1];
      for (uint32_t p = lo; p < hi; ++p)
        if (win[p] == 0) { *where = p; return true; }
    }
  }
  return false;
}

template <bool COUNT_ONLY>
__global__ __launch_bounds__(256) void seed_general_kernel(const SeedGeneralArgs* __restrict__ ap)
{
  const SeedGeneralArgs& a = *ap;
  const uint64_t item = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= a.n_reads) return;
  const uint64_t rid = a.read_list ? a.read_list[item] : item;
  uint64_t start, len;
  if (a.offsets) {
    start = a.offsets[rid];
    const uint64_t end = a.ends ? a.ends[rid] : a.offsets[rid + 1];
    len = end > start ? end - start : 0;
  } else {
    start = rid * a.stride;
    len = a.len;
  }
  const uint8_t* s = a.seqs + start;
  const uint32_t k = a.k;
  const uint32_t per = a.n_seeds * a.m2;
  uint64_t emitted = 0;
  if (len >= k) {
    bool has_nul = false;
    for (uint64_t i = 0; i < len; ++i) has_nul |= (s[i] == 0);
    const uint64_t base = COUNT_ONLY ? 0 : a.read_off[rid];
    uint64_t pos = 0;
    // SeedNtHash::init (src/seed.cpp:493-516)
    auto init = [&]() -> bool {
      uint32_t where = 0;
      while (pos < len - k + 1 && has_nul && seed_first_nul(a, s + pos, &where)) pos += where + 1;
      return !(pos > len - k);
    };
    bool ok = init();
    while (ok) {
      if (!COUNT_ONLY) {
        const uint64_t o = base + emitted;
        if (o < a.capacity) {
          const uint8_t* win = s + pos;
          // a window of bases only (almost all of them) is hashed from the byte tables, 4 bases per lookup;
          // one with a non-base needs the per-character seed values (SEED_TAB), position by position
          bool all_bases = a.tables != nullptr;
          for (uint32_t p = 0; p < k && all_bases; ++p) all_bases = is_base(win[p]);
          for (uint32_t sd = 0; sd < a.n_seeds; ++sd) {
            const uint32_t* care = a.care_bits + sd * a.care_words;
            uint64_t fh = 0, rh = 0;
            if (all_bases) {
              const uint4* ts = a.tables + (size_t)sd * a.ntab * 256u;
              for (uint32_t jt = 0; jt < a.ntab; ++jt) {
                uint32_t byte = 0;
                for (uint32_t q = 0; q < 4u && 4u * jt + q < k; ++q) byte |= ((win[4u * jt + q] >> 1) & 3u) << (2u * q);
                const uint4 e = ts[jt * 256u + byte];
                fh ^= ((uint64_t)e.y << 32) | e.x;
                rh ^= ((uint64_t)e.w << 32) | e.z;
              }
            } else {
              for (uint32_t p = 0; p < k; ++p) { // Horner: F = XOR srol^{k-1-p}(S[c_p])
                const bool c = (care[p >> 5] >> (p & 31u)) & 1u;
                fh = srol1(fh) ^ (c ? fwd_seed(win[p]) : 0);
              }
              for (uint32_t p = k; p-- > 0;) {   // R = XOR srol^{p}(S[c_p & 7])
                const bool c = (care[p >> 5] >> (p & 31u)) & 1u;
                rh = srol1(rh) ^ (c ? rc_seed(win[p]) : 0);
              }
            }
            const uint64_t h0 = fh + rh;
            if (a.fwd) a.fwd[o * a.n_seeds + sd] = fh;
            if (a.rev) a.rev[o * a.n_seeds + sd] = rh;
            uint64_t* dst = a.hashes + o * per + sd * a.m2;
            dst[0] = h0;
            for (uint32_t jj = 1; jj < a.m2; ++jj) dst[jj] = mix_hash(h0, a.mult[jj]);
          }
          if (a.pos) a.pos[o] = (uint32_t)pos;
        }
      }
      emitted++;
      // SeedNtHash::roll (src/seed.cpp:518-544)
      if (pos >= len - k) break;
      if (!is_base(s[pos + k])) {
        pos += k;
        ok = init();
      } else {
        pos++;
      }
    }
  }
  if (a.counts) a.counts[rid] = emitted;
}


// --------------------------------------------------------------------------
// seed_wave_kernel: one WAVE per read, one lane per window -- SeedNtHash on reads of any (bounded) length,
// given as offsets, spans or fixed length, with or without non-bases.  The exact semantics of the reference
// without its sequential walk:
//   * the emitted positions.  SeedNtHash::roll() only looks at the INCOMING character (src/seed.cpp:518-544):
//     a non-base at q makes it jump to the window that starts AT q, but only if position q-k was visited,
//     i.e. q >= (previous jump target) + k, the first one q >= k.  These "triggers" are found by one pass
//     over the read's non-bases (usually none); window p is emitted iff no trigger lies in (p, p+k).
//     (Reads containing NUL follow init()'s extra rule, src/seed.cpp:146-158,493-516: lane 0 walks those.)
//   * the values.  A window of bases only is hashed from the per-seed byte tables as in seed_fixed_kernel;
//     a window holding a non-base (emitted all the same) needs the per-character seed values of
//     SEED_TAB: Horner over its k raw bytes (rare, divergent).
// Emitted records are compacted per 64 windows through a wave-private LDS tile and written contiguously at
// the read's scanned offset.  COUNT_ONLY: just the emitted count per read.
// --------------------------------------------------------------------------
template <bool COUNT_ONLY, int NW>
__global__ __launch_bounds__(1024) void seed_wave_kernel(const SeedGeneralArgs* __restrict__ ap)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  const SeedGeneralArgs& a = *ap;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t k = a.k, per = a.n_seeds * a.m2, lmax = a.wave_lmax;
  // LDS: [tables] | per wave { raw bytes | codes | validity | triggers / emission | record tile }
  uint4* tabs = (uint4*)lds_dyn;
  const uint32_t n_entries = (COUNT_ONLY || NW == 0) ? 0u : a.n_seeds * a.ntab * 256u;
  const uint32_t raw_dw = (lmax + 64u) >> 2, code_dw = (lmax >> 4) + 8u, bit_dw = (lmax >> 5) + 8u;
  const uint32_t tile_dw = COUNT_ONLY ? 0u : 64u * per * 2u;
  const uint32_t per_wave = (raw_dw + code_dw + 2u * bit_dw + tile_dw + 3u) & ~3u;
  uint32_t* wb = lds_dyn + n_entries * 4u + wave * per_wave;
  uint8_t* raw = (uint8_t*)wb;
  uint32_t* bits = wb + raw_dw;
  uint32_t* vbits = bits + code_dw;
  uint32_t* tbits = vbits + bit_dw;
  uint64_t* otile = (uint64_t*)(wb + ((raw_dw + code_dw + 2u * bit_dw + 3u) & ~3u));
  if (!COUNT_ONLY) {
    for (uint32_t i = tid; i < n_entries; i += blockDim.x) tabs[i] = a.tables[i];
  }
  __syncthreads();
  auto lds_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  };
  const uint64_t per_block = (a.n_reads + gridDim.x - 1) / gridDim.x;
  const uint64_t i_begin = (uint64_t)blockIdx.x * per_block;
  const uint64_t i_end = i_begin + per_block < a.n_reads ? i_begin + per_block : a.n_reads;
  for (uint64_t item = i_begin + wave; item < i_end; item += a.wave_waves) {
    const uint64_t rid = a.read_list ? a.read_list[item] : item;
    uint64_t start, len64;
    if (a.offsets) {
      start = a.offsets[rid];
      const uint64_t end = a.ends ? a.ends[rid] : a.offsets[rid + 1];
      len64 = end > start ? end - start : 0;
    } else {
      start = rid * a.stride;
      len64 = a.len;
    }
    // reads longer than the staging buffers go through it in segments of seg_w windows; the state of the
    // reference's walk between two segments is (last init position, next position that may be emitted,
    // "that position still needs init()'s NUL check")
    const uint64_t len = len64;
    if (len < k) {
      if (COUNT_ONLY && lane == 0) a.counts[rid] = 0;
      continue;
    }
    const uint64_t nwin = len - k + 1u;
    const uint32_t seg_w = lmax - k;
    uint64_t last_init = 0, next_pos = 0;
    uint32_t need_init = 1;
    uint64_t emitted_before = 0;
    const uint64_t obase = COUNT_ONLY ? 0 : a.read_off[rid];
    for (uint64_t ws = 0; ws < nwin; ws += seg_w) {
      const uint64_t we = ws + seg_w < nwin ? ws + seg_w : nwin;
      // staged bases: [ws, we + k) -- the k bases of the last window and the character that comes in after it
      const uint32_t nb = (uint32_t)(len - ws < we - ws + k ? len - ws : we - ws + k);
      const uint32_t seg_win = (uint32_t)(we - ws);
      const uint8_t* s = a.seqs + start + ws;
      lds_sync();
      // ---- stage the segment: raw bytes, 2-bit codes, validity bits ------------------------------------
      uint32_t f_bad = 0, f_nul = 0;
      const uint32_t n_vec = (nb + 15u) >> 4;
      for (uint32_t i = lane; i < n_vec; i += 64u) {
        uint4 x;
        if (16u * i + 16u <= nb) {
          __builtin_memcpy(&x, s + 16u * i, 16);
        } else {
          uint32_t wv[4] = {0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u}; // 'A' padding past the segment
          for (uint32_t b = 0; 16u * i + b < nb; ++b) {
            wv[b >> 2] &= ~(0xFFu << ((b & 3u) * 8u));
            wv[b >> 2] |= (uint32_t)s[16u * i + b] << ((b & 3u) * 8u);
          }
          x = make_uint4(wv[0], wv[1], wv[2], wv[3]);
        }
        *(uint4*)(raw + 16u * i) = x;
        uint32_t i0, i1, i2, i3;
        const uint32_t c0 = pack4v(x.x, i0), c1 = pack4v(x.y, i1), c2 = pack4v(x.z, i2), c3 = pack4v(x.w, i3);
        bits[i] = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
        const uint32_t inv16 = i0 | (i1 << 4) | (i2 << 8) | (i3 << 12);
        ((uint16_t*)vbits)[i] = (uint16_t)inv16;
        f_bad |= inv16;
        auto has_zero = [](uint32_t w) { return ((w - 0x01010101u) & ~w & 0x80808080u) != 0u; };
        if (inv16 && (has_zero(x.x) || has_zero(x.y) || has_zero(x.z) || has_zero(x.w))) f_nul = 1;
      }
      if (lane < 12u) { bits[n_vec + (lane & 7u)] = 0; ((uint16_t*)vbits)[n_vec + lane] = 0; }
      for (uint32_t i = lane; i < ((nb + 31u) >> 5) + 6u; i += 64u) tbits[i] = 0;
      const bool any_bad = __ballot(f_bad != 0) != 0;
      const bool any_nul = __ballot(f_nul != 0) != 0;
      lds_sync();
      const uint64_t next_pos_in = next_pos;
      if (lane == 0) {
        if (any_nul) {
          // init()'s NUL rule makes the walk sequential: the reference's state machine over positions only,
          // emitted windows marked in tbits (relative to ws)
          uint64_t pos = next_pos > ws ? next_pos : ws;
          while (pos < we) {
            if (need_init) {
              uint32_t where = 0;
              while (pos < nwin && pos < we && seed_first_nul(a, raw + (pos - ws), &where)) pos += where + 1;
              if (pos >= we || pos >= nwin) break; // init() continues in the next segment / the read is over
              last_init = pos;
              need_init = 0;
            }
            tbits[(pos - ws) >> 5] |= 1u << ((pos - ws) & 31u);
            if (pos >= len - k) { pos = nwin; break; }
            if (!is_base(raw[pos + k - ws])) { pos += k; need_init = 1; }
            else pos++;
          }
          next_pos = pos;
        } else {
          if (need_init) { last_init = next_pos > ws ? next_pos : ws; need_init = 0; } // no NUL here: init() passes
          if (any_bad) {
            // triggers: a non-base q decides when position q-k is visited (q >= ws + k: not an earlier segment's),
            // and jumps iff q >= last init + k
            uint64_t last_trigger = 0;
            bool have = false;
            // from trigger to trigger: the next one is the first non-base at or after (last init + k), so a run of
            // thousands of non-bases costs one step per k of them, not one per character
            uint64_t from = last_init + k > ws + k ? last_init + k : ws + k;
            while (from < ws + nb) {
              uint32_t qr = (uint32_t)(from - ws);
              uint32_t w = qr >> 5;
              uint32_t word = vbits[w] & (~0u << (qr & 31u));
              while (!word && 32u * (w + 1u) < nb) word = vbits[++w];
              if (!word) break;
              qr = 32u * w + (uint32_t)__builtin_ctz(word);
              if (qr >= nb) break;
              const uint64_t q = ws + qr;
              last_init = q;
              last_trigger = q;
              have = true;
              tbits[qr >> 5] |= 1u << (qr & 31u);
              from = q + k;
            }
            if (have && last_trigger > next_pos) next_pos = last_trigger;
            if (have && last_trigger >= we) need_init = 1; // its window reaches into the next segment
          }
        }
      }
      last_init = ((uint64_t)__shfl((uint32_t)(last_init >> 32), 0, 64) << 32) | __shfl((uint32_t)last_init, 0, 64);
      next_pos = ((uint64_t)__shfl((uint32_t)(next_pos >> 32), 0, 64) << 32) | __shfl((uint32_t)next_pos, 0, 64);
      need_init = __shfl(need_init, 0, 64);
      lds_sync();
      // ---- the windows of the segment, 64 at a time ----------------------------------------------------
      for (uint32_t w0 = 0; w0 < seg_win; w0 += 64u) {
        const uint32_t p = w0 + lane;
        const bool in = p < seg_win;
        const uint32_t pc = in ? p : 0u;
        bool emit;
        if (any_nul) emit = in && ((tbits[pc >> 5] >> (pc & 31u)) & 1u);
        else emit = in && ws + pc >= next_pos_in &&
                    (!any_bad || ((k <= 65u ? windows_with_non_base(tbits, pc + 1u, k - 1u)
                                            : windows_with_non_base_long(tbits, pc + 1u, k - 1u, 1u)) & 1u) == 0u);
        const uint64_t eb = __ballot(emit);
        const uint32_t n_e = (uint32_t)__builtin_popcountll(eb);
        if (COUNT_ONLY) { emitted_before += n_e; continue; }
        if (n_e == 0) continue;
        const uint32_t slot = (uint32_t)__builtin_popcountll(eb & ((1ull << lane) - 1ull));
        if (emit) {
          // NW == 0 (k > 64): every window by Horner over its raw bytes, no tables
          const bool dirty_win = NW == 0 || (any_bad && (windows_with_non_base(vbits, pc, k) & 1u));
          // a window inside a run of 'N' (the emitted one of every k there): nothing to add up.  Only for the letter N /
          // n itself -- bytes 1, 3, 4, 5, 7 are non-bases that still carry a value on the reverse strand (SEED_TAB)
          bool blank_win = false;
          if (NW != 0 && dirty_win) {
            blank_win = true;
            for (uint32_t q = 0; q < k && blank_win; q += 4u) {
              uint32_t w4;
              __builtin_memcpy(&w4, raw + pc + q, 4); // (raw holds 64 bytes past the segment)
              w4 |= 0x20202020u;                        // lower case
              const uint32_t keep = k - q >= 4u ? 0xFFFFFFFFu : (1u << (8u * (k - q))) - 1u;
              blank_win = ((w4 ^ 0x6E6E6E6Eu) & keep) == 0u;
            }
          }
          uint64_t* mine = otile + slot * per;
          uint32_t wwords[NW ? NW : 1];
          if (!dirty_win) {
            const uint32_t d = pc >> 4, sh = (pc & 15u) << 1;
            uint32_t lo = bits[d];
#pragma unroll
            for (int i = 0; i < NW; ++i) {
              const uint32_t hi = bits[d + i + 1];
              wwords[i] = funnel(hi, lo, sh);
              lo = hi;
            }
          }
          for (uint32_t sdx = 0; sdx < a.n_seeds; ++sdx) {
            uint64_t fh = 0, rh = 0;
            if (blank_win) {
              // (every character a non-base other than the table's odd entries: all seed values are 0)
            } else if (!dirty_win) {
              const uint4* ts = tabs + sdx * a.ntab * 256u;
              uint32_t f0 = 0, f1 = 0, r0 = 0, r1 = 0;
#pragma unroll
              for (int jt = 0; jt < 4 * NW; ++jt) { // a.ntab == 4 * NW (zero tables past ceil(k/4))
                const uint32_t byte = (wwords[jt >> 2] >> ((jt & 3) * 8)) & 0xFFu;
                const uint4 e = ts[(uint32_t)jt * 256u + byte];
                f0 ^= e.x; f1 ^= e.y; r0 ^= e.z; r1 ^= e.w;
              }
              fh = ((uint64_t)f1 << 32) | f0;
              rh = ((uint64_t)r1 << 32) | r0;
            } else {
              // (the care word in a register: one global load per 32 positions, not one per position -- a segment
              //  inside a long run of N took a millisecond with the latter)
              const uint32_t* care = a.care_bits + sdx * a.care_words;
              const uint8_t* win = raw + pc;
              uint32_t cw = 0;
              for (uint32_t q = 0; q < k; ++q) {
                if ((q & 31u) == 0u) cw = care[q >> 5];
                const bool c = (cw >> (q & 31u)) & 1u;
                fh = srol1(fh) ^ (c ? fwd_seed(win[q]) : 0);
              }
              cw = care[(k - 1u) >> 5];
              for (uint32_t q = k; q-- > 0;) {
                if ((q & 31u) == 31u) cw = care[q >> 5];
                const bool c = (cw >> (q & 31u)) & 1u;
                rh = srol1(rh) ^ (c ? rc_seed(win[q]) : 0);
              }
            }
            const uint64_t h0 = fh + rh;
            mine[sdx * a.m2] = h0;
            for (uint32_t jj = 1; jj < a.m2; ++jj) mine[sdx * a.m2 + jj] = mix_hash(h0, a.mult[jj]);
            // SeedNtHash::get_forward_hash / get_reverse_hash: one value per seed and k-mer
            if (a.fwd) a.fwd[(obase + emitted_before + slot) * a.n_seeds + sdx] = fh;
            if (a.rev) a.rev[(obase + emitted_before + slot) * a.n_seeds + sdx] = rh;
          }
          if (a.pos) a.pos[obase + emitted_before + slot] = (uint32_t)(ws + pc);
        }
        lds_sync();
        const uint32_t n_vals = n_e * per;
        uint64_t* dst = a.hashes + (obase + emitted_before) * per;
        if ((((uint64_t)(uintptr_t)dst) & 15u) == 0u) {
          for (uint32_t pi = lane; 2u * pi < n_vals; pi += 64u) {
            const uint4 dv = *(const uint4*)(otile + 2u * pi);
            if (2u * pi + 1u < n_vals) *(uint4*)(dst + 2u * pi) = dv;
            else *(uint2*)(dst + 2u * pi) = make_uint2(dv.x, dv.y);
          }
        } else {
          for (uint32_t v = lane; v < n_vals; v += 64u) dst[v] = otile[v];
        }
        emitted_before += n_e;
        lds_sync();
      }
    }
    if (COUNT_ONLY && lane == 0) a.counts[rid] = emitted_before;
  }
}

} // namespace ntamd
