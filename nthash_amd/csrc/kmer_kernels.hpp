// kmer_kernels.hpp -- CDNA4 (gfx950) kernels for contiguous k-mer hashing.
//
// Replaces, for a whole batch of reads, the reference's per-read loop
//     NtHash h(seq, len, m, k); while (h.roll()) use(h.hashes());
// (src/kmer.cpp:200-264): next_forward_hash / next_reverse_hash
// (src/kmer.cpp:84-94, 164-174), canonical() and extend_hashes()
// (src/internal.hpp:24-29, 104-118).
//
// Two kernels:
//   kmer_fixed_kernel   the hot path.  Fixed-length reads (optionally
//                       overlapping runs of one long sequence via `stride`),
//                       one lane rolls one run, state in VGPRs; ASCII is
//                       staged through LDS as a 2-bit stream, the hash stream
//                       leaves through an XOR-swizzled LDS tile so that every
//                       global store is a 16-byte piece of a 128-byte row.
//   kmer_general_kernel the N-aware path for ragged / dirty batches: one lane
//                       per read, exact reference emission order, compact
//                       output through per-read offsets.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "nt_math.hpp"

namespace ntamd {

constexpr int KF_THREADS = 256;              // 4 wavefronts
constexpr int KF_RUNS_PER_BLOCK = 256;       // one run (read) per lane
constexpr int KF_ROW_U64 = 16;               // tile row = 16 hashes = 128 B
constexpr int KF_TILE_U64 = 64 * KF_ROW_U64; // per-wave tile (8 KiB)
constexpr int KF_MAX_RUNTIME_M = 8;

struct KmerFixedArgs {
  const uint8_t* seqs;   // device, ASCII
  uint64_t* hashes;      // device, dense [run][window][m]
  uint32_t* dirty;       // device flag: set when a non-ACGTU byte is seen
  uint64_t n_runs;       // number of reads / runs
  uint32_t len;          // bases per run
  uint32_t stride;       // bases between run starts (== len for plain reads)
  uint32_t k, m;
  uint32_t nwin;         // len - k + 1
  uint32_t pad_dwords;   // front pad of the LDS bit stream, >= ceil(k/16)+1
  uint32_t n_tiles;
  uint32_t reserved;
  uint64_t f_init, r_init; // strand hashes of k virtual 'A's
  uint64_t tab[16][2];     // [(in<<2)|out] -> {fwd term, rev term}
  uint64_t mult[KF_MAX_RUNTIME_M]; // i ^ k*MULTISEED, i = 0..7 (runtime-m variant)
};

// ---- 2-bit packing of 16 ASCII bytes + validity accumulation -------------
// code = (c >> 1) & 3.  A byte is a base iff, lower-cased and with 'u' folded
// onto 't', it equals the canonical letter of its own code.
__device__ __forceinline__ uint32_t pack4(uint32_t w, uint32_t& bad)
{
  const uint32_t t = (w >> 1) & 0x03030303u;
  uint32_t x = w | 0x20202020u;
  const uint32_t ubit = (x >> 4) & 0x01010101u; // set for 0x7_ letters (t,u)
  x = x & ~ubit;                                // 'u'(0x75) -> 't'(0x74)
  const uint32_t canon = __builtin_amdgcn_perm(0u, 0x67746361u, t); // a,c,t,g by code
  bad |= x ^ canon;
  // gather the four 2-bit fields (one per byte of t) into one byte:
  // f0 + 4*f1 + 16*f2 + 64*f3 is a single 4-way byte dot product
  return __builtin_amdgcn_udot4(t, 0x40100401u, 0u, false);
}

__device__ __forceinline__ uint32_t pack16(uint4 v, uint32_t& bad)
{
  return pack4(v.x, bad) | (pack4(v.y, bad) << 8) | (pack4(v.z, bad) << 16) |
         (pack4(v.w, bad) << 24);
}

__device__ __forceinline__ uint32_t funnel(uint32_t hi, uint32_t lo, uint32_t sh)
{
  return __builtin_amdgcn_alignbit(hi, lo, sh); // ({hi,lo} >> (sh & 31))[31:0]
}

// Split rotates on a (lo, hi) register pair, 5 VALU ops each (the generic
// 64-bit C++ in nt_math.hpp compiles to ~10 ops with 64-bit shifts).
//   srol: bits 63..33 rotate left as a 31-bit word, bits 32..0 as a 33-bit word
__device__ __forceinline__ void srol_pair(uint32_t& lo, uint32_t& hi)
{
  const uint32_t t = __builtin_amdgcn_alignbit(hi, lo, 31); // (hi << 1) | (lo >> 31)
  const uint32_t nlo = (lo << 1) | (hi & 1u);               // bit 32 -> bit 0
  hi = (t & ~2u) | ((hi >> 30) & 2u);                       // bit 63 -> bit 33
  lo = nlo;
}
// canonical hash f + r (mod 2^64) from the 32-bit halves as ONE carry chain: the 64-bit C++ form
// (((u64)f_hi << 32) | f_lo) + ... compiles to two v_lshl_add_u64 plus moves and ors on gfx950
#ifndef NT_CANON_ASM
#define NT_CANON_ASM 1
#endif
__device__ __forceinline__ uint64_t canon_pair(uint32_t f_lo, uint32_t f_hi, uint32_t r_lo, uint32_t r_hi)
{
#if NT_CANON_ASM
  uint32_t lo, hi;
  asm("v_add_co_u32 %0, vcc, %2, %3\n\tv_addc_co_u32 %1, vcc, %4, %5, vcc"
      : "=&v"(lo), "=v"(hi)
      : "v"(f_lo), "v"(r_lo), "v"(f_hi), "v"(r_hi)
      : "vcc");
  return ((uint64_t)hi << 32) | lo;
#else
  return (((uint64_t)f_hi << 32) | f_lo) + (((uint64_t)r_hi << 32) | r_lo);
#endif
}
// 16-byte store of hash-stream data with the write-through policy (sc0 sc1), for stores that cover whole,
// aligned cache lines (the headline kernel's copy-out: +1.2 % in-process A/B; nt: no change).  NOT for the
// copy-outs whose pieces start anywhere: without L2 write merging the k=64/m=3 shape lost 20 %.
#ifndef NT_STREAM_STORE_POLICY
#define NT_STREAM_STORE_POLICY " sc0 sc1"
#endif
typedef uint32_t nt_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void stream_store16(void* p, const uint4 v)
{
  const nt_v4u sv = {v.x, v.y, v.z, v.w};
  // s_nop: a store of more than 8 bytes still reads its data registers for two cycles after issue; the
  // compiler pads that hazard for its own stores but cannot see one inside inline asm
  asm volatile("global_store_dwordx4 %0, %1, off" NT_STREAM_STORE_POLICY "\n\ts_nop 1" ::"v"(p), "v"(sv) : "memory");
}
//   sror: the inverse
__device__ __forceinline__ void sror_pair(uint32_t& lo, uint32_t& hi)
{
  const uint32_t nlo = __builtin_amdgcn_alignbit(hi, lo, 1);     // bits 32..1 -> 31..0
  const uint32_t t = __builtin_amdgcn_alignbit(hi >> 1, hi, 1);  // bits 63..34 -> 62..33, bit 33 -> bit 63 (and bit 32, replaced:)
  hi = (t & ~1u) | (lo & 1u);                                    // bit 0 -> bit 32
  lo = nlo;
}

// One step of the roll (next_forward_hash / next_reverse_hash, src/kmer.cpp:84-94, 164-174) on the register pairs, with the
// pair-table term {f.lo, f.hi, r.lo, r.hi} of the bases that enter and leave:
//     f = srol(f) ^ term.f          r = sror(r ^ term.r)
// Round 5: written out.  From the C++ above hipcc made 19-20 instructions of a step (it spreads the XORs over v_bitop3 and
// then computes both the folded and the unfolded form of the rotated halves); the split rotates are 5 and 4 instructions
//     srol: lo' = (hi & 1) | lo << 1;  hi' = bfi(2, hi >> 30, alignbit(hi, lo, 31))          (bit 63 -> 33, bit 32 -> 0)
//     sror: lo' = alignbit(hi, lo, 1); hi' = bfi(1, lo, alignbit(hi >> 1, hi, 1))            (bit 33 -> 63, bit 0 -> 32)
// and a step is 7 + 6 (+ 2 for the canonical sum): the kernels that are bound by their instruction stream (reads of any
// lengths, minimizers, MinHash, the fused Bloom level, k > 32) get that back.  Two statements: the strands are independent.
#ifndef NT_ROLL_ASM
#define NT_ROLL_ASM 1
#endif
// (ASM = false: the instantiations that sit at 128 registers with the compiler's form and would spill with this one)
template <bool ASM = true>
__device__ __forceinline__ void roll_step(uint32_t& f_lo, uint32_t& f_hi, uint32_t& r_lo, uint32_t& r_hi, const uint4 term)
{
#if NT_ROLL_ASM
  if constexpr (!ASM) {
    srol_pair(f_lo, f_hi);
    f_lo ^= term.x;
    f_hi ^= term.y;
    r_lo ^= term.z;
    r_hi ^= term.w;
    sror_pair(r_lo, r_hi);
    return;
  }
  // (in place where the old value is not needed again: the kernels around this are at their register limit)
  uint32_t t0, t1, nr_lo;
  asm("v_lshrrev_b32 %2, 30, %1\n\t"
      "v_alignbit_b32 %3, %1, %0, 31\n\t"
      "v_lshlrev_b32 %0, 1, %0\n\t"
      "v_and_or_b32 %0, %1, 1, %0\n\t"
      "v_bfi_b32 %1, 2, %2, %3\n\t"
      "v_xor_b32 %0, %0, %4\n\t"
      "v_xor_b32 %1, %1, %5"
      : "+v"(f_lo), "+v"(f_hi), "=&v"(t0), "=&v"(t1)
      : "v"(term.x), "v"(term.y));
  asm("v_xor_b32 %3, %4, %5\n\t"
      "v_xor_b32 %1, %1, %6\n\t"
      "v_lshrrev_b32 %2, 1, %1\n\t"
      "v_alignbit_b32 %0, %1, %3, 1\n\t"
      "v_alignbit_b32 %2, %2, %1, 1\n\t"
      "v_bfi_b32 %1, 1, %3, %2"
      : "=&v"(nr_lo), "+v"(r_hi), "=&v"(t0), "=&v"(t1)
      : "v"(r_lo), "v"(term.z), "v"(term.w));
  r_lo = nr_lo;
#else
  srol_pair(f_lo, f_hi);
  f_lo ^= term.x;
  f_hi ^= term.y;
  r_lo ^= term.z;
  r_hi ^= term.w;
  sror_pair(r_lo, r_hi);
#endif
}

// Tiles -> waves for the kernels whose waves own whole tiles: the blocks of the grid are split into `groups` groups of
// consecutive blocks, every group streams through its own contiguous range of tiles with all its waves interleaved in
// it (groups = 0 or >= blocks: one range per block).  Fewer, wider streams touch fewer pages at any moment.
struct TileRange {
  uint64_t first, end, step;
};
__device__ __forceinline__ TileRange tile_range(uint64_t n_tiles, uint32_t waves, uint32_t wave, uint32_t groups)
{
  const uint32_t nb = gridDim.x;
  const uint32_t G = groups != 0u && groups < nb ? groups : nb;
  const uint32_t bpg = nb / G; // blocks per group (the last group takes the remainder)
  uint32_t g = blockIdx.x / bpg;
  if (g >= G) g = G - 1u;
  const uint32_t b_in_g = blockIdx.x - g * bpg;
  const uint32_t g_blocks = g == G - 1u ? nb - g * bpg : bpg;
  const uint64_t per = (n_tiles + G - 1u) / G;
  const uint64_t t0 = (uint64_t)g * per;
  TileRange r;
  r.first = t0 + (uint64_t)b_in_g * waves + wave;
  r.end = t0 + per < n_tiles ? t0 + per : n_tiles;
  r.step = (uint64_t)g_blocks * waves;
  return r;
}

// ---- hidden loads and the ISA lint (nthash_amd/isa_lint.py, run by build.py on every unit) ----------------------------
// Several kernels issue the NEXT tile's global loads in inline asm and consume them after this tile's stores behind a
// counted s_waitcnt (vmcnt retires in order; an ordinary load there would make every wave wait for its own store
// acknowledgements).  hipcc believes the destination registers are ready at the asm statement: any copy, spill or use it
// inserts before the wait reads stale data (it happened once, round 2).  So every such site ends in a marker
//     asm volatile("; NTLINT_CONSUME %0 %1 ..." : "+v"(regs) ...)
// right behind its wait.  The marker is only a comment in the emitted ISA, but it names the registers hipcc holds the
// values in AT THE WAIT; the lint disassembles every kernel and refuses the build unless, on every path from a hidden
// load to the marker that consumes it, (1) no instruction reads or writes the load's destination registers, (2) the
// marker names exactly those registers (no copy in between) and (3) the marker sits behind an inline s_waitcnt vmcnt.
// -DNT_LINT_SELFTEST=1 breaks rule (1) on purpose in every kernel that has such a site (tests/test_isa_lint.py).
#ifndef NT_LINT_SELFTEST
#define NT_LINT_SELFTEST 0
#endif
#if NT_LINT_SELFTEST
#define NT_LINT_SELFTEST_TOUCH(reg32) \
  do { uint32_t lint_tmp_; asm volatile("v_mov_b32 %0, %1" : "=v"(lint_tmp_) : "v"(reg32)); asm volatile("" ::"v"(lint_tmp_)); } while (0)
#else
#define NT_LINT_SELFTEST_TOUCH(reg32) do { } while (0)
#endif

// word modes of the per-lane rolling loop
enum : int { W_NOEMIT = 0, W_EMIT = 1, W_BOUNDARY = 2, W_CHECKED = 3 };

// K_T / M_T: compile-time k and m (0 = read from args)
template <int K_T, int M_T>
__global__ __launch_bounds__(KF_THREADS) void kmer_fixed_kernel(const KmerFixedArgs a)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  __shared__ __attribute__((aligned(16))) uint64_t tile_all[4 * KF_TILE_U64];
  __shared__ __attribute__((aligned(16))) uint4 tab[16];
  uint32_t* bits = lds_dyn;

  const uint32_t k = K_T ? (uint32_t)K_T : a.k;
  const uint32_t m = M_T ? (uint32_t)M_T : a.m;
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = tid >> 6;
  uint64_t* tile = tile_all + wave * KF_TILE_U64;

  if (tid < 16) {
    tab[tid] = make_uint4((uint32_t)a.tab[tid][0], (uint32_t)(a.tab[tid][0] >> 32),
                          (uint32_t)a.tab[tid][1], (uint32_t)(a.tab[tid][1] >> 32));
  }
  // front pad of the bit stream = virtual 'A's (code 0)
  for (uint32_t i = tid; i < a.pad_dwords; i += KF_THREADS) bits[i] = 0;

  // swizzled slot of this lane's row: conflict-free 8-byte writes, 16-byte reads
  const uint32_t sw = ((lane & 7u) << 1) | ((lane >> 3) & 1u);
  const uint32_t wr_base = lane * KF_ROW_U64 + sw;
  const uint32_t vpr = a.nwin * m; // values per run in the output stream
  const uint32_t kmod = (k - 1u) & 15u;
  const uint32_t jb = (k - 1u) >> 4; // word holding the first emitting step
  const uint32_t n_full = a.len >> 4; // words with all 16 steps
  const uint32_t n_words = (a.len + 15u) >> 4;
  uint32_t bad = 0;

  for (uint32_t t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
    const uint64_t run0 = (uint64_t)t * KF_RUNS_PER_BLOCK;
    const uint64_t left = a.n_runs - run0;
    const uint32_t runs_here = left < KF_RUNS_PER_BLOCK ? (uint32_t)left : KF_RUNS_PER_BLOCK;
    // ---- phase 1: slab of ASCII -> 2-bit stream in LDS ---------------------
    // 16-byte vectors aligned in memory; a vector that holds one valid byte
    // lies inside that byte's page, so edge vectors are safe to load whole.
    const uint64_t byte0 = run0 * a.stride;
    const uint64_t addr0 = (uint64_t)(a.seqs + byte0);
    const uint32_t shift = (uint32_t)(addr0 & 15u); // foreign bytes in the first vector
    const uint4* vsrc = (const uint4*)(addr0 - shift);
    const uint32_t slab_bytes = (runs_here - 1u) * a.stride + a.len;
    const uint32_t n_vec = (shift + slab_bytes + 15u) >> 4;
    __syncthreads(); // previous tile consumed; pad/tab visible on the first pass
    for (uint32_t i = tid; i < n_vec; i += KF_THREADS) {
      const uint4 v = vsrc[i];
      uint32_t b = 0;
      const uint32_t p = pack16(v, b);
      const int32_t lo_cut = (int32_t)shift - (int32_t)(i << 4);
      const int32_t hi_cut = (int32_t)(shift + slab_bytes) - (int32_t)(i << 4);
      if (lo_cut > 0 || hi_cut < 16) {
        // edge vector: bytes outside the slab are somebody else's, do not judge them
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
      bits[a.pad_dwords + i] = p;
    }
    if (tid < 2) bits[a.pad_dwords + n_vec + tid] = 0; // funnel reads one word ahead
    __syncthreads();

    // ---- phase 2: every lane rolls its run --------------------------------
    const uint32_t lrun = wave * 64u + lane; // run index inside the tile
    const uint32_t bl = a.pad_dwords * 16u + shift + lrun * a.stride; // first base (stream index)
    const uint32_t in_d = bl >> 4, in_sh = (bl & 15u) << 1;
    const uint32_t ob = bl - k; // never negative thanks to the pad
    const uint32_t out_d = ob >> 4, out_sh = (ob & 15u) << 1;
    const uint64_t wave_run0 = run0 + wave * 64u;

    uint32_t f_lo = (uint32_t)a.f_init, f_hi = (uint32_t)(a.f_init >> 32);
    uint32_t r_lo = (uint32_t)a.r_init, r_hi = (uint32_t)(a.r_init >> 32);
    uint32_t in_lo = bits[in_d], out_lo = bits[out_d];
    const bool wave_full = wave_run0 + 64u <= a.n_runs;
    // this lane's slice of the wave's output rows: row R = 8*s + (lane >> 3), 16-byte chunk lane & 7
    uint64_t* const out_lane = a.hashes + (wave_run0 + (lane >> 3)) * vpr + 2u * (lane & 7u);
    const uint32_t rd_R = lane >> 3, rd_ch = lane & 7u;

    // write one 128-byte row per lane to global memory: 8 x (64 lanes x 16 B).
    // All eight LDS reads are issued before the first store so that their
    // latencies overlap; full waves take the unpredicated path.
    auto flush_row = [&](uint32_t row_v0, uint32_t nvalid) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      uint4 d[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const uint32_t R = (uint32_t)s * 8u + rd_R;
        const uint32_t chp = rd_ch ^ (R & 7u);
        const uint4 q = *(const uint4*)(tile + R * KF_ROW_U64 + 2u * chp);
        d[s] = (s & 1) ? make_uint4(q.z, q.w, q.x, q.y) : q;
      }
      uint64_t* const dst0 = out_lane + row_v0;
      if (wave_full && nvalid == 16u) {
#pragma unroll
        for (int s = 0; s < 8; ++s) *(uint4*)(dst0 + (uint64_t)s * 8u * vpr) = d[s];
      } else {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const uint64_t run = wave_run0 + (uint32_t)s * 8u + rd_R;
          uint64_t* dst = dst0 + (uint64_t)s * 8u * vpr;
          if (run < a.n_runs) {
            if (2u * rd_ch + 1u < nvalid) {
              *(uint4*)dst = d[s];
            } else if (2u * rd_ch < nvalid) {
              *(uint2*)dst = make_uint2(d[s].x, d[s].y);
            }
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    };

    auto emit = [&](uint32_t step, uint32_t i) {
      const uint64_t h0 = canon_pair(f_lo, f_hi, r_lo, r_hi);
      const uint32_t p = step - (k - 1u);           // window index inside the run
      const uint32_t pi = (i + 16u - kmod) & 15u;   // == p & 15
      uint64_t tb = 0;
      if (M_T > 1) tb = h0 * multiplier(k, 0);
#pragma unroll
      for (uint32_t jj = 0; jj < (M_T ? (uint32_t)M_T : m); ++jj) {
        uint64_t val;
        if (jj == 0) {
          val = h0;
        } else if (M_T) {
          // i ^ B differs from B = k*MULTISEED only in its low bits, so
          // h0*(i^B) = h0*B + h0*delta with a tiny compile-time delta
          const int64_t delta = (int64_t)(multiplier(k, jj) - multiplier(k, 0));
          const uint64_t tv = tb + (uint64_t)delta * h0;
          val = tv ^ (tv >> MULTISHIFT);
        } else {
          val = mix_hash(h0, a.mult[jj & (KF_MAX_RUNTIME_M - 1)]);
        }
        const uint32_t slot = (pi * m + jj) & 15u;
        tile[wr_base ^ slot] = val;
        if (slot == 15u) flush_row((p * m + jj) & ~15u, 16u);
      }
    };

    auto word = [&](auto mode_tag, uint32_t j) {
      constexpr int MODE = decltype(mode_tag)::value;
      const uint32_t in_hi = bits[in_d + j + 1];
      const uint32_t out_hi = bits[out_d + j + 1];
      const uint32_t w_in = funnel(in_hi, in_lo, in_sh);
      uint32_t w_out = funnel(out_hi, out_lo, out_sh);
      in_lo = in_hi;
      out_lo = out_hi;
      const uint32_t s0 = j << 4;
      // steps whose outgoing base lies before the run start see a virtual 'A'
      if (s0 + 16u <= k) w_out = 0;
      else if (s0 < k) w_out &= ~0u << ((k - s0) << 1);
      // nibble streams: u = even steps, v = odd steps; nibble = (in<<2)|out
      const uint32_t u = ((w_in & 0x33333333u) << 2) | (w_out & 0x33333333u);
      const uint32_t v = (w_in & 0xCCCCCCCCu) | ((w_out >> 2) & 0x33333333u);
      auto lookup = [&](uint32_t i) -> uint4 {
        const uint32_t src = (i & 1u) ? v : u;
        const uint32_t off = ((src >> ((i >> 1) * 4u)) & 0xFu) << 4;
        return *(const uint4*)((const char*)tab + off);
      };
      auto roll = [&](const uint4 term) {
        roll_step(f_lo, f_hi, r_lo, r_hi, term);
      };
      if constexpr (MODE == W_CHECKED) {
        const uint32_t steps = (a.len - s0) < 16u ? (a.len - s0) : 16u;
#pragma unroll 1
        for (uint32_t i = 0; i < steps; ++i) {
          roll(lookup(i));
          if (s0 + i >= k - 1u) emit(s0 + i, i);
        }
      } else {
        // the 16 table terms do not depend on the hash state: fetch them all
        // first so their LDS latencies overlap, then run the dependent chain
        uint4 terms[16];
#pragma unroll
        for (uint32_t i = 0; i < 16; ++i) terms[i] = lookup(i);
#pragma unroll
        for (uint32_t i = 0; i < 16; ++i) {
          roll(terms[i]);
          if constexpr (MODE == W_EMIT) emit(s0 + i, i);
          if constexpr (MODE == W_BOUNDARY) {
            if (i >= kmod) emit(s0 + i, i);
          }
        }
      }
    };

    uint32_t j = 0;
    const uint32_t pre_end = jb < n_full ? jb : n_full;
    for (; j < pre_end; ++j) word(std::integral_constant<int, W_NOEMIT>{}, j);
    if (j < n_full) { word(std::integral_constant<int, W_BOUNDARY>{}, j); ++j; }
#pragma unroll 1
    for (; j < n_full; ++j) word(std::integral_constant<int, W_EMIT>{}, j);
    for (; j < n_words; ++j) word(std::integral_constant<int, W_CHECKED>{}, j);
    if (vpr & 15u) flush_row(vpr & ~15u, vpr & 15u); // last, partial row
  }
  // any non-base byte in this block's slabs -> the caller re-runs the general path
  if (__ballot(bad != 0) != 0 && lane == 0) atomicOr(a.dirty, 1u);
}

// --------------------------------------------------------------------------
// General path: exact NtHash emission order on arbitrary bytes and lengths.
// --------------------------------------------------------------------------
constexpr uint32_t KG_SEG_WINDOWS = 1024; // long reads are cut into segments of this many windows

struct KmerGeneralArgs {
  const uint8_t* seqs;
  const uint64_t* offsets; // n_reads+1, or nullptr with fixed len/stride
  uint64_t n_reads;
  uint32_t len, stride;    // used when offsets == nullptr
  uint32_t k, m;
  // segmented mode (seg_base != nullptr): work item = one segment of a read;
  // seg_base[r] = index of read r's first segment (exclusive scan), n_items segments
  const uint64_t* seg_base;
  uint64_t n_items;
  const uint64_t* item_off; // exclusive scan of the per-item counts (hash pass only)
  uint64_t* counts;         // per-item emitted windows (count pass)
  uint64_t* hashes;
  uint32_t* pos;
  uint64_t* fwd;
  uint64_t* rev;
  uint64_t capacity;        // k-mers
  uint64_t sk_fwd[4];       // srol^k(seed[code])
  uint64_t sk_rc[4];        // srol^k(seed[code^2])
  uint64_t mult[256];
};

// segments per read: ceil(windows / KG_SEG_WINDOWS)
static __global__ __launch_bounds__(256) void kmer_seg_count_kernel(const uint64_t* __restrict__ offsets, uint64_t n_reads,
                                                            uint32_t len, uint32_t k, uint64_t* __restrict__ segs)
{
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_reads) return;
  const uint64_t l = offsets ? offsets[r + 1] - offsets[r] : len;
  segs[r] = l >= k ? (l - k + 1 + KG_SEG_WINDOWS - 1) / KG_SEG_WINDOWS : 0;
}

// per-read counts from per-item offsets: counts[r] = item_off[seg_base[r+1]] - item_off[seg_base[r]]
static __global__ __launch_bounds__(256) void kmer_seg_read_counts_kernel(const uint64_t* __restrict__ seg_base,
                                                                  const uint64_t* __restrict__ item_off,
                                                                  uint64_t n_reads, uint64_t n_items, uint64_t total,
                                                                  uint64_t* __restrict__ counts)
{
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_reads) return;
  const uint64_t a = seg_base[r], b = r + 1 < n_reads ? seg_base[r + 1] : n_items;
  const uint64_t oa = a < n_items ? item_off[a] : total, ob = b < n_items ? item_off[b] : total;
  counts[r] = ob - oa;
}

// One lane per read (or per segment of a read).  A window is emitted iff its k
// bytes are all bases -- the net effect of the reference's init()/roll() skipping
// (src/kmer.cpp:228-264).  Non-base bytes contribute nothing on entry and on
// exit, so the rolled state is exact again as soon as a clean window is reached.
// A segment starts its roll k-1 bytes before its first window's last byte, i.e.
// from scratch at its first window: no state crosses segments.
template <bool COUNT_ONLY>
__global__ __launch_bounds__(256) void kmer_general_kernel(const KmerGeneralArgs* __restrict__ ap)
{
  const KmerGeneralArgs& a = *ap;
  const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t k = a.k;
  uint64_t rid = id, w0 = 0, w1 = ~0ull; // read, first window, one past the last window of this item
  if (a.seg_base) {
    if (id >= a.n_items) return;
    // the read whose segment range contains id: last r with seg_base[r] <= id
    uint64_t lo = 0, hi = a.n_reads;
    while (hi - lo > 1) {
      const uint64_t mid = lo + (hi - lo) / 2;
      if (a.seg_base[mid] <= id) lo = mid; else hi = mid;
    }
    rid = lo;
    w0 = (id - a.seg_base[rid]) * KG_SEG_WINDOWS;
    w1 = w0 + KG_SEG_WINDOWS;
  } else if (id >= a.n_reads) {
    return;
  }
  uint64_t start, len;
  if (a.offsets) {
    start = a.offsets[rid];
    len = a.offsets[rid + 1] - start;
  } else {
    start = rid * a.stride;
    len = a.len;
  }
  const uint8_t* s = a.seqs + start;
  uint64_t emitted = 0;
  if (len >= k) {
    const uint64_t nwin = len - k + 1;
    if (w1 > nwin) w1 = nwin;
    uint64_t f = 0, r = 0;
    uint64_t run = 0; // consecutive base bytes ending at i
    const uint64_t base = COUNT_ONLY ? 0 : a.item_off[id];
    const uint64_t i_end = w1 + k - 1; // bytes [w0, i_end) cover windows [w0, w1)
    for (uint64_t i = w0; i < i_end; ++i) {
      const uint8_t cin = s[i];
      const bool vin = is_base(cin);
      run = vin ? run + 1 : 0;
      if (!COUNT_ONLY) {
        const uint32_t ci = code_of(cin);
        uint64_t tf = vin ? seed_of_code(ci) : 0;
        uint64_t tr = vin ? a.sk_rc[ci] : 0;
        if (i >= w0 + k) {
          const uint8_t cout = s[i - k];
          if (is_base(cout)) {
            const uint32_t co = code_of(cout);
            tf ^= a.sk_fwd[co];
            tr ^= seed_of_code(co ^ 2u);
          }
        }
        f = srol1(f) ^ tf;
        r = sror1(r ^ tr);
      }
      if (run >= k) {
        if (!COUNT_ONLY) {
          const uint64_t o = base + emitted;
          if (o < a.capacity) {
            const uint64_t h0 = f + r;
            a.hashes[o * a.m] = h0;
            for (uint32_t jj = 1; jj < a.m; ++jj) a.hashes[o * a.m + jj] = mix_hash(h0, a.mult[jj]);
            if (a.pos) a.pos[o] = (uint32_t)(i + 1 - k);
            if (a.fwd) a.fwd[o] = f;
            if (a.rev) a.rev[o] = r;
          }
        }
        emitted++;
      }
    }
  }
  if (a.counts) a.counts[id] = emitted;
}

} // namespace ntamd
