// kmer_ragged_kernel.hpp -- run-split k-mer hashing of VARIABLE-LENGTH reads
// with N-aware compaction: the path real FASTQ batches take.  A read is a span
// [starts[r], ends[r]) of one device buffer -- n_reads+1 offsets of concatenated reads
// (starts = offsets, ends = offsets + 1) or the sequence lines of a raw FASTQ chunk
// indexed on the device (fastx_kernels.hpp).
//
// Same decomposition as kmer_runs_na_kernel (runs of up to C windows, 64 runs per
// wave tile, count pass -> scan -> compact hash pass), but a read contributes
// ceil(windows / C) runs and the last one may be short.  A device pre-pass
// (ragged_* kernels below) lists the reads that have at least one window, scans
// their run counts and tells every tile which listed read its first run belongs
// to.  A tile then touches at most 64 listed reads; each of them stages exactly
// the bytes its runs in this tile need, packed back to back (16-base aligned per
// read) in the wave's LDS stream, so reads without windows, and the parts of a
// long read that belong to other tiles, cost nothing.
#pragma once

#include <hip/hip_runtime.h>

#include "kmer_runs_gen_kernel.hpp"

namespace ntamd {

enum : int { NA_MODE_COUNT = 1, NA_MODE_HASH = 2 };

// ---- pre-pass ---------------------------------------------------------------
// runs per read and a 0/1 flag "has a window"
static __global__ __launch_bounds__(256) void ragged_runs_kernel(const uint64_t* __restrict__ starts,
                                                         const uint64_t* __restrict__ ends, uint64_t n_reads,
                                                         uint32_t k, uint32_t C, uint64_t* __restrict__ rc,
                                                         uint64_t* __restrict__ flag)
{
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_reads) return;
  const uint64_t len = ends[r] > starts[r] ? ends[r] - starts[r] : 0;
  const uint64_t runs = len >= k ? (len - k + 1 + C - 1) / C : 0;
  rc[r] = runs;
  flag[r] = runs ? 1 : 0;
}

// everything a tile needs to know about a listed read, in one 32-byte record (one load level)
struct __attribute__((aligned(32))) NzMeta {
  uint64_t read;  // index in the caller's batch
  uint64_t rc;    // runs
  uint64_t start; // first byte in the buffer
  uint64_t len;   // bytes
};

// compact list of the reads that have runs: nz_meta[j], and nz_rc[j] on its own for the scan
static __global__ __launch_bounds__(256) void ragged_scatter_kernel(const uint64_t* __restrict__ rc,
                                                            const uint64_t* __restrict__ nz_idx,
                                                            const uint64_t* __restrict__ starts,
                                                            const uint64_t* __restrict__ ends, uint64_t n_reads,
                                                            NzMeta* __restrict__ nz_meta,
                                                            uint64_t* __restrict__ nz_rc)
{
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_reads || rc[r] == 0) return;
  const uint64_t j = nz_idx[r];
  NzMeta mm;
  mm.read = r;
  mm.rc = rc[r];
  mm.start = starts[r];
  mm.len = ends[r] - starts[r];
  nz_meta[j] = mm;
  nz_rc[j] = rc[r];
}

// per tile: the listed read that holds run 64*t, and how many of its runs precede it
static __global__ __launch_bounds__(256) void ragged_tiles_kernel(const uint64_t* __restrict__ nz_run_base, uint64_t n_nz,
                                                          uint64_t n_tiles, uint64_t* __restrict__ tile_j0,
                                                          uint64_t* __restrict__ tile_rem0)
{
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tiles) return;
  const uint64_t g0 = t * 64u;
  uint64_t lo = 0, hi = n_nz; // last j with nz_run_base[j] <= g0
  while (hi - lo > 1) {
    const uint64_t mid = lo + (hi - lo) / 2;
    if (nz_run_base[mid] <= g0) lo = mid; else hi = mid;
  }
  tile_j0[t] = lo;
  tile_rem0[t] = g0 - nz_run_base[lo];
}

// ---- main kernel --------------------------------------------------------------
struct KmerRaggedArgs {
  const uint8_t* seqs;
  const uint64_t* starts;    // read r = bytes [starts[r], ends[r]) of seqs
  const uint64_t* ends;
  uint64_t total_bytes;      // size of the seqs buffer (no load goes past it)
  uint64_t* hashes;
  uint32_t* pos;
  uint64_t* counts;          // optional (count pass, zeroed by the host): per-read emitted windows
  uint64_t* tile_counts;
  const uint64_t* tile_off;
  const NzMeta* nz_meta;
  const uint64_t* tile_j0;
  const uint64_t* tile_rem0;
  const uint4* init_tab;
  uint64_t n_nz, total_runs, n_wtiles;
  uint32_t k, m, C, ntab;
  uint32_t waves, bits_dwords, vbits_dwords, tile_u64; // count pass: tile_u64 = bits_dwords = 0 (validity only)
  uint32_t ptile_dwords;                               // position tile, 0 unless pos is wanted
  uint32_t value_sel;                                  // 0: canonical hash (+ mixes), 1: forward, 2: reverse strand (m == 1)
  uint64_t tab[16][2];
  uint64_t mult[KF_MAX_RUNTIME_M];
};

template <int MODE, int NW>
__global__ __launch_bounds__(KR_MAX_THREADS) void kmer_ragged_kernel(const KmerRaggedArgs a)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  const uint32_t k = a.k, m = a.m, C = a.C;
  const uint32_t inv_m = 0xFFFFFFFFu / m + 1u; // v / m == umulhi(v, inv_m) for v < 2^29
  const uint64_t kmul = (uint64_t)k * MULTISEED; // h[i] = mix(h[0] * (i ^ k*MULTISEED)), any m
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6); // uniform: tile bookkeeping on the scalar unit

  // LDS: tables | pair table | multipliers | per wave {hash tile, pos tile, bits, vbits, 2 read tables}
  uint4* itab = (uint4*)lds_dyn;
  uint4* ptab = itab + a.ntab * 256u;
  uint64_t* mults = (uint64_t*)(ptab + 16);
  // per listed read: run begin, vector begin, q_lo, windows, read index (u64), first staged byte (u64)
  constexpr uint32_t RT_DWORDS = 64 * 8;
  const uint32_t per_wave = a.tile_u64 * 2u + a.ptile_dwords + a.bits_dwords + a.vbits_dwords + 2u * RT_DWORDS;
  // the count pass keeps no tables, tile or code stream: a few KB per wave, so it runs at full occupancy
  uint32_t* wave_base = (MODE == NA_MODE_HASH ? (uint32_t*)(mults + KF_MAX_RUNTIME_M) : lds_dyn) + wave * per_wave;
  uint64_t* tile = (uint64_t*)wave_base;
  uint32_t* ptile = wave_base + a.tile_u64 * 2u;
  uint32_t* bits = ptile + a.ptile_dwords;
  uint16_t* vbits = (uint16_t*)(bits + a.bits_dwords);
  uint32_t* rt_base = bits + a.bits_dwords + a.vbits_dwords; // two tables: this tile's and the next one's
#define RT_BEGIN(sel) (rt_base + (sel) * RT_DWORDS)                     /* first run (tile-relative) of listed read j */
#define RT_VBEG(sel) (rt_base + (sel) * RT_DWORDS + 64)                 /* first staged vector of read j */
#define RT_QLO(sel) (rt_base + (sel) * RT_DWORDS + 128)                 /* first run of read j in this tile */
#define RT_NWIN(sel) (rt_base + (sel) * RT_DWORDS + 192)                /* windows of read j (saturated) */
#define RT_READ(sel) ((uint64_t*)(rt_base + (sel) * RT_DWORDS + 256))   /* read index */
#define RT_ADDR(sel) ((uint64_t*)(rt_base + (sel) * RT_DWORDS + 384))   /* byte offset of the first staged byte */

  if (MODE == NA_MODE_HASH) {
    for (uint32_t i = tid; i < a.ntab * 256u; i += blockDim.x) itab[i] = a.init_tab[i];
    if (tid < 16)
      ptab[tid] = make_uint4((uint32_t)a.tab[tid][0], (uint32_t)(a.tab[tid][0] >> 32),
                             (uint32_t)a.tab[tid][1], (uint32_t)(a.tab[tid][1] >> 32));
    if (tid < KF_MAX_RUNTIME_M) mults[tid] = a.mult[tid];
  }
  __syncthreads();

  auto lds_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  };
  auto wave_incl_scan32 = [&](uint32_t v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(v, d, 64);
      if ((int)lane >= d) v += o;
    }
    return v;
  };
  // last j in [0,64) with arr[j] <= x (arr non-decreasing, arr[0] <= x)
  auto search64 = [&](const uint32_t* arr, uint32_t x) {
    uint32_t lo = 0;
#pragma unroll
    for (uint32_t step = 32; step > 0; step >>= 1)
      if (lo + step < 64u && arr[lo + step] <= x) lo += step;
    return lo;
  };

  const uint64_t per_block = (a.n_wtiles + gridDim.x - 1) / gridDim.x;
  const uint64_t t_begin = (uint64_t)blockIdx.x * per_block;
  const uint64_t t_end = t_begin + per_block < a.n_wtiles ? t_begin + per_block : a.n_wtiles;
  const uint64_t wstride = a.waves;
  uint64_t wt = t_begin + wave;
  if (wt >= t_end) return;

  // Software pipeline (the wave is alone with its latencies: 8 waves per CU).  A tile's bytes sit
  // behind three dependent loads: tile -> first listed read (scalar loads, two tiles ahead), that
  // read's 32-byte record (one tile ahead), the bytes themselves (issued for the NEXT tile before
  // this one is hashed).  Record and byte loads are inline asm consumed after this tile's stores
  // with a counted s_waitcnt, as in kmer_runs_kernel.hpp (vmcnt retires in order).
  v4u meta_lo, meta_hi;        // NzMeta of the tile whose table is built next
  v4u st0, st1, st2;           // staged vectors lane, lane+64, lane+128 of the current tile
  uint32_t st_unsafe = 0;      // bit i: vector i was not loaded (16 bytes would cross the end of the buffer)
  auto issue_meta = [&](uint64_t j0_) {
    uint64_t jj = j0_ + lane;
    if (jj >= a.n_nz) jj = a.n_nz - 1;
    const NzMeta* p = a.nz_meta + jj;
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"
                 : "=&v"(meta_lo), "=&v"(meta_hi)
                 : "v"(p)
                 : "memory");
  };
  auto runs_of = [&](uint64_t t) -> uint32_t {
    const uint64_t left = a.total_runs - t * 64u;
    return left < 64u ? (uint32_t)left : 64u;
  };
  // read table of a tile from the records in meta_lo/meta_hi; returns the number of staged vectors
  auto build_table = [&](uint32_t sel, uint64_t j0, uint32_t rem0, uint32_t runs_here) -> uint32_t {
    uint32_t take = 0, q_lo = 0, nv = 0, nwin32 = 0;
    uint64_t rd = 0, byte0 = 0;
    uint64_t avail = 0, len = 0, start = 0;
    if (j0 + lane < a.n_nz) {
      rd = ((uint64_t)meta_lo.y << 32) | meta_lo.x;
      const uint64_t rcj = ((uint64_t)meta_lo.w << 32) | meta_lo.z;
      start = ((uint64_t)meta_hi.y << 32) | meta_hi.x;
      len = ((uint64_t)meta_hi.w << 32) | meta_hi.z;
      q_lo = lane == 0 ? rem0 : 0u;
      avail = rcj - q_lo;
    }
    const uint32_t av32 = avail > 64u ? 64u : (uint32_t)avail; // a tile never takes more than 64 runs
    const uint32_t end = wave_incl_scan32(av32);
    const uint32_t begin = end - av32;
    take = begin < runs_here ? (runs_here - begin < av32 ? runs_here - begin : av32) : 0u;
    const uint64_t nwin = len >= k ? len - k + 1 : 0;
    nwin32 = nwin > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)nwin;
    if (take) {
      const uint64_t first_b = (uint64_t)q_lo * C;
      uint64_t last_b = (uint64_t)(q_lo + take) * C + k - 1; // one past the last needed byte
      if (last_b > len) last_b = len;
      nv = (uint32_t)((last_b - first_b + 15u) >> 4);
      byte0 = start + first_b;
    }
    const uint32_t vend = wave_incl_scan32(nv);
    RT_BEGIN(sel)[lane] = take ? begin : 0xFFFFFFFFu; // reads past the tile never match a search
    RT_VBEG(sel)[lane] = take ? vend - nv : 0xFFFFFFFFu;
    RT_QLO(sel)[lane] = q_lo;
    RT_NWIN(sel)[lane] = nwin32;
    RT_READ(sel)[lane] = rd;
    RT_ADDR(sel)[lane] = byte0;
    return __shfl(vend, 63, 64);
  };
  // byte offset of staged vector v of the tile whose table is `sel`
  auto vec_offset = [&](uint32_t sel, uint32_t v) -> uint64_t {
    const uint32_t j = search64(RT_VBEG(sel), v);
    return RT_ADDR(sel)[j] + ((uint64_t)(v - RT_VBEG(sel)[j]) << 4);
  };
  auto issue_stage = [&](uint32_t sel, uint32_t v_total) {
    const uint8_t* p[3];
    st_unsafe = 0;
#pragma unroll
    for (uint32_t i = 0; i < 3; ++i) {
      const uint32_t v = lane + 64u * i;
      uint64_t off = 0;
      if (v < v_total) {
        off = vec_offset(sel, v);
        if (off + 16u > a.total_bytes) { // the very end of the buffer: load nothing there, redo bytewise
          st_unsafe |= 1u << i;
          off = 0;
        }
      }
      p[i] = a.seqs + off;
    }
    // (a batch shorter than 16 bytes never gets here with a safe vector; offset 0 of such a batch is
    // read as the aligned 16 bytes around seqs, which stay inside its page)
    if (a.total_bytes < 16u) {
#pragma unroll
      for (uint32_t i = 0; i < 3; ++i) p[i] = (const uint8_t*)((uintptr_t)a.seqs & ~(uintptr_t)15);
    }
    asm volatile("global_load_dwordx4 %0, %3, off" KRG_LOAD_NT "\n\tglobal_load_dwordx4 %1, %4, off" KRG_LOAD_NT "\n\t"
                 "global_load_dwordx4 %2, %5, off" KRG_LOAD_NT
                 : "=&v"(st0), "=&v"(st1), "=&v"(st2)
                 : "v"(p[0]), "v"(p[1]), "v"(p[2])
                 : "memory");
  };
  auto pack_one = [&](uint32_t v, uint4 x) {
    uint32_t i0, i1, i2, i3;
    const uint32_t c0 = pack4v(x.x, i0), c1 = pack4v(x.y, i1), c2 = pack4v(x.z, i2), c3 = pack4v(x.w, i3);
    if (MODE == NA_MODE_HASH) bits[v] = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
    vbits[v] = (uint16_t)(i0 | (i1 << 4) | (i2 << 8) | (i3 << 12));
  };
  auto load_plain = [&](uint32_t sel, uint32_t v) -> uint4 {
    const uint64_t off = vec_offset(sel, v);
    uint4 x;
    if (off + 16u <= a.total_bytes) {
      __builtin_memcpy(&x, a.seqs + off, 16); // unaligned 16-byte load (one global_load_dwordx4)
    } else { // never read past the caller's buffer
      uint32_t wv[4] = {0, 0, 0, 0};
      for (uint32_t b = 0; b < 16u && off + b < a.total_bytes; ++b)
        wv[b >> 2] |= (uint32_t)a.seqs[off + b] << ((b & 3u) * 8u);
      x = make_uint4(wv[0], wv[1], wv[2], wv[3]);
    }
    return x;
  };

  // ---- prologue: tile records of the first two tiles, first table, first staging loads ----
  uint64_t j0_cur = a.tile_j0[wt], j0_nxt = 0;
  uint32_t rem0_cur = (uint32_t)a.tile_rem0[wt], rem0_nxt = 0;
  if (wt + wstride < t_end) { j0_nxt = a.tile_j0[wt + wstride]; rem0_nxt = (uint32_t)a.tile_rem0[wt + wstride]; }
  uint32_t cur = 0;
  issue_meta(j0_cur);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("; NTLINT_CONSUME %0 %1" : "+v"(meta_lo), "+v"(meta_hi)::"memory");
  uint32_t v_total = build_table(cur, j0_cur, rem0_cur, runs_of(wt));
  lds_sync();
  issue_stage(cur, v_total);
  if (wt + wstride < t_end) issue_meta(j0_nxt);
  uint32_t n_counted = 0;                  // stores surely issued after the loads in flight
  uint64_t acc_read = ~0ull, acc_cnt = 0;  // count pass: running per-read count of this wave

  for (; wt < t_end; wt += wstride) {
    const uint32_t runs_here = runs_of(wt);
    const bool have_next = wt + wstride < t_end;
    // ---- the staged vectors of this tile (and the records of the next) have landed ------------
    wait_vmcnt_upto15(n_counted < 15u ? n_counted : 15u);
    asm volatile("; NTLINT_CONSUME %0 %1 %2 %3 %4" : "+v"(st0), "+v"(st1), "+v"(st2), "+v"(meta_lo), "+v"(meta_hi)::"memory");
    {
      const uint4 x0 = make_uint4(st0.x, st0.y, st0.z, st0.w), x1 = make_uint4(st1.x, st1.y, st1.z, st1.w),
                  x2 = make_uint4(st2.x, st2.y, st2.z, st2.w);
      if (lane < v_total) pack_one(lane, (st_unsafe & 1u) ? load_plain(cur, lane) : x0);
      if (lane + 64u < v_total) pack_one(lane + 64u, (st_unsafe & 2u) ? load_plain(cur, lane + 64u) : x1);
      if (lane + 128u < v_total) pack_one(lane + 128u, (st_unsafe & 4u) ? load_plain(cur, lane + 128u) : x2);
      for (uint32_t v = lane + 192u; v < v_total; v += 64u) pack_one(v, load_plain(cur, v)); // long k only
    }
    if (MODE == NA_MODE_HASH && lane < (uint32_t)NW + 3u) bits[v_total + lane] = 0;
    if (lane < 10u) vbits[v_total + lane] = 0xFFFFu;
    // ---- next tile: table, staging loads, and the records of the tile after it ------------------
    uint32_t v_total_nxt = 0;
    if (have_next) v_total_nxt = build_table(cur ^ 1u, j0_nxt, rem0_nxt, runs_of(wt + wstride));
    lds_sync();
    uint64_t j0_n2 = 0;
    uint32_t rem0_n2 = 0;
    if (have_next) {
      issue_stage(cur ^ 1u, v_total_nxt);
      if (wt + 2u * wstride < t_end) {
        j0_n2 = a.tile_j0[wt + 2u * wstride];
        rem0_n2 = (uint32_t)a.tile_rem0[wt + 2u * wstride];
        issue_meta(j0_n2);
      }
    }

    // ---- this lane's run -------------------------------------------------------------
    const bool live = lane < runs_here;
    const uint32_t j = search64(RT_BEGIN(cur), live ? lane : 0u);
    const uint32_t q = RT_QLO(cur)[j] + ((live ? lane : 0u) - RT_BEGIN(cur)[j]);
    const uint32_t nwin_j = RT_NWIN(cur)[j];
    const uint64_t w_first = (uint64_t)q * C;
    const uint32_t c_run = !live ? 0u : (nwin_j - w_first < C ? (uint32_t)(nwin_j - w_first) : C);
    const uint32_t b0 = (RT_VBEG(cur)[j] << 4) + (q - RT_QLO(cur)[j]) * C;
    const uint32_t valid = ~(k <= 64u ? windows_with_non_base((const uint32_t*)vbits, b0, k)
                                      : windows_with_non_base_long((const uint32_t*)vbits, b0, k, C)) &
                           ((1u << c_run) - 1u); // C <= 16
    const uint32_t cnt = __builtin_popcount(valid);
    const uint32_t incl = wave_incl_scan32(cnt);
    const uint32_t lane_off = incl - cnt;
    const uint32_t total = __shfl(incl, 63, 64);

    if (MODE == NA_MODE_COUNT) {
      if (lane == 0) a.tile_counts[wt] = total;
      if (a.counts) {
        // Per-read counts.  A long read owns thousands of consecutive tiles: adding every lane's count to
        // its one address serialises the whole pass (1 s for a 3 Gbp genome).  A wave keeps the count of
        // the read it is in and adds it once, when the read changes; tiles that mix reads (short reads,
        // distinct addresses) add per lane.
        const uint64_t my_read = RT_READ(cur)[j];
        const uint64_t r0 = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(my_read >> 32)) << 32) |
                            __builtin_amdgcn_readfirstlane((uint32_t)my_read);
        if (__ballot(live && my_read != r0) == 0) {
          if (r0 != acc_read) {
            if (acc_cnt && lane == 0) atomicAdd((unsigned long long*)&a.counts[acc_read], (unsigned long long)acc_cnt);
            acc_read = r0;
            acc_cnt = 0;
          }
          acc_cnt += total;
        } else if (cnt) {
          atomicAdd((unsigned long long*)&a.counts[my_read], (unsigned long long)cnt);
        }
      }
      n_counted = 0;
    } else {
      // ---- hash the run, drop valid hashes at their compacted slots ------------------------
      const uint32_t d0 = b0 >> 4, sh0 = (b0 & 15u) << 1;
      uint32_t f_lo = 0, f_hi = 0, r_lo = 0, r_hi = 0;
      if constexpr (NW == 0) { // any k: Horner first window (kmer_runs_gen_kernel.hpp)
        any_k_first_window(bits, itab, b0, k, f_lo, f_hi, r_lo, r_hi);
      } else {
        uint32_t w[NW];
        uint32_t lo = bits[d0];
#pragma unroll
        for (int i = 0; i < NW; ++i) {
          const uint32_t hi = bits[d0 + i + 1];
          w[i] = funnel(hi, lo, sh0);
          lo = hi;
        }
#pragma unroll
        for (int jt = 0; jt < 4 * NW; ++jt) { // a.ntab == 4 * NW (zero tables past ceil(k/4)): no branch, lookups in flight together
          const uint32_t byte = (w[jt >> 2] >> ((jt & 3) * 8)) & 0xFFu;
          const uint4 e = itab[(uint32_t)jt * 256u + byte];
          f_lo ^= e.x; f_hi ^= e.y; r_lo ^= e.z; r_hi ^= e.w;
        }
      }
      uint32_t slot = lane_off;
      const bool want_pos = a.pos != nullptr;
      const uint32_t p_first = (uint32_t)w_first;
      const uint64_t o0 = a.tile_off[wt];
      const uint32_t tpar = m == 1u ? (uint32_t)(o0 & (KRG_ALIGN_U64 - 1u)) : 0u;
      auto emit = [&](uint32_t jw) {
        if ((valid >> jw) & 1u) {
          tile[tpar + slot] = a.value_sel == 0u   ? canon_pair(f_lo, f_hi, r_lo, r_hi)
                              : a.value_sel == 1u ? (((uint64_t)f_hi << 32) | f_lo)
                                                  : (((uint64_t)r_hi << 32) | r_lo);
          if (want_pos) ptile[slot] = p_first + jw;
          ++slot;
        }
      };
      emit(0u);
      const uint32_t bi = b0 + k;
      const uint32_t di = bi >> 4, shi = (bi & 15u) << 1;
      for (uint32_t jw = 0; jw * 16u + 1u < C; ++jw) {
        const uint32_t w_in = funnel(bits[di + jw + 1], bits[di + jw], shi);
        const uint32_t w_out = funnel(bits[d0 + jw + 1], bits[d0 + jw], sh0);
        const uint32_t u = ((w_in & 0x33333333u) << 2) | (w_out & 0x33333333u);
        const uint32_t v = (w_in & 0xCCCCCCCCu) | ((w_out >> 2) & 0x33333333u);
        auto lookup = [&](uint32_t i) -> uint4 {
          const uint32_t src = (i & 1u) ? v : u;
          const uint32_t toff = ((src >> ((i >> 1) * 4u)) & 0xFu) << 4;
          return *(const uint4*)((const char*)ptab + toff);
        };
        auto roll = [&](const uint4 term) {
          roll_step(f_lo, f_hi, r_lo, r_hi, term);
        };
        // table terms do not depend on the hash state: fetch a batch ahead of the dependent chain
        auto batch = [&](uint32_t i0, auto n_tag) {
          constexpr uint32_t N = decltype(n_tag)::value;
          uint4 terms[N];
#pragma unroll
          for (uint32_t i = 0; i < N; ++i) terms[i] = lookup(i0 + i);
#pragma unroll
          for (uint32_t i = 0; i < N; ++i) {
            roll(terms[i]);
            emit(jw * 16u + i0 + i + 1u);
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
      lds_sync();
      // copy-out as in kmer_runs_gen_kernel.hpp: the tile was built shifted by tpar = o0 mod 128, every store
      // instruction covers one aligned KiB of the stream
      if (m == 1) {
        const uint32_t span = tpar + total;
        const uint32_t pieces = (span + 1u) >> 1;
        uint64_t* const base = a.hashes + (o0 - tpar);
        for (uint32_t pi = lane; pi < pieces; pi += 64u) {
          const uint4 dv = *(const uint4*)(tile + 2u * pi);
          const bool lo_ok = 2u * pi >= tpar && 2u * pi < span;
          const bool hi_ok = 2u * pi + 1u >= tpar && 2u * pi + 1u < span;
#if defined(KRR_ST_PLAIN)
          if (lo_ok && hi_ok) *(uint4*)(base + 2u * pi) = dv;
#else     // streaming stores for the whole pieces, as in the general kernel
          if (lo_ok && hi_ok) __builtin_nontemporal_store(*(const nt_v4u*)&dv, (nt_v4u*)(base + 2u * pi));
#endif
          else if (lo_ok) *(uint2*)(base + 2u * pi) = make_uint2(dv.x, dv.y);
          else if (hi_ok) *(uint2*)(base + 2u * pi + 1u) = make_uint2(dv.z, dv.w);
        }
        const uint32_t it_lo = ((tpar >> 1) + 64u) >> 6, it_hi = pieces >> 6;
        n_counted = it_hi > it_lo ? it_hi - it_lo : 0u;
      } else {
        n_counted = multi_hash_copy_out<false>(tile, a.hashes, o0, total, m, inv_m, kmul, lane);
      }
      if (want_pos)
        for (uint32_t e = lane; e < total; e += 64u) a.pos[o0 + e] = ptile[e];
      lds_sync(); // tile, bits and this tile's table are free again
    }
    // ---- rotate the pipeline -----------------------------------------------------------------
    cur ^= 1u;
    v_total = v_total_nxt;
    j0_cur = j0_nxt; rem0_cur = rem0_nxt;
    j0_nxt = j0_n2; rem0_nxt = rem0_n2;
  }
  // (no hidden load is in flight here -- the last iteration issues none -- but that is a property of the loop's
  //  conditions, not of the flow graph: the registers are reused below, so say it in a way the ISA lint can see)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (MODE == NA_MODE_COUNT && a.counts && acc_cnt && lane == 0)
    atomicAdd((unsigned long long*)&a.counts[acc_read], (unsigned long long)acc_cnt);
#undef RT_BEGIN
#undef RT_VBEG
#undef RT_QLO
#undef RT_NWIN
#undef RT_READ
#undef RT_ADDR
}

} // namespace ntamd
