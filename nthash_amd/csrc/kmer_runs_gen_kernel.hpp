// kmer_runs_gen_kernel.hpp -- the run-split kernel for ANY fixed read shape.
//
// kmer_runs_kernel.hpp needs the run length C to divide the window count and
// stages whole reads; that leaves cliffs (a prime window count, reads of 10 kb).
// This kernel keeps its structure -- 64 lanes own 64 consecutive runs, wave-private
// 2-bit slab + output tile, prefetch of the next slab with a counted s_waitcnt --
// and generalises the geometry:
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
// Hash arithmetic is the same as in kmer_runs_kernel.hpp (first window from the byte
// tables, src/kmer.cpp:43-73,123-152; the rest rolled, src/kmer.cpp:84-94,164-174).
#pragma once

#include <hip/hip_runtime.h>

#include "kmer_runs_kernel.hpp"

namespace ntamd {

struct KmerRunsGenArgs {
  const uint8_t* seqs;
  uint64_t* hashes;      // dense [read][window][m]
  uint32_t* dirty;
  const uint4* init_tab; // global [ntab][256] {f.lo,f.hi,r.lo,r.hi}
  uint64_t n_reads;
  uint64_t n_runs;       // n_reads * rpr
  uint64_t n_wtiles;     // ceil(n_runs / 64)
  uint64_t total_bytes;  // (n_reads - 1) * stride + len
  uint32_t len, stride, k, m;
  uint32_t nwin;
  uint32_t C;            // windows per full run
  uint32_t rpr;          // runs per read = ceil(nwin / C)
  uint32_t last_start;   // first window of a read's last run = nwin - C
  uint32_t ntab;         // ceil(k/4)
  uint32_t waves;        // waves per block
  uint32_t bits_dwords;  // per-wave bit-stream capacity
  uint32_t tile_u64;     // per-wave tile capacity (64*C + 128)
  uint32_t inv_rpr;      // floor(65536 / rpr) + 1 (used when rpr <= 64)
  uint32_t tile_map;     // wave groups of the tile -> wave mapping (as in kmer_runs_kernel)
  uint64_t tab[16][2];
  uint64_t mult[KF_MAX_RUNTIME_M];
};

// s_waitcnt needs an immediate: wait until at most n (0..15) vector-memory operations are in flight
#define KRG_WAITCASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
__device__ __forceinline__ void wait_vmcnt_upto15(uint32_t n)
{
  switch (n) {
    KRG_WAITCASE(1) KRG_WAITCASE(2) KRG_WAITCASE(3) KRG_WAITCASE(4) KRG_WAITCASE(5)
    KRG_WAITCASE(6) KRG_WAITCASE(7) KRG_WAITCASE(8) KRG_WAITCASE(9) KRG_WAITCASE(10)
    KRG_WAITCASE(11) KRG_WAITCASE(12) KRG_WAITCASE(13) KRG_WAITCASE(14) KRG_WAITCASE(15)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}
#undef KRG_WAITCASE

// NW: window words, k <= 16*NW; DT: every slab is <= 1280 bytes (tail = one dword per lane)
template <int NW, bool DT>
__global__ __launch_bounds__(KR_MAX_THREADS) void kmer_runs_gen_kernel(const KmerRunsGenArgs a)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
  const uint32_t k = a.k, m = a.m, C = a.C, ntab = a.ntab, rpr = a.rpr;
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = tid >> 6;

  uint4* itab = (uint4*)lds_dyn;
  uint4* ptab = itab + ntab * 256u;
  uint64_t* mults = (uint64_t*)(ptab + 16);
  uint32_t* wave_base = (uint32_t*)(mults + KF_MAX_RUNTIME_M) + wave * (a.tile_u64 * 2u + a.bits_dwords);
  uint64_t* tile = (uint64_t*)wave_base;
  uint32_t* bits = wave_base + a.tile_u64 * 2u;

  for (uint32_t i = tid; i < ntab * 256u; i += blockDim.x) itab[i] = a.init_tab[i];
  if (tid < 16)
    ptab[tid] = make_uint4((uint32_t)a.tab[tid][0], (uint32_t)(a.tab[tid][0] >> 32),
                           (uint32_t)a.tab[tid][1], (uint32_t)(a.tab[tid][1] >> 32));
  if (tid < KF_MAX_RUNTIME_M) mults[tid] = a.mult[tid];
  __syncthreads(); // the only block-wide barrier

  uint32_t bad = 0;
  uint64_t wt, wstride, wt_end;
  {
    const uint64_t n_waves_total = (uint64_t)gridDim.x * a.waves;
    const uint64_t gw = (uint64_t)blockIdx.x * a.waves + wave;
    uint64_t groups = a.tile_map ? a.tile_map : 1u;
    if (groups > n_waves_total) groups = n_waves_total;
    const uint64_t wpg = n_waves_total / groups;
    uint64_t g = gw / wpg;
    if (g >= groups) g = groups - 1;
    const uint64_t w_in_g = gw - g * wpg;
    const uint64_t g_waves = g == groups - 1 ? n_waves_total - g * wpg : wpg;
    const uint64_t per = (a.n_wtiles + groups - 1) / groups;
    const uint64_t t0 = g * per;
    wt = t0 + w_in_g;
    wstride = g_waves;
    wt_end = t0 + per < a.n_wtiles ? t0 + per : a.n_wtiles;
  }
  uint64_t r_first = (wt * 64u) / rpr;
  uint32_t rem0 = (uint32_t)(wt * 64u - r_first * rpr);
  const uint64_t step_q = (wstride * 64u) / rpr;
  const uint32_t step_r = (uint32_t)(wstride * 64u - step_q * rpr);

  // gl = run index counted from run 0 of read r_first, gl < 64 + rpr:
  // read (relative) and first window of the run inside it
  auto split = [&](uint32_t gl, uint32_t& lr, uint32_t& w0) {
    lr = rpr > 64u ? (gl >= rpr ? 1u : 0u) : (gl * a.inv_rpr) >> 16;
    const uint32_t q = gl - lr * rpr;
    w0 = q == rpr - 1u ? a.last_start : q * C;
  };

  // geometry of the tile whose first run is run rm of read rf
  struct Geo {
    uint64_t byte0;  // offset of the first 16-byte vector from a.seqs (wraps below 0 by < 16)
    uint64_t out0;   // index of the tile's first k-mer in the dense stream
    uint32_t shift, slab_bytes, n_vec, runs_here;
    uint32_t n_kmers; // k-mers of this tile
    uint32_t w_first; // first window (inside its read) of the tile's first run
    uint32_t edge;    // the vectors of the slab reach outside the caller's buffer
  };
  auto geo_of = [&](uint64_t g0, uint64_t rf, uint32_t rm) -> Geo {
    Geo g;
    const uint64_t runs_left = a.n_runs - g0;
    g.runs_here = runs_left < 64u ? (uint32_t)runs_left : 64u;
    uint32_t lre, we;
    split(rm + g.runs_here - 1u, lre, we);
    const uint32_t w_first = rm == rpr - 1u ? a.last_start : rm * C;
    const uint64_t start = rf * a.stride + w_first;
    g.w_first = w_first;
    g.slab_bytes = lre * a.stride + we + C + k - 1u - w_first;
    g.n_kmers = lre * a.nwin + we + C - w_first;
    g.out0 = rf * a.nwin + w_first;
    g.shift = (uint32_t)(((uint64_t)a.seqs + start) & 15u);
    g.byte0 = start - g.shift;
    g.n_vec = (g.shift + g.slab_bytes + 15u) >> 4;
    g.edge = (start < g.shift || g.byte0 + ((uint64_t)g.n_vec << 4) > a.total_bytes) ? 1u : 0u;
    return g;
  };
  // Bytes of the batch next to the slab inside its first / last vector are judged
  // too (a non-base there makes the batch dirty anyway); bytes outside the caller's
  // buffer are not: they exist only in the slabs flagged `edge`.
  auto pack_vec = [&](const Geo& sl, uint32_t i, const uint4 v) {
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
  auto pack_dword = [&](const Geo& sl, uint32_t j, const uint32_t wv) {
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
    bad |= b;
    ((uint8_t*)bits)[256u + j] = (uint8_t)p;
  };
  auto stage = [&](const Geo& sl, uint32_t first) {
    for (uint32_t i = first + lane; i < sl.n_vec; i += 64u)
      pack_vec(sl, i, *(const uint4*)(a.seqs + sl.byte0 + ((uint64_t)i << 4)));
    if (lane < (uint32_t)NW + 5u) bits[sl.n_vec + lane] = 0; // funnels read a little ahead
  };
  auto lds_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  };

  Geo cur;
  cur.byte0 = cur.out0 = 0;
  cur.shift = cur.slab_bytes = cur.n_vec = cur.runs_here = cur.n_kmers = cur.w_first = 0;
  cur.edge = 1u;
  if (wt < wt_end) {
    cur = geo_of(wt * 64u, r_first, rem0);
    stage(cur, 0u);
  }
  for (; wt < wt_end; wt += wstride) {
    lds_sync();
    const uint32_t shift = cur.shift, runs_here = cur.runs_here;
    const uint32_t my_rem0 = rem0;
    // ---- issue the loads of the NEXT tile's slab ---------------------------------
    r_first += step_q;
    rem0 += step_r;
    if (rem0 >= rpr) { rem0 -= rpr; r_first += 1; }
    const uint64_t nwt = wt + wstride;
    const bool have_next = nwt < wt_end;
    Geo nxt = cur;
    if (have_next) nxt = geo_of(nwt * 64u, r_first, rem0);
    v4u pv0, pv1;
    uint32_t pw;
    uint32_t dirty_seen;
    {
      const uint32_t i0 = lane < nxt.n_vec ? lane : 0u;
      const uint8_t* p0 = a.seqs + nxt.byte0 + ((uint64_t)i0 << 4);
      if constexpr (DT) {
        const uint32_t n_dw = (nxt.shift + nxt.slab_bytes + 3u) >> 2;
        const uint32_t j = 256u + lane < n_dw ? 256u + lane : 0u;
        const uint8_t* p1 = a.seqs + nxt.byte0 + ((uint64_t)j << 2);
        asm volatile("global_load_dword %2, %5, off sc1\n\tglobal_load_dwordx4 %0, %3, off\n\t"
                     "global_load_dword %1, %4, off"
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
    }

    // ---- this lane's run ----------------------------------------------------------
    // lanes past the end of the last tile redo the tile's first run (same values, same addresses)
    uint32_t lr, w0;
    split(lane < runs_here ? my_rem0 + lane : my_rem0, lr, w0);
    const uint32_t b0 = shift + lr * a.stride + w0 - cur.w_first; // first base of the first window
    // the tile is built shifted by the position of its first stream element inside a
    // 1 KiB block of the output (m == 1): every store instruction of the copy-out then
    // covers one aligned KiB, instead of every wave splitting cache lines with its neighbours
    const uint32_t tpar = m == 1u ? (uint32_t)(cur.out0 & 127u) : 0u;
    uint64_t* my_row = tile + tpar + (lr * a.nwin + w0 - cur.w_first);
    const uint32_t d0 = b0 >> 4, sh0 = (b0 & 15u) << 1;
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
    uint32_t f_lo = 0, f_hi = 0, r_lo = 0, r_hi = 0;
#pragma unroll
    for (int jt = 0; jt < 4 * NW; ++jt) {
      if ((uint32_t)jt < ntab) {
        const uint32_t byte = (w[jt >> 2] >> ((jt & 3) * 8)) & 0xFFu;
        const uint4 e = itab[(uint32_t)jt * 256u + byte];
        f_lo ^= e.x; f_hi ^= e.y; r_lo ^= e.z; r_hi ^= e.w;
      }
    }
    my_row[0] = (((uint64_t)f_hi << 32) | f_lo) + (((uint64_t)r_hi << 32) | r_lo);

    // remaining C-1 windows: roll.  step t: in = base b0+k-1+t, out = base b0+t-1
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
        srol_pair(f_lo, f_hi);
        f_lo ^= term.x;
        f_hi ^= term.y;
        r_lo ^= term.z;
        r_hi ^= term.w;
        sror_pair(r_lo, r_hi);
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
          my_row[jw * 16u + i0 + i + 1u] = (((uint64_t)f_hi << 32) | f_lo) + (((uint64_t)r_hi << 32) | r_lo);
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

    // ---- copy the tile out: n_kmers * m consecutive values of the hash stream -----
    lds_sync();
    uint32_t n_counted; // store instructions surely issued after the prefetch loads
    if (m == 1u) {
      const uint32_t span = tpar + cur.n_kmers;
      const uint32_t pieces = (span + 1u) >> 1;
      const uint32_t first = tpar >> 1;                    // first piece that holds a value
      uint64_t* const base = a.hashes + (cur.out0 - tpar); // 1 KiB aligned
      for (uint32_t pi = lane; pi < pieces; pi += 64u) {
        const uint4 dv = *(const uint4*)(tile + 2u * pi);
        const bool lo_ok = 2u * pi >= tpar;
        const bool hi_ok = 2u * pi + 1u >= tpar && 2u * pi + 1u < span;
        if (lo_ok && hi_ok) *(uint4*)(base + 2u * pi) = dv;
        else if (lo_ok) *(uint2*)(base + 2u * pi) = make_uint2(dv.x, dv.y);
        else if (hi_ok) *(uint2*)(base + 2u * pi + 1u) = make_uint2(dv.z, dv.w);
      }
      // iterations in which some lane surely stores a whole piece
      const uint32_t it_lo = (first + 64u) >> 6, it_hi = pieces >> 6;
      n_counted = it_hi > it_lo ? it_hi - it_lo : 0u;
    } else {
      // multi-hash expansion (extend_hashes, src/internal.hpp:104-118) fused into the
      // copy-out: stream value v is h[v % m] of k-mer v / m
      const uint64_t v0 = cur.out0 * m;
      const uint32_t vpar = (uint32_t)(v0 & 1u);
      const uint32_t n_vals = cur.n_kmers * m;
      const uint32_t span = vpar + n_vals;
      const uint32_t pieces = (span + 1u) >> 1;
      uint64_t* const base = a.hashes + (v0 - vpar);
      for (uint32_t pi = lane; pi < pieces; pi += 64u) {
        uint64_t o[2];
        bool ok[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const uint32_t sv = 2u * pi + (uint32_t)h - vpar; // wraps for the skipped head half
          ok[h] = sv < n_vals;
          const uint32_t e = ok[h] ? sv / m : 0u, jj = ok[h] ? sv - e * m : 0u;
          const uint64_t h0 = tile[e];
          o[h] = jj == 0 ? h0 : mix_hash(h0, mults[jj & (KF_MAX_RUNTIME_M - 1)]);
        }
        if (ok[0] && ok[1])
          *(uint4*)(base + 2u * pi) =
              make_uint4((uint32_t)o[0], (uint32_t)(o[0] >> 32), (uint32_t)o[1], (uint32_t)(o[1] >> 32));
        else if (ok[0]) *(uint2*)(base + 2u * pi) = make_uint2((uint32_t)o[0], (uint32_t)(o[0] >> 32));
        else if (ok[1]) *(uint2*)(base + 2u * pi + 1u) = make_uint2((uint32_t)o[1], (uint32_t)(o[1] >> 32));
      }
      n_counted = pieces >> 6;
    }
    lds_sync(); // tile and bits are free again

    // ---- consume the prefetched slab ------------------------------------------------
    // vmcnt retires in order: once at most n_counted operations are in flight, the
    // three loads issued before those stores have landed (never count a store that
    // might not have been issued: an iteration with all 64 lanes active always is)
    wait_vmcnt_upto15(n_counted < 15u ? n_counted : 15u);
    if constexpr (DT) asm volatile("" : "+v"(pv0), "+v"(pw), "+v"(dirty_seen)::"memory");
    else asm volatile("" : "+v"(pv0), "+v"(pv1), "+v"(dirty_seen)::"memory");
    if (__builtin_amdgcn_readfirstlane(dirty_seen) != 0u) break;
    if (have_next) {
      cur = nxt;
      if (lane < cur.n_vec) pack_vec(cur, lane, make_uint4(pv0.x, pv0.y, pv0.z, pv0.w));
      if constexpr (DT) {
        const uint32_t n_dw = (cur.shift + cur.slab_bytes + 3u) >> 2;
        if (256u + lane < n_dw) pack_dword(cur, lane, pw);
        if (lane < (uint32_t)NW + 5u) bits[cur.n_vec + lane] = 0;
      } else {
        if (lane + 64u < cur.n_vec) pack_vec(cur, lane + 64u, make_uint4(pv1.x, pv1.y, pv1.z, pv1.w));
        stage(cur, 128u);
      }
      if (__ballot(bad != 0) != 0) {
        if (lane == 0) atomicOr(a.dirty, 1u);
        break;
      }
    }
  }
  if (__ballot(bad != 0) != 0 && lane == 0) atomicOr(a.dirty, 1u);
}

} // namespace ntamd
