// capi_kmer_na.hip -- N-aware run-split path for fixed-length reads: count pass -> scan -> compact hash pass
// Part of libnthash_hip.so (include/nthash_hip.h); see capi_internal.hpp for the file map.
#include "capi_internal.hpp"
#include "util_kernels.hpp"

using namespace ntamd;
using namespace ntamd::host;

int ntamd::host::run_kmer_na(nthip_ctx* c, const Staged& st, const nthip_reads* rd, uint32_t k, uint32_t m,
                const NaPlan& plan, const KmerFixedArgs& consts, uint64_t capacity, uint64_t* total, const uint16_t* invalid)
{
  const bool packed = invalid != nullptr;
  KmerRunsGenArgs a;
  fill_gen_args(a, c, st, rd, k, m, plan.g, consts);
  a.invalid = invalid;
  NTCHK(get_kmer_tab(c, k, &a.init_tab));
  a.pos = st.pos;
  a.counts = st.counts;
  a.vbits_dwords = plan.vbits_dwords;
  a.ptile_dwords = plan.ptile_dwords;
  a.tile_u64 = plan.tile_u64;
  const uint64_t nt = a.n_wtiles;
  const uint64_t nb = (nt + SCAN_TILE - 1) / SCAN_TILE;
  NTCHK(ensure_scratch(c, 2 * nt + nb + 16));
  a.tile_counts = c->d_scratch;
  uint64_t* d_off = c->d_scratch + nt;
  uint64_t* d_sums = c->d_scratch + 2 * nt;
  uint64_t* d_total = (uint64_t*)(c->d_small + 8);
  a.tile_off = d_off;
  if (st.counts) HIPCHK(hipMemsetAsync(st.counts, 0, rd->n_reads * sizeof(uint64_t), c->stream));
  {
    // count pass: 16 waves per block, validity bits only
    KmerRunsGenArgs ca = a;
    ca.waves = 16;
    while (ca.waves > 1 && (size_t)ca.waves * ca.vbits_dwords * 4 + 64 > 150 * 1024) ca.waves /= 2; // long k
    const size_t lds = (size_t)ca.waves * ca.vbits_dwords * 4 + 64;
    int per_cu = 1;
    auto count_kernel = packed ? kmer_runs_count_kernel<true> : kmer_runs_count_kernel<false>;
    NTCHK(blocks_per_cu(c, count_kernel, (int)ca.waves * 64, lds, &per_cu));
    const uint64_t need = (ca.n_wtiles + ca.waves - 1) / ca.waves;
    uint64_t grid = (uint64_t)c->n_cu * per_cu;
    if (grid > need) grid = need;
    hipLaunchKernelGGL(count_kernel, dim3((unsigned)grid), dim3(ca.waves * 64), lds, c->stream, ca);
    HIPCHK(hipGetLastError());
  }
  NTCHK(device_exclusive_scan(c, a.tile_counts, d_off, nt, d_sums, d_total));
  HIPCHK(hipMemcpyAsync(c->h_small + 8, d_total, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  memcpy(total, c->h_small + 8, 8);
  if (*total > capacity)
    return fail(NTHIP_ERR_CAPACITY, "output capacity %llu k-mers < %llu needed",
                (unsigned long long)capacity, (unsigned long long)*total);
  a.counts = nullptr;
  a.waves = plan.waves;
  if (packed) {
    NTCHK((launch_kmer_runs_gen_nw<true, SINK_NONE, true>(c, a, plan.lds, plan.g.nw, plan.g.dword_tail != 0)));
    HIPCHK(hipStreamSynchronize(c->stream));
    return NTHIP_OK; // (no strand outputs with packed input)
  }
  NTCHK(launch_kmer_runs_gen_nw<true>(c, a, plan.lds, plan.g.nw, plan.g.dword_tail != 0));
  // strand hashes (get_forward_hash / get_reverse_hash): the same pass again with another value selected,
  // one value per k-mer at the same compact offsets
  for (uint32_t sel = 1; sel <= 2; ++sel) {
    uint64_t* dst = sel == 1 ? st.fwd : st.rev;
    if (!dst) continue;
    KmerRunsGenArgs b = a;
    b.hashes = dst;
    b.pos = nullptr;
    b.m = 1;
    b.value_sel = sel;
    NTCHK(launch_kmer_runs_gen_nw<true>(c, b, plan.lds, plan.g.nw, plan.g.dword_tail != 0));
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  return NTHIP_OK;
}

int ntamd::host::run_kmer_na_special(nthip_ctx* c, const Staged& st, const nthip_reads* rd, uint32_t k, uint32_t m,
                                     const RunsPlan& plan, const KmerRunsArgs& ra0, bool dt, const KmerFixedArgs& consts,
                                     uint64_t capacity, uint64_t* total, bool* handled)
{
  *handled = false;
  if (m != 1 || plan.ph_tiles != (uint32_t)KR_BURST || st.pos || st.fwd || st.rev || c->tune.no_na_special) return NTHIP_OK;
  const uint32_t len = rd->fixed_len, stride = rd->stride ? rd->stride : len;
  NaPlan q; // the count pass and the listed tiles: the general kernel on the SAME tiles (run length, runs per read)
  if (!kmer_na_plan(c, len, stride, k, m, false, &q, 0, plan.C) || q.g.C != plan.C || q.g.rpr != plan.rpr ||
      q.g.rpr * q.g.C != len - k + 1)
    return NTHIP_OK;
  KmerRunsGenArgs a;
  fill_gen_args(a, c, st, rd, k, m, q.g, consts);
  NTCHK(get_kmer_tab(c, k, &a.init_tab));
  a.counts = st.counts;
  a.vbits_dwords = q.vbits_dwords;
  a.ptile_dwords = q.ptile_dwords;
  a.tile_u64 = q.tile_u64;
  const uint64_t nt = a.n_wtiles;
  if (nt != ra0.n_wtiles || plan.lds + (size_t)plan.waves * 128 > lds_cap_of(c) + 512) return NTHIP_OK;
  const uint64_t nb = (nt + SCAN_TILE - 1) / SCAN_TILE;
  NTCHK(ensure_scratch(c, 3 * nt + nb + 16 + (nt + 63) / 64 + 2)); // (+ a bit per tile: the count pass's flags)
  a.tile_counts = c->d_scratch;
  uint64_t* d_off = c->d_scratch + nt;
  uint64_t* d_list = c->d_scratch + 2 * nt;
  uint64_t* d_sums = c->d_scratch + 3 * nt;
  uint64_t* d_total = (uint64_t*)(c->d_small + 8);
  unsigned long long* d_nlist = (unsigned long long*)(c->d_small + 32);
  uint32_t* d_zero = (uint32_t*)(c->d_small + 40); // (the specialised kernel polls a "batch is dirty" word: it stays 0)
  a.tile_off = d_off;
  HIPCHK(hipMemsetAsync(c->d_small + 32, 0, 16, c->stream));
  if (st.counts) HIPCHK(hipMemsetAsync(st.counts, 0, rd->n_reads * sizeof(uint64_t), c->stream));
  {
    KmerRunsGenArgs ca = a;
    ca.waves = 16;
    while (ca.waves > 1 && (size_t)ca.waves * ca.vbits_dwords * 4 + 64 > 150 * 1024) ca.waves /= 2;
    const size_t lds = (size_t)ca.waves * ca.vbits_dwords * 4 + 64;
    int per_cu = 1;
    // tiles of whole reads lying back to back: flag the tiles with a non-base in one sweep over the batch's vectors, count
    // only those (kmer_runs_gen_kernel.hpp); the flags live behind the scan's sums in the scratch
    const bool flagged = stride == len && 64u % q.g.rpr == 0u && !c->tune.no_tiles_flag;
    if (flagged) {
      uint32_t* const d_flags = (uint32_t*)(c->d_scratch + 3 * nt + nb + 16);
      const size_t flag_words = (size_t)((nt + 31) / 32) + 1;
      HIPCHK(hipMemsetAsync(d_flags, 0, flag_words * 4, c->stream));
      hipLaunchKernelGGL(tiles_flag_kernel, dim3((unsigned)c->n_cu * 8), dim3(256), 0, c->stream, (const uint8_t*)ca.seqs, ca.total_bytes,
                         (uint32_t)(64u / q.g.rpr) * stride, d_flags);
      NTCHK(blocks_per_cu(c, tiles_count_flagged_kernel, (int)ca.waves * 64, lds, &per_cu));
      const uint64_t need = ((ca.n_wtiles + 63) / 64 + ca.waves - 1) / ca.waves;
      uint64_t grid = (uint64_t)c->n_cu * per_cu;
      if (grid > need) grid = need;
      hipLaunchKernelGGL(tiles_count_flagged_kernel, dim3((unsigned)grid), dim3(ca.waves * 64), lds, c->stream, ca, (const uint32_t*)d_flags);
    } else {
      NTCHK(blocks_per_cu(c, kmer_runs_count_kernel<false>, (int)ca.waves * 64, lds, &per_cu));
      const uint64_t need = (ca.n_wtiles + ca.waves - 1) / ca.waves;
      uint64_t grid = (uint64_t)c->n_cu * per_cu;
      if (grid > need) grid = need;
      hipLaunchKernelGGL(kmer_runs_count_kernel<false>, dim3((unsigned)grid), dim3(ca.waves * 64), lds, c->stream, ca);
    }
    HIPCHK(hipGetLastError());
  }
  NTCHK(device_exclusive_scan(c, a.tile_counts, d_off, nt, d_sums, d_total));
  hipLaunchKernelGGL(list_short_tiles_kernel, dim3((unsigned)(c->n_cu * 4)), dim3(256), 0, c->stream, a.tile_counts, nt, a.n_runs,
                     plan.C, d_list, d_nlist);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(c->h_small + 8, d_total, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(c->h_small + 32, d_nlist, 8, hipMemcpyDeviceToHost, c->stream));
  *handled = true;
  // Round 5: when the caller's capacity holds every window of the batch nothing can overflow, and the two passes below go out
  // WITHOUT the host waiting for the counts (the listed pass reads its number of tiles on the device): one wait per call
  // instead of two, the kernels back to back.  Else: wait, check, as before.
  const bool no_wait = capacity >= rd->n_reads * (uint64_t)(len - k + 1);
  uint64_t n_list = 0;
  if (!no_wait) {
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(total, c->h_small + 8, 8);
    memcpy(&n_list, c->h_small + 32, 8);
    if (*total > capacity)
      return fail(NTHIP_ERR_CAPACITY, "output capacity %llu k-mers < %llu needed",
                  (unsigned long long)capacity, (unsigned long long)*total);
  }
  // the tiles that lost nothing: the specialised kernel, at the compact offsets
  KmerRunsArgs ra = ra0;
  RunsPlan p2 = plan;
  ra.hashes = st.hashes;
  ra.dirty = d_zero;
  ra.vecmap = nullptr;
  ra.tile_off = d_off;
  ra.tile_counts = a.tile_counts;
  ra.tile_u64 = plan.tile_u64 + 16; // (a tile starts anywhere in a 128-byte line of the stream: built that far up)
  p2.tile_u64 = ra.tile_u64;
  p2.lds = plan.lds + (size_t)plan.waves * 128;
  NTCHK(launch_kmer_runs_special(c, ra, p2, dt));
  if (no_wait || n_list) { // the others (an N in one read of 1000: under 1 % of the tiles): the N-aware kernel
    a.counts = nullptr;
    a.waves = q.waves;
    a.tile_list = d_list;
    a.n_list = n_list;
    a.n_list_dev = no_wait ? d_nlist : nullptr;
    const bool prof = c->profiling; // (last_kernel_ms names and times the pass over the bulk of the batch)
    c->profiling = false;
    const int rc = launch_kmer_runs_gen_nw<true>(c, a, q.lds, q.g.nw, q.g.dword_tail != 0);
    c->profiling = prof;
    NTCHK(rc);
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  if (no_wait) memcpy(total, c->h_small + 8, 8);
  return NTHIP_OK;
}
