// capi_sink_minimizer.hip -- per-read (w, k)-minimizers: nthip_kmer_minimizers
// Part of libnthash_hip.so (include/nthash_hip.h); see capi_internal.hpp for the file map.
#include "capi_internal.hpp"

#include <algorithm>

#include "minimizer_fused_kernel.hpp"
#include "minimizer_w_kernel.hpp"
#include "minimizer_kernels.hpp"
#include "util_kernels.hpp"

using namespace ntamd;
using namespace ntamd::host;

namespace {

// ---- the one-pass kernel (minimizer_fused_kernel.hpp): fixed-length reads lying back to back, k within the position
// tables, w <= the read's windows.  The run length C is the block of the sliding minimum: C <= w.
struct MzfPlan {
  uint32_t C = 0, rpr = 0, R = 0, m0 = 0, r = 0, extra = 0, waves = 0, bits_dwords = 0, pitch_h = 0, pitch_b = 0, per_wave_dwords = 0,
           nw = 0, stash_cap = 0, byte_dwords = 0;
  size_t lds = 0;
};
bool minimizer_fused_plan(const nthip_ctx* c, uint32_t len, uint32_t k, uint32_t w, MzfPlan* p)
{
  if (len < k || k > KMER_TABLE_K_MAX || w < 2 || w > 128) return false;
  const uint32_t nwin = len - k + 1;
  if (nwin < w) return false; // (a read with fewer windows than w is one window: the round-3 kernels)
  const uint32_t nw = kmer_nw(k);
  double best = 0;
  uint32_t best_c = 0;
  for (uint32_t C = 2; C <= 16 && C <= w; ++C) {
    const uint32_t rpr = (nwin + C - 1) / C, m0 = (w - 1) / C;
    if (rpr > 64 || m0 + 1 > MZF_MAX_MM) continue;
    if (c->tune.mz_c && C != c->tune.mz_c) continue;
    // lane-instructions per read: a first window, C - 1 rolls, the two sweeps; idle lanes of a tile of whole reads; an
    // even row pitch costs the tile writes bank conflicts, which the padding to an odd one removes
    const uint32_t R = 64 / rpr;
    const double per_block = 60.0 + 25.0 * (C - 1) + (m0 ? 38.0 : 30.0) * C + (m0 ? 12.0 * m0 : 0.0);
    const double cost = per_block * rpr * 64.0 / (R * rpr);
    if (best_c == 0 || cost < best) {
      best = cost;
      best_c = C;
    }
  }
  if (best_c == 0) return false;
  const uint32_t C = best_c;
  p->C = C;
  p->nw = nw;
  p->rpr = (nwin + C - 1) / C;
  p->R = 64 / p->rpr;
  p->m0 = (w - 1) / C;
  p->r = (w - 1) % C;
  p->extra = p->rpr * C - nwin;
  p->pitch_h = C | 1u;
  p->pitch_b = 4u * (((C + 3u) / 4u) | 1u);
  p->bits_dwords = ((15u + p->R * len + p->extra + 15u) >> 4) + nw + 8u;
  // the stash parks one tile's picks: three times the expected 2 / (w + 1) of the tile's windows, at least 256
  const uint32_t max_picks = p->R * (nwin - w + 1u);
  uint32_t scap = 6u * p->R * nwin / (w + 1u);
  scap = scap < 256u ? 256u : scap;
  scap = scap > max_picks ? max_picks : scap;
  p->stash_cap = (scap + 3u) & ~3u;
  // (the byte arrays before the stash: (MZF_ROWS + 64) * pitch_b = 129 * 4 * odd bytes -- padded to 8 below)
  p->byte_dwords = (((MZF_ROWS + 64u) * p->pitch_b + 7u) & ~7u) / 4u;
  p->per_wave_dwords = 2u * MZF_ROWS * p->pitch_h + 3u * MZF_FULL + p->byte_dwords + 2u * p->stash_cap + p->stash_cap / 2u + p->bits_dwords +
                       (p->bits_dwords + 8u + 1u) / 2u;
  p->per_wave_dwords = (p->per_wave_dwords + 3u) & ~3u;
  const size_t tables = (size_t)4 * nw * 256 * 16 + 256 + BR_CTRL_DWORDS * 4;
  const size_t cap = lds_cap_of(c);
  if (cap < tables + (size_t)p->per_wave_dwords * 4 * 2) return false;
  uint32_t waves = (uint32_t)((cap - tables) / ((size_t)p->per_wave_dwords * 4));
  if (waves > (uint32_t)MZF_MAX_THREADS / 64) waves = MZF_MAX_THREADS / 64;
  if (c->tune.mz_waves && c->tune.mz_waves < waves) waves = c->tune.mz_waves;
  p->waves = waves;
  p->lds = tables + (size_t)waves * p->per_wave_dwords * 4;
  return true;
}

// *handled = false (and nothing written that the caller relies on) when the shape is outside the kernel
int minimizers_fused(nthip_ctx* c, const uint8_t* d_seqs, uint64_t n, uint32_t len, uint32_t k, uint32_t w, uint64_t* d_min_hashes,
                     uint32_t* d_min_pos, uint64_t* d_min_offsets, uint64_t capacity, uint64_t* total_out, bool* handled)
{
  *handled = false;
  MzfPlan p;
  if (!minimizer_fused_plan(c, len, k, w, &p)) return NTHIP_OK;
  const uint64_t n_tiles = (n + p.R - 1) / p.R;
  if (n_tiles >= 0x7FFFFFFFull) return NTHIP_OK;
  // one block per CU, every block resident: the look-back needs the blocks of a round to run together (block_rounds.hpp)
  uint64_t grid = c->tune.mz_grid ? (uint64_t)c->tune.mz_grid : (uint64_t)c->n_cu;
  {
    const uint64_t need = (n_tiles + p.waves - 1) / p.waves;
    if (grid > need) grid = need;
  }
  const uint32_t n_rounds = (uint32_t)((n_tiles + grid * p.waves - 1) / (grid * p.waves));
  const uint64_t n_status = ((uint64_t)n_rounds + 1) * grid;
  NTCHK(ensure_scratch(c, n_status + 8));
  // scratch: [0] total, [1] abort (u32) | dirty (u32), [2 ...) the block-rounds' look-back words
  HIPCHK(hipMemsetAsync(c->d_scratch, 0, (n_status + 2) * sizeof(uint64_t), c->stream));
  MinimizerFusedArgs a;
  memset(&a, 0, sizeof a);
  a.seqs = d_seqs;
  NTCHK(get_kmer_tab(c, k, &a.init_tab));
  a.total = c->d_scratch;
  a.abort = (uint32_t*)(c->d_scratch + 1);
  a.timeout_us = c->tune.mz_timeout_us;
  a.dirty = a.abort + 1;
  a.status = (unsigned long long*)(c->d_scratch + 2);
  a.out_hashes = d_min_hashes;
  a.out_pos = d_min_pos;
  a.out_offsets = d_min_offsets;
  a.capacity = capacity;
  a.n_reads = n;
  a.total_bytes = n * len;
  a.n_tiles = (uint32_t)n_tiles;
  a.n_rounds = n_rounds;
  a.stash_cap = p.stash_cap;
  a.len = len;
  a.k = k;
  a.w = w;
  a.nwin = len - k + 1;
  a.nwv = a.nwin - w + 1;
  a.C = p.C;
  a.rpr = p.rpr;
  a.inv_rpr = 65536u / p.rpr + 1u;
  a.R = p.R;
  a.m0 = p.m0;
  a.r = p.r;
  a.extra = p.extra;
  a.waves = p.waves;
  a.bits_dwords = p.bits_dwords;
  a.pitch_h = p.pitch_h;
  a.pitch_b = p.pitch_b;
  a.per_wave_dwords = p.per_wave_dwords;
  KmerFixedArgs consts;
  memset(&consts, 0, sizeof consts);
  fill_kmer_consts(k, 1, consts);
  memcpy(a.tab, consts.tab, sizeof a.tab);
  bool launched = false;
  auto launch = [&](auto kernel) -> int {
    int per_cu = 1;
    NTCHK(blocks_per_cu(c, kernel, (int)p.waves * 64, p.lds, &per_cu));
    (void)per_cu;
    prof_begin(c, "minimizer_fused_kernel");
    NTCHK(launch_resident(c, kernel, (unsigned)grid, p.waves * 64, p.lds, a, c->tune.mz_grid == 0, &launched));
    prof_end(c);
    return NTHIP_OK;
  };
  const bool mid = p.m0 != 0;
  switch (p.nw * 2 + (mid ? 1 : 0)) {
    case 2: NTCHK(launch(minimizer_fused_kernel<1, false>)); break;
    case 3: NTCHK(launch(minimizer_fused_kernel<1, true>)); break;
    case 4: NTCHK(launch(minimizer_fused_kernel<2, false>)); break;
    case 5: NTCHK(launch(minimizer_fused_kernel<2, true>)); break;
    case 6: NTCHK(launch(minimizer_fused_kernel<3, false>)); break;
    case 7: NTCHK(launch(minimizer_fused_kernel<3, true>)); break;
    case 8: NTCHK(launch(minimizer_fused_kernel<4, false>)); break;
    default: NTCHK(launch(minimizer_fused_kernel<4, true>)); break;
  }
  if (!launched) return NTHIP_OK; // (the device does not hold the grid right now: the caller goes on with the round-3 kernels)
  HIPCHK(hipMemcpyAsync(c->h_small + 32, c->d_scratch, 16, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  uint64_t total = 0;
  uint32_t td[2];
  memcpy(&total, c->h_small + 32, 8);
  memcpy(td, c->h_small + 40, 8);
  if (td[0] != 0) return NTHIP_OK; // (the grid was not resident: the caller goes on with the round-3 kernels)
  *handled = true;
  if (total_out) *total_out = total;
  if (total > capacity)
    return fail(NTHIP_ERR_CAPACITY, "output capacity %llu minimizers < %llu needed", (unsigned long long)capacity,
                (unsigned long long)total);
  return NTHIP_OK;
}

// ---- the record form (minimizer_w_kernel.hpp): run length = w (4 ... 16), k <= 32 ----
template <int C>
int launch_minimizer_w(nthip_ctx* c, const MinimizerWArgs& a0, size_t lds_fixed, uint64_t n_tiles, bool* launched)
{
  MinimizerWArgs a = a0;
  auto kernel = minimizer_w_kernel<C>;
  const size_t lds = lds_fixed + (size_t)a.waves * a.per_wave_dwords * 4;
  int per_cu = 1;
  NTCHK(blocks_per_cu(c, kernel, (int)a.waves * 64, lds, &per_cu));
  (void)per_cu;
  uint64_t grid = c->tune.mz_grid ? (uint64_t)c->tune.mz_grid : (uint64_t)c->n_cu; // one block per CU, every block resident: the look-back needs the blocks of a round to run together
  const uint64_t need = (n_tiles + a.waves - 1) / a.waves;
  if (grid > need) grid = need;
  a.n_rounds = (uint32_t)((n_tiles + grid * a.waves - 1) / (grid * a.waves));
  const uint64_t n_status = ((uint64_t)a.n_rounds + 1) * grid;
  NTCHK(ensure_scratch(c, n_status + 8));
  // scratch: [0] total, [1] abort (u32), [2 ...) the block-rounds' look-back words
  HIPCHK(hipMemsetAsync(c->d_scratch, 0, (n_status + 2) * sizeof(uint64_t), c->stream));
  a.total = c->d_scratch;
  a.abort = (uint32_t*)(c->d_scratch + 1);
  a.timeout_us = c->tune.mz_timeout_us;
  a.status = (unsigned long long*)(c->d_scratch + 2);
  prof_begin(c, "minimizer_w_kernel");
  NTCHK(launch_resident(c, kernel, (unsigned)grid, a.waves * 64, lds, a, c->tune.mz_grid == 0, launched));
  prof_end(c);
  return NTHIP_OK;
}

// *handled = false when the shape is outside the kernel (reads with non-bases are its business: a k-mer that holds one is
// no candidate)
int minimizers_w(nthip_ctx* c, const uint8_t* d_seqs, uint64_t n, uint32_t len, uint32_t k, uint32_t w, uint64_t* d_min_hashes,
                 uint32_t* d_min_pos, uint64_t* d_min_offsets, uint64_t capacity, uint64_t* total_out, bool* handled)
{
  *handled = false;
  if (len < k || k > 32 || k > KMER_TABLE_K_MAX || w < 4 || w > 16) return NTHIP_OK;
  if (c->tune.mz_c && c->tune.mz_c != w) return NTHIP_OK; // (a forced run length: the any-run-length form)
  const uint32_t nwin = len - k + 1;
  if (nwin < w) return NTHIP_OK;
  const uint32_t C = w, rpr = (nwin + C - 1) / C;
  if (rpr > 64) return NTHIP_OK;
  MinimizerWArgs a;
  memset(&a, 0, sizeof a);
  a.R = 64 / rpr;
  const uint64_t n_tiles = (n + a.R - 1) / a.R;
  if (n_tiles >= 0x7FFFFFFFull) return NTHIP_OK;
  a.seqs = d_seqs;
  NTCHK(get_kmer_tab(c, k, &a.init_tab));
  a.src_tabs = kmer_ntab(k);
  a.out_hashes = d_min_hashes;
  a.out_pos = d_min_pos;
  a.out_offsets = d_min_offsets;
  a.capacity = capacity;
  a.n_reads = n;
  a.total_bytes = n * len;
  a.n_tiles = (uint32_t)n_tiles;
  a.len = len;
  a.k = k;
  a.nwin = nwin;
  a.nwv = nwin - w + 1;
  a.rpr = rpr;
  a.inv_rpr = 65536u / rpr + 1u;
  a.extra = rpr * C - nwin;
  a.bits_dwords = ((15u + a.R * len + a.extra + 15u) >> 4) + 2u + 8u;
  // the stash parks one tile's picks: three times the expected 2 / (w + 1) of the tile's windows, at least 256; a tile with
  // more writes them itself, a round late
  const uint32_t max_picks = a.R * a.nwv;
  uint32_t cap = 6u * a.R * nwin / (w + 1u);
  cap = cap < 256u ? 256u : cap;
  cap = cap > max_picks ? max_picks : cap;
  a.stash_cap = (cap + 1u) & ~1u;
  a.per_wave_dwords = 2u * a.stash_cap + a.stash_cap / 2u + a.bits_dwords + (a.bits_dwords + 8u + 1u) / 2u;
  a.per_wave_dwords = (a.per_wave_dwords + 1u) & ~1u;
  const size_t fixed = (size_t)8 * 256 * 16 + 256 + MZW_CTRL_DWORDS * 4;
  const size_t cap_lds = lds_cap_of(c);
  if (cap_lds < fixed + (size_t)a.per_wave_dwords * 4 * 2) return NTHIP_OK;
  uint32_t waves = (uint32_t)((cap_lds - fixed) / ((size_t)a.per_wave_dwords * 4));
  if (waves > mzw_max_waves((int)C)) waves = mzw_max_waves((int)C);
  if (c->tune.mz_waves && c->tune.mz_waves < waves) waves = c->tune.mz_waves;
  a.waves = waves;
  KmerFixedArgs consts;
  memset(&consts, 0, sizeof consts);
  fill_kmer_consts(k, 1, consts);
  memcpy(a.tab, consts.tab, sizeof a.tab);
  bool launched = false;
  switch (C) {
#define MZW_CASE(CC) case CC: NTCHK(launch_minimizer_w<CC>(c, a, fixed, n_tiles, &launched)); break;
    MZW_CASE(4) MZW_CASE(5) MZW_CASE(6) MZW_CASE(7) MZW_CASE(8) MZW_CASE(9) MZW_CASE(10) MZW_CASE(11) MZW_CASE(12)
    MZW_CASE(13) MZW_CASE(14) MZW_CASE(15) MZW_CASE(16)
#undef MZW_CASE
    default: return NTHIP_OK;
  }
  if (!launched) return NTHIP_OK; // (the device does not hold the grid right now: the caller goes on with the round-3 kernels)
  HIPCHK(hipMemcpyAsync(c->h_small + 32, c->d_scratch, 16, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  uint64_t total = 0;
  uint32_t td[2];
  memcpy(&total, c->h_small + 32, 8);
  memcpy(td, c->h_small + 40, 8);
  if (td[0] != 0) return NTHIP_OK; // (the grid was not resident: the caller goes on with the round-3 kernels)
  *handled = true;
  if (total_out) *total_out = total;
  if (total > capacity)
    return fail(NTHIP_ERR_CAPACITY, "output capacity %llu minimizers < %llu needed", (unsigned long long)capacity,
                (unsigned long long)total);
  return NTHIP_OK;
}

// one round of reads of at most MZ_REG_POS windows through minimizer_reg_kernel + minimizer_gather_kernel
// d_slot_counts: the read-slots form (read r's k-mers at r * nwin); else d_roff; both NULL: every read emits every one of its nwin windows (positions = indices); d_lpre / d_ctot / d_coff: n_reads u64 each
int minimizers_reg_round(nthip_ctx* c, uint64_t* d_h, uint32_t* d_pos, const uint64_t* d_roff, const uint64_t* d_slot_counts, uint64_t n_kmers, const uint64_t* d_offsets, const uint64_t* d_ends,
                         uint32_t k, uint64_t nr, uint32_t nwin, uint32_t w, uint64_t* d_lpre, uint64_t* d_ctot, uint64_t* d_coff,
                         uint64_t* d_sums, uint64_t* d_tot, uint64_t base, uint64_t capacity, uint64_t* d_min_hashes,
                         uint32_t* d_min_pos, uint64_t* d_min_offsets, uint64_t* round_total)
{
  const unsigned grid = (unsigned)(c->n_cu * 8);
  const uint64_t waves = (uint64_t)grid * 4;
  uint64_t rb = nr / (waves * 4);
  rb = rb < 1 ? 1 : rb > 256 ? 256 : rb;
  const uint64_t n_chunks = (nr + rb - 1) / rb;
  MinimizerDenseArgs da;
  memset(&da, 0, sizeof da);
  da.hashes = d_h;
  da.tpos = d_pos;
  da.n_reads = nr;
  da.nwin = nwin;
  da.w = w;
  da.rb = (uint32_t)rb;
  da.roff = d_roff;
  da.counts = d_slot_counts;
  da.n_kmers = n_kmers;
  da.offsets = d_offsets;
  da.ends = d_ends;
  da.k = k;
  da.lpre = d_lpre;
  da.ctot = d_ctot;
  da.coff = d_coff;
  da.base = base;
  da.capacity = capacity;
  da.out_hashes = d_min_hashes;
  da.out_pos = d_min_pos;
  da.out_offsets = d_min_offsets;
  prof_begin(c, "minimizer_reg_kernel");
  const bool sparse = d_roff || d_slot_counts, one = nwin <= 64; // (nwin: of the longest read)
  if (nwin > MZ_REG_POS) { // up to 256 windows: four register sets
    if (sparse) hipLaunchKernelGGL((minimizer_regn_kernel<true, 4>), dim3(grid), dim3(256), 0, c->stream, da);
    else hipLaunchKernelGGL((minimizer_regn_kernel<false, 4>), dim3(grid), dim3(256), 0, c->stream, da);
  } else
  if (sparse && one) hipLaunchKernelGGL((minimizer_reg_kernel<true, true>), dim3(grid), dim3(256), 0, c->stream, da);
  else if (sparse) hipLaunchKernelGGL((minimizer_reg_kernel<true, false>), dim3(grid), dim3(256), 0, c->stream, da);
  else if (one) hipLaunchKernelGGL((minimizer_reg_kernel<false, true>), dim3(grid), dim3(256), 0, c->stream, da);
  else hipLaunchKernelGGL((minimizer_reg_kernel<false, false>), dim3(grid), dim3(256), 0, c->stream, da);
  prof_end(c);
  NTCHK(device_exclusive_scan(c, d_ctot, d_coff, n_chunks, d_sums, d_tot));
  hipLaunchKernelGGL(minimizer_gather_kernel, dim3(grid), dim3(256), 0, c->stream, da);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(c->h_small + 8, d_tot, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  memcpy(round_total, c->h_small + 8, 8);
  return NTHIP_OK;
}

// reads of any lengths, device-resident -- read r = [d_starts[r], d_ends ? d_ends[r] : d_starts[r + 1]) of d_seqs --: ONE
// round, the emitted stream of the whole batch (at most one k-mer per byte of the buffer) in the context's scratch; the
// kernels take every read's window count from its length
int minimizers_of_spans(nthip_ctx* c, const uint8_t* d_seqs, uint64_t total_bytes, const uint64_t* d_starts, const uint64_t* d_ends,
                        uint64_t n, uint64_t max_len, uint16_t k16, uint32_t w, uint64_t* d_min_hashes, uint32_t* d_min_pos,
                        uint64_t* d_min_offsets, uint64_t capacity, uint64_t* total_out, uint64_t base = 0, bool last = true,
                        uint64_t round_bases = 0)
{
  // (base: minimizers of the rounds before -- this round's go behind them, its offsets count from there; last: write the
  // closing offset; round_bases: the bases of this round's reads when it is a piece of a batch -- what sizes the scratch)
  const uint32_t k = k16;
  const uint64_t max_nwin64 = max_len >= k ? max_len - k + 1 : 0;
  if (max_nwin64 > 0xFFFFFFFFull) return fail(NTHIP_ERR_UNSUPPORTED, "a read of %llu bases: window positions are 32 bits wide", (unsigned long long)max_len);
  const uint32_t max_nwin = (uint32_t)max_nwin64;
  if (max_nwin == 0) { // no read has a window
    HIPCHK(hipMemsetAsync(d_min_offsets, 0, (n + 1) * sizeof(uint64_t), c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return NTHIP_OK;
  }
  const uint64_t cap_kmers = round_bases ? round_bases : total_bytes; // (a k-mer starts at a base)
  const uint32_t chunks = (max_nwin + 63u) / 64u;
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  // the pick masks belong to the LDS-table kernels (reads of more than MZ_REGN_POS windows); with offsets (not spans) a read's
  // words start at (its first byte >> 6) + its index: total_bytes / 8 + 8 n bytes, whatever the longest read is
  const bool reg_path = max_nwin <= MZ_REGN_POS && !c->tune.mz_table;
  const bool by_start = !d_ends;
  const size_t b_fl = reg_path ? 0 : by_start ? al(((cap_kmers >> 6) + n + 2) * 8) : al(n * (size_t)chunks * 8);
  const size_t b_h = al(cap_kmers * 8), b_pos = al(cap_kmers * 4), b_rd = al(n * 8);
  const size_t b_sums = al((n / SCAN_TILE + cap_kmers / SCAN_TILE + 64) * 8);
  const size_t need = b_h + b_pos + b_fl + 4 * b_rd + b_sums + 256;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
  if (c->bloom_tmp_bytes < need) {
    if (need > free_b + c->bloom_tmp_bytes)
      return fail(NTHIP_ERR_UNSUPPORTED, "minimizers of reads given by offsets: the batch needs %llu MB of scratch in one round; split it",
                  (unsigned long long)(need >> 20));
    if (c->bloom_tmp) HIPCHK(hipFree(c->bloom_tmp));
    c->bloom_tmp = nullptr;
    c->bloom_tmp_bytes = 0;
    HIPCHK(hipMalloc((void**)&c->bloom_tmp, need));
    c->bloom_tmp_bytes = need;
  }
  uint8_t* p = c->bloom_tmp;
  uint64_t* d_h = (uint64_t*)p; p += b_h;
  uint32_t* d_pos = (uint32_t*)p; p += b_pos;
  uint64_t* d_masks = (uint64_t*)p; p += b_fl;
  uint64_t* d_counts = (uint64_t*)p; p += b_rd;
  uint64_t* d_roff = (uint64_t*)p; p += b_rd;
  uint64_t* d_picked = (uint64_t*)p; p += b_rd;
  uint64_t* d_ooff = (uint64_t*)p; p += b_rd;
  uint64_t* d_sums = (uint64_t*)p; p += b_sums;
  uint64_t* d_tot = (uint64_t*)p;
  nthip_out out;
  memset(&out, 0, sizeof out);
  out.hashes = d_h;
  out.capacity = cap_kmers;
  out.counts = d_counts;
  out.pos = d_pos;
  uint64_t n_kmers = 0;
  if (d_ends) {
    NTCHK(nthip_kmer_hash_spans(c, (const char*)d_seqs, total_bytes, d_starts, d_ends, n, k16, 1, &out, &n_kmers, 0));
  } else {
    nthip_reads dr;
    memset(&dr, 0, sizeof dr);
    dr.seqs = (const char*)d_seqs;
    dr.offsets = d_starts;
    dr.n_reads = n;
    NTCHK(nthip_kmer_hash(c, &dr, k16, 1, &out, &n_kmers, 0));
  }
  NTCHK(device_exclusive_scan(c, d_counts, d_roff, n, d_sums, d_tot));
  if (max_nwin <= MZ_REGN_POS && !c->tune.mz_table) { // short reads (a FASTQ batch): the tables in registers
    uint64_t total = 0;
    NTCHK(minimizers_reg_round(c, d_h, d_pos, d_roff, nullptr, n_kmers, d_starts, d_ends, k, n, max_nwin, w, d_picked, d_counts, d_ooff, d_sums,
                               d_tot + 1, base, capacity, d_min_hashes, d_min_pos, d_min_offsets, &total));
    total += base;
    if (last) HIPCHK(hipMemcpyAsync(d_min_offsets + n, &total, sizeof total, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (total_out) *total_out = total;
    if (total > capacity)
      return fail(NTHIP_ERR_CAPACITY, "output capacity %llu minimizers < %llu needed", (unsigned long long)capacity,
                  (unsigned long long)total);
    return NTHIP_OK;
  }
  MinimizerArgs a;
  memset(&a, 0, sizeof a);
  a.hashes = d_h;
  a.pos = d_pos;
  a.roff = d_roff;
  a.n_reads = n;
  a.n_kmers = n_kmers;
  a.nwin = max_nwin;
  a.w = w;
  a.offsets = d_starts;
  a.ends = d_ends;
  a.k = k;
  a.masks = d_masks;
  a.chunks = chunks;
  a.mask_by_start = by_start ? 1u : 0u;
  a.picked = d_picked;
  a.out_off = d_ooff;
  a.base = base;
  a.capacity = capacity;
  a.out_hashes = d_min_hashes;
  a.out_pos = d_min_pos;
  a.out_offsets = d_min_offsets;
  const unsigned grid = (unsigned)(c->n_cu * 16);
  if (max_nwin > 256) HIPCHK(hipMemsetAsync(d_masks, 0, b_fl, c->stream)); // (reads that take the walk OR bits in)
  prof_begin(c, "minimizer_flag_kernel");
  if (max_nwin <= 256) hipLaunchKernelGGL((minimizer_flag_kernel<false, 256>), dim3(grid * 2), dim3(64 * MZ_WAVES), 0, c->stream, a);
  else hipLaunchKernelGGL((minimizer_flag_kernel<false>), dim3(grid), dim3(64 * MZ_WAVES), 0, c->stream, a);
  prof_end(c);
  NTCHK(device_exclusive_scan(c, d_picked, d_ooff, n, d_sums, d_tot + 1));
  hipLaunchKernelGGL(minimizer_write_kernel<false>, dim3(grid), dim3(256), 0, c->stream, a);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(c->h_small + 8, d_tot + 1, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  uint64_t total = 0;
  memcpy(&total, c->h_small + 8, 8);
  total += base;
  if (last) HIPCHK(hipMemcpyAsync(d_min_offsets + n, &total, sizeof total, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (total_out) *total_out = total;
  if (total > capacity)
    return fail(NTHIP_ERR_CAPACITY, "output capacity %llu minimizers < %llu needed", (unsigned long long)capacity,
                (unsigned long long)total);
  return NTHIP_OK;
}

// reads given by offsets (host or device): staged, surveyed, then as spans whose ends are the next starts
int minimizers_of_offsets(nthip_ctx* c, const nthip_reads* rd, uint16_t k16, uint32_t w, uint64_t* d_min_hashes, uint32_t* d_min_pos,
                          uint64_t* d_min_offsets, uint64_t capacity, uint64_t* total_out, uint32_t flags)
{
  const uint64_t n = rd->n_reads;
  if (n == 0) {
    HIPCHK(hipMemsetAsync(d_min_offsets, 0, sizeof(uint64_t), c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return NTHIP_OK;
  }
  uint64_t total_bytes = 0;
  NTCHK(reads_total_bytes(c, rd, flags, &total_bytes));
  Staged st;
  NTCHK(stage_inputs(c, rd, flags & NTHIP_HOST_INPUT, total_bytes, st));
  OffsetsSurvey sv;
  NTCHK(offsets_survey_device(c, st.offsets, n, total_bytes, &sv));
  if (sv.bad) return fail(NTHIP_ERR_ARG, "offsets are not non-decreasing or reach outside the read buffer");
  if (sv.uniform && sv.len0 >= k16 && sv.len0 <= 0xFFFFFFFFull && c->tune.mz_fused == 0) {
    // offsets that are one length in disguise (untrimmed reads lying back to back): the one-pass kernel of fixed-length reads
    bool handled = false;
    const int rc = minimizers_w(c, st.seqs + sv.off0, n, (uint32_t)sv.len0, k16, w, d_min_hashes, d_min_pos, d_min_offsets, capacity, total_out,
                                &handled);
    if (rc != NTHIP_OK || handled) return rc;
  }
  // rounds of reads whose emitted stream (hash 8 + position 4 bytes per base at most, 40 bytes per read) fits the device
  // (round 4: one round until then, NTHIP_ERR_UNSUPPORTED beyond it)
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)4 << 30;
  free_b += c->bloom_tmp_bytes;
  uint64_t round_bases = (uint64_t)(free_b / 10 * 8) / 14;
  if (c->tune.bloom_round) round_bases = c->tune.bloom_round; // (tests: several rounds on a small batch)
  if (total_bytes - sv.off0 <= round_bases && n * 40 <= free_b / 10)
    return minimizers_of_spans(c, st.seqs, total_bytes, st.offsets, nullptr, n, sv.max_len, k16, w, d_min_hashes, d_min_pos, d_min_offsets,
                               capacity, total_out);
  std::vector<uint64_t> ho(n + 1);
  if (flags & NTHIP_HOST_INPUT) memcpy(ho.data(), rd->offsets, (n + 1) * 8);
  else HIPCHK(hipMemcpy(ho.data(), st.offsets, (n + 1) * 8, hipMemcpyDeviceToHost));
  const uint64_t reads_max = std::max<uint64_t>(1, (free_b / 10) / 40);
  uint64_t base = 0;
  bool overflow = false;
  for (uint64_t r0 = 0; r0 < n;) {
    uint64_t r1 = (uint64_t)(std::upper_bound(ho.begin() + r0, ho.end(), ho[r0] + round_bases) - ho.begin()) - 1;
    if (r1 <= r0) r1 = r0 + 1; // (a read longer than a round: alone)
    if (r1 - r0 > reads_max) r1 = r0 + reads_max;
    if (r1 > n) r1 = n;
    uint64_t tot = 0;
    const int rc = minimizers_of_spans(c, st.seqs, ho[r1], st.offsets + r0, nullptr, r1 - r0, sv.max_len, k16, w, d_min_hashes, d_min_pos,
                                       d_min_offsets + r0, capacity, &tot, base, r1 == n, ho[r1] - ho[r0]);
    if (rc == NTHIP_ERR_CAPACITY) overflow = true;
    else if (rc != NTHIP_OK) return rc;
    base = tot;
    r0 = r1;
  }
  if (total_out) *total_out = base;
  if (overflow)
    return fail(NTHIP_ERR_CAPACITY, "output capacity %llu minimizers < %llu needed", (unsigned long long)capacity, (unsigned long long)base);
  return NTHIP_OK;
}

} // namespace

extern "C" int nthip_kmer_minimizers(nthip_ctx* c, const nthip_reads* rd, uint16_t k16, uint32_t w, uint64_t* d_min_hashes,
                                     uint32_t* d_min_pos, uint64_t* d_min_offsets, uint64_t capacity, uint64_t* total_out,
                                     uint32_t flags)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  NTCHK(check_reads(rd));
  const uint32_t k = k16;
  if (k == 0) return fail(NTHIP_ERR_ARG, "k must be greater than 0");
  if (k < 3) return fail(NTHIP_ERR_UNSUPPORTED, "k < 3 is undefined in the reference (src/kmer.cpp:47)");
  if (w == 0) return fail(NTHIP_ERR_ARG, "w must be greater than 0");
  if (c->async_pending) return fail(NTHIP_ERR_ARG, "NTHIP_ASYNC batches are pending: call nthip_ctx_take_dirty first");
  if (!d_min_offsets || (capacity && !d_min_hashes)) return fail(NTHIP_ERR_ARG, "min_offsets / min_hashes is NULL");
  HIPCHK(hipSetDevice(c->device));
  if (total_out) *total_out = 0;
  if (rd->offsets) return minimizers_of_offsets(c, rd, k16, w, d_min_hashes, d_min_pos, d_min_offsets, capacity, total_out, flags);
  const uint64_t n = rd->n_reads;
  const uint32_t len = rd->fixed_len, stride = rd->stride ? rd->stride : len;
  if (n == 0 || len < k) {
    HIPCHK(hipMemsetAsync(d_min_offsets, 0, (n + 1) * sizeof(uint64_t), c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return NTHIP_OK;
  }
  const uint32_t nwin = len - k + 1;
  if (!(flags & NTHIP_HOST_INPUT) && stride == len && c->tune.mz_fused != 2) {
    // one pass over the bases, nothing of the hash stream in HBM; a batch with a non-base comes back unhandled
    bool handled = false;
    if (c->tune.mz_fused != 1) { // the record form: run length = w (4 ... 16), k <= 32; reads with non-bases included
      const int rcw = minimizers_w(c, (const uint8_t*)rd->seqs, n, len, k, w, d_min_hashes, d_min_pos, d_min_offsets, capacity, total_out,
                                   &handled);
      if (rcw != NTHIP_OK || handled) return rcw;
    }
    // the any-run-length form (w up to 7 x 16, k within the position tables); a batch with a non-base comes back unhandled
    // (NTHIP_TUNE_MZ_FUSED=1: this form on every shape it takes -- the cross-check of the record form)
    const int rc = minimizers_fused(c, (const uint8_t*)rd->seqs, n, len, k, w, d_min_hashes, d_min_pos, d_min_offsets, capacity, total_out,
                                    &handled);
    if (rc != NTHIP_OK || handled) return rc;
  }
  // rounds of reads: the emitted stream of a round (hash 8, position 4 -- only written for a round that has a read with a
  // non-base --, flag 1 byte per k-mer; three 8-byte values per read) in the context's consumer scratch
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)4 << 30;
  free_b += c->bloom_tmp_bytes;
  uint64_t round_kmers = (uint64_t)(free_b / 2) / 13;
  if (round_kmers > (1ull << 32)) round_kmers = 1ull << 32;
  if (c->tune.bloom_round) round_kmers = c->tune.bloom_round; // (tests: several rounds on a small batch)
  uint64_t reads_per_round = round_kmers / nwin;
  if (reads_per_round == 0) return fail(NTHIP_ERR_UNSUPPORTED, "reads too long for the minimizer rounds");
  if (reads_per_round > n) reads_per_round = n;
  const uint64_t cap_round = reads_per_round * nwin;
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const uint32_t chunks = (nwin + 63u) / 64u;
  const size_t b_h = al(cap_round * 8), b_pos = al(cap_round * 4), b_fl = al(reads_per_round * chunks * 8), b_rd = al(reads_per_round * 8);
  const size_t b_sums = al((reads_per_round / SCAN_TILE + 64) * 8);
  const size_t need = b_h + b_pos + b_fl + 4 * b_rd + b_sums + 256;
  if (c->bloom_tmp_bytes < need) {
    if (c->bloom_tmp) HIPCHK(hipFree(c->bloom_tmp));
    c->bloom_tmp = nullptr;
    c->bloom_tmp_bytes = 0;
    HIPCHK(hipMalloc((void**)&c->bloom_tmp, need));
    c->bloom_tmp_bytes = need;
  }
  uint8_t* p = c->bloom_tmp;
  uint64_t* d_h = (uint64_t*)p; p += b_h;
  uint32_t* d_pos = (uint32_t*)p; p += b_pos;
  uint64_t* d_masks = (uint64_t*)p; p += b_fl;
  uint64_t* d_counts = (uint64_t*)p; p += b_rd;
  uint64_t* d_roff = (uint64_t*)p; p += b_rd;
  uint64_t* d_picked = (uint64_t*)p; p += b_rd;
  uint64_t* d_ooff = (uint64_t*)p; p += b_rd;
  uint64_t* d_sums = (uint64_t*)p; p += b_sums;
  uint64_t* d_tot = (uint64_t*)p;
  uint64_t base = 0;
  bool overflow = false;
  for (uint64_t r0 = 0; r0 < n; r0 += reads_per_round) {
    const uint64_t nr = n - r0 < reads_per_round ? n - r0 : reads_per_round;
    nthip_reads part = *rd;
    part.seqs = rd->seqs + r0 * stride;
    part.n_reads = nr;
    nthip_out out;
    memset(&out, 0, sizeof out);
    out.hashes = d_h;
    out.capacity = nr * (uint64_t)nwin;
    uint64_t n_kmers = 0;
    bool dense = false, slots = false, settled = false;
    if (!(flags & NTHIP_HOST_INPUT) && nwin <= MZ_REGN_POS && !c->tune.mz_table) {
      // the dense pass alone, as if no read of the round had a non-base (NTHIP_ASYNC: nothing is redone), then the context's
      // flag; a round with a non-base: once more under the read-slots contract (the dense pass + the reads concerned redone
      // in their slots, with positions) -- not the count -> scan -> hash of the compact stream
      int rc = nthip_kmer_hash(c, &part, k16, 1, &out, &n_kmers, NTHIP_ASYNC);
      if (rc == NTHIP_OK) {
        int dirty = 0;
        NTCHK(nthip_ctx_take_dirty(c, &dirty));
        if (!dirty) {
          dense = settled = true;
          n_kmers = nr * (uint64_t)nwin;
        } else {
          out.counts = d_counts;
          out.pos = d_pos;
          c->pos_listed_only = true; // (a read that fills its slot has its k-mers at their window indices: the kernel knows)
          rc = nthip_kmer_hash(c, &part, k16, 1, &out, &n_kmers, NTHIP_OUT_READ_SLOTS);
          c->pos_listed_only = false;
          if (rc == NTHIP_OK) slots = settled = true;
          else if (rc != NTHIP_ERR_UNSUPPORTED) return rc;
          out.counts = nullptr;
          out.pos = nullptr;
        }
      } else if (rc != NTHIP_ERR_UNSUPPORTED) {
        return rc;
      }
    }
    if (!settled) {
      // optimistic: no read of the round has a non-base -- every read emits every window, positions are indices
      NTCHK(nthip_kmer_hash(c, &part, k16, 1, &out, &n_kmers, flags & NTHIP_HOST_INPUT));
      dense = n_kmers == nr * (uint64_t)nwin;
      if (!dense) { // (rare: again, with the positions and the per-read counts)
        out.counts = d_counts;
        out.pos = d_pos;
        NTCHK(nthip_kmer_hash(c, &part, k16, 1, &out, &n_kmers, flags & NTHIP_HOST_INPUT));
        NTCHK(device_exclusive_scan(c, d_counts, d_roff, nr, d_sums, d_tot));
      }
    }
    if (nwin <= MZ_REGN_POS && !c->tune.mz_table) {
      // short reads: the table in registers, the picks compacted in place chunk by chunk, then gathered
      uint64_t round_total = 0;
      // (read-slots form: the counts are an input, the chunks' totals go where the compact form's read offsets would be)
      NTCHK(minimizers_reg_round(c, d_h, d_pos, dense || slots ? nullptr : d_roff, slots ? d_counts : nullptr, n_kmers, nullptr, nullptr, k, nr, nwin,
                                 w, d_picked, slots ? d_roff : d_counts, d_ooff, d_sums, d_tot + 1, base, capacity, d_min_hashes,
                                 d_min_pos, d_min_offsets + r0, &round_total));
      if (base + round_total > capacity) overflow = true;
      base += round_total;
      continue;
    }
    MinimizerArgs a;
    memset(&a, 0, sizeof a);
    a.hashes = d_h;
    a.pos = dense ? nullptr : d_pos;
    a.roff = dense ? nullptr : d_roff;
    a.n_reads = nr;
    a.n_kmers = n_kmers;
    a.nwin = nwin;
    a.w = w;
    a.masks = d_masks;
    a.chunks = chunks;
    a.picked = d_picked;
    a.out_off = d_ooff;
    a.base = base;
    a.capacity = capacity;
    a.out_hashes = d_min_hashes;
    a.out_pos = d_min_pos;
    a.out_offsets = d_min_offsets + r0;
    const unsigned grid = (unsigned)(c->n_cu * 16);
    if (nwin > MZ_LDS_POS) HIPCHK(hipMemsetAsync(d_masks, 0, nr * (size_t)chunks * 8, c->stream)); // (the walks OR bits in)
    prof_begin(c, "minimizer_flag_kernel");
    if (nwin <= 256) {
      if (dense) hipLaunchKernelGGL((minimizer_flag_kernel<true, 256>), dim3(grid * 2), dim3(64 * MZ_WAVES), 0, c->stream, a);
      else hipLaunchKernelGGL((minimizer_flag_kernel<false, 256>), dim3(grid * 2), dim3(64 * MZ_WAVES), 0, c->stream, a);
    } else {
      if (dense) hipLaunchKernelGGL((minimizer_flag_kernel<true>), dim3(grid), dim3(64 * MZ_WAVES), 0, c->stream, a);
      else hipLaunchKernelGGL((minimizer_flag_kernel<false>), dim3(grid), dim3(64 * MZ_WAVES), 0, c->stream, a);
    }
    prof_end(c);
    NTCHK(device_exclusive_scan(c, d_picked, d_ooff, nr, d_sums, d_tot + 1));
    if (dense) hipLaunchKernelGGL(minimizer_write_kernel<true>, dim3(grid), dim3(256), 0, c->stream, a);
    else hipLaunchKernelGGL(minimizer_write_kernel<false>, dim3(grid), dim3(256), 0, c->stream, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->h_small + 8, d_tot + 1, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    uint64_t round_total = 0;
    memcpy(&round_total, c->h_small + 8, 8);
    if (base + round_total > capacity) overflow = true;
    base += round_total;
  }
  HIPCHK(hipMemcpyAsync(d_min_offsets + n, &base, sizeof base, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (total_out) *total_out = base;
  if (overflow)
    return fail(NTHIP_ERR_CAPACITY, "output capacity %llu minimizers < %llu needed", (unsigned long long)capacity,
                (unsigned long long)base);
  return NTHIP_OK;
}

extern "C" int nthip_kmer_minimizers_spans(nthip_ctx* c, const char* d_buf, uint64_t buf_bytes, const uint64_t* d_starts,
                                           const uint64_t* d_ends, uint64_t n_reads, uint16_t k, uint32_t w, uint64_t* d_min_hashes,
                                           uint32_t* d_min_pos, uint64_t* d_min_offsets, uint64_t capacity, uint64_t* total_out)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  if (k == 0) return fail(NTHIP_ERR_ARG, "k must be greater than 0");
  if (k < 3) return fail(NTHIP_ERR_UNSUPPORTED, "k < 3 is undefined in the reference (src/kmer.cpp:47)");
  if (w == 0) return fail(NTHIP_ERR_ARG, "w must be greater than 0");
  if (!d_min_offsets || (capacity && !d_min_hashes)) return fail(NTHIP_ERR_ARG, "min_offsets / min_hashes is NULL");
  if (n_reads && (!d_buf || !d_starts || !d_ends)) return fail(NTHIP_ERR_ARG, "buf / starts / ends is NULL");
  if (c->async_pending) return fail(NTHIP_ERR_ARG, "NTHIP_ASYNC batches are pending: call nthip_ctx_take_dirty first");
  HIPCHK(hipSetDevice(c->device));
  if (total_out) *total_out = 0;
  if (n_reads == 0) {
    HIPCHK(hipMemsetAsync(d_min_offsets, 0, sizeof(uint64_t), c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return NTHIP_OK;
  }
  uint64_t max_len = 0;
  NTCHK(check_offsets_device(c, d_starts, d_ends, n_reads, buf_bytes, false, &max_len));
  return minimizers_of_spans(c, (const uint8_t*)d_buf, buf_bytes, d_starts, d_ends, n_reads, max_len, k, w, d_min_hashes, d_min_pos,
                             d_min_offsets, capacity, total_out);
}
