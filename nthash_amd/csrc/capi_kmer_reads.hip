// capi_kmer_reads.hip -- variable-length SHORT reads (offsets / sorted spans): tiles of whole reads, clean reads on the
// run-split kernel, reads with a non-base on a lane-per-read kernel (kmer_reads_kernel.hpp).
// Part of libnthash_hip.so (include/nthash_hip.h); see capi_internal.hpp for the file map.
#include "capi_internal.hpp"
#include "kmer_reads_kernel.hpp"
#include "util_kernels.hpp"

#include <algorithm>

using namespace ntamd;
using namespace ntamd::host;

namespace {

template <int NW>
int launch_kmer_reads(nthip_ctx* c, int mode, const KmerReadsArgs& a, size_t dyn_lds)
{
  auto kernel = mode == RD_MODE_MARK    ? kmer_reads_kernel<RD_MODE_MARK, NW, false>
                : mode == RD_MODE_SLOTS ? (a.pos ? kmer_reads_kernel<RD_MODE_SLOTS, NW, true> : kmer_reads_kernel<RD_MODE_SLOTS, NW, false>)
                : a.pos                 ? kmer_reads_kernel<RD_MODE_HASH, NW, true>
                                        : kmer_reads_kernel<RD_MODE_HASH, NW, false>;
  int per_cu = 1;
  NTCHK(blocks_per_cu(c, kernel, (int)a.waves * 64, dyn_lds, &per_cu));
  const uint64_t need = (a.n_tiles + a.waves - 1) / a.waves;
  uint64_t grid = (uint64_t)c->n_cu * per_cu;
  if (grid > need) grid = need;
  if (mode != RD_MODE_MARK) prof_begin(c, mode == RD_MODE_SLOTS ? "kmer_reads_kernel(read slots)" : "kmer_reads_kernel");
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(a.waves * 64), dyn_lds, c->stream, a);
  if (mode != RD_MODE_MARK) prof_end(c);
  HIPCHK(hipGetLastError());
  return NTHIP_OK;
}

} // namespace

// *handled = false: the batch is outside this path (long reads, spans out of order, LDS); nothing was written
int ntamd::host::run_kmer_reads(nthip_ctx* c, const Staged& st, const uint64_t* d_starts, const uint64_t* d_ends,
                                uint64_t n_reads, uint64_t total_bytes, uint32_t k, uint32_t m, uint64_t capacity,
                                uint64_t* total, bool* handled, const ReadsShape* shape, bool slots)
{
  *handled = false;
  if (c->tune.no_kmer_reads || n_reads == 0) return NTHIP_OK;
  const uint64_t n = n_reads;
  // ---- shape of the batch: longest read, largest distance between starts, order ----
  unsigned long long* d_res = (unsigned long long*)(c->d_small + 96); // [0] max length [1] max pitch [2] out of order
  unsigned long long* d_ndirty = (unsigned long long*)(c->d_small + 128);
  HIPCHK(hipMemsetAsync(c->d_small + 96, 0, 56, c->stream)); // (+ the sum of the windows at 144)
  uint64_t res[5] = {0, 0, 0, 0, 0}; // max length, max pitch, out of order, sum of lengths, sum of windows
  // NTHIP_OUT_READ_SLOTS: the survey also sums the windows of every tile, for the usual tile of 32 reads (decided below:
  // a batch whose slabs would not fit gets fewer reads per tile and its sums from a pass of their own)
  const uint32_t R_guess = c->tune.reads_per_tile ? (c->tune.reads_per_tile > 64 ? 64 : c->tune.reads_per_tile) : 32;
  unsigned long long* d_guess = nullptr;
  if (shape) { // back-to-back reads already surveyed by the caller
    res[0] = shape->max_len;
    res[1] = shape->max_pitch;
    res[3] = shape->sum_len;
    res[4] = shape->sum_len; // (an upper bound: a window starts at a base)
  } else {
    uint64_t blocks = (n + READS_PREP_THREADS - 1) / READS_PREP_THREADS;
    if (blocks > (uint64_t)c->n_cu * 2) blocks = (uint64_t)c->n_cu * 2;
    if (slots) {
      const uint64_t nt_guess = (n + R_guess - 1) / R_guess;
      NTCHK(ensure_scratch2(c, nt_guess + 16));
      d_guess = (unsigned long long*)c->d_scratch2;
      HIPCHK(hipMemsetAsync(d_guess, 0, nt_guess * sizeof(uint64_t), c->stream));
    }
    hipLaunchKernelGGL(reads_prep_kernel, dim3((unsigned)blocks), dim3(READS_PREP_THREADS), 0, c->stream, d_starts, d_ends, n, total_bytes, d_res,
                       0u, R_guess, k, d_guess, (unsigned long long*)(c->d_small + 144));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->h_small + 96, d_res, 56, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(res, c->h_small + 96, 32);
    memcpy(&res[4], c->h_small + 144, 8);
  }
  const uint64_t max_len = res[0], max_pitch = res[1] > res[0] ? res[1] : res[0];
  if (res[2] || max_len > RD_MAX_LEN) return NTHIP_OK; // long reads: kmer_ragged_kernel spreads them over tiles
  if (max_len < k) { // no read has a window
    *handled = true;
    *total = 0;
    if (st.counts) HIPCHK(hipMemsetAsync(st.counts, 0, n * sizeof(uint64_t), c->stream));
    return NTHIP_OK;
  }
  // ---- tile geometry ----
  // Run length, odd (the tile writes of a wave -- 8-byte entries, C entries apart -- then fall on 16 different bank
  // pairs).  Mixed lengths: 9 (in-process A/B on 100-150 bp: 11 -3 %, 15 -6 %: a longer run wastes more of a read's
  // overlapping last run and costs waves).  Reads (nearly) all of one length -- a sequencer's output before or after a
  // light trim: what minimises the lane work of a full-length read, runs x (C + ~8 for the first window and the run's
  // set-up: 151 bp measured 15 > 9 > 11); 150 bp: 15 (+4 % against 9)
  uint32_t C = 9;
  if (res[3] >= (uint64_t)(0.95 * (double)max_len * (double)n) && max_len >= k) {
    const uint32_t W = (uint32_t)(max_len - k + 1);
    uint32_t best_cost = ~0u;
    for (uint32_t cand : {15u, 13u, 11u, 9u}) {
      const uint32_t cost = ((W + cand - 1) / cand) * (cand + 8);
      if (cost < best_cost) { best_cost = cost; C = cand; }
    }
  }
  if (c->tune.reads_run_len) C = c->tune.reads_run_len; // <= 16 (one word of roll steps + the first window)
  const uint32_t nw = kmer_nw(k);
  const uint64_t slab_cap = 16384; // bytes of reads (and what lies between them) a tile stages
  uint32_t R = c->tune.reads_per_tile ? c->tune.reads_per_tile : 32;
  if (R > 64) R = 64;
  while (R > 1 && (uint64_t)(R - 1) * max_pitch + max_len + 16 > slab_cap) --R;
  if ((uint64_t)(R - 1) * max_pitch + max_len + 16 > slab_cap) return NTHIP_OK;
  const uint32_t max_vec = (uint32_t)(((uint64_t)(R - 1) * max_pitch + max_len + 15 + 15) / 16 + 1);
  const uint32_t bits_dwords = (max_vec + nw + 8 + 3u) & ~3u;
  // (a listed read inside a pass leaves a hole; what does not fit the tile waits for the next pass)
  const uint32_t tile_u64 = (64 * C + RD_ALIGN_U64 + 48 + 1u) & ~1u;
  const uint32_t ptile_dwords = st.pos ? tile_u64 : 0;
  const uint32_t max_runs = R * (uint32_t)((max_len - k + 1 + C - 1) / C);
  const uint32_t rmap_dwords = ((max_runs + 64 + 3) / 4 + 3u) & ~3u;
  const size_t fixed = (size_t)kmer_ntab(k) * 4096 + 256 + 64;
  const size_t per_wave = (size_t)tile_u64 * 8 + ((size_t)ptile_dwords + bits_dwords + 256 + rmap_dwords) * 4;
  const size_t cap = lds_cap_of(c);
  uint32_t waves = 0;
  uint32_t w_top = c->tune.reads_waves ? c->tune.reads_waves : 16;
  if (slots && nw >= 4 && w_top > 12) w_top = 12; // (launch bounds of that instantiation: rd_max_threads)
  for (uint32_t w = w_top; w >= 1; --w)
    if (fixed + per_wave * w <= cap) { waves = w; break; }
  if (!waves) return NTHIP_OK;
  *handled = true;

  const uint64_t n_tiles = (n + R - 1) / R;
  const uint64_t nb = (n_tiles + SCAN_TILE - 1) / SCAN_TILE;
  NTCHK(ensure_scratch(c, 2 * n + 2 * n_tiles + nb + n / 8 + 64));
  uint64_t* d_cnt = st.counts ? st.counts : c->d_scratch;
  uint64_t* d_list = c->d_scratch + n;
  uint64_t* d_tsum = c->d_scratch + 2 * n;
  uint64_t* d_toff = d_tsum + n_tiles;
  uint64_t* d_sums = d_toff + n_tiles;
  uint8_t* d_flags = (uint8_t*)(d_sums + nb + 8);
  uint64_t* d_total = (uint64_t*)(c->d_small + 8);

  KmerReadsArgs a;
  memset(&a, 0, sizeof a);
  KmerFixedArgs consts;
  memset(&consts, 0, sizeof consts);
  fill_kmer_consts(k, m, consts);
  a.seqs = st.seqs;
  a.starts = d_starts;
  a.ends = d_ends;
  a.n_reads = n;
  a.R = R;
  a.n_tiles = n_tiles;
  a.cnt = d_cnt;
  a.flags = d_flags;
  a.dirty_list = d_list;
  a.dirty_count = d_ndirty;
  a.tile_sum = d_tsum;
  a.tile_off = d_toff;
  a.hashes = st.hashes;
  a.pos = st.pos;
  NTCHK(get_kmer_tab(c, k, &a.init_tab));
  a.k = k;
  a.m = m;
  a.C = C;
  a.ntab = kmer_ntab(k);
  a.waves = waves;
  a.bits_dwords = bits_dwords;
  a.tile_u64 = tile_u64;
  a.ptile_dwords = ptile_dwords;
  a.rmap_dwords = rmap_dwords;
  a.groups = c->tune.has_tile_map ? c->tune.tile_map : 32u; // (in process: 32 / 64 groups 0.2-0.7 % ahead of one range per block)
  memcpy(a.tab, consts.tab, sizeof a.tab);
  memcpy(a.mult, consts.mult, sizeof a.mult);
  const size_t lds = fixed + per_wave * waves;
  // NTHIP_OUT_READ_SLOTS: one pass over the bases -- slots from the spans alone, then hash; the reads the pass lists
  // (a non-base inside) are redone in their slots afterwards
  if (slots) {
    if (!st.counts || st.fwd || st.rev) return fail(NTHIP_ERR_ARG, "NTHIP_OUT_READ_SLOTS needs out->counts and has no strand outputs");
    const uint64_t* tile_windows = d_tsum;
    if (d_guess && R == R_guess) {
      tile_windows = (const uint64_t*)d_guess; // (the survey summed them)
    } else {
      const unsigned tblocks = (unsigned)std::min<uint64_t>((n_tiles + 255) / 256, (uint64_t)c->n_cu * 8);
      hipLaunchKernelGGL(reads_tile_windows_kernel, dim3(tblocks), dim3(256), 0, c->stream, d_starts, d_ends, n, R, k, n_tiles, d_tsum);
      HIPCHK(hipGetLastError());
    }
    NTCHK(device_exclusive_scan(c, tile_windows, d_toff, n_tiles, d_sums, d_total));
    HIPCHK(hipMemcpyAsync(c->h_small + 8, d_total, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(total, c->h_small + 8, 8); // the extent of the slot array: every window of every read
    if (*total > capacity)
      return fail(NTHIP_ERR_CAPACITY, "output capacity %llu k-mers < %llu slots needed", (unsigned long long)capacity,
                  (unsigned long long)*total);
    KmerReadsArgs sa = a;
    auto launch_slots = [&]() -> int {
      switch (nw) {
        case 0: return launch_kmer_reads<0>(c, RD_MODE_SLOTS, sa, lds);
        case 1: return launch_kmer_reads<1>(c, RD_MODE_SLOTS, sa, lds);
        case 2: return launch_kmer_reads<2>(c, RD_MODE_SLOTS, sa, lds);
        case 3: return launch_kmer_reads<3>(c, RD_MODE_SLOTS, sa, lds);
        default: return launch_kmer_reads<4>(c, RD_MODE_SLOTS, sa, lds);
      }
    };
    NTCHK(launch_slots());
    KmerDirtyReadsArgs da;
    memset(&da, 0, sizeof da);
    da.seqs = st.seqs;
    da.starts = d_starts;
    da.ends = d_ends;
    da.list = d_list;
    da.n_list = d_ndirty;
    da.k = k;
    da.m = m;
    da.cnt = d_cnt;
    da.tile_sum = d_tsum;
    da.tile_off = d_toff;
    da.R = R;
    da.hashes = st.hashes;
    da.pos = st.pos;
    da.slots = 1;
    NTCHK(get_fw_tab(c, &da.horner_tab));
    hipLaunchKernelGGL(kmer_dirty_reads_kernel<false>, dim3((unsigned)c->n_cu * 4), dim3(256), 0, c->stream, da);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return NTHIP_OK;
  }
  auto launch = [&](int mode, const KmerReadsArgs& args, size_t bytes) -> int {
    switch (nw) {
      case 0: return launch_kmer_reads<0>(c, mode, args, bytes); // any k
      case 1: return launch_kmer_reads<1>(c, mode, args, bytes);
      case 2: return launch_kmer_reads<2>(c, mode, args, bytes);
      case 3: return launch_kmer_reads<3>(c, mode, args, bytes);
      default: return launch_kmer_reads<4>(c, mode, args, bytes);
    }
  };
  // ---- mark: which reads have a non-base; counts of the others ----
  {
    KmerReadsArgs ma = a;
    ma.tile_u64 = 0;
    ma.ptile_dwords = 0;
    ma.rmap_dwords = 0;
    ma.bits_dwords = ((max_vec + 4) / 2 + 3u) & ~3u; // 16 validity bits per vector
    ma.waves = 16;
    const size_t mlds = ((size_t)ma.bits_dwords + 256) * 4 * ma.waves + 64;
    NTCHK(launch_kmer_reads<1>(c, RD_MODE_MARK, ma, mlds));
  }
  // ---- exact counts of the listed reads, offsets of all ----
  KmerDirtyReadsArgs da;
  memset(&da, 0, sizeof da);
  da.seqs = st.seqs;
  da.starts = d_starts;
  da.ends = d_ends;
  da.list = d_list;
  da.n_list = d_ndirty;
  da.k = k;
  da.m = m;
  da.cnt = d_cnt;
  da.tile_sum = d_tsum;
  da.tile_off = d_toff;
  da.R = R;
  da.hashes = st.hashes;
  da.pos = st.pos;
  da.fwd = st.fwd;
  da.rev = st.rev;
  for (uint32_t cde = 0; cde < 4; ++cde) {
    da.sk_fwd[cde] = srol_n(seed_of_code(cde), k);
    da.sk_rc[cde] = srol_n(seed_of_code(cde ^ 2u), k);
  }
  NTCHK(get_fw_tab(c, &da.horner_tab));
  const unsigned dblocks = (unsigned)c->n_cu * 4;
  hipLaunchKernelGGL(kmer_dirty_reads_kernel<true>, dim3(dblocks), dim3(256), 0, c->stream, da);
  HIPCHK(hipGetLastError());
  NTCHK(device_exclusive_scan(c, d_tsum, d_toff, n_tiles, d_sums, d_total));
  HIPCHK(hipMemcpyAsync(c->h_small + 8, d_total, 8, hipMemcpyDeviceToHost, c->stream));
  // (round 5) a capacity that holds every window of every read cannot overflow: the hash pass goes out behind the scan without
  // the host waiting for the count -- one wait fewer per call, the kernels back to back
  const bool no_wait = capacity >= res[4];
  if (!no_wait) {
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(total, c->h_small + 8, 8);
    if (*total > capacity)
      return fail(NTHIP_ERR_CAPACITY, "output capacity %llu k-mers < %llu needed", (unsigned long long)capacity,
                  (unsigned long long)*total);
  }
  // ---- hash: the clean reads, then the listed ones into the holes they left ----
  NTCHK(launch(RD_MODE_HASH, a, lds));
  for (uint32_t sel = 1; sel <= 2; ++sel) { // strand hashes: the hash pass again with another value selected
    uint64_t* dst = sel == 1 ? st.fwd : st.rev;
    if (!dst) continue;
    KmerReadsArgs sa = a;
    sa.hashes = dst;
    sa.pos = nullptr;
    sa.ptile_dwords = a.ptile_dwords; // (same LDS layout)
    sa.m = 1;
    sa.value_sel = sel;
    NTCHK(launch(RD_MODE_HASH, sa, lds));
  }
  hipLaunchKernelGGL(kmer_dirty_reads_kernel<false>, dim3(dblocks), dim3(256), 0, c->stream, da);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  if (no_wait) memcpy(total, c->h_small + 8, 8);
  return NTHIP_OK;
}

// ---- NTHIP_OUT_READ_SLOTS on fixed-length reads ------------------------------------------------------------------------
bool ntamd::host::kmer_fixed_slots_len_ok(uint32_t len) { return len <= RD_MAX_LEN; }

int ntamd::host::kmer_fixed_slots_begin(nthip_ctx* c, uint64_t n_reads, uint64_t total_bytes, FixedSlots* fs)
{
  const uint64_t n_vec = (total_bytes >> 4) + 4;           // vectors of the batch (from the reads' first byte rounded down to 16)
  const uint64_t vwords = (n_vec + 31) / 32, rwords = (n_reads + 31) / 32;
  const uint64_t elems = (vwords + rwords + 1) / 2 + n_reads + 8;
  NTCHK(ensure_scratch2(c, elems));
  fs->d_vecmap = (uint32_t*)c->d_scratch2;
  fs->d_readmap = fs->d_vecmap + vwords;
  fs->d_list = c->d_scratch2 + (vwords + rwords + 1) / 2;
  fs->d_count = (unsigned long long*)(fs->d_list + n_reads);
  fs->n_words = vwords;
  HIPCHK(hipMemsetAsync(fs->d_vecmap, 0, (vwords + rwords) * sizeof(uint32_t), c->stream));
  HIPCHK(hipMemsetAsync(fs->d_count, 0, sizeof(unsigned long long), c->stream));
  return NTHIP_OK;
}

// after the dense pass (counts = windows and, when wanted, positions = window indices already filled for every read):
// list the reads a marked vector touches, redo them -- the windows the reference emits at the front of the slot, the exact
// count, zeros behind
int ntamd::host::kmer_fixed_slots_finish(nthip_ctx* c, const Staged& st, const FixedSlots& fs, uint64_t n_reads, uint32_t len,
                                         uint32_t stride, uint32_t k, uint32_t m, uint64_t total_bytes, uint64_t* n_redone)
{
  uint64_t blocks = (fs.n_words + 255) / 256;
  if (blocks > (uint64_t)c->n_cu * 16) blocks = (uint64_t)c->n_cu * 16;
  hipLaunchKernelGGL(slots_list_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, (const uint32_t*)fs.d_vecmap, fs.n_words,
                     (uint32_t)((uintptr_t)st.seqs & 15u), stride, n_reads, total_bytes, fs.d_readmap, fs.d_list, fs.d_count);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(c->h_small + 128, fs.d_count, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  uint64_t n_list = 0;
  memcpy(&n_list, c->h_small + 128, 8);
  if (n_redone) *n_redone = n_list;
  if (n_list == 0) return NTHIP_OK;
  KmerDirtyReadsArgs da;
  memset(&da, 0, sizeof da);
  da.seqs = st.seqs;
  da.list = fs.d_list;
  da.n_list = fs.d_count;
  da.k = k;
  da.m = m;
  da.cnt = st.counts;
  da.hashes = st.hashes;
  da.pos = st.pos;
  da.slots = 1;
  da.R = 1;
  da.fixed_len = len;
  da.fixed_stride = stride;
  NTCHK(get_fw_tab(c, &da.horner_tab));
  // (not the kernel of record: nthip_last_kernel_ms keeps the dense pass)
  hipLaunchKernelGGL(kmer_dirty_reads_kernel<false>, dim3((unsigned)c->n_cu * 4), dim3(256), 0, c->stream, da);
  HIPCHK(hipGetLastError());
  return NTHIP_OK;
}

