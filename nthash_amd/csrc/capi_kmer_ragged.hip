// capi_kmer_ragged.hip -- variable-length reads (offsets / spans): kmer_ragged_kernel
// Part of libnthash_hip.so (include/nthash_hip.h); see capi_internal.hpp for the file map.
#include "capi_internal.hpp"
#include "kmer_ragged_kernel.hpp"
#include "util_kernels.hpp"

using namespace ntamd;
using namespace ntamd::host;

namespace {

// Variable-length reads: pre-pass (runs per read, list of reads with windows, tile table) ->
// count pass -> scan -> compact hash pass (kmer_ragged_kernel.hpp).  *handled = false when the
// shape is outside this path (k, m, LDS), the caller then uses the general kernel.
template <int NW>
int launch_kmer_ragged(nthip_ctx* c, int mode, const KmerRaggedArgs& a, size_t dyn_lds)
{
  auto kernel = mode == NA_MODE_COUNT ? kmer_ragged_kernel<NA_MODE_COUNT, NW> : kmer_ragged_kernel<NA_MODE_HASH, NW>;
  int per_cu = 1;
  NTCHK(blocks_per_cu(c, kernel, (int)a.waves * 64, dyn_lds, &per_cu));
  const uint64_t need = (a.n_wtiles + a.waves - 1) / a.waves;
  uint64_t grid = (uint64_t)c->n_cu * per_cu;
  if (grid > need) grid = need;
  if (mode == NA_MODE_HASH) prof_begin(c, "kmer_ragged_kernel");
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(a.waves * 64), dyn_lds, c->stream, a);
  if (mode == NA_MODE_HASH) prof_end(c);
  HIPCHK(hipGetLastError());
  return NTHIP_OK;
}

} // namespace

// reads = spans [starts[r], ends[r]) of the device buffer st.seqs (total_bytes long)
int ntamd::host::run_kmer_ragged(nthip_ctx* c, const Staged& st, const uint64_t* d_starts, const uint64_t* d_ends, uint64_t n_reads,
                    uint64_t total_bytes, uint32_t k, uint32_t m, uint64_t capacity, uint64_t* total, bool* handled,
                    const ReadsShape* shape, bool checked)
{
  *handled = false;
  // short reads in order (every offsets batch, the sequence lines of a FASTQ chunk): tiles of whole reads
  NTCHK(run_kmer_reads(c, st, d_starts, d_ends, n_reads, total_bytes, k, m, capacity, total, handled, shape));
  if (*handled) return NTHIP_OK;
  // (the short-read path validates the spans in its own survey pass; this one trusts them: one pass first, unless the caller did)
  if (!checked) NTCHK(check_offsets_device(c, d_starts, d_ends, n_reads, total_bytes, false));
  const uint32_t C = 15; // run length; the last run of a read may be shorter
  const uint64_t n = n_reads;
  const uint32_t nw = kmer_nw(k);
  // per-wave LDS: a tile touches <= 64 listed reads, each staging its runs' bytes rounded up to 16
  const uint32_t max_vec = (64 * C + 64 * (k - 1 + 15 + 15)) / 16 + 64;
  const uint32_t bits_dwords = (max_vec + nw + 8 + 3u) & ~3u;
  const uint32_t vbits_dwords = ((max_vec + 12) / 2 + 2 + 3u) & ~3u;
  const uint32_t tile_u64 = 64 * C + KRG_ALIGN_U64;
  const uint32_t ptile_dwords = st.pos ? 64 * C : 0;
  const size_t fixed = (size_t)kmer_ntab(k) * 4096 + 256 + 64;
  const size_t per_wave = (size_t)tile_u64 * 8 + ((size_t)ptile_dwords + bits_dwords + vbits_dwords + 2 * 512) * 4;
  const size_t cap = (c->lds_max < 160 * 1024 ? c->lds_max : 160 * 1024) - 512;
  uint32_t waves = 0;
  for (uint32_t w = 8; w >= 1; --w)
    if (fixed + per_wave * w <= cap) { waves = w; break; }
  if (!waves) return NTHIP_OK;
  *handled = true;

  // ---- pre-pass over reads ----------------------------------------------------------------
  const uint64_t nb_r = (n + SCAN_TILE - 1) / SCAN_TILE;
  NTCHK(ensure_scratch(c, 8 * n + nb_r + 16));
  uint64_t* d_rc = c->d_scratch;
  uint64_t* d_flag = c->d_scratch + n;      // flag, then (scanned) nz index
  uint64_t* d_nz_rc = c->d_scratch + 2 * n;
  NzMeta* d_nz_meta = (NzMeta*)(c->d_scratch + 4 * n); // 32-byte records (d_scratch comes from hipMalloc: aligned)
  uint64_t* d_run_base = c->d_scratch + 3 * n;
  uint64_t* d_sums = c->d_scratch + 8 * n;
  uint64_t* d_total = (uint64_t*)(c->d_small + 8);
  const unsigned rblocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(ragged_runs_kernel, dim3(rblocks), dim3(256), 0, c->stream, d_starts, d_ends, n, k, C, d_rc, d_flag);
  NTCHK(device_exclusive_scan(c, d_flag, d_flag, n, d_sums, d_total));
  HIPCHK(hipMemcpyAsync(c->h_small + 8, d_total, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  uint64_t n_nz = 0;
  memcpy(&n_nz, c->h_small + 8, 8);
  *total = 0;
  if (n_nz == 0) { // no read has a window
    if (st.counts) HIPCHK(hipMemsetAsync(st.counts, 0, n * sizeof(uint64_t), c->stream));
    return NTHIP_OK;
  }
  hipLaunchKernelGGL(ragged_scatter_kernel, dim3(rblocks), dim3(256), 0, c->stream, d_rc, d_flag, d_starts, d_ends, n,
                     d_nz_meta, d_nz_rc);
  NTCHK(device_exclusive_scan(c, d_nz_rc, d_run_base, n_nz, d_sums, d_total));
  HIPCHK(hipMemcpyAsync(c->h_small + 8, d_total, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  uint64_t total_runs = 0;
  memcpy(&total_runs, c->h_small + 8, 8);
  const uint64_t nt = (total_runs + 63) / 64;
  const uint64_t nb_t = (nt + SCAN_TILE - 1) / SCAN_TILE;
  NTCHK(ensure_scratch2(c, 4 * nt + nb_t + 16));
  uint64_t* d_tile_j0 = c->d_scratch2;
  uint64_t* d_tile_rem0 = c->d_scratch2 + nt;
  uint64_t* d_tile_cnt = c->d_scratch2 + 2 * nt;
  uint64_t* d_tile_off = c->d_scratch2 + 3 * nt;
  uint64_t* d_sums2 = c->d_scratch2 + 4 * nt;
  hipLaunchKernelGGL(ragged_tiles_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, c->stream, d_run_base,
                     n_nz, nt, d_tile_j0, d_tile_rem0);
  HIPCHK(hipGetLastError());

  KmerRaggedArgs a;
  memset(&a, 0, sizeof a);
  KmerFixedArgs consts;
  memset(&consts, 0, sizeof consts);
  fill_kmer_consts(k, m, consts);
  a.seqs = st.seqs;
  a.starts = d_starts;
  a.ends = d_ends;
  a.total_bytes = total_bytes;
  a.hashes = st.hashes;
  a.pos = st.pos;
  a.counts = st.counts;
  a.tile_counts = d_tile_cnt;
  a.tile_off = d_tile_off;
  a.nz_meta = d_nz_meta;
  a.tile_j0 = d_tile_j0;
  a.tile_rem0 = d_tile_rem0;
  NTCHK(get_kmer_tab(c, k, &a.init_tab));
  a.n_nz = n_nz;
  a.total_runs = total_runs;
  a.n_wtiles = nt;
  a.k = k;
  a.m = m;
  a.C = C;
  a.ntab = kmer_ntab(k);
  a.waves = waves;
  a.bits_dwords = bits_dwords;
  a.vbits_dwords = vbits_dwords;
  a.tile_u64 = tile_u64;
  a.ptile_dwords = ptile_dwords;
  memcpy(a.tab, consts.tab, sizeof a.tab);
  memcpy(a.mult, consts.mult, sizeof a.mult);
  const size_t lds = fixed + per_wave * waves;
  auto launch = [&](int mode) -> int {
    switch (nw) {
      case 0: return launch_kmer_ragged<0>(c, mode, a, lds); // any k
      case 1: return launch_kmer_ragged<1>(c, mode, a, lds);
      case 2: return launch_kmer_ragged<2>(c, mode, a, lds);
      case 3: return launch_kmer_ragged<3>(c, mode, a, lds);
      default: return launch_kmer_ragged<4>(c, mode, a, lds);
    }
  };
  if (st.counts) HIPCHK(hipMemsetAsync(st.counts, 0, n * sizeof(uint64_t), c->stream));
  {
    // count pass: validity bits and the read table only, 16 waves per block
    KmerRaggedArgs ca = a;
    ca.tile_u64 = 0;
    ca.ptile_dwords = 0;
    ca.bits_dwords = 0;
    ca.waves = 16;
    while (ca.waves > 1 && ((size_t)ca.vbits_dwords + 2 * 512) * 4 * ca.waves + 64 > cap) ca.waves /= 2; // long k
    const size_t clds = ((size_t)ca.vbits_dwords + 2 * 512) * 4 * ca.waves + 64;
    NTCHK(launch_kmer_ragged<1>(c, NA_MODE_COUNT, ca, clds));
  }
  NTCHK(device_exclusive_scan(c, d_tile_cnt, d_tile_off, nt, d_sums2, d_total));
  HIPCHK(hipMemcpyAsync(c->h_small + 8, d_total, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  memcpy(total, c->h_small + 8, 8);
  if (*total > capacity)
    return fail(NTHIP_ERR_CAPACITY, "output capacity %llu k-mers < %llu needed",
                (unsigned long long)capacity, (unsigned long long)*total);
  a.counts = nullptr;
  NTCHK(launch(NA_MODE_HASH));
  for (uint32_t sel = 1; sel <= 2; ++sel) { // strand hashes: the hash pass again with another value selected
    uint64_t* dst = sel == 1 ? st.fwd : st.rev;
    if (!dst) continue;
    a.hashes = dst;
    a.pos = nullptr;
    a.m = 1;
    a.value_sel = sel;
    NTCHK(launch(NA_MODE_HASH));
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  return NTHIP_OK;
}
