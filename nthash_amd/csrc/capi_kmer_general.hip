// capi_kmer_general.hip -- lane-per-read kernels (exact reference order on anything) and the row-per-read kernel
// Part of libnthash_hip.so (include/nthash_hip.h); see capi_internal.hpp for the file map.
#include "capi_internal.hpp"
#include "util_kernels.hpp"

using namespace ntamd;
using namespace ntamd::host;

namespace {

template <typename K>
int launch_kmer_fixed(nthip_ctx* c, K kernel, const KmerFixedArgs& a, size_t dyn_lds)
{
  int per_cu = 1;
  NTCHK(blocks_per_cu(c, kernel, KF_THREADS, dyn_lds, &per_cu));
  uint64_t grid = (uint64_t)c->n_cu * per_cu;
  if (grid > a.n_tiles) grid = a.n_tiles;
  prof_begin(c, "kmer_fixed_kernel");
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(KF_THREADS), dyn_lds, c->stream, a);
  prof_end(c);
  HIPCHK(hipGetLastError());
  return NTHIP_OK;
}

} // namespace

int ntamd::host::run_kmer_general(nthip_ctx* c, const Staged& st, const nthip_reads* rd, uint32_t k, uint32_t m,
                     uint64_t capacity, uint64_t* total)
{
  const uint64_t n = rd->n_reads;
  KmerGeneralArgs h;
  memset(&h, 0, sizeof h);
  h.seqs = st.seqs;
  h.offsets = st.offsets;
  h.n_reads = n;
  h.len = rd->fixed_len;
  h.stride = rd->stride ? rd->stride : rd->fixed_len;
  h.k = k;
  h.m = m;
  for (unsigned cde = 0; cde < 4; ++cde) {
    h.sk_fwd[cde] = srol_n(seed_of_code(cde), k);
    h.sk_rc[cde] = srol_n(seed_of_code(cde ^ 2u), k);
  }
  for (uint32_t i = 0; i < 256; ++i) h.mult[i] = multiplier(k, i);
  uint64_t* d_total = (uint64_t*)(c->d_small + 8);
  NTCHK(ensure_args(c, sizeof(KmerGeneralArgs)));
  const unsigned rblocks = (unsigned)((n + 255) / 256);

  // ---- work items: one per read, or one per 1024-window segment when a read is long ----
  // scratch layout: seg[n] | seg_base[n] | item_cnt[items] | item_off[items] | sums
  uint64_t n_items = n;
  bool segmented = false;
  const bool may_be_long = st.offsets || (rd->fixed_len >= k && rd->fixed_len - k + 1 > KG_SEG_WINDOWS);
  const uint64_t nb_r = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (may_be_long) {
    NTCHK(ensure_scratch(c, 2 * n + nb_r + 16));
    uint64_t* d_seg = c->d_scratch;
    uint64_t* d_seg_base = c->d_scratch + n;
    hipLaunchKernelGGL(kmer_seg_count_kernel, dim3(rblocks), dim3(256), 0, c->stream, st.offsets, n,
                       rd->fixed_len, k, d_seg);
    HIPCHK(hipGetLastError());
    NTCHK(device_exclusive_scan(c, d_seg, d_seg_base, n, c->d_scratch + 2 * n, d_total));
    HIPCHK(hipMemcpyAsync(c->h_small + 8, d_total, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    uint64_t items = 0;
    memcpy(&items, c->h_small + 8, 8);
    if (items > n) { // at least one read needs more than one segment
      segmented = true;
      n_items = items;
    }
  }
  const uint64_t nb_i = (n_items + SCAN_TILE - 1) / SCAN_TILE;
  if (segmented) {
    // re-create the segment table at the front of a scratch area that also holds the item arrays
    // (ensure_scratch may reallocate, so redo the two cheap kernels after growing it)
    NTCHK(ensure_scratch(c, 2 * n + 2 * n_items + nb_i + nb_r + 32));
    uint64_t* d_seg = c->d_scratch;
    uint64_t* d_seg_base = c->d_scratch + n;
    hipLaunchKernelGGL(kmer_seg_count_kernel, dim3(rblocks), dim3(256), 0, c->stream, st.offsets, n,
                       rd->fixed_len, k, d_seg);
    NTCHK(device_exclusive_scan(c, d_seg, d_seg_base, n, c->d_scratch + 2 * n + 2 * n_items, d_total));
    h.seg_base = d_seg_base;
    h.n_items = n_items;
  } else {
    NTCHK(ensure_scratch(c, 2 * n + 2 * n_items + nb_i + 32));
  }
  uint64_t* d_cnt = (!segmented && st.counts) ? st.counts : c->d_scratch + 2 * n;
  uint64_t* d_off = c->d_scratch + 2 * n + n_items;
  uint64_t* d_sums = c->d_scratch + 2 * n + 2 * n_items;
  const unsigned iblocks = (unsigned)((n_items + 255) / 256);

  // pass 1: per-item counts
  h.counts = d_cnt;
  HIPCHK(hipMemcpyAsync(c->d_args, &h, sizeof h, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(kmer_general_kernel<true>, dim3(iblocks), dim3(256), 0, c->stream,
                     (const KmerGeneralArgs*)c->d_args);
  HIPCHK(hipGetLastError());
  NTCHK(device_exclusive_scan(c, d_cnt, d_off, n_items, d_sums, d_total));
  HIPCHK(hipMemcpyAsync(c->h_small + 8, d_total, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream)); // also makes reuse of the stack copy `h` safe
  memcpy(total, c->h_small + 8, 8);
  if (segmented && st.counts) {
    hipLaunchKernelGGL(kmer_seg_read_counts_kernel, dim3(rblocks), dim3(256), 0, c->stream, h.seg_base, d_off, n,
                       n_items, *total, st.counts);
    HIPCHK(hipGetLastError());
  }
  if (*total > capacity)
    return fail(NTHIP_ERR_CAPACITY, "output capacity %llu k-mers < %llu needed",
                (unsigned long long)capacity, (unsigned long long)*total);
  // pass 2: hashes at their compact offsets
  h.counts = nullptr;
  h.item_off = d_off;
  h.hashes = st.hashes;
  h.pos = st.pos;
  h.fwd = st.fwd;
  h.rev = st.rev;
  h.capacity = capacity;
  HIPCHK(hipMemcpyAsync(c->d_args, &h, sizeof h, hipMemcpyHostToDevice, c->stream));
  prof_begin(c, "kmer_general_kernel");
  hipLaunchKernelGGL(kmer_general_kernel<false>, dim3(iblocks), dim3(256), 0, c->stream,
                     (const KmerGeneralArgs*)c->d_args);
  prof_end(c);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  return NTHIP_OK;
}

// row-per-read kernel (kmer_kernels.hpp): reads overlapping by more than k-1 bases, NTHIP_FORCE_ROWS
int ntamd::host::launch_kmer_rows(nthip_ctx* c, const KmerFixedArgs& a, size_t dyn)
{
  if (a.k == 31 && a.m == 1) return launch_kmer_fixed(c, kmer_fixed_kernel<31, 1>, a, dyn);
  if (a.k == 31 && a.m == 4) return launch_kmer_fixed(c, kmer_fixed_kernel<31, 4>, a, dyn);
  return launch_kmer_fixed(c, kmer_fixed_kernel<0, 0>, a, dyn);
}
