// capi_sink_minhash.hip -- fused consumer of the hash stream: per-read MinHash signatures
// Part of libnthash_hip.so (include/nthash_hip.h); see capi_internal.hpp for the file map.
#include "capi_internal.hpp"
#include "minimizer_kernels.hpp" // (stream_minhash_kernel)
#include "util_kernels.hpp"      // (SCAN_TILE)

using namespace ntamd;
using namespace ntamd::host;

namespace {

// per-read MinHash signatures: the k-mer hashes never leave the registers (kmer_runs_gen_kernel, SINK_MINHASH)
int run_kmer_minhash(nthip_ctx* c, const nthip_reads* rd, uint16_t k16, uint8_t m8, uint64_t* sig, uint64_t* total_out,
                     uint32_t flags)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  NTCHK(check_reads(rd));
  const uint32_t k = k16, m = m8;
  if (k == 0) return fail(NTHIP_ERR_ARG, "k must be greater than 0");
  if (k < 3) return fail(NTHIP_ERR_UNSUPPORTED, "k < 3 is undefined in the reference (src/kmer.cpp:47)");
  if (m == 0) return fail(NTHIP_ERR_UNSUPPORTED, "num_hashes must be >= 1");
  if (rd->n_reads && !sig) return fail(NTHIP_ERR_ARG, "signatures is NULL");
  HIPCHK(hipSetDevice(c->device));
  if (total_out) *total_out = 0;
  if (rd->n_reads == 0) return NTHIP_OK;
  if (rd->offsets) { // reads of any lengths: h[0] of the compact stream of a round of them, the signatures from the stream
    uint64_t sum_kmers = 0;
    const bool host = (flags & NTHIP_HOST_OUTPUT) != 0;
    NTCHK(offsets_in_rounds(c, rd, flags, 8, [&](const nthip_reads* part, uint64_t r0, uint64_t bases) -> int {
      Staged keep;
      const uint64_t nr = part->n_reads;
      uint64_t *d_h = nullptr, *d_counts = nullptr, *d_roff = nullptr, *d_sums = nullptr, n_kmers = 0;
      NTCHK(stream_of_offsets(c, part, k16, 1, flags, keep, &d_h, &d_counts, &n_kmers, bases ? bases : 1));
      sum_kmers += n_kmers;
      const size_t sb = nr * (size_t)m * sizeof(uint64_t);
      uint64_t* d_sig = sig + r0 * m;
      if (host) NTCHK(own_alloc(keep, sb, (void**)&d_sig));
      NTCHK(own_alloc(keep, (size_t)(nr + 1) * 8, (void**)&d_roff));
      NTCHK(own_alloc(keep, (size_t)(nr / SCAN_TILE + 64) * 8, (void**)&d_sums));
      NTCHK(device_exclusive_scan(c, d_counts, d_roff, nr, d_sums, (uint64_t*)(c->d_small + 16)));
      hipLaunchKernelGGL(stream_minhash_kernel, dim3((unsigned)(c->n_cu * 8)), dim3(256), 0, c->stream, d_h, d_roff, nr, n_kmers, m,
                         (uint64_t)k * MULTISEED, d_sig);
      HIPCHK(hipGetLastError());
      if (host) HIPCHK(hipMemcpyAsync(sig + r0 * m, d_sig, sb, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      return NTHIP_OK;
    }));
    if (total_out) *total_out = sum_kmers;
    return NTHIP_OK;
  }
  const uint32_t len = rd->fixed_len, stride = rd->stride ? rd->stride : len;
  const bool host_sig = (flags & NTHIP_HOST_OUTPUT) != 0;
  const size_t sig_bytes = rd->n_reads * (size_t)m * sizeof(uint64_t);
  if (len < k) { // no read has a k-mer
    if (host_sig) memset(sig, 0xFF, sig_bytes);
    else {
      HIPCHK(hipMemsetAsync(sig, 0xFF, sig_bytes, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
    }
    return NTHIP_OK;
  }
  const uint32_t per_launch = m < KRG_SIG_MAX ? m : KRG_SIG_MAX;
  NaPlan plan;
  if (!kmer_na_plan(c, len, stride, k, m, false, &plan, KRG_TILE_READS * per_launch))
    return fail(NTHIP_ERR_UNSUPPORTED, "shape outside the fused consumer kernels (stride >= windows)");
  uint64_t total_bytes = 0;
  NTCHK(reads_total_bytes(c, rd, flags, &total_bytes));
  Staged st;
  NTCHK(stage_inputs(c, rd, flags, total_bytes, st));
  uint64_t* d_sig = sig;
  if (host_sig) {
    HIPCHK(hipMalloc((void**)&d_sig, sig_bytes));
    st.owned.push_back(d_sig);
  }
  HIPCHK(hipMemsetAsync(d_sig, 0xFF, sig_bytes, c->stream));
  KmerFixedArgs consts;
  memset(&consts, 0, sizeof consts);
  fill_kmer_consts(k, m, consts);
  KmerRunsGenArgs a;
  fill_gen_args(a, c, st, rd, k, m, plan.g, consts);
  NTCHK(get_kmer_tab(c, k, &a.init_tab));
  a.hashes = nullptr;
  a.vbits_dwords = plan.vbits_dwords;
  a.ptile_dwords = plan.ptile_dwords;
  a.tile_u64 = plan.tile_u64;
  a.waves = plan.waves;
  a.sig = d_sig;
  a.sink_totals = (uint64_t*)(c->d_small + 16);
  HIPCHK(hipMemsetAsync(c->d_small + 16, 0, 16, c->stream));
  uint32_t launches = 0;
  for (uint32_t first = 0; first < m; first += KRG_SIG_MAX, ++launches) { // KRG_SIG_MAX entries per pass
    a.sig_first = first;
    a.sig_n = m - first < KRG_SIG_MAX ? m - first : KRG_SIG_MAX;
    if (m == 1) NTCHK((launch_kmer_runs_gen_nw<true, SINK_MINHASH1>(c, a, plan.lds, plan.g.nw, plan.g.dword_tail != 0)));
    else NTCHK((launch_kmer_runs_gen_nw<true, SINK_MINHASH>(c, a, plan.lds, plan.g.nw, plan.g.dword_tail != 0)));
  }
  HIPCHK(hipMemcpyAsync(c->h_small + 16, c->d_small + 16, 16, hipMemcpyDeviceToHost, c->stream));
  if (host_sig) HIPCHK(hipMemcpyAsync(sig, d_sig, sig_bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  uint64_t tot = 0;
  memcpy(&tot, c->h_small + 16, 8);
  if (total_out) *total_out = tot / launches; // every pass consumes every k-mer
  return NTHIP_OK;
}

} // namespace

extern "C" int nthip_kmer_minhash(nthip_ctx* c, const nthip_reads* rd, uint16_t k, uint8_t m, uint64_t* signatures,
                                  uint64_t* total, uint32_t flags)
{
  return run_kmer_minhash(c, rd, k, m, signatures, total, flags);
}
