// capi_sink_bloom.hip -- fused consumers of the hash stream: Bloom filter insert / query
// Part of libnthash_hip.so (include/nthash_hip.h); see capi_internal.hpp for the file map.
#include "capi_internal.hpp"

using namespace ntamd;
using namespace ntamd::host;

namespace {

uint64_t bloom_magic_of(uint64_t n_bits)
{
  return (n_bits & (n_bits - 1)) == 0 ? 0ull : ~0ull / n_bits;
}

// shared body: SINK_BLOOM_INSERT or SINK_BLOOM_QUERY over fixed-length reads
int run_kmer_bloom(nthip_ctx* c, const nthip_reads* rd, uint16_t k16, uint8_t m8, uint32_t* d_filter, uint64_t n_bits,
                   uint64_t* hits, uint64_t* total_out, uint64_t* total_hits, uint32_t flags, bool query)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  NTCHK(check_reads(rd));
  const uint32_t k = k16, m = m8;
  if (k == 0) return fail(NTHIP_ERR_ARG, "k must be greater than 0");
  if (k < 3) return fail(NTHIP_ERR_UNSUPPORTED, "k < 3 is undefined in the reference (src/kmer.cpp:47)");
  if (m == 0) return fail(NTHIP_ERR_UNSUPPORTED, "num_hashes must be >= 1");
  if (!d_filter || n_bits == 0) return fail(NTHIP_ERR_ARG, "filter is NULL / n_bits is 0");
  if ((uintptr_t)d_filter & 3u) return fail(NTHIP_ERR_ARG, "filter must be 4-byte aligned");
  if (rd->offsets) return fail(NTHIP_ERR_UNSUPPORTED, "fused consumers take fixed-length reads (offsets == NULL)");
  HIPCHK(hipSetDevice(c->device));
  if (total_out) *total_out = 0;
  if (total_hits) *total_hits = 0;
  const uint32_t len = rd->fixed_len, stride = rd->stride ? rd->stride : len;
  const bool host_hits = query && hits && (flags & NTHIP_HOST_OUTPUT);
  if (rd->n_reads == 0) return NTHIP_OK;
  if (len < k) {
    if (query && hits) {
      if (host_hits) memset(hits, 0, rd->n_reads * sizeof(uint64_t));
      else HIPCHK(hipMemsetAsync(hits, 0, rd->n_reads * sizeof(uint64_t), c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
    }
    return NTHIP_OK;
  }
  NaPlan plan;
  if (!kmer_na_plan(c, len, stride, k, m, /*want_pos (the k-mer's read)*/ query, &plan))
    return fail(NTHIP_ERR_UNSUPPORTED, "shape outside the fused consumer kernels (k <= 64, m <= 8, stride >= windows)");
  uint64_t total_bytes = 0;
  NTCHK(reads_total_bytes(c, rd, flags, &total_bytes));
  Staged st;
  NTCHK(stage_inputs(c, rd, flags, total_bytes, st));
  uint64_t* d_hits = hits;
  if (host_hits) {
    HIPCHK(hipMalloc((void**)&d_hits, rd->n_reads * sizeof(uint64_t)));
    st.owned.push_back(d_hits);
  }
  if (query && d_hits) HIPCHK(hipMemsetAsync(d_hits, 0, rd->n_reads * sizeof(uint64_t), c->stream));
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
  a.bloom = d_filter;
  a.n_bits = n_bits;
  a.bloom_magic = bloom_magic_of(n_bits);
  a.hits = query ? d_hits : nullptr;
  a.sink_totals = (uint64_t*)(c->d_small + 16);
  HIPCHK(hipMemsetAsync(c->d_small + 16, 0, 16, c->stream));
  if (query) NTCHK((launch_kmer_runs_gen_nw<true, SINK_BLOOM_QUERY>(c, a, plan.lds, plan.g.nw, plan.g.dword_tail != 0)));
  else NTCHK((launch_kmer_runs_gen_nw<true, SINK_BLOOM_INSERT>(c, a, plan.lds, plan.g.nw, plan.g.dword_tail != 0)));
  HIPCHK(hipMemcpyAsync(c->h_small + 16, c->d_small + 16, 16, hipMemcpyDeviceToHost, c->stream));
  if (host_hits) HIPCHK(hipMemcpyAsync(hits, d_hits, rd->n_reads * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  uint64_t tot[2];
  memcpy(tot, c->h_small + 16, 16);
  if (total_out) *total_out = tot[0];
  if (total_hits) *total_hits = tot[1];
  return NTHIP_OK;
}

} // namespace

extern "C" int nthip_kmer_bloom_insert(nthip_ctx* c, const nthip_reads* rd, uint16_t k, uint8_t m, uint8_t* d_filter,
                                       uint64_t n_bits, uint64_t* total, uint32_t flags)
{
  return run_kmer_bloom(c, rd, k, m, (uint32_t*)d_filter, n_bits, nullptr, total, nullptr, flags, false);
}

extern "C" int nthip_kmer_bloom_query(nthip_ctx* c, const nthip_reads* rd, uint16_t k, uint8_t m,
                                      const uint8_t* d_filter, uint64_t n_bits, uint64_t* hits, uint64_t* total,
                                      uint64_t* total_hits, uint32_t flags)
{
  return run_kmer_bloom(c, rd, k, m, (uint32_t*)d_filter, n_bits, hits, total, total_hits, flags, true);
}

extern "C" int nthip_stream_bloom_insert(nthip_ctx* c, const uint64_t* d_hashes, uint64_t n_values, uint8_t* d_filter,
                                         uint64_t n_bits)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  if (!d_filter || n_bits == 0) return fail(NTHIP_ERR_ARG, "filter is NULL / n_bits is 0");
  if ((uintptr_t)d_filter & 3u) return fail(NTHIP_ERR_ARG, "filter must be 4-byte aligned");
  if (n_values && !d_hashes) return fail(NTHIP_ERR_ARG, "hashes is NULL");
  HIPCHK(hipSetDevice(c->device));
  if (n_values == 0) return NTHIP_OK;
  prof_begin(c, "stream_bloom_insert_kernel");
  hipLaunchKernelGGL(stream_bloom_insert_kernel, dim3(c->n_cu * 8), dim3(256), 0, c->stream, d_hashes, n_values,
                     (uint32_t*)d_filter, n_bits, bloom_magic_of(n_bits));
  prof_end(c);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  return NTHIP_OK;
}
