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
