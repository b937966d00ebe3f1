// capi_kmer_runs.hip -- the k = 31 instantiations of kmer_runs_kernel (BASELINE configs 2, 3, 5)
// Part of libnthash_hip.so (include/nthash_hip.h); see capi_internal.hpp for the file map.
#include "capi_internal.hpp"

using namespace ntamd;
using namespace ntamd::host;

namespace {

template <typename K>
int launch_kmer_runs(nthip_ctx* c, K kernel, KmerRunsArgs a, size_t dyn_lds)
{
  int per_cu = 1;
  NTCHK(blocks_per_cu(c, kernel, (int)a.waves * 64, dyn_lds, &per_cu));
  const uint64_t need = (a.n_wtiles + a.waves - 1) / a.waves;
  uint64_t grid = (uint64_t)c->n_cu * per_cu;
  if (grid > need) grid = need;
  if (a.tile_map == 0xFFFFFFFFu) a.tile_map = (uint32_t)grid; // every block streams its own range
  prof_begin(c, "kmer_runs_kernel");
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(a.waves * 64), dyn_lds, c->stream, a);
  prof_end(c);
  HIPCHK(hipGetLastError());
  return NTHIP_OK;
}

} // namespace


// ra.m selects the instantiation: m = 1 (configs 2 / 5; run length 15 or 30), m = 4 compile-time (config 3),
// any other m at run time
int ntamd::host::launch_kmer_runs_special(nthip_ctx* c, const KmerRunsArgs& ra, const RunsPlan& plan, bool dt)
{
#define NT_RUNS(KT, MT, CT, NWT) \
  (dt ? launch_kmer_runs(c, kmer_runs_kernel<KT, MT, CT, NWT, true>, ra, plan.lds) \
      : launch_kmer_runs(c, kmer_runs_kernel<KT, MT, CT, NWT, false>, ra, plan.lds))
  if (ra.k != 31 || !(plan.C == 15 || (plan.C == 30 && ra.m == 1)))
    return fail(NTHIP_ERR_HIP, "no specialised run-split kernel for k=%u, run length %u", ra.k, plan.C);
  if (ra.m == 1 && plan.C == 15) return NT_RUNS(31, 1, 15, 2);
  if (ra.m == 1 && plan.C == 30) return NT_RUNS(31, 1, 30, 2);
  if (ra.m == 4 && !c->tune.no_m4) return NT_RUNS(31, 4, 15, 2); // BASELINE config 3
  return NT_RUNS(31, 0, 15, 2);
#undef NT_RUNS
}
