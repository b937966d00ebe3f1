// capi_kmer_gen.hip -- kmer_runs_gen_kernel, dense: any fixed read shape
// Part of libnthash_hip.so (include/nthash_hip.h); see capi_internal.hpp for the file map.
#include "capi_internal.hpp"

using namespace ntamd;
using namespace ntamd::host;

int ntamd::host::launch_kmer_gen_dense(nthip_ctx* c, const KmerRunsGenArgs& ga, size_t lds, uint32_t nw, bool dt, bool packed)
{
  if (packed) return launch_kmer_runs_gen_nw<false, SINK_NONE, true>(c, ga, lds, nw, dt);
  if (ga.fh) { // (forward-half tables: the plan sized the LDS for them)
#define NT_FH(NWT) \
  (dt ? launch_kmer_runs_gen(c, kmer_runs_gen_kernel<NWT, true, false, SINK_NONE, false, true>, ga, lds, "kmer_runs_gen_kernel") \
      : launch_kmer_runs_gen(c, kmer_runs_gen_kernel<NWT, false, false, SINK_NONE, false, true>, ga, lds, "kmer_runs_gen_kernel"))
    switch (nw) {
      case 1: return NT_FH(1);
      case 2: return NT_FH(2);
      case 3: return NT_FH(3);
      case 4: return NT_FH(4);
      default: break; // (any k: no position tables)
    }
#undef NT_FH
  }
  return launch_kmer_runs_gen_nw<false>(c, ga, lds, nw, dt);
}
