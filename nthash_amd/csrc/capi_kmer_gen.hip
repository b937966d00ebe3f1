// capi_kmer_gen.hip -- kmer_runs_gen_kernel, dense: any fixed read shape
// Part of libnthash_hip.so (include/nthash_hip.h); see capi_internal.hpp for the file map.
#include "capi_internal.hpp"

using namespace ntamd;
using namespace ntamd::host;

int ntamd::host::launch_kmer_gen_dense(nthip_ctx* c, const KmerRunsGenArgs& ga, size_t lds, uint32_t nw, bool dt, bool packed)
{
  if (packed) return launch_kmer_runs_gen_nw<false, SINK_NONE, true>(c, ga, lds, nw, dt);
  return launch_kmer_runs_gen_nw<false>(c, ga, lds, nw, dt);
}
