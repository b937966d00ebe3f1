// capi_kmer_runs.hip -- the k = 31 instantiations of kmer_runs_kernel (BASELINE configs 2, 3, 5)
// Part of libnthash_hip.so (include/nthash_hip.h); see capi_internal.hpp for the file map.
#include "capi_internal.hpp"

using namespace ntamd;
using namespace ntamd::host;

namespace {

// EXPERIMENT (NTHIP_TUNE_PF_GBPS; off by default): one block per tile group reads the group's input range ahead of the
// hashing waves, a chunk at a time, at the pace the main kernel is expected to consume it -- large sequential reads
// instead of thousands of 9 KiB pieces, to see whether HBM / the Infinity Cache take the read share of the mix better
// that way.  Loads only; the data is dropped.
__global__ __launch_bounds__(256) void input_prefetch_kernel(const uint8_t* base, uint64_t total_bytes, uint64_t group_bytes,
                                                              uint32_t chunk_bytes, uint64_t lead_bytes, double ticks_per_byte,
                                                              uint32_t* sink)
{
  const uint64_t g0 = (uint64_t)blockIdx.x * group_bytes;
  if (g0 >= total_bytes) return;
  const uint64_t g1 = g0 + group_bytes < total_bytes ? g0 + group_bytes : total_bytes;
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime(); // 100 MHz
  uint32_t acc = 0;
  for (uint64_t off = g0; off < g1; off += chunk_bytes) {
    const uint64_t done = off - g0;
    const uint64_t due = done > lead_bytes ? t0 + (uint64_t)((double)(done - lead_bytes) * ticks_per_byte) : t0;
    while (__builtin_amdgcn_s_memrealtime() < due) __builtin_amdgcn_s_sleep(32);
    const uint64_t end = off + chunk_bytes < g1 ? off + chunk_bytes : g1;
    for (uint64_t p = (off & ~15ull) + (uint64_t)threadIdx.x * 16u; p + 16u <= end; p += 256u * 16u) {
      const uint4 v = *(const uint4*)(base + p);
      acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
  }
  if (acc == 0x9E3779B9u) sink[0] = acc; // (keeps the loads)
}

template <typename K>
int launch_kmer_runs(nthip_ctx* c, K kernel, KmerRunsArgs a, size_t dyn_lds)
{
  int per_cu = 1;
  NTCHK(blocks_per_cu(c, kernel, (int)a.waves * 64, dyn_lds, &per_cu));
  const uint64_t need = (a.n_wtiles + a.waves - 1) / a.waves;
  uint64_t grid = (uint64_t)c->n_cu * per_cu;
  if (grid > need) grid = need;
  // 32 groups of 8 blocks, each group streaming through its own range of tiles with its waves interleaved.  Round 1 had
  // one group per block; in-process over fresh allocations (tools/alloc_ab.py) 16-128 groups are 0-0.9 % faster on the
  // 150 bp shape (nothing on a box where the allocation is slow anyway), 2 % on 151 bp / run length 11; 1 group
  // (plain grid stride) is 4-5 % slower on a good allocation and 2-4 % faster on a bad one; m > 1: no difference
  if (a.tile_map == 0xFFFFFFFFu) a.tile_map = grid >= 64 ? 32u : (uint32_t)grid;
  if (a.ph_tiles && !c->tune.no_pacing) {
    // period of the phased path = the time HBM needs for what the chip reads and writes in one period, apart:
    // reads at ~6.0 TB/s, write-through stores at ~6.8 TB/s (measured with the hash switched off), in ticks of 10 ns
    const double kmers = (double)grid * a.waves * a.ph_tiles * 64.0 * a.C;
    const double t_r = kmers * a.len / a.nwin / 6.0e12, t_w = kmers * 8.0 * a.m / 6.8e12;
    a.ph_read = c->tune.ph_read ? c->tune.ph_read : (uint32_t)((t_r + 1.5e-6) * 1e8);
    a.ph_period = c->tune.ph_period ? c->tune.ph_period : (uint32_t)((t_r + t_w + 1.0e-6) * 1e8);
    if (a.ph_read >= a.ph_period) a.ph_read = a.ph_period / 4;
  }
#if KR_DEBUG_TIMES
  HIPCHK(hipMemsetAsync(c->d_small + 64, 0, 64, c->stream));
  HIPCHK(hipMemsetAsync(c->d_small + 64 + 48, 0xFF, 8, c->stream));
#endif
  bool prefetching = false;
  if (c->tune.pf_gbps && a.ph_tiles == (uint32_t)KR_BURST && a.tile_map && ((uintptr_t)a.seqs & 15u) == 0) {
    if (!c->aux_stream) {
      HIPCHK(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&c->aux_done, hipEventDisableTiming));
    }
    const uint64_t tile_bytes = (uint64_t)(64u / a.rpr) * a.stride;
    const uint64_t per = (a.n_wtiles + a.tile_map - 1) / a.tile_map; // tiles per group (tile_range)
    const uint64_t group_bytes = per * tile_bytes, total_bytes = a.n_reads * (uint64_t)a.stride;
    const uint32_t chunk = (c->tune.pf_chunk_kb ? c->tune.pf_chunk_kb : 256u) << 10;
    const uint64_t lead = (uint64_t)(c->tune.pf_lead_kb ? c->tune.pf_lead_kb : 2048u) << 10;
    // bytes per second of ONE group = rate / groups; ticks of 10 ns per byte
    const double ticks_per_byte = 1e8 * (double)a.tile_map / ((double)c->tune.pf_gbps * 1e9);
    hipLaunchKernelGGL(input_prefetch_kernel, dim3(a.tile_map), dim3(256), 0, c->aux_stream, (const uint8_t*)a.seqs, total_bytes,
                       group_bytes, chunk, lead, ticks_per_byte, (uint32_t*)(c->d_small + 200));
    prefetching = true;
  }
  prof_begin(c, "kmer_runs_kernel");
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(a.waves * 64), dyn_lds, c->stream, a);
  prof_end(c);
  HIPCHK(hipGetLastError());
  if (prefetching) { // (what follows on the context's stream follows the reader too)
    HIPCHK(hipEventRecord(c->aux_done, c->aux_stream));
    HIPCHK(hipStreamWaitEvent(c->stream, c->aux_done, 0));
  }
#if KR_DEBUG_TIMES
  {
    uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(t, c->d_small + 64, 64, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    const double nw = (double)grid * a.waves;
    fprintf(stderr, "[kmer_runs phased] per wave, us: store drain %.0f  wait read window %.0f  load+pack %.0f  wait write window %.0f  "
                    "hash+write %.0f  | wave elapsed min %.0f avg %.0f max %.0f  (period %u read %u ticks, P %u)\n",
            t[0] / nw / 100.0, t[1] / nw / 100.0, t[2] / nw / 100.0, t[3] / nw / 100.0, t[4] / nw / 100.0, t[6] / 100.0,
            t[7] / nw / 100.0, t[5] / 100.0, a.ph_period, a.ph_read, a.ph_tiles);
  }
#endif
  return NTHIP_OK;
}

} // namespace


bool ntamd::host::kmer_runs_chunked_compiled() { return KR_CHUNKED != 0; }

// ra.m selects the instantiation: m = 1 (configs 2 / 5; run length 15 or 30), m = 4 compile-time (config 3),
// any other m at run time
// shapes launch_kmer_runs_special has a runtime-k instantiation for
bool ntamd::host::kmer_runs_any_k_compiled(uint32_t k, uint32_t m, uint32_t C)
{
  if (m != 1 || k < 17 || k > 32) return false;
  switch (C) {
    case 9: case 10: case 11: case 12: case 13: case 14: case 15: case 16: case 17: case 19: case 20: case 21: case 22: case 23:
    case 25: return true;
    default: return false;
  }
}

int ntamd::host::launch_kmer_runs_special(nthip_ctx* c, const KmerRunsArgs& ra, const RunsPlan& plan, bool dt)
{
#define NT_RUNS(KT, MT, CT, NWT) \
  (dt ? launch_kmer_runs(c, kmer_runs_kernel<KT, MT, CT, NWT, true>, ra, plan.lds) \
      : launch_kmer_runs(c, kmer_runs_kernel<KT, MT, CT, NWT, false>, ra, plan.lds))
  // m = 1, 17 <= k <= 32, a run length that divides the window count: k at run time, run length compile-time
  // (151 bp / k31: 11, 100 bp: 14, 76 bp: 23, 250 bp: 11, 125 bp: 19, 150 bp / k21: 13, ...)
  if (ra.m == 1 && !(ra.k == 31 && (plan.C == 15 || plan.C == 30)) && kmer_runs_any_k_compiled(ra.k, 1, plan.C)) {
    switch (plan.C) {
      case 9: return NT_RUNS(0, 1, 9, 2);
      case 17: return NT_RUNS(0, 1, 17, 2);
      case 21: return NT_RUNS(0, 1, 21, 2);
      case 10: return NT_RUNS(0, 1, 10, 2);
      case 11: return NT_RUNS(0, 1, 11, 2);
      case 12: return NT_RUNS(0, 1, 12, 2);
      case 13: return NT_RUNS(0, 1, 13, 2);
      case 14: return NT_RUNS(0, 1, 14, 2);
      case 15: return NT_RUNS(0, 1, 15, 2);
      case 16: return NT_RUNS(0, 1, 16, 2);
      case 19: return NT_RUNS(0, 1, 19, 2);
      case 20: return NT_RUNS(0, 1, 20, 2);
      case 22: return NT_RUNS(0, 1, 22, 2);
      case 23: return NT_RUNS(0, 1, 23, 2);
      default: return NT_RUNS(0, 1, 25, 2);
    }
  }
  if (ra.k != 31 || !(plan.C == 15 || (plan.C == 30 && ra.m == 1)))
    return fail(NTHIP_ERR_HIP, "no specialised run-split kernel for k=%u, run length %u", ra.k, plan.C);
  if (ra.m == 1 && plan.C == 15) return NT_RUNS(31, 1, 15, 2);
  if (ra.m == 1 && plan.C == 30) return NT_RUNS(31, 1, 30, 2);
  if (ra.m == 4 && !c->tune.no_m4) return NT_RUNS(31, 4, 15, 2); // BASELINE config 3
  if (ra.m == 2 && !c->tune.no_m4) return NT_RUNS(31, 2, 15, 2); // (compile-time m: the copy-out's division and
  if (ra.m == 3 && !c->tune.no_m4) return NT_RUNS(31, 3, 15, 2); //  multiplier fetch fold into constants)
  return NT_RUNS(31, 0, 15, 2);
#undef NT_RUNS
}

int ntamd::host::launch_kmer_runs_packed(nthip_ctx* c, const KmerRunsArgs& ra, const RunsPlan& plan)
{
#if KR_CHUNKED
  (void)c; (void)ra; (void)plan;
  return fail(NTHIP_ERR_UNSUPPORTED, "packed input: not in a windowed (KR_CHUNKED) build");
#else
  return launch_kmer_runs(c, kmer_runs_kernel<31, 1, 15, 2, true, true>, ra, plan.lds);
#endif
}
