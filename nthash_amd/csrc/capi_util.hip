// capi_util.hip -- device scan, graph-extension query, measurement helpers
// Part of libnthash_hip.so (include/nthash_hip.h); see capi_internal.hpp for the file map.
#include "capi_internal.hpp"

#include <vector>
#include "util_kernels.hpp"

using namespace ntamd;
using namespace ntamd::host;

namespace ntamd {
namespace host {

// exclusive scan of n u64 on the device: out[i] = sum(in[0..i)), *d_total = sum
// scratch: needs ceil(n/1024) (+ recursion) extra u64, taken from `sums`
int device_exclusive_scan(nthip_ctx* c, const uint64_t* d_in, uint64_t* d_out, uint64_t n,
                          uint64_t* d_sums, uint64_t* d_total)
{
  const uint64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (nb == 0) {
    HIPCHK(hipMemsetAsync(d_total, 0, sizeof(uint64_t), c->stream));
    return NTHIP_OK;
  }
  hipLaunchKernelGGL(scan_tiles_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, c->stream, d_in, d_out,
                     d_sums, n);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(SCAN_THREADS), 0, c->stream, d_sums, nb, d_total);
  hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, c->stream, d_out, d_sums, n);
  HIPCHK(hipGetLastError());
  return NTHIP_OK;
}

int launch_fill_u64(nthip_ctx* c, uint64_t* d_dst, uint64_t n, uint64_t value)
{
  hipLaunchKernelGGL(fill_u64_kernel, dim3(1024), dim3(256), 0, c->stream, d_dst, n, value);
  HIPCHK(hipGetLastError());
  return NTHIP_OK;
}

int launch_fill_window_pos(nthip_ctx* c, uint32_t* d_pos, uint64_t n_reads, uint32_t nwin, const uint64_t* d_flags,
                           const uint64_t* d_offsets)
{
  hipLaunchKernelGGL(fill_window_pos_kernel, dim3(c->n_cu * 8), dim3(256), 0, c->stream, d_pos, n_reads, nwin, d_flags,
                     d_offsets);
  HIPCHK(hipGetLastError());
  return NTHIP_OK;
}

int offsets_survey_device(nthip_ctx* c, const uint64_t* d_offsets, uint64_t n_reads, uint64_t buf_bytes, OffsetsSurvey* sv)
{
  unsigned long long* d_res = (unsigned long long*)(c->d_small + 160); // see offsets_survey_kernel
  HIPCHK(hipMemsetAsync(d_res, 0, 40, c->stream));
  uint64_t blocks = (n_reads + 255) / 256;
  if (blocks > (uint64_t)c->n_cu * 8) blocks = (uint64_t)c->n_cu * 8;
  hipLaunchKernelGGL(offsets_survey_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, d_offsets, n_reads, buf_bytes,
                     d_res);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(c->h_small + 160, d_res, 40, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  uint64_t res[5];
  memcpy(res, c->h_small + 160, 40);
  sv->off0 = res[0];
  sv->len0 = res[1] >= res[0] ? res[1] - res[0] : 0;
  sv->bad = res[3] != 0;
  sv->uniform = !sv->bad && res[2] == 0;
  sv->max_len = res[4];
  return NTHIP_OK;
}

// One pass over the offsets / spans before a kernel trusts them (a decreasing pair would underflow a length and
// read far outside the buffer): costs one round trip, only on the paths that take caller-made offsets.
int check_offsets_device(nthip_ctx* c, const uint64_t* d_starts, const uint64_t* d_ends, uint64_t n_reads,
                         uint64_t buf_bytes, bool contiguous, uint64_t* max_len)
{
  (void)contiguous;
  if (max_len) *max_len = 0;
  if (n_reads == 0) return NTHIP_OK;
  uint32_t* d_bad = (uint32_t*)(c->d_small + 32); // [32] bad flag (u32), [40] longest span (u64)
  HIPCHK(hipMemsetAsync(d_bad, 0, 16, c->stream));
  uint64_t blocks = (n_reads + 255) / 256;
  if (blocks > (uint64_t)c->n_cu * 8) blocks = (uint64_t)c->n_cu * 8;
  hipLaunchKernelGGL(check_spans_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, d_starts, d_ends, n_reads,
                     buf_bytes, d_bad, (unsigned long long*)(c->d_small + 40));
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(c->h_small + 32, d_bad, 16, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  uint32_t bad = 0;
  memcpy(&bad, c->h_small + 32, 4);
  if (max_len) memcpy(max_len, c->h_small + 40, 8);
  if (bad) return fail(NTHIP_ERR_ARG, "offsets / spans are not non-decreasing or reach outside the read buffer");
  return NTHIP_OK;
}

} // namespace host
} // namespace ntamd

// ==========================================================================
// batched graph-extension query
// ==========================================================================
extern "C" int nthip_kmer_extend(nthip_ctx* c, const char* kmers, uint64_t n, uint16_t k16, uint8_t m8,
                                 uint64_t* self, uint64_t* next, uint64_t* prev, uint32_t flags)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  const uint32_t k = k16, m = m8;
  if (k == 0) return fail(NTHIP_ERR_ARG, "k must be greater than 0"); // src/kmer.cpp:347-349
  if (m == 0) return fail(NTHIP_ERR_UNSUPPORTED, "num_hashes must be >= 1");
  if (n && !kmers) return fail(NTHIP_ERR_ARG, "kmers is NULL");
  if (!self && !next && !prev) return fail(NTHIP_ERR_ARG, "no output requested");
  HIPCHK(hipSetDevice(c->device));
  if (n == 0) return NTHIP_OK;
  std::vector<void*> owned;
  auto cleanup = [&]() { for (void* p : owned) (void)hipFree(p); };
  const uint8_t* d_in = (const uint8_t*)kmers;
  uint64_t *d_self = self, *d_next = next, *d_prev = prev;
  int rc = NTHIP_OK;
  auto dev_alloc = [&](size_t bytes, void** p) -> int {
    HIPCHK(hipMalloc(p, bytes));
    owned.push_back(*p);
    return NTHIP_OK;
  };
  if (flags & NTHIP_HOST_INPUT) {
    void* p = nullptr;
    rc = dev_alloc(n * k, &p);
    if (rc == NTHIP_OK && hipMemcpyAsync(p, kmers, n * k, hipMemcpyHostToDevice, c->stream) != hipSuccess)
      rc = fail(NTHIP_ERR_HIP, "H2D copy failed");
    d_in = (const uint8_t*)p;
  }
  if (rc == NTHIP_OK && (flags & NTHIP_HOST_OUTPUT)) {
    if (self) rc = dev_alloc(n * m * 8, (void**)&d_self);
    if (rc == NTHIP_OK && next) rc = dev_alloc(n * 4 * m * 8, (void**)&d_next);
    if (rc == NTHIP_OK && prev) rc = dev_alloc(n * 4 * m * 8, (void**)&d_prev);
  }
  if (rc != NTHIP_OK) { cleanup(); return rc; }
  const bool aligned16 = (!d_next || ((uintptr_t)d_next & 15u) == 0) && (!d_prev || ((uintptr_t)d_prev & 15u) == 0);
  // byte tables in LDS (beyond the position tables: the fw tables + a 2-bit stream of the wave's k-mers), 16-byte stores
  const uint32_t ntab = kmer_ntab(k);
  const bool any_k = kmer_nw(k) == 0;
  const size_t wave_bits = any_k ? ((((size_t)64 * k + 30) >> 4) + 4) * 4 : 0;
  const size_t lds_fixed = (size_t)ntab * 4096 + 16 * 2048; // tables + a 2 KiB exchange tile per wave
  const size_t lds_cap = (c->lds_max < 160 * 1024 ? c->lds_max : 160 * 1024) - 512;
  uint32_t waves = k > 32 && !any_k ? KX_WIDE_THREADS / 64 : 16;
  while (waves > 1 && lds_fixed + wave_bits * waves > lds_cap) waves /= 2;
  if (aligned16 && lds_fixed + wave_bits * waves <= lds_cap) {
    const uint4* tab = nullptr;
    if (get_kmer_tab(c, k, &tab) != NTHIP_OK) { cleanup(); return NTHIP_ERR_HIP; }
    const size_t lds = lds_fixed + wave_bits * waves;
    const uint32_t threads = waves * 64;
    uint64_t blocks = (n + threads - 1) / threads;
    if (blocks > (uint64_t)c->n_cu * 2) blocks = (uint64_t)c->n_cu * 2;
    auto go = [&](auto kernel) {
      (void)raise_max_dynamic_lds(c->device, reinterpret_cast<const void*>(kernel), lds);
      prof_begin(c, "kmer_extend_tab_kernel");
      hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(threads), lds, c->stream, d_in, n, k, m, tab, ntab, d_self,
                         d_next, d_prev);
      prof_end(c);
    };
    if (any_k) go(kmer_extend_tab_kernel<0>);
    else if (k <= 16) go(kmer_extend_tab_kernel<1>);
    else if (k <= 32) go(kmer_extend_tab_kernel<2>);
    else if (k <= 48) go(kmer_extend_tab_kernel<3>);
    else go(kmer_extend_tab_kernel<4>);
  } else {
    prof_begin(c, "kmer_extend_kernel");
    hipLaunchKernelGGL(kmer_extend_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_in, n, k, m,
                       d_self, d_next, d_prev);
    prof_end(c);
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && (flags & NTHIP_HOST_OUTPUT)) {
    if (self) e = hipMemcpyAsync(self, d_self, n * m * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && next) e = hipMemcpyAsync(next, d_next, n * 4 * m * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && prev) e = hipMemcpyAsync(prev, d_prev, n * 4 * m * 8, hipMemcpyDeviceToHost, c->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  cleanup();
  if (e != hipSuccess) return fail(NTHIP_ERR_HIP, "kmer_extend failed: %s", hipGetErrorString(e));
  return NTHIP_OK;
}

// ==========================================================================
// measurement helpers
// ==========================================================================
extern "C" int nthip_synth_reads(nthip_ctx* c, char* d_dst, uint64_t first_read, uint64_t n_reads,
                                 uint32_t len, uint64_t seed)
{
  if (!c || !d_dst) return fail(NTHIP_ERR_ARG, "ctx/dst is NULL");
  HIPCHK(hipSetDevice(c->device));
  if (n_reads == 0 || len == 0) return NTHIP_OK;
  hipLaunchKernelGGL(synth_reads_kernel, dim3(c->n_cu * 8), dim3(256), 0, c->stream, (uint8_t*)d_dst,
                     first_read, n_reads, len, seed);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  return NTHIP_OK;
}

extern "C" int nthip_checksum(nthip_ctx* c, const uint64_t* d_vals, uint64_t n, uint64_t* sum, uint64_t* xr)
{
  if (!c || !sum || !xr) return fail(NTHIP_ERR_ARG, "ctx/sum/xor is NULL");
  HIPCHK(hipSetDevice(c->device));
  *sum = 0;
  *xr = 0;
  if (n == 0) return NTHIP_OK;
  const unsigned blocks = (unsigned)c->n_cu * 8;
  NTCHK(ensure_scratch(c, 2 * blocks + 16));
  hipLaunchKernelGGL(checksum_kernel, dim3(blocks), dim3(256), 0, c->stream, d_vals, n, c->d_scratch);
  HIPCHK(hipGetLastError());
  std::vector<uint64_t> part(2 * blocks);
  HIPCHK(hipMemcpyAsync(part.data(), c->d_scratch, part.size() * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  for (unsigned b = 0; b < blocks; ++b) {
    *sum += part[2 * b];
    *xr ^= part[2 * b + 1];
  }
  return NTHIP_OK;
}

extern "C" int nthip_copy_bench(nthip_ctx* c, void* d_dst, const void* d_src, size_t bytes, int reps,
                                float* best_ms)
{
  if (!c || !d_dst || !d_src || !best_ms) return fail(NTHIP_ERR_ARG, "NULL argument");
  HIPCHK(hipSetDevice(c->device));
  float best = 1e30f;
  for (int i = 0; i < (reps > 0 ? reps : 1); ++i) {
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    hipLaunchKernelGGL(copy_kernel, dim3(c->n_cu * 8), dim3(256), 0, c->stream, (uint4*)d_dst,
                       (const uint4*)d_src, (uint64_t)(bytes / 16));
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipEventSynchronize(c->ev1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    if (ms < best) best = ms;
  }
  c->ev_valid = false;
  *best_ms = best;
  return NTHIP_OK;
}

extern "C" int nthip_fill_bench(nthip_ctx* c, void* d_dst, size_t bytes, int reps, float* best_ms)
{
  if (!c || !d_dst || !best_ms) return fail(NTHIP_ERR_ARG, "NULL argument");
  HIPCHK(hipSetDevice(c->device));
  float best = 1e30f;
  for (int i = 0; i < (reps > 0 ? reps : 1); ++i) {
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    #ifndef FILL_WAVES
#define FILL_WAVES 8
#endif
    hipLaunchKernelGGL(fill16_kernel, dim3(c->n_cu), dim3(FILL_WAVES * 64), 0, c->stream, (uint4*)d_dst,
                       (uint64_t)(bytes / 16), (uint32_t)i);
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    HIPCHK(hipEventSynchronize(c->ev1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    if (ms < best) best = ms;
  }
  c->ev_valid = false;
  *best_ms = best;
  return NTHIP_OK;
}

// Placement-aware allocation.  On MI355X the page set hipMalloc hands out decides how fast a buffer streams: the same
// write-only fill runs at 5.6-7.1 TB/s over fresh allocations of one size in one process, the hash kernels follow it
// (532-632 G k-mers/s on the headline shape), offsets inside an allocation and physically contiguous allocations do
// not change it (profiles/r02_notes.md 11).  For a long-lived buffer -- the hash stream of a pipeline -- it pays to
// look: up to `candidates` allocations are made (as many at a time as the free memory holds), each is filled once or
// twice with the write-only yardstick, the fastest is kept and the others are freed.  The buffer's content is garbage.
// (Round 5 mapped big buffers from small physical pieces through HIP's virtual-memory API -- one big physical block spreads over
// the memory channels worse than many small ones -- and found buffers mapped, released and mapped again coming back with foreign
// bytes in them (profiles/r05_vmm_reuse_check.txt).  A wrong hash costs more than 3 % of bandwidth: round 6 removed that code
// and its knobs; every buffer of this library is a plain hipMalloc.)
// The hash stream of a round and the answers of a stream query were hipMalloc'ed and freed per call: 53 GB + 6.6 GB for config 4's
// seed pair on 5 M reads.  Memory given back is not free at once -- the next allocation of that size waits for the driver: every
// second or third call of nthip_seed_bloom_query took 3.5-6.5 s instead of 83 ms (tools/seed_query_loop.py).  The context keeps
// them (grow-only; nthip_ctx_trim gives them back).
size_t ntamd::host::scratch_limit_of(nthip_ctx* c)
{
  if (c->scratch_limit) return c->scratch_limit;
  if (!c->device_mem) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
      (void)hipGetLastError();
      total_b = (size_t)64 << 30;
    }
    c->device_mem = total_b;
  }
  return c->device_mem / 2;
}
size_t ntamd::host::round_memory(nthip_ctx* c, size_t reusable, size_t fallback_free)
{
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
    (void)hipGetLastError();
    free_b = fallback_free;
  }
  free_b += reusable;
  const size_t lim = scratch_limit_of(c);
  return free_b < lim ? free_b : lim;
}
extern "C" int nthip_ctx_set_scratch_limit(nthip_ctx* c, size_t bytes)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  if (bytes != 0 && bytes < ((size_t)256 << 20)) return fail(NTHIP_ERR_ARG, "a scratch limit under 256 MiB leaves the consumers' rounds no room");
  HIPCHK(hipSetDevice(c->device));
  c->scratch_limit = bytes;
  // what is held over the new limit goes back now (the next round makes what it needs)
  if (reusable_bytes(c) > scratch_limit_of(c)) return nthip_ctx_trim(c);
  return NTHIP_OK;
}
extern "C" int nthip_ctx_scratch_info(nthip_ctx* c, size_t* kept_bytes, size_t* limit_bytes)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  HIPCHK(hipSetDevice(c->device));
  if (kept_bytes) *kept_bytes = reusable_bytes(c);
  if (limit_bytes) *limit_bytes = scratch_limit_of(c);
  return NTHIP_OK;
}

int ntamd::host::kept_alloc(nthip_ctx* c, int slot, size_t bytes, void** p)
{
  *p = nullptr;
  if (bytes == 0) bytes = 16;
  if (bytes > scratch_limit_of(c)) // (the rounds of the consumers are sized under the limit: a caller that is not asks for too much)
    return fail(NTHIP_ERR_HIP, "%zu MB of a round's hash stream / answers are over the context's scratch limit (%zu MB: nthip_ctx_set_scratch_limit)",
                bytes >> 20, scratch_limit_of(c) >> 20);
  if (c->kept_bytes[slot] < bytes) {
    if (c->kept[slot]) HIPCHK(hipFree(c->kept[slot]));
    c->kept[slot] = nullptr;
    c->kept_bytes[slot] = 0;
    const size_t want = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    if (hipMalloc(&c->kept[slot], want) != hipSuccess) {
      (void)hipGetLastError();
      c->kept[slot] = nullptr;
      return fail(NTHIP_ERR_HIP, "no device memory for %zu MB of a round's hash stream / answers", want >> 20);
    }
    c->kept_bytes[slot] = want;
  }
  *p = c->kept[slot];
  return NTHIP_OK;
}

int ntamd::host::default_alloc(nthip_ctx* c, size_t bytes, void** out)
{
  (void)c;
  HIPCHK(hipMalloc(out, bytes ? bytes : 16));
  return NTHIP_OK;
}

extern "C" int nthip_malloc_probed(nthip_ctx* c, size_t bytes, int candidates, void** out, double* gbps, int* tried)
{
  if (!c || !out) return fail(NTHIP_ERR_ARG, "ctx/dptr is NULL");
  HIPCHK(hipSetDevice(c->device));
  *out = nullptr;
  if (gbps) *gbps = 0;
  if (tried) *tried = 0;
  if (bytes < (size_t)(64u << 20) || candidates <= 1) { // nothing to measure on a small buffer / one candidate: what nthip_malloc gives
    NTCHK(default_alloc(c, bytes, out));
    if (tried) *tried = 1;
    return NTHIP_OK;
  }
  void* best = nullptr;
  float best_ms = 1e30f;
  std::vector<void*> held; // candidates are kept until the end so that the next one gets other pages
  auto give_back = [&](void* p) {
    if (p) (void)hipFree(p);
  };
  int n = 0;
  for (; n < candidates; ++n) {
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    if (n > 0 && free_b < bytes + ((size_t)2 << 30)) {
      // no room for another candidate next to the ones held: give the slower ones back and try once more
      bool freed = false;
      for (void*& p : held)
        if (p && p != best) { give_back(p); p = nullptr; freed = true; }
      if (!freed) break;
      HIPCHK(hipMemGetInfo(&free_b, &total_b));
      if (free_b < bytes + ((size_t)2 << 30)) break;
    }
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
      (void)hipGetLastError();
      if (n == 0) return fail(NTHIP_ERR_HIP, "hipMalloc of %zu bytes failed", bytes);
      break;
    }
    held.push_back(p);
    float ms = 0;
    const int rc = nthip_fill_bench(c, p, bytes, 2, &ms);
    if (rc != NTHIP_OK) {
      for (void* q : held) give_back(q);
      return rc;
    }
    if (ms < best_ms) { best_ms = ms; best = p; }
  }
  for (void* p : held)
    if (p && p != best) give_back(p);
  if (!best) return fail(NTHIP_ERR_HIP, "hipMalloc of %zu bytes failed", bytes);
  *out = best;
  if (gbps) *gbps = (double)bytes / (best_ms * 1e-3) / 1e9;
  if (tried) *tried = n;
  return NTHIP_OK;
}

