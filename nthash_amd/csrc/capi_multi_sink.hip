// capi_multi_sink.hip -- one node, several GPUs: device-resident shards, consumers, and the ONLY inter-GPU step of the
// whole design -- the merge of the consumers' results over peer copies (SURVEY.md 5 "distributed", 8e, 8f-1).
// Part of libnthash_hip.so (include/nthash_hip.h); see capi_internal.hpp for the file map.
//
// Reads are independent (reference include/nthash/nthash.hpp:196-204: all state is per object), so every device consumes
// its own shard into its own table -- Bloom filter, counting sketch, MinHash signature of the set -- exactly as a single
// device does (capi_sink_bloom.hip, capi_sink_minhash.hip); what then crosses xGMI is the RESULT, never the hash stream
// (8 m bytes per k-mer: bandwidth-inverted).  The merge is a ring reduce-scatter with an element-wise
// operator that RCCL does not have for these types -- OR of filter words, saturating add of one-byte counters, minimum
// of 64-bit signature entries: in step s device g receives segment (g - 1 - s) mod G of its left neighbour's table
// (hipMemcpyPeerAsync into a staging buffer: one xGMI link per device and step, all links busy at once) and folds it into
// its own copy; after G - 1 steps device g holds the finished segment (g + 1) mod G.  Then either those segments are
// copied to the first device (result there only: (G - 1) / G of a table over its links) or an all-gather ring leaves the
// merged table on every device (NTHIP_MULTI_ALLGATHER: what a sharded QUERY needs next).  Host threads, one per device,
// meet at a barrier between steps; a table of 4 GiB over 8 devices is 7 + 7 steps of 512 MiB.
#include "capi_internal.hpp"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

using namespace ntamd;
using namespace ntamd::host;

namespace {

// dst = dst (op) src over n 16-byte vectors + a tail of single bytes / words handled by the caller's alignment
template <int OP>
__global__ __launch_bounds__(256) void merge_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, uint64_t n_vec)
{
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (uint64_t)gridDim.x * blockDim.x) {
    uint4 a = dst[i];
    const uint4 b = src[i];
    if (OP == NTHIP_MERGE_OR) {
      a.x |= b.x; a.y |= b.y; a.z |= b.z; a.w |= b.w;
    } else if (OP == NTHIP_MERGE_ADD_SAT_U8) {
      // four one-byte counters per word: add the low 7 bits of every byte, then the top bits with their carries; a
      // byte whose sum passes 255 becomes 255
      auto sat4 = [](uint32_t x, uint32_t y) -> uint32_t {
        const uint32_t lo = (x & 0x7f7f7f7fu) + (y & 0x7f7f7f7fu);       // bit 7 of a byte: carry out of its low 7 bits
        const uint32_t hx = x & 0x80808080u, hy = y & 0x80808080u, c = lo & 0x80808080u;
        const uint32_t over = (hx & hy) | (hx & c) | (hy & c);            // carry out of bit 7: the byte overflows
        const uint32_t sum = (lo & 0x7f7f7f7fu) | ((hx ^ hy ^ c) & 0x80808080u);
        const uint32_t full = (over >> 7) * 0xFFu;                        // 0xFF in every overflowing byte
        return sum | full;
      };
      a.x = sat4(a.x, b.x); a.y = sat4(a.y, b.y); a.z = sat4(a.z, b.z); a.w = sat4(a.w, b.w);
    } else { // NTHIP_MERGE_MIN_U64
      const uint64_t a0 = ((uint64_t)a.y << 32) | a.x, a1 = ((uint64_t)a.w << 32) | a.z;
      const uint64_t b0 = ((uint64_t)b.y << 32) | b.x, b1 = ((uint64_t)b.w << 32) | b.z;
      const uint64_t m0 = b0 < a0 ? b0 : a0, m1 = b1 < a1 ? b1 : a1;
      a = make_uint4((uint32_t)m0, (uint32_t)(m0 >> 32), (uint32_t)m1, (uint32_t)(m1 >> 32));
    }
    dst[i] = a;
  }
}

// sig[i] = min over the reads of per_read[r * m + i] (the MinHash signature of the whole SET of k-mers: what a sample
// keeps, next to the per-read signatures of nthip_kmer_minhash)
__global__ __launch_bounds__(256) void column_min_kernel(const uint64_t* __restrict__ per_read, uint64_t n_reads, uint32_t m,
                                                         unsigned long long* __restrict__ sig)
{
  __shared__ uint64_t part[256];
  for (uint32_t i = 0; i < m; ++i) {
    uint64_t mn = ~0ull;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += (uint64_t)gridDim.x * blockDim.x) {
      const uint64_t v = per_read[r * m + i];
      mn = v < mn ? v : mn;
    }
    part[threadIdx.x] = mn;
    __syncthreads();
    for (uint32_t d = 128; d > 0; d >>= 1) {
      if (threadIdx.x < d) part[threadIdx.x] = part[threadIdx.x + d] < part[threadIdx.x] ? part[threadIdx.x + d] : part[threadIdx.x];
      __syncthreads();
    }
    if (threadIdx.x == 0) atomicMin(sig + i, (unsigned long long)part[0]);
    __syncthreads();
  }
}

int launch_merge(nthip_ctx* c, int op, void* dst, const void* src, uint64_t bytes)
{
  const uint64_t n_vec = bytes / 16;
  if (n_vec == 0) return NTHIP_OK;
  const unsigned grid = (unsigned)std::min<uint64_t>((n_vec + 255) / 256, (uint64_t)c->n_cu * 16);
  if (op == NTHIP_MERGE_OR) hipLaunchKernelGGL(merge_kernel<NTHIP_MERGE_OR>, dim3(grid), dim3(256), 0, c->stream, (uint4*)dst, (const uint4*)src, n_vec);
  else if (op == NTHIP_MERGE_ADD_SAT_U8)
    hipLaunchKernelGGL(merge_kernel<NTHIP_MERGE_ADD_SAT_U8>, dim3(grid), dim3(256), 0, c->stream, (uint4*)dst, (const uint4*)src, n_vec);
  else hipLaunchKernelGGL(merge_kernel<NTHIP_MERGE_MIN_U64>, dim3(grid), dim3(256), 0, c->stream, (uint4*)dst, (const uint4*)src, n_vec);
  HIPCHK(hipGetLastError());
  return NTHIP_OK;
}

struct Barrier {
  std::mutex mu;
  std::condition_variable cv;
  size_t n, waiting = 0, gen = 0;
  explicit Barrier(size_t n_) : n(n_) {}
  void wait()
  {
    std::unique_lock<std::mutex> lk(mu);
    const size_t g = gen;
    if (++waiting == n) {
      waiting = 0;
      ++gen;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return gen != g; });
    }
  }
};

// one host thread per device runs `work(g)`; the first failure is reported (its thread-local message carried over)
template <typename F>
int on_every_device(nthip_multi* m, F work)
{
  const size_t G = m->ctx.size();
  std::vector<int> rc(G, NTHIP_OK);
  std::vector<std::string> err(G);
  std::vector<std::thread> th;
  for (size_t g = 0; g < G; ++g)
    th.emplace_back([&, g] {
      rc[g] = work(g);
      if (rc[g] != NTHIP_OK) err[g] = nthip_last_error();
    });
  for (auto& t : th) t.join();
  for (size_t g = 0; g < G; ++g)
    if (rc[g] != NTHIP_OK) return fail(rc[g], "device %zu of the set (HIP device %d): %s", g, m->ctx[g]->device, err[g].c_str());
  return NTHIP_OK;
}

// the ring: tables[g] on device g, `bytes` each (a multiple of 16).  Inside a device thread: `bar` is shared by all of them,
// `failed` makes every thread go through the remaining barriers without working.
int ring_merge(nthip_multi* m, size_t g, void* const* tables, uint64_t bytes, int op, bool allgather, Barrier& bar,
               std::atomic<int>& failed)
{
  const size_t G = m->ctx.size();
  nthip_ctx* c = m->ctx[g];
  int rc = NTHIP_OK;
  auto seg_off = [&](size_t s) { return (bytes / 16 * s / G) * 16; }; // segment s = [seg_off(s), seg_off(s + 1))
  uint64_t seg_max = 0;
  for (size_t s = 0; s < G; ++s) seg_max = std::max(seg_max, seg_off(s + 1) - seg_off(s));
  void* tmp = nullptr;
  auto step = [&](auto body) { // every thread passes every barrier, working or not
    if (rc == NTHIP_OK && failed.load() == 0) {
      rc = body();
      if (rc != NTHIP_OK) failed.store(1);
    }
    bar.wait();
  };
  step([&]() -> int {
    HIPCHK(hipSetDevice(c->device));
    if (G > 1 && seg_max) HIPCHK(hipMalloc(&tmp, seg_max));
    HIPCHK(hipStreamSynchronize(c->stream)); // (the consumer's kernels on this device are done: the neighbours may read)
    return NTHIP_OK;
  });
  const size_t left = (g + G - 1) % G;
  // reduce-scatter
  for (size_t s = 0; s + 1 < G; ++s)
    step([&]() -> int {
      const size_t idx = (left + G - s) % G;
      const uint64_t off = seg_off(idx), len = seg_off(idx + 1) - off;
      if (len == 0) return NTHIP_OK;
      HIPCHK(hipMemcpyPeerAsync(tmp, c->device, (const char*)tables[left] + off, m->ctx[left]->device, len, c->stream));
      NTCHK(launch_merge(c, op, (char*)tables[g] + off, tmp, len));
      HIPCHK(hipStreamSynchronize(c->stream));
      return NTHIP_OK;
    });
  // device g now holds the finished segment (g + 1) mod G
  if (allgather) {
    for (size_t s = 0; s + 1 < G; ++s)
      step([&]() -> int {
        const size_t idx = (left + 1 + G - s) % G;
        const uint64_t off = seg_off(idx), len = seg_off(idx + 1) - off;
        if (len == 0) return NTHIP_OK;
        HIPCHK(hipMemcpyPeerAsync((char*)tables[g] + off, c->device, (const char*)tables[left] + off, m->ctx[left]->device, len, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        return NTHIP_OK;
      });
  } else {
    step([&]() -> int { // the first device collects the finished segments
      if (g != 0) return NTHIP_OK;
      for (size_t o = 1; o < G; ++o) {
        const size_t idx = (o + 1) % G;
        const uint64_t off = seg_off(idx), len = seg_off(idx + 1) - off;
        if (len) HIPCHK(hipMemcpyPeerAsync((char*)tables[0] + off, c->device, (const char*)tables[o] + off, m->ctx[o]->device, len, c->stream));
      }
      HIPCHK(hipStreamSynchronize(c->stream));
      return NTHIP_OK;
    });
  }
  if (tmp) (void)hipFree(tmp);
  return rc;
}

int check_multi_tables(nthip_multi* m, const void* const* tables, const char* what)
{
  if (!m) return fail(NTHIP_ERR_ARG, "multi is NULL");
  if (!tables) return fail(NTHIP_ERR_ARG, "%s is NULL", what);
  for (size_t g = 0; g < m->ctx.size(); ++g)
    if (!tables[g]) return fail(NTHIP_ERR_ARG, "%s[%zu] is NULL", what, g);
  return NTHIP_OK;
}

// consume(g) on every device, then the merge of tables[] -- one set of threads for both
template <typename Consume>
int consume_and_merge(nthip_multi* m, void* const* tables, uint64_t bytes, int op, uint32_t flags, uint64_t* total, Consume consume)
{
  const size_t G = m->ctx.size();
  Barrier bar(G);
  std::atomic<int> failed{0};
  std::vector<uint64_t> tot(G, 0);
  const int rc = on_every_device(m, [&](size_t g) -> int {
    int r = consume(g, &tot[g]);
    if (r != NTHIP_OK) failed.store(1);
    const int rm = ring_merge(m, g, tables, bytes, op, (flags & NTHIP_MULTI_ALLGATHER) != 0, bar, failed);
    return r != NTHIP_OK ? r : rm;
  });
  if (total) {
    *total = 0;
    for (uint64_t t : tot) *total += t;
  }
  return rc;
}

} // namespace

extern "C" int nthip_multi_ctx(nthip_multi* m, int index, nthip_ctx** ctx)
{
  if (!m || !ctx) return fail(NTHIP_ERR_ARG, "multi / ctx is NULL");
  if (index < 0 || (size_t)index >= m->ctx.size()) return fail(NTHIP_ERR_ARG, "device index %d outside the set of %zu", index, m->ctx.size());
  *ctx = m->ctx[(size_t)index];
  return NTHIP_OK;
}

extern "C" int nthip_multi_merge(nthip_multi* m, void* const* d_tables, uint64_t bytes, int op, uint32_t flags)
{
  NTCHK(check_multi_tables(m, (const void* const*)d_tables, "tables"));
  if (bytes % 16) return fail(NTHIP_ERR_ARG, "the tables' size must be a multiple of 16 bytes");
  if (op != NTHIP_MERGE_OR && op != NTHIP_MERGE_ADD_SAT_U8 && op != NTHIP_MERGE_MIN_U64) return fail(NTHIP_ERR_ARG, "unknown merge operator %d", op);
  Barrier bar(m->ctx.size());
  std::atomic<int> failed{0};
  return on_every_device(m, [&](size_t g) { return ring_merge(m, g, d_tables, bytes, op, (flags & NTHIP_MULTI_ALLGATHER) != 0, bar, failed); });
}

extern "C" int nthip_multi_kmer_hash_shards(nthip_multi* m, const nthip_reads* shards, uint16_t k, uint8_t mh, const nthip_out* outs,
                                            uint64_t* totals, uint32_t flags)
{
  if (!m || !shards || !outs) return fail(NTHIP_ERR_ARG, "multi / shards / outs is NULL");
  return on_every_device(m, [&](size_t g) -> int {
    uint64_t t = 0;
    const int rc = shards[g].n_reads ? nthip_kmer_hash(m->ctx[g], &shards[g], k, mh, &outs[g], &t, flags) : NTHIP_OK;
    if (totals) totals[g] = t;
    return rc;
  });
}

extern "C" int nthip_multi_kmer_bloom_insert(nthip_multi* m, const nthip_reads* shards, uint16_t k, uint8_t mh, uint8_t* const* d_filters,
                                             uint64_t n_bits, uint64_t* total, uint32_t flags)
{
  NTCHK(check_multi_tables(m, (const void* const*)d_filters, "filters"));
  if (!shards) return fail(NTHIP_ERR_ARG, "shards is NULL");
  if (n_bits == 0 || n_bits % 128) return fail(NTHIP_ERR_ARG, "the multi-device filter must be a multiple of 128 bits (16 bytes)");
  return consume_and_merge(m, (void* const*)d_filters, n_bits / 8, NTHIP_MERGE_OR, flags, total, [&](size_t g, uint64_t* t) -> int {
    return shards[g].n_reads ? nthip_kmer_bloom_insert(m->ctx[g], &shards[g], k, mh, d_filters[g], n_bits, t, flags & NTHIP_HOST_INPUT) : NTHIP_OK;
  });
}

extern "C" int nthip_multi_kmer_count_insert(nthip_multi* m, const nthip_reads* shards, uint16_t k, uint8_t mh, uint8_t* const* d_counters,
                                             uint64_t n_counters, uint64_t* total, uint32_t flags)
{
  NTCHK(check_multi_tables(m, (const void* const*)d_counters, "counters"));
  if (!shards) return fail(NTHIP_ERR_ARG, "shards is NULL");
  if (n_counters == 0 || n_counters % 16) return fail(NTHIP_ERR_ARG, "the multi-device sketch must be a multiple of 16 counters");
  return consume_and_merge(m, (void* const*)d_counters, n_counters, NTHIP_MERGE_ADD_SAT_U8, flags, total, [&](size_t g, uint64_t* t) -> int {
    return shards[g].n_reads ? nthip_kmer_count_insert(m->ctx[g], &shards[g], k, mh, d_counters[g], n_counters, t, flags & NTHIP_HOST_INPUT)
                             : NTHIP_OK;
  });
}

extern "C" int nthip_multi_kmer_minhash_set(nthip_multi* m, const nthip_reads* shards, uint16_t k, uint8_t mh, uint64_t* const* d_sigs,
                                            uint64_t* total, uint32_t flags)
{
  NTCHK(check_multi_tables(m, (const void* const*)d_sigs, "signatures"));
  if (!shards) return fail(NTHIP_ERR_ARG, "shards is NULL");
  if (mh == 0) return fail(NTHIP_ERR_UNSUPPORTED, "num_hashes must be >= 1");
  const uint32_t mm = mh, padded = (mm + 1u) & ~1u; // (16-byte merge vectors: an odd m has one entry of padding, all ones)
  return consume_and_merge(m, (void* const*)d_sigs, (uint64_t)padded * 8, NTHIP_MERGE_MIN_U64, flags, total, [&](size_t g, uint64_t* t) -> int {
    nthip_ctx* c = m->ctx[g];
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemsetAsync(d_sigs[g], 0xFF, (size_t)padded * 8, c->stream));
    const uint64_t n = shards[g].n_reads;
    if (n == 0) return NTHIP_OK;
    uint64_t* per_read = nullptr;
    HIPCHK(hipMalloc((void**)&per_read, n * mm * sizeof(uint64_t)));
    int rc = nthip_kmer_minhash(c, &shards[g], k, mh, per_read, t, flags & NTHIP_HOST_INPUT);
    if (rc == NTHIP_OK) {
      hipLaunchKernelGGL(column_min_kernel, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, (uint64_t)c->n_cu * 4)), dim3(256), 0, c->stream,
                         per_read, n, mm, (unsigned long long*)d_sigs[g]);
      if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(NTHIP_ERR_HIP, "column_min_kernel failed");
    }
    (void)hipFree(per_read);
    return rc;
  });
}

// ---- the sharded QUERY that the all-gathered tables are for (round 5) ---------------------------------------------------------
// Reads sharded as for the insert, every device asks ITS copy of the table (nthip_multi_kmer_bloom_insert / _count_insert
// with NTHIP_MULTI_ALLGATHER left the merged table on every device); the answers stay where the reads are: hits[g] /
// estimates[g] are memory of device g (host memory with NTHIP_HOST_OUTPUT).  Nothing crosses a link.
extern "C" int nthip_multi_kmer_bloom_query(nthip_multi* m, const nthip_reads* shards, uint16_t k, uint8_t mh, const uint8_t* const* d_filters,
                                            uint64_t n_bits, uint64_t* const* hits, uint64_t* total, uint64_t* total_hits, uint32_t flags)
{
  NTCHK(check_multi_tables(m, (const void* const*)d_filters, "filters"));
  if (!shards) return fail(NTHIP_ERR_ARG, "shards is NULL");
  const size_t G = m->ctx.size();
  std::vector<uint64_t> tot(G, 0), found(G, 0);
  const int rc = on_every_device(m, [&](size_t g) -> int {
    if (shards[g].n_reads == 0) return NTHIP_OK;
    return nthip_kmer_bloom_query(m->ctx[g], &shards[g], k, mh, d_filters[g], n_bits, hits ? hits[g] : nullptr, &tot[g], &found[g],
                                  flags & (NTHIP_HOST_INPUT | NTHIP_HOST_OUTPUT));
  });
  if (total) {
    *total = 0;
    for (uint64_t t : tot) *total += t;
  }
  if (total_hits) {
    *total_hits = 0;
    for (uint64_t t : found) *total_hits += t;
  }
  return rc;
}

extern "C" int nthip_multi_kmer_count_query(nthip_multi* m, const nthip_reads* shards, uint16_t k, uint8_t mh, const uint8_t* const* d_counters,
                                            uint64_t n_counters, uint8_t* const* estimates, uint64_t* total, uint32_t flags)
{
  NTCHK(check_multi_tables(m, (const void* const*)d_counters, "counters"));
  if (!shards || !estimates) return fail(NTHIP_ERR_ARG, "shards / estimates is NULL");
  const size_t G = m->ctx.size();
  std::vector<uint64_t> tot(G, 0);
  const int rc = on_every_device(m, [&](size_t g) -> int {
    if (shards[g].n_reads == 0) return NTHIP_OK;
    return nthip_kmer_count_query(m->ctx[g], &shards[g], k, mh, d_counters[g], n_counters, estimates[g], &tot[g],
                                  flags & (NTHIP_HOST_INPUT | NTHIP_HOST_OUTPUT));
  });
  if (total) {
    *total = 0;
    for (uint64_t t : tot) *total += t;
  }
  return rc;
}
