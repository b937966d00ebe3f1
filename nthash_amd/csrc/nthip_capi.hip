// nthip_capi.hip -- implementation of include/nthash_hip.h (libnthash_hip.so).
//
// Host-side orchestration of the gfx950 kernels: argument validation (the
// conditions the reference raise_error()s on, src/kmer.cpp:212-225,
// src/seed.cpp:90-95), per-k constant tables, path selection (optimistic dense
// fixed-length kernel, N-aware general kernels as the device-side fallback),
// staging for host buffers, and HIP-event timing of the dominant kernel.
// There is no CPU hashing path in this file or anywhere under nthash_amd/.
#include "../../include/nthash_hip.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <array>
#include <chrono>
#include <map>
#include <string>
#include <vector>

#include "kmer_kernels.hpp"
#include "kmer_runs_kernel.hpp"
#include "kmer_runs_gen_kernel.hpp"
#include "kmer_ragged_kernel.hpp"
#include "nt_math.hpp"
#include "seed_kernels.hpp"
#include "seed_parse.hpp"
#include "util_kernels.hpp"

using namespace ntamd;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIPCHK(call)                                                                    \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess)                                                               \
      return fail(NTHIP_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                  __FILE__, __LINE__);                                                  \
  } while (0)

#define NTCHK(call)             \
  do {                          \
    int rc_ = (call);           \
    if (rc_ != NTHIP_OK) return rc_; \
  } while (0)

} // namespace

struct nthip_ctx {
  int device = 0;
  int n_cu = 0;
  size_t lds_max = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  // small device scratch: [0] dirty flag (u32), [8] total (u64)
  uint8_t* d_small = nullptr;
  uint8_t* h_small = nullptr; // pinned mirror
  // device copy of the large argument blocks of the general kernels
  void* d_args = nullptr;
  size_t d_args_bytes = 0;
  // scratch for counts / scan
  uint64_t* d_scratch = nullptr;
  size_t d_scratch_elems = 0;
  uint64_t* d_scratch2 = nullptr; // second area (tile-level arrays next to read-level ones)
  size_t d_scratch2_elems = 0;
  bool profiling = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool ev_valid = false;
  const char* last_kernel = "";
  bool async_pending = false; // NTHIP_ASYNC launches since the last nthip_ctx_take_dirty: d_small[0] accumulates
  // blocks per CU of (kernel, dynamic LDS) pairs already configured
  std::map<std::pair<const void*, size_t>, int> occ_cache;
  // all-care byte tables for the first window of a run, per k (device memory)
  std::map<uint32_t, uint4*> init_tabs;
  // run length of the general dense kernel per (len, stride, k, m), measured on the first big batch of that shape
  // (the cost model does not see what a longer run costs in waves per CU or LDS conflicts: +-10 % either way)
  std::map<std::array<uint32_t, 4>, uint32_t> run_len_cache;
  // staging arena of the NTHIP_HOST_INPUT / NTHIP_HOST_OUTPUT calls: small host-buffer calls (the C++ facade makes
  // one per object) carve their device copies out of it instead of paying five hipMalloc / hipFree pairs each.
  // Grow-only up to STAGE_ARENA_MAX; calls that need more allocate as before.
  uint8_t* stage_buf = nullptr;
  size_t stage_cap = 0, stage_used = 0, stage_want = 0;
  // buffers of the FASTQ / FASTA streaming driver, kept between calls (pinning and mapping half a GiB costs more
  // than streaming a few GB through it); released by nthip_ctx_trim / nthip_ctx_destroy
  struct FastxBuffers {
    uint8_t* pinned[2] = {nullptr, nullptr};
    uint8_t* d_raw[2] = {nullptr, nullptr};
    uint64_t *d_starts = nullptr, *d_ends = nullptr, *d_counts = nullptr, *d_hashes = nullptr;
    uint64_t pinned_bytes = 0, raw_bytes = 0, reads_cap = 0, hashes_cap = 0;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_h2d[2] = {nullptr, nullptr};
  } fx;
};

namespace {
void fastx_buffers_release(nthip_ctx* c)
{
  auto& b = c->fx;
  if (b.copy_stream) (void)hipStreamSynchronize(b.copy_stream);
  for (int i = 0; i < 2; ++i) {
    if (b.pinned[i]) (void)hipHostFree(b.pinned[i]);
    if (b.d_raw[i]) (void)hipFree(b.d_raw[i]);
    if (b.ev_h2d[i]) (void)hipEventDestroy(b.ev_h2d[i]);
  }
  if (b.d_starts) (void)hipFree(b.d_starts);
  if (b.d_ends) (void)hipFree(b.d_ends);
  if (b.d_counts) (void)hipFree(b.d_counts);
  if (b.d_hashes) (void)hipFree(b.d_hashes);
  if (b.copy_stream) (void)hipStreamDestroy(b.copy_stream);
  b = nthip_ctx::FastxBuffers();
}
} // namespace

struct nthip_seeds {
  nthip_ctx* ctx = nullptr;
  uint32_t n_seeds = 0, k = 0, ntab = 0, care_words = 0;
  bool asymmetric = false;
  uint4* d_tables = nullptr;      // [seed][ntab][256]
  uint32_t* d_care = nullptr;     // [seed][care_words]
  uint32_t* d_blk_start = nullptr;
  uint32_t* d_blk_count = nullptr;
  uint32_t* d_blk_pairs = nullptr;
};

namespace {

int ensure_scratch(nthip_ctx* c, size_t elems)
{
  if (c->d_scratch_elems >= elems) return NTHIP_OK;
  if (c->d_scratch) HIPCHK(hipFree(c->d_scratch));
  c->d_scratch = nullptr;
  c->d_scratch_elems = 0;
  HIPCHK(hipMalloc((void**)&c->d_scratch, elems * sizeof(uint64_t)));
  c->d_scratch_elems = elems;
  return NTHIP_OK;
}

int ensure_scratch2(nthip_ctx* c, size_t elems)
{
  if (c->d_scratch2_elems >= elems) return NTHIP_OK;
  if (c->d_scratch2) HIPCHK(hipFree(c->d_scratch2));
  c->d_scratch2 = nullptr;
  c->d_scratch2_elems = 0;
  HIPCHK(hipMalloc((void**)&c->d_scratch2, elems * sizeof(uint64_t)));
  c->d_scratch2_elems = elems;
  return NTHIP_OK;
}

int ensure_args(nthip_ctx* c, size_t bytes)
{
  if (c->d_args_bytes >= bytes) return NTHIP_OK;
  if (c->d_args) HIPCHK(hipFree(c->d_args));
  c->d_args = nullptr;
  HIPCHK(hipMalloc(&c->d_args, bytes));
  c->d_args_bytes = bytes;
  return NTHIP_OK;
}

void prof_begin(nthip_ctx* c, const char* name)
{
  c->last_kernel = name;
  if (c->profiling) {
    (void)hipEventRecord(c->ev0, c->stream);
    c->ev_valid = false;
  }
}
void prof_end(nthip_ctx* c)
{
  if (c->profiling) {
    (void)hipEventRecord(c->ev1, c->stream);
    c->ev_valid = true;
  }
}

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

struct Staged {
  // device views of the caller's buffers (staged copies when host flags are set)
  const uint8_t* seqs = nullptr;
  const uint64_t* offsets = nullptr;
  uint64_t* hashes = nullptr;
  uint64_t* counts = nullptr;
  uint32_t* pos = nullptr;
  uint64_t* fwd = nullptr;
  uint64_t* rev = nullptr;
  std::vector<void*> owned;
  ~Staged()
  {
    for (void* p : owned) (void)hipFree(p);
  }
};

constexpr size_t STAGE_ARENA_MAX = 64u << 20;

// a call that stages host buffers starts here (stage_inputs is its first staging step): the arena is free again
// (staged calls end synchronised), and grows to what the previous call would have liked
int stage_begin(nthip_ctx* c)
{
  if (c->stage_want > c->stage_cap && c->stage_want <= STAGE_ARENA_MAX) {
    if (c->stage_buf) HIPCHK(hipFree(c->stage_buf));
    c->stage_buf = nullptr;
    c->stage_cap = 0;
    size_t cap = c->stage_want + c->stage_want / 4 + 4096;
    if (cap > STAGE_ARENA_MAX) cap = STAGE_ARENA_MAX;
    HIPCHK(hipMalloc((void**)&c->stage_buf, cap));
    c->stage_cap = cap;
  }
  c->stage_used = c->stage_want = 0;
  return NTHIP_OK;
}

// device memory for one staged buffer of this call: from the arena when it fits, its own allocation otherwise
int stage_alloc(nthip_ctx* c, Staged& st, size_t bytes, void** p)
{
  const size_t need = ((bytes ? bytes : 16) + 255) & ~(size_t)255;
  c->stage_want += need;
  if (c->stage_buf && c->stage_used + need <= c->stage_cap) {
    *p = c->stage_buf + c->stage_used;
    c->stage_used += need;
    return NTHIP_OK;
  }
  HIPCHK(hipMalloc(p, need));
  st.owned.push_back(*p);
  return NTHIP_OK;
}

int stage_inputs(nthip_ctx* c, const nthip_reads* rd, uint32_t flags, uint64_t total_bytes, Staged& st)
{
  if (flags & (NTHIP_HOST_INPUT | NTHIP_HOST_OUTPUT)) NTCHK(stage_begin(c));
  if (flags & NTHIP_HOST_INPUT) {
    void* d = nullptr;
    NTCHK(stage_alloc(c, st, total_bytes, &d));
    if (total_bytes) HIPCHK(hipMemcpyAsync(d, rd->seqs, total_bytes, hipMemcpyHostToDevice, c->stream));
    st.seqs = (const uint8_t*)d;
    if (rd->offsets) {
      void* o = nullptr;
      NTCHK(stage_alloc(c, st, (rd->n_reads + 1) * sizeof(uint64_t), &o));
      HIPCHK(hipMemcpyAsync(o, rd->offsets, (rd->n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice,
                            c->stream));
      st.offsets = (const uint64_t*)o;
    }
  } else {
    st.seqs = (const uint8_t*)rd->seqs;
    st.offsets = rd->offsets;
  }
  return NTHIP_OK;
}

int stage_outputs(nthip_ctx* c, const nthip_out* out, uint32_t flags, uint64_t n_reads, uint32_t per,
                  Staged& st, uint32_t strands_per = 1)
{
  if (flags & NTHIP_HOST_OUTPUT) {
    auto alloc = [&](size_t bytes, void** p) -> int { return stage_alloc(c, st, bytes, p); };
    NTCHK(alloc(out->capacity * per * sizeof(uint64_t), (void**)&st.hashes));
    if (out->counts) NTCHK(alloc(n_reads * sizeof(uint64_t), (void**)&st.counts));
    if (out->pos) NTCHK(alloc(out->capacity * sizeof(uint32_t), (void**)&st.pos));
    if (out->fwd) NTCHK(alloc(out->capacity * strands_per * sizeof(uint64_t), (void**)&st.fwd));
    if (out->rev) NTCHK(alloc(out->capacity * strands_per * sizeof(uint64_t), (void**)&st.rev));
  } else {
    st.hashes = out->hashes;
    st.counts = out->counts;
    st.pos = out->pos;
    st.fwd = out->fwd;
    st.rev = out->rev;
  }
  return NTHIP_OK;
}

int unstage_outputs(nthip_ctx* c, const nthip_out* out, uint32_t flags, uint64_t n_reads, uint32_t per,
                    uint64_t total, const Staged& st, uint32_t strands_per = 1)
{
  if (!(flags & NTHIP_HOST_OUTPUT)) return NTHIP_OK;
  if (total) {
    HIPCHK(hipMemcpyAsync(out->hashes, st.hashes, total * per * sizeof(uint64_t), hipMemcpyDeviceToHost,
                          c->stream));
    if (out->pos)
      HIPCHK(hipMemcpyAsync(out->pos, st.pos, total * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    if (out->fwd)
      HIPCHK(hipMemcpyAsync(out->fwd, st.fwd, total * strands_per * sizeof(uint64_t), hipMemcpyDeviceToHost,
                            c->stream));
    if (out->rev)
      HIPCHK(hipMemcpyAsync(out->rev, st.rev, total * strands_per * sizeof(uint64_t), hipMemcpyDeviceToHost,
                            c->stream));
  }
  if (out->counts && n_reads)
    HIPCHK(hipMemcpyAsync(out->counts, st.counts, n_reads * sizeof(uint64_t), hipMemcpyDeviceToHost,
                          c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return NTHIP_OK;
}

// total bytes of the read buffer (needs the last offset when offsets are on the device)
int reads_total_bytes(nthip_ctx* c, const nthip_reads* rd, uint32_t flags, uint64_t* out)
{
  if (rd->n_reads == 0) { *out = 0; return NTHIP_OK; }
  if (rd->offsets) {
    if (flags & NTHIP_HOST_INPUT) {
      *out = rd->offsets[rd->n_reads];
    } else {
      uint64_t last = 0;
      HIPCHK(hipMemcpyAsync(&last, rd->offsets + rd->n_reads, sizeof last, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      *out = last;
    }
  } else {
    const uint32_t stride = rd->stride ? rd->stride : rd->fixed_len;
    *out = (rd->n_reads - 1) * (uint64_t)stride + rd->fixed_len;
  }
  return NTHIP_OK;
}

int check_reads(const nthip_reads* rd)
{
  if (!rd) return fail(NTHIP_ERR_ARG, "reads is NULL");
  if (rd->n_reads && !rd->seqs) return fail(NTHIP_ERR_ARG, "reads->seqs is NULL");
  if (!rd->offsets && rd->fixed_len == 0 && rd->n_reads)
    return fail(NTHIP_ERR_ARG, "reads needs either offsets or fixed_len");
  if (rd->offsets && rd->fixed_len) return fail(NTHIP_ERR_ARG, "reads has both offsets and fixed_len");
  return NTHIP_OK;
}

template <typename K>
int set_max_lds(K kernel, size_t bytes)
{
  // static + dynamic LDS beyond the 64 KiB default needs the opt-in attribute
  if (bytes > 24 * 1024) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  }
  return NTHIP_OK;
}

// blocks per CU for a persistent-style grid; the LDS opt-in and the occupancy
// query are host calls worth ~1 ms, so they are cached per (kernel, LDS size)
template <typename K>
int blocks_per_cu(nthip_ctx* c, K kernel, int threads, size_t dyn_lds, int* out)
{
  const auto key = std::make_pair(reinterpret_cast<const void*>(kernel), dyn_lds);
  auto it = c->occ_cache.find(key);
  if (it == c->occ_cache.end()) {
    NTCHK(set_max_lds(kernel, dyn_lds));
    int per_cu = 0;
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, dyn_lds));
    if (per_cu < 1) per_cu = 1;
    it = c->occ_cache.emplace(key, per_cu).first;
  }
  *out = it->second;
  return NTHIP_OK;
}

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

// ==========================================================================
// library / context
// ==========================================================================
extern "C" const char* nthip_version(void) { return "nthash_amd 0.1 (gfx950; ntHash_v2 bit-exact)"; }
extern "C" const char* nthip_last_error(void) { return g_err.c_str(); }

extern "C" int nthip_device_count(int* count)
{
  if (!count) return fail(NTHIP_ERR_ARG, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(NTHIP_ERR_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = n;
  return NTHIP_OK;
}

extern "C" int nthip_ctx_create(int device, nthip_ctx** out)
{
  if (!out) return fail(NTHIP_ERR_ARG, "ctx out pointer is NULL");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(NTHIP_ERR_NODEVICE, "no HIP device available (nthash_amd has no CPU fallback)");
  if (device < 0 || device >= n) return fail(NTHIP_ERR_ARG, "device %d out of range [0,%d)", device, n);
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  nthip_ctx* c = new nthip_ctx();
  c->device = device;
  c->n_cu = prop.multiProcessorCount;
  c->lds_max = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor : 65536;
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc((void**)&c->d_small, 64) != hipSuccess ||
      hipHostMalloc((void**)&c->h_small, 64) != hipSuccess ||
      hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
    delete c;
    return fail(NTHIP_ERR_HIP, "context resource creation failed: %s", hipGetErrorString(hipGetLastError()));
  }
  c->stream = c->own_stream;
  *out = c;
  return NTHIP_OK;
}

extern "C" int nthip_ctx_destroy(nthip_ctx* c)
{
  if (!c) return NTHIP_OK;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->d_small) (void)hipFree(c->d_small);
  if (c->h_small) (void)hipHostFree(c->h_small);
  if (c->d_args) (void)hipFree(c->d_args);
  if (c->d_scratch) (void)hipFree(c->d_scratch);
  if (c->d_scratch2) (void)hipFree(c->d_scratch2);
  for (auto& kv : c->init_tabs) (void)hipFree(kv.second);
  fastx_buffers_release(c);
  if (c->stage_buf) (void)hipFree(c->stage_buf);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
  return NTHIP_OK;
}

extern "C" int nthip_ctx_trim(nthip_ctx* c)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  fastx_buffers_release(c);
  if (c->stage_buf) (void)hipFree(c->stage_buf);
  c->stage_buf = nullptr;
  c->stage_cap = c->stage_used = c->stage_want = 0;
  if (c->d_scratch) (void)hipFree(c->d_scratch);
  if (c->d_scratch2) (void)hipFree(c->d_scratch2);
  c->d_scratch = c->d_scratch2 = nullptr;
  c->d_scratch_elems = c->d_scratch2_elems = 0;
  return NTHIP_OK;
}

extern "C" int nthip_ctx_set_stream(nthip_ctx* c, void* s)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  c->stream = s ? (hipStream_t)s : c->own_stream;
  return NTHIP_OK;
}

extern "C" int nthip_ctx_synchronize(nthip_ctx* c)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  return NTHIP_OK;
}

extern "C" int nthip_ctx_take_dirty(nthip_ctx* c, int* dirty)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  HIPCHK(hipSetDevice(c->device));
  uint32_t d = 0;
  if (c->async_pending) {
    HIPCHK(hipMemcpyAsync(c->h_small, c->d_small, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(&d, c->h_small, 4);
    c->async_pending = false;
  } else {
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  if (dirty) *dirty = d ? 1 : 0;
  return NTHIP_OK;
}

extern "C" int nthip_ctx_set_profiling(nthip_ctx* c, int on)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  c->profiling = on != 0;
  c->ev_valid = false;
  return NTHIP_OK;
}

extern "C" int nthip_last_kernel_ms(nthip_ctx* c, float* ms, const char** name)
{
  if (!c || !ms) return fail(NTHIP_ERR_ARG, "ctx/ms is NULL");
  if (!c->ev_valid) return fail(NTHIP_ERR_ARG, "no profiled kernel recorded (enable profiling first)");
  HIPCHK(hipEventSynchronize(c->ev1));
  HIPCHK(hipEventElapsedTime(ms, c->ev0, c->ev1));
  if (name) *name = c->last_kernel;
  return NTHIP_OK;
}

extern "C" int nthip_malloc(nthip_ctx* c, size_t bytes, void** p)
{
  if (!c || !p) return fail(NTHIP_ERR_ARG, "ctx/dptr is NULL");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMalloc(p, bytes ? bytes : 16));
  return NTHIP_OK;
}
extern "C" int nthip_free(nthip_ctx* c, void* p)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipFree(p));
  return NTHIP_OK;
}
extern "C" int nthip_memcpy_h2d(nthip_ctx* c, void* dst, const void* src, size_t bytes)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return NTHIP_OK;
}
extern "C" int nthip_memcpy_d2h(nthip_ctx* c, void* dst, const void* src, size_t bytes)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return NTHIP_OK;
}

extern "C" int nthip_memset(nthip_ctx* c, void* d_dst, int byte_value, size_t bytes)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMemsetAsync(d_dst, byte_value, bytes, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return NTHIP_OK;
}

// ==========================================================================
// k-mer hashing
// ==========================================================================
namespace {

void fill_kmer_consts(uint32_t k, uint32_t m, KmerFixedArgs& a)
{
  // (in,out) pair terms of the roll (src/kmer.cpp:84-94, 164-174):
  //   F' = srol(F) ^ S[in] ^ srol^k(S[out]);  R' = sror(R ^ srol^k(S[~in]) ^ S[~out])
  for (unsigned in = 0; in < 4; ++in)
    for (unsigned o = 0; o < 4; ++o) {
      a.tab[(in << 2) | o][0] = seed_of_code(in) ^ srol_n(seed_of_code(o), k);
      a.tab[(in << 2) | o][1] = srol_n(seed_of_code(in ^ 2u), k) ^ seed_of_code(o ^ 2u);
    }
  // strand hashes of a window of k 'A's: the state before the first real base
  uint64_t f = 0, r = 0;
  for (uint32_t i = 0; i < k; ++i) {
    f = srol1(f) ^ SEED_A;
    r = srol1(r) ^ SEED_T; // all terms equal, so the order of rotation does not matter
  }
  a.f_init = f;
  a.r_init = r;
  for (uint32_t i = 0; i < (uint32_t)KF_MAX_RUNTIME_M; ++i) a.mult[i] = multiplier(k, i);
  (void)m;
}

// Can the fixed-length kernel take this batch?  Returns the dynamic LDS size.
bool kmer_fixed_eligible(const nthip_ctx* c, uint32_t len, uint32_t stride, uint32_t k, uint32_t m,
                         uint32_t* pad_dwords, size_t* dyn_lds)
{
  if (len < k || m > (uint32_t)KF_MAX_RUNTIME_M) return false;
  if (stride > len) return false; // gaps between reads: general path
  const uint32_t pad = (k + 15u) / 16u + 1u;
  const uint64_t slab = 15ull + (uint64_t)(KF_RUNS_PER_BLOCK - 1) * stride + len;
  const uint64_t n_vec = (slab + 15u) >> 4;
  const uint64_t dwords = pad + n_vec + 2;
  const size_t bytes = dwords * 4;
  const size_t static_lds = 4 * KF_TILE_U64 * 8 + 256;
  if (bytes + static_lds > c->lds_max || bytes + static_lds > 160 * 1024) return false;
  *pad_dwords = pad;
  *dyn_lds = bytes;
  return true;
}

// byte tables: entry [jt][byte] = XOR over the byte's 4 bases of the rotated seeds
// of window positions 4jt..4jt+3 (care[p] == 0 drops position p), both strands
void build_byte_tables(uint32_t k, const uint8_t* care, uint4* out)
{
  const uint32_t ntab = (k + 3) / 4;
  for (uint32_t jt = 0; jt < ntab; ++jt)
    for (uint32_t byte = 0; byte < 256; ++byte) {
      uint64_t f = 0, r = 0;
      for (uint32_t q = 0; q < 4; ++q) {
        const uint32_t p = 4 * jt + q;
        if (p >= k || (care && !care[p])) continue;
        const uint32_t code = (byte >> (2 * q)) & 3u;
        f ^= srol_n(seed_of_code(code), k - 1 - p);
        r ^= srol_n(seed_of_code(code ^ 2u), p);
      }
      out[(size_t)jt * 256 + byte] = make_uint4((uint32_t)f, (uint32_t)(f >> 32), (uint32_t)r, (uint32_t)(r >> 32));
    }
}

int get_init_tab(nthip_ctx* c, uint32_t k, const uint4** out)
{
  auto it = c->init_tabs.find(k);
  if (it == c->init_tabs.end()) {
    const uint32_t ntab = 4u * ((k + 15) / 16); // zero tables past ceil(k/4)
    std::vector<uint4> h((size_t)ntab * 256, make_uint4(0, 0, 0, 0));
    build_byte_tables(k, nullptr, h.data());
    uint4* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, h.size() * sizeof(uint4)));
    HIPCHK(hipMemcpy(d, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice));
    it = c->init_tabs.emplace(k, d).first;
  }
  *out = it->second;
  return NTHIP_OK;
}

// k > 64: the run-split kernels hash a run's first window by Horner with ONE k-independent byte table
// (a 4-mer's) plus a 1-mer's for the k % 4 leftover bases -- [0..255] and [256..511] of the same array
constexpr uint32_t KMER_TABLE_K_MAX = 64; // beyond: Horner
int get_kmer_tab(nthip_ctx* c, uint32_t k, const uint4** out)
{
  if (k <= KMER_TABLE_K_MAX) return get_init_tab(c, k, out);
  const uint32_t key = 0xFFFF0004u;
  auto it = c->init_tabs.find(key);
  if (it == c->init_tabs.end()) {
    std::vector<uint4> h(512);
    build_byte_tables(4, nullptr, h.data());
    build_byte_tables(1, nullptr, h.data() + 256);
    uint4* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, h.size() * sizeof(uint4)));
    HIPCHK(hipMemcpy(d, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice));
    it = c->init_tabs.emplace(key, d).first;
  }
  *out = it->second;
  return NTHIP_OK;
}
// byte tables a kernel instantiated for NW = ceil(k/16) window words looks up: 4 per word, the ones past
// ceil(k/4) all zero (so that no lookup needs a branch); k > 64: the two Horner tables
inline uint32_t kmer_ntab(uint32_t k) { return k <= KMER_TABLE_K_MAX ? 4u * ((k + 15) / 16) : 2u; }
inline uint32_t kmer_nw(uint32_t k) { return k <= KMER_TABLE_K_MAX ? (k + 15) / 16 : 0u; }

// Plan for the run-split kernel: run length C | nwin, waves per block, LDS bytes.
struct RunsPlan {
  uint32_t C = 0, rpr = 0, waves = 0, bits_dwords = 0, tile_u64 = 0, nw = 0, dword_tail = 0;
  size_t lds = 0;
};

bool kmer_runs_plan(const nthip_ctx* c, uint32_t len, uint32_t stride, uint32_t k, uint32_t m, RunsPlan* p)
{
  if (len < k || k > 64 || m > (uint32_t)KF_MAX_RUNTIME_M || stride > len || stride == 0) return false;
  const uint32_t nwin = len - k + 1;
  uint32_t best = 0;
  for (uint32_t d = 16; d >= 4; --d)
    if (nwin % d == 0) { best = d; break; }
  if (best < 6)
    for (uint32_t d = 17; d <= 64; ++d) // e.g. a prime window count: the whole read is one run
      if (nwin % d == 0) { best = d; break; }
  // NTHIP_TUNE_RUN_LEN: A/B override of the run length (must divide the window count)
  if (const char* t = getenv("NTHIP_TUNE_RUN_LEN")) {
    const uint32_t d = (uint32_t)atoi(t);
    if (d >= 2 && d <= 64 && nwin % d == 0) best = d;
  }
  if (best == 0) return false;
  p->C = best;
  p->rpr = nwin / best;
  p->nw = (k + 15) / 16;
  p->tile_u64 = 64 * best;
  // reads touched by one wave tile (64 consecutive runs)
  const uint32_t slab_reads = (64 % p->rpr == 0) ? 64 / p->rpr : (p->rpr - 1 + 63) / p->rpr + 1;
  const uint64_t slab_bytes = (uint64_t)(slab_reads - 1) * stride + len;
  uint32_t bd = (uint32_t)((15 + slab_bytes + 15) >> 4) + p->nw + 4;
  bd = (bd + 3u) & ~3u;
  p->bits_dwords = bd;
  p->dword_tail = (15 + slab_bytes <= 1280) ? 1u : 0u;
  const size_t fixed = (size_t)((k + 3) / 4) * 4096 + 256 + 64;
  const size_t per_wave = (size_t)p->tile_u64 * 8 + (size_t)bd * 4;
  const size_t cap = (c->lds_max < 160 * 1024 ? c->lds_max : 160 * 1024) - 512;
  // 8 waves per CU measured 1.5-2 % faster than 16 on the HBM-bound C2 shape; with m > 1 the copy-out does the
  // multi-hash expansion and more waves hide it: 16 waves +6 % (m=4), +9 % (m=8)  (profiles/r01_notes.md)
  uint32_t w_max = m == 1 ? 8 : 16;
  if (const char* t = getenv("NTHIP_TUNE_WAVES")) {
    const uint32_t w = (uint32_t)atoi(t);
    if (w >= 1 && w <= 16) w_max = w;
  }
  for (uint32_t w = w_max; w >= 1; --w) {
    if (fixed + per_wave * w <= cap) {
      p->waves = w;
      p->lds = fixed + per_wave * w;
      return true;
    }
  }
  return false;
}

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

// Geometry of the general run-split kernel (kmer_runs_gen_kernel.hpp): any window count.
// The run length minimises lane work per k-mer: one first-window evaluation (about
// 2 + ntab/2 roll steps' worth) plus C-1 rolls per run, the windows a read's last run recomputes included.
struct GenPlan {
  uint32_t C = 0, rpr = 0, last_start = 0, waves = 0, bits_dwords = 0, tile_u64 = 0, nw = 0, dword_tail = 0;
  size_t lds = 0;
};

// gaps_ok: rows may be padded (stride > len).  The padding travels through the slab like any other byte, so
// the dense pass -- which flags every non-base it stages -- does not take such batches; the N-aware passes do
// (no window reaches into the padding).
// force_c: run length to use (0: the model's choice); model_cap: longest run the model may pick (0: its default)
// no_tile: a consumer that keeps the hashes in registers (MinHash) -- a longer run then costs neither LDS nor
// bank conflicts, only fewer first windows: the model may go to 31 (measured: 150 bp m=1 798 -> 853-868 G k-mers/s,
// m=2 +16 %, m=4 +12 %)
bool kmer_gen_plan(const nthip_ctx* c, uint32_t len, uint32_t stride, uint32_t k, uint32_t m, GenPlan* p,
                   bool gaps_ok = false, uint32_t force_c = 0, uint32_t model_cap = 0, bool no_tile = false)
{
  if (len < k || m == 0 || (stride > len && !gaps_ok) || len >= (1u << 30) || stride >= (1u << 30)) return false;
  const uint32_t nwin = len - k + 1;
  if (stride < nwin) return false; // reads overlapping by more than k-1 bases: other paths
  const uint32_t ntab = (k + 3) / 4; // first-window cost in the model (table lookups or Horner steps)
  uint32_t best = 0;
  double best_cost = 1e30;
  // The kernels take runs of up to 31 windows (the window masks of the N-aware pass are 32 bits wide), but the
  // model stops at 16: it does not see what a larger tile costs in waves per CU.  In-process A/B over 24 shapes
  // (profiles/r01_notes.md): longer runs win 3-8 % where they cut the runs per read sharply (100 bp/k64: 13 -> 19,
  // 1 kb reads, k <= 15) and lose 5-28 % elsewhere (k = 63/64 at 150 bp: -27 %).
  uint32_t c_cap = model_cap ? model_cap : no_tile ? 31 : 16;
  if (const char* t = getenv("NTHIP_TUNE_RUN_MAX")) { // A/B knob: longest run the model may pick
    const uint32_t d = (uint32_t)atoi(t);
    if (d >= 1 && d <= 31) c_cap = d;
  }
  const uint32_t c_hi = nwin < c_cap ? nwin : c_cap;
  for (uint32_t C = c_hi; C >= 1; --C) {
    const uint32_t rpr = (nwin + C - 1) / C;
    // the 64 rows of a tile are C*8 bytes apart: an even C puts several lanes of a ds_write_b64 on the
    // same LDS banks (16-way for C = 16), an odd C none.  Measured: 2-way is nearly free (C = 18 on 101 bp
    // +7 % over C = 15), 4-way is not (C = 20 on 50 bp -12 % against two runs of 10)
    uint32_t g = 2 * C, ways = 1;
    while (ways < 32 && (g & 1) == 0) { g >>= 1; ways <<= 1; }
    ways = ways > 2 ? ways / 2 : 1;
    const double conflict = no_tile || ways <= 1 ? 0.0 : ways == 2 ? 0.05 : ways == 4 ? 0.45 : 1.0;
    const double cost = (double)rpr * (2.0 + 0.5 * ntab + (C - 1) * (1.0 + conflict)) / nwin;
    if (cost < best_cost - 1e-9) { best_cost = cost; best = C; }
  }
  if (force_c >= 1 && force_c <= 31 && force_c <= nwin) best = force_c;
  if (const char* t = getenv("NTHIP_TUNE_RUN_LEN")) { // A/B override
    const uint32_t d = (uint32_t)atoi(t);
    if (d >= 1 && d <= 31 && d <= nwin) best = d; // (the model itself stays at <= 16)
  }
  if (best == 0) return false;
  p->C = best;
  p->rpr = (nwin + best - 1) / best;
  p->last_start = nwin - best;
  p->nw = kmer_nw(k);
  p->tile_u64 = 64 * best + 128;
  // longest slab: 63 run-to-run steps of at most C bases, C + stride - nwin across a read boundary, plus
  // the last run
  const uint64_t crossings = (63 + p->rpr - 1) / p->rpr;
  const uint64_t slab_bytes = 64ull * best + k - 1 + crossings * (uint64_t)(stride - nwin);
  uint32_t bd = (uint32_t)((15 + slab_bytes + 15) >> 4) + p->nw + 6;
  bd = (bd + 3u) & ~3u;
  p->bits_dwords = bd;
  p->dword_tail = (15 + slab_bytes <= 1280) ? 1u : 0u;
  const size_t fixed = (size_t)kmer_ntab(k) * 4096 + 256 + 64;
  const size_t per_wave = (size_t)p->tile_u64 * 8 + (size_t)bd * 4;
  const size_t cap = (c->lds_max < 160 * 1024 ? c->lds_max : 160 * 1024) - 512;
  uint32_t w_max = m == 1 ? 8 : 16;
  if (const char* t = getenv("NTHIP_TUNE_WAVES")) {
    const uint32_t w = (uint32_t)atoi(t);
    if (w >= 1 && w <= 16) w_max = w;
  }
  for (uint32_t w = w_max; w >= 1; --w)
    if (fixed + per_wave * w <= cap) {
      p->waves = w;
      p->lds = fixed + per_wave * w;
      return true;
    }
  return false;
}

template <typename K>
int launch_kmer_runs_gen(nthip_ctx* c, K kernel, KmerRunsGenArgs a, size_t dyn_lds, const char* label)
{
  int per_cu = 1;
  NTCHK(blocks_per_cu(c, kernel, (int)a.waves * 64, dyn_lds, &per_cu));
  const uint64_t need = (a.n_wtiles + a.waves - 1) / a.waves;
  uint64_t grid = (uint64_t)c->n_cu * per_cu;
  if (grid > need) grid = need;
  if (a.tile_map == 0xFFFFFFFFu) a.tile_map = (uint32_t)grid;
  prof_begin(c, label);
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(a.waves * 64), dyn_lds, c->stream, a);
  prof_end(c);
  HIPCHK(hipGetLastError());
  return NTHIP_OK;
}

// N-aware run-split path for fixed-length reads: count pass -> scan -> compact hash pass
// (kmer_runs_gen_kernel.hpp, NA = true).  Same geometry as the dense general kernel.
struct NaPlan {
  GenPlan g;
  uint32_t vbits_dwords = 0, ptile_dwords = 0, waves = 0, tile_u64 = 0;
  size_t lds = 0;
};

bool kmer_na_plan(const nthip_ctx* c, uint32_t len, uint32_t stride, uint32_t k, uint32_t m, bool want_pos,
                  NaPlan* p, uint32_t register_sink_u64 = 0, uint32_t force_c = 0, uint32_t model_cap = 0)
{
  if (!kmer_gen_plan(c, len, stride, k, m, &p->g, /*gaps_ok*/ true, force_c, model_cap, register_sink_u64 != 0))
    return false;
  const GenPlan& g = p->g;
  p->tile_u64 = 64 * g.C + KRG_ALIGN_U64 + KRG_SLACK_U64;
  // a consumer that keeps the hashes in registers (MinHash) needs no tile, only its fold area
  if (register_sink_u64) p->tile_u64 = (register_sink_u64 + 1u) & ~1u;
  p->ptile_dwords = want_pos ? (64 * g.C + KRG_SLACK_U64 + 3u) & ~3u : 0u;
  p->vbits_dwords = (g.bits_dwords / 2 + 8 + 3u) & ~3u; // 16 validity bits per 32 stream bits, read 4 dwords ahead
  const size_t fixed = (size_t)kmer_ntab(k) * 4096 + 256 + 64;
  const size_t per_wave = (size_t)p->tile_u64 * 8 + ((size_t)p->ptile_dwords + g.bits_dwords + p->vbits_dwords) * 4;
  const size_t cap = (c->lds_max < 160 * 1024 ? c->lds_max : 160 * 1024) - 512;
  uint32_t w_max = register_sink_u64 ? 16 : 8;
  if (const char* t = getenv("NTHIP_TUNE_NA_WAVES")) {
    const uint32_t w = (uint32_t)atoi(t);
    if (w >= 1 && w <= 16) w_max = w;
  }
  for (uint32_t w = w_max; w >= 1; --w)
    if (fixed + per_wave * w <= cap) {
      p->waves = w;
      p->lds = fixed + per_wave * w;
      return true;
    }
  return false;
}

void fill_gen_args(KmerRunsGenArgs& ga, nthip_ctx* c, const Staged& st, const nthip_reads* rd, uint32_t k,
                   uint32_t m, const GenPlan& g, const KmerFixedArgs& consts)
{
  memset(&ga, 0, sizeof ga);
  const uint32_t len = rd->fixed_len, stride = rd->stride ? rd->stride : len;
  ga.seqs = st.seqs;
  ga.hashes = st.hashes;
  ga.dirty = (uint32_t*)c->d_small;
  ga.n_reads = rd->n_reads;
  ga.n_runs = rd->n_reads * g.rpr;
  ga.n_wtiles = (ga.n_runs + 63) / 64;
  ga.total_bytes = (rd->n_reads - 1) * (uint64_t)stride + len;
  ga.len = len;
  ga.stride = stride;
  ga.k = k;
  ga.m = m;
  ga.nwin = len - k + 1;
  ga.C = g.C;
  ga.rpr = g.rpr;
  ga.last_start = g.last_start;
  ga.last_dup = g.rpr * g.C - ga.nwin;
  ga.ntab = kmer_ntab(k);
  ga.waves = g.waves;
  ga.bits_dwords = g.bits_dwords;
  ga.tile_u64 = g.tile_u64;
  ga.inv_rpr = 65536u / g.rpr + 1u;
  { const char* t2 = getenv("NTHIP_TUNE_TILE_MAP"); ga.tile_map = t2 ? (uint32_t)atoi(t2) : 0xFFFFFFFFu; }
  memcpy(ga.tab, consts.tab, sizeof ga.tab);
  memcpy(ga.mult, consts.mult, sizeof ga.mult);
}

template <bool NA, int SINK = SINK_NONE>
int launch_kmer_runs_gen_nw(nthip_ctx* c, const KmerRunsGenArgs& ga, size_t lds, uint32_t nw, bool dt)
{
  const char* label = SINK == SINK_BLOOM_INSERT  ? "kmer_runs_gen_kernel(bloom insert)"
                      : SINK == SINK_MINHASH     ? "kmer_runs_gen_kernel(minhash)"
                      : SINK == SINK_MINHASH1    ? "kmer_runs_gen_kernel(minhash, m = 1)"
                      : SINK == SINK_BLOOM_QUERY ? "kmer_runs_gen_kernel(bloom query)"
                      : NA                       ? "kmer_runs_gen_kernel(N-aware)"
                                                 : "kmer_runs_gen_kernel";
#define NT_GEN(NWT) \
  (dt ? launch_kmer_runs_gen(c, kmer_runs_gen_kernel<NWT, true, NA, SINK>, ga, lds, label) \
      : launch_kmer_runs_gen(c, kmer_runs_gen_kernel<NWT, false, NA, SINK>, ga, lds, label))
  switch (nw) {
    case 0: return NT_GEN(0); // any k
    case 1: return NT_GEN(1);
    case 2: return NT_GEN(2);
    case 3: return NT_GEN(3);
    default: return NT_GEN(4);
  }
#undef NT_GEN
}

int run_kmer_na(nthip_ctx* c, const Staged& st, const nthip_reads* rd, uint32_t k, uint32_t m,
                const NaPlan& plan, const KmerFixedArgs& consts, uint64_t capacity, uint64_t* total)
{
  KmerRunsGenArgs a;
  fill_gen_args(a, c, st, rd, k, m, plan.g, consts);
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
    NTCHK(blocks_per_cu(c, kmer_runs_count_kernel, (int)ca.waves * 64, lds, &per_cu));
    const uint64_t need = (ca.n_wtiles + ca.waves - 1) / ca.waves;
    uint64_t grid = (uint64_t)c->n_cu * per_cu;
    if (grid > need) grid = need;
    hipLaunchKernelGGL(kmer_runs_count_kernel, dim3((unsigned)grid), dim3(ca.waves * 64), lds, c->stream, ca);
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

// Variable-length reads: pre-pass (runs per read, list of reads with windows, tile table) ->
// count pass -> scan -> compact hash pass (kmer_ragged_kernel.hpp).  *handled = false when the
// shape is outside this path (k, m, LDS), the caller then uses the general kernel.
template <int NW>
int launch_kmer_ragged(nthip_ctx* c, int mode, const KmerRaggedArgs& a, size_t dyn_lds)
{
  auto kernel = mode == NA_MODE_COUNT ? kmer_ragged_kernel<NA_MODE_COUNT, NW> : kmer_ragged_kernel<NA_MODE_HASH, NW>;
  int per_cu = 1;
  NTCHK(blocks_per_cu(c, kernel, (int)a.waves * 64, dyn_lds, &per_cu));
  const uint64_t need = (a.n_wtiles + a.waves - 1) / a.waves;
  uint64_t grid = (uint64_t)c->n_cu * per_cu;
  if (grid > need) grid = need;
  if (mode == NA_MODE_HASH) prof_begin(c, "kmer_ragged_kernel");
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(a.waves * 64), dyn_lds, c->stream, a);
  if (mode == NA_MODE_HASH) prof_end(c);
  HIPCHK(hipGetLastError());
  return NTHIP_OK;
}

// reads = spans [starts[r], ends[r]) of the device buffer st.seqs (total_bytes long)
int run_kmer_ragged(nthip_ctx* c, const Staged& st, const uint64_t* d_starts, const uint64_t* d_ends, uint64_t n_reads,
                    uint64_t total_bytes, uint32_t k, uint32_t m, uint64_t capacity, uint64_t* total, bool* handled)
{
  *handled = false;
  const uint32_t C = 15; // run length; the last run of a read may be shorter
  const uint64_t n = n_reads;
  const uint32_t nw = kmer_nw(k);
  // per-wave LDS: a tile touches <= 64 listed reads, each staging its runs' bytes rounded up to 16
  const uint32_t max_vec = (64 * C + 64 * (k - 1 + 15 + 15)) / 16 + 64;
  const uint32_t bits_dwords = (max_vec + nw + 8 + 3u) & ~3u;
  const uint32_t vbits_dwords = ((max_vec + 12) / 2 + 2 + 3u) & ~3u;
  const uint32_t tile_u64 = 64 * C + KRG_ALIGN_U64;
  const uint32_t ptile_dwords = st.pos ? 64 * C : 0;
  const size_t fixed = (size_t)kmer_ntab(k) * 4096 + 256 + 64;
  const size_t per_wave = (size_t)tile_u64 * 8 + ((size_t)ptile_dwords + bits_dwords + vbits_dwords + 2 * 512) * 4;
  const size_t cap = (c->lds_max < 160 * 1024 ? c->lds_max : 160 * 1024) - 512;
  uint32_t waves = 0;
  for (uint32_t w = 8; w >= 1; --w)
    if (fixed + per_wave * w <= cap) { waves = w; break; }
  if (!waves) return NTHIP_OK;
  *handled = true;

  // ---- pre-pass over reads ----------------------------------------------------------------
  const uint64_t nb_r = (n + SCAN_TILE - 1) / SCAN_TILE;
  NTCHK(ensure_scratch(c, 8 * n + nb_r + 16));
  uint64_t* d_rc = c->d_scratch;
  uint64_t* d_flag = c->d_scratch + n;      // flag, then (scanned) nz index
  uint64_t* d_nz_rc = c->d_scratch + 2 * n;
  NzMeta* d_nz_meta = (NzMeta*)(c->d_scratch + 4 * n); // 32-byte records (d_scratch comes from hipMalloc: aligned)
  uint64_t* d_run_base = c->d_scratch + 3 * n;
  uint64_t* d_sums = c->d_scratch + 8 * n;
  uint64_t* d_total = (uint64_t*)(c->d_small + 8);
  const unsigned rblocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(ragged_runs_kernel, dim3(rblocks), dim3(256), 0, c->stream, d_starts, d_ends, n, k, C, d_rc, d_flag);
  NTCHK(device_exclusive_scan(c, d_flag, d_flag, n, d_sums, d_total));
  HIPCHK(hipMemcpyAsync(c->h_small + 8, d_total, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  uint64_t n_nz = 0;
  memcpy(&n_nz, c->h_small + 8, 8);
  *total = 0;
  if (n_nz == 0) { // no read has a window
    if (st.counts) HIPCHK(hipMemsetAsync(st.counts, 0, n * sizeof(uint64_t), c->stream));
    return NTHIP_OK;
  }
  hipLaunchKernelGGL(ragged_scatter_kernel, dim3(rblocks), dim3(256), 0, c->stream, d_rc, d_flag, d_starts, d_ends, n,
                     d_nz_meta, d_nz_rc);
  NTCHK(device_exclusive_scan(c, d_nz_rc, d_run_base, n_nz, d_sums, d_total));
  HIPCHK(hipMemcpyAsync(c->h_small + 8, d_total, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  uint64_t total_runs = 0;
  memcpy(&total_runs, c->h_small + 8, 8);
  const uint64_t nt = (total_runs + 63) / 64;
  const uint64_t nb_t = (nt + SCAN_TILE - 1) / SCAN_TILE;
  NTCHK(ensure_scratch2(c, 4 * nt + nb_t + 16));
  uint64_t* d_tile_j0 = c->d_scratch2;
  uint64_t* d_tile_rem0 = c->d_scratch2 + nt;
  uint64_t* d_tile_cnt = c->d_scratch2 + 2 * nt;
  uint64_t* d_tile_off = c->d_scratch2 + 3 * nt;
  uint64_t* d_sums2 = c->d_scratch2 + 4 * nt;
  hipLaunchKernelGGL(ragged_tiles_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, c->stream, d_run_base,
                     n_nz, nt, d_tile_j0, d_tile_rem0);
  HIPCHK(hipGetLastError());

  KmerRaggedArgs a;
  memset(&a, 0, sizeof a);
  KmerFixedArgs consts;
  memset(&consts, 0, sizeof consts);
  fill_kmer_consts(k, m, consts);
  a.seqs = st.seqs;
  a.starts = d_starts;
  a.ends = d_ends;
  a.total_bytes = total_bytes;
  a.hashes = st.hashes;
  a.pos = st.pos;
  a.counts = st.counts;
  a.tile_counts = d_tile_cnt;
  a.tile_off = d_tile_off;
  a.nz_meta = d_nz_meta;
  a.tile_j0 = d_tile_j0;
  a.tile_rem0 = d_tile_rem0;
  NTCHK(get_kmer_tab(c, k, &a.init_tab));
  a.n_nz = n_nz;
  a.total_runs = total_runs;
  a.n_wtiles = nt;
  a.k = k;
  a.m = m;
  a.C = C;
  a.ntab = kmer_ntab(k);
  a.waves = waves;
  a.bits_dwords = bits_dwords;
  a.vbits_dwords = vbits_dwords;
  a.tile_u64 = tile_u64;
  a.ptile_dwords = ptile_dwords;
  memcpy(a.tab, consts.tab, sizeof a.tab);
  memcpy(a.mult, consts.mult, sizeof a.mult);
  const size_t lds = fixed + per_wave * waves;
  auto launch = [&](int mode) -> int {
    switch (nw) {
      case 0: return launch_kmer_ragged<0>(c, mode, a, lds); // any k
      case 1: return launch_kmer_ragged<1>(c, mode, a, lds);
      case 2: return launch_kmer_ragged<2>(c, mode, a, lds);
      case 3: return launch_kmer_ragged<3>(c, mode, a, lds);
      default: return launch_kmer_ragged<4>(c, mode, a, lds);
    }
  };
  if (st.counts) HIPCHK(hipMemsetAsync(st.counts, 0, n * sizeof(uint64_t), c->stream));
  {
    // count pass: validity bits and the read table only, 16 waves per block
    KmerRaggedArgs ca = a;
    ca.tile_u64 = 0;
    ca.ptile_dwords = 0;
    ca.bits_dwords = 0;
    ca.waves = 16;
    while (ca.waves > 1 && ((size_t)ca.vbits_dwords + 2 * 512) * 4 * ca.waves + 64 > cap) ca.waves /= 2; // long k
    const size_t clds = ((size_t)ca.vbits_dwords + 2 * 512) * 4 * ca.waves + 64;
    NTCHK(launch_kmer_ragged<1>(c, NA_MODE_COUNT, ca, clds));
  }
  NTCHK(device_exclusive_scan(c, d_tile_cnt, d_tile_off, nt, d_sums2, d_total));
  HIPCHK(hipMemcpyAsync(c->h_small + 8, d_total, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  memcpy(total, c->h_small + 8, 8);
  if (*total > capacity)
    return fail(NTHIP_ERR_CAPACITY, "output capacity %llu k-mers < %llu needed",
                (unsigned long long)capacity, (unsigned long long)*total);
  a.counts = nullptr;
  NTCHK(launch(NA_MODE_HASH));
  for (uint32_t sel = 1; sel <= 2; ++sel) { // strand hashes: the hash pass again with another value selected
    uint64_t* dst = sel == 1 ? st.fwd : st.rev;
    if (!dst) continue;
    a.hashes = dst;
    a.pos = nullptr;
    a.m = 1;
    a.value_sel = sel;
    NTCHK(launch(NA_MODE_HASH));
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  return NTHIP_OK;
}

int run_kmer_general(nthip_ctx* c, const Staged& st, const nthip_reads* rd, uint32_t k, uint32_t m,
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

} // namespace

extern "C" int nthip_kmer_hash(nthip_ctx* c, const nthip_reads* rd, uint16_t k16, uint8_t m8,
                               const nthip_out* out, uint64_t* total_out, uint32_t flags)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  NTCHK(check_reads(rd));
  if (!out || !out->hashes) return fail(NTHIP_ERR_ARG, "out->hashes is NULL");
  const uint32_t k = k16, m = m8;
  if (k == 0) return fail(NTHIP_ERR_ARG, "k must be greater than 0"); // src/kmer.cpp:212-214
  if (k < 3) return fail(NTHIP_ERR_UNSUPPORTED, "k < 3 is undefined in the reference (src/kmer.cpp:47)");
  if (m == 0) return fail(NTHIP_ERR_UNSUPPORTED, "num_hashes must be >= 1");
  HIPCHK(hipSetDevice(c->device));
  uint64_t total = 0;
  if (total_out) *total_out = 0;
  if (rd->n_reads == 0) return NTHIP_OK;

  uint64_t total_bytes = 0;
  NTCHK(reads_total_bytes(c, rd, flags, &total_bytes));
  Staged st;
  NTCHK(stage_inputs(c, rd, flags, total_bytes, st));
  NTCHK(stage_outputs(c, out, flags, rd->n_reads, m, st));

  const uint32_t len = rd->fixed_len;
  const uint32_t stride = rd->stride ? rd->stride : len;
  uint32_t pad = 0;
  size_t dyn = 0;
  bool done = false;
  // optimistic dense pass: wanted when the dense stream fits the caller's capacity (a batch with non-bases
  // may still fit when the dense stream does not: the counting paths below decide that)
  const bool rows_ok = !rd->offsets && kmer_fixed_eligible(c, len, stride, k, m, &pad, &dyn);
  // (positions do not need the N-aware pass when the batch turns out clean: every window is emitted)
  const bool want_fast = !rd->offsets && !(flags & NTHIP_FORCE_GENERAL) && !st.fwd && !st.rev &&
                         !(st.pos && (flags & NTHIP_ASYNC)) &&
                         len >= k && rd->n_reads * (uint64_t)(len - k + 1) <= out->capacity;
  // fixed-length reads that are (or may be) dirty, or whose positions are wanted: N-aware run-split path
  NaPlan na_plan;
  const bool na_ok = !rd->offsets && !(flags & (NTHIP_FORCE_GENERAL | NTHIP_FORCE_ROWS)) &&
                     len >= k && kmer_na_plan(c, len, stride, k, m, st.pos != nullptr, &na_plan);
  if (!rd->offsets && len < k) {
    // every read shorter than k: nothing is emitted
    if (st.counts) {
      hipLaunchKernelGGL(fill_u64_kernel, dim3(1024), dim3(256), 0, c->stream, st.counts, rd->n_reads, 0ull);
      HIPCHK(hipGetLastError());
    }
    done = true;
  } else if (want_fast) {
    if ((flags & NTHIP_ASYNC) && (flags & (NTHIP_HOST_INPUT | NTHIP_HOST_OUTPUT)))
      return fail(NTHIP_ERR_UNSUPPORTED, "NTHIP_ASYNC takes device-resident buffers");
    const uint32_t nwin = len - k + 1;
    const uint64_t dense = rd->n_reads * (uint64_t)nwin;
    bool fast_ran = true;
    KmerFixedArgs a;
    memset(&a, 0, sizeof a);
    a.seqs = st.seqs;
    a.hashes = st.hashes;
    a.dirty = (uint32_t*)c->d_small;
    a.n_runs = rd->n_reads;
    a.len = len;
    a.stride = stride;
    a.k = k;
    a.m = m;
    a.nwin = nwin;
    a.pad_dwords = pad;
    const uint64_t n_tiles = (rd->n_reads + KF_RUNS_PER_BLOCK - 1) / KF_RUNS_PER_BLOCK;
    if (n_tiles > 0xFFFFFFFFull) return fail(NTHIP_ERR_UNSUPPORTED, "too many reads for one call");
    a.n_tiles = (uint32_t)n_tiles;
    fill_kmer_consts(k, m, a);
    const bool async = (flags & NTHIP_ASYNC) != 0;
    if (!async && c->async_pending)
      return fail(NTHIP_ERR_ARG, "NTHIP_ASYNC batches are pending: call nthip_ctx_take_dirty first");
    if (!c->async_pending) HIPCHK(hipMemsetAsync(c->d_small, 0, 4, c->stream));
    int rc;
    RunsPlan plan;
    const bool rows_only = (flags & NTHIP_FORCE_ROWS) != 0;
    GenPlan gplan;
    // the specialised k=31 instantiations (run length 15 / 30 dividing the window count); everything
    // else goes to the general run-split kernel
    const char* no_special = getenv("NTHIP_TUNE_NO_SPECIAL"); // A/B: general kernel on the k=31 shapes too
    const bool special = !rows_only && k == 31 && kmer_runs_plan(c, len, stride, k, m, &plan) &&
                         (plan.C == 15 || (plan.C == 30 && m == 1)) && !(no_special && no_special[0] == '1');
    if (special) {
      // run-split kernel: contiguous write-out (see kmer_runs_kernel.hpp)
      KmerRunsArgs ra;
      memset(&ra, 0, sizeof ra);
      ra.seqs = st.seqs;
      ra.hashes = st.hashes;
      ra.dirty = (uint32_t*)c->d_small;
      NTCHK(get_init_tab(c, k, &ra.init_tab));
      ra.n_reads = rd->n_reads;
      ra.n_runs = rd->n_reads * plan.rpr;
      ra.n_wtiles = (ra.n_runs + 63) / 64;
      ra.len = len;
      ra.stride = stride;
      ra.k = k;
      ra.m = m;
      ra.nwin = nwin;
      ra.C = plan.C;
      ra.rpr = plan.rpr;
      ra.ntab = (k + 3) / 4;
      ra.waves = plan.waves;
      ra.bits_dwords = plan.bits_dwords;
      ra.tile_u64 = plan.tile_u64;
      ra.inv_rpr = 65536u / plan.rpr + 1u;
      ra.dword_tail = plan.dword_tail;
      // one tile group per block (set in launch_kmer_runs); NTHIP_TUNE_TILE_MAP overrides for A/B runs
      { const char* t2 = getenv("NTHIP_TUNE_TILE_MAP"); ra.tile_map = t2 ? (uint32_t)atoi(t2) : 0xFFFFFFFFu; }
      memcpy(ra.tab, a.tab, sizeof ra.tab);
      memcpy(ra.mult, a.mult, sizeof ra.mult);
      // NTHIP_TUNE_NO_DWORD_TAIL=1: A/B switch for the slab-tail staging variant (tools/ablate.py)
      const char* tune = getenv("NTHIP_TUNE_NO_DWORD_TAIL");
      const bool dt = plan.dword_tail != 0 && !(tune && tune[0] == '1');
#define NT_RUNS(KT, MT, CT, NWT) \
  (dt ? launch_kmer_runs(c, kmer_runs_kernel<KT, MT, CT, NWT, true>, ra, plan.lds) \
      : launch_kmer_runs(c, kmer_runs_kernel<KT, MT, CT, NWT, false>, ra, plan.lds))
      if (k == 31 && m == 1 && plan.C == 15) rc = NT_RUNS(31, 1, 15, 2);
      else if (k == 31 && m == 1 && plan.C == 30) rc = NT_RUNS(31, 1, 30, 2);
      else if (m == 4 && !getenv("NTHIP_TUNE_NO_M4")) rc = NT_RUNS(31, 4, 15, 2); // BASELINE config 3
      else rc = NT_RUNS(31, 0, 15, 2);
#undef NT_RUNS
    } else if (!rows_only && kmer_gen_plan(c, len, stride, k, m, &gplan)) {
      // any other shape: general run-split kernel (kmer_runs_gen_kernel.hpp)
      KmerRunsGenArgs ga;
      const uint4* gen_tab = nullptr;
      NTCHK(get_kmer_tab(c, k, &gen_tab));
      // big batch of a shape not seen before: time the model's run length against the longer ones it may not pick
      // on a slice of the batch (a few launches of ~1 ms), keep the fastest for this context
      const std::array<uint32_t, 4> shape_key = {len, stride, k, m};
      auto tuned = c->run_len_cache.find(shape_key);
      if (tuned == c->run_len_cache.end() && dense >= (1ull << 30) && !async && !getenv("NTHIP_TUNE_RUN_LEN") &&
          !getenv("NTHIP_TUNE_RUN_MAX") && !getenv("NTHIP_TUNE_NO_AUTOTUNE")) {
        uint32_t cand[4] = {gplan.C, 0, 0, 0};
        const uint32_t caps[3] = {19, 23, 31};
        uint32_t n_cand = 1;
        for (uint32_t cap : caps) {
          GenPlan q;
          if (!kmer_gen_plan(c, len, stride, k, m, &q, false, 0, cap)) continue;
          bool seen = false;
          for (uint32_t i = 0; i < n_cand; ++i) seen = seen || cand[i] == q.C;
          if (!seen) cand[n_cand++] = q.C;
        }
        uint32_t best_c = gplan.C;
        if (n_cand > 1) {
          nthip_reads slice = *rd;
          const uint64_t want = (128ull << 20) / nwin + 1; // ~128 M k-mers (a quarter of a millisecond) per trial
          slice.n_reads = rd->n_reads < want ? rd->n_reads : want;
          hipEvent_t e0 = nullptr, e1 = nullptr;
          HIPCHK(hipEventCreate(&e0));
          HIPCHK(hipEventCreate(&e1));
          float best_ms = 1e30f;
          bool clean = true;
          for (uint32_t i = 0; i < n_cand && clean; ++i) {
            GenPlan q;
            if (!kmer_gen_plan(c, len, stride, k, m, &q, false, cand[i])) continue;
            fill_gen_args(ga, c, st, &slice, k, m, q, a);
            ga.init_tab = gen_tab;
            float ms = 1e30f;
            for (int rep = 0; rep < 2 && clean; ++rep) { // the first launch warms the tables and the clocks
              HIPCHK(hipEventRecord(e0, c->stream));
              const bool prof = c->profiling;
              c->profiling = false;
              const int trc = launch_kmer_runs_gen_nw<false>(c, ga, q.lds, q.nw, q.dword_tail != 0);
              c->profiling = prof;
              NTCHK(trc);
              HIPCHK(hipEventRecord(e1, c->stream));
              HIPCHK(hipMemcpyAsync(c->h_small, c->d_small, 4, hipMemcpyDeviceToHost, c->stream));
              HIPCHK(hipStreamSynchronize(c->stream));
              uint32_t d = 0;
              memcpy(&d, c->h_small, 4);
              if (d) clean = false; // a non-base: the dense kernel stopped early, the times mean nothing
              HIPCHK(hipEventElapsedTime(&ms, e0, e1));
            }
            // (the model's choice is the first candidate: another one has to beat it by 3 % to replace it)
            if (clean && ms < (i == 0 ? best_ms : 0.97f * best_ms)) { best_ms = ms; best_c = cand[i]; }
          }
          (void)hipEventDestroy(e0);
          (void)hipEventDestroy(e1);
          if (clean) tuned = c->run_len_cache.emplace(shape_key, best_c).first;
          else HIPCHK(hipMemsetAsync(c->d_small, 0, 4, c->stream)); // the real pass below finds it again
        } else {
          tuned = c->run_len_cache.emplace(shape_key, gplan.C).first;
        }
      }
      if (tuned != c->run_len_cache.end() && tuned->second != gplan.C &&
          !kmer_gen_plan(c, len, stride, k, m, &gplan, false, tuned->second))
        return fail(NTHIP_ERR_HIP, "run-split plan failed for a tuned run length");
      fill_gen_args(ga, c, st, rd, k, m, gplan, a);
      ga.init_tab = gen_tab;
      rc = launch_kmer_runs_gen_nw<false>(c, ga, gplan.lds, gplan.nw, gplan.dword_tail != 0);
    } else if (!rows_ok) {
      rc = NTHIP_OK;
      fast_ran = false;
    } else if (k == 31 && m == 1) rc = launch_kmer_fixed(c, kmer_fixed_kernel<31, 1>, a, dyn);
    else if (k == 31 && m == 4) rc = launch_kmer_fixed(c, kmer_fixed_kernel<31, 4>, a, dyn);
    else rc = launch_kmer_fixed(c, kmer_fixed_kernel<0, 0>, a, dyn);
    NTCHK(rc);
    if (async) {
      if (!fast_ran) return fail(NTHIP_ERR_UNSUPPORTED, "NTHIP_ASYNC: no dense kernel takes this shape");
      c->async_pending = true;
      if (st.counts) {
        hipLaunchKernelGGL(fill_u64_kernel, dim3(1024), dim3(256), 0, c->stream, st.counts, rd->n_reads,
                           (uint64_t)nwin);
        HIPCHK(hipGetLastError());
      }
      if (total_out) *total_out = dense;
      return NTHIP_OK;
    }
    uint32_t dirty = 1;
    if (fast_ran) {
      HIPCHK(hipMemcpyAsync(c->h_small, c->d_small, 4, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      memcpy(&dirty, c->h_small, 4);
    }
    if (!dirty) {
      total = dense;
      if (st.counts) {
        hipLaunchKernelGGL(fill_u64_kernel, dim3(1024), dim3(256), 0, c->stream, st.counts, rd->n_reads,
                           (uint64_t)nwin);
        HIPCHK(hipGetLastError());
      }
      if (st.pos) { // get_pos() of a read of bases only: the window index
        hipLaunchKernelGGL(fill_window_pos_kernel, dim3(c->n_cu * 8), dim3(256), 0, c->stream, st.pos, rd->n_reads, nwin,
                           (const uint64_t*)nullptr, (const uint64_t*)nullptr);
        HIPCHK(hipGetLastError());
      }
      done = true;
    }
    // dirty: some byte is not ACGTU -> redo on an N-aware path (device side)
  }
  if (!done && (flags & NTHIP_ASYNC))
    return fail(NTHIP_ERR_UNSUPPORTED, "NTHIP_ASYNC: not a plain dense call (offsets, pos / strand outputs, capacity)");
  if (!done && na_ok) {
    KmerFixedArgs consts;
    memset(&consts, 0, sizeof consts);
    fill_kmer_consts(k, m, consts);
    // the run length of the N-aware passes, measured like the dense kernel's (see there): the first big batch of a
    // shape runs count -> scan -> hash on a slice for every candidate
    const uint64_t na_dense = rd->n_reads * (uint64_t)(len - k + 1);
    const std::array<uint32_t, 4> na_key = {len, stride | 0x80000000u, k, m | (st.pos ? 0x100u : 0u)};
    auto na_tuned = c->run_len_cache.find(na_key);
    if (na_tuned == c->run_len_cache.end() && na_dense >= (1ull << 30) && !st.fwd && !st.rev &&
        !getenv("NTHIP_TUNE_RUN_LEN") && !getenv("NTHIP_TUNE_RUN_MAX") && !getenv("NTHIP_TUNE_NO_AUTOTUNE")) {
      uint32_t cand[4] = {na_plan.g.C, 0, 0, 0}, n_cand = 1;
      for (uint32_t cap : {19u, 23u, 31u}) {
        NaPlan q;
        if (!kmer_na_plan(c, len, stride, k, m, st.pos != nullptr, &q, 0, 0, cap)) continue;
        bool seen = false;
        for (uint32_t i = 0; i < n_cand; ++i) seen = seen || cand[i] == q.g.C;
        if (!seen) cand[n_cand++] = q.g.C;
      }
      uint32_t best_c = na_plan.g.C;
      if (n_cand > 1) {
        nthip_reads slice = *rd;
        const uint64_t want = (256ull << 20) / (len - k + 1) + 1; // wall-clock timing (host round trips inside): longer trials
        slice.n_reads = rd->n_reads < want ? rd->n_reads : want;
        double best_s = 1e30;
        const bool prof = c->profiling;
        c->profiling = false;
        for (uint32_t i = 0; i < n_cand; ++i) {
          NaPlan q;
          if (!kmer_na_plan(c, len, stride, k, m, st.pos != nullptr, &q, 0, cand[i])) continue;
          double sec = 1e30;
          int trc = NTHIP_OK;
          for (int rep = 0; rep < 2 && trc == NTHIP_OK; ++rep) {
            uint64_t tt = 0;
            const auto t0 = std::chrono::steady_clock::now();
            trc = run_kmer_na(c, st, &slice, k, m, q, consts, out->capacity, &tt);
            sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
          }
          if (trc != NTHIP_OK) { c->profiling = prof; return trc; }
          if (sec < (i == 0 ? best_s : 0.96 * best_s)) { best_s = sec; best_c = cand[i]; }
        }
        c->profiling = prof;
      }
      na_tuned = c->run_len_cache.emplace(na_key, best_c).first;
    }
    if (na_tuned != c->run_len_cache.end() && na_tuned->second != na_plan.g.C &&
        !kmer_na_plan(c, len, stride, k, m, st.pos != nullptr, &na_plan, 0, na_tuned->second))
      return fail(NTHIP_ERR_HIP, "N-aware plan failed for a tuned run length");
    int rc = run_kmer_na(c, st, rd, k, m, na_plan, consts, out->capacity, &total);
    if (rc == NTHIP_ERR_CAPACITY && total_out) *total_out = total;
    NTCHK(rc);
    done = true;
  }
  if (!done && rd->offsets && !(flags & NTHIP_FORCE_GENERAL)) {
    bool handled = false;
    int rc = run_kmer_ragged(c, st, st.offsets, st.offsets + 1, rd->n_reads, total_bytes, k, m, out->capacity,
                             &total, &handled);
    if (rc == NTHIP_ERR_CAPACITY && total_out) *total_out = total;
    NTCHK(rc);
    done = handled;
  }
  if (!done) NTCHK(run_kmer_general(c, st, rd, k, m, out->capacity, &total));
  if (total_out) *total_out = total;
  NTCHK(unstage_outputs(c, out, flags, rd->n_reads, m, total, st));
  HIPCHK(hipStreamSynchronize(c->stream));
  return NTHIP_OK;
}

// ==========================================================================
// spaced seeds
// ==========================================================================
namespace {

} // namespace

extern "C" int nthip_seeds_create(nthip_ctx* c, const char* const* seeds, uint32_t n_seeds, uint16_t k16,
                                  nthip_seeds** out, int* asymmetric)
{
  if (!c || !out) return fail(NTHIP_ERR_ARG, "ctx/out is NULL");
  *out = nullptr;
  if (!seeds || n_seeds == 0) return fail(NTHIP_ERR_ARG, "no seeds given");
  const uint32_t k = k16;
  if (k == 0) return fail(NTHIP_ERR_ARG, "k must be greater than 0");
  HIPCHK(hipSetDevice(c->device));
  // byte tables per seed: 4 per 16-base window word for k <= 64 (zero past ceil(k/4): the kernels look all of them up)
  const uint32_t ntab = k <= 64 ? 4u * ((k + 15) / 16) : (k + 3) / 4, cw = (k + 31) / 32;
  std::vector<uint4> tables((size_t)n_seeds * ntab * 256, make_uint4(0, 0, 0, 0));
  std::vector<uint32_t> care((size_t)n_seeds * cw, 0), blk_start(n_seeds), blk_count(n_seeds), blk_pairs;
  bool asym = false;
  for (uint32_t s = 0; s < n_seeds; ++s) {
    if (!seeds[s]) return fail(NTHIP_ERR_ARG, "seed %u is NULL", s);
    const std::string str(seeds[s]);
    if (str.size() != k) // src/seed.cpp:90-95
      return fail(NTHIP_ERR_ARG, "Spaced seed string length (%zu) not equal to k=%u in %s", str.size(), k,
                  str.c_str());
    if (!seed_is_symmetric(str)) asym = true; // src/seed.cpp:96-102
    const SeedShape shape = parse_seed_shape(str);
    const std::vector<uint32_t>& pairs = shape.block_pairs;
    const std::vector<uint8_t>& par = shape.care;
    blk_start[s] = (uint32_t)(blk_pairs.size() / 2);
    blk_count[s] = (uint32_t)(pairs.size() / 2);
    blk_pairs.insert(blk_pairs.end(), pairs.begin(), pairs.end());
    for (uint32_t p = 0; p < k; ++p)
      if (par[p]) care[(size_t)s * cw + (p >> 5)] |= 1u << (p & 31);
    // byte tables: entry = XOR over the byte's 4 bases of the masked rotated seeds
    build_byte_tables(k, par.data(), tables.data() + (size_t)s * ntab * 256);
  }
  if (blk_pairs.empty()) blk_pairs.push_back(0);
  nthip_seeds* sd = new nthip_seeds();
  sd->ctx = c;
  sd->n_seeds = n_seeds;
  sd->k = k;
  sd->ntab = ntab;
  sd->care_words = cw;
  sd->asymmetric = asym;
  auto up = [&](const void* src, size_t bytes, void** dst) -> int {
    HIPCHK(hipMalloc(dst, bytes));
    HIPCHK(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return NTHIP_OK;
  };
  int rc = up(tables.data(), tables.size() * sizeof(uint4), (void**)&sd->d_tables);
  if (rc == NTHIP_OK) rc = up(care.data(), care.size() * 4, (void**)&sd->d_care);
  if (rc == NTHIP_OK) rc = up(blk_start.data(), blk_start.size() * 4, (void**)&sd->d_blk_start);
  if (rc == NTHIP_OK) rc = up(blk_count.data(), blk_count.size() * 4, (void**)&sd->d_blk_count);
  if (rc == NTHIP_OK) rc = up(blk_pairs.data(), blk_pairs.size() * 4, (void**)&sd->d_blk_pairs);
  if (rc != NTHIP_OK) {
    nthip_seeds_destroy(sd);
    return rc;
  }
  if (asymmetric) *asymmetric = asym ? 1 : 0;
  *out = sd;
  return NTHIP_OK;
}

extern "C" int nthip_seeds_destroy(nthip_seeds* sd)
{
  if (!sd) return NTHIP_OK;
  if (sd->ctx) (void)hipSetDevice(sd->ctx->device);
  if (sd->d_tables) (void)hipFree(sd->d_tables);
  if (sd->d_care) (void)hipFree(sd->d_care);
  if (sd->d_blk_start) (void)hipFree(sd->d_blk_start);
  if (sd->d_blk_count) (void)hipFree(sd->d_blk_count);
  if (sd->d_blk_pairs) (void)hipFree(sd->d_blk_pairs);
  delete sd;
  return NTHIP_OK;
}

namespace {

// seed_wave_kernel (one wave per read, staged in segments of SEED_WAVE_LMAX bytes): k <= 64, no strand outputs.
constexpr uint32_t SEED_WAVE_LMAX = 2048;
struct SeedWavePlan {
  uint32_t nw = 0, waves_count = 0, waves_hash = 0;
  size_t lds_count = 0, lds_hash = 0;
};
bool seed_wave_plan(const nthip_ctx* c, const nthip_seeds* sd, uint32_t m2, SeedWavePlan* p)
{
  if (sd->k + 64u > SEED_WAVE_LMAX || sd->k < 2) return false; // a segment must hold some windows
  const uint32_t per = sd->n_seeds * m2, lmax = SEED_WAVE_LMAX;
  const uint32_t raw_dw = (lmax + 64u) >> 2, code_dw = (lmax >> 4) + 8u, bit_dw = (lmax >> 5) + 8u;
  const size_t pw_count = (size_t)((raw_dw + code_dw + 2u * bit_dw + 3u) & ~3u) * 4;
  const size_t pw_hash = (size_t)((raw_dw + code_dw + 2u * bit_dw + 64u * per * 2u + 3u) & ~3u) * 4;
  const size_t tables = sd->k <= 64 ? (size_t)sd->n_seeds * sd->ntab * 256 * sizeof(uint4) : 0; // k > 64: Horner, no tables
  const size_t cap = (c->lds_max < 160 * 1024 ? c->lds_max : 160 * 1024) - 512;
  p->nw = sd->k <= 64 ? (sd->k + 15) / 16 : 0;
  for (uint32_t w = 16; w >= 1; --w)
    if (pw_count * w <= cap) { p->waves_count = w; p->lds_count = pw_count * w; break; }
  for (uint32_t w = 16; w >= 2; --w)
    if (tables + pw_hash * w <= cap) { p->waves_hash = w; p->lds_hash = tables + pw_hash * w; break; }
  return p->waves_count && p->waves_hash;
}
template <bool COUNT_ONLY>
int launch_seed_wave(nthip_ctx* c, const SeedWavePlan& plan, uint64_t n_items, bool record = true)
{
  const uint32_t waves = COUNT_ONLY ? plan.waves_count : plan.waves_hash;
  const size_t lds = COUNT_ONLY ? plan.lds_count : plan.lds_hash;
  auto go = [&](auto kernel) -> int {
    int per_cu = 1;
    NTCHK(blocks_per_cu(c, kernel, (int)waves * 64, lds, &per_cu));
    const uint64_t need = (n_items + waves - 1) / waves;
    uint64_t grid = (uint64_t)c->n_cu * per_cu;
    if (grid > need) grid = need;
    if (!COUNT_ONLY && record) prof_begin(c, "seed_wave_kernel");
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(waves * 64), lds, c->stream,
                       (const SeedGeneralArgs*)c->d_args);
    if (!COUNT_ONLY && record) prof_end(c);
    HIPCHK(hipGetLastError());
    return NTHIP_OK;
  };
  switch (plan.nw) {
    case 0: return go(seed_wave_kernel<COUNT_ONLY, 0>); // k > 64
    case 1: return go(seed_wave_kernel<COUNT_ONLY, 1>);
    case 2: return go(seed_wave_kernel<COUNT_ONLY, 2>);
    case 3: return go(seed_wave_kernel<COUNT_ONLY, 3>);
    default: return go(seed_wave_kernel<COUNT_ONLY, 4>);
  }
}

int run_seed_general(nthip_ctx* c, const Staged& st, const nthip_reads* rd, const nthip_seeds* sd,
                     uint32_t m2, uint64_t capacity, uint64_t* total, const uint64_t* d_ends = nullptr)
{
  const uint64_t n = rd->n_reads;
  SeedGeneralArgs h;
  memset(&h, 0, sizeof h);
  h.seqs = st.seqs;
  h.offsets = st.offsets;
  h.ends = d_ends; // spans: st.offsets holds the starts
  h.n_reads = n;
  h.len = rd->fixed_len;
  h.stride = rd->stride ? rd->stride : rd->fixed_len;
  h.k = sd->k;
  h.m2 = m2;
  h.n_seeds = sd->n_seeds;
  h.care_words = sd->care_words;
  h.care_bits = sd->d_care;
  h.blk_start = sd->d_blk_start;
  h.blk_count = sd->d_blk_count;
  h.blk_pairs = sd->d_blk_pairs;
  h.tables = sd->d_tables;
  h.ntab = sd->ntab;
  for (uint32_t i = 0; i < 256; ++i) h.mult[i] = multiplier(sd->k, i);
  const uint64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  NTCHK(ensure_scratch(c, 2 * n + nb + 16));
  uint64_t* d_counts = st.counts ? st.counts : c->d_scratch;
  uint64_t* d_off = c->d_scratch + n;
  uint64_t* d_sums = c->d_scratch + 2 * n;
  uint64_t* d_total = (uint64_t*)(c->d_small + 8);
  NTCHK(ensure_args(c, sizeof(SeedGeneralArgs)));
  const unsigned blocks = (unsigned)((n + 255) / 256);
  // one wave per read (seed_wave_kernel), k <= 64; the lane-per-read kernel otherwise
  SeedWavePlan wplan;
  bool use_wave = !getenv("NTHIP_TUNE_NO_SEED_WAVE") && seed_wave_plan(c, sd, m2, &wplan);
  h.wave_lmax = SEED_WAVE_LMAX;
  h.counts = d_counts;
  if (use_wave) {
    h.wave_waves = wplan.waves_count;
    HIPCHK(hipMemcpyAsync(c->d_args, &h, sizeof h, hipMemcpyHostToDevice, c->stream));
    NTCHK(launch_seed_wave<true>(c, wplan, n));
  } else {
    HIPCHK(hipMemcpyAsync(c->d_args, &h, sizeof h, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(seed_general_kernel<true>, dim3(blocks), dim3(256), 0, c->stream,
                       (const SeedGeneralArgs*)c->d_args);
    HIPCHK(hipGetLastError());
  }
  NTCHK(device_exclusive_scan(c, d_counts, d_off, n, d_sums, d_total));
  HIPCHK(hipMemcpyAsync(c->h_small + 8, d_total, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  memcpy(total, c->h_small + 8, 8);
  if (*total > capacity)
    return fail(NTHIP_ERR_CAPACITY, "output capacity %llu k-mers < %llu needed",
                (unsigned long long)capacity, (unsigned long long)*total);
  h.counts = nullptr;
  h.read_off = d_off;
  h.hashes = st.hashes;
  h.pos = st.pos;
  h.fwd = st.fwd;
  h.rev = st.rev;
  h.capacity = capacity;
  if (use_wave) {
    h.wave_waves = wplan.waves_hash;
    HIPCHK(hipMemcpyAsync(c->d_args, &h, sizeof h, hipMemcpyHostToDevice, c->stream));
    NTCHK(launch_seed_wave<false>(c, wplan, n));
  } else {
    HIPCHK(hipMemcpyAsync(c->d_args, &h, sizeof h, hipMemcpyHostToDevice, c->stream));
    prof_begin(c, "seed_general_kernel");
    hipLaunchKernelGGL(seed_general_kernel<false>, dim3(blocks), dim3(256), 0, c->stream,
                       (const SeedGeneralArgs*)c->d_args);
    prof_end(c);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  return NTHIP_OK;
}

template <typename K>
int launch_seed_fixed(nthip_ctx* c, K kernel, const SeedFixedArgs& a, size_t dyn_lds)
{
  int per_cu = 1;
  NTCHK(blocks_per_cu(c, kernel, SF_THREADS, dyn_lds, &per_cu));
  uint64_t grid = (uint64_t)c->n_cu * per_cu;
  if (grid > a.n_tiles) grid = a.n_tiles;
  prof_begin(c, "seed_fixed_kernel");
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(SF_THREADS), dyn_lds, c->stream, a);
  prof_end(c);
  HIPCHK(hipGetLastError());
  return NTHIP_OK;
}

} // namespace

extern "C" int nthip_seed_hash(nthip_ctx* c, const nthip_reads* rd, const nthip_seeds* sd, uint8_t m28,
                               const nthip_out* out, uint64_t* total_out, uint32_t flags)
{
  if (!c || !sd) return fail(NTHIP_ERR_ARG, "ctx/seeds is NULL");
  NTCHK(check_reads(rd));
  if (!out || !out->hashes) return fail(NTHIP_ERR_ARG, "out->hashes is NULL");
  const uint32_t m2 = m28, k = sd->k;
  if (m2 == 0) return fail(NTHIP_ERR_UNSUPPORTED, "num_hashes_per_seed must be >= 1");
  if (c->async_pending) return fail(NTHIP_ERR_ARG, "NTHIP_ASYNC batches are pending: call nthip_ctx_take_dirty first");
  HIPCHK(hipSetDevice(c->device));
  uint64_t total = 0;
  if (total_out) *total_out = 0;
  if (rd->n_reads == 0) return NTHIP_OK;
  const uint32_t per = sd->n_seeds * m2;

  uint64_t total_bytes = 0;
  NTCHK(reads_total_bytes(c, rd, flags, &total_bytes));
  Staged st;
  NTCHK(stage_inputs(c, rd, flags, total_bytes, st));
  NTCHK(stage_outputs(c, out, flags, rd->n_reads, per, st, sd->n_seeds));

  const uint32_t len = rd->fixed_len;
  const uint32_t stride = rd->stride ? rd->stride : len;
  bool done = false;
  if (!rd->offsets && len < k) {
    if (st.counts) {
      hipLaunchKernelGGL(fill_u64_kernel, dim3(1024), dim3(256), 0, c->stream, st.counts, rd->n_reads, 0ull);
      HIPCHK(hipGetLastError());
    }
    done = true;
  } else if (!rd->offsets && !(flags & NTHIP_FORCE_GENERAL) && !st.fwd && !st.rev && m2 <= (uint32_t)SF_MAX_RUNTIME_M &&
             k <= 64 && stride <= len) {
    const uint32_t nwin = len - k + 1;
    const uint32_t nh = (k + 7) / 8; // 16-bit halves of the window; 2*nh byte tables per seed in LDS (zero-padded)
    const size_t table_bytes = (size_t)sd->n_seeds * 2 * nh * 256 * sizeof(uint4);
    // tile = as many runs as give a ~8 KiB bit stream (32 Ki bases), at most 256
    uint32_t rpt = 32768u / stride;
    if (rpt > 256) rpt = 256;
    if (rpt < 1) rpt = 1;
    {
      // a tile's records are one contiguous piece of the stream: make every tile start on a KiB of it (or the
      // largest power of two below that the read count allows), so that no store splits lines with another block's
      const uint64_t tile_unit = (uint64_t)nwin * per * 8;
      uint32_t mult_of = 1;
      while (mult_of < 128 && ((tile_unit * mult_of) & 1023u) != 0) mult_of <<= 1;
      while (mult_of > 1 && mult_of > rpt) mult_of >>= 1;
      rpt -= rpt % mult_of;
      if (const char* t = getenv("NTHIP_TUNE_SEED_RPT")) { // A/B override
        const uint32_t v = (uint32_t)atoi(t);
        if (v >= 1 && v <= 256) rpt = v;
      }
    }
    const uint64_t slab = 15ull + (uint64_t)(rpt - 1) * stride + len;
    const uint32_t bits_dwords = (uint32_t)((((slab + 15) >> 4) + 8 + 3) & ~3ull);
    const size_t dyn = table_bytes + (size_t)bits_dwords * 4 + (size_t)(SF_THREADS / 64) * (64 * per + 2) * 8;
    const uint64_t dense = rd->n_reads * (uint64_t)nwin;
    if (dyn <= 158 * 1024 && dyn <= c->lds_max && (uint64_t)rpt * nwin < 0x7FFFFFFFull) {
      if (dense > out->capacity) {
        if (total_out) *total_out = dense;
        return fail(NTHIP_ERR_CAPACITY, "output capacity %llu k-mers < %llu needed",
                    (unsigned long long)out->capacity, (unsigned long long)dense);
      }
      SeedFixedArgs a;
      memset(&a, 0, sizeof a);
      a.seqs = st.seqs;
      a.hashes = st.hashes;
      a.dirty = (uint32_t*)c->d_small;
      a.tables = sd->d_tables;
      a.n_runs = rd->n_reads;
      a.len = len;
      a.stride = stride;
      a.k = k;
      a.m2 = m2;
      a.n_seeds = sd->n_seeds;
      a.ntab = sd->ntab;
      a.nwin = nwin;
      a.runs_per_tile = rpt;
      const uint64_t n_tiles = (rd->n_reads + rpt - 1) / rpt;
      if (n_tiles > 0xFFFFFFFFull) return fail(NTHIP_ERR_UNSUPPORTED, "too many reads for one call");
      a.n_tiles = (uint32_t)n_tiles;
      a.inv_nwin = (uint32_t)((1ull << 32) / nwin + 1);
      a.bits_dwords = bits_dwords;
      for (uint32_t i = 0; i < (uint32_t)SF_MAX_RUNTIME_M; ++i) a.mult[i] = multiplier(k, i);
      HIPCHK(hipMemsetAsync(c->d_small, 0, 4, c->stream));
      int rc;
#define NT_SEED_FIXED(SPLIT_T, DYN) \
  (nh == 1   ? launch_seed_fixed(c, seed_fixed_kernel<1, SPLIT_T>, a, DYN) \
   : nh == 2 ? launch_seed_fixed(c, seed_fixed_kernel<2, SPLIT_T>, a, DYN) \
   : nh == 3 ? launch_seed_fixed(c, seed_fixed_kernel<3, SPLIT_T>, a, DYN) \
   : nh == 4 ? launch_seed_fixed(c, seed_fixed_kernel<4, SPLIT_T>, a, DYN) \
   : nh == 5 ? launch_seed_fixed(c, seed_fixed_kernel<5, SPLIT_T>, a, DYN) \
   : nh == 6 ? launch_seed_fixed(c, seed_fixed_kernel<6, SPLIT_T>, a, DYN) \
   : nh == 7 ? launch_seed_fixed(c, seed_fixed_kernel<7, SPLIT_T>, a, DYN) \
             : launch_seed_fixed(c, seed_fixed_kernel<8, SPLIT_T>, a, DYN))
      rc = NT_SEED_FIXED(false, dyn);
      NTCHK(rc);
      HIPCHK(hipMemcpyAsync(c->h_small, c->d_small, 4, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      uint32_t dirty = 0;
      memcpy(&dirty, c->h_small, 4);
      if (!dirty) {
        total = dense;
        if (st.counts) {
          hipLaunchKernelGGL(fill_u64_kernel, dim3(1024), dim3(256), 0, c->stream, st.counts, rd->n_reads,
                             (uint64_t)nwin);
          HIPCHK(hipGetLastError());
        }
        if (st.pos) { // every read emits every window: get_pos() is the window index
          hipLaunchKernelGGL(fill_window_pos_kernel, dim3(c->n_cu * 8), dim3(256), 0, c->stream, st.pos, rd->n_reads, nwin,
                             (const uint64_t*)nullptr, (const uint64_t*)nullptr);
          HIPCHK(hipGetLastError());
        }
        done = true;
      } else if (dyn + (size_t)rpt * 8 <= 158 * 1024 && dyn + (size_t)rpt * 8 <= c->lds_max) {
        // Batch with non-bases.  SeedNtHash's position state machine (App. B Q3) only matters for the reads
        // that HAVE a non-base: those (usually a fraction of a percent) go through seed_general_kernel, every
        // other read emits all its windows and stays on the fast kernel, writing at its place in the compact
        // stream (per-read counts -> scan -> offsets).
        const uint64_t n = rd->n_reads;
        const uint64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
        NTCHK(ensure_scratch(c, 5 * n + nb + 16));
        uint64_t* d_flags = c->d_scratch;
        uint64_t* d_idx = c->d_scratch + n;
        uint64_t* d_list = c->d_scratch + 2 * n;
        uint64_t* d_cnt = st.counts ? st.counts : c->d_scratch + 3 * n;
        uint64_t* d_roff = c->d_scratch + 4 * n;
        uint64_t* d_sums = c->d_scratch + 5 * n;
        uint64_t* d_total = (uint64_t*)(c->d_small + 8);
        const unsigned rblocks = (unsigned)((n + 255) / 256);
        HIPCHK(hipMemsetAsync(d_flags, 0, n * sizeof(uint64_t), c->stream));
        hipLaunchKernelGGL(seed_mark_dirty_kernel, dim3(c->n_cu * 8), dim3(256), 0, c->stream, st.seqs, total_bytes, len,
                           stride, n, d_flags);
        NTCHK(device_exclusive_scan(c, d_flags, d_idx, n, d_sums, d_total));
        HIPCHK(hipMemcpyAsync(c->h_small + 8, d_total, 8, hipMemcpyDeviceToHost, c->stream));
        hipLaunchKernelGGL(seed_list_kernel, dim3(rblocks), dim3(256), 0, c->stream, d_flags, d_idx, n, d_list);
        hipLaunchKernelGGL(fill_u64_kernel, dim3(1024), dim3(256), 0, c->stream, d_cnt, n, (uint64_t)nwin);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(c->stream));
        uint64_t n_dirty = 0;
        memcpy(&n_dirty, c->h_small + 8, 8);
        SeedGeneralArgs h;
        memset(&h, 0, sizeof h);
        h.seqs = st.seqs;
        h.read_list = d_list;
        h.n_reads = n_dirty;
        h.len = len;
        h.stride = stride;
        h.k = k;
        h.m2 = m2;
        h.n_seeds = sd->n_seeds;
        h.care_words = sd->care_words;
        h.care_bits = sd->d_care;
        h.blk_start = sd->d_blk_start;
        h.blk_count = sd->d_blk_count;
        h.blk_pairs = sd->d_blk_pairs;
        h.tables = sd->d_tables;
        h.ntab = sd->ntab;
        for (uint32_t i = 0; i < 256; ++i) h.mult[i] = multiplier(k, i);
        NTCHK(ensure_args(c, sizeof(SeedGeneralArgs)));
        const unsigned lblocks = (unsigned)((n_dirty + 255) / 256);
        SeedWavePlan wplan;
        const bool list_wave = !getenv("NTHIP_TUNE_NO_SEED_WAVE") && seed_wave_plan(c, sd, m2, &wplan);
        h.wave_lmax = SEED_WAVE_LMAX;
        if (n_dirty) {
          h.counts = d_cnt;
          h.wave_waves = wplan.waves_count;
          HIPCHK(hipMemcpyAsync(c->d_args, &h, sizeof h, hipMemcpyHostToDevice, c->stream));
          if (list_wave) {
            NTCHK(launch_seed_wave<true>(c, wplan, n_dirty));
          } else {
            hipLaunchKernelGGL(seed_general_kernel<true>, dim3(lblocks), dim3(256), 0, c->stream,
                               (const SeedGeneralArgs*)c->d_args);
            HIPCHK(hipGetLastError());
          }
        }
        NTCHK(device_exclusive_scan(c, d_cnt, d_roff, n, d_sums, d_total));
        HIPCHK(hipMemcpyAsync(c->h_small + 8, d_total, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        memcpy(&total, c->h_small + 8, 8);
        if (total > out->capacity) {
          if (total_out) *total_out = total;
          return fail(NTHIP_ERR_CAPACITY, "output capacity %llu k-mers < %llu needed",
                      (unsigned long long)out->capacity, (unsigned long long)total);
        }
        a.read_dirty = d_flags;
        a.read_off = d_roff;
        const size_t dyn2 = dyn + (size_t)rpt * 8;
        rc = NT_SEED_FIXED(true, dyn2);
#undef NT_SEED_FIXED
        NTCHK(rc);
        if (st.pos) {
          hipLaunchKernelGGL(fill_window_pos_kernel, dim3(c->n_cu * 8), dim3(256), 0, c->stream, st.pos, n, nwin,
                             (const uint64_t*)d_flags, (const uint64_t*)d_roff);
          HIPCHK(hipGetLastError());
        }
        if (n_dirty) {
          h.counts = nullptr;
          h.read_off = d_roff;
          h.hashes = st.hashes;
          h.pos = st.pos;
          h.capacity = out->capacity;
          h.wave_waves = wplan.waves_hash;
          HIPCHK(hipMemcpyAsync(c->d_args, &h, sizeof h, hipMemcpyHostToDevice, c->stream));
          if (list_wave) {
            NTCHK(launch_seed_wave<false>(c, wplan, n_dirty, /*record*/ false)); // kernel of record: seed_fixed_kernel
          } else {
            hipLaunchKernelGGL(seed_general_kernel<false>, dim3(lblocks), dim3(256), 0, c->stream,
                               (const SeedGeneralArgs*)c->d_args);
            HIPCHK(hipGetLastError());
          }
        }
        HIPCHK(hipStreamSynchronize(c->stream));
        done = true;
      }
    }
  }
  if (!done) NTCHK(run_seed_general(c, st, rd, sd, m2, out->capacity, &total));
  if (total_out) *total_out = total;
  NTCHK(unstage_outputs(c, out, flags, rd->n_reads, per, total, st, sd->n_seeds));
  HIPCHK(hipStreamSynchronize(c->stream));
  return NTHIP_OK;
}

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
  // byte tables in LDS (k > 64: the two Horner tables + a 2-bit stream of the wave's k-mers), 16-byte stores
  const uint32_t ntab = kmer_ntab(k);
  const size_t wave_bits = k > 64 ? ((((size_t)64 * k + 30) >> 4) + 4) * 4 : 0;
  const size_t lds_fixed = (size_t)ntab * 4096 + 16 * 2048; // tables + a 2 KiB exchange tile per wave
  const size_t lds_cap = (c->lds_max < 160 * 1024 ? c->lds_max : 160 * 1024) - 512;
  uint32_t waves = k > 32 && k <= 64 ? KX_WIDE_THREADS / 64 : 16;
  while (waves > 1 && lds_fixed + wave_bits * waves > lds_cap) waves /= 2;
  if (aligned16 && lds_fixed + wave_bits * waves <= lds_cap) {
    const uint4* tab = nullptr;
    if (get_kmer_tab(c, k, &tab) != NTHIP_OK) { cleanup(); return NTHIP_ERR_HIP; }
    const size_t lds = lds_fixed + wave_bits * waves;
    const uint32_t threads = waves * 64;
    uint64_t blocks = (n + threads - 1) / threads;
    if (blocks > (uint64_t)c->n_cu * 2) blocks = (uint64_t)c->n_cu * 2;
    auto go = [&](auto kernel) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      prof_begin(c, "kmer_extend_tab_kernel");
      hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(threads), lds, c->stream, d_in, n, k, m, tab, ntab, d_self,
                         d_next, d_prev);
      prof_end(c);
    };
    if (k > 64) go(kmer_extend_tab_kernel<0>);
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
// fused consumers of the hash stream: Bloom filter insert / query
// ==========================================================================
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
  if (rd->offsets) return fail(NTHIP_ERR_UNSUPPORTED, "fused consumers take fixed-length reads (offsets == NULL)");
  HIPCHK(hipSetDevice(c->device));
  if (total_out) *total_out = 0;
  if (rd->n_reads == 0) return NTHIP_OK;
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

extern "C" int nthip_kmer_minhash(nthip_ctx* c, const nthip_reads* rd, uint16_t k, uint8_t m, uint64_t* signatures,
                                  uint64_t* total, uint32_t flags)
{
  return run_kmer_minhash(c, rd, k, m, signatures, total, flags);
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

// ==========================================================================
// FASTQ / FASTA -> device batches: spans entry point, device indexer, streaming driver
// ==========================================================================
#include "fastx_stream.hpp"

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
