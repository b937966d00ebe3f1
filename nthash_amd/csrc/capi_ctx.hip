// capi_ctx.hip -- context, error string, staging of host buffers, memory helpers
// Part of libnthash_hip.so (include/nthash_hip.h); see capi_internal.hpp for the file map.
#include "capi_internal.hpp"

#include <mutex>

using namespace ntamd;
using namespace ntamd::host;

namespace {
thread_local std::string g_err;
} // namespace

int ntamd::host::fail(int code, const char* fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

namespace ntamd {
namespace host {
uint32_t g_kmer_table_k_max = KMER_TABLE_K_MAX_N;
// hipFuncAttributeMaxDynamicSharedMemorySize is one value per (device, function) for the process: raise-only, locked
int raise_max_dynamic_lds(int device, const void* kernel, size_t bytes)
{
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, size_t> set;
  std::lock_guard<std::mutex> lock(mu);
  size_t& have = set[std::make_pair(device, kernel)];
  if (bytes <= have) return NTHIP_OK;
  HIPCHK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  have = bytes;
  return NTHIP_OK;
}

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
} // namespace host
} // namespace ntamd

namespace ntamd {
namespace host {

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
                  Staged& st, uint32_t strands_per)
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
                    uint64_t total, const Staged& st, uint32_t strands_per)
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

} // namespace host
} // namespace ntamd

// ==========================================================================
// library / context
// ==========================================================================
extern "C" const char* nthip_version(void) { return "nthash_amd 0.3 (gfx950; ntHash_v2 bit-exact)"; }
extern "C" const char* nthip_last_error(void) { return g_err.c_str(); }

// the A/B knobs of the measurement tools; production runs have none of them set
void ntamd::host::load_tuning(nthip_tune& t)
{
  t = nthip_tune();
  auto num = [](const char* name, uint32_t lo, uint32_t hi) -> uint32_t {
    const char* v = getenv(name);
    if (!v) return 0u;
    const long x = atol(v);
    return x >= (long)lo && x <= (long)hi ? (uint32_t)x : 0u;
  };
  auto is_one = [](const char* name) { const char* v = getenv(name); return v && v[0] == '1'; };
  // (round 6: 29 knobs of concluded experiments -- wave counts, the phased / prefetching experiments of the headline kernel,
  //  tile maps, per-kernel on/off switches nobody measures any more -- are no longer read from the environment: their
  //  fields keep the defaults the measurements settled on.  DESIGN.md "tuning knobs" lists what is left and who sets it.)
  auto is_set = [](const char* name) { return getenv(name) != nullptr; };
  t.run_len = num("NTHIP_TUNE_RUN_LEN", 1, 64);
  t.run_max = num("NTHIP_TUNE_RUN_MAX", 1, 31);
  t.waves = num("NTHIP_TUNE_WAVES", 1, 16);
  t.no_special = is_one("NTHIP_TUNE_NO_SPECIAL");
  t.no_autotune = is_set("NTHIP_TUNE_NO_AUTOTUNE");
  t.no_dirty_memory = is_set("NTHIP_TUNE_NO_DIRTY_MEMORY");
  t.no_na_special = is_set("NTHIP_TUNE_NO_NA_SPECIAL");
  t.no_seed_wave = is_set("NTHIP_TUNE_NO_SEED_WAVE");
  t.no_seed_rot = is_one("NTHIP_TUNE_NO_SEED_ROT");
  t.seed_pass = num("NTHIP_TUNE_SEED_PASS", 1, 255);
  t.malloc_probe = num("NTHIP_TUNE_MALLOC_PROBE", 1, 8);
  t.no_seed_long = is_one("NTHIP_TUNE_NO_SEED_LONG");
  t.no_kmer_reads = is_one("NTHIP_TUNE_NO_KMER_READS");
  t.bloom_fused = num("NTHIP_TUNE_BLOOM_FUSED", 1, 2);
  t.mz_fused = num("NTHIP_TUNE_MZ_FUSED", 1, 2);
  t.mz_grid = num("NTHIP_TUNE_MZ_GRID", 1, 1 << 20);
  t.mz_timeout_us = num("NTHIP_TUNE_MZ_TIMEOUT_US", 1, 100000000);
  t.fw = num("NTHIP_TUNE_FW", 1, 2);
  t.seed_any = num("NTHIP_TUNE_SEED_ANY", 1, 2);
  t.seed_roll = num("NTHIP_TUNE_SEED_ROLL", 1, 2);
  t.seed_px = num("NTHIP_TUNE_SEED_PX", 1, 2);
  t.seed_ps = num("NTHIP_TUNE_SEED_PS", 1, 2);
  if (const char* v = getenv("NTHIP_SEED_JIT")) t.seed_jit = v[0] == '0' ? 2u : v[0] == '1' ? 1u : 0u; // (2: never)
  t.seed_ps_lanes = num("NTHIP_TUNE_SEED_PS_LANES", 16, 64);
  t.seed_px_array = num("NTHIP_TUNE_SEED_PX_ARRAY", 1, 300);
  t.seed_px_reads = num("NTHIP_TUNE_SEED_PX_READS", 1, 64);
  t.seed_px_waves = num("NTHIP_TUNE_SEED_PX_WAVES", 1, 8);
  t.bloom_binned = num("NTHIP_TUNE_BLOOM_BINNED", 1, 2);
  t.no_tiles_flag = is_one("NTHIP_TUNE_NO_TILES_FLAG");
  t.bloom_slots = num("NTHIP_TUNE_BLOOM_SLOTS", 1, 2);
  t.bloom_slot_tight = num("NTHIP_TUNE_BLOOM_SLOT_TIGHT", 1, 2);
  t.bloom_round = num("NTHIP_TUNE_BLOOM_ROUND", 1024, 0x7FFFFFFF);
  t.bloom_query = num("NTHIP_TUNE_BLOOM_QUERY", 1, 2);
  t.bloom_pieces = num("NTHIP_TUNE_BLOOM_PIECES", 1, 2);
  const uint32_t tk = num("NTHIP_TUNE_TABLE_K_MAX", 16, 64);
  ntamd::host::g_kmer_table_k_max = tk ? tk : (uint32_t)KMER_TABLE_K_MAX_N;
}
namespace {
} // namespace

extern "C" int nthip_ctx_reload_tuning(nthip_ctx* c)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  load_tuning(c->tune);
  c->run_len_cache.clear();
  return NTHIP_OK;
}

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
      hipMalloc((void**)&c->d_small, 256) != hipSuccess ||
      hipHostMalloc((void**)&c->h_small, 256) != hipSuccess ||
      hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
    delete c;
    return fail(NTHIP_ERR_HIP, "context resource creation failed: %s", hipGetErrorString(hipGetLastError()));
  }
  c->stream = c->own_stream;
  load_tuning(c->tune);
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
  if (c->bloom_tmp) (void)hipFree(c->bloom_tmp);
  for (int i = 0; i < 2; ++i)
    if (c->kept[i]) (void)hipFree(c->kept[i]);
  for (auto& kv : c->init_tabs) (void)hipFree(kv.second);
  fastx_buffers_release(c);
  if (c->stage_buf) (void)hipFree(c->stage_buf);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->aux_done) (void)hipEventDestroy(c->aux_done);
  if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
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
  if (c->bloom_tmp) (void)hipFree(c->bloom_tmp);
  c->bloom_tmp = nullptr;
  c->bloom_tmp_bytes = 0;
  for (int i = 0; i < 2; ++i) {
    if (c->kept[i]) (void)hipFree(c->kept[i]);
    c->kept[i] = nullptr;
    c->kept_bytes[i] = 0;
  }
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
  // Which pages hipMalloc hands out puts a big buffer into one of two classes -- five 15 GB allocations of one process: the
  // compact-stream pass of 20 M reads 4.14-4.21 ms into three of them, 4.68-4.70 into the other two, every time
  // (tools/var_alloc_spread.py) -- and a fill kernel tells them apart (2.16-2.44 against 2.67-2.68 ms).
  // NTHIP_TUNE_MALLOC_PROBE=<n> makes a buffer of 1 GiB and more the fastest of n plain allocations (held together, two fills
  // each, the others given back: nthip_malloc_probed).  NOT the default: the memory given back is not free at once -- a process
  // that releases tens of GB and allocates again waits for the driver, 2-6 s per 50 GB (tools/probe_cost.py, profiles/r05_notes.md
  // §12) -- so a caller who wants the fast class asks for it once, at start-up.  (Mapping from physical pieces: default_alloc.)
  const uint32_t probe = c->tune.malloc_probe ? c->tune.malloc_probe : 1u;
  if (probe > 1 && bytes >= ((size_t)1 << 30)) return nthip_malloc_probed(c, bytes, (int)probe, p, nullptr, nullptr);
  return default_alloc(c, bytes, p);
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

extern "C" int nthip_host_alloc(size_t bytes, void** hptr)
{
  if (!hptr) return fail(NTHIP_ERR_ARG, "hptr is NULL");
  *hptr = nullptr;
  if (bytes == 0) return NTHIP_OK;
  HIPCHK(hipHostMalloc(hptr, bytes, hipHostMallocPortable)); // (any device of the process copies to / from it)
  return NTHIP_OK;
}

extern "C" int nthip_host_free(void* hptr)
{
  if (hptr) HIPCHK(hipHostFree(hptr));
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
