// capi_internal.hpp -- what the translation units of libnthash_hip.so share.
//
// The C-ABI (include/nthash_hip.h) is implemented by one .hip file per path:
//   capi_ctx.hip           context, error string, staging of host buffers, memory helpers, tuning knobs
//   capi_util.hip          device scan, synthetic reads, checksums, copy / fill yardsticks, nthip_kmer_extend
//   capi_kmer_plan.hip     per-k constants and tables, geometry plans of the run-split kernels (host only)
//   capi_kmer.hip          nthip_kmer_hash: path selection
//   capi_kmer_runs.hip     kmer_runs_kernel instantiations (headline shapes)
//   capi_kmer_gen.hip      kmer_runs_gen_kernel, dense
//   capi_kmer_na.hip       kmer_runs_gen_kernel, N-aware: count -> scan -> hash
//   capi_kmer_reads.hip    kmer_reads_kernel (offsets / spans in order, short reads): tiles of whole reads
//   capi_kmer_ragged.hip   kmer_ragged_kernel (offsets / spans, any lengths)
//   capi_kmer_general.hip  lane-per-read kernels (correctness paths) and the row-per-read kernel
//   capi_seed_extend.hip   nthip_seed_extend: the 4 successors / predecessors of n windows through spaced seeds
//   capi_seed.hip          spaced seeds: dense (seed_wtile / seed_fixed), variable-length (seed_rtile), one wave per read
//                          (seed_wave), long reads cut into pieces (seed_long_kernels.hpp), lane per read (seed_general)
//   capi_sink_bloom.hip / capi_sink_minhash.hip   fused consumers
//   capi_fastx.hip         FASTQ / FASTA indexing and the file streaming driver
//   capi_packed.hip        2-bit packed input: nthip_pack_reads, nthip_kmer_hash with NTHIP_PACKED_INPUT
//   capi_multi.hip         several devices of one node: shards of a host batch, one thread + context per device
//   capi_multi_sink.hip    ... device-resident shards, consumers per device, the merge of their tables over peer copies
// Kernels live in the *_kernel(s).hpp headers; every TU instantiates only the ones it launches.
// There is no CPU hashing path in any of them.
#pragma once

#include "../../include/nthash_hip.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <array>
#include <chrono>
#include <functional>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "kmer_kernels.hpp"
#include "kmer_runs_kernel.hpp"
#include "kmer_runs_gen_kernel.hpp"
#include "nt_math.hpp"
#include "seed_px_plan.hpp"

namespace ntamd {
namespace host {

// sets the thread-local error string returned by nthip_last_error() and returns `code`
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

} // namespace host
} // namespace ntamd

#define HIPCHK(call)                                                                                 \
  do {                                                                                               \
    hipError_t e_ = (call);                                                                          \
    if (e_ != hipSuccess)                                                                            \
      return ::ntamd::host::fail(NTHIP_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                                 __FILE__, __LINE__);                                                \
  } while (0)

#define NTCHK(call)                  \
  do {                               \
    int rc_ = (call);                \
    if (rc_ != NTHIP_OK) return rc_; \
  } while (0)

// A/B knobs (NTHIP_TUNE_* environment variables), read ONCE per context at nthip_ctx_create -- not on the hot
// calls -- and again by nthip_ctx_reload_tuning (the measurement tools change them between variants).
struct nthip_tune {
  uint32_t run_len = 0;     // NTHIP_TUNE_RUN_LEN: run length override (0 = the plan's choice)
  uint32_t run_max = 0;     // NTHIP_TUNE_RUN_MAX: longest run the cost model may pick
  uint32_t waves = 0;       // NTHIP_TUNE_WAVES: waves per block of the dense run-split kernels
  uint32_t na_waves = 0;    // NTHIP_TUNE_NA_WAVES: ... of the N-aware pass
  uint32_t seed_rpt = 0;    // NTHIP_TUNE_SEED_RPT: reads per tile of seed_fixed_kernel / seed_wtile_kernel
  uint32_t seed_waves = 0;  // NTHIP_TUNE_SEED_WAVES: waves per block of seed_wtile_kernel (4, 8, 12, 16)
  uint32_t read_threads = 0; // NTHIP_TUNE_READ_THREADS: pread threads of the file driver
  bool has_tile_map = false;
  uint32_t tile_map = 0;    // NTHIP_TUNE_TILE_MAP
  bool no_special = false;  // NTHIP_TUNE_NO_SPECIAL=1: general kernel on the k=31 shapes too
  bool no_dword_tail = false; // NTHIP_TUNE_NO_DWORD_TAIL=1
  bool no_m4 = false;       // NTHIP_TUNE_NO_M4 (set): runtime-m instantiation for m = 4
  bool no_autotune = false; // NTHIP_TUNE_NO_AUTOTUNE (set)
  bool no_na_special = false;   // NTHIP_TUNE_NO_NA_SPECIAL (set): batches with non-bases keep the general N-aware kernel for every tile
  bool no_dirty_memory = false; // NTHIP_TUNE_NO_DIRTY_MEMORY (set): every batch tries the dense pass first
  bool no_seed_wave = false; // NTHIP_TUNE_NO_SEED_WAVE (set)
  bool no_seed_wtile = false; // NTHIP_TUNE_NO_SEED_WTILE=1: the block-tile dense seed kernel instead of the wave-tile one
  bool mz_table = false;      // NTHIP_TUNE_MZ_TABLE=1: minimizers of clean short reads through the LDS tables too (A/B)
  uint32_t mz_fused = 0;      // NTHIP_TUNE_MZ_FUSED=2: never the one-pass minimizer kernel (minimizer_fused_kernel.hpp; A/B, tests)
  uint32_t mz_grid = 0;       // NTHIP_TUNE_MZ_GRID=<blocks>: the one-pass minimizer kernels on a PLAIN launch of that many blocks (tests: more than the device holds)
  uint32_t mz_timeout_us = 0; // NTHIP_TUNE_MZ_TIMEOUT_US: how long a look-back waits for a predecessor before the launch is given up (tests; 0: 50 ms)
  uint32_t mz_c = 0, mz_waves = 0; // NTHIP_TUNE_MZ_C / _MZ_WAVES: its run length and waves per block (0: planned)
  bool no_fh = false;         // NTHIP_TUNE_NO_FH=1: full position tables for k = 49 ... 64 (A/B)
  bool no_any_k_runs = false; // NTHIP_TUNE_NO_ANY_K_RUNS=1: only the k = 31 / run length 15, 30 instantiations of kmer_runs_kernel
  bool no_seed_align = false; // NTHIP_TUNE_NO_SEED_ALIGN=1: seed_rtile_kernel's groups end anywhere (A/B)
  bool no_seed_reads = false; // NTHIP_TUNE_NO_SEED_READS=1: variable-length reads of SeedNtHash on seed_wave_kernel only
  bool no_kmer_reads = false; // NTHIP_TUNE_NO_KMER_READS=1: variable-length reads on kmer_ragged_kernel only
  uint32_t reads_run_len = 0, reads_per_tile = 0, reads_waves = 0; // NTHIP_TUNE_READS_RUN_LEN / _PER_TILE / _WAVES (kmer_reads_kernel)
  bool no_seed_w6 = false;    // NTHIP_TUNE_NO_SEED_W6=1: seed_wtile_kernel with 4 waves where 6 would fit (A/B)
  bool no_seed_long = false;  // NTHIP_TUNE_NO_SEED_LONG=1: long reads of SeedNtHash stay on one wave per read
  uint32_t malloc_probe = 0;  // NTHIP_TUNE_MALLOC_PROBE=<n>: plain allocations nthip_malloc measures for a buffer of 1 GiB and more, the fastest kept (1 or unset: none -- one plain allocation)
  uint32_t seed_pass = 0;     // NTHIP_TUNE_SEED_PASS=n: seed_wtile_kernel hashes n seeds per pass (A/B; 0: planned)
  bool no_seed_rot = false;   // NTHIP_TUNE_NO_SEED_ROT=1: the plain [table][entry] layout of the byte tables in LDS
  // phased headline kernel (kmer_runs_kernel.hpp): tiles per wave and period, period / read window in 10 ns ticks
  bool no_phases = false;   // NTHIP_TUNE_NO_PHASES=1: the static loop (one tile ahead) instead of dynamic chunks
  bool no_pacing = false;   // NTHIP_TUNE_NO_PACING=1 (windowed builds): groups of tiles, but no waiting for the clock
  uint32_t ph_tiles = 0;    // NTHIP_TUNE_PH_TILES
  uint32_t ph_period = 0;   // NTHIP_TUNE_PH_PERIOD
  uint32_t ph_read = 0;     // NTHIP_TUNE_PH_READ
  // experiment (profiles/r03_notes.md 16): a second kernel reads the headline kernel's input ahead of it, in large sequential
  // chunks per tile group, paced by the clock -- NTHIP_TUNE_PF_GBPS = the input rate the main kernel is expected to consume
  // (GB/s; 0: off), NTHIP_TUNE_PF_LEAD_KB = how far ahead per group, NTHIP_TUNE_PF_CHUNK_KB = bytes per burst
  uint32_t pf_gbps = 0, pf_lead_kb = 0, pf_chunk_kb = 0;
  uint32_t bloom_round = 0;  // NTHIP_TUNE_BLOOM_ROUND=<values>: rounds of the binned consumers no longer than this (tests: several rounds on a small batch)
  uint32_t bloom_fused = 0;  // NTHIP_TUNE_BLOOM_FUSED=1: the stream-less binned insert on every shape it can take, 2: never (A/B, tests)
  bool no_tiles_flag = false; // NTHIP_TUNE_NO_TILES_FLAG=1: the count pass of run_kmer_na_special tile by tile (A/B, tests)
  uint32_t bloom_slots = 0;  // NTHIP_TUNE_BLOOM_SLOTS=1: slots mode (no histogram) whenever it applies, even after a failed round, 2: never (A/B, tests)
  uint32_t bloom_slot_tight = 0; // NTHIP_TUNE_BLOOM_SLOT_TIGHT=1: buckets of the mean exactly (the overflow list in use), 2: of half the mean (rounds fail) -- tests
  uint32_t bloom_binned = 0; // NTHIP_TUNE_BLOOM_BINNED=1: the binned insert whenever the filter allows it, 2: never (A/B, tests)
  uint32_t bloom_pieces = 0; // NTHIP_TUNE_BLOOM_PIECES=2: the two-level binned rounds on slots behind shared cursors, not on block-private pieces (A/B, tests)
  uint32_t bloom_query_passes = 0; // NTHIP_TUNE_BLOOM_QUERY_PASSES=1: the binned query of m > 1 one hash per pass whatever survives, 2: all m in one pass (A/B, tests)
  uint32_t bloom_query = 0;  // NTHIP_TUNE_BLOOM_QUERY=1: the binned query (bloom_query_kernels.hpp) on every batch it can take, 2: never (A/B, tests)
  uint32_t seed_roll_waves = 0; // NTHIP_TUNE_SEED_ROLL_WAVES=2..8: waves per block of seed_roll_kernel (A/B)
  uint32_t seed_roll = 0;   // NTHIP_TUNE_SEED_ROLL=1: every dense seed batch the block-rolling kernel takes goes there, 2: none (A/B, tests)
  uint32_t seed_px = 0;     // NTHIP_TUNE_SEED_PX=1: every dense seed batch seed_px_kernel takes goes there, 2: none (A/B, tests)
  uint32_t seed_px_array = 0; // NTHIP_TUNE_SEED_PX_ARRAY=1 + f: px_make_plan's force_array = f (tests: every array form against the oracle)
  uint32_t seed_px_reads = 0; // NTHIP_TUNE_SEED_PX_READS=R: reads per tile of seed_px_kernel (A/B; 0: planned)
  uint32_t seed_jit = 0;    // NTHIP_SEED_JIT=0: no kernel is compiled at run time (2 here), 1: for every batch the specialised kernel takes, compiled on the spot; unset: large batches, compiled on a thread of its own while the precompiled kernels go on hashing
  uint32_t seed_ps = 0;     // NTHIP_TUNE_SEED_PS=1: every dense seed batch seed_ps_kernel takes goes there, 2: none (A/B, tests)
  uint32_t seed_ps_lanes = 0; // NTHIP_TUNE_SEED_PS_LANES=16 / 32 / 64: window lanes per read of seed_ps_kernel (A/B; 0: planned)
  uint32_t seed_px_waves = 0; // NTHIP_TUNE_SEED_PX_WAVES=1..8: waves per block of seed_px_kernel (A/B; 0: planned)
  uint32_t seed_any = 0;    // NTHIP_TUNE_SEED_ANY=1: dense seed batches on the any-seed form whatever the seed set, 2: none of k <= 128 (A/B, tests)
  uint32_t fw = 0;          // NTHIP_TUNE_FW: first window beyond the position tables -- 1 grouped, 2 prefix scan (0: cost model)
};

struct nthip_ctx {
  int device = 0;
  int n_cu = 0;
  size_t lds_max = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t aux_stream = nullptr; // (the input-prefetch experiment)
  hipEvent_t aux_done = nullptr;
  hipStream_t stream = nullptr;
  // small device scratch: [0] dirty flag (u32), [8] total (u64), [16..32) sink totals
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
  // the binned Bloom insert's lists (capi_sink_bloom.hip): grow-only, released by nthip_ctx_trim / nthip_ctx_destroy
  uint8_t* bloom_tmp = nullptr;
  size_t bloom_tmp_bytes = 0;
  // the big buffers the consumers' rounds used to allocate and free per call -- [0] the hash stream of a round (reads by offsets,
  // spaced seeds), [1] the answers of a stream query: grow-only, released by nthip_ctx_trim / nthip_ctx_destroy.  (Round 5: a
  // process that frees tens of GB and allocates again waits for the driver -- seconds: kept_alloc, capi_util.hip)
  void* kept[2] = {nullptr, nullptr};
  size_t kept_bytes[2] = {0, 0};
  uint32_t bloom_slots_backoff = 0; // calls of the binned consumers that keep to the exact lists (a slots-mode round failed)
  bool profiling = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool ev_valid = false;
  const char* last_kernel = "";
  bool pos_listed_only = false; // (internal, nthip_kmer_minimizers) NTHIP_OUT_READ_SLOTS on fixed-length reads: positions of the redone reads only
  bool async_pending = false; // NTHIP_ASYNC launches since the last nthip_ctx_take_dirty: d_small[0] accumulates
  nthip_tune tune;
  // blocks per CU of (kernel, dynamic LDS) pairs already configured
  std::map<std::pair<const void*, size_t>, int> occ_cache;
  // all-care byte tables for the first window of a run, per k (device memory)
  std::map<uint32_t, uint4*> init_tabs;
  // what the consumers' rounds may hold in device scratch -- the kept buffers above, the lists of the binned consumers, the
  // temporaries of a round (nthip_ctx_set_scratch_limit; 0: half of the device's memory); rounds are sized to fit
  size_t scratch_limit = 0, device_mem = 0;
  // run length of the general dense kernel per (len, stride, k, m), measured on the first big batch of that shape
  // (the cost model does not see what a longer run costs in waves per CU or LDS conflicts: +-10 % either way)
  std::map<std::array<uint32_t, 4>, uint32_t> run_len_cache;
  // shapes whose last fixed-length batch held a non-base: the next batch of the shape starts on the N-aware passes (the
  // dense pass would only find out the same again); a batch that loses no window takes the shape off the list
  std::set<std::array<uint32_t, 4>> dirty_shapes;
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

// several devices of one node (capi_multi.hip, capi_multi_sink.hip): one context per listed device
struct nthip_multi {
  std::vector<nthip_ctx*> ctx;
};

struct nthip_seeds {
  nthip_ctx* ctx = nullptr; // the creating context: only compared, never dereferenced after creation (it may be gone)
  int device = 0;           // where the tables live (nthip_seeds_destroy frees them there)
  uint32_t n_seeds = 0, k = 0, ntab = 0, care_words = 0;
  bool asymmetric = false;
  uint4* d_tables = nullptr;      // [seed][ntab][256]
  uint32_t* d_care = nullptr;     // [seed][care_words]
  uint32_t* d_blk_start = nullptr;
  uint32_t* d_blk_count = nullptr;
  uint32_t* d_blk_pairs = nullptr;
  // the any-seed form of seed_wtile_kernel (NH == 0): per seed and 16-base group of the window, the care positions as a
  // 2-bit-per-base mask and the code-0 contribution of the others (first_window.hpp's tables)
  uint32_t any_groups = 0;
  uint32_t* d_any_mask = nullptr;
  uint4* d_any_acorr = nullptr;
  // nthip_seed_extend (capi_seed_extend.hip): which positions are covered by an odd number of blocks / are monomers in the
  // reference's description of every seed (get_blocks, src/seed.cpp:19-66), [n_seeds][k]; the three mask sets of the
  // kernel -- every contributing position, block positions, monomers -- as [3][n_seeds][any_groups], made on first use
  std::vector<uint8_t> h_blk_parity, h_is_mono;
  uint32_t* d_ext_mask = nullptr;
  uint4* d_ext_acorr = nullptr;
  // rolling run by run (seed_roll_kernel.hpp): every seed as the XOR of its care runs [a, b), each with a 16-entry (in, out)
  // pair table; roll_terms == 0: more runs than the kernel takes.  roll_in = b, roll_out = a.
  uint32_t roll_terms = 0, roll_first[65] = {}, roll_in[64] = {}, roll_out[64] = {};
  uint4* d_roll_tabs = nullptr;
  // sparse sums over scanned arrays (seed_px_plan.hpp / seed_px_kernel.hpp): the care positions of every seed, and the plan
  // made for the last read shape asked for (a plan weighs the arrays' cost per position against the reads per window)
  std::vector<std::vector<uint8_t>> h_care;
  mutable ntamd::PxPlan px_plan;
  mutable uint32_t px_plan_len = 0;
  mutable int px_plan_force = -2;
  // seed_ps_kernel.hpp: the same reads as byte offsets per step of a segment, on the device, for the last geometry asked for
  // the specialised kernels this seed set has met, (len, m2, waves, device) -> hipFunction_t: the registry of capi_seed_jit.hip is
  // keyed by the source TEXT, which a call should not have to write out again
  mutable std::map<uint64_t, std::pair<void*, uint32_t>> psj_ready; // (+ the waves per block it was compiled for)
  mutable uint32_t* d_ps_off = nullptr;
  mutable uint64_t ps_key = 0;
  mutable ntamd::PxPlan ps_plan;
  mutable uint32_t ps_plan_len = 0;
  mutable int ps_plan_force = -2;
};

namespace ntamd {
namespace host {

// ---- capi_seed_jit.hip: the kernel specialisation cache --------------------------------------------------------
struct SeedJitShape { // what seed_psj_kernel.inc is compiled for
  uint32_t len = 0, k = 0, nwin = 0, m2 = 0, n_seeds = 0, W = 0, nb_log = 0, lpr_log = 0, n_arrays = 0, segs_b = 0, waves = 0;
  std::vector<uint32_t> term_arr, term_e, seed_first;
};
std::string seed_psj_source(const SeedJitShape& g);
void* seed_psj_get(nthip_ctx* c, const nthip_seeds* sd, const SeedJitShape& g, bool wait, std::string* why); // hipFunction_t or nullptr
void seed_jit_release(const nthip_seeds* sd);
bool seed_jit_shape(nthip_ctx* c, const nthip_seeds* sd, uint32_t len, uint32_t m2, SeedJitShape* out); // capi_seed.hip

// ---- capi_ctx.hip -------------------------------------------------------------------------------------------
void load_tuning(nthip_tune& t);
int ensure_scratch(nthip_ctx* c, size_t elems);
int ensure_scratch2(nthip_ctx* c, size_t elems);
int ensure_args(nthip_ctx* c, size_t bytes);
void fastx_buffers_release(nthip_ctx* c); // the file driver's pinned / device buffers
// capi_util.hip: what nthip_malloc does -- hipMalloc
int default_alloc(nthip_ctx* c, size_t bytes, void** out);
// the context's scratch limit in bytes (its default: half of the device's memory)
size_t scratch_limit_of(nthip_ctx* c);
// the memory a consumer's round may plan with: what is free now + `reusable` (what the context already holds and the round
// reuses), capped by the scratch limit
size_t round_memory(nthip_ctx* c, size_t reusable, size_t fallback_free);

inline void prof_begin(nthip_ctx* c, const char* name)
{
  c->last_kernel = name;
  if (c->profiling) {
    (void)hipEventRecord(c->ev0, c->stream);
    c->ev_valid = false;
  }
}
inline void prof_end(nthip_ctx* c)
{
  if (c->profiling) {
    (void)hipEventRecord(c->ev1, c->stream);
    c->ev_valid = true;
  }
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

int stage_alloc(nthip_ctx* c, Staged& st, size_t bytes, void** p);
int stage_inputs(nthip_ctx* c, const nthip_reads* rd, uint32_t flags, uint64_t total_bytes, Staged& st);
int stage_outputs(nthip_ctx* c, const nthip_out* out, uint32_t flags, uint64_t n_reads, uint32_t per, Staged& st,
                  uint32_t strands_per = 1);
int unstage_outputs(nthip_ctx* c, const nthip_out* out, uint32_t flags, uint64_t n_reads, uint32_t per,
                    uint64_t total, const Staged& st, uint32_t strands_per = 1);
// total bytes of the read buffer (needs the last offset when offsets are on the device)
int reads_total_bytes(nthip_ctx* c, const nthip_reads* rd, uint32_t flags, uint64_t* out);
int check_reads(const nthip_reads* rd);
// offsets / spans sanity on the device (non-decreasing, inside the buffer): NTHIP_ERR_ARG instead of a wild read
// *max_len (optional): the longest span
int check_offsets_device(nthip_ctx* c, const uint64_t* d_starts, const uint64_t* d_ends, uint64_t n_reads,
                         uint64_t buf_bytes, bool contiguous, uint64_t* max_len = nullptr);
inline size_t lds_cap_of(const nthip_ctx* c) { return (c->lds_max < 160 * 1024 ? c->lds_max : 160 * 1024) - 512; }

// ---- capi_sink_bloom.hip: consumers on reads of any lengths ----
// device memory that `keep` frees (not the staging arena, which starts over in every staged call)
int own_alloc(Staged& keep, size_t bytes, void** p);
// *p = the context's kept buffer `slot` (KEPT_STREAM / KEPT_ANSWERS), at least `bytes` long; grown (freed and allocated anew: its
// contents are gone) when smaller.  NTHIP_ERR_HIP when the device does not have the memory (the slot is empty then).
enum : int { KEPT_STREAM = 0, KEPT_ANSWERS = 1 };
int kept_alloc(nthip_ctx* c, int slot, size_t bytes, void** p);
// what the context holds and a consumer's round may count as free memory: its lists and its kept buffers are reused, not added to
inline size_t reusable_bytes(const nthip_ctx* c) { return c->bloom_tmp_bytes + c->kept_bytes[0] + c->kept_bytes[1]; }
// the compact hash stream (m values per k-mer) of a batch given by offsets, hashed in ONE round into memory `keep` owns;
// d_counts (optional): per-read counts.  NTHIP_ERR_UNSUPPORTED when the stream does not fit the device
int stream_of_offsets(nthip_ctx* c, const nthip_reads* rd, uint16_t k, uint8_t m, uint32_t flags, Staged& keep, uint64_t** d_h,
                      uint64_t** d_counts, uint64_t* n_kmers, uint64_t round_bases = 0);
// A batch given by offsets, piece by piece (round 4: the consumers took it in ONE round and refused what did not fit): fn(part,
// r0, bases) for consecutive ranges of reads -- part: the range as a batch of its own (device-resident: the same seqs and
// offsets + r0; host: rebased), r0: its first read, bases: its bases -- sized so that scratch_per_base bytes of device scratch
// per base + 48 per read fit the free memory.  One piece when everything fits.
int offsets_in_rounds(nthip_ctx* c, const nthip_reads* rd, uint32_t flags, size_t scratch_per_base,
                      const std::function<int(const nthip_reads*, uint64_t, uint64_t)>& fn);

// ---- capi_sink_query.hip: the binned read side of the Bloom filter / counting sketch (bloom_query_kernels.hpp) ----
// Fixed-length device-resident reads [*first, n_reads) against the table (kind: BQ_BLOOM bits / BQ_COUNT one-byte counters),
// round by round while the rounds go through: *first is advanced past them, *kmers by the k-mers they emit, *hits_sum by
// their hits (BQ_BLOOM; d_hits[r], optional, per read -- device memory) or d_est[r * windows + w] written (BQ_COUNT).
// Stops early -- *first < n_reads, not an error -- when the shape, the table or the memory is not the binned query's, or a
// round's overflow list overflowed (skewed values): the caller takes the rest through its direct kernels.
int bloom_query_binned(nthip_ctx* c, const nthip_reads* rd, uint32_t k, uint32_t m, const uint32_t* d_table, uint64_t n_slots, int kind,
                       uint64_t* d_hits, uint8_t* d_est, uint64_t* first, uint64_t* kmers, uint64_t* hits_sum);

// The binned query of a hash STREAM (device memory, not inside the context's list buffer): d_ans[i] = the answer of value i
// (filter: its bit; sketch: its counter), region by region as above (slots mode: level 1 is the stream's partition).
// *done = false (nothing written): not a table / a stream for it, or skewed values -- the caller keeps its direct kernel.
// (expand_m = 2 ... 4: d_hashes holds hashes()[0] of n_values INPUTS and level 1 makes the other values of each -- d_ans[input *
//  expand_m + j], kmul = k * MULTISEED; two-level tables in pieces mode only, *done = false otherwise)
int stream_query_binned(nthip_ctx* c, const uint64_t* d_hashes, uint64_t n_values, const uint32_t* d_table, uint64_t n_slots, int kind,
                        uint8_t* d_ans, bool* done, uint32_t expand_m = 1, uint64_t kmul = 0);
// (its tests that need only the shapes -- table kind 0 filter / 1 sketch, alignment, sizes: asked before the kept answers are taken)
bool stream_query_applies(const nthip_ctx* c, const void* d_table, uint64_t n_slots, int kind, uint64_t n_values);
// the answers of a stream's values, k-mer by k-mer (m consecutive values each): filter: flags[i] = all set, *found their number;
// sketch: out[i] = the smallest.  hits per read (roff: first k-mer of every read): answers_hits_per_read.  Launches only.
int answers_per_kmer(nthip_ctx* c, const uint8_t* d_ans, uint64_t n_kmers, uint32_t m, int kind, uint8_t* d_out, unsigned long long* d_found);
// hits per read of a stream of k-mers' values against a filter: stream_query_binned + answers_hits_per_read when that applies,
// stream_bloom_query_kernel (a filter line per value) otherwise.  *d_total += the hits.  Launches, and waits when it held scratch.
int stream_hits_per_read(nthip_ctx* c, const uint64_t* d_h, const uint64_t* d_roff, uint64_t n_reads, uint64_t n_kmers, uint32_t m,
                         const uint32_t* d_filter, uint64_t n_bits, uint64_t* d_hits, unsigned long long* d_total, const char* direct_label,
                         uint32_t expand_m = 1, uint64_t kmul = 0, bool* expanded = nullptr);
// the stream insert / the binned stream query of a stream that holds hashes()[0] only: level 1 makes the other expand_m - 1 values of
// every input (bloom_part_stream_pieces_kernel<.., M>; expand_m 2 ... 4, kmul = k * MULTISEED).  *done = false: the caller hashes
// the full stream and takes the usual road.
bool bloom_binned_applies(const nthip_ctx* c, const void* d_filter, uint64_t n_bits, uint64_t n_values); // (the binned insert takes such a batch)
int stream_bloom_insert_expand(nthip_ctx* c, const uint64_t* d_h0, uint64_t n_inputs, uint32_t expand_m, uint64_t kmul, uint32_t* d_filter,
                               uint64_t n_bits, bool* done);
int answers_hits_per_read(nthip_ctx* c, const uint8_t* d_ans, const uint64_t* d_roff, uint64_t n_reads, uint64_t n_kmers, uint32_t m,
                          uint64_t* d_hits, unsigned long long* d_total_hits);

// ---- capi_util.hip ------------------------------------------------------------------------------------------
// exclusive scan of n u64 on the device: out[i] = sum(in[0..i)), *d_total = sum; d_sums: ceil(n/1024) + 16 u64
// (in-place is allowed: d_out == d_in)
int device_exclusive_scan(nthip_ctx* c, const uint64_t* d_in, uint64_t* d_out, uint64_t n, uint64_t* d_sums,
                          uint64_t* d_total);
int launch_fill_u64(nthip_ctx* c, uint64_t* d_dst, uint64_t n, uint64_t value);
// do the n reads of a device offsets array all have one length (offsets[r + 1] - offsets[r] == offsets[1] - offsets[0])?
struct OffsetsSurvey {
  uint64_t off0 = 0, len0 = 0, max_len = 0;
  bool uniform = false, bad = false;
};
int offsets_survey_device(nthip_ctx* c, const uint64_t* d_offsets, uint64_t n_reads, uint64_t buf_bytes, OffsetsSurvey* sv);
// longest read / largest distance between starts of a batch, when the caller already knows them
struct ReadsShape {
  uint64_t max_len = 0, max_pitch = 0, sum_len = 0;
};
// get_pos() of reads that emit every window (flags/offsets: only the reads with flags[r] == 0, at offsets[r])
int launch_fill_window_pos(nthip_ctx* c, uint32_t* d_pos, uint64_t n_reads, uint32_t nwin, const uint64_t* d_flags,
                           const uint64_t* d_offsets);

// ---- capi_kmer_plan.hip (host only) ---------------------------------------------------------------------------
#ifndef KMER_TABLE_K_MAX_N
#define KMER_TABLE_K_MAX_N 64
#endif
// largest k whose first window comes from position-specific byte tables (4 KiB per 4 bases of k in LDS); beyond it the
// k-independent first window of first_window.hpp.  A process-wide value: 64 unless NTHIP_TUNE_TABLE_K_MAX (16..64) says
// otherwise when a context is created (A/B: the k-independent forms on k <= 64 shapes)
extern uint32_t g_kmer_table_k_max;
#define KMER_TABLE_K_MAX (::ntamd::host::g_kmer_table_k_max)
inline uint32_t kmer_ntab(uint32_t k) { return k <= KMER_TABLE_K_MAX ? 4u * ((k + 15) / 16) : FW_ENTRIES / 256u; }
inline uint32_t kmer_nw(uint32_t k) { return k <= KMER_TABLE_K_MAX ? (k + 15) / 16 : 0u; }

void fill_kmer_consts(uint32_t k, uint32_t m, KmerFixedArgs& a);
bool kmer_fixed_eligible(const nthip_ctx* c, uint32_t len, uint32_t stride, uint32_t k, uint32_t m,
                         uint32_t* pad_dwords, size_t* dyn_lds);
void build_byte_tables(uint32_t k, const uint8_t* care, uint4* out);
int get_init_tab(nthip_ctx* c, uint32_t k, const uint4** out);
int get_kmer_tab(nthip_ctx* c, uint32_t k, const uint4** out);
int get_fw_tab(nthip_ctx* c, const uint4** out); // the k-independent first-window tables (first_window.hpp), FW_ENTRIES entries
// NTHIP_OUT_READ_SLOTS on fixed-length reads (capi_kmer_reads.hip): the dense pass marks the 16-byte vectors that hold a
// non-base in a bit map instead of giving up; afterwards the reads those vectors touch are redone in their slots.
struct FixedSlots {
  uint32_t* d_vecmap = nullptr; // what the dense kernels get (KmerRunsArgs::vecmap / KmerRunsGenArgs::vecmap)
  uint32_t* d_readmap = nullptr;
  uint64_t* d_list = nullptr;
  unsigned long long* d_count = nullptr;
  uint64_t n_words = 0;
};
bool kmer_fixed_slots_len_ok(uint32_t len);
int kmer_fixed_slots_begin(nthip_ctx* c, uint64_t n_reads, uint64_t total_bytes, FixedSlots* fs);
int kmer_fixed_slots_finish(nthip_ctx* c, const Staged& st, const FixedSlots& fs, uint64_t n_reads, uint32_t len, uint32_t stride,
                            uint32_t k, uint32_t m, uint64_t total_bytes, uint64_t* n_redone);

// Plan for the headline run-split kernel: run length C | nwin, waves per block, LDS bytes.
struct RunsPlan {
  uint32_t C = 0, rpr = 0, waves = 0, bits_dwords = 0, tile_u64 = 0, nw = 0, dword_tail = 0, ph_tiles = 0;
  size_t lds = 0;
};
bool kmer_runs_plan(const nthip_ctx* c, uint32_t len, uint32_t stride, uint32_t k, uint32_t m, RunsPlan* p);

struct GenPlan {
  uint32_t C = 0, rpr = 0, last_start = 0, waves = 0, bits_dwords = 0, tile_u64 = 0, nw = 0, dword_tail = 0;
  uint32_t fw_scan = 0, uw_dwords = 0; // nw == 0: the prefix-scan first window and its per-wave LDS (first_window.hpp)
  uint32_t fh = 0;                     // the forward-half tables of the dense pass (k = 49 ... 64: kmer_runs_gen_kernel FH)
  size_t lds = 0;
};
bool kmer_gen_plan(const nthip_ctx* c, uint32_t len, uint32_t stride, uint32_t k, uint32_t m, GenPlan* p,
                   bool gaps_ok = false, uint32_t force_c = 0, uint32_t model_cap = 0, bool no_tile = false);

struct NaPlan {
  GenPlan g;
  uint32_t vbits_dwords = 0, ptile_dwords = 0, waves = 0, tile_u64 = 0;
  size_t lds = 0;
};
bool kmer_na_plan(const nthip_ctx* c, uint32_t len, uint32_t stride, uint32_t k, uint32_t m, bool want_pos,
                  NaPlan* p, uint32_t register_sink_u64 = 0, uint32_t force_c = 0, uint32_t model_cap = 0);
void fill_gen_args(KmerRunsGenArgs& ga, nthip_ctx* c, const Staged& st, const nthip_reads* rd, uint32_t k,
                   uint32_t m, const GenPlan& g, const KmerFixedArgs& consts);

// ---- kernel launchers, one TU each ----------------------------------------------------------------------------
// capi_kmer_runs.hip: the k = 31 instantiations of kmer_runs_kernel (C = 15 | nwin, or 30 for m = 1)
bool kmer_runs_any_k_compiled(uint32_t k, uint32_t m, uint32_t C);
int launch_kmer_runs_special(nthip_ctx* c, const KmerRunsArgs& ra, const RunsPlan& plan, bool dword_tail);
// the packed-input instantiation of the headline shape (k = 31, m = 1, run length 15, dword-tail slabs)
int launch_kmer_runs_packed(nthip_ctx* c, const KmerRunsArgs& ra, const RunsPlan& plan);
// whether that unit was built with the chunked path (KR_CHUNKED; an A/B build of the unit may differ from the plan's)
bool kmer_runs_chunked_compiled();
// capi_kmer_gen.hip: kmer_runs_gen_kernel<NW, DT, false>
int launch_kmer_gen_dense(nthip_ctx* c, const KmerRunsGenArgs& ga, size_t lds, uint32_t nw, bool dt, bool packed = false);
// capi_kmer_na.hip
// invalid != nullptr: packed input (st.seqs = the code stream, invalid = its validity stream)
int run_kmer_na(nthip_ctx* c, const Staged& st, const nthip_reads* rd, uint32_t k, uint32_t m, const NaPlan& plan,
                const KmerFixedArgs& consts, uint64_t capacity, uint64_t* total, const uint16_t* invalid = nullptr);
// the same contract with the clean tiles on the specialised kernel (kmer_runs_kernel's burst path writing the compact
// stream at the count pass's offsets), the tiles that lost a window on the N-aware kernel from a list.  ra: the dense
// launch's arguments for the shape.  *handled = false: not a shape of that path, nothing done
int run_kmer_na_special(nthip_ctx* c, const Staged& st, const nthip_reads* rd, uint32_t k, uint32_t m, const RunsPlan& plan,
                        const KmerRunsArgs& ra, bool dt, const KmerFixedArgs& consts, uint64_t capacity, uint64_t* total,
                        bool* handled);
// capi_packed.hip: nthip_kmer_hash with NTHIP_PACKED_INPUT (st: the staged outputs)
int run_kmer_packed(nthip_ctx* c, const nthip_reads* rd, uint32_t k, uint32_t m, const nthip_out* out, const Staged& st,
                    uint32_t flags, uint64_t* total);
// capi_kmer_ragged.hip: reads = spans [starts[r], ends[r]) of the device buffer st.seqs (total_bytes long)
int run_kmer_reads(nthip_ctx* c, const Staged& st, const uint64_t* d_starts, const uint64_t* d_ends, uint64_t n_reads,
                   uint64_t total_bytes, uint32_t k, uint32_t m, uint64_t capacity, uint64_t* total, bool* handled,
                   const ReadsShape* shape, bool slots = false);
int run_kmer_ragged(nthip_ctx* c, const Staged& st, const uint64_t* d_starts, const uint64_t* d_ends, uint64_t n_reads,
                    uint64_t total_bytes, uint32_t k, uint32_t m, uint64_t capacity, uint64_t* total, bool* handled,
                    const ReadsShape* shape = nullptr, bool checked = true);
// capi_kmer_general.hip
int run_kmer_general(nthip_ctx* c, const Staged& st, const nthip_reads* rd, uint32_t k, uint32_t m,
                     uint64_t capacity, uint64_t* total);
int launch_kmer_rows(nthip_ctx* c, const KmerFixedArgs& a, size_t dyn_lds);
// capi_seed.hip
int run_seed_general(nthip_ctx* c, const Staged& st, const nthip_reads* rd, const nthip_seeds* sd, uint32_t m2,
                     uint64_t capacity, uint64_t* total, const uint64_t* d_ends = nullptr, bool fixed_as_spans = false);

// reads of >= 16 384 bases cut into independent pieces (seed_long_kernels.hpp); *handled = false: no such read, nothing done
int run_seed_long(nthip_ctx* c, const Staged& st, const nthip_reads* rd, const nthip_seeds* sd, uint32_t m2, uint64_t capacity,
                  uint64_t* total, bool* handled, const uint64_t* d_ends = nullptr);
constexpr uint64_t SEED_LONG_MIN = 16384; // reads from this length on are worth cutting into pieces

// ---- capi_fastx.hip (fastx_stream.hpp): the streaming pipeline of ONE context over caller-chosen pieces of a file ----
struct FastxRange {
  uint64_t off, len; // bytes [off, off + len): begins at a record start, ends where a record ends
};
// called with the index of the piece in the list and its batch (device pointers of the context; also for an empty piece)
typedef std::function<int(uint64_t, nthip_fastx_batch&)> FastxDeliver;
int fastx_stream_ranges(nthip_ctx* c, const char* path, uint32_t format, uint16_t k, uint8_t m, const nthip_seeds* seeds,
                        uint64_t chunk_bytes, const std::vector<FastxRange>& ranges, const FastxDeliver& deliver,
                        nthip_fastx_stats* stats);
int64_t fastx_find_record_start(int fd, uint64_t file_size, uint64_t pos, uint32_t format);
// does the file begin with the gzip magic (1f 8b)?  1 / 0, < 0: read error.  Such a file is inflated by one host thread
// and has no record boundaries to seek to: it feeds ONE device
int fastx_file_is_gzip(int fd);

// ---- templates every launching TU uses --------------------------------------------------------------------------
// Dynamic LDS beyond the default needs hipFuncAttributeMaxDynamicSharedMemorySize, and that attribute is ONE value per
// (device, function) for the whole process -- not per context, not per launch size.  So it is only ever RAISED, under a
// process-wide lock: a context that configures 100 KiB after another configured 150 KiB must not lower it under the
// other's (cached) launches (several contexts per device are normal: one per facade thread, nthip_multi listing a
// device twice).  capi_ctx.hip owns the table.
int raise_max_dynamic_lds(int device, const void* kernel, size_t bytes);

template <typename K>
int set_max_lds(const nthip_ctx* c, K kernel, size_t bytes)
{
  if (bytes <= 24 * 1024) return NTHIP_OK;
  return raise_max_dynamic_lds(c->device, reinterpret_cast<const void*>(kernel), bytes);
}

// A grid whose blocks wait for each other (block_rounds.hpp): every block must be resident.  hipLaunchCooperativeKernel
// promises that or refuses the launch -- *launched = false, no error: the caller takes a path that does not need it.
template <typename K, typename A>
int launch_resident(nthip_ctx* c, K kernel, unsigned grid, unsigned threads, size_t dyn_lds, A& args, bool cooperative, bool* launched)
{
  *launched = false;
  if (!cooperative) { // (tests: a plain launch, the grid whatever the knob says)
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), dyn_lds, c->stream, args);
    HIPCHK(hipGetLastError());
    *launched = true;
    return NTHIP_OK;
  }
  void* argv[1] = {(void*)&args};
  const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kernel), dim3(grid), dim3(threads), argv, (unsigned)dyn_lds, c->stream);
  if (e != hipSuccess) {
    (void)hipGetLastError(); // (too large for the device as it is right now, or no cooperative launches at all: not an error of the call)
    return NTHIP_OK;
  }
  *launched = true;
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
    NTCHK(set_max_lds(c, kernel, dyn_lds));
    int per_cu = 0;
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, dyn_lds));
    if (per_cu < 1) per_cu = 1;
    it = c->occ_cache.emplace(key, per_cu).first;
  }
  *out = it->second;
  return NTHIP_OK;
}

template <typename K>
int launch_kmer_runs_gen(nthip_ctx* c, K kernel, KmerRunsGenArgs a, size_t dyn_lds, const char* label)
{
  int per_cu = 1;
  NTCHK(blocks_per_cu(c, kernel, (int)a.waves * 64, dyn_lds, &per_cu));
  const uint64_t need = ((a.tile_list && !a.n_list_dev ? a.n_list : a.n_wtiles) + a.waves - 1) / a.waves;
  uint64_t grid = (uint64_t)c->n_cu * per_cu;
  if (grid > need) grid = need;
  if (a.tile_map == 0xFFFFFFFFu) a.tile_map = (uint32_t)grid;
  prof_begin(c, label);
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(a.waves * 64), dyn_lds, c->stream, a);
  prof_end(c);
  HIPCHK(hipGetLastError());
  return NTHIP_OK;
}

template <bool NA, int SINK = SINK_NONE, bool PK = false>
int launch_kmer_runs_gen_nw(nthip_ctx* c, const KmerRunsGenArgs& ga, size_t lds, uint32_t nw, bool dt)
{
  const char* label = SINK == SINK_BLOOM_INSERT  ? "kmer_runs_gen_kernel(bloom insert)"
                      : SINK == SINK_MINHASH     ? "kmer_runs_gen_kernel(minhash)"
                      : SINK == SINK_MINHASH1    ? "kmer_runs_gen_kernel(minhash, m = 1)"
                      : SINK == SINK_BLOOM_QUERY ? "kmer_runs_gen_kernel(bloom query)"
                      : NA                       ? (PK ? "kmer_runs_gen_kernel(N-aware, packed input)" : "kmer_runs_gen_kernel(N-aware)")
                      : PK                       ? "kmer_runs_gen_kernel(packed input)"
                                                 : "kmer_runs_gen_kernel";
#define NT_GEN(NWT) \
  (dt ? launch_kmer_runs_gen(c, kmer_runs_gen_kernel<NWT, true, NA, SINK, PK>, ga, lds, label) \
      : launch_kmer_runs_gen(c, kmer_runs_gen_kernel<NWT, false, NA, SINK, PK>, ga, lds, label))
  switch (nw) {
    case 0: return NT_GEN(0); // any k
    case 1: return NT_GEN(1);
    case 2: return NT_GEN(2);
    case 3: return NT_GEN(3);
    default: return NT_GEN(4);
  }
#undef NT_GEN
}

} // namespace host
} // namespace ntamd
