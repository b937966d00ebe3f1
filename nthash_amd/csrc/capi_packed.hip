// capi_packed.hip -- 2-bit packed input: nthip_pack_reads / nthip_packed_size, and nthip_kmer_hash with NTHIP_PACKED_INPUT
// Part of libnthash_hip.so (include/nthash_hip.h); see capi_internal.hpp for the file map.
//
// The reference converts bases to 2-bit codes on every call (CONVERT_TAB / RC_CONVERT_TAB, src/internal.hpp:350-418,
// feed its di- / tri- / tetramer tables); pipelines that hash the same reads at several k (ntCard, multi-k assembly) pay
// that -- and on this GPU the 1.25 bytes of ASCII per k-mer that are 13.5 % of the headline kernel's traffic -- once per
// k.  Packing is that step done ONCE: the batch's bytes as a 2-bit code stream (0.25 B per base; 16 bases per dword,
// the format the kernels stage a slab into anyway) plus a validity stream (1 bit per base) for the N-skipping rule.
#include "capi_internal.hpp"
#include "util_kernels.hpp"

using namespace ntamd;
using namespace ntamd::host;

namespace {

inline uint64_t packed_codes_bytes(uint64_t n_bases) { return ((n_bases + 63) / 64) * 16 + 16; }   // + one spare vector
inline uint64_t packed_invalid_bytes(uint64_t n_bases) { return ((n_bases + 127) / 128) * 16 + 16; }
// The last 8 bytes of the code stream's spare vector say how many bases were packed: the validity stream's place follows
// from that number, and a hash call that describes other reads (a subset, another stride) would read validity bits from
// the wrong place without anybody noticing -- now it is refused (NTHIP_ERR_ARG).  NTHIP_PACKED_CLEAN calls never read it.
constexpr uint64_t PACKED_MAGIC = 0x6e74504bull << 40; // "ntPK" above 40 bits of base count

// one thread per 16 bytes of the batch: a dword of codes ((c >> 1) & 3, as everywhere), 16 validity bits (1 = not a base)
__global__ __launch_bounds__(256) void pack_reads_kernel(const uint8_t* __restrict__ src, uint64_t n_bytes,
                                                         uint32_t* __restrict__ codes, uint16_t* __restrict__ invalid,
                                                         uint64_t n_vec_out, unsigned long long* __restrict__ n_invalid)
{
  uint32_t mine = 0;
  for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n_vec_out;
       v += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t off = v << 4;
    uint32_t w[4] = {0, 0, 0, 0};
    uint32_t have = 16;
    if (off + 16u <= n_bytes) {
      uint4 x;
      __builtin_memcpy(&x, src + off, 16); // (the caller's buffer starts anywhere)
      w[0] = x.x; w[1] = x.y; w[2] = x.z; w[3] = x.w;
    } else {
      have = off < n_bytes ? (uint32_t)(n_bytes - off) : 0u;
      for (uint32_t b = 0; b < have; ++b) w[b >> 2] |= (uint32_t)src[off + b] << ((b & 3u) * 8u);
    }
    uint32_t i0, i1, i2, i3;
    const uint32_t c0 = pack4v(w[0], i0), c1 = pack4v(w[1], i1), c2 = pack4v(w[2], i2), c3 = pack4v(w[3], i3);
    uint32_t inv = i0 | (i1 << 4) | (i2 << 8) | (i3 << 12);
    const uint32_t real = have >= 16u ? 0xFFFFu : (1u << have) - 1u;
    mine += (uint32_t)__builtin_popcount(inv & real);
    inv |= ~real & 0xFFFFu; // positions past the batch: never a base
    uint32_t cw = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
    if (have < 16u) cw &= have ? (1u << (2u * have)) - 1u : 0u;
    codes[v] = cw;
    invalid[v] = (uint16_t)inv;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) mine += (uint32_t)__shfl_xor((int)mine, d, 64);
  if ((threadIdx.x & 63u) == 0 && mine) atomicAdd(n_invalid, (unsigned long long)mine);
}

} // namespace

extern "C" int nthip_packed_size(uint64_t n_bases, size_t* total_bytes, size_t* invalid_offset)
{
  if (!total_bytes) return fail(NTHIP_ERR_ARG, "total_bytes is NULL");
  *total_bytes = (size_t)(packed_codes_bytes(n_bases) + packed_invalid_bytes(n_bases));
  if (invalid_offset) *invalid_offset = (size_t)packed_codes_bytes(n_bases);
  return NTHIP_OK;
}

extern "C" int nthip_pack_reads(nthip_ctx* c, const nthip_reads* rd, void* d_packed, uint64_t* n_invalid, uint32_t flags)
{
  if (!c) return fail(NTHIP_ERR_ARG, "ctx is NULL");
  NTCHK(check_reads(rd));
  if (!d_packed) return fail(NTHIP_ERR_ARG, "d_packed is NULL");
  if (((uintptr_t)d_packed & 15u) != 0) return fail(NTHIP_ERR_ARG, "d_packed must be 16-byte aligned");
  HIPCHK(hipSetDevice(c->device));
  if (n_invalid) *n_invalid = 0;
  uint64_t n_bytes = 0;
  if (rd->n_reads) NTCHK(reads_total_bytes(c, rd, flags, &n_bytes));
  Staged st;
  if (rd->n_reads) NTCHK(stage_inputs(c, rd, flags & NTHIP_HOST_INPUT, n_bytes, st));
  uint32_t* codes = (uint32_t*)d_packed;
  uint16_t* invalid = (uint16_t*)((uint8_t*)d_packed + packed_codes_bytes(n_bytes));
  const uint64_t n_vec_out = packed_codes_bytes(n_bytes) / 4 < packed_invalid_bytes(n_bytes) / 2
                                 ? packed_codes_bytes(n_bytes) / 4 : packed_invalid_bytes(n_bytes) / 2; // incl. the spare vectors
  unsigned long long* d_cnt = (unsigned long long*)(c->d_small + 8);
  HIPCHK(hipMemsetAsync(d_cnt, 0, 8, c->stream));
  uint64_t blocks = (n_vec_out + 255) / 256;
  if (blocks > (uint64_t)c->n_cu * 16) blocks = (uint64_t)c->n_cu * 16;
  prof_begin(c, "pack_reads_kernel");
  hipLaunchKernelGGL(pack_reads_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, st.seqs, n_bytes, codes, invalid,
                     n_vec_out, d_cnt);
  prof_end(c);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(c->h_small + 8, d_cnt, 8, hipMemcpyDeviceToHost, c->stream));
  const uint64_t stamp = PACKED_MAGIC | (n_bytes & ((1ull << 40) - 1));
  memcpy(c->h_small + 48, &stamp, 8);
  HIPCHK(hipMemcpyAsync((uint8_t*)d_packed + packed_codes_bytes(n_bytes) - 8, c->h_small + 48, 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (n_invalid) memcpy(n_invalid, c->h_small + 8, 8);
  return NTHIP_OK;
}

// nthip_kmer_hash, NTHIP_PACKED_INPUT: fixed-length reads whose bases are positions r * stride + j of the code stream
int ntamd::host::run_kmer_packed(nthip_ctx* c, const nthip_reads* rd, uint32_t k, uint32_t m, const nthip_out* out,
                                 const Staged& st_out, uint32_t flags, uint64_t* total)
{
  if (rd->offsets || rd->fixed_len == 0)
    return fail(NTHIP_ERR_UNSUPPORTED, "NTHIP_PACKED_INPUT takes fixed-length reads (fixed_len / stride)");
  if (flags & (NTHIP_HOST_INPUT | NTHIP_ASYNC | NTHIP_FORCE_GENERAL | NTHIP_FORCE_ROWS))
    return fail(NTHIP_ERR_UNSUPPORTED, "NTHIP_PACKED_INPUT: device-resident packed buffer, synchronous call");
  if (out->fwd || out->rev) return fail(NTHIP_ERR_UNSUPPORTED, "NTHIP_PACKED_INPUT: no strand outputs");
  if (((uintptr_t)rd->seqs & 15u) != 0) return fail(NTHIP_ERR_ARG, "the packed buffer must be 16-byte aligned");
  const uint32_t len = rd->fixed_len, stride = rd->stride ? rd->stride : len;
  const uint64_t n = rd->n_reads;
  *total = 0;
  if (len < k) {
    if (st_out.counts) {
      hipLaunchKernelGGL(fill_u64_kernel, dim3(1024), dim3(256), 0, c->stream, st_out.counts, n, 0ull);
      HIPCHK(hipGetLastError());
    }
    return NTHIP_OK;
  }
  const uint32_t nwin = len - k + 1;
  const uint64_t n_bases = (n - 1) * (uint64_t)stride + len;
  Staged st; // the views the launchers take: the code stream as `seqs`, the caller's (staged) outputs
  st.seqs = (const uint8_t*)rd->seqs;
  st.hashes = st_out.hashes;
  st.counts = st_out.counts;
  st.pos = st_out.pos;
  const uint16_t* invalid = (const uint16_t*)((const uint8_t*)rd->seqs + packed_codes_bytes(n_bases));
  KmerFixedArgs consts;
  memset(&consts, 0, sizeof consts);
  fill_kmer_consts(k, m, consts);
  const uint64_t dense = n * (uint64_t)nwin;
  if (flags & NTHIP_PACKED_CLEAN) {
    // the caller knows (nthip_pack_reads said so) that every byte is a base: every window is emitted
    if (dense > out->capacity) {
      *total = dense;
      return fail(NTHIP_ERR_CAPACITY, "output capacity %llu k-mers < %llu needed", (unsigned long long)out->capacity,
                  (unsigned long long)dense);
    }
    RunsPlan plan;
    GenPlan gplan;
    if (k == 31 && m == 1 && !c->tune.no_special && !kmer_runs_chunked_compiled() &&
        kmer_runs_plan(c, len, stride, k, m, &plan) && plan.C == 15 && plan.dword_tail && plan.rpr <= 128) {
      KmerRunsArgs ra;
      memset(&ra, 0, sizeof ra);
      ra.seqs = st.seqs;
      ra.hashes = st.hashes;
      ra.dirty = (uint32_t*)c->d_small;
      NTCHK(get_init_tab(c, k, &ra.init_tab));
      ra.n_reads = n;
      ra.n_runs = n * plan.rpr;
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
      ra.dword_tail = 1;
      ra.ph_tiles = plan.ph_tiles; // (burst path: pieces of 8 consecutive tiles, one contiguous run of the code stream)
      ra.tile_map = c->tune.has_tile_map ? c->tune.tile_map : 0xFFFFFFFFu;
      memcpy(ra.tab, consts.tab, sizeof ra.tab);
      memcpy(ra.mult, consts.mult, sizeof ra.mult);
      NTCHK(launch_kmer_runs_packed(c, ra, plan));
    } else if (kmer_gen_plan(c, len, stride, k, m, &gplan, /*gaps_ok*/ true)) {
      KmerRunsGenArgs ga;
      fill_gen_args(ga, c, st, rd, k, m, gplan, consts);
      NTCHK(get_kmer_tab(c, k, &ga.init_tab));
      NTCHK(launch_kmer_gen_dense(c, ga, gplan.lds, gplan.nw, gplan.dword_tail != 0, /*packed*/ true));
    } else {
      return fail(NTHIP_ERR_UNSUPPORTED, "NTHIP_PACKED_INPUT: shape outside the run-split kernels (reads overlapping by more than k - 1 bases)");
    }
    *total = dense;
    if (st.counts) {
      hipLaunchKernelGGL(fill_u64_kernel, dim3(1024), dim3(256), 0, c->stream, st.counts, n, (uint64_t)nwin);
      HIPCHK(hipGetLastError());
    }
    if (st.pos) {
      hipLaunchKernelGGL(fill_window_pos_kernel, dim3(c->n_cu * 8), dim3(256), 0, c->stream, st.pos, n, nwin,
                         (const uint64_t*)nullptr, (const uint64_t*)nullptr);
      HIPCHK(hipGetLastError());
    }
    return NTHIP_OK;
  }
  // batches that hold (or may hold) a non-base: count -> scan -> compact hash pass, validity from the companion stream --
  // whose place follows from the number of bases that were PACKED: it must be the number this call describes
  {
    uint64_t stamp = 0;
    HIPCHK(hipMemcpyAsync(c->h_small + 48, (const uint8_t*)rd->seqs + packed_codes_bytes(n_bases) - 8, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(&stamp, c->h_small + 48, 8);
    if (stamp != (PACKED_MAGIC | (n_bases & ((1ull << 40) - 1))))
      return fail(NTHIP_ERR_ARG, "NTHIP_PACKED_INPUT without NTHIP_PACKED_CLEAN: these reads (%llu bases) are not the batch nthip_pack_reads "
                  "packed into this buffer -- its validity stream lies elsewhere", (unsigned long long)n_bases);
  }
  NaPlan na;
  if (!kmer_na_plan(c, len, stride, k, m, st.pos != nullptr, &na))
    return fail(NTHIP_ERR_UNSUPPORTED, "NTHIP_PACKED_INPUT: shape outside the run-split kernels (reads overlapping by more than k - 1 bases)");
  return run_kmer_na(c, st, rd, k, m, na, consts, out->capacity, total, invalid);
}
