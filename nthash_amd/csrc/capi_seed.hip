// capi_seed.hip -- spaced seeds: nthip_seeds_*, nthip_seed_hash
// Part of libnthash_hip.so (include/nthash_hip.h); see capi_internal.hpp for the file map.
#include "capi_internal.hpp"
#include "seed_long_kernels.hpp"
#include "seed_kernels.hpp"
#include "seed_roll_kernel.hpp"
#include "seed_px_kernel.hpp"
#include "seed_ps_kernel.hpp"
#include "kmer_reads_kernel.hpp" // the mark pass (reads with a non-base) is shared with the k-mer path
#include "seed_parse.hpp"
#include "util_kernels.hpp"

using namespace ntamd;
using namespace ntamd::host;

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
  std::vector<uint8_t> h_blk, h_mono;
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
    h_blk.insert(h_blk.end(), shape.blk_parity.begin(), shape.blk_parity.end());
    h_mono.insert(h_mono.end(), shape.is_mono.begin(), shape.is_mono.end());
    blk_start[s] = (uint32_t)(blk_pairs.size() / 2);
    blk_count[s] = (uint32_t)(pairs.size() / 2);
    blk_pairs.insert(blk_pairs.end(), pairs.begin(), pairs.end());
    for (uint32_t p = 0; p < k; ++p)
      if (par[p]) care[(size_t)s * cw + (p >> 5)] |= 1u << (p & 31);
    // byte tables: entry = XOR over the byte's 4 bases of the masked rotated seeds
    build_byte_tables(k, par.data(), tables.data() + (size_t)s * ntab * 256);
  }
  if (blk_pairs.empty()) blk_pairs.push_back(0);
  // the any-seed form: care masks and code-0 corrections per 16-base group (first_window.hpp: position u of a word
  // contributes sror^{u+1}(S[c]) / srol^{u}(S[~c]))
  const uint32_t G = (k + 15) / 16;
  std::vector<uint32_t> any_mask((size_t)n_seeds * G, 0);
  std::vector<uint4> any_acorr((size_t)n_seeds * G, make_uint4(0, 0, 0, 0));
  for (uint32_t s = 0; s < n_seeds; ++s)
    for (uint32_t g = 0; g < G; ++g) {
      uint64_t f = 0, r = 0;
      uint32_t mask = 0;
      for (uint32_t u = 0; u < 16; ++u) {
        const uint32_t p = 16 * g + u;
        if (p < k && ((care[(size_t)s * cw + (p >> 5)] >> (p & 31)) & 1u)) {
          mask |= 3u << (2 * u);
        } else {
          f ^= srol_n(seed_of_code(0), 1023u - (u + 1u));
          r ^= srol_n(seed_of_code(2), u);
        }
      }
      any_mask[(size_t)s * G + g] = mask;
      any_acorr[(size_t)s * G + g] = make_uint4((uint32_t)f, (uint32_t)(f >> 32), (uint32_t)r, (uint32_t)(r >> 32));
    }
  // block rolling: the care runs of every seed and their pair tables
  std::vector<uint4> roll_tabs;
  uint32_t roll_terms = 0, roll_first[SR_MAX_RUNS + 1] = {}, roll_in[SR_MAX_RUNS] = {}, roll_out[SR_MAX_RUNS] = {};
  if (n_seeds <= 8) {
    bool ok = true;
    for (uint32_t s = 0; s < n_seeds && ok; ++s) {
      auto is_care = [&](uint32_t p) { return ((care[(size_t)s * cw + (p >> 5)] >> (p & 31)) & 1u) != 0; };
      std::vector<uint32_t> terms; // the care runs as [a, b) pairs (runs alternate, so the description by gaps -- the whole
      for (uint32_t p = 0; p < k;) { // window XOR its don't-care runs -- never has fewer terms)
        uint32_t q = p;
        while (q < k && is_care(q) == is_care(p)) ++q;
        if (is_care(p)) {
          terms.push_back(p);
          terms.push_back(q);
        }
        p = q;
      }
      if (roll_terms + terms.size() / 2 > SR_MAX_RUNS) { ok = false; break; }
      for (size_t t = 0; t < terms.size(); t += 2) {
        const uint32_t a = terms[t], b = terms[t + 1];
        roll_in[roll_terms] = b;
        roll_out[roll_terms] = a;
        for (uint32_t e = 0; e < 16; ++e) {
          const uint32_t in = e >> 2, out = e & 3u;
          const uint64_t f = srol_n(seed_of_code(in), k - b) ^ srol_n(seed_of_code(out), k - a);
          const uint64_t r = srol_n(seed_of_code(in ^ 2u), b) ^ srol_n(seed_of_code(out ^ 2u), a);
          roll_tabs.push_back(make_uint4((uint32_t)f, (uint32_t)(f >> 32), (uint32_t)r, (uint32_t)(r >> 32)));
        }
        ++roll_terms;
      }
      roll_first[s + 1] = roll_terms;
    }
    if (!ok) roll_terms = 0;
  }
  nthip_seeds* sd = new nthip_seeds();
  sd->ctx = c;
  sd->device = c->device;
  sd->n_seeds = n_seeds;
  sd->k = k;
  sd->ntab = ntab;
  sd->care_words = cw;
  sd->asymmetric = asym;
  sd->h_blk_parity = h_blk;
  sd->h_is_mono = h_mono;
  sd->h_care.resize(n_seeds);
  for (uint32_t s = 0; s < n_seeds; ++s) {
    sd->h_care[s].resize(k);
    for (uint32_t p = 0; p < k; ++p) sd->h_care[s][p] = (uint8_t)((care[(size_t)s * cw + (p >> 5)] >> (p & 31)) & 1u);
  }
  auto up = [&](const void* src, size_t bytes, void** dst) -> int {
    HIPCHK(hipMalloc(dst, bytes));
    HIPCHK(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return NTHIP_OK;
  };
  int rc = up(tables.data(), tables.size() * sizeof(uint4), (void**)&sd->d_tables);
  if (rc == NTHIP_OK && roll_terms) {
    rc = up(roll_tabs.data(), roll_tabs.size() * sizeof(uint4), (void**)&sd->d_roll_tabs);
    sd->roll_terms = roll_terms;
    memcpy(sd->roll_first, roll_first, sizeof roll_first);
    memcpy(sd->roll_in, roll_in, sizeof roll_in);
    memcpy(sd->roll_out, roll_out, sizeof roll_out);
  }
  if (rc == NTHIP_OK) rc = up(care.data(), care.size() * 4, (void**)&sd->d_care);
  if (rc == NTHIP_OK) rc = up(blk_start.data(), blk_start.size() * 4, (void**)&sd->d_blk_start);
  if (rc == NTHIP_OK) rc = up(blk_count.data(), blk_count.size() * 4, (void**)&sd->d_blk_count);
  if (rc == NTHIP_OK) rc = up(blk_pairs.data(), blk_pairs.size() * 4, (void**)&sd->d_blk_pairs);
  sd->any_groups = G;
  if (rc == NTHIP_OK) rc = up(any_mask.data(), any_mask.size() * 4, (void**)&sd->d_any_mask);
  if (rc == NTHIP_OK) rc = up(any_acorr.data(), any_acorr.size() * sizeof(uint4), (void**)&sd->d_any_acorr);
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
  (void)hipSetDevice(sd->device); // (not sd->ctx->device: a seed set may outlive the context that made it)
  if (sd->d_tables) (void)hipFree(sd->d_tables);
  if (sd->d_care) (void)hipFree(sd->d_care);
  if (sd->d_blk_start) (void)hipFree(sd->d_blk_start);
  if (sd->d_blk_count) (void)hipFree(sd->d_blk_count);
  if (sd->d_blk_pairs) (void)hipFree(sd->d_blk_pairs);
  if (sd->d_any_mask) (void)hipFree(sd->d_any_mask);
  if (sd->d_any_acorr) (void)hipFree(sd->d_any_acorr);
  if (sd->d_ext_mask) (void)hipFree(sd->d_ext_mask);
  if (sd->d_ext_acorr) (void)hipFree(sd->d_ext_acorr);
  if (sd->d_roll_tabs) (void)hipFree(sd->d_roll_tabs);
  if (sd->d_ps_off) (void)hipFree(sd->d_ps_off);
  seed_jit_release(sd);
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

} // namespace

namespace {

// A seed set that needs several passes over the position tables is ONE pass in the any-seed form (whole records; k / 16
// table words per seed and window instead of one table byte per 8 bases).  Picoseconds per k-mer, fitted on
// tools/seed_sweep.py (profiles/r03_seed_any_sweep.txt, 14 multi-pass shapes, every one within 15 %): the passes are bound by
// their partial-record writes (1.25 TB/s: 6.4 ps per hash) or their table lookups (0.85 ps per seed and table), the any form
// by its arithmetic (2.1 ps per seed and group of 16 bases).  6 seeds of 31: 25.6 -> 38.1 G k-mers/s; 5 x 2: 15.1 -> 37.6;
// 2 x 2 of 80 bases: 38.2 -> 46.7; 4 seeds of 64 and 2 of 100 stay on the passes (37.2 / 30.5, 51.1 / 40.4)
bool seed_any_cheaper(uint32_t n_seeds, uint32_t m2, uint32_t k)
{
  const double per_h = (double)n_seeds * m2;
  const double t_lookups = 0.85 * n_seeds * ((k + 7) / 8);
  const double t_pass = 6.4 * per_h > t_lookups ? 6.4 * per_h : t_lookups;
  const double t_any = 2.1 * n_seeds * ((k + 15) / 16) + 0.5 * per_h;
  return t_any < t_pass;
}

// Variable-length short reads in order (offsets, or the sequence lines of a FASTQ chunk): the reads without a non-base
// on seed_rtile_kernel (tiles of whole reads), the others -- SeedNtHash's position state machine -- on seed_wave_kernel
// from a list, into the holes they left.  *handled = false: outside this path, nothing written.
int run_seed_reads(nthip_ctx* c, const Staged& st, const uint64_t* d_starts, const uint64_t* d_ends, uint64_t n,
                   const nthip_seeds* sd, uint32_t m2, uint64_t capacity, uint64_t* total, bool* handled,
                   SeedGeneralArgs& h, const SeedWavePlan& wplan)
{
  *handled = false;
  const uint32_t k = sd->k;
  if (c->tune.no_seed_reads || n == 0 || st.fwd || st.rev || m2 > (uint32_t)SF_MAX_RUNTIME_M) return NTHIP_OK;
  if (k > 64 && c->tune.seed_any == 2) return NTHIP_OK; // (A/B: the position tables end at 64 bases here)
  unsigned long long* d_res = (unsigned long long*)(c->d_small + 96);
  unsigned long long* d_ndirty = (unsigned long long*)(c->d_small + 128);
  HIPCHK(hipMemsetAsync(c->d_small + 96, 0, 48, c->stream));
  {
    uint64_t blocks = (n + READS_PREP_THREADS - 1) / READS_PREP_THREADS;
    if (blocks > (uint64_t)c->n_cu * 2) blocks = (uint64_t)c->n_cu * 2;
    hipLaunchKernelGGL(reads_prep_kernel, dim3((unsigned)blocks), dim3(READS_PREP_THREADS), 0, c->stream, d_starts, d_ends, n, ~0ull, d_res, 1u);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipMemcpyAsync(c->h_small + 96, d_res, 24, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  uint64_t res[3];
  memcpy(res, c->h_small + 96, 24);
  const uint64_t max_len = res[0], max_pitch = res[1] > res[0] ? res[1] : res[0];
  if (res[2] || max_len > RD_MAX_LEN || max_len < k) return NTHIP_OK;
  // ---- geometry ----
  const bool rot_ok = !c->tune.no_seed_rot && k <= 32 && m2 <= 4;
  const uint32_t nh_plain = (k + 7) / 8;
  const uint32_t per = sd->n_seeds * m2;
  uint32_t R = c->tune.seed_rpt ? (c->tune.seed_rpt < 64u ? c->tune.seed_rpt : 64u) : 16;
  const uint64_t slab_cap = 8192;
  while (R > 1 && (uint64_t)(R - 1) * max_pitch + max_len + 16 > slab_cap) --R;
  if ((uint64_t)(R - 1) * max_pitch + max_len + 16 > slab_cap) return NTHIP_OK;
  const uint32_t max_vec = (uint32_t)(((uint64_t)(R - 1) * max_pitch + max_len + 15 + 15) / 16 + 1);
  const uint32_t bits_dwords = (max_vec + (nh_plain + 1) / 2 + 8 + 3u) & ~3u;
  const uint32_t otile_recs = 64 + 16;
  const uint32_t max_win = R * (uint32_t)(max_len - k + 1);
  const uint32_t wmap_dwords = ((max_win / 16 + 8 + 3) / 4 + 3u) & ~3u;
  const uint32_t rt_n = R <= 16 ? 16 : R <= 32 ? 32 : 64;
  const size_t cap = lds_cap_of(c);
  // passes over the seeds, planned as in launch_seed_wtile: one pass writing whole records when the tables of all the
  // seeds fit in LDS beside 4 waves, otherwise as few passes as possible, each writing its part of every record
  auto per_wave_of = [&](uint32_t seeds_here) {
    return ((size_t)(otile_recs * seeds_here * m2 + 2) * 2 + bits_dwords + 4 * rt_n + wmap_dwords) * 4;
  };
  auto plain_bytes = [&](uint32_t seeds_here) { return (size_t)seeds_here * 2 * nh_plain * 256 * sizeof(uint4); };
  auto waves_for = [&](size_t tb, uint32_t seeds_here) -> uint32_t { // (as in launch_seed_wtile)
    for (uint32_t w = 16; w >= 4; w -= (w > 8 ? 4 : 1))
      if (tb + per_wave_of(seeds_here) * w <= cap) return w;
    return 0;
  };
  uint32_t pass_seeds;
  bool rot;
  if (c->tune.seed_pass) {
    pass_seeds = c->tune.seed_pass < sd->n_seeds ? c->tune.seed_pass : sd->n_seeds;
    rot = rot_ok && pass_seeds <= 2;
  } else if (rot_ok && sd->n_seeds <= 2) {
    pass_seeds = sd->n_seeds;
    rot = true;
  } else if (rot_ok && sd->n_seeds <= 4 && waves_for(2 * 65536, sd->n_seeds) != 0 &&
             (m2 == 1 || waves_for(plain_bytes(sd->n_seeds), sd->n_seeds) < 8)) {
    pass_seeds = sd->n_seeds; // (as in launch_seed_wtile: two table sets)
    rot = true;
  } else {
    rot = false;
    uint32_t most = sd->n_seeds;
    while (most > 1 && waves_for(plain_bytes(most), most) == 0) --most;
    const uint32_t passes = (sd->n_seeds + most - 1) / most;
    pass_seeds = (sd->n_seeds + passes - 1) / passes;
  }
  // the any-seed form (seed_rtile_kernel<0>): seeds beyond 64 bases, and seed sets of several table passes where the cost
  // model of launch_seed_wtile puts one pass of it ahead
  bool use_any = k > 64 || c->tune.seed_any == 1;
  if (!use_any && !rot && pass_seeds < sd->n_seeds && c->tune.seed_any != 2 && !c->tune.seed_pass)
    use_any = seed_any_cheaper(sd->n_seeds, m2, k);
  size_t any_bytes = 0;
  if (use_any) {
    const uint32_t n_grp = sd->n_seeds * sd->any_groups;
    any_bytes = ((size_t)SA_TAB_ENTRIES + n_grp + ((n_grp + 3) >> 2)) * sizeof(uint4);
    pass_seeds = sd->n_seeds;
    rot = false;
  }
  const uint32_t nh = use_any ? 0u : rot ? 4u : nh_plain;
  const size_t table_bytes = use_any ? any_bytes : rot ? (size_t)65536 * ((pass_seeds + 1) / 2) : plain_bytes(pass_seeds);
  uint32_t waves = waves_for(table_bytes, pass_seeds);
  if (use_any && !waves) // (many hashes per window: fewer than 4 waves still beat one wave per read)
    for (uint32_t w = 3; w >= 1 && !waves; --w)
      if (table_bytes + per_wave_of(pass_seeds) * w <= cap) waves = w;
  if (!waves) return NTHIP_OK;
  *handled = true;

  const uint64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  const uint64_t n_tiles = (n + R - 1) / R;
  NTCHK(ensure_scratch(c, 3 * n + n_tiles + nb + n / 8 + 64));
  uint64_t* d_cnt = st.counts ? st.counts : c->d_scratch;
  uint64_t* d_off = c->d_scratch + n;
  uint64_t* d_list = c->d_scratch + 2 * n;
  uint64_t* d_tsum = c->d_scratch + 3 * n; // (the mark pass writes per-tile sums; this path scans per-read counts)
  uint64_t* d_sums = d_tsum + n_tiles;
  uint8_t* d_flags = (uint8_t*)(d_sums + nb + 8);
  uint64_t* d_total = (uint64_t*)(c->d_small + 8);
  // ---- mark: reads with a non-base -> list; windows of the others ----
  {
    KmerReadsArgs ma;
    memset(&ma, 0, sizeof ma);
    ma.seqs = st.seqs;
    ma.starts = d_starts;
    ma.ends = d_ends;
    ma.n_reads = n;
    ma.R = R;
    ma.n_tiles = n_tiles;
    ma.cnt = d_cnt;
    ma.flags = d_flags;
    ma.dirty_list = d_list;
    ma.dirty_count = d_ndirty;
    ma.tile_sum = d_tsum;
    ma.k = k;
    ma.m = 1;
    ma.C = 8;
    ma.bits_dwords = ((max_vec + 4) / 2 + 3u) & ~3u;
    ma.waves = 16;
    const size_t mlds = ((size_t)ma.bits_dwords + 256) * 4 * ma.waves + 64;
    auto kernel = kmer_reads_kernel<RD_MODE_MARK, 1, false>;
    int per_cu = 1;
    NTCHK(blocks_per_cu(c, kernel, (int)ma.waves * 64, mlds, &per_cu));
    uint64_t grid = (uint64_t)c->n_cu * per_cu;
    const uint64_t need = (n_tiles + ma.waves - 1) / ma.waves;
    if (grid > need) grid = need;
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(ma.waves * 64), mlds, c->stream, ma);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipMemcpyAsync(c->h_small + 128, d_ndirty, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  uint64_t n_dirty = 0;
  memcpy(&n_dirty, c->h_small + 128, 8);
  // ---- the listed reads: exact counts (the reference's state machine) ----
  h.read_list = d_list;
  h.n_reads = n_dirty;
  h.counts = d_cnt;
  NTCHK(ensure_args(c, sizeof(SeedGeneralArgs)));
  const unsigned lblocks = (unsigned)((n_dirty + 255) / 256);
  const bool list_wave = wplan.waves_count != 0;
  if (n_dirty) {
    h.wave_waves = wplan.waves_count;
    HIPCHK(hipMemcpyAsync(c->d_args, &h, sizeof h, hipMemcpyHostToDevice, c->stream));
    if (list_wave) {
      NTCHK(launch_seed_wave<true>(c, wplan, n_dirty));
    } else {
      hipLaunchKernelGGL(seed_general_kernel<true>, dim3(lblocks), dim3(256), 0, c->stream, (const SeedGeneralArgs*)c->d_args);
      HIPCHK(hipGetLastError());
    }
  }
  NTCHK(device_exclusive_scan(c, d_cnt, d_off, n, d_sums, d_total));
  HIPCHK(hipMemcpyAsync(c->h_small + 8, d_total, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  memcpy(total, c->h_small + 8, 8);
  if (*total > capacity)
    return fail(NTHIP_ERR_CAPACITY, "output capacity %llu k-mers < %llu needed", (unsigned long long)capacity,
                (unsigned long long)*total);
  // ---- the clean reads ----
  SeedRtileArgs ra;
  memset(&ra, 0, sizeof ra);
  ra.seqs = st.seqs;
  ra.starts = d_starts;
  ra.ends = d_ends;
  ra.flags = d_flags;
  ra.read_off = d_off;
  ra.hashes = st.hashes;
  ra.pos = st.pos;
  ra.tables = sd->d_tables;
  ra.n_reads = n;
  ra.n_tiles = n_tiles;
  ra.R = R;
  ra.k = k;
  ra.m2 = m2;
  ra.ntab = sd->ntab;
  ra.bits_dwords = bits_dwords;
  ra.otile_recs = otile_recs;
  ra.wmap_dwords = wmap_dwords;
  ra.waves = waves;
  for (uint32_t i = 0; i < (uint32_t)SF_MAX_RUNTIME_M; ++i) ra.mult[i] = multiplier(k, i);
  const uint4* fw_tab = nullptr;
  if (use_any) {
    NTCHK(get_fw_tab(c, &fw_tab));
    ra.any_mask = sd->d_any_mask;
    ra.any_acorr = sd->d_any_acorr;
    ra.any_groups = sd->any_groups;
  }
  int rc = NTHIP_OK;
  for (uint32_t s0 = 0; s0 < sd->n_seeds && rc == NTHIP_OK; s0 += pass_seeds) {
    const uint32_t ns = sd->n_seeds - s0 < pass_seeds ? sd->n_seeds - s0 : pass_seeds;
    const uint32_t per_here = ns * m2;
    ra.n_seeds = ns;
    ra.tables = use_any ? fw_tab : sd->d_tables + (size_t)s0 * sd->ntab * 256;
    ra.pos = s0 == 0 ? st.pos : nullptr;
    if (ns == sd->n_seeds) {
      uint32_t g = per_here, h2 = 16;
      while (h2) { const uint32_t t2 = g % h2; g = h2; h2 = t2; } // gcd(per, 16)
      ra.align_recs = c->tune.no_seed_align ? 1u : 16u / g;
      ra.rec_stride = 0;
      ra.rec_off = 0;
    } else {
      ra.align_recs = 1;
      ra.rec_stride = per;
      ra.rec_off = s0 * m2;
    }
    const size_t lds = table_bytes + per_wave_of(ns) * waves;
    auto go = [&](auto kernel) -> int {
      int per_cu = 1;
      NTCHK(blocks_per_cu(c, kernel, (int)waves * 64, lds, &per_cu));
      const uint64_t need = (n_tiles + waves - 1) / waves;
      uint64_t grid = (uint64_t)c->n_cu * per_cu;
      if (grid > need) grid = need;
      if (s0 == 0) prof_begin(c, use_any ? "seed_rtile_kernel(any seed set)" : "seed_rtile_kernel"); // (all passes in one measurement)
      hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(waves * 64), lds, c->stream, ra);
      prof_end(c);
      HIPCHK(hipGetLastError());
      return NTHIP_OK;
    };
    if (rot) {
      switch (ns * 8 + m2) {
        case 8 + 1: rc = go(seed_rtile_kernel<4, 1, 1>); break;
        case 8 + 2: rc = go(seed_rtile_kernel<4, 1, 2>); break;
        case 8 + 3: rc = go(seed_rtile_kernel<4, 1, 3>); break;
        case 8 + 4: rc = go(seed_rtile_kernel<4, 1, 4>); break;
        case 16 + 1: rc = go(seed_rtile_kernel<4, 2, 1>); break;
        case 16 + 2: rc = go(seed_rtile_kernel<4, 2, 2>); break;
        case 16 + 3: rc = go(seed_rtile_kernel<4, 2, 3>); break;
        case 16 + 4: rc = go(seed_rtile_kernel<4, 2, 4>); break;
        case 24 + 1: rc = go(seed_rtile_kernel<4, 3, 1>); break;
        case 24 + 2: rc = go(seed_rtile_kernel<4, 3, 2>); break;
        case 24 + 3: rc = go(seed_rtile_kernel<4, 3, 3>); break;
        case 24 + 4: rc = go(seed_rtile_kernel<4, 3, 4>); break;
        case 32 + 1: rc = go(seed_rtile_kernel<4, 4, 1>); break;
        case 32 + 2: rc = go(seed_rtile_kernel<4, 4, 2>); break;
        case 32 + 3: rc = go(seed_rtile_kernel<4, 4, 3>); break;
        default: rc = go(seed_rtile_kernel<4, 4, 4>); break;
      }
    } else {
      switch (nh) {
        case 0: rc = go(seed_rtile_kernel<0>); break; // the any-seed form
        case 1: rc = go(seed_rtile_kernel<1>); break;
        case 2: rc = go(seed_rtile_kernel<2>); break;
        case 3: rc = go(seed_rtile_kernel<3>); break;
        case 4: rc = go(seed_rtile_kernel<4>); break;
        case 5: rc = go(seed_rtile_kernel<5>); break;
        case 6: rc = go(seed_rtile_kernel<6>); break;
        case 7: rc = go(seed_rtile_kernel<7>); break;
        default: rc = go(seed_rtile_kernel<8>); break;
      }
    }
  }
  NTCHK(rc);
  // ---- the listed reads, into their holes ----
  if (n_dirty) {
    h.counts = nullptr;
    h.read_off = d_off;
    h.hashes = st.hashes;
    h.pos = st.pos;
    h.capacity = capacity;
    h.wave_waves = wplan.waves_hash;
    HIPCHK(hipMemcpyAsync(c->d_args, &h, sizeof h, hipMemcpyHostToDevice, c->stream));
    if (list_wave) {
      NTCHK(launch_seed_wave<false>(c, wplan, n_dirty, /*record*/ false));
    } else {
      hipLaunchKernelGGL(seed_general_kernel<false>, dim3(lblocks), dim3(256), 0, c->stream, (const SeedGeneralArgs*)c->d_args);
      HIPCHK(hipGetLastError());
    }
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  return NTHIP_OK;
}

} // namespace

int ntamd::host::run_seed_general(nthip_ctx* c, const Staged& st, const nthip_reads* rd, const nthip_seeds* sd,
                     uint32_t m2, uint64_t capacity, uint64_t* total, const uint64_t* d_ends, bool fixed_as_spans)
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
  bool use_wave = !c->tune.no_seed_wave && seed_wave_plan(c, sd, m2, &wplan);
  h.wave_lmax = SEED_WAVE_LMAX;
  if (st.offsets && !c->tune.no_seed_wave) { // variable-length reads: the short-read path first (NTHIP_TUNE_NO_SEED_WAVE=1
                                             // keeps the lane-per-read kernel alone: the reference of the stress tools)
    bool handled = false;
    SeedGeneralArgs hl = h;
    SeedWavePlan lp = wplan;
    if (!use_wave) lp.waves_count = 0;
    NTCHK(run_seed_reads(c, st, st.offsets, d_ends ? d_ends : st.offsets + 1, n, sd, m2, capacity, total, &handled, hl, lp));
    if (handled) return NTHIP_OK;
  } else if (fixed_as_spans && !st.offsets && !c->tune.no_seed_wave && h.stride >= h.len && n != 0) {
    // reads of one length that left the dense kernel (a non-base somewhere in the batch and a shape the block-tile
    // kernel's split pass has no room for): the same path over spans made here -- the clean reads stay on tiles of whole
    // reads, only the reads with a non-base go one by one
    NTCHK(ensure_scratch2(c, 2 * n));
    uint64_t* d_s = c->d_scratch2;
    uint64_t* d_e = c->d_scratch2 + n;
    hipLaunchKernelGGL(fill_spans_kernel, dim3(c->n_cu * 8), dim3(256), 0, c->stream, d_s, d_e, n, (uint64_t)h.stride,
                       (uint64_t)h.len);
    HIPCHK(hipGetLastError());
    bool handled = false;
    SeedGeneralArgs hl = h;
    SeedWavePlan lp = wplan;
    if (!use_wave) lp.waves_count = 0;
    NTCHK(run_seed_reads(c, st, d_s, d_e, n, sd, m2, capacity, total, &handled, hl, lp));
    if (handled) return NTHIP_OK;
  }
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

namespace {

// seed_wtile_kernel: plan (reads per wave tile, LDS) + launch; *ran = false when the shape is outside it
int launch_seed_wtile(nthip_ctx* c, const SeedFixedArgs& f, const nthip_seeds* sd, uint32_t nh, bool* ran,
                      bool* prefer_any = nullptr)
{
  *ran = false;
  if (c->tune.no_seed_wtile) return NTHIP_OK;
  const uint32_t per = f.n_seeds * f.m2;
  const uint64_t rec_bytes = (uint64_t)f.nwin * per * 8; // a read's records
  // reads per tile: ~4 KB of reads, rounded up so that a tile's records are a whole number of KiB when the
  // slab stays under 8 KiB (SW_MAX_VEC_ROUNDS x 64 vectors)
  uint32_t R = 4096u / f.stride;
  if (R < 1) R = 1;
  if (c->tune.seed_rpt) {
    R = c->tune.seed_rpt; // A/B override
  } else {
    uint64_t g = rec_bytes, h = 1024;
    while (h) { const uint64_t t2 = g % h; g = h; h = t2; } // gcd(rec_bytes, 1024)
    const uint32_t unit = (uint32_t)(1024 / g);
    const uint32_t Ra = (R + unit - 1) / unit * unit;
    if (15ull + (uint64_t)(Ra - 1) * f.stride + f.len <= SW_MAX_VEC_ROUNDS * 1024ull) R = Ra;
  }
  const uint64_t slab = 15ull + (uint64_t)(R - 1) * f.stride + f.len;
  if (slab > SW_MAX_VEC_ROUNDS * 1024ull || (uint64_t)R * f.nwin >= 0x7FFFFFFFull) return NTHIP_OK;
  const uint32_t bits_dwords = (uint32_t)((((slab + 15) >> 4) + 8 + 3) & ~3ull);
  const size_t cap = lds_cap_of(c);
  // ---- passes.  The byte tables of a pass live in LDS: 8 KiB x nh per seed in the plain layout, 64 KiB for one or two
  // seeds of k <= 32 bases in the rotated-slot layout (no LDS bank conflicts; few hashes per seed).  A seed set that
  // fits whole, even with only 4 waves beside it, is ONE pass writing whole records; a larger one is hashed a few seeds
  // at a time, every pass writing its part of each record -- 8-byte pieces with gaps, which HBM takes at 0.9-1.4 TB/s
  // against 2.6-5 TB/s for whole records (tools/seed_sweep.py; 4 seeds x 2 hashes, k = 31: 40.7 G k-mers/s in one pass of
  // 4 waves, 21.4 G as two rotated-slot passes), so as few passes as possible. ----
  const bool rot_ok = !c->tune.no_seed_rot && f.k <= 32 && f.m2 <= 4;
  auto per_wave_of = [&](uint32_t seeds_here) { return (size_t)(64 * seeds_here * f.m2 + 2) * 8 + (size_t)bits_dwords * 4; };
  auto plain_bytes = [&](uint32_t seeds_here) { return (size_t)seeds_here * 2 * nh * 256 * sizeof(uint4); };
  auto waves_for = [&](size_t table_bytes, uint32_t seeds_here) -> uint32_t {
    // 16 / 12 / 8 waves, below that as many as fit (in process, 128 KiB of tables: 4 -> 5 -> 6 -> 7 waves 48.9 -> 53.6 ->
    // 61.4 -> 63.0 G k-mers/s at k = 64, 2 seeds x 3; between 8 and 12 the count does not matter).  The knob: any count (A/B)
    for (uint32_t w = c->tune.seed_waves ? c->tune.seed_waves : 16u; w >= 4; w -= (w > 8 ? 4 : 1)) {
      if (table_bytes + per_wave_of(seeds_here) * w <= cap) return w;
      if (c->tune.no_seed_w6 && w <= 8 && w > 4) w = 5; // (A/B: straight from 8 to 4)
    }
    return 0;
  };
  uint32_t pass_seeds; // seeds per pass
  bool rot;
  if (c->tune.seed_pass) {
    pass_seeds = c->tune.seed_pass < f.n_seeds ? c->tune.seed_pass : f.n_seeds;
    rot = rot_ok && pass_seeds <= 2;
  } else if (rot_ok && f.n_seeds <= 2) {
    pass_seeds = f.n_seeds;
    rot = true;
  } else if (rot_ok && f.n_seeds <= 4 && waves_for(2 * 65536, f.n_seeds) != 0 &&
             (f.m2 == 1 || waves_for(plain_bytes(f.n_seeds), f.n_seeds) < 8)) {
    // three or four seeds: two table sets of 64 KiB (seeds {0, 1} and {2, 3}), 4 waves beside them.  Pays where the
    // plain layout is LDS-bound (one hash per seed: 3 seeds 111 -> 144 G k-mers/s, 4 seeds 80 -> 117) or has no more
    // waves itself (4 seeds x 2: 42 -> 60 G); with 8+ waves beside plain tables and several hashes per seed the plain
    // layout is ahead (3 x 3: 64 against 62 G; 3 x 2, k = 24: 93 against 81)
    pass_seeds = f.n_seeds;
    rot = true;
  } else {
    rot = false;
    uint32_t most = f.n_seeds; // the most seeds whose tables fit beside 4 waves
    while (most > 1 && waves_for(plain_bytes(most), most) == 0) --most;
    const uint32_t passes = (f.n_seeds + most - 1) / most;
    pass_seeds = (f.n_seeds + passes - 1) / passes;
    if (passes > 1 && prefer_any && seed_any_cheaper(f.n_seeds, f.m2, f.k)) { // (one pass of the any-seed form instead)
      *prefer_any = true;
      return NTHIP_OK;
    }
  }
  const size_t table_bytes = rot ? (size_t)65536 * ((pass_seeds + 1) / 2) : (size_t)pass_seeds * 2 * nh * 256 * sizeof(uint4);
  const uint32_t waves = waves_for(table_bytes, pass_seeds);
  if (!waves) return NTHIP_OK;
  SeedWtileArgs a;
  memset(&a, 0, sizeof a);
  a.seqs = f.seqs;
  a.hashes = f.hashes;
  a.dirty = f.dirty;
  a.n_reads = f.n_runs;
  a.n_tiles = (f.n_runs + R - 1) / R;
  a.len = f.len;
  a.stride = f.stride;
  a.k = f.k;
  a.m2 = f.m2;
  a.ntab = f.ntab;
  a.nwin = f.nwin;
  a.reads_per_tile = R;
  a.inv_nwin = f.inv_nwin;
  a.bits_dwords = bits_dwords;
  a.waves = waves;
  a.groups = c->tune.has_tile_map ? c->tune.tile_map : 32u; // (in process: 32 / 64 groups 0.2-0.4 % ahead of one range per block)
  memcpy(a.mult, f.mult, sizeof a.mult);
  for (uint32_t s0 = 0; s0 < f.n_seeds; s0 += pass_seeds) {
    const uint32_t ns = f.n_seeds - s0 < pass_seeds ? f.n_seeds - s0 : pass_seeds;
    const uint32_t per_here = ns * f.m2;
    a.n_seeds = ns;
    a.tables = f.tables + (size_t)s0 * f.ntab * 256;
    if (ns == f.n_seeds) { // whole records
      a.rec_stride = 0;
      a.rec_off = 0;
      uint32_t g = per_here, h = 16;
      while (h) { const uint32_t t2 = g % h; g = h; h = t2; } // gcd(per, 16)
      a.align_recs = c->tune.no_seed_align ? 1u : 16u / g;
    } else {
      a.rec_stride = per;
      a.rec_off = s0 * f.m2;
      a.align_recs = 1;
    }
    const size_t lds = table_bytes + per_wave_of(ns) * waves;
    auto go = [&](auto kernel) -> int {
      int per_cu = 1;
      NTCHK(blocks_per_cu(c, kernel, (int)waves * 64, lds, &per_cu));
      const uint64_t need = (a.n_tiles + waves - 1) / waves;
      uint64_t grid = (uint64_t)c->n_cu * per_cu;
      if (grid > need) grid = need;
      if (s0 == 0) prof_begin(c, "seed_wtile_kernel"); // (all passes in one measurement)
      hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(waves * 64), lds, c->stream, a);
      prof_end(c);
      HIPCHK(hipGetLastError());
      return NTHIP_OK;
    };
    int rc;
    const bool sub = a.rec_stride != 0;
#define NT_SW(...) (sub ? go(seed_wtile_kernel<__VA_ARGS__, true>) : go(seed_wtile_kernel<__VA_ARGS__, false>))
    if (rot) {
      switch (ns * 8 + f.m2) {
        case 8 + 1: rc = NT_SW(4, 1, 1); break;
        case 8 + 2: rc = NT_SW(4, 1, 2); break;
        case 8 + 3: rc = NT_SW(4, 1, 3); break;
        case 8 + 4: rc = NT_SW(4, 1, 4); break;
        case 16 + 1: rc = NT_SW(4, 2, 1); break;
        case 16 + 2: rc = NT_SW(4, 2, 2); break;
        case 16 + 3: rc = NT_SW(4, 2, 3); break;
        case 16 + 4: rc = NT_SW(4, 2, 4); break;
        case 24 + 1: rc = go(seed_wtile_kernel<4, 3, 1>); break; // (three / four seeds: always the whole seed set)
        case 24 + 2: rc = go(seed_wtile_kernel<4, 3, 2>); break;
        case 24 + 3: rc = go(seed_wtile_kernel<4, 3, 3>); break;
        case 24 + 4: rc = go(seed_wtile_kernel<4, 3, 4>); break;
        case 32 + 1: rc = go(seed_wtile_kernel<4, 4, 1>); break;
        case 32 + 2: rc = go(seed_wtile_kernel<4, 4, 2>); break;
        case 32 + 3: rc = go(seed_wtile_kernel<4, 4, 3>); break;
        default: rc = go(seed_wtile_kernel<4, 4, 4>); break;
      }
    } else {
      switch (nh) {
        case 1: rc = NT_SW(1, 0, 0); break;
        case 2: rc = NT_SW(2, 0, 0); break;
        case 3: rc = NT_SW(3, 0, 0); break;
        case 4: rc = NT_SW(4, 0, 0); break;
        case 5: rc = NT_SW(5, 0, 0); break;
        case 6: rc = NT_SW(6, 0, 0); break;
        case 7: rc = NT_SW(7, 0, 0); break;
        case 8: rc = NT_SW(8, 0, 0); break;
        case 10: rc = NT_SW(10, 0, 0); break; // seeds of 65..128 bases
        case 12: rc = NT_SW(12, 0, 0); break;
        case 14: rc = NT_SW(14, 0, 0); break;
        default: rc = NT_SW(16, 0, 0); break;
      }
    }
#undef NT_SW
    NTCHK(rc);
  }
  *ran = true;
  return NTHIP_OK;
}

// the any-seed form (seed_wtile_kernel<0>): ONE pass over all the seeds whatever their number and length
int launch_seed_any(nthip_ctx* c, const SeedFixedArgs& f, const nthip_seeds* sd, bool* ran)
{
  *ran = false;
  const uint32_t per = f.n_seeds * f.m2;
  uint32_t R = 4096u / f.stride;
  if (R < 1) R = 1;
  const uint64_t slab = 15ull + (uint64_t)(R - 1) * f.stride + f.len;
  if (slab > SW_MAX_VEC_ROUNDS * 1024ull || (uint64_t)R * f.nwin >= 0x7FFFFFFFull) return NTHIP_OK;
  const uint32_t bits_dwords = (uint32_t)((((slab + 15) >> 4) + 8 + 3) & ~3ull);
  const uint32_t n_grp = f.n_seeds * sd->any_groups;
  const size_t table_bytes = ((size_t)SA_TAB_ENTRIES + n_grp + ((n_grp + 3) >> 2)) * sizeof(uint4);
  const size_t per_wave = (size_t)(64 * per + 2) * 8 + (size_t)bits_dwords * 4;
  const size_t cap = lds_cap_of(c);
  uint32_t waves = 0;
  for (uint32_t w = 16; w >= 1; --w)
    if (table_bytes + per_wave * w <= cap) { waves = w; break; }
  if (!waves) return NTHIP_OK;
  const uint4* fw = nullptr;
  NTCHK(get_fw_tab(c, &fw));
  SeedWtileArgs a;
  memset(&a, 0, sizeof a);
  a.seqs = f.seqs;
  a.hashes = f.hashes;
  a.dirty = f.dirty;
  a.tables = fw;
  a.n_reads = f.n_runs;
  a.n_tiles = (f.n_runs + R - 1) / R;
  a.len = f.len;
  a.stride = f.stride;
  a.k = f.k;
  a.m2 = f.m2;
  a.n_seeds = f.n_seeds;
  a.ntab = f.ntab;
  a.nwin = f.nwin;
  a.reads_per_tile = R;
  a.inv_nwin = f.inv_nwin;
  a.bits_dwords = bits_dwords;
  a.waves = waves;
  a.groups = c->tune.has_tile_map ? c->tune.tile_map : 32u;
  uint32_t g = per, h = 16;
  while (h) { const uint32_t t2 = g % h; g = h; h = t2; } // gcd(per, 16)
  a.align_recs = c->tune.no_seed_align ? 1u : 16u / g;
  a.any_mask = sd->d_any_mask;
  a.any_acorr = sd->d_any_acorr;
  a.any_groups = sd->any_groups;
  memcpy(a.mult, f.mult, sizeof a.mult);
  const size_t lds = table_bytes + per_wave * waves;
  auto kernel = seed_wtile_kernel<0, 0, 0, false>;
  int per_cu = 1;
  NTCHK(blocks_per_cu(c, kernel, (int)waves * 64, lds, &per_cu));
  const uint64_t need = (a.n_tiles + waves - 1) / waves;
  uint64_t grid = (uint64_t)c->n_cu * per_cu;
  if (grid > need) grid = need;
  prof_begin(c, "seed_wtile_kernel(any seed set)");
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(waves * 64), lds, c->stream, a);
  prof_end(c);
  HIPCHK(hipGetLastError());
  *ran = true;
  return NTHIP_OK;
}

// Seeds rolled run by run (seed_roll_kernel.hpp).  The cost model, per window in units of one run's roll (fitted to
// tools/seed_roll_sweep.py, profiles/r04_seed_roll_sweep.txt): rolling pays 114 + 6.7 per 16 bases of k once per segment and
// seed (the any-seed first window, the tile's bases and bookkeeping), 3.3 per seed and window, 1 per run, 5 per hash after a
// seed's first, and 30 % on top when a window has several values (the stage takes the place of waves); the position-table
// form pays 1.34 per byte table and seed, the any-seed form 6.2 per 16 bases.  *ran = false: not this kernel's shape, or
// the direct form is the cheaper one.
int launch_seed_roll(nthip_ctx* c, const SeedFixedArgs& f, const nthip_seeds* sd, bool* ran)
{
  *ran = false;
  const uint32_t per = f.n_seeds * f.m2;
  if (!sd->roll_terms || c->tune.seed_roll == 2 || c->tune.seed_any == 1 || per > 8 || f.stride != f.len) return NTHIP_OK;
  const uint32_t SW = per <= 2 ? 16u : per <= 4 ? 8u : 4u;
  const uint32_t segs = (f.nwin + SW - 1) / SW, G = sd->any_groups;
  if (c->tune.seed_roll != 1) {
    const double sw_eff = (double)f.nwin / segs, extra = per - f.n_seeds;
    const double roll = (f.n_seeds * ((114.0 + 6.7 * G) / sw_eff + 3.3) + sd->roll_terms + 5.0 * extra) * (per > 1 ? 1.3 : 1.0);
    const uint32_t tables = f.k <= 64 ? 2 * ((f.k + 7) / 8) : 4 * ((f.k + 15) / 16);
    const double direct = (f.k <= 128 ? 1.34 * tables : 6.2 * G) * f.n_seeds + 4.5 * extra;
    if (roll >= direct) return NTHIP_OK;
  }
  const uint64_t slab_max = 64ull * SW + (uint64_t)(63u / segs + 2u) * (f.k - 1u) + 16u;
  const uint64_t n_vec_max = ((slab_max + 15) >> 4) + 1;
  if (n_vec_max + 2 > SR_VEC_ROUNDS * 64ull) return NTHIP_OK;
  const uint32_t bits_dwords = (uint32_t)((n_vec_max + 4 + 3) & ~3ull);
  const uint32_t stage_vals = (64u * SW * per + 16u + 511u) & ~511u;
  const size_t per_wave = (size_t)stage_vals * 8 + (size_t)bits_dwords * 8;
  const uint32_t n_grp = f.n_seeds * G;
  const size_t fixed = (((size_t)SA_TAB_ENTRIES + n_grp + ((n_grp + 3) >> 2) + 15) & ~(size_t)15) * sizeof(uint4) +
                       (size_t)sd->roll_terms * 256;
  const size_t cap = lds_cap_of(c);
  if (fixed + 2 * per_wave > cap) return NTHIP_OK;
  uint32_t waves = (uint32_t)((cap - fixed) / per_wave);
  if (waves > SR_MAX_WAVES) waves = SR_MAX_WAVES;
  if (c->tune.seed_roll_waves && c->tune.seed_roll_waves < waves) waves = c->tune.seed_roll_waves;
  const uint4* fw = nullptr;
  NTCHK(get_fw_tab(c, &fw));
  SeedRollArgs a;
  memset(&a, 0, sizeof a);
  a.seqs = f.seqs;
  a.hashes = f.hashes;
  a.dirty = f.dirty;
  a.pair_tabs = sd->d_roll_tabs;
  a.fw_tabs = fw;
  a.any_mask = sd->d_any_mask;
  a.any_acorr = sd->d_any_acorr;
  a.n_items = f.n_runs * segs;
  a.n_tiles = (a.n_items + 63) / 64;
  a.len = f.len;
  a.k = f.k;
  a.m2 = f.m2;
  a.n_seeds = f.n_seeds;
  a.nwin = f.nwin;
  a.any_groups = G;
  a.segs = segs;
  a.inv_segs = (uint32_t)((1ull << 32) / segs + 1);
  a.n_runs = sd->roll_terms;
  a.waves = waves;
  a.bits_dwords = bits_dwords;
  a.stage_vals = stage_vals;
  memcpy(a.seed_first, sd->roll_first, sizeof a.seed_first);
  memcpy(a.run_end, sd->roll_in, sizeof a.run_end);
  memcpy(a.run_first, sd->roll_out, sizeof a.run_first);
  for (uint32_t i = 0; i < (uint32_t)SF_MAX_RUNTIME_M; ++i) a.mult[i] = multiplier(f.k, i);
  const size_t lds = fixed + per_wave * waves;
  const uint64_t need = (a.n_tiles + waves - 1) / waves;
  uint64_t grid = (uint64_t)c->n_cu; // (one block per CU: the tables alone are 64 KiB)
  if (grid > need) grid = need;
  a.total_bytes = f.n_runs * (uint64_t)f.len;
  a.step_reads = (64ull * grid * waves) / segs;
  a.step_segs = (uint32_t)((64ull * grid * waves) % segs);
  auto go = [&](auto kernel) -> int {
    NTCHK(set_max_lds(c, kernel, lds));
    prof_begin(c, "seed_roll_kernel");
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(waves * 64), lds, c->stream, a);
    prof_end(c);
    HIPCHK(hipGetLastError());
    return NTHIP_OK;
  };
  if (SW == 16) NTCHK(go(seed_roll_kernel<16>));
  else if (SW == 8) NTCHK(go(seed_roll_kernel<8>));
  else NTCHK(go(seed_roll_kernel<4>));
  *ran = true;
  return NTHIP_OK;
}

// seed_ps_kernel (seed_ps_kernel.hpp): the seeds as sparse sums over the prefix XOR (and the terms) of a read, a lane per
// segment of W positions / windows; compiled for the very seed set and read shape at run time when the batch is worth it
// (capi_seed_jit.hip).  The geometry and the reads of a shape:
struct PsGeo {
  uint32_t W = 0, nb_log = 0, lpr_log = 0, segs_b = 0, n_arrays = 0, waves = 0;
  size_t wave_bytes = 0, fixed = 0;
  double cost = 0;
  std::vector<uint32_t> first;                      // seed s reads [first[s], first[s + 1])
  std::vector<std::pair<uint32_t, uint32_t>> reads; // (array: 0 prefix, 1 terms; e)
};
// (jit: the specialised kernel, which stages a tile's values in LDS -- seed_psj_kernel.inc: STAGE_U64)
bool ps_geometry(nthip_ctx* c, const nthip_seeds* sd, uint32_t len, uint32_t m2, bool jit, PsGeo* out)
{
  const uint32_t k = sd->k;
  if (len < k || sd->h_care.empty() || sd->n_seeds > PX_MAX_SEEDS || m2 == 0 || m2 > (uint32_t)SF_MAX_RUNTIME_M) return false;
  const uint32_t nwin = len - k + 1, n_seeds = sd->n_seeds, per = n_seeds * m2;
  int force = c->tune.seed_px_array ? (int)c->tune.seed_px_array - 1 : -1;
  if (force > 2) return false; // (a stride form: seed_px_kernel's)
  if (sd->ps_plan_len != len || sd->ps_plan_force != force) {
    sd->ps_plan = px_make_plan(sd->h_care, k, (double)len / nwin, force, /*allow_strides=*/false);
    sd->ps_plan_len = len;
    sd->ps_plan_force = force;
  }
  const PxPlan& plan = sd->ps_plan;
  if (!plan.ok) return false;
  // the kernel's arrays: the prefix first, the terms second
  const bool has_raw = std::find(plan.arrays.begin(), plan.arrays.end(), PxArray{0, 0}) != plan.arrays.end();
  const uint32_t n_arrays = has_raw ? 2u : 1u;
  const size_t cap = lds_cap_of(c);
  PsGeo best;
  bool found = false;
  for (uint32_t lpr_log = 4; lpr_log <= 6; ++lpr_log) {
    if (c->tune.seed_ps_lanes && c->tune.seed_ps_lanes != (1u << lpr_log)) continue;
    const uint32_t lpr = 1u << lpr_log, W = (nwin + lpr - 1) / lpr;
    if (W == 0 || W > PS_MAX_W) continue;
    const uint32_t segs_b = (len + 1 + W - 1) / W;
    uint32_t nb_log = 4;
    while ((1u << nb_log) < segs_b) ++nb_log;
    if (nb_log > 6) continue;
    const uint32_t rpw = 64u >> lpr_log, epr = W << nb_log;
    if ((uint64_t)rpw * len + 16 > PX_VEC_ROUNDS * 1024ull - 64) continue;
    const uint32_t codes_dw = ((rpw * len + 16u + 15u) >> 4) + 4u;
    PsGeo g;
    g.W = W; g.nb_log = nb_log; g.lpr_log = lpr_log; g.segs_b = segs_b; g.n_arrays = n_arrays;
    g.wave_bytes = (size_t)n_arrays * rpw * epr * 16 + ((codes_dw * 4 + 15) & ~15u);
    if (jit) { // (the stage takes the place of the arrays)
      const uint32_t nv = rpw * nwin * per;
      const uint32_t seg_vals = W * per; // (seed_psj_kernel.inc: STAGE_U64 -- a slot of padding per segment's worth of values)
      const size_t stage = (((size_t)nv + ((seg_vals & 1u) ? 0 : nv / seg_vals + 1) + 2 + 1) / 2) * 16, arrays = (size_t)n_arrays * rpw * epr * 16;
      if (stage > arrays) g.wave_bytes += stage - arrays;
      if ((uint64_t)W * per > 64) continue; // (a segment's values wait in registers)
    }
    g.fixed = (size_t)4 * epr * 16 + (64 + (k + W) / W) * 16; // (+ room behind the last wave for the reads of its idle lanes)
    if (g.fixed + g.wave_bytes > cap) continue;
    // wave instructions per window: the build's two passes per round, W steps of the seeds' reads and rotations
    const uint32_t rounds = (rpw + (64u >> nb_log) - 1) / (64u >> nb_log);
    const double step = 12.0 + 38.0 * n_seeds + 3.2 * plan.n_terms() + 6.0 * (per - n_seeds);
    g.cost = (rounds * (20.0 * W + 45.0) + W * step + 60.0) / ((double)rpw * nwin);
    const size_t waves_fit = (cap - g.fixed) / g.wave_bytes;
    if (waves_fit < 12) g.cost *= 1.0 + 0.06 * (12 - waves_fit); // (few waves: the phases of a tile are not hidden)
    if (!found || g.cost < best.cost) { best = g; found = true; }
  }
  if (!found) return false;
  uint32_t waves = (uint32_t)((cap - best.fixed) / best.wave_bytes);
  if (waves > PS_MAX_WAVES) waves = PS_MAX_WAVES;
  if (c->tune.seed_px_waves && c->tune.seed_px_waves < waves) waves = c->tune.seed_px_waves;
  if (waves == 0) return false;
  best.waves = waves;
  best.first.assign(n_seeds + 1, 0);
  for (uint32_t s = 0; s < n_seeds; ++s) {
    for (uint32_t i = plan.seed_first[s]; i < plan.seed_first[s + 1]; ++i) {
      const PxArray& y = plan.arrays[plan.terms[i].arr];
      best.reads.push_back({y.d1 == 1u ? 0u : 1u, plan.terms[i].e});
    }
    best.first[s + 1] = (uint32_t)best.reads.size();
  }
  *out = best;
  return true;
}
} // namespace
bool ntamd::host::seed_jit_shape(nthip_ctx* c, const nthip_seeds* sd, uint32_t len, uint32_t m2, SeedJitShape* out)
{
  PsGeo g;
  if (!ps_geometry(c, sd, len, m2, true, &g)) return false;
  if ((uint64_t)g.W * g.reads.size() > 2048) return false; // (straight-line code: reads x steps)
  SeedJitShape j;
  j.len = len; j.k = sd->k; j.nwin = len - sd->k + 1; j.m2 = m2; j.n_seeds = sd->n_seeds; j.W = g.W; j.nb_log = g.nb_log;
  j.lpr_log = g.lpr_log; j.n_arrays = g.n_arrays; j.segs_b = g.segs_b;
  j.waves = g.waves;
  for (auto& r : g.reads) {
    j.term_arr.push_back(r.first);
    j.term_e.push_back(r.second);
  }
  j.seed_first = g.first;
  *out = j;
  return true;
}
namespace {
// *ran = false: not this kernel's shape (reads of more than ~2000 bases or 64 segments, more than 8 hashes per seed, a gap
// between the reads), or another form is cheaper.
int launch_seed_ps(nthip_ctx* c, const SeedFixedArgs& f, const nthip_seeds* sd, bool* ran)
{
  *ran = false;
  if (c->tune.seed_ps == 2 || c->tune.seed_px == 1 || c->tune.seed_any == 1 || c->tune.seed_roll == 1 || f.stride != f.len || f.n_runs == 0)
    return NTHIP_OK;
  PsGeo best;
  if (!ps_geometry(c, sd, f.len, f.m2, false, &best)) return NTHIP_OK;
  const uint32_t per = f.n_seeds * f.m2;
  const uint32_t W = best.W, NB = 1u << best.nb_log, rpw = 64u >> best.lpr_log, epr = W << best.nb_log, waves = best.waves;
  const uint32_t n_terms = (uint32_t)best.reads.size();
  // compiled for the shape (a second or two once per seed set and read shape, then the disk cache -- on a thread of its own
  // unless NTHIP_SEED_JIT=1: until the code object is there the other forms hash): straight-line code, every read an
  // immediate offset
  const bool jit_wanted = c->tune.seed_jit != 2 && (c->tune.seed_jit == 1 || f.n_runs * (uint64_t)f.nwin >= (1ull << 22)) &&
                          (uint64_t)W * n_terms <= 2048;
  if (c->tune.seed_ps != 1) {
    // Against the other dense forms, picoseconds per window, fitted on tools/seed_sweep.py and seed_roll_sweep.py
    // (profiles/r06_seed_sweep*.txt, r06_seed_roll_sweep.txt).  The precompiled segment kernel is behind them on every
    // shape measured (its reads cost an address and a scalar load each): it hashes only when asked for.
    if (!jit_wanted) return NTHIP_OK;
    const double pos_per_win = (double)f.len / f.nwin, extra = per - f.n_seeds;
    // (the specialised kernel: a tile's fixed work, a seed's rotations, 0.11 per read of 16 bytes, the build per position, 1.5 per
    //  further hash -- over the root of the fraction of its window lanes that have a window: idle lanes cost, but less than
    //  their share)
    double fill = 1.0;
    {
      PsGeo gj;
      if (ps_geometry(c, sd, f.len, f.m2, true, &gj)) fill = (double)f.nwin / ((double)(1u << gj.lpr_log) * gj.W);
    }
    double psj = (1.9 + 1.4 * f.n_seeds + 0.11 * n_terms + 0.25 * pos_per_win + 1.5 * extra) / sqrt(fill);
    const double hbm = (8.0 * per + pos_per_win) / 5.5; // (bytes per window at 5.5 TB/s)
    if (psj < hbm) psj = hbm;
    double direct;
    if (f.k <= 32) {
      static const double few[5] = {0, 2.8, 4.15, 6.7, 8.5}; // (the rotated-slot tables: one to four seeds)
      direct = (f.n_seeds <= 4 ? few[f.n_seeds] : 4.3 * f.n_seeds) + 1.4 * extra;
    } else {
      const size_t tab_bytes = (size_t)f.n_seeds * ((f.k + 7) / 8) * 2 * 4096; // (8 KiB of byte tables per 8 bases and seed)
      direct = 0.36 * ((f.k + 3) / 4) * f.n_seeds * (tab_bytes > 120 * 1024 ? 1.25 : 1.0) + 0.8 * extra;
      if (f.k > 128) direct = 1.0 * ((f.k + 15) / 16) * f.n_seeds + 0.8 * extra; // (the any-seed form)
    }
    uint32_t runs = 0;
    for (const auto& care : sd->h_care)
      for (uint32_t p = 0; p < f.k; ++p) runs += care[p] && (p == 0 || !care[p - 1]);
    const double roll = per <= 8 && runs <= SR_MAX_RUNS ? f.n_seeds * (3.0 + 0.004 * f.k) + 0.27 * runs + 3.0 * extra : 1e9;
    static const bool verbose = getenv("NTHIP_JIT_VERBOSE") != nullptr;
    if (verbose)
      fprintf(stderr, "nthash_amd: seed forms, ps per window: specialised %.2f (fill %.2f, %u reads), tables %.2f, roll %.2f (%u runs)\n", psj, fill,
              n_terms, direct, roll, runs);
    if (psj >= 0.97 * (direct < roll ? direct : roll)) return NTHIP_OK;
  }
  const uint64_t n_tiles = (f.n_runs + rpw - 1) / rpw, need = (n_tiles + waves - 1) / waves;
  if (jit_wanted) {
    SeedJitShape j;
    std::string why;
    if (seed_jit_shape(c, sd, f.len, f.m2, &j)) {
      HIPCHK(hipSetDevice(c->device));
      if (j.waves > 12 && j.waves < 16) j.waves = 12; // (13-15 waves are four on one SIMD: 128 registers, as for 16)
      const bool wait = c->tune.seed_jit == 1;
      const int force_j = c->tune.seed_px_array ? (int)c->tune.seed_px_array : 0;
      const uint64_t fn_key = ((uint64_t)f.len << 32) ^ ((uint64_t)f.m2 << 24) ^ ((uint64_t)c->device << 16) ^ ((uint64_t)force_j << 8) ^
                              (c->tune.seed_ps_lanes | (c->tune.seed_px_waves << 4));
      auto known = sd->psj_ready.find(fn_key);
      if (known != sd->psj_ready.end()) j.waves = known->second.second;
      hipFunction_t fn = known != sd->psj_ready.end() ? (hipFunction_t)known->second.first : (hipFunction_t)seed_psj_get(c, sd, j, wait, &why);
      // a kernel that spills at this block size gets the registers of a smaller one
      while (!fn && why.find("spills") != std::string::npos && j.waves > 4) {
        j.waves = j.waves > 12 ? 12 : j.waves > 8 ? 8 : 4;
        why.clear();
        fn = (hipFunction_t)seed_psj_get(c, sd, j, wait, &why);
      }
      if (fn) {
        sd->psj_ready[fn_key] = {(void*)fn, j.waves};
        const uint32_t jw = j.waves, jrpw = 64u >> j.lpr_log; // (the specialised kernel's own tiles: its geometry counts the stage)
        const uint64_t n_tiles_j = (f.n_runs + jrpw - 1) / jrpw;
        int lds_static = 0;
        (void)hipFuncGetAttribute(&lds_static, HIP_FUNC_ATTRIBUTE_SHARED_SIZE_BYTES, fn);
        uint64_t per_cu = lds_static > 0 ? lds_cap_of(c) / (size_t)lds_static : 1;
        if (per_cu < 1) per_cu = 1;
        if (per_cu * jw > 32) per_cu = 32 / jw ? 32 / jw : 1;
        uint64_t grid = (uint64_t)c->n_cu * per_cu;
        const uint64_t need_j = (n_tiles_j + jw - 1) / jw;
        if (grid > need_j) grid = need_j;
        const uint8_t* seqs = f.seqs;
        uint64_t* hashes = f.hashes;
        uint32_t* dirty = f.dirty;
        uint64_t n_reads = f.n_runs, nt = n_tiles_j, total_bytes = f.n_runs * (uint64_t)f.len;
        void* args[] = {&seqs, &hashes, &dirty, &n_reads, &nt, &total_bytes};
        prof_begin(c, "seed_psj_kernel");
        const hipError_t e = hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, jw * 64, 1, 1, 0, c->stream, args, nullptr);
        prof_end(c);
        if (e == hipSuccess) {
          *ran = true;
          return NTHIP_OK;
        }
        (void)hipGetLastError(); // (a launch the device refuses: the precompiled kernel below)
      } else if (c->tune.seed_jit == 1 && getenv("NTHIP_JIT_VERBOSE")) {
        fprintf(stderr, "nthash_amd: no specialised seed kernel (%s)\n", why.c_str());
      }
    }
  }
  // (no specialised kernel, or not yet -- it is being compiled on a thread of its own: the other dense forms hash this batch)
  if (c->tune.seed_ps != 1) return NTHIP_OK;
  // the reads, per step of a segment: term t at step i is entry ((i + e) % W) * NB + (i + e) / W of its array
  const uint32_t arr_bytes = rpw * epr * 16u;
  const int force = c->tune.seed_px_array ? (int)c->tune.seed_px_array - 1 : -1;
  const uint64_t key = ((uint64_t)f.len << 40) ^ ((uint64_t)W << 32) ^ ((uint64_t)best.nb_log << 24) ^ ((uint64_t)best.lpr_log << 16) ^
                       ((uint64_t)(force + 2) << 8) ^ best.n_arrays;
  if (!sd->d_ps_off || sd->ps_key != key) {
    std::vector<uint32_t> off((size_t)W * n_terms + 4, 0);
    for (uint32_t i = 0; i < W; ++i)
      for (uint32_t t = 0; t < n_terms; ++t) {
        const uint32_t j = i + best.reads[t].second;
        off[(size_t)i * n_terms + t] = best.reads[t].first * arr_bytes + ((j % W) * NB + j / W) * 16u;
      }
    HIPCHK(hipSetDevice(c->device));
    if (sd->d_ps_off) {
      HIPCHK(hipStreamSynchronize(c->stream));
      (void)hipFree(sd->d_ps_off);
      sd->d_ps_off = nullptr;
    }
    HIPCHK(hipMalloc((void**)&sd->d_ps_off, off.size() * 4));
    HIPCHK(hipMemcpy(sd->d_ps_off, off.data(), off.size() * 4, hipMemcpyHostToDevice));
    sd->ps_key = key;
  }
  SeedPsArgs a;
  memset(&a, 0, sizeof a);
  a.seqs = f.seqs;
  a.hashes = f.hashes;
  a.dirty = f.dirty;
  a.step_off = sd->d_ps_off;
  a.n_reads = f.n_runs;
  a.n_tiles = n_tiles;
  a.total_bytes = f.n_runs * (uint64_t)f.len;
  a.len = f.len;
  a.k = f.k;
  a.m2 = f.m2;
  a.n_seeds = f.n_seeds;
  a.nwin = f.nwin;
  a.W = W;
  a.nb_log = best.nb_log;
  a.lpr_log = best.lpr_log;
  a.n_arrays = best.n_arrays;
  a.n_terms = n_terms;
  a.waves = waves;
  a.segs_b = best.segs_b;
  a.k31 = (f.k - 1u) % 31u;
  a.k33 = (f.k - 1u) % 33u;
  for (uint32_t s = 0; s <= f.n_seeds; ++s) a.seed_first[s] = best.first[s];
  for (uint32_t i = 0; i < (uint32_t)SF_MAX_RUNTIME_M; ++i) a.mult[i] = multiplier(f.k, i);
  const size_t lds = best.fixed + best.wave_bytes * waves;
  const int perc = per <= 2 ? (int)per : 0;
  auto go = [&](auto kernel) -> int {
    int per_cu = 1;
    NTCHK(blocks_per_cu(c, kernel, (int)waves * 64, lds, &per_cu));
    uint64_t grid = (uint64_t)c->n_cu * per_cu;
    if (grid > need) grid = need;
    prof_begin(c, "seed_ps_kernel");
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(waves * 64), lds, c->stream, a);
    prof_end(c);
    HIPCHK(hipGetLastError());
    return NTHIP_OK;
  };
  if (perc == 1) NTCHK(go(seed_ps_kernel<1>));
  else if (perc == 2) NTCHK(go(seed_ps_kernel<2>));
  else NTCHK(go(seed_ps_kernel<0>));
  *ran = true;
  return NTHIP_OK;
}

// seed_px_kernel (seed_px_plan.hpp): the seeds as sparse sums over scanned term arrays.  *ran = false: not this kernel's
// shape (reads of more than its slab, more than 8 hashes per seed, a gap between the reads), or another form is cheaper.
int launch_seed_px(nthip_ctx* c, const SeedFixedArgs& f, const nthip_seeds* sd, bool* ran)
{
  *ran = false;
  // (only when asked for: two positions per lane and step with a wave-wide scan -- ~85 instructions per position -- leave it
  //  behind the other dense forms on every shape of tools/seed_sweep.py / seed_roll_sweep.py; it is the form that takes the
  //  stride-d scans of the plan and reads of up to ~2000 bases, kept bit-exact by tests/test_gpu_seed_px.py)
  if (c->tune.seed_px != 1 || c->tune.seed_any == 1 || c->tune.seed_roll == 1 || f.stride != f.len || f.m2 > (uint32_t)SF_MAX_RUNTIME_M ||
      sd->h_care.empty() || f.n_seeds > PX_MAX_SEEDS || f.n_runs == 0)
    return NTHIP_OK;
  const int force = c->tune.seed_px_array ? (int)c->tune.seed_px_array - 1 : -1;
  if (sd->px_plan_len != f.len || sd->px_plan_force != force) {
    sd->px_plan = px_make_plan(sd->h_care, f.k, (double)f.len / f.nwin, force);
    sd->px_plan_len = f.len;
    sd->px_plan_force = force;
  }
  const PxPlan& plan = sd->px_plan;
  if (!plan.ok) return NTHIP_OK;
  const uint32_t per = f.n_seeds * f.m2, n_arrays = (uint32_t)plan.arrays.size();
  const int perc = per <= 2 ? (int)per : 0;
  // reads per tile: a wave's arrays, stage and codes in a 16th of the CU's LDS where a read allows it (two waves per SIMD
  // hide nothing: a tile is a chain of short phases), then the count nearby that leaves the fewest idle lanes in a
  // tile's last group of 64 windows
  const size_t cap = lds_cap_of(c);
  const uint32_t stage_vals = perc ? 0u : (64u * per + 31u + 127u) & ~127u;
  auto entries_of = [&](uint32_t R) { return (15u + R * f.len + plan.reach + 1u + 127u) & ~127u; };
  auto wave_bytes_of = [&](uint32_t R) {
    const uint32_t ne = entries_of(R);
    return (size_t)n_arrays * ne * 16 + (size_t)stage_vals * 8 + ((ne >> 4) + 4) * 4;
  };
  const size_t target = (cap - 256) / 16;
  uint32_t R_max = 0;
  for (uint32_t R = 1; R <= 64; ++R) {
    if (15ull + (uint64_t)R * f.len > PX_VEC_ROUNDS * 1024ull || entries_of(R) > 2048u) break;
    if (wave_bytes_of(R) > (R == 1 ? cap - 256 : target)) break;
    R_max = R;
  }
  if (R_max == 0) return NTHIP_OK;
  uint32_t R = R_max;
  if (c->tune.seed_px_reads) R = c->tune.seed_px_reads < R_max ? c->tune.seed_px_reads : R_max;
  else {
    double best = -1;
    for (uint32_t r = R_max; r >= 1 && r + 3 >= R_max; --r) {
      const uint32_t w = r * f.nwin, groups = (w + 63u) / 64u;
      const double fill = (double)w / (groups * 64.0) - 0.01 * (R_max - r); // (a smaller tile: more of the per-tile work)
      if (fill > best) {
        best = fill;
        R = r;
      }
    }
  }
  if (c->tune.seed_px != 1) { // the cost model against the other dense forms: picoseconds per window (tools/seed_px_fit.py)
    const double pos_per_win = (double)f.len / f.nwin;
    double arr = 0;
    for (const PxArray& y : plan.arrays) arr += (y.d1 == 1 ? 1.4 : 1.0) + (y.d1 > 1 ? 2.0 : 0.0) + (y.d2 > 1 ? 2.0 : 0.0);
    const double px = 0.45 * pos_per_win * arr + 0.085 * plan.n_terms() + 0.55 * f.n_seeds + 0.25 * (per - f.n_seeds) + 0.6;
    const uint32_t tables = f.k <= 64 ? 2 * ((f.k + 7) / 8) : 4 * ((f.k + 15) / 16);
    const double direct = ((f.k <= 128 ? 1.34 * tables : 6.2 * sd->any_groups) * f.n_seeds + 4.5 * (per - f.n_seeds)) * 0.205;
    if (px >= direct) return NTHIP_OK;
  }
  const size_t per_wave = wave_bytes_of(R);
  uint32_t waves = (uint32_t)((cap - 256) / per_wave);
  if (waves > PX_MAX_WAVES) waves = PX_MAX_WAVES;
  if (c->tune.seed_px_waves && c->tune.seed_px_waves < waves) waves = c->tune.seed_px_waves;
  if (waves == 0) return NTHIP_OK;
  SeedPxArgs a;
  memset(&a, 0, sizeof a);
  a.seqs = f.seqs;
  a.hashes = f.hashes;
  a.dirty = f.dirty;
  a.n_reads = f.n_runs;
  a.n_tiles = (f.n_runs + R - 1) / R;
  a.total_bytes = f.n_runs * (uint64_t)f.len;
  a.len = f.len;
  a.k = f.k;
  a.m2 = f.m2;
  a.n_seeds = f.n_seeds;
  a.nwin = f.nwin;
  a.inv_nwin = (uint32_t)((1ull << 32) / f.nwin + 1);
  a.R = R;
  a.n_entries = entries_of(R);
  a.n_arrays = n_arrays;
  a.waves = waves;
  a.stage_vals = stage_vals;
  uint32_t g = per, h = 16;
  while (h) { const uint32_t t = g % h; g = h; h = t; }
  a.align_win = 16u / g;
  a.reach = plan.reach + 1u;
  a.k31 = (f.k - 1u) % 31u;
  a.k33 = (f.k - 1u) % 33u;
  // the kernel's order: the arrays that start from the prefix XOR first
  uint32_t slot_of[PX_MAX_ARRAYS] = {}, n_slot = 0;
  for (int pre = 1; pre >= 0; --pre)
    for (uint32_t y = 0; y < n_arrays; ++y)
      if ((plan.arrays[y].d1 == 1u) == (pre == 1)) {
        slot_of[y] = n_slot;
        a.arr_d[n_slot] = pre ? plan.arrays[y].d2 : plan.arrays[y].d1;
        if (!pre && plan.arrays[y].d2) return NTHIP_OK; // (two strides on the terms: not a form the planner makes)
        ++n_slot;
        if (pre) ++a.n_pre;
      }
  for (uint32_t s = 0; s <= f.n_seeds; ++s) a.seed_first[s] = plan.seed_first[s];
  for (uint32_t i = 0; i < plan.n_terms(); ++i)
    a.term_off[i] = slot_of[plan.terms[i].arr] * a.n_entries * 16u + (uint32_t)plan.terms[i].e * 16u;
  for (uint32_t i = 0; i < (uint32_t)SF_MAX_RUNTIME_M; ++i) a.mult[i] = multiplier(f.k, i);
  const size_t lds = 256 + per_wave * waves;
  auto go = [&](auto kernel) -> int {
    int per_cu = 1;
    NTCHK(blocks_per_cu(c, kernel, (int)waves * 64, lds, &per_cu));
    const uint64_t need = (a.n_tiles + waves - 1) / waves;
    uint64_t grid = (uint64_t)c->n_cu * per_cu;
    if (grid > need) grid = need;
    prof_begin(c, "seed_px_kernel");
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(waves * 64), lds, c->stream, a);
    prof_end(c);
    HIPCHK(hipGetLastError());
    return NTHIP_OK;
  };
  if (perc == 1) NTCHK(go(seed_px_kernel<1>));
  else if (perc == 2) NTCHK(go(seed_px_kernel<2>));
  else NTCHK(go(seed_px_kernel<0>));
  *ran = true;
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

// Long reads (seed_long_kernels.hpp): the batch's reads are cut into independent pieces of about SEED_LONG_S windows at
// positions with 2k bases around them, the pieces go through the span path (a wave per piece instead of a wave per
// read), counts and positions are folded back per read.  *handled = false: nothing done.
namespace {
constexpr uint64_t SEED_LONG_DENSE_MAX = 131072; // fixed-length reads from this length on skip the dense kernels
struct DevTemp { // device temporaries of one call
  std::vector<void*> ptrs;
  ~DevTemp() { for (void* p : ptrs) (void)hipFree(p); }
  template <typename T> int get(T** out, size_t n)
  {
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, (n ? n : 1) * sizeof(T)));
    ptrs.push_back(p);
    *out = (T*)p;
    return NTHIP_OK;
  }
};
} // namespace
int ntamd::host::run_seed_long(nthip_ctx* c, const Staged& st, const nthip_reads* rd, const nthip_seeds* sd, uint32_t m2,
                               uint64_t capacity, uint64_t* total, bool* handled, const uint64_t* d_ends)
{
  *handled = false;
  const uint64_t n = rd->n_reads;
  if (n == 0 || c->tune.no_seed_long || c->tune.no_seed_wave) return NTHIP_OK;
  SeedLongArgs a;
  memset(&a, 0, sizeof a);
  a.seqs = st.seqs;
  a.offsets = st.offsets;
  a.ends = d_ends;
  a.n_reads = n;
  a.len = rd->fixed_len;
  a.stride = rd->stride ? rd->stride : rd->fixed_len;
  a.k = sd->k;
  a.S = sd->k * 4u > 1280u ? sd->k * 4u : 1280u;
  DevTemp tmp;
  const unsigned gblocks = (unsigned)c->n_cu * 8;
  uint64_t* d_tot = nullptr;
  NTCHK(tmp.get(&d_tot, 2));
  uint64_t P = 0;
  if (st.offsets) {
    uint64_t *d_pieces = nullptr, *d_pbase = nullptr, *d_sums = nullptr;
    NTCHK(tmp.get(&d_pieces, n));
    NTCHK(tmp.get(&d_pbase, n + 1));
    NTCHK(tmp.get(&d_sums, (n + SCAN_TILE - 1) / SCAN_TILE + 16));
    hipLaunchKernelGGL(seed_long_count_kernel, dim3(gblocks), dim3(256), 0, c->stream, a, d_pieces);
    HIPCHK(hipGetLastError());
    NTCHK(device_exclusive_scan(c, d_pieces, d_pbase, n, d_sums, d_tot));
    HIPCHK(hipMemcpyAsync(d_pbase + n, d_tot, 8, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->h_small + 8, d_tot, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(&P, c->h_small + 8, 8);
    a.pbase = d_pbase;
  } else {
    a.pieces_per_read = a.len < a.k ? 1 : (a.len - a.k + 1 + a.S - 1) / a.S;
    P = n * a.pieces_per_read;
  }
  a.n_pieces = P;
  if (P <= n) return NTHIP_OK; // no read is longer than a piece
  uint64_t *d_valid = nullptr, *d_cut = nullptr, *d_idx = nullptr, *d_sums2 = nullptr;
  NTCHK(tmp.get(&d_valid, P));
  NTCHK(tmp.get(&d_cut, P));
  NTCHK(tmp.get(&d_idx, P));
  NTCHK(tmp.get(&d_sums2, (P + SCAN_TILE - 1) / SCAN_TILE + 16));
  hipLaunchKernelGGL(seed_long_lastbase_kernel, dim3(gblocks), dim3(256), 0, c->stream, a, d_idx); // (d_idx: free until the scan)
  hipLaunchKernelGGL(seed_long_cut_kernel, dim3(gblocks), dim3(256), 0, c->stream, a, (const uint64_t*)d_idx, d_valid, d_cut);
  HIPCHK(hipGetLastError());
  NTCHK(device_exclusive_scan(c, d_valid, d_idx, P, d_sums2, d_tot));
  HIPCHK(hipMemcpyAsync(c->h_small + 8, d_tot, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  uint64_t m = 0;
  memcpy(&m, c->h_small + 8, 8);
  uint64_t *d_ss = nullptr, *d_se = nullptr, *d_sr = nullptr, *d_rel = nullptr, *d_cnt = nullptr;
  NTCHK(tmp.get(&d_ss, m));
  NTCHK(tmp.get(&d_se, m));
  NTCHK(tmp.get(&d_sr, m));
  NTCHK(tmp.get(&d_rel, m));
  NTCHK(tmp.get(&d_cnt, m));
  hipLaunchKernelGGL(seed_long_spans_kernel, dim3(gblocks), dim3(256), 0, c->stream, a, d_valid, d_idx, d_cut, d_ss, d_se,
                     d_sr, d_rel);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemsetAsync(d_cnt, 0, m * 8, c->stream));
  // ---- the pieces as a batch of spans ----
  Staged st2;
  st2.seqs = st.seqs;
  st2.offsets = d_ss;
  st2.hashes = st.hashes;
  st2.counts = d_cnt;
  st2.pos = st.pos;
  st2.fwd = st.fwd;
  st2.rev = st.rev;
  nthip_reads rd2 = *rd;
  rd2.n_reads = m;
  rd2.offsets = d_ss; // (a device pointer: only its being non-NULL matters from here on)
  rd2.fixed_len = 0;
  rd2.stride = 0;
  NTCHK(run_seed_general(c, st2, &rd2, sd, m2, capacity, total, d_se, false));
  // ---- back to the caller's reads ----
  if (st.counts) {
    HIPCHK(hipMemsetAsync(st.counts, 0, n * 8, c->stream));
    hipLaunchKernelGGL(seed_long_fold_counts_kernel, dim3(gblocks), dim3(256), 0, c->stream, d_cnt, d_sr, m, st.counts);
    HIPCHK(hipGetLastError());
  }
  if (st.pos) {
    uint64_t *d_off = nullptr, *d_sums3 = nullptr;
    NTCHK(tmp.get(&d_off, m));
    NTCHK(tmp.get(&d_sums3, (m + SCAN_TILE - 1) / SCAN_TILE + 16));
    NTCHK(device_exclusive_scan(c, d_cnt, d_off, m, d_sums3, d_tot));
    hipLaunchKernelGGL(seed_long_fold_pos_kernel, dim3(gblocks), dim3(256), 0, c->stream, d_cnt, d_off, d_rel, m, st.pos);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  *handled = true;
  return NTHIP_OK;
}

extern "C" int nthip_seed_hash(nthip_ctx* c, const nthip_reads* rd_in, const nthip_seeds* sd, uint8_t m28,
                               const nthip_out* out, uint64_t* total_out, uint32_t flags)
{
  if (!c || !sd) return fail(NTHIP_ERR_ARG, "ctx/seeds is NULL");
  NTCHK(check_reads(rd_in));
  nthip_reads eff = *rd_in; // what the paths below see: offsets of equal-length, back-to-back reads become a fixed length
  const nthip_reads* rd = &eff;
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

  // offsets: one pass over them before anything trusts them (order, bounds), and reads that all have one length and
  // lie back to back are a fixed-length batch (as in nthip_kmer_hash): the dense kernel instead of the variable-length path
  uint64_t max_len = rd->fixed_len; // the longest read of the batch
  if (st.offsets) {
    OffsetsSurvey sv;
    NTCHK(offsets_survey_device(c, st.offsets, rd->n_reads, total_bytes, &sv));
    if (sv.bad) return fail(NTHIP_ERR_ARG, "offsets / spans are not non-decreasing or reach outside the read buffer");
    max_len = sv.max_len;
    if (st.pos && sv.max_len > 0xFFFFFFFFull) // (as nthip_kmer_hash: the façade hashes long sequences window by window)
      return fail(NTHIP_ERR_UNSUPPORTED, "out->pos is 32 bits wide: a read of %llu bases cannot report its positions",
                  (unsigned long long)sv.max_len);
    if (sv.uniform && !(flags & NTHIP_FORCE_GENERAL) && rd->n_reads >= 1024 && sv.len0 >= 1 && sv.len0 < (1ull << 30) &&
        sv.off0 + rd->n_reads * sv.len0 <= total_bytes) {
      st.seqs += sv.off0;
      st.offsets = nullptr;
      eff.offsets = nullptr;
      eff.fixed_len = (uint32_t)sv.len0;
      eff.stride = 0;
      total_bytes = rd->n_reads * sv.len0;
    }
  }

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
             stride <= len && (len < SEED_LONG_DENSE_MAX || k > 64 || c->tune.no_seed_long)) {
    // (reads of 128 Ki bases and more go to the pieces below even when clean: the block-tile kernel keeps the whole read's
    //  bit stream in LDS and stops fitting near 400 kbase.  Whole calls, 4 GiB of records, tools/seed_long_wholecall.py:
    //  40 kbase dense 83 G k-mers/s / pieces 77 G; 100 kbase 79 / 81; 300 kbase 7.6 (one wave per read) / 79)
    const uint32_t nwin = len - k + 1;
    // 16-bit halves of the window; 2*nh byte tables per seed in LDS (zero-padded).  Seeds of 65..128 bases: the wave-tile
    // kernel only (instantiated for nh = 10, 12, 14, 16), one seed of up to 128 KiB of tables per pass
    const uint32_t nh = k <= 64 ? (k + 7) / 8 : ((k + 15) / 16) * 2;
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
      if (c->tune.seed_rpt) rpt = c->tune.seed_rpt; // A/B override
    }
    const uint64_t slab = 15ull + (uint64_t)(rpt - 1) * stride + len;
    const uint32_t bits_dwords = (uint32_t)((((slab + 15) >> 4) + 8 + 3) & ~3ull);
    const size_t dyn = table_bytes + (size_t)bits_dwords * 4 + (size_t)(SF_THREADS / 64) * (64 * per + 2) * 8;
    const uint64_t dense = rd->n_reads * (uint64_t)nwin;
    // (the block-tile kernel needs its 16 waves' tiles beside ALL the tables; the wave-tile kernel plans its own LDS)
    const bool block_fits = nh <= 8 && dyn <= 158 * 1024 && dyn <= c->lds_max;
    if ((uint64_t)rpt * nwin < 0x7FFFFFFFull) {
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
      // clean batches (the optimistic pass): one wave per tile of reads, no block barriers (seed_wtile_kernel);
      // shapes outside it (slabs of more than 8 KiB per tile, LDS) keep the block-tile kernel
      // seeds beyond 128 bases have no position tables at all: the any-seed form (k-independent tables, one pass);
      // NTHIP_TUNE_SEED_ANY=1 sends every dense batch there, =2 none of k <= 128 (A/B, tests)
      bool wtile_ran = false;
      bool prefer_any = false;
      NTCHK(launch_seed_ps(c, a, sd, &wtile_ran)); // (sparse sums over the prefix XOR of a read: the cost follows the seed)
      if (!wtile_ran) NTCHK(launch_seed_px(c, a, sd, &wtile_ran)); // (... over stride scans; longer reads)
      if (!wtile_ran) NTCHK(launch_seed_roll(c, a, sd, &wtile_ran)); // (seeds of few runs: rolled, whatever k)
      if (!wtile_ran && (k > 128 || c->tune.seed_any == 1)) NTCHK(launch_seed_any(c, a, sd, &wtile_ran));
      if (!wtile_ran && k <= 128)
        NTCHK(launch_seed_wtile(c, a, sd, nh, &wtile_ran, c->tune.seed_any == 2 ? nullptr : &prefer_any));
      if (!wtile_ran && prefer_any) { // (a seed set of several position-table passes)
        NTCHK(launch_seed_any(c, a, sd, &wtile_ran));
        if (!wtile_ran) NTCHK(launch_seed_wtile(c, a, sd, nh, &wtile_ran));
      }
      if (!wtile_ran && block_fits) {
        rc = NT_SEED_FIXED(false, dyn);
        NTCHK(rc);
      }
      uint32_t dirty = 1; // (no dense kernel for the shape: the general path below)
      if (wtile_ran || block_fits) {
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
        if (st.pos) { // every read emits every window: get_pos() is the window index
          hipLaunchKernelGGL(fill_window_pos_kernel, dim3(c->n_cu * 8), dim3(256), 0, c->stream, st.pos, rd->n_reads, nwin,
                             (const uint64_t*)nullptr, (const uint64_t*)nullptr);
          HIPCHK(hipGetLastError());
        }
        done = true;
      } else if (block_fits && dyn + (size_t)rpt * 8 <= 158 * 1024 && dyn + (size_t)rpt * 8 <= c->lds_max &&
                 (len < SEED_LONG_MIN || c->tune.no_seed_long)) { // (long reads with a non-base: cut into pieces below)
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
        const bool list_wave = !c->tune.no_seed_wave && seed_wave_plan(c, sd, m2, &wplan);
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
  if (!done && max_len >= SEED_LONG_MIN && !(flags & NTHIP_FORCE_GENERAL)) {
    // chromosomes / contigs: one wave per read would leave the chip idle; independent pieces instead
    bool handled = false;
    const int rc = run_seed_long(c, st, rd, sd, m2, out->capacity, &total, &handled);
    if (rc == NTHIP_ERR_CAPACITY && total_out) *total_out = total; // (the need, as the dense and k-mer paths report it)
    NTCHK(rc);
    done = handled;
  }
  if (!done) {
    const int rc = run_seed_general(c, st, rd, sd, m2, out->capacity, &total, nullptr, !(flags & NTHIP_FORCE_GENERAL));
    if (rc == NTHIP_ERR_CAPACITY && total_out) *total_out = total;
    NTCHK(rc);
  }
  if (total_out) *total_out = total;
  NTCHK(unstage_outputs(c, out, flags, rd->n_reads, per, total, st, sd->n_seeds));
  HIPCHK(hipStreamSynchronize(c->stream));
  return NTHIP_OK;
}
