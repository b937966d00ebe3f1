// capi_kmer_plan.hip -- per-k constants and tables, geometry plans of the run-split kernels (host code only)
// Part of libnthash_hip.so (include/nthash_hip.h); see capi_internal.hpp for the file map.
#include "capi_internal.hpp"

using namespace ntamd;
using namespace ntamd::host;

namespace ntamd {
namespace host {

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

// k > 64: the first window of a run comes from the k-independent fw tables (first_window.hpp: 16 bases per step, or the
// prefix-scan form) -- FW_ENTRIES entries, five 4 KiB tables' worth of LDS whatever k is
int get_kmer_tab(nthip_ctx* c, uint32_t k, const uint4** out)
{
  if (k <= KMER_TABLE_K_MAX) return get_init_tab(c, k, out);
  return get_fw_tab(c, out);
}
int get_fw_tab(nthip_ctx* c, const uint4** out)
{
  const uint32_t key = 0xFFFF0010u;
  auto it = c->init_tabs.find(key);
  if (it == c->init_tabs.end()) {
    std::vector<uint4> h(FW_ENTRIES);
    build_fw_tables(h.data());
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

bool kmer_runs_plan(const nthip_ctx* c, uint32_t len, uint32_t stride, uint32_t k, uint32_t m, RunsPlan* p)
{
  if (len < k || k > KMER_TABLE_K_MAX || m > (uint32_t)KF_MAX_RUNTIME_M || stride > len || stride == 0) return false;
  const uint32_t nwin = len - k + 1;
  uint32_t best = 0;
  // An ODD run length first: a lane's tile row is C 8-byte entries, and the 16 lanes of a ds_write_b64 group then fall
  // on 16 different bank pairs; C = 16 is a 16-way conflict on every tile write, 12 and 20 are 4-way, 10 / 14 / 22 2-way.
  static const uint32_t pref[] = {15, 13, 11, 17, 19, 21, 23, 25, 9, 14, 10, 22};
  for (uint32_t d : pref)
    if (nwin % d == 0) { best = d; break; }
  if (best == 0)
    for (uint32_t d = 16; d >= 4; --d)
      if (nwin % d == 0) { best = d; break; }
  if (best < 6)
    for (uint32_t d = 17; d <= 64; ++d) // e.g. a prime window count: the whole read is one run
      if (nwin % d == 0) { best = d; break; }
  // NTHIP_TUNE_RUN_LEN: A/B override of the run length (must divide the window count)
  if (const uint32_t d = c->tune.run_len) // A/B override of the run length (must divide the window count)
    if (d >= 2 && d <= 64 && nwin % d == 0) best = d;
  if (best == 0) return false;
  p->C = best;
  p->rpr = nwin / best;
  p->nw = (k + 15) / 16;
  p->tile_u64 = 64 * best;
  // reads touched by one wave tile (64 consecutive runs)
  const uint32_t slab_reads = (64 % p->rpr == 0) ? 64 / p->rpr : (p->rpr - 1 + 63) / p->rpr + 1;
  const uint64_t slab_bytes = (uint64_t)(slab_reads - 1) * stride + len;
  constexpr uint32_t AL = KR_SLAB_ALIGN - 1; // a slab starts on an aligned address at or below its first byte
  uint32_t bd = (uint32_t)((AL + slab_bytes + 15) >> 4) + p->nw + 4;
  bd = (bd + 3u) & ~3u;
  p->bits_dwords = bd;
  p->dword_tail = (AL + slab_bytes <= 1280 && !c->tune.no_dword_tail) ? 1u : 0u;
  const size_t fixed = (size_t)(4 * p->nw) * 4096 + 256 + 64; // (room for the padded tables of the runtime-k instantiations)
  // chunked path (kmer_runs_kernel.hpp, only when compiled in): every wave keeps the bit streams of a chunk's tiles
  // in LDS and is limited to 8 waves per CU
  const bool chunked = kmer_runs_chunked_compiled() && p->dword_tail && !c->tune.no_phases && m == 1 &&
                       64 % p->rpr == 0 && stride == len; // (tiles are whole reads)
  p->ph_tiles = chunked ? (c->tune.ph_tiles ? (c->tune.ph_tiles < 16u ? c->tune.ph_tiles : 16u) : 16u) : 0u;
  // burst path (kmer_runs_kernel.hpp, round 3): pieces of KR_BURST consecutive tiles per wave -- run length 15 (8 store
  // instructions per tile), tiles of whole reads, dword-tail slabs; NTHIP_TUNE_NO_PHASES=1 keeps the static loop (A/B)
  // (m = 1 only: in-process A/B, 60 M reads, +3.0 % on 150 bp / k = 31; with four hashes per k-mer the copy-out's
  //  arithmetic needs the 16 waves the longer LDS streams do not leave: -3 %)
  if (!kmer_runs_chunked_compiled() && p->dword_tail && !c->tune.no_phases && best == 15 && 64 % p->rpr == 0 && stride == len &&
      k >= 17 && m == 1) // (the instantiations with a compile-time run length: k = 31 and the runtime-k ones)
    p->ph_tiles = KR_BURST;
  const size_t per_wave = (size_t)p->tile_u64 * 8 + (size_t)bd * 4 * (p->ph_tiles ? p->ph_tiles : 1u);
  const size_t cap = (c->lds_max < 160 * 1024 ? c->lds_max : 160 * 1024) - 512;
  // m = 1: 16 waves per CU measured 1.3-2.9 % faster than 8 once the tile geometry runs on the scalar unit (round 1:
  // 8 were 2 % ahead); m > 1: the copy-out does the multi-hash expansion and more waves hide it (+6 % m=4, +9 % m=8)
  uint32_t w_max = 16;
  if (c->tune.waves) w_max = c->tune.waves;
  if (kmer_runs_chunked_compiled() && m == 1 && w_max > 8) w_max = 8; // (launch bounds of a windowed build's m = 1 kernels)
  for (uint32_t w = w_max; w >= 1; --w) {
    if (fixed + per_wave * w <= cap) {
      p->waves = w;
      p->lds = fixed + per_wave * w;
      return true;
    }
  }
  return false;
}

// Geometry of the general run-split kernel (kmer_runs_gen_kernel.hpp): any window count.
// The run length minimises lane work per k-mer: one first-window evaluation (about
// 2 + ntab/2 roll steps' worth) plus C-1 rolls per run, the windows a read's last run recomputes included.
// gaps_ok: rows may be padded (stride > len).  The padding travels through the slab like any other byte, so
// the dense pass -- which flags every non-base it stages -- does not take such batches; the N-aware passes do
// (no window reaches into the padding).
// force_c: run length to use (0: the model's choice); model_cap: longest run the model may pick (0: its default)
// no_tile: a consumer that keeps the hashes in registers (MinHash) -- a longer run then costs neither LDS nor
// bank conflicts, only fewer first windows: the model may go to 31 (measured: 150 bp m=1 798 -> 853-868 G k-mers/s,
// m=2 +16 %, m=4 +12 %)
bool kmer_gen_plan(const nthip_ctx* c, uint32_t len, uint32_t stride, uint32_t k, uint32_t m, GenPlan* p,
                   bool gaps_ok, uint32_t force_c, uint32_t model_cap, bool no_tile)
{
  if (len < k || m == 0 || (stride > len && !gaps_ok) || len >= (1u << 30) || stride >= (1u << 30)) return false;
  const uint32_t nwin = len - k + 1;
  if (stride < nwin) return false; // reads overlapping by more than k-1 bases: other paths
  const uint32_t ntab = (k + 3) / 4; // first-window cost in the model: table lookups (k within the position tables)
  const bool any_k = kmer_nw(k) == 0;
  uint32_t best = 0;
  double best_cost = 1e30;
  // The kernels take runs of up to 31 windows (the window masks of the N-aware pass are 32 bits wide), but the
  // model stops at 16: it does not see what a larger tile costs in waves per CU.  In-process A/B over 24 shapes
  // (profiles/r01_notes.md): longer runs win 3-8 % where they cut the runs per read sharply (100 bp/k64: 13 -> 19,
  // 1 kb reads, k <= 15) and lose 5-28 % elsewhere (k = 63/64 at 150 bp: -27 %).  Beyond the position tables the
  // tables are 20 KB whatever k is and the first window is dearer: the model may go to 19 there when a read has few windows
  // (in-process A/B, profiles/r03_notes.md: 19 wins on 150 bp / k = 65, 17 on the 51-window shapes -- no recomputed
  // windows --, 15 on 1 kb / k = 500; 23 and more lose 8-25 % everywhere).
  uint32_t c_cap = model_cap ? model_cap : no_tile ? 31 : any_k && nwin <= 128 ? 19 : 16;
  if (c->tune.run_max) c_cap = c->tune.run_max; // A/B knob: longest run the model may pick
  const uint32_t c_hi = nwin < c_cap ? nwin : c_cap;
  // slab of a tile of 64 runs of C windows, in dwords of the 2-bit stream: 63 run-to-run steps of at most C bases,
  // C + stride - nwin across a read boundary, plus the last run
  auto slab_bytes_of = [&](uint32_t C) -> uint64_t {
    const uint32_t rpr = (nwin + C - 1) / C;
    const uint64_t crossings = (63 + rpr - 1) / rpr;
    return 64ull * C + k - 1 + crossings * (uint64_t)(stride - nwin);
  };
  auto slab_dwords = [&](uint32_t C) -> uint64_t { return (15 + slab_bytes_of(C) + 15) >> 4; };
  // first window of a run beyond the position tables, in VALU instructions (counted on the ISA, fitted to in-process
  // A/B on seven shapes, profiles/r03_notes.md): grouped ~43 per 16 bases of k + 13; prefix scan ~230 + 60 per word of
  // the slab a lane takes, whatever k is
  auto fw_cost = [&](uint32_t C, bool* scan) -> double {
    const double grouped = 43.0 * (k / 16 + 1) + 13.0;
    const double sc = 230.0 + 60.0 * (double)((slab_dwords(C) + 1 + 63) / 64);
    *scan = c->tune.fw ? c->tune.fw == 2 : sc < grouped;
    return *scan ? sc : grouped;
  };
  for (uint32_t C = c_hi; C >= 1; --C) {
    const uint32_t rpr = (nwin + C - 1) / C;
    // the 64 rows of a tile are C*8 bytes apart: an even C puts several lanes of a ds_write_b64 on the
    // same LDS banks (16-way for C = 16), an odd C none.  Measured: 2-way is nearly free (C = 18 on 101 bp
    // +7 % over C = 15), 4-way is not (C = 20 on 50 bp -12 % against two runs of 10)
    uint32_t g = 2 * C, ways = 1;
    while (ways < 32 && (g & 1) == 0) { g >>= 1; ways <<= 1; }
    ways = ways > 2 ? ways / 2 : 1;
    const double conflict = no_tile || ways <= 1 ? 0.0 : ways == 2 ? 0.05 : ways == 4 ? 0.45 : 1.0;
    bool scan = false;
    const double first = any_k ? fw_cost(C, &scan) / 35.0 : 2.0 + 0.5 * ntab; // in roll steps (35 instructions each)
    const double cost = (double)rpr * (first + (C - 1) * (1.0 + conflict)) / nwin;
    if (cost < best_cost - 1e-9) { best_cost = cost; best = C; }
  }
  if (force_c >= 1 && force_c <= 31 && force_c <= nwin) best = force_c;
  if (const uint32_t d = c->tune.run_len) // A/B override
    if (d >= 1 && d <= 31 && d <= nwin) best = d; // (the model itself stays at <= 16)
  if (best == 0) return false;
  p->C = best;
  p->rpr = (nwin + best - 1) / best;
  p->last_start = nwin - best;
  p->nw = kmer_nw(k);
  p->tile_u64 = 64 * best + 128;
  const uint64_t slab_vecs = slab_dwords(best);
  uint32_t bd = (uint32_t)slab_vecs + p->nw + 6;
  bd = (bd + 3u) & ~3u;
  p->bits_dwords = bd;
  p->dword_tail = (15 + slab_bytes_of(best) <= 1280) ? 1u : 0u;
  // k beyond the position tables: which first window (first_window.hpp), and the scan's {U, V} per word of the slab
  p->fw_scan = p->uw_dwords = 0;
  if (any_k) {
    bool scan = false;
    (void)fw_cost(best, &scan);
    if (bd >= 65536u) scan = false; // (positions of the scan are folded below 2^20)
    p->fw_scan = scan ? 1u : 0u;
    if (scan) p->uw_dwords = 4u * bd;
  }
  // forward-half tables (kmer_runs_gen_kernel FH): the dense pass of k = 49 ... 64, and of every k within the position
  // tables when there are several hashes per k-mer (16 waves wanted).  Half the table bytes -- and, several hashes per
  // k-mer, no alignment slack in the tile (the copy-out shifts nothing there) -- are 12 waves per CU instead of 8 on the
  // reference's benchmark shape (100 bp, k = 64, m = 3) and 16 instead of 13 at k = 31
  p->fh = (!gaps_ok && !no_tile && !any_k && (p->nw == 4 || m > 1) && !c->tune.no_fh) ? 1u : 0u;
  if (m > 1 && !gaps_ok && !no_tile) p->tile_u64 = 64 * best + 2; // (the dense copy-out shifts the tile only for m = 1)
  const size_t fixed = (size_t)kmer_ntab(k) * (p->fh ? 2048 : 4096) + 256 + 64;
  const size_t per_wave = (size_t)p->tile_u64 * 8 + (size_t)bd * 4 + (size_t)p->uw_dwords * 4;
  const size_t cap = (c->lds_max < 160 * 1024 ? c->lds_max : 160 * 1024) - 512;
  // m = 1: 12 waves per CU (3 per SIMD) measured +3 ... +9 % over 8 on 150 / 151 bp, 16 no better (round 2)
  uint32_t w_max = m == 1 ? 12 : 16;
  if (c->tune.waves) w_max = c->tune.waves;
  for (uint32_t w = w_max; w >= 1; --w)
    if (fixed + per_wave * w <= cap) {
      p->waves = w;
      p->lds = fixed + per_wave * w;
      return true;
    }
  return false;
}

// N-aware run-split path for fixed-length reads: count pass -> scan -> compact hash pass
// (kmer_runs_gen_kernel.hpp, NA = true).  Same geometry as the dense general kernel.
bool kmer_na_plan(const nthip_ctx* c, uint32_t len, uint32_t stride, uint32_t k, uint32_t m, bool want_pos,
                  NaPlan* p, uint32_t register_sink_u64, uint32_t force_c, uint32_t model_cap)
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
  const size_t per_wave = (size_t)p->tile_u64 * 8 + ((size_t)p->ptile_dwords + g.bits_dwords + p->vbits_dwords + g.uw_dwords) * 4;
  const size_t cap = (c->lds_max < 160 * 1024 ? c->lds_max : 160 * 1024) - 512;
  // (tools/na_waves.sh, batches with an N in one read of 1000: 12 waves per CU are 13-14 % ahead of 8 on 150 / 151 bp, k = 31,
  //  one hash; with several hashes per k-mer or longer k-mers 8 stay level or ahead)
  uint32_t w_max = register_sink_u64 ? 16 : (m == 1 && k <= 32) ? 12 : 8;
  if (c->tune.na_waves) w_max = c->tune.na_waves;
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
  ga.tile_map = c->tune.has_tile_map ? c->tune.tile_map : 0xFFFFFFFFu;
  ga.fw_scan = g.fw_scan;
  ga.uw_dwords = g.uw_dwords;
  ga.fh = g.fh;
  memcpy(ga.tab, consts.tab, sizeof ga.tab);
  memcpy(ga.mult, consts.mult, sizeof ga.mult);
}

} // namespace host
} // namespace ntamd
