/*
 * oracle/nthash_oracle.c -- TEST INFRASTRUCTURE ONLY (see nthash_oracle.h).
 *
 * CPU restatement of the ntHash v2 hot path, written from the mathematical
 * specification and the reference's control flow.  Every
 * function cites the reference file:line whose behaviour it restates.  The
 * arithmetic is deliberately done a different way from the reference (closed
 * form split-rotates and direct XOR sums instead of lookup tables), so that a
 * transcription error in either shows up as a mismatch against
 * oracle/_ref/libnthash_ref.so and the golden vectors.
 */
#include "nthash_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* primitives                                                               */
/* ------------------------------------------------------------------------ */

#define NTO_MASK31 0x7FFFFFFFULL
#define NTO_MASK33 0x1FFFFFFFFULL

/* src/internal.hpp:124-128 */
static const uint64_t NTO_A = 0x3c8bfbb395c60474ULL;
static const uint64_t NTO_C = 0x3193c18562a02b4cULL;
static const uint64_t NTO_G = 0x20323ed082572324ULL;
static const uint64_t NTO_T = 0x295549f54be24456ULL;

/* src/internal.hpp:91,94 */
#define NTO_MULTISHIFT 27
static const uint64_t NTO_MULTISEED = 0x90b45d39fb6da1faULL;

static uint64_t rotl_bits(uint64_t v, unsigned width, unsigned d)
{
  const uint64_t mask = (width == 64) ? ~0ULL : ((1ULL << width) - 1);
  d %= width;
  if (d == 0) return v & mask;
  return ((v << d) | (v >> (width - d))) & mask;
}

/* internal.hpp:343-348: srol_table(c, d) == srol^d(SEED_TAB[c]) for any d.
 * Closed form: the top 31 bits and the low 33 bits rotate independently. */
uint64_t nto_srol_n(uint64_t x, unsigned d)
{
  const uint64_t hi = rotl_bits(x >> 33, 31, d);
  const uint64_t lo = rotl_bits(x & NTO_MASK33, 33, d);
  return (hi << 33) | lo;
}

/* internal.hpp:41-47 */
uint64_t nto_srol(uint64_t x) { return nto_srol_n(x, 1); }

/* internal.hpp:83-88: inverse of srol = rotate each part right by one */
uint64_t nto_sror(uint64_t x)
{
  const uint64_t hi = rotl_bits(x >> 33, 31, 30);
  const uint64_t lo = rotl_bits(x & NTO_MASK33, 33, 32);
  return (hi << 33) | lo;
}

/* internal.hpp:132-165: SEED_TAB indexed by the raw byte.  Letters ACGTU in
 * either case carry their seed.  The table also has entries at bytes 1,3,4,5,7
 * (the complement slots reached through `c & CP_OFF`); they are reproduced
 * because SeedNtHash feeds `c & 7` of *any* byte through them (seed.cpp:156). */
static uint64_t seed_tab(unsigned char c)
{
  switch (c) {
    case 'A': case 'a': return NTO_A;
    case 'C': case 'c': return NTO_C;
    case 'G': case 'g': return NTO_G;
    case 'T': case 't': case 'U': case 'u': return NTO_T;
    case 1: return NTO_T; /* 'A' & 7 -> complement of A */
    case 3: return NTO_G; /* 'C' & 7 */
    case 4: return NTO_A; /* 'T' & 7 */
    case 5: return NTO_A; /* 'U' & 7 */
    case 7: return NTO_C; /* 'G' & 7 */
    default: return 0;    /* SEED_N */
  }
}

uint64_t nto_seed_fwd(unsigned char c) { return seed_tab(c); }
uint64_t nto_seed_rc(unsigned char c) { return seed_tab((unsigned char)(c & 7)); }

static int is_valid(unsigned char c) { return seed_tab(c) != 0; }

/* internal.hpp:104-118 */
void nto_extend(uint64_t fwd, uint64_t rev, unsigned k, unsigned m, uint64_t* h)
{
  const uint64_t base = (uint64_t)k * NTO_MULTISEED;
  h[0] = fwd + rev; /* canonical(), internal.hpp:24-29 */
  for (unsigned i = 1; i < m; i++) {
    uint64_t t = h[0] * ((uint64_t)i ^ base);
    t ^= t >> NTO_MULTISHIFT;
    h[i] = t;
  }
}

/* kmer.cpp:43-73 computes F0 through tetramer tables; the value is
 * XOR_i srol^{k-1-i}(S[s_i]). */
uint64_t nto_base_fwd(const char* s, unsigned k)
{
  uint64_t h = 0;
  for (unsigned i = 0; i < k; i++)
    h ^= nto_srol_n(nto_seed_fwd((unsigned char)s[i]), k - 1 - i);
  return h;
}

/* kmer.cpp:123-152: R0 = XOR_i srol^{i}(S[comp s_i]) */
uint64_t nto_base_rev(const char* s, unsigned k)
{
  uint64_t h = 0;
  for (unsigned i = 0; i < k; i++)
    h ^= nto_srol_n(nto_seed_rc((unsigned char)s[i]), i);
  return h;
}

/* kmer.cpp:84-94 */
static uint64_t next_fwd(uint64_t f, unsigned k, unsigned char out, unsigned char in)
{
  return nto_srol(f) ^ nto_seed_fwd(in) ^ nto_srol_n(nto_seed_fwd(out), k);
}
/* kmer.cpp:164-174 */
static uint64_t next_rev(uint64_t r, unsigned k, unsigned char out, unsigned char in)
{
  return nto_sror(r ^ nto_srol_n(nto_seed_rc(in), k) ^ nto_seed_rc(out));
}
/* kmer.cpp:104-114 */
static uint64_t prev_fwd(uint64_t f, unsigned k, unsigned char out, unsigned char in)
{
  return nto_sror(f ^ nto_srol_n(nto_seed_fwd(in), k) ^ nto_seed_fwd(out));
}
/* kmer.cpp:184-194 */
static uint64_t prev_rev(uint64_t r, unsigned k, unsigned char out, unsigned char in)
{
  return nto_srol(r) ^ nto_seed_rc(in) ^ nto_srol_n(nto_seed_rc(out), k);
}

/* ------------------------------------------------------------------------ */
/* NtHash iterator (kmer.cpp:200-336)                                        */
/* ------------------------------------------------------------------------ */

int nto_nthash_init(nto_nthash* it, const char* seq, size_t len, unsigned m,
                    unsigned k, size_t pos, uint64_t* hbuf)
{
  /* kmer.cpp:212-225: the three raise_error conditions */
  if (k == 0 || len < k || pos > len - k) return -1;
  it->seq = seq; it->len = len; it->k = k; it->m = m; it->pos = pos;
  it->initialized = 0; it->fwd = 0; it->rev = 0; it->h = hbuf;
  return 0;
}

/* kmer.cpp:25-35 restated on a window that may touch index `len`; that one
 * byte is treated as invalid (a std::string's terminator).  The reference
 * reads it (quirk Q6); with a NUL-terminated buffer the behaviour is this. */
static int window_invalid(const nto_nthash* it, size_t at, size_t* bad)
{
  for (size_t i = it->k; i-- > 0;) {
    const size_t idx = at + i;
    if (idx >= it->len || !is_valid((unsigned char)it->seq[idx])) {
      *bad = i;
      return 1;
    }
  }
  return 0;
}

/* kmer.cpp:228-244 */
static int nthash_do_init(nto_nthash* it)
{
  size_t bad = 0;
  while (it->pos <= it->len - it->k + 1 && window_invalid(it, it->pos, &bad))
    it->pos += bad + 1;
  if (it->pos > it->len - it->k) return 0;
  it->fwd = nto_base_fwd(it->seq + it->pos, it->k);
  it->rev = nto_base_rev(it->seq + it->pos, it->k);
  nto_extend(it->fwd, it->rev, it->k, it->m, it->h);
  it->initialized = 1;
  return 1;
}

/* kmer.cpp:246-264 */
int nto_nthash_roll(nto_nthash* it)
{
  if (!it->initialized) return nthash_do_init(it);
  if (it->pos >= it->len - it->k) return 0;
  const unsigned char in = (unsigned char)it->seq[it->pos + it->k];
  if (!is_valid(in)) {
    it->pos += it->k;
    return nthash_do_init(it);
  }
  const unsigned char out = (unsigned char)it->seq[it->pos];
  it->fwd = next_fwd(it->fwd, it->k, out, in);
  it->rev = next_rev(it->rev, it->k, out, in);
  nto_extend(it->fwd, it->rev, it->k, it->m, it->h);
  it->pos++;
  return 1;
}

/* kmer.cpp:266-287 */
int nto_nthash_roll_back(nto_nthash* it)
{
  if (!it->initialized) return nthash_do_init(it);
  if (it->pos == 0) return 0;
  const unsigned char in = (unsigned char)it->seq[it->pos - 1];
  if (!is_valid(in) && it->pos >= it->k) {
    it->pos -= it->k;
    return nthash_do_init(it);
  }
  if (!is_valid(in)) return 0;
  const unsigned char out = (unsigned char)it->seq[it->pos + it->k - 1];
  it->fwd = prev_fwd(it->fwd, it->k, out, in);
  it->rev = prev_rev(it->rev, it->k, out, in);
  nto_extend(it->fwd, it->rev, it->k, it->m, it->h);
  it->pos--;
  return 1;
}

/* kmer.cpp:298-311 */
int nto_nthash_peek_char(nto_nthash* it, char c)
{
  if (!it->initialized) return nthash_do_init(it);
  if (!is_valid((unsigned char)c)) return 0;
  const unsigned char out = (unsigned char)it->seq[it->pos];
  nto_extend(next_fwd(it->fwd, it->k, out, (unsigned char)c),
             next_rev(it->rev, it->k, out, (unsigned char)c), it->k, it->m, it->h);
  return 1;
}

/* kmer.cpp:289-296 */
int nto_nthash_peek(nto_nthash* it)
{
  if (it->pos >= it->len - it->k) return 0;
  return nto_nthash_peek_char(it, it->seq[it->pos + it->k]);
}

/* kmer.cpp:322-336 */
int nto_nthash_peek_back_char(nto_nthash* it, char c)
{
  if (!it->initialized) return nthash_do_init(it);
  if (!is_valid((unsigned char)c)) return 0;
  const unsigned char out = (unsigned char)it->seq[it->pos + it->k - 1];
  nto_extend(prev_fwd(it->fwd, it->k, out, (unsigned char)c),
             prev_rev(it->rev, it->k, out, (unsigned char)c), it->k, it->m, it->h);
  return 1;
}

/* kmer.cpp:313-320 */
int nto_nthash_peek_back(nto_nthash* it)
{
  if (it->pos == 0) return 0;
  return nto_nthash_peek_back_char(it, it->seq[it->pos - 1]);
}

/* the caller loop of examples/benchmark.cpp:34-39, generalised to a batch */
uint64_t nto_kmer_batch(const char* seqs, const uint64_t* offsets,
                        uint64_t n_reads, unsigned k, unsigned m,
                        uint64_t* hashes, uint32_t* pos, uint64_t* fwd,
                        uint64_t* rev, uint64_t* counts)
{
  uint64_t total = 0;
  uint64_t* hbuf = (uint64_t*)malloc(sizeof(uint64_t) * (m ? m : 1));
  for (uint64_t r = 0; r < n_reads; r++) {
    const char* s = seqs + offsets[r];
    const size_t len = (size_t)(offsets[r + 1] - offsets[r]);
    uint64_t n = 0;
    nto_nthash it;
    if (nto_nthash_init(&it, s, len, m, k, 0, hbuf) == 0) {
      while (nto_nthash_roll(&it)) {
        if (hashes) memcpy(hashes + (total + n) * m, hbuf, sizeof(uint64_t) * m);
        if (pos) pos[total + n] = (uint32_t)it.pos;
        if (fwd) fwd[total + n] = it.fwd;
        if (rev) rev[total + n] = it.rev;
        n++;
      }
    }
    if (counts) counts[r] = n;
    total += n;
  }
  free(hbuf);
  return total;
}

/* ------------------------------------------------------------------------ */
/* spaced seeds (seed.cpp)                                                   */
/* ------------------------------------------------------------------------ */

/* seed.cpp:19-66.  The reference appends a sentinel that differs from the last
 * character and walks the runs: a care run ends at a literal '0', a don't-care
 * run ends at a literal '1' (any other character extends the current run).
 * Runs of length one become monomers, longer runs become blocks.  If the
 * don't-care description is cheaper (2*blocks + monos + 2 < 2*blocks + monos
 * of the care description) it is used instead, together with the whole-k-mer
 * block [0,k) pushed LAST. */
unsigned nto_get_blocks(const char* seed, unsigned k, unsigned* blocks,
                        unsigned* monos, unsigned* n_monos)
{
  unsigned* cb = (unsigned*)malloc(sizeof(unsigned) * 2 * (k + 2));
  unsigned* ib = (unsigned*)malloc(sizeof(unsigned) * 2 * (k + 2));
  unsigned* cm = (unsigned*)malloc(sizeof(unsigned) * (k + 2));
  unsigned* im = (unsigned*)malloc(sizeof(unsigned) * (k + 2));
  unsigned ncb = 0, nib = 0, ncm = 0, nim = 0;
  const char sentinel = (seed[k - 1] == '1') ? '0' : '1';
  int in_care = (seed[0] == '1');
  unsigned run_start = 0;
  for (unsigned p = 0; p <= k; p++) {
    const char ch = (p < k) ? seed[p] : sentinel;
    if (in_care && ch == '0') {
      if (p - run_start == 1) cm[ncm++] = run_start;
      else { cb[2 * ncb] = run_start; cb[2 * ncb + 1] = p; ncb++; }
      run_start = p;
      in_care = 0;
    } else if (!in_care && ch == '1') {
      if (p - run_start == 1) im[nim++] = run_start;
      else { ib[2 * nib] = run_start; ib[2 * nib + 1] = p; nib++; }
      run_start = p;
      in_care = 1;
    }
  }
  const unsigned cost_care = 2 * ncb + ncm;
  const unsigned cost_ignore = 2 * nib + nim + 2;
  unsigned nb;
  if (cost_ignore < cost_care) {
    memcpy(blocks, ib, sizeof(unsigned) * 2 * nib);
    blocks[2 * nib] = 0; blocks[2 * nib + 1] = k;
    nb = nib + 1;
    memcpy(monos, im, sizeof(unsigned) * nim);
    *n_monos = nim;
  } else {
    memcpy(blocks, cb, sizeof(unsigned) * 2 * ncb);
    nb = ncb;
    memcpy(monos, cm, sizeof(unsigned) * ncm);
    *n_monos = ncm;
  }
  free(cb); free(ib); free(cm); free(im);
  return nb;
}

typedef struct {
  unsigned nb, nm;
  unsigned* blocks; /* 2*nb */
  unsigned* monos;
  unsigned char* parity; /* k entries: 1 if the position contributes */
} seed_desc;

static void seed_desc_make(seed_desc* d, const char* seed, unsigned k)
{
  d->blocks = (unsigned*)malloc(sizeof(unsigned) * 2 * (k + 2));
  d->monos = (unsigned*)malloc(sizeof(unsigned) * (k + 2));
  d->parity = (unsigned char*)calloc(k, 1);
  d->nb = nto_get_blocks(seed, k, d->blocks, d->monos, &d->nm);
  /* every block / monomer XORs its positions in (seed.cpp:149-164); with the
   * ignore description the don't-care positions appear twice and cancel. */
  for (unsigned b = 0; b < d->nb; b++)
    for (unsigned p = d->blocks[2 * b]; p < d->blocks[2 * b + 1]; p++) d->parity[p] ^= 1;
  for (unsigned i = 0; i < d->nm; i++) d->parity[d->monos[i]] ^= 1;
}

static void seed_desc_free(seed_desc* d)
{
  free(d->blocks); free(d->monos); free(d->parity);
}

/* seed.cpp:130-175 (first window) and :177-207 (rolled window) both evaluate
 * F = XOR_{p contributing} srol^{k-1-p}(S[c_p]),  R = XOR srol^{p}(S[c_p & 7]). */
static void seed_hash_window(const seed_desc* d, const char* win, unsigned k,
                             uint64_t* f, uint64_t* r)
{
  uint64_t fh = 0, rh = 0;
  for (unsigned p = 0; p < k; p++) {
    if (!d->parity[p]) continue;
    fh ^= nto_srol_n(nto_seed_fwd((unsigned char)win[p]), k - 1 - p);
    rh ^= nto_srol_n(nto_seed_rc((unsigned char)win[p]), p);
  }
  *f = fh; *r = rh;
}

void nto_seed_window(const char* win, const char* seed, unsigned k,
                     uint64_t* fwd, uint64_t* rev)
{
  seed_desc d;
  seed_desc_make(&d, seed, k);
  seed_hash_window(&d, win, k, fwd, rev);
  seed_desc_free(&d);
}

/* BlindSeedNtHash::roll(c) / roll_back(c) from the window `kmer`, for c = A, C, G, T (seed.cpp:701-737 through the NTMSM64
 * macro, seed.cpp:177-207).  Forward, the macro rolls every block (out = old[b0], in = the window's base at b1) and adds
 * the monomers from kmer_seq[pos + 1] -- the NEW window: the result is the masked formula of kmer[1..k) + c.  Backward
 * (ntmsm64l on the deque with c pushed in front) the blocks are those of the new window c + kmer[0..k-1), but the monomers
 * are read at the same kmer_seq[pos + 1] -- now the OLD window's base at pos (seed.cpp:195-198 reused): reproduced. */
void nto_seed_extend(const char* kmer, unsigned k, const char* const* seeds, unsigned n_seeds, unsigned m2,
                     uint64_t* self, uint64_t* next, uint64_t* prev)
{
  static const char bases[4] = { 'A', 'C', 'G', 'T' };
  char* win = (char*)malloc(k + 1);
  const size_t per = (size_t)n_seeds * m2;
  for (unsigned s = 0; s < n_seeds; s++) {
    seed_desc d;
    seed_desc_make(&d, seeds[s], k);
    unsigned char* mono = (unsigned char*)calloc(k, 1);
    for (unsigned i = 0; i < d.nm; i++) mono[d.monos[i]] ^= 1;
    uint64_t f, r;
    if (self) {
      seed_hash_window(&d, kmer, k, &f, &r);
      nto_extend(f, r, k, m2, self + (size_t)s * m2);
    }
    for (unsigned b = 0; b < 4; b++) {
      if (next) {
        memcpy(win, kmer + 1, k - 1);
        win[k - 1] = bases[b];
        seed_hash_window(&d, win, k, &f, &r);
        nto_extend(f, r, k, m2, next + b * per + (size_t)s * m2);
      }
      if (prev) {
        win[0] = bases[b];
        memcpy(win + 1, kmer, k - 1);
        f = r = 0;
        for (unsigned p = 0; p < k; p++) {
          const unsigned blk = d.parity[p] ^ mono[p]; /* covered by an odd number of blocks */
          if (blk) {
            f ^= nto_srol_n(nto_seed_fwd((unsigned char)win[p]), k - 1 - p);
            r ^= nto_srol_n(nto_seed_rc((unsigned char)win[p]), p);
          }
          if (mono[p]) {
            f ^= nto_srol_n(nto_seed_fwd((unsigned char)kmer[p]), k - 1 - p);
            r ^= nto_srol_n(nto_seed_rc((unsigned char)kmer[p]), p);
          }
        }
        nto_extend(f, r, k, m2, prev + b * per + (size_t)s * m2);
      }
    }
    free(mono);
    seed_desc_free(&d);
  }
  free(win);
}

/* seed.cpp:146-158: the first-window routine fails on the first NUL byte met
 * while walking seeds -> blocks -> positions in that order. */
static int seed_first_nul(const seed_desc* ds, unsigned n_seeds, const char* win,
                          unsigned* where)
{
  for (unsigned s = 0; s < n_seeds; s++)
    for (unsigned b = 0; b < ds[s].nb; b++)
      for (unsigned p = ds[s].blocks[2 * b]; p < ds[s].blocks[2 * b + 1]; p++)
        if (win[p] == 0) { *where = p; return 1; }
  return 0;
}

static void seed_emit(const seed_desc* ds, unsigned n_seeds, const char* win,
                      unsigned k, unsigned m2, uint64_t* out)
{
  for (unsigned s = 0; s < n_seeds; s++) {
    uint64_t f, r;
    seed_hash_window(&ds[s], win, k, &f, &r);
    nto_extend(f, r, k, m2, out + (size_t)s * m2); /* seed.cpp:167-172 */
  }
}

/* seed.cpp:493-516 */
static int seed_do_init(const seed_desc* ds, unsigned n_seeds, const char* s,
                        size_t len, unsigned k, size_t* pos)
{
  unsigned where = 0;
  while (*pos < len - k + 1 && seed_first_nul(ds, n_seeds, s + *pos, &where))
    *pos += where + 1;
  return !(*pos > len - k);
}

uint64_t nto_seed_batch(const char* seqs, const uint64_t* offsets,
                        uint64_t n_reads, const char* const* seeds,
                        unsigned n_seeds, unsigned k, unsigned m2,
                        uint64_t* hashes, uint32_t* pos_out, uint64_t* counts)
{
  seed_desc* ds = (seed_desc*)malloc(sizeof(seed_desc) * n_seeds);
  for (unsigned s = 0; s < n_seeds; s++) seed_desc_make(&ds[s], seeds[s], k);
  const size_t per = (size_t)n_seeds * m2;
  uint64_t* tmp = (uint64_t*)malloc(sizeof(uint64_t) * (per ? per : 1));
  uint64_t total = 0;
  for (uint64_t r = 0; r < n_reads; r++) {
    const char* s = seqs + offsets[r];
    const size_t len = (size_t)(offsets[r + 1] - offsets[r]);
    uint64_t n = 0;
    if (k != 0 && len >= k) {
      size_t pos = 0;
      int ok = seed_do_init(ds, n_seeds, s, len, k, &pos); /* first roll() */
      while (ok) {
        seed_emit(ds, n_seeds, s + pos, k, m2, tmp);
        if (hashes) memcpy(hashes + (total + n) * per, tmp, sizeof(uint64_t) * per);
        if (pos_out) pos_out[total + n] = (uint32_t)pos;
        n++;
        /* seed.cpp:518-544 */
        if (pos >= len - k) break;
        if (!is_valid((unsigned char)s[pos + k])) {
          pos += k;
          ok = seed_do_init(ds, n_seeds, s, len, k, &pos);
        } else {
          pos++;
        }
      }
    }
    if (counts) counts[r] = n;
    total += n;
  }
  free(tmp);
  for (unsigned s = 0; s < n_seeds; s++) seed_desc_free(&ds[s]);
  free(ds);
  return total;
}

/* ------------------------------------------------------------------------ */
/* synthetic reads, checksums, timing helper                                 */
/* ------------------------------------------------------------------------ */

uint64_t nto_splitmix64(uint64_t x)
{
  uint64_t z = x + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

/* read r, 32-base word w -> splitmix64(seed + r*W + w) */
void nto_synth_reads(char* dst, uint64_t first_read, uint64_t n_reads,
                     unsigned len, uint64_t seed)
{
  static const char ACGT[4] = { 'A', 'C', 'G', 'T' };
  const uint64_t W = (len + 31) / 32;
  for (uint64_t i = 0; i < n_reads; i++) {
    const uint64_t r = first_read + i;
    char* out = dst + i * (uint64_t)len;
    for (uint64_t w = 0; w < W; w++) {
      const uint64_t x = nto_splitmix64(seed + r * W + w);
      for (unsigned j = 0; j < 32 && w * 32 + j < len; j++)
        out[w * 32 + j] = ACGT[(x >> (2 * j)) & 3];
    }
  }
}

void nto_checksum(const uint64_t* v, uint64_t n, uint64_t* sum, uint64_t* x)
{
  uint64_t s = 0, q = 0;
  for (uint64_t i = 0; i < n; i++) { s += v[i]; q ^= v[i]; }
  *sum = s; *x = q;
}

uint64_t nto_bench_kmer(const char* seqs, uint64_t n_reads, unsigned len,
                        unsigned k, unsigned m, uint64_t* n_kmers)
{
  uint64_t acc = 0, cnt = 0;
  uint64_t hbuf[256];
  for (uint64_t r = 0; r < n_reads; r++) {
    nto_nthash it;
    if (nto_nthash_init(&it, seqs + r * (uint64_t)len, len, m, k, 0, hbuf)) continue;
    while (nto_nthash_roll(&it)) {
      for (unsigned j = 0; j < m; j++) acc += hbuf[j];
      cnt++;
    }
  }
  *n_kmers = cnt;
  return acc;
}
