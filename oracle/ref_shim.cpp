// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A thin extern "C" driver around the REAL reference library.  oracle/Makefile
// compiles this file together with /root/reference/src/{kmer,seed}.cpp (read
// where they lie, never copied) into oracle/_ref/libnthash_ref.so.  The shim
// contains no hashing logic of its own: it only constructs the reference's
// iterator classes and records what they return, so that
//   (a) oracle/nthash_oracle.c can be validated against the true reference,
//   (b) golden fixtures can be generated (tests/golden/gen_golden.py), and
//   (c) bench.py can time the true reference as the CPU baseline.
#include <nthash/nthash.hpp>

#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

extern "C" {

const char* ref_fn_name() { return nthash::NTHASH_FN_NAME; }

// for each read: NtHash h(seq, len, m, k); while (h.roll()) record
uint64_t ref_kmer_batch(const char* seqs, const uint64_t* offsets, uint64_t n_reads,
                        unsigned k, unsigned m, uint64_t* hashes, uint32_t* pos,
                        uint64_t* fwd, uint64_t* rev, uint64_t* counts)
{
  uint64_t total = 0;
  for (uint64_t r = 0; r < n_reads; r++) {
    const char* s = seqs + offsets[r];
    const size_t len = offsets[r + 1] - offsets[r];
    uint64_t n = 0;
    if (k != 0 && len >= k) { // the reference would exit(1) otherwise
      nthash::NtHash h(s, len, (uint8_t)m, (uint16_t)k);
      while (h.roll()) {
        if (hashes) std::memcpy(hashes + (total + n) * m, h.hashes(), 8 * m);
        if (pos) pos[total + n] = (uint32_t)h.get_pos();
        if (fwd) fwd[total + n] = h.get_forward_hash();
        if (rev) rev[total + n] = h.get_reverse_hash();
        n++;
      }
    }
    if (counts) counts[r] = n;
    total += n;
  }
  return total;
}

uint64_t ref_seed_batch(const char* seqs, const uint64_t* offsets, uint64_t n_reads,
                        const char* const* seeds, unsigned n_seeds, unsigned k,
                        unsigned m2, uint64_t* hashes, uint32_t* pos, uint64_t* counts)
{
  std::vector<std::string> sv;
  for (unsigned i = 0; i < n_seeds; i++) sv.emplace_back(seeds[i]);
  const size_t per = (size_t)n_seeds * m2;
  uint64_t total = 0;
  for (uint64_t r = 0; r < n_reads; r++) {
    const char* s = seqs + offsets[r];
    const size_t len = offsets[r + 1] - offsets[r];
    uint64_t n = 0;
    if (k != 0 && len >= k) {
      nthash::SeedNtHash h(s, len, sv, (uint8_t)m2, (uint16_t)k);
      while (h.roll()) {
        if (hashes) std::memcpy(hashes + (total + n) * per, h.hashes(), 8 * per);
        if (pos) pos[total + n] = (uint32_t)h.get_pos();
        n++;
      }
    }
    if (counts) counts[r] = n;
    total += n;
  }
  return total;
}

// ---- iterator "scripts": run a sequence of API calls, record everything ----
// ops: 'r' roll, 'b' roll_back, 'p' peek, 'q' peek_back, 'P'+c peek(c),
// 'Q'+c peek_back(c).  Per op: ret, pos, fwd, rev, m hashes.
static size_t n_ops_of(const char* ops, size_t ops_len)
{
  size_t n = 0;
  for (size_t i = 0; i < ops_len; i++) {
    if (ops[i] == 'P' || ops[i] == 'Q' || ops[i] == 'R' || ops[i] == 'B') i++;
    n++;
  }
  return n;
}

uint64_t ref_nthash_script(const char* seq, uint64_t len, unsigned m, unsigned k,
                           uint64_t pos0, const char* ops, uint64_t ops_len,
                           int32_t* ret, uint64_t* pos, uint64_t* fwd, uint64_t* rev,
                           uint64_t* hashes)
{
  nthash::NtHash h(seq, len, (uint8_t)m, (uint16_t)k, pos0);
  uint64_t n = 0;
  for (uint64_t i = 0; i < ops_len; i++, n++) {
    bool ok = false;
    switch (ops[i]) {
      case 'r': ok = h.roll(); break;
      case 'b': ok = h.roll_back(); break;
      case 'p': ok = h.peek(); break;
      case 'q': ok = h.peek_back(); break;
      case 'P': ok = h.peek(ops[++i]); break;
      case 'Q': ok = h.peek_back(ops[++i]); break;
      default: return n;
    }
    ret[n] = ok;
    pos[n] = h.get_pos();
    fwd[n] = h.get_forward_hash();
    rev[n] = h.get_reverse_hash();
    std::memcpy(hashes + n * m, h.hashes(), 8 * m);
  }
  (void)n_ops_of;
  return n;
}

// BlindNtHash: 'R'+c roll(c), 'B'+c roll_back(c), 'P'+c peek(c), 'Q'+c peek_back(c)
uint64_t ref_blind_script(const char* seq, unsigned m, unsigned k, int64_t pos0,
                          const char* ops, uint64_t ops_len, int64_t* pos,
                          uint64_t* fwd, uint64_t* rev, uint64_t* hashes)
{
  nthash::BlindNtHash h(seq, (uint8_t)m, (uint16_t)k, (ssize_t)pos0);
  uint64_t n = 0;
  // record the state right after construction as entry 0
  pos[n] = h.get_pos(); fwd[n] = h.get_forward_hash(); rev[n] = h.get_reverse_hash();
  std::memcpy(hashes + n * m, h.hashes(), 8 * m);
  n++;
  for (uint64_t i = 0; i + 1 < ops_len; i += 2, n++) {
    switch (ops[i]) {
      case 'R': h.roll(ops[i + 1]); break;
      case 'B': h.roll_back(ops[i + 1]); break;
      case 'P': h.peek(ops[i + 1]); break;
      case 'Q': h.peek_back(ops[i + 1]); break;
      default: return n;
    }
    pos[n] = h.get_pos(); fwd[n] = h.get_forward_hash(); rev[n] = h.get_reverse_hash();
    std::memcpy(hashes + n * m, h.hashes(), 8 * m);
  }
  return n;
}

// SeedNtHash script; per op: ret, pos, per-seed fwd/rev, n_seeds*m2 hashes
uint64_t ref_seed_script(const char* seq, uint64_t len, const char* const* seeds,
                         unsigned n_seeds, unsigned m2, unsigned k, uint64_t pos0,
                         const char* ops, uint64_t ops_len, int32_t* ret, uint64_t* pos,
                         uint64_t* fwd, uint64_t* rev, uint64_t* hashes)
{
  std::vector<std::string> sv;
  for (unsigned i = 0; i < n_seeds; i++) sv.emplace_back(seeds[i]);
  nthash::SeedNtHash h(seq, len, sv, (uint8_t)m2, (uint16_t)k, pos0);
  const size_t per = (size_t)n_seeds * m2;
  uint64_t n = 0;
  for (uint64_t i = 0; i < ops_len; i++, n++) {
    bool ok = false;
    switch (ops[i]) {
      case 'r': ok = h.roll(); break;
      case 'b': ok = h.roll_back(); break;
      case 'p': ok = h.peek(); break;
      case 'q': ok = h.peek_back(); break;
      case 'P': ok = h.peek(ops[++i]); break;
      case 'Q': ok = h.peek_back(ops[++i]); break;
      default: return n;
    }
    ret[n] = ok;
    pos[n] = h.get_pos();
    std::memcpy(fwd + n * n_seeds, h.get_forward_hash(), 8 * n_seeds);
    std::memcpy(rev + n * n_seeds, h.get_reverse_hash(), 8 * n_seeds);
    std::memcpy(hashes + n * per, h.hashes(), 8 * per);
  }
  return n;
}

// BlindSeedNtHash: 'R'+c roll(c), 'B'+c roll_back(c); entry 0 = after ctor
uint64_t ref_blindseed_script(const char* seq, const char* const* seeds, unsigned n_seeds,
                              unsigned m2, unsigned k, int64_t pos0, const char* ops,
                              uint64_t ops_len, int64_t* pos, uint64_t* fwd,
                              uint64_t* rev, uint64_t* hashes)
{
  std::vector<std::string> sv;
  for (unsigned i = 0; i < n_seeds; i++) sv.emplace_back(seeds[i]);
  nthash::BlindSeedNtHash h(seq, sv, (uint8_t)m2, (uint16_t)k, (ssize_t)pos0);
  const size_t per = (size_t)n_seeds * m2;
  uint64_t n = 0;
  auto rec = [&]() {
    pos[n] = h.get_pos();
    std::memcpy(fwd + n * n_seeds, h.get_forward_hash(), 8 * n_seeds);
    std::memcpy(rev + n * n_seeds, h.get_reverse_hash(), 8 * n_seeds);
    std::memcpy(hashes + n * per, h.hashes(), 8 * per);
    n++;
  };
  rec();
  for (uint64_t i = 0; i + 1 < ops_len; i += 2) {
    switch (ops[i]) {
      case 'R': h.roll(ops[i + 1]); break;
      case 'B': h.roll_back(ops[i + 1]); break;
      default: return n;
    }
    rec();
  }
  return n;
}

// parse_seeds (include/nthash/nthash.hpp:59-60): flattened output
uint64_t ref_parse_seeds(const char* seed, uint32_t* out, uint64_t cap)
{
  auto v = nthash::parse_seeds({ std::string(seed) });
  uint64_t n = 0;
  for (unsigned p : v[0]) { if (n < cap) out[n] = p; n++; }
  return n;
}

// ---- CPU baseline: the reference used the way examples/benchmark.cpp uses it
// (iterator per read, consume every hash).  threads<=1: single thread;
// otherwise an OpenMP parallel-for over reads ADDED BY THIS HARNESS (the
// reference itself is single-threaded).
uint64_t ref_bench_kmer(const char* seqs, uint64_t n_reads, unsigned len, unsigned k,
                        unsigned m, int threads, uint64_t* n_kmers)
{
  uint64_t acc = 0, cnt = 0;
#ifdef _OPENMP
  if (threads > 1) omp_set_num_threads(threads);
#pragma omp parallel for schedule(static) reduction(+ : acc, cnt) if (threads > 1)
#endif
  for (int64_t r = 0; r < (int64_t)n_reads; r++) {
    nthash::NtHash h(seqs + (uint64_t)r * len, len, (uint8_t)m, (uint16_t)k);
    while (h.roll()) {
      const uint64_t* hv = h.hashes();
      for (unsigned j = 0; j < m; j++) acc += hv[j];
      cnt++;
    }
  }
  *n_kmers = cnt;
  return acc;
}

uint64_t ref_bench_seed(const char* seqs, uint64_t n_reads, unsigned len,
                        const char* const* seeds, unsigned n_seeds, unsigned k,
                        unsigned m2, int threads, uint64_t* n_kmers)
{
  std::vector<std::string> sv;
  for (unsigned i = 0; i < n_seeds; i++) sv.emplace_back(seeds[i]);
  const unsigned per = n_seeds * m2;
  uint64_t acc = 0, cnt = 0;
#ifdef _OPENMP
  if (threads > 1) omp_set_num_threads(threads);
#pragma omp parallel for schedule(static) reduction(+ : acc, cnt) if (threads > 1)
#endif
  for (int64_t r = 0; r < (int64_t)n_reads; r++) {
    nthash::SeedNtHash h(seqs + (uint64_t)r * len, len, sv, (uint8_t)m2, (uint16_t)k);
    while (h.roll()) {
      const uint64_t* hv = h.hashes();
      for (unsigned j = 0; j < per; j++) acc += hv[j];
      cnt++;
    }
  }
  *n_kmers = cnt;
  return acc;
}

// ---- whole-workload helpers on the counter-based synthetic reads (read r, 32-base word w ->
// splitmix64(seed + r*W + w), 2 bits per base).  The generator is ours; every hash comes from the reference.
static inline uint64_t shim_splitmix64(uint64_t x)
{
  uint64_t z = x + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static inline void shim_synth_read(char* out, uint64_t r, unsigned len, uint64_t seed)
{
  static const char ACGT[4] = { 'A', 'C', 'G', 'T' };
  const uint64_t W = (len + 31) / 32;
  for (uint64_t w = 0; w < W; w++) {
    const uint64_t x = shim_splitmix64(seed + r * W + w);
    for (unsigned j = 0; j < 32 && w * 32 + j < len; j++) out[w * 32 + j] = ACGT[(x >> (2 * j)) & 3];
  }
}

// wrapping sum and XOR of EVERY hash the reference emits for reads [first_read, first_read + n_reads) of the
// synthetic workload; seeds == NULL: NtHash(k, m), else SeedNtHash(seeds, m per seed).  Reads are generated by
// the thread that hashes them (no 15 GB buffer), OpenMP over reads.
void ref_synth_checksum(uint64_t first_read, uint64_t n_reads, unsigned len, uint64_t seed,
                        const char* const* seeds, unsigned n_seeds, unsigned k, unsigned m, int threads,
                        uint64_t* sum_out, uint64_t* xor_out, uint64_t* total_out)
{
  std::vector<std::string> sv;
  for (unsigned i = 0; i < n_seeds; i++) sv.emplace_back(seeds[i]);
  const unsigned per = n_seeds ? n_seeds * m : m;
  uint64_t sum = 0, xr = 0, cnt = 0;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel reduction(+ : sum, cnt) reduction(^ : xr)
#endif
  {
    std::vector<char> buf(len + 1);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int64_t i = 0; i < (int64_t)n_reads; i++) {
      shim_synth_read(buf.data(), first_read + (uint64_t)i, len, seed);
      if (n_seeds) {
        nthash::SeedNtHash h(buf.data(), len, sv, (uint8_t)m, (uint16_t)k);
        while (h.roll()) {
          const uint64_t* hv = h.hashes();
          for (unsigned j = 0; j < per; j++) { sum += hv[j]; xr ^= hv[j]; }
          cnt++;
        }
      } else {
        nthash::NtHash h(buf.data(), len, (uint8_t)m, (uint16_t)k);
        while (h.roll()) {
          const uint64_t* hv = h.hashes();
          for (unsigned j = 0; j < per; j++) { sum += hv[j]; xr ^= hv[j]; }
          cnt++;
        }
      }
    }
  }
  *sum_out = sum;
  *xor_out = xr;
  *total_out = cnt;
}

// Variable-length version of the synthetic workload (bench.py's "var" line): read r is the first len_r bytes of the
// synthetic read r of length len_max, len_r = len_min + x % (len_max - len_min + 1) with x = splitmix64(seed + 0xABCDEF + r);
// one read in ~997 ((x >> 32) % 997 == 0) has its byte (x >> 16) % len_r replaced by 'N'.  Every hash from the reference.
void ref_synth_var_checksum(uint64_t first_read, uint64_t n_reads, unsigned len_min, unsigned len_max, uint64_t seed,
                            unsigned k, unsigned m, int threads, uint64_t* sum_out, uint64_t* xor_out,
                            uint64_t* total_out)
{
  uint64_t sum = 0, xr = 0, cnt = 0;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel reduction(+ : sum, cnt) reduction(^ : xr)
#endif
  {
    std::vector<char> buf(len_max + 1);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int64_t i = 0; i < (int64_t)n_reads; i++) {
      const uint64_t r = first_read + (uint64_t)i;
      shim_synth_read(buf.data(), r, len_max, seed);
      const uint64_t x = shim_splitmix64(seed + 0xABCDEFULL + r);
      const unsigned len = len_min + (unsigned)(x % (uint64_t)(len_max - len_min + 1));
      if ((x >> 32) % 997 == 0) buf[(x >> 16) % len] = 'N';
      if (len < k) continue; // (the reference refuses sequences shorter than k)
      nthash::NtHash h(buf.data(), len, (uint8_t)m, (uint16_t)k);
      while (h.roll()) {
        const uint64_t* hv = h.hashes();
        for (unsigned j = 0; j < m; j++) { sum += hv[j]; xr ^= hv[j]; }
        cnt++;
      }
    }
  }
  *sum_out = sum;
  *xor_out = xr;
  *total_out = cnt;
}

// CPU baseline, timed inside: the reads are generated (untimed) by the threads that will hash them -- first touch
// on their own NUMA node, thread pool warm -- then the same static partition is hashed the way
// examples/benchmark.cpp uses the library (iterator per read, every hash consumed).  Returns seconds.
double ref_bench_synth(uint64_t first_read, uint64_t n_reads, unsigned len, uint64_t seed,
                       const char* const* seeds, unsigned n_seeds, unsigned k, unsigned m, int threads,
                       int repeats, uint64_t* n_kmers, uint64_t* acc_out, int* threads_used)
{
  std::vector<std::string> sv;
  for (unsigned i = 0; i < n_seeds; i++) sv.emplace_back(seeds[i]);
  const unsigned per = n_seeds ? n_seeds * m : m;
  char* data = (char*)malloc((size_t)n_reads * len + 64); // untouched pages: placed by the first writer
  if (!data) return -1.0;
  uint64_t acc = 0, cnt = 0;
  int used = 1;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
  {
#pragma omp single
    used = omp_get_num_threads();
#pragma omp for schedule(static)
    for (int64_t i = 0; i < (int64_t)n_reads; i++) shim_synth_read(data + (uint64_t)i * len, first_read + (uint64_t)i, len, seed);
  }
  const double t0 = omp_get_wtime();
#else
  for (uint64_t i = 0; i < n_reads; i++) shim_synth_read(data + i * len, first_read + i, len, seed);
  const auto c0 = std::chrono::steady_clock::now();
#endif
  for (int rep = 0; rep < (repeats > 0 ? repeats : 1); rep++) { // the same reads again: a longer, steadier sample
#ifdef _OPENMP
#pragma omp parallel for schedule(static) reduction(+ : acc, cnt)
#endif
  for (int64_t r = 0; r < (int64_t)n_reads; r++) {
    if (n_seeds) {
      nthash::SeedNtHash h(data + (uint64_t)r * len, len, sv, (uint8_t)m, (uint16_t)k);
      while (h.roll()) {
        const uint64_t* hv = h.hashes();
        for (unsigned j = 0; j < per; j++) acc += hv[j];
        cnt++;
      }
    } else {
      nthash::NtHash h(data + (uint64_t)r * len, len, (uint8_t)m, (uint16_t)k);
      while (h.roll()) {
        const uint64_t* hv = h.hashes();
        for (unsigned j = 0; j < per; j++) acc += hv[j];
        cnt++;
      }
    }
  }
  }
#ifdef _OPENMP
  const double sec = omp_get_wtime() - t0;
#else
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count();
#endif
  free(data);
  *n_kmers = cnt;
  *acc_out = acc;
  *threads_used = used;
  return sec;
}

int ref_max_threads()
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

} // extern "C"
