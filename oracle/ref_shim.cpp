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

#include <cstdint>
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

int ref_max_threads()
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

} // extern "C"
