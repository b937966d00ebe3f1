// TEST INFRASTRUCTURE ONLY (tests/test_sanitizers.py): the host logic of the C++ facade against the C restatement
// (oracle/nthash_oracle.c), both compiled with -fsanitize=address,undefined.  Exits non-zero on the first mismatch;
// the sanitizers abort on the first bad access / UB.
#include <nthash/nthash.hpp>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

extern "C" {
#include "../../oracle/nthash_oracle.h"
}

static std::mt19937_64 rng(12345);
static int fails = 0;
#define CHECK(cond, ...) do { if (!(cond)) { std::fprintf(stderr, "MISMATCH %s:%d: ", __FILE__, __LINE__); \
  std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); if (++fails > 5) std::exit(1); } } while (0)

static std::string rand_seq(size_t len, int dirty)
{
  static const char clean[] = "ACGTacgtUu";
  static const char mixed[] = "ACGTACGTACGTNnRY-";
  std::string s(len, 'A');
  for (auto& c : s) c = dirty ? mixed[rng() % (sizeof mixed - 1)] : clean[rng() % (sizeof clean - 1)];
  return s;
}

// NtHash: full forward walk + random call sequences, against the oracle's iterator
static void test_nthash()
{
  for (int it = 0; it < 400; ++it) {
    const size_t len = 3 + rng() % 300;
    const unsigned k = 3 + rng() % (len - 2 < 70 ? len - 2 : 70);
    const unsigned m = 1 + rng() % 4;
    const std::string s = rand_seq(len, it & 1);
    std::vector<uint64_t> hb(m);
    {
      nthash::NtHash h(s, (uint8_t)m, (uint16_t)k);
      nto_nthash o;
      CHECK(nto_nthash_init(&o, s.data(), s.size(), m, k, 0, hb.data()) == 0, "init");
      for (;;) {
        const bool a = h.roll();
        const int b = nto_nthash_roll(&o);
        CHECK(a == (b != 0), "roll() return, len %zu k %u", len, k);
        if (!a) break;
        CHECK(h.get_pos() == o.pos, "pos %zu vs %zu", h.get_pos(), o.pos);
        CHECK(h.get_forward_hash() == o.fwd && h.get_reverse_hash() == o.rev, "strands at %zu", o.pos);
        CHECK(std::memcmp(h.hashes(), hb.data(), 8 * m) == 0, "hashes at %zu", o.pos);
      }
    }
    {
      const size_t pos0 = rng() % (len - k + 1);
      nthash::NtHash h(s.data(), s.size(), (uint8_t)m, (uint16_t)k, pos0);
      nthash::NtHash copy(h); // copies before and after use
      nto_nthash o;
      nto_nthash_init(&o, s.data(), s.size(), m, k, pos0, hb.data());
      bool seen = false;
      for (int step = 0; step < 120; ++step) {
        const int op = rng() % 8;
        const char c = "ACGTN"[rng() % 5];
        bool a;
        int b;
        switch (op) {
          case 0: case 1: case 2: case 3: a = h.roll(); b = nto_nthash_roll(&o); break;
          case 4: a = h.roll_back(); b = nto_nthash_roll_back(&o); break;
          case 5: a = h.peek(); b = nto_nthash_peek(&o); break;
          case 6: a = h.peek(c); b = nto_nthash_peek_char(&o, c); break;
          default: a = h.peek_back(c); b = nto_nthash_peek_back_char(&o, c); break;
        }
        if (o.pos > len - k) break; // a failed skip left pos past the last window: the reference reads out of bounds from here
        CHECK(a == (b != 0), "op %d return", op);
        CHECK(h.get_pos() == o.pos, "op %d pos", op);
        seen = seen || b;
        if (seen && b) CHECK(std::memcmp(h.hashes(), hb.data(), 8 * m) == 0, "op %d hashes", op);
      }
      nthash::NtHash moved(std::move(h));
      (void)moved.get_pos();
      (void)copy.roll();
    }
  }
}

// BlindNtHash: caller-fed characters, against direct hashes of the window it must hold
static void test_blind()
{
  for (int it = 0; it < 200; ++it) {
    const unsigned k = 3 + rng() % 60;
    const unsigned m = 1 + rng() % 3;
    std::string s = rand_seq(k + 200, 0);
    nthash::BlindNtHash h(s.data(), (uint8_t)m, (uint16_t)k, 0);
    std::string win = s.substr(0, k);
    std::vector<uint64_t> want(m);
    for (int step = 0; step < 150; ++step) {
      const char c = "ACGT"[rng() % 4];
      if (rng() % 3) { h.roll(c); win = win.substr(1) + c; }
      else { h.roll_back(c); win = std::string(1, c) + win.substr(0, k - 1); }
      const uint64_t f = nto_base_fwd(win.data(), k), r = nto_base_rev(win.data(), k);
      nto_extend(f, r, k, m, want.data());
      CHECK(h.get_forward_hash() == f && h.get_reverse_hash() == r, "blind strands k %u", k);
      CHECK(std::memcmp(h.hashes(), want.data(), 8 * m) == 0, "blind hashes k %u", k);
    }
    nthash::BlindNtHash copy(h);
    copy.peek('A');
    copy.peek_back('T');
  }
}

static std::string rand_seed(unsigned k)
{
  std::string half((k + 1) / 2, '1');
  for (auto& c : half) c = (rng() % 4) ? '1' : '0';
  std::string s = half;
  for (unsigned i = k / 2; i-- > 0;) s.push_back(half[i]);
  s.resize(k, '1');
  return s; // palindromic: no warning noise
}

// SeedNtHash (short sequences: host path) + BlindSeedNtHash, against the oracle's batch walk / window formula
static void test_seeds()
{
  for (int it = 0; it < 200; ++it) {
    const unsigned k = 4 + rng() % 40;
    const size_t len = k + rng() % 200;
    const unsigned m2 = 1 + rng() % 3, ns = 1 + rng() % 3;
    std::vector<std::string> seeds;
    for (unsigned i = 0; i < ns; ++i) seeds.push_back(rand_seed(k));
    std::vector<const char*> sp;
    for (auto& x : seeds) sp.push_back(x.c_str());
    const std::string s = rand_seq(len, it & 1);
    const uint64_t offs[2] = {0, s.size()};
    std::vector<uint64_t> want((len - k + 1) * ns * m2);
    std::vector<uint32_t> wpos(len - k + 1);
    const uint64_t total = nto_seed_batch(s.data(), offs, 1, sp.data(), ns, k, m2, want.data(), wpos.data(), nullptr);
    nthash::SeedNtHash h(s, seeds, (uint8_t)m2, (uint16_t)k);
    uint64_t n = 0;
    while (h.roll()) {
      CHECK(n < total, "seed walk emits too many");
      if (n >= total) break;
      CHECK(h.get_pos() == wpos[n], "seed pos %zu vs %u", h.get_pos(), wpos[n]);
      CHECK(std::memcmp(h.hashes(), want.data() + n * ns * m2, 8 * ns * m2) == 0, "seed hashes at %u", wpos[n]);
      ++n;
    }
    CHECK(n == total, "seed walk count %llu vs %llu", (unsigned long long)n, (unsigned long long)total);
    nthash::SeedNtHash c2(h);
    (void)c2.roll_back();
    (void)c2.peek();
    (void)c2.peek_back();
    const auto parsed = nthash::parse_seeds(seeds);
    nthash::SeedNtHash from_parsed(s.data(), s.size(), parsed, (uint8_t)m2, (uint16_t)k, 0);
    if (total) {
      CHECK(from_parsed.roll(), "parsed-seed ctor roll");
      CHECK(std::memcmp(from_parsed.hashes(), want.data(), 8 * ns * m2) == 0, "parsed-seed ctor hashes");
    }
    // BlindSeedNtHash: window formula after every roll
    std::string clean = rand_seq(k + 60, 0);
    nthash::BlindSeedNtHash b(clean.data(), seeds, (uint8_t)m2, (uint16_t)k, 0);
    std::string win = clean.substr(0, k);
    for (int step = 0; step < 40; ++step) {
      const char c = "ACGT"[rng() % 4];
      b.roll(c);
      win = win.substr(1) + c;
      for (unsigned si = 0; si < ns; ++si) {
        uint64_t f, r;
        nto_seed_window(win.data(), seeds[si].c_str(), k, &f, &r);
        CHECK(b.get_forward_hash()[si] == f && b.get_reverse_hash()[si] == r, "blind seed strands");
      }
    }
    b.roll_back('A');
  }
}

int main()
{
  test_nthash();
  test_blind();
  test_seeds();
  if (fails) return 1;
  std::puts("sanitize driver OK");
  return 0;
}
