// Host-side check of nthash_amd/csrc/seed_px_plan.hpp: a spaced seed's strand hashes as sparse sums over scanned term
// arrays, evaluated exactly the way seed_px_kernel.hpp does it (terms in one frame, exclusive scans of stride 1 / d, the
// reads of the plan, one pair of split rotates per window) against the masked direct formula on nt_math.hpp (which
// tests/test_oracle.py pins to the reference).  Every array form: the terms, the prefix, both mixed, strides of the
// terms and of the prefix, and the planner's own choice.  Prints "ok <n checks>" or the first mismatch.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "../../nthash_amd/csrc/nt_math.hpp"
#include "../../nthash_amd/csrc/seed_px_plan.hpp"

using namespace ntamd;

struct Ent {
  uint64_t t = 0, u = 0;
};

static std::vector<Ent> scan(const std::vector<Ent>& x, uint32_t d) // Y(q) = XOR_{j >= 1} X(q - j d)
{
  std::vector<Ent> y(x.size());
  for (size_t q = d; q < x.size(); ++q) {
    y[q].t = y[q - d].t ^ x[q - d].t;
    y[q].u = y[q - d].u ^ x[q - d].u;
  }
  return y;
}

int main()
{
  std::mt19937_64 rng(2026);
  const char letters[4] = {'A', 'C', 'T', 'G'}; // code = (ascii >> 1) & 3
  unsigned long checks = 0;
  auto rnd_seed = [&](uint32_t k, int kind) {
    std::string s(k, '1');
    if (kind == 0) {
      for (uint32_t i = 1; i + 1 < k; ++i) s[i] = (rng() % 10) < 7 ? '1' : '0';
    } else if (kind == 1) { // a few blocks
      const uint32_t gaps = 1 + (uint32_t)(rng() % 4);
      for (uint32_t g = 0; g < gaps; ++g) {
        const uint32_t a = 1 + (uint32_t)(rng() % (k - 2)), n = 1 + (uint32_t)(rng() % (k / 4 + 1));
        for (uint32_t i = a; i < a + n && i + 1 < k; ++i) s[i] = '0';
      }
    } else if (kind == 2) { // periodic
      const uint32_t p = 2 + (uint32_t)(rng() % 12), on = 1 + (uint32_t)(rng() % (p - 1));
      for (uint32_t i = 0; i < k; ++i) s[i] = (i % p) < on ? '1' : '0';
      if (rng() & 1) s[k / 2] = s[k / 2] == '1' ? '0' : '1';
    } else { // don't-cares at both ends
      for (uint32_t i = 0; i < k; ++i) s[i] = (i < 2 || i + 3 > k || i == k / 2) ? '0' : '1';
      if (k < 8) s[k / 3] = '1';
    }
    return s;
  };
  const uint32_t ks[] = {3, 4, 7, 16, 24, 31, 32, 33, 48, 64, 65, 100, 128, 160, 200};
  for (int trial = 0; trial < 400; ++trial) {
    const uint32_t k = ks[rng() % (sizeof ks / sizeof ks[0])];
    const uint32_t ns = 1 + (uint32_t)(rng() % 4);
    std::vector<std::string> seeds;
    std::vector<std::vector<uint8_t>> care;
    for (uint32_t s = 0; s < ns; ++s) {
      seeds.push_back(rnd_seed(k, (int)(rng() % 4)));
      std::vector<uint8_t> c(k);
      for (uint32_t p = 0; p < k; ++p) c[p] = seeds[s][p] == '1';
      care.push_back(c);
    }
    const uint32_t shift = (uint32_t)(rng() % 16), L = k + (uint32_t)(rng() % 120), n_reads = 1 + (uint32_t)(rng() % 3);
    const uint32_t NQ = shift + n_reads * L;
    std::string slab(NQ, 'A');
    for (auto& ch : slab) ch = letters[rng() & 3];
    int force = -1;
    switch (trial % 7) {
      case 0: force = 0; break;
      case 1: force = 1; break;
      case 2: force = 2; break;
      case 3: force = 3 + 2 + (int)(rng() % 20); break;   // prefix, then stride d
      case 4: force = 100 + 2 + (int)(rng() % 20); break; // the terms at stride d
      default: force = -1;
    }
    if (force >= 100 && (uint32_t)(force - 100) >= k) force = 0;
    if (force >= 3 && force < 100 && (uint32_t)(force - 3) >= k) force = 1;
    const PxPlan plan = px_make_plan(care, k, (double)L / (L - k + 1), force);
    if (!plan.ok) {
      if (force < 0) continue; // (more reads than the kernel's list holds)
      printf("no plan: k=%u force=%d\n", k, force);
      return 1;
    }
    const uint32_t n_e = NQ + plan.reach + 1;
    std::vector<Ent> raw(n_e);
    for (uint32_t q = 0; q < n_e; ++q) {
      const unsigned code = q < NQ ? code_of((unsigned char)slab[q]) : 0u; // (behind the slab: anything, it cancels)
      raw[q].t = srol_n(seed_of_code(code), 1023u - q % 1023u);
      raw[q].u = srol_n(seed_of_code(code ^ 2u), q);
    }
    std::vector<std::vector<Ent>> arrays;
    for (const PxArray& y : plan.arrays) {
      std::vector<Ent> a = raw;
      if (y.d1) a = scan(a, y.d1);
      if (y.d2) a = scan(a, y.d2);
      arrays.push_back(a);
    }
    for (uint32_t r = 0; r < n_reads; ++r)
      for (uint32_t p = 0; p + k <= L; ++p) {
        const uint32_t q = shift + r * L + p;
        for (uint32_t s = 0; s < ns; ++s) {
          uint64_t gt = 0, gu = 0;
          for (uint32_t i = plan.seed_first[s]; i < plan.seed_first[s + 1]; ++i) {
            const Ent& e = arrays[plan.terms[i].arr][q + plan.terms[i].e];
            gt ^= e.t;
            gu ^= e.u;
          }
          const uint64_t f = srol_n(gt, q + k - 1), rv = srol_n(gu, 1023u - q % 1023u);
          uint64_t wf = 0, wr = 0;
          for (uint32_t i = 0; i < k; ++i)
            if (care[s][i]) {
              const unsigned code = code_of((unsigned char)slab[q + i]);
              wf ^= srol_n(seed_of_code(code), k - 1 - i);
              wr ^= srol_n(seed_of_code(code ^ 2u), i);
            }
          if (f != wf || rv != wr) {
            printf("mismatch: k=%u seed=%s force=%d q=%u (arrays %zu, terms %u)\n", k, seeds[s].c_str(), force, q,
                   plan.arrays.size(), plan.n_terms());
            return 1;
          }
          ++checks;
        }
      }
  }
  // the collapse the plan is there for
  {
    std::vector<uint8_t> alt(31), blocks(128, 1);
    for (uint32_t i = 0; i < 31; ++i) alt[i] = (i & 1) == 0;
    PxPlan p = px_make_plan({alt}, 31, 250.0 / 220.0);
    if (!p.ok || p.n_terms() != 2 || p.arrays.size() != 1 || !(p.arrays[0] == PxArray{2, 0})) {
      printf("1010...1 should be two reads of the stride-2 scan, got %u reads\n", p.n_terms());
      return 1;
    }
    for (uint32_t i = 40; i < 88; ++i) blocks[i] = 0;
    p = px_make_plan({blocks}, 128, 250.0 / 123.0);
    if (!p.ok || p.n_terms() != 4 || !(p.arrays[0] == PxArray{1, 0})) {
      printf("two blocks should be four reads of the prefix, got %u\n", p.n_terms());
      return 1;
    }
  }
  printf("ok %lu\n", checks);
  return 0;
}
