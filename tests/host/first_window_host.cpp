// Host-side check of nthash_amd/csrc/first_window.hpp: the grouped and the scan form of a run's first window against
// the direct hashes of nt_math.hpp (which tests/test_oracle.py pins to the reference), on the CPU.  The same source
// compiles into the kernels; here the wave is a loop.  Prints "ok <n checks>" or the first mismatch.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "../../nthash_amd/csrc/first_window.hpp"

using namespace ntamd;

int main()
{
  std::vector<fw_u4> tab(FW_ENTRIES);
  build_fw_tables(tab.data());
  std::mt19937_64 rng(12345);
  const char letters[4] = {'A', 'C', 'T', 'G'}; // code = (ascii >> 1) & 3
  unsigned long checks = 0;
  const uint32_t ks[] = {3, 4, 5, 15, 16, 17, 31, 32, 33, 47, 48, 63, 64, 65, 80, 96, 100, 127, 128, 129, 200, 255, 256, 500,
                         1022, 1023, 1024, 1025, 1100, 2047, 4099};
  for (int trial = 0; trial < 60; ++trial) {
    const uint32_t n = 64 + (uint32_t)(rng() % 9000);
    const uint32_t shift = (uint32_t)(rng() % 16); // the slab starts anywhere inside its first vector
    const uint32_t total = shift + n;
    const uint32_t n_words = (total + 15) / 16;
    std::vector<uint32_t> bits(n_words + 8, 0);
    std::string seq(total, 'A');
    for (uint32_t i = 0; i < total; ++i) {
      const uint32_t c = (uint32_t)(rng() & 3);
      seq[i] = letters[c];
      bits[i >> 4] |= c << (2 * (i & 15));
    }
    // prefixes at every word boundary: uw[w] = {U(16 w), V(16 w)}, as the kernel builds them (lane chunks + XOR scan)
    std::vector<fw_u4> uw(n_words + 2);
    const uint32_t W = (n_words + 1 + 63) / 64;
    std::vector<fw_u4> lane_tot(64, fw_make(0, 0, 0, 0));
    for (uint32_t lane = 0; lane < 64; ++lane) {
      fw_u4 run = fw_make(0, 0, 0, 0);
      for (uint32_t w = 0; w < W; ++w) {
        const uint32_t wi = lane * W + w;
        if (wi > n_words) break;
        uw[wi] = run;
        const fw_u4 e = fw_scan_word(tab.data(), bits[wi], wi);
        run.x ^= e.x; run.y ^= e.y; run.z ^= e.z; run.w ^= e.w;
      }
      lane_tot[lane] = run;
    }
    fw_u4 excl = fw_make(0, 0, 0, 0);
    for (uint32_t lane = 0; lane < 64; ++lane) {
      for (uint32_t w = 0; w < W; ++w) {
        const uint32_t wi = lane * W + w;
        if (wi > n_words) break;
        uw[wi].x ^= excl.x; uw[wi].y ^= excl.y; uw[wi].z ^= excl.z; uw[wi].w ^= excl.w;
      }
      excl.x ^= lane_tot[lane].x; excl.y ^= lane_tot[lane].y; excl.z ^= lane_tot[lane].z; excl.w ^= lane_tot[lane].w;
    }
    for (uint32_t k : ks) {
      if (k > n) continue;
      for (int rep = 0; rep < 12; ++rep) {
        const uint32_t p = shift + (uint32_t)(rng() % (n - k + 1));
        const uint64_t f0 = direct_fwd(seq.data() + p, k), r0 = direct_rev(seq.data() + p, k);
        uint32_t fl, fh, rl, rh;
        grouped_first_window(bits.data(), tab.data(), p, k, k % 31u, k % 33u, fl, fh, rl, rh);
        if ((((uint64_t)fh << 32) | fl) != f0 || (((uint64_t)rh << 32) | rl) != r0) {
          printf("grouped mismatch: n=%u k=%u p=%u\n", n, k, p);
          return 1;
        }
        scan_first_window(bits.data(), tab.data(), uw.data(), p, k, k % 1023u, k % 31u, k % 33u, fl, fh, rl, rh);
        if ((((uint64_t)fh << 32) | fl) != f0 || (((uint64_t)rh << 32) | rl) != r0) {
          printf("scan mismatch: n=%u k=%u p=%u\n", n, k, p);
          return 1;
        }
        checks += 2;
      }
      // the last window of the slab: its end is the slab's end
      {
        const uint32_t p = shift + n - k;
        const uint64_t f0 = direct_fwd(seq.data() + p, k), r0 = direct_rev(seq.data() + p, k);
        uint32_t fl, fh, rl, rh;
        scan_first_window(bits.data(), tab.data(), uw.data(), p, k, k % 1023u, k % 31u, k % 33u, fl, fh, rl, rh);
        if ((((uint64_t)fh << 32) | fl) != f0 || (((uint64_t)rh << 32) | rl) != r0) {
          printf("scan mismatch at the slab's end: n=%u k=%u p=%u\n", n, k, p);
          return 1;
        }
        ++checks;
      }
    }
  }
  printf("ok %lu\n", checks);
  return 0;
}
