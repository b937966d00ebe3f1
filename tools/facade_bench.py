#!/usr/bin/env python3
"""Whole-walk rate of the C++ facade on one long sequence: `NtHash h(seq, m, k); while (h.roll()) sum += h.hashes()[0];`
with and without the helper thread that hashes the next window (NTHASH_AMD_PREFETCH=0/1).

    python tools/facade_bench.py [Mbases=256] [k=31] [m=1]
"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r"""
#include <nthash/nthash.hpp>
#include <chrono>
#include <cstdio>
#include <string>
int main(int argc, char** argv) {
  const size_t n = std::stoull(argv[1]) << 20;
  const unsigned k = std::stoi(argv[2]), m = std::stoi(argv[3]);
  std::string s(n, 'A');
  unsigned long long x = 88172645463325252ull;
  for (size_t i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; s[i] = "ACGT"[x & 3]; }
  if (argc > 4) { // spaced seeds: two seeds of k bases (every other position / two of three), m hashes each
    std::string s1(k, '1'), s2(k, '1');
    for (unsigned i = 1; i + 1 < k; i += 2) s1[i] = '0';
    for (unsigned i = 2; i + 1 < k && k - 1 - i > 1; i += 3) { s2[i] = '0'; s2[k - 1 - i] = '0'; }
    for (int rep = 0; rep < 3; ++rep) {
      const auto t0 = std::chrono::steady_clock::now();
      nthash::SeedNtHash h(s, {s1, s2}, m, k);
      unsigned long long sum = 0, cnt = 0;
      while (h.roll()) { sum += h.hashes()[0]; ++cnt; }
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      std::printf("SeedNtHash rep %d: %llu k-mers, %.3f s, %.1f M k-mers/s, sum %016llx\n", rep, cnt, dt, cnt / dt / 1e6, sum);
    }
    return 0;
  }
  for (int rep = 0; rep < 3; ++rep) {
    const auto t0 = std::chrono::steady_clock::now();
    nthash::NtHash h(s, m, k);
    unsigned long long sum = 0, cnt = 0;
    while (h.roll()) { sum += h.hashes()[0]; ++cnt; }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("rep %d: %llu k-mers, %.3f s, %.1f M k-mers/s, sum %016llx\n", rep, cnt, dt, cnt / dt / 1e6, sum);
  }
}
"""
mb, k, m = (sys.argv[1:] + ["256", "31", "1"][len(sys.argv) - 1:])[:3]
with tempfile.TemporaryDirectory() as d:
    src, exe = os.path.join(d, "fb.cpp"), os.path.join(d, "fb")
    open(src, "w").write(SRC)
    lib = os.path.join(ROOT, "nthash_amd", "lib")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), src, "-o", exe, "-L" + lib,
                           "-lnthash", "-lnthash_hip", "-Wl,-rpath," + lib, "-pthread"])
    for pf in ("0", "1"):
        env = dict(os.environ, NTHASH_AMD_PREFETCH=pf)
        print("NTHASH_AMD_PREFETCH=" + pf, flush=True)
        subprocess.run([exe, mb, k, m], env=env)
    if os.environ.get("FACADE_SEEDS") == "1":
        for pf in ("0", "1"):
            print("SeedNtHash, 2 seeds x m, NTHASH_AMD_PREFETCH=" + pf, flush=True)
            subprocess.run([exe, str(min(int(mb), 128)), k, m, "seeds"], env=dict(os.environ, NTHASH_AMD_PREFETCH=pf))
    if os.environ.get("FACADE_TRACE") == "1":
        for pf in ("0", "1"):
            print("trace, NTHASH_AMD_PREFETCH=" + pf, flush=True)
            r = subprocess.run([exe, "64", k, m], env=dict(os.environ, NTHASH_AMD_PREFETCH=pf, NTHASH_AMD_TRACE="1"),
                               capture_output=True, text=True)
            print("\n".join(r.stderr.splitlines()[:12]))
