// seed_px_plan.hpp -- host side of seed_px_kernel.hpp: a spaced seed as a SPARSE sum over scanned arrays.
//
// Every masked strand hash is linear over XOR.  With the per-position terms of a tile taken in ONE common frame,
//     T(q) = srol^{-q}(S[c_q])          U(q) = srol^{q}(S[comp c_q])             (q: position in the tile's slab)
// the forward hash of the window at q under care set C is srol^{q+k-1}( XOR_{i in C} T(q+i) ) and the reverse one
// srol^{-q}( XOR_{i in C} U(q+i) )  (SURVEY.md App. A.4 with the rotations factored out: srol is a bit permutation).
// What is left per window is a sliding masked XOR -- the product of the mask polynomial C(x) with the term sequence over
// GF(2) -- and it can be taken from any array Y with T = B * Y for a polynomial B:
//     XOR_{i in C} T(q+i) = XOR_{e in supp(C B)} Y(q+e).
// B = 1: the terms themselves, |C| reads per window (the masked direct formula, src/seed.cpp:130-175).
// B = 1 + x: the exclusive prefix XOR; supp(C (1 + x)) = the edges of the care runs, 2 reads per run -- what the
//   reference's roll pays (one base in, one out per block, src/seed.cpp:177-207), without its chain from window to window.
// B = 1 + x^d, (1 + x)(1 + x^d): stride-d scans (of the terms / of the prefix); a seed that repeats under a shift by d
//   collapses: 1010...1 with d = 2 is 2 reads instead of 16, evenly spaced blocks of period d are 4.
// The plan picks at most PX_MAX_ARRAYS arrays for a seed set and, per seed, the array with the fewest reads; reads of one
// seed may also mix the raw terms with the prefix (an edge pair (e, e+1) of the prefix is the single term e).
// Host only (no HIP): nthip_seeds_create builds it, tests/host/seed_px_host.cpp checks it against the direct formula.
#pragma once

#include <algorithm>
#include <cstdint>
#include <vector>

namespace ntamd {

constexpr uint32_t PX_MAX_ARRAYS = 3;
constexpr uint32_t PX_MAX_TERMS = 400;  // all seeds of a set (the kernel takes the list by value: 4 KiB of arguments)
constexpr uint32_t PX_MAX_SEEDS = 32;
constexpr uint32_t PX_MAX_STRIDE = 64;

struct PxArray {
  uint32_t d1 = 0, d2 = 0; // exclusive scans applied to the raw terms, in this order (0: none); {0,0} raw, {1,0} prefix
  bool operator==(const PxArray& o) const { return d1 == o.d1 && d2 == o.d2; }
  uint32_t reach() const { return d1 + d2; } // a window's reads end this far behind its last base
};

struct PxTerm {
  uint16_t e;  // the read is Y(q + e)
  uint8_t arr; // index into PxPlan::arrays
};

struct PxPlan {
  bool ok = false;
  uint32_t k = 0;
  std::vector<PxArray> arrays;
  std::vector<uint32_t> seed_first; // terms of seed s: [seed_first[s], seed_first[s + 1])
  std::vector<PxTerm> terms;
  uint32_t reach = 0;               // max over the arrays in use
  uint32_t n_terms() const { return (uint32_t)terms.size(); }
};

namespace px {

using Poly = std::vector<uint8_t>; // coefficient of x^i at [i]

inline Poly mul_1pxd(const Poly& a, uint32_t d)
{
  if (d == 0) return a;
  Poly r(a.size() + d, 0);
  for (size_t i = 0; i < a.size(); ++i) {
    r[i] ^= a[i];
    r[i + d] ^= a[i];
  }
  return r;
}
inline std::vector<uint16_t> support(const Poly& a)
{
  std::vector<uint16_t> s;
  for (size_t i = 0; i < a.size(); ++i)
    if (a[i]) s.push_back((uint16_t)i);
  return s;
}
inline std::vector<uint16_t> reads_of(const Poly& care, const PxArray& y) { return support(mul_1pxd(mul_1pxd(care, y.d1), y.d2)); }

// the prefix edges with every adjacent pair (e, e + 1) replaced by the raw term e
inline void mixed_reads(const Poly& care, std::vector<uint16_t>& raw, std::vector<uint16_t>& pre)
{
  const std::vector<uint16_t> edges = support(mul_1pxd(care, 1));
  raw.clear();
  pre.clear();
  for (size_t i = 0; i < edges.size();) {
    if (i + 1 < edges.size() && edges[i + 1] == edges[i] + 1) {
      raw.push_back(edges[i]);
      i += 2;
    } else {
      pre.push_back(edges[i]);
      ++i;
    }
  }
}

} // namespace px

// care[s][p] != 0: position p of seed s is a care position.  pos_per_win = read length / windows per read: what one more
// array costs (every position of the slab is written once per array) against what a read costs.
// force_array >= 0 (tests): 0 raw only, 1 prefix only, 2 raw + prefix mixed, 3 + d: the stride-d scan of the prefix only,
// 100 + d: the stride-d scan of the raw terms only.
inline PxPlan px_make_plan(const std::vector<std::vector<uint8_t>>& care, uint32_t k, double pos_per_win, int force_array = -1,
                           bool allow_strides = true)
{
  PxPlan best;
  const uint32_t ns = (uint32_t)care.size();
  if (ns == 0 || ns > PX_MAX_SEEDS || k == 0 || k > 4096) return best;
  // what a slab position costs per array, in reads: its 16-byte write against a read's 16 bytes at a fourth of the
  // rate, the scan's instructions on top (a stride-d scan is a pass of its own over the array: twice that)
  const double W_RAW = 4.0, W_PRE = 5.5, W_STRIDE = 9.0;
  auto array_cost = [&](const PxArray& y) {
    double c = (y.d1 == 0 && y.d2 == 0) ? W_RAW : W_PRE;
    if (y.d1 > 1) c += W_STRIDE - W_PRE + W_RAW;
    if (y.d2 > 1) c += W_STRIDE;
    return c * pos_per_win;
  };
  struct Cand {
    std::vector<PxArray> arrays;
    std::vector<std::vector<PxTerm>> per_seed;
    double cost = 0;
  };
  auto finish = [&](Cand& c) {
    c.cost = 0;
    for (const PxArray& y : c.arrays) c.cost += array_cost(y);
    for (const auto& t : c.per_seed) c.cost += (double)t.size() + 6.0; // (the seed's rotations and its value's way out)
  };
  auto single_array = [&](const PxArray& y) {
    Cand c;
    c.arrays = {y};
    for (uint32_t s = 0; s < ns; ++s) {
      std::vector<PxTerm> t;
      for (uint16_t e : px::reads_of(care[s], y)) t.push_back({e, 0});
      c.per_seed.push_back(t);
    }
    finish(c);
    return c;
  };
  auto mixed = [&]() {
    Cand c;
    c.arrays = {PxArray{0, 0}, PxArray{1, 0}};
    for (uint32_t s = 0; s < ns; ++s) {
      std::vector<uint16_t> raw, pre;
      px::mixed_reads(care[s], raw, pre);
      std::vector<PxTerm> t;
      for (uint16_t e : raw) t.push_back({e, 0});
      for (uint16_t e : pre) t.push_back({e, 1});
      c.per_seed.push_back(t);
    }
    finish(c);
    return c;
  };
  // every seed on the cheapest of a given set of arrays
  auto best_of = [&](const std::vector<PxArray>& ys) {
    Cand c;
    std::vector<bool> used(ys.size(), false);
    std::vector<std::vector<std::pair<uint16_t, uint32_t>>> pick(ns);
    for (uint32_t s = 0; s < ns; ++s) {
      size_t bi = 0, bn = ~(size_t)0;
      std::vector<uint16_t> br;
      for (size_t i = 0; i < ys.size(); ++i) {
        std::vector<uint16_t> r = px::reads_of(care[s], ys[i]);
        if (r.size() < bn) {
          bn = r.size();
          bi = i;
          br.swap(r);
        }
      }
      used[bi] = true;
      for (uint16_t e : br) pick[s].push_back({e, (uint32_t)bi});
    }
    std::vector<uint32_t> remap(ys.size(), 0);
    for (size_t i = 0; i < ys.size(); ++i)
      if (used[i]) {
        remap[i] = (uint32_t)c.arrays.size();
        c.arrays.push_back(ys[i]);
      }
    for (uint32_t s = 0; s < ns; ++s) {
      std::vector<PxTerm> t;
      for (auto& p : pick[s]) t.push_back({p.first, (uint8_t)remap[p.second]});
      c.per_seed.push_back(t);
    }
    finish(c);
    return c;
  };

  std::vector<Cand> cands;
  if (force_array == 0) cands.push_back(single_array({0, 0}));
  else if (force_array == 1) cands.push_back(single_array({1, 0}));
  else if (force_array == 2) cands.push_back(mixed());
  else if (force_array >= 100) cands.push_back(single_array({(uint32_t)force_array - 100u, 0}));
  else if (force_array >= 3) cands.push_back(single_array({1, (uint32_t)force_array - 3u}));
  else {
    cands.push_back(single_array({0, 0}));
    cands.push_back(single_array({1, 0}));
    cands.push_back(mixed());
    if (allow_strides) {
      // per seed the stride that leaves the fewest reads (of the terms, of the prefix); then the sets of up to
      // PX_MAX_ARRAYS arrays made of the plain two and the seeds' favourites
      std::vector<PxArray> fav;
      auto add = [&](const PxArray& y) {
        if (std::find(fav.begin(), fav.end(), y) == fav.end()) fav.push_back(y);
      };
      for (uint32_t s = 0; s < ns; ++s) {
        size_t bn = std::min(px::reads_of(care[s], {0, 0}).size(), px::reads_of(care[s], {1, 0}).size());
        PxArray by{0, 0};
        bool found = false;
        for (uint32_t d = 2; d <= PX_MAX_STRIDE && d < k; ++d)
          for (int with_pre = 0; with_pre < 2; ++with_pre) {
            const PxArray y = with_pre ? PxArray{1, d} : PxArray{d, 0};
            const size_t n = px::reads_of(care[s], y).size();
            if (n + 2 <= bn) { // (a scan of its own has to save more than a read or two)
              bn = n;
              by = y;
              found = true;
            }
          }
        if (found) add(by);
      }
      if (fav.size() > 6) fav.resize(6);
      const std::vector<PxArray> plain = {PxArray{0, 0}, PxArray{1, 0}};
      for (size_t i = 0; i < fav.size(); ++i) {
        cands.push_back(best_of({fav[i]}));
        for (const PxArray& p : plain) cands.push_back(best_of({fav[i], p}));
        for (size_t j = i + 1; j < fav.size(); ++j) {
          cands.push_back(best_of({fav[i], fav[j]}));
          for (const PxArray& p : plain) cands.push_back(best_of({fav[i], fav[j], p}));
        }
      }
    }
  }
  const Cand* pick = nullptr;
  for (const Cand& c : cands) {
    size_t n = 0;
    for (const auto& t : c.per_seed) n += t.size();
    if (n > PX_MAX_TERMS || c.arrays.size() > PX_MAX_ARRAYS) continue;
    if (!pick || c.cost < pick->cost) pick = &c;
  }
  if (!pick) return best;
  best.ok = true;
  best.k = k;
  best.arrays = pick->arrays;
  best.seed_first.push_back(0);
  for (const auto& t : pick->per_seed) {
    best.terms.insert(best.terms.end(), t.begin(), t.end());
    best.seed_first.push_back((uint32_t)best.terms.size());
  }
  for (const PxArray& y : best.arrays) best.reach = std::max(best.reach, y.reach());
  return best;
}

} // namespace ntamd
