// capi_seed_extend.hip -- nthip_seed_extend: the 4 successors / predecessors of n windows through spaced seeds
// Part of libnthash_hip.so (include/nthash_hip.h); see capi_internal.hpp for the file map.
#include "capi_internal.hpp"
#include "seed_extend_kernel.hpp"

#include <mutex>

using namespace ntamd;
using namespace ntamd::host;

namespace {

// the kernel's three mask sets of a seed set, once per nthip_seeds: per 16-base group a 2-bit-per-base mask and what the
// masked-out positions contribute when they read as code 0 (first_window.hpp: position u of a word contributes
// sror^{u+1}(S[c]) / srol^{u}(S[~c]); the same correction nthip_seeds_create makes for the any-seed form)
// (made on first use, under a lock: two host threads with their own contexts may share one seed set -- ADVICE r04; both
// pointers are published only when both tables are on the device)
std::mutex g_ext_mu;
int ext_masks(nthip_seeds* sd)
{
  std::lock_guard<std::mutex> lk(g_ext_mu);
  if (sd->d_ext_mask && sd->d_ext_acorr) return NTHIP_OK;
  const uint32_t k = sd->k, S = sd->n_seeds, G = sd->any_groups;
  std::vector<uint32_t> mask((size_t)3 * S * G, 0);
  std::vector<uint4> acorr((size_t)3 * S * G, make_uint4(0, 0, 0, 0));
  for (uint32_t t = 0; t < 3; ++t)
    for (uint32_t s = 0; s < S; ++s)
      for (uint32_t g = 0; g < G; ++g) {
        uint64_t f = 0, r = 0;
        uint32_t m = 0;
        for (uint32_t u = 0; u < 16; ++u) {
          const uint32_t p = 16 * g + u;
          bool in = false;
          if (p < k) {
            const bool blk = sd->h_blk_parity[(size_t)s * k + p] != 0, mono = sd->h_is_mono[(size_t)s * k + p] != 0;
            in = t == 0 ? (blk != mono) : t == 1 ? blk : mono;
          }
          if (in) {
            m |= 3u << (2 * u);
          } else {
            f ^= srol_n(seed_of_code(0), 1023u - (u + 1u));
            r ^= srol_n(seed_of_code(2), u);
          }
        }
        mask[((size_t)t * S + s) * G + g] = m;
        acorr[((size_t)t * S + s) * G + g] = make_uint4((uint32_t)f, (uint32_t)(f >> 32), (uint32_t)r, (uint32_t)(r >> 32));
      }
  uint32_t* d_mask = nullptr;
  uint4* d_acorr = nullptr;
  HIPCHK(hipMalloc((void**)&d_mask, mask.size() * 4));
  if (hipMemcpy(d_mask, mask.data(), mask.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
      hipMalloc((void**)&d_acorr, acorr.size() * sizeof(uint4)) != hipSuccess ||
      hipMemcpy(d_acorr, acorr.data(), acorr.size() * sizeof(uint4), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(d_mask);
    if (d_acorr) (void)hipFree(d_acorr);
    return fail(NTHIP_ERR_HIP, "the mask tables of nthip_seed_extend could not be put on the device");
  }
  sd->d_ext_acorr = d_acorr;
  sd->d_ext_mask = d_mask;
  return NTHIP_OK;
}

} // namespace

extern "C" int nthip_seed_extend(nthip_ctx* c, const char* kmers, uint64_t n, const nthip_seeds* seeds, uint8_t m2, uint64_t* self,
                                 uint64_t* next, uint64_t* prev, uint32_t flags)
{
  if (!c || !seeds) return fail(NTHIP_ERR_ARG, "ctx / seeds is NULL");
  if (m2 == 0) return fail(NTHIP_ERR_UNSUPPORTED, "num_hashes_per_seed must be >= 1");
  if (n && !kmers) return fail(NTHIP_ERR_ARG, "kmers is NULL");
  if (!self && !next && !prev) return fail(NTHIP_ERR_ARG, "no output requested");
  if (seeds->device != c->device) return fail(NTHIP_ERR_ARG, "the seed set lives on another device");
  const uint32_t k = seeds->k, S = seeds->n_seeds, G = seeds->any_groups;
  if (k > 128) return fail(NTHIP_ERR_UNSUPPORTED, "nthip_seed_extend takes seeds of at most 128 bases");
  HIPCHK(hipSetDevice(c->device));
  if (n == 0) return NTHIP_OK;
  nthip_seeds* sd = const_cast<nthip_seeds*>(seeds); // (the masks are a cache)
  NTCHK(ext_masks(sd));
  Staged keep;
  const uint8_t* d_in = (const uint8_t*)kmers;
  const size_t per = (size_t)S * m2;
  uint64_t *d_self = self, *d_next = next, *d_prev = prev;
  if (flags & NTHIP_HOST_INPUT) {
    void* p = nullptr;
    NTCHK(own_alloc(keep, n * k, &p));
    HIPCHK(hipMemcpyAsync(p, kmers, n * k, hipMemcpyHostToDevice, c->stream));
    d_in = (const uint8_t*)p;
  }
  if (flags & NTHIP_HOST_OUTPUT) {
    if (self) NTCHK(own_alloc(keep, n * per * 8, (void**)&d_self));
    if (next) NTCHK(own_alloc(keep, n * 4 * per * 8, (void**)&d_next));
    if (prev) NTCHK(own_alloc(keep, n * 4 * per * 8, (void**)&d_prev));
  }
  SeedExtendArgs a;
  memset(&a, 0, sizeof a);
  a.kmers = d_in;
  a.n = n;
  NTCHK(get_fw_tab(c, &a.fw));
  a.mask = sd->d_ext_mask;
  a.acorr = sd->d_ext_acorr;
  a.self = d_self;
  a.next = d_next;
  a.prev = d_prev;
  a.k = k;
  a.n_seeds = S;
  a.m2 = m2;
  a.G = G;
  a.bits_dwords = ((64u * k + 15u) >> 4) + G + 3u;
  const uint32_t n_grp = 3u * S * G;
  const size_t lds = (size_t)SA_TAB_ENTRIES * 16 + (size_t)n_grp * 16 + (size_t)((n_grp + 3u) & ~3u) * 4 + (size_t)4 * a.bits_dwords * 4;
  if (lds > lds_cap_of(c)) return fail(NTHIP_ERR_UNSUPPORTED, "nthip_seed_extend: %u seeds of %u bases need %zu bytes of LDS", S, k, lds);
  uint64_t blocks = (n + 255) / 256;
  if (blocks > (uint64_t)c->n_cu * 2) blocks = (uint64_t)c->n_cu * 2;
  auto go = [&](auto kernel) -> int {
    NTCHK(raise_max_dynamic_lds(c->device, reinterpret_cast<const void*>(kernel), lds));
    prof_begin(c, "seed_extend_kernel");
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(256), lds, c->stream, a);
    prof_end(c);
    HIPCHK(hipGetLastError());
    return NTHIP_OK;
  };
  if (G <= 2) NTCHK(go(seed_extend_kernel<2>));
  else if (G <= 4) NTCHK(go(seed_extend_kernel<4>));
  else NTCHK(go(seed_extend_kernel<8>));
  if (flags & NTHIP_HOST_OUTPUT) {
    if (self) HIPCHK(hipMemcpyAsync(self, d_self, n * per * 8, hipMemcpyDeviceToHost, c->stream));
    if (next) HIPCHK(hipMemcpyAsync(next, d_next, n * 4 * per * 8, hipMemcpyDeviceToHost, c->stream));
    if (prev) HIPCHK(hipMemcpyAsync(prev, d_prev, n * 4 * per * 8, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  return NTHIP_OK;
}
