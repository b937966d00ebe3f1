/* TEST INFRASTRUCTURE ONLY (tests/test_sanitizers.py): a stand-in for libnthash_hip.so on a box without a GPU, so that
 * the HOST logic of the C++ facade -- position state machines, the tabulated recurrences behind roll_back()/peek()/
 * Blind* and behind roll() on short sequences, seed parsing, copies and moves -- can run under ASan/UBSan in the CPU
 * test tier.  It hashes nothing: a context "exists", every hashing entry point fails with NTHIP_ERR_NODEVICE. */
#include <stdlib.h>
#include "nthash_hip.h"
#include <stddef.h>

static int g_ctx_storage;
const char* nthip_version(void) { return "stub (no device): sanitizer job only"; }
const char* nthip_last_error(void) { return "stub C-ABI: no device (sanitizer job)"; }
int nthip_ctx_create(int device, nthip_ctx** out) { (void)device; *out = (nthip_ctx*)&g_ctx_storage; return NTHIP_OK; }
int nthip_ctx_destroy(nthip_ctx* c) { (void)c; return NTHIP_OK; }
int nthip_host_alloc(size_t bytes, void** p) { *p = malloc(bytes ? bytes : 1); return *p ? NTHIP_OK : NTHIP_ERR_HIP; }
int nthip_host_free(void* p) { free(p); return NTHIP_OK; }
int nthip_kmer_hash(nthip_ctx* c, const nthip_reads* rd, uint16_t k, uint8_t m, const nthip_out* out, uint64_t* total,
                    uint32_t flags)
{
  (void)c; (void)rd; (void)k; (void)m; (void)out; (void)flags;
  if (total) *total = 0;
  return NTHIP_ERR_NODEVICE;
}
int nthip_seeds_create(nthip_ctx* c, const char* const* seeds, uint32_t n, uint16_t k, nthip_seeds** out, int* asym)
{
  (void)c; (void)seeds; (void)n; (void)k; (void)asym;
  *out = NULL;
  return NTHIP_ERR_NODEVICE;
}
int nthip_seeds_destroy(nthip_seeds* s) { (void)s; return NTHIP_OK; }
int nthip_seed_hash(nthip_ctx* c, const nthip_reads* rd, const nthip_seeds* sd, uint8_t m2, const nthip_out* out,
                    uint64_t* total, uint32_t flags)
{
  (void)c; (void)rd; (void)sd; (void)m2; (void)out; (void)flags;
  if (total) *total = 0;
  return NTHIP_ERR_NODEVICE;
}
int nthip_multi_create(const int* devices, int n, nthip_multi** out) { (void)devices; (void)n; *out = NULL; return NTHIP_ERR_NODEVICE; }
int nthip_multi_destroy(nthip_multi* m) { (void)m; return NTHIP_OK; }
int nthip_multi_kmer_hash(nthip_multi* m, const nthip_reads* rd, uint16_t k, uint8_t mh, const nthip_out* out, uint64_t* total)
{
  (void)m; (void)rd; (void)k; (void)mh; (void)out;
  if (total) *total = 0;
  return NTHIP_ERR_NODEVICE;
}
