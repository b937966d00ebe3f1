// Does the way device memory is obtained decide the "page class" (profiles/r02_notes.md sections 11, 18, 25)?
// Write-only fill rate of a buffer from: hipMalloc; the virtual-memory API with one physical handle; with handles of
// 2 MiB ... 1 GiB mapped back to back; virtual addresses aligned to 2 MiB / 1 GiB.
//   hipcc --offload-arch=gfx950 -O3 tools/bench_micro/vmm_alloc.hip -o tools/bench_micro/vmm_alloc_bin && tools/bench_micro/vmm_alloc_bin [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void fill(uint4* p, size_t n, uint32_t v)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    const v4 x = {v, v + 1, (uint32_t)i, v};
    __builtin_nontemporal_store(x, (v4*)&p[i]);
  }
}

static double fill_rate(void* p, size_t bytes)
{
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  fill<<<256 * 8, 256>>>((uint4*)p, bytes / 16, 1);
  (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(a);
    fill<<<256 * 8, 256>>>((uint4*)p, bytes / 16, r);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  return bytes / best / 1e6;
}

int main(int argc, char** argv)
{
  const size_t gib = argc > 1 ? atoi(argv[1]) : 32;
  const size_t bytes = gib << 30;
  CK(hipSetDevice(0));
  for (int rep = 0; rep < 3; ++rep) {
    void* p = nullptr;
    CK(hipMalloc(&p, bytes));
    printf("hipMalloc %zu GiB #%d               va %p: %7.0f GB/s\n", gib, rep, p, fill_rate(p, bytes));
    CK(hipFree(p));
  }
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran_min = 0, gran_rec = 0;
  CK(hipMemGetAllocationGranularity(&gran_min, &prop, hipMemAllocationGranularityMinimum));
  CK(hipMemGetAllocationGranularity(&gran_rec, &prop, hipMemAllocationGranularityRecommended));
  printf("granularity: minimum %zu, recommended %zu\n", gran_min, gran_rec);
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  const size_t chunks[] = {(size_t)1 << 30, (size_t)256 << 20, (size_t)64 << 20, (size_t)1 << 30, (size_t)256 << 20, (size_t)64 << 20, (size_t)1 << 30, (size_t)256 << 20, (size_t)2 << 30, (size_t)4 << 30, (size_t)8 << 30, (size_t)2 << 30};
  const size_t aligns[] = {0};
  for (size_t chunk : chunks) {
    {
      void* p = nullptr;
      CK(hipMalloc(&p, bytes));
      printf("hipMalloc %zu GiB                  va %p: %7.0f GB/s\n", gib, p, fill_rate(p, bytes));
      CK(hipFree(p));
    }
    for (size_t align : aligns) {
      void* va = nullptr;
      CK(hipMemAddressReserve(&va, bytes, align, nullptr, 0));
      std::vector<hipMemGenericAllocationHandle_t> hs;
      bool ok = true;
      for (size_t off = 0; off < bytes && ok; off += chunk) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { ok = false; break; }
        hs.push_back(h);
        if (hipMemMap((char*)va + off, chunk, 0, h, 0) != hipSuccess) { ok = false; break; }
      }
      if (ok && hipMemSetAccess(va, bytes, &acc, 1) != hipSuccess) ok = false;
      if (ok) printf("vmm chunk %8zu KiB align %8zu KiB va %p: %7.0f GB/s\n", chunk >> 10, align >> 10, va, fill_rate(va, bytes));
      else printf("vmm chunk %8zu KiB align %8zu KiB: failed (%s)\n", chunk >> 10, align >> 10, hipGetErrorString(hipGetLastError()));
      (void)hipMemUnmap(va, bytes);
      for (auto h : hs) (void)hipMemRelease(h);
      (void)hipMemAddressFree(va, bytes);
    }
  }
  return 0;
}
