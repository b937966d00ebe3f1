// Random atomic-OR rate into a bit table: device (agent) scope vs workgroup scope.
// Workgroup-scope atomics execute in the issuing XCD's L2; they are only correct when every
// address is touched from ONE XCD -- this benchmark measures what that would buy.
//   hipcc --offload-arch=gfx950 -O3 tools/bench_micro/atomics.hip -o /tmp/atomics && /tmp/atomics
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__device__ inline uint64_t sm64(uint64_t x)
{
  uint64_t z = x + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
__device__ inline uint32_t xcc_id()
{
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xF;
}

// MODE 0: agent scope, whole table.  MODE 1: workgroup scope, whole table (NOT coherent across XCDs: rate only).
// MODE 2: workgroup scope, every XCD works on its own eighth of the table (the coherent way to use MODE 1).
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* tab, uint64_t words, uint64_t per_thread, uint32_t* xcd_seen)
{
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t x = xcc_id();
  if (threadIdx.x == 0) atomicOr(&xcd_seen[x], 1u);
  uint64_t slice = words / 8;
  for (uint64_t i = 0; i < per_thread; ++i) {
    const uint64_t h = sm64(gid * per_thread + i);
    uint64_t w = (h >> 5) % (MODE == 2 ? slice : words);
    if (MODE == 2) w += x * slice;
    const uint32_t bit = 1u << (h & 31);
    if (MODE == 0) __hip_atomic_fetch_or(&tab[w], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_or(&tab[w], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}
__global__ void popc(const uint32_t* tab, uint64_t words, unsigned long long* out)
{
  unsigned long long s = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (uint64_t)gridDim.x * blockDim.x)
    s += __builtin_popcount(tab[i]);
  atomicAdd(out, s);
}

int main()
{
  const uint64_t per_thread = 256;
  const unsigned blocks = 256 * 16;
  uint32_t* seen;
  unsigned long long* d_pc;
  hipMalloc(&seen, 64);
  hipMalloc(&d_pc, 8);
  for (uint64_t bytes : {8ull << 20, 128ull << 20, 1024ull << 20}) {
    const uint64_t words = bytes / 4;
    uint32_t* tab;
    hipMalloc(&tab, bytes);
    unsigned long long pc[3] = {0, 0, 0};
    for (int mode = 0; mode < 3; ++mode) {
      hipMemset(tab, 0, bytes);
      hipMemset(seen, 0, 64);
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, tab, words, per_thread, seen);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, tab, words, per_thread, seen);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, tab, words, per_thread, seen);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      hipMemset(d_pc, 0, 8);
      hipLaunchKernelGGL(popc, dim3(1024), dim3(256), 0, 0, tab, words, d_pc);
      hipMemcpy(&pc[mode], d_pc, 8, hipMemcpyDeviceToHost);
      uint32_t hs[16];
      hipMemcpy(hs, seen, 64, hipMemcpyDeviceToHost);
      int nx = 0;
      for (int i = 0; i < 16; ++i) nx += hs[i] != 0;
      const double n = (double)blocks * 256 * per_thread;
      printf("table %5llu MiB mode %d: %8.3f ms  %7.1f G atomics/s  bits set %llu  (XCDs seen %d)\n",
             (unsigned long long)(bytes >> 20), mode, ms, n / ms / 1e6, pc[mode], nx);
    }
    printf("   lost updates with workgroup scope over the whole table: %lld bits\n", (long long)pc[0] - (long long)pc[1]);
    hipFree(tab);
  }
  return 0;
}
