// scan_rate.hip -- how fast can 3 GB of bases be LOOKED AT (a non-base anywhere in a 1200-byte tile -> a bit)?  The count
// pass of the compact contract reads every base once more (0.8 ms per 3 GB: 3.7 TB/s); this is the floor of such a pass:
// grid-stride 16-byte loads, U in flight per thread, one byte test each, a flag per chunk of 75 vectors.
//   hipcc --offload-arch=gfx950 -O3 tools/bench_micro/scan_rate.hip -o tools/bench_micro/scan_rate_bin && tools/bench_micro/scan_rate_bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t non_base4(uint32_t w)
{
  const uint32_t t = (w >> 1) & 0x03030303u;
  uint32_t x = w | 0x20202020u;
  const uint32_t ubit = (x >> 4) & 0x01010101u;
  x = x & ~ubit;
  return x ^ __builtin_amdgcn_perm(0u, 0x67746361u, t);
}
template <int U>
__global__ __launch_bounds__(256) void scan_kernel(const uint4* __restrict__ v, uint64_t n_vec, uint32_t* __restrict__ flags)
{
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n_vec; i0 += stride * U) {
    uint4 x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = i0 + u * stride < n_vec ? v[i0 + u * stride] : make_uint4(0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t bad = non_base4(x[u].x) | non_base4(x[u].y) | non_base4(x[u].z) | non_base4(x[u].w);
      if (bad) atomicOr(&flags[(i0 + u * stride) / 75 / 32], 1u << (((i0 + u * stride) / 75) & 31));
    }
  }
}
int main()
{
  const uint64_t bytes = 3000000000ull, n_vec = bytes / 16;
  uint4* d; uint32_t* f;
  hipMalloc(&d, bytes); hipMalloc(&f, n_vec / 75 / 8 + 64);
  hipMemset(d, 'A', bytes); hipMemset(f, 0, n_vec / 75 / 8 + 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](auto kern, const char* name, int blocks_per_cu) {
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(256 * blocks_per_cu), dim3(256), 0, 0, d, n_vec, f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%s blocks/CU %d: %.3f ms  %.2f TB/s\n", name, blocks_per_cu, best, bytes / best / 1e9);
  };
  for (int b : {4, 8, 16}) { run(scan_kernel<1>, "U=1", b); run(scan_kernel<2>, "U=2", b); run(scan_kernel<4>, "U=4", b); run(scan_kernel<8>, "U=8", b); }
  return 0;
}
