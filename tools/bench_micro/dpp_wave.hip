// which lane does lane i read under the whole-wave DPP shifts of gfx9 (wave_shl:1 / wave_rol:1 / wave_shr:1 / wave_ror:1)?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* p)
{
  const int v = (int)threadIdx.x;
  p[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);       // wave_shl:1
  p[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x134, 0xf, 0xf, false);  // wave_rol:1
  p[128 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false); // wave_shr:1
  p[192 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x13C, 0xf, 0xf, false); // wave_ror:1
}
int main()
{
  int* d;
  int h[256];
  hipMalloc(&d, sizeof h);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* names[4] = {"wave_shl:1", "wave_rol:1", "wave_shr:1", "wave_ror:1"};
  for (int t = 0; t < 4; ++t) {
    printf("%s: lane 0 <- %d, lane 1 <- %d, lane 15 <- %d, lane 16 <- %d, lane 31 <- %d, lane 32 <- %d, lane 62 <- %d, lane 63 <- %d\n", names[t],
           h[64 * t], h[64 * t + 1], h[64 * t + 15], h[64 * t + 16], h[64 * t + 31], h[64 * t + 32], h[64 * t + 62], h[64 * t + 63]);
  }
  return 0;
}
