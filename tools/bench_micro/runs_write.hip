// What does HBM make of scattered RUNS of 4-byte entries?  The partition levels of the binned consumers append runs of
// ~64 entries (256 B) at cursors that sit anywhere: every run touches three 128-byte lines, two of them in part.
//   hipcc --offload-arch=gfx950 -O3 tools/bench_micro/runs_write.hip -o /tmp/runs_write && /tmp/runs_write
// MODE 0: runs of LEN dwords at dword-granular places (slot * LEN + a shift of 0..31 dwords)
// MODE 1: the same runs at places aligned to 128 B
// A wave writes 4 runs per step (as bloom_copy_out does), lanes 0..LEN-1 a dword each.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int LEN, int MODE>
__global__ __launch_bounds__(1024) void k(uint32_t* out, uint64_t n_runs, uint64_t slots)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (uint64_t r = wave * 4; r < n_runs; r += n_waves * 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint64_t rr = r + u;
      const uint64_t slot = (rr * 0x9E3779B97F4A7C15ull) % slots; // scattered
      const uint32_t shift = MODE == 0 ? (uint32_t)((rr * 2654435761u) >> 7) & 31u : 0u;
      uint32_t* dst = out + slot * (LEN + 32) + shift;
      for (uint32_t j = lane; j < LEN; j += 64) dst[j] = (uint32_t)rr;
    }
  }
}

template <int LEN, int MODE>
void run(uint32_t* buf, uint64_t bytes_total)
{
  const uint64_t n_runs = bytes_total / (LEN * 4);
  const uint64_t slots = n_runs; // every run its own place (no overlap), places LEN + 32 dwords apart
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<LEN, MODE>), dim3(256 * 2), dim3(1024), 0, 0, buf, n_runs, slots);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  printf("runs of %4d B %s: %7.3f ms  %6.2f TB/s of payload\n", LEN * 4, MODE ? "aligned to 128 B" : "at any dword     ", best,
         (double)n_runs * LEN * 4 / best / 1e9);
}

int main()
{
  const uint64_t payload = 8ull << 30;
  uint32_t* buf;
  if (hipMalloc(&buf, payload * 2 + (1 << 20)) != hipSuccess) return 1; // (LEN + 32 spacing: at most 2 x)
  hipMemset(buf, 0, payload * 2);
  run<32, 0>(buf, payload);
  run<32, 1>(buf, payload);
  run<64, 0>(buf, payload);
  run<64, 1>(buf, payload);
  run<128, 0>(buf, payload);
  run<128, 1>(buf, payload);
  run<256, 0>(buf, payload);
  run<256, 1>(buf, payload);
  return 0;
}
