// Appending runs of ~64 four-byte entries to 256 bucket lists, three ways (what the partition levels of the binned Bloom
// consumers do; tools/bench_micro/runs_write.hip measured single scattered runs):
//   MODE 0  shared cursors: a bucket's next run goes behind the last one of ANY block (one global atomic per run) -- today
//   MODE 1  block-private chunks: every block appends to its OWN piece of every bucket, run behind run (partial lines are
//           completed by the same CU a few microseconds later: do the L2s merge them?)
//   MODE 2  as 1, every run padded to whole 128-byte lines (aligned pieces, ~25 % more bytes)
// A block = 16 waves; per step wave w writes one run to each of its 16 buckets; runs of 48..80 entries.
//   hipcc --offload-arch=gfx950 -O3 tools/bench_micro/runs_append.hip -o /tmp/ra && /tmp/ra
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ inline uint32_t mix(uint32_t x)
{
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int MODE>
__global__ __launch_bounds__(1024) void k(uint32_t* out, uint32_t* cursors, uint32_t steps, uint64_t bucket_cap, uint64_t piece_cap, uint32_t sleep)
{
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t mycur = 0; // lane u < 16: the private cursor of bucket wave * 16 + u
  for (uint32_t s = 0; s < steps; ++s) {
#pragma unroll 4
    for (uint32_t u = 0; u < 16; ++u) {
      const uint32_t b = wave * 16u + u;
      const uint32_t len = 48u + (mix(s * 7919u + b * 31u + blockIdx.x) % 33u);
      uint64_t at;
      if (MODE == 0) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&cursors[b * 32u], len);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        at = (uint64_t)b * bucket_cap + base;
      } else {
        const uint32_t cur = (uint32_t)__builtin_amdgcn_readlane((int)mycur, (int)u);
        at = (uint64_t)b * bucket_cap + (uint64_t)blockIdx.x * piece_cap + cur;
        const uint32_t adv = MODE == 2 ? (len + 31u) & ~31u : len;
        if (lane == u) mycur += adv;
      }
      uint32_t* dst = out + at;
      const uint32_t n = MODE == 2 ? (len + 31u) & ~31u : len;
      if (lane < n) dst[lane] = s;
      if (lane + 64u < n) dst[lane + 64u] = s;
    }
    if (sleep) __builtin_amdgcn_s_sleep(127);
  }
}

template <int MODE>
void run(uint32_t* buf, uint32_t* cursors, uint64_t total_entries, uint32_t blocks, uint32_t sleep)
{
  const uint32_t steps = (uint32_t)(total_entries / ((uint64_t)blocks * 256 * 64));
  const uint64_t piece_cap = ((uint64_t)steps * 96 + 127) & ~127ull; // room for the longest (padded) runs
  const uint64_t bucket_cap = piece_cap * blocks;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int it = 0; it < 3; ++it) {
    (void)hipMemset(cursors, 0, 256 * 32 * 4);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, buf, cursors, steps, bucket_cap, piece_cap, sleep);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const char* names[3] = {"shared cursors (atomics)     ", "block-private, run behind run", "block-private, whole lines   "};
  printf("blocks %4u sleep %u  %s: %7.3f ms  %6.2f TB/s of payload\n", blocks, sleep, names[MODE], best,
         (double)steps * blocks * 256 * 64 * 4 / best / 1e9);
}

int main()
{
  const uint64_t entries = 2ull << 30; // 8 GiB of payload
  uint32_t *buf, *cursors;
  if (hipMalloc(&buf, entries * 4 * 2) != hipSuccess) return 1;
  (void)hipMalloc(&cursors, 256 * 32 * 4);
  (void)hipMemset(buf, 0, entries * 4 * 2);
  for (uint32_t sleep = 0; sleep < 2; ++sleep)
    for (uint32_t blocks : {256u, 512u}) {
      run<0>(buf, cursors, entries, blocks, sleep);
      run<1>(buf, cursors, entries, blocks, sleep);
      run<2>(buf, cursors, entries, blocks, sleep);
    }
  return 0;
}
