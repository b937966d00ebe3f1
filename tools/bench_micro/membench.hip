// membench.hip -- HBM write-pattern microbenchmarks for MI355X (measurement aid,
// not part of the product): what write bandwidth do different store shapes reach?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

// fully coalesced fill: consecutive lanes write consecutive 16 B
template <bool NT>
__global__ __launch_bounds__(256) void fill_linear(uint4* dst, uint64_t n16, uint32_t v)
{
  const uint4 val = make_uint4(v, v + 1, v + 2, v + 3);
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
    if (NT) __builtin_nontemporal_store(*(const v4u*)&val, (v4u*)(dst + i)); else dst[i] = val;
  }
}

// "rows" pattern: a wave owns 64 records of REC bytes (contiguous region of 64*REC bytes);
// it writes them in passes; each pass writes ROW bytes of every record (ROW/16 lanes per record).
// Emulates the kmer kernel's flush: REC=960, ROW=128.
template <int ROW, bool NT>
__global__ __launch_bounds__(256) void fill_rows(uint8_t* dst, uint64_t n_rec, uint32_t rec_bytes, uint32_t v)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t wave_global = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  constexpr uint32_t LPR = ROW / 16;      // lanes per record-row
  constexpr uint32_t RPI = 64 / LPR;      // records per store instruction
  const uint4 val = make_uint4(v, lane, v + 2, v + 3);
  const uint32_t passes = (rec_bytes + ROW - 1) / ROW;
  for (uint64_t w = wave_global; w * 64 < n_rec; w += n_waves) {
    uint8_t* base = dst + w * 64 * (uint64_t)rec_bytes;
    for (uint32_t p = 0; p < passes; ++p) {
#pragma unroll
      for (uint32_t s = 0; s < 64 / RPI; ++s) {
        const uint32_t rec = s * RPI + lane / LPR;
        const uint32_t off = p * ROW + (lane % LPR) * 16;
        if (off < rec_bytes && w * 64 + rec < n_rec) {
          uint4* q = (uint4*)(base + (uint64_t)rec * rec_bytes + off);
          if (NT) __builtin_nontemporal_store(*(const v4u*)&val, (v4u*)q); else *q = val;
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void copy_linear(uint4* dst, const uint4* src, uint64_t n16)
{
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void read_linear(const uint4* src, uint64_t n16, uint32_t* out)
{
  uint32_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) { uint4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345u) *out = acc;
}

template <typename F> float timeit(F f, int reps = 5)
{
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int i = 0; i < reps; i++) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
  return best;
}

int main(int argc, char** argv)
{
  const uint64_t n_rec = argc > 1 ? strtoull(argv[1], 0, 10) : 20000000ull;
  const uint32_t rec = 960;
  const uint64_t bytes = n_rec * rec;
  uint8_t *d, *s; uint32_t* flag;
  CK(hipMalloc(&d, bytes + 4096)); CK(hipMalloc(&s, bytes + 4096)); CK(hipMalloc(&flag, 4));
  CK(hipMemset(s, 1, bytes));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s CUs %d, buffer %.2f GB\n", prop.name, cus, bytes / 1e9);
  for (int bpc : {2, 4, 8, 16}) {
    const int grid = cus * bpc;
    float ms;
    ms = timeit([&] { hipLaunchKernelGGL(fill_linear<false>, dim3(grid), dim3(256), 0, 0, (uint4*)d, bytes / 16, 7u); });
    printf("fill_linear      grid=%5d  %.3f ms  %.0f GB/s\n", grid, ms, bytes / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(fill_linear<true>, dim3(grid), dim3(256), 0, 0, (uint4*)d, bytes / 16, 7u); });
    printf("fill_linear NT   grid=%5d  %.3f ms  %.0f GB/s\n", grid, ms, bytes / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(copy_linear, dim3(grid), dim3(256), 0, 0, (uint4*)d, (const uint4*)s, bytes / 16); });
    printf("copy_linear      grid=%5d  %.3f ms  %.0f GB/s (r+w)\n", grid, ms, 2 * bytes / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(read_linear, dim3(grid), dim3(256), 0, 0, (const uint4*)s, bytes / 16, flag); });
    printf("read_linear      grid=%5d  %.3f ms  %.0f GB/s\n", grid, ms, bytes / ms / 1e6);
  }
  for (int bpc : {3, 6, 8}) {
    const int grid = cus * bpc;
    float ms;
#define ROWS(R, NT) ms = timeit([&] { hipLaunchKernelGGL((fill_rows<R, NT>), dim3(grid), dim3(256), 0, 0, d, n_rec, rec, 7u); }); \
    printf("fill_rows ROW=%4d %s grid=%5d  %.3f ms  %.0f GB/s\n", R, NT ? "NT" : "  ", grid, ms, bytes / ms / 1e6);
    ROWS(64, false) ROWS(128, false) ROWS(256, false) ROWS(512, false) ROWS(1024, false)
    ROWS(128, true) ROWS(256, true)
  }
  return 0;
}
