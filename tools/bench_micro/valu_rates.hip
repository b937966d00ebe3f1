// Issue cost of the integer VALU instructions the hash kernels are made of, in cycles per wave64
// instruction per SIMD (all SIMDs busy, 4 waves per SIMD, 8 independent chains per lane).
//   hipcc --offload-arch=gfx950 -O3 tools/bench_micro/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t iters, uint32_t seed)
{
  uint32_t a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = seed * (threadIdx.x + 1) + i;
    b[i] = seed ^ (i * 0x9E3779B9u + threadIdx.x);
  }
  uint32_t c = seed | 1u, d = seed + 17;
  for (uint32_t it = 0; it < iters; ++it) {
#define ONE(i)                                                                                                   \
  if (OP == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c));                                       \
  if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));                                    \
  if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));                                    \
  if (OP == 3) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(uint64_t*)&a[i & 6]) : "v"(c), "v"(d) : "vcc"); \
  if (OP == 4) asm volatile("v_alignbit_b32 %0, %0, %1, 27" : "+v"(a[i]) : "v"(b[i]));                          \
  if (OP == 5) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(b[i]), "v"(c));                       \
  if (OP == 6) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, %2, %3, vcc"                  \
                            : "+v"(a[i]), "+v"(b[i]) : "v"(c), "v"(d) : "vcc");                                   \
  if (OP == 7) asm volatile("v_bfe_u32 %0, %0, 3, 29" : "+v"(a[i]));                                             \
  if (OP == 8) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(c));                                \
  if (OP == 9) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(c));                               \
  if (OP == 10) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c));                    \
  if (OP == 11) asm volatile("v_cmp_lt_u64 vcc, %0, %1\n\tv_cndmask_b32 %2, %2, %3, vcc"                          \
                             : : "v"(*(uint64_t*)&a[i & 6]), "v"(*(uint64_t*)&b[i & 6]), "v"(c), "v"(d) : "vcc"); \
  if (OP == 12) asm volatile("v_lshlrev_b64 %0, 5, %0" : "+v"(*(uint64_t*)&a[i & 6]));                           \
  if (OP == 13) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(c));                                  \
  if (OP == 14) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));                      \
  if (OP == 15) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c));                      \
  if (OP == 16) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c));                       \
  if (OP == 17) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c) : );                          \
  if (OP == 18) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b[i]));                                        \
  if (OP == 19) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
    REP8(ONE) REP8(ONE) REP8(ONE) REP8(ONE)
#undef ONE
  }
  uint32_t r = 0;
  for (int i = 0; i < 8; ++i) r ^= a[i] ^ b[i];
  if (r == 0x12345678u) out[threadIdx.x] = r + c + d;
}

template <int OP>
double run(const char* name, int instr_per_rep, uint32_t* d_out)
{
  const uint32_t iters = 4000;
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int blocks = prop.multiProcessorCount * 4; // 4 blocks x 4 waves per CU = 4 waves per SIMD
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 100u, 3u);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 3u);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  // wave-instructions per SIMD: 4 waves x iters x 32 reps x instr_per_rep
  const double wave_instr = 4.0 * iters * 32.0 * instr_per_rep;
  const double clk = prop.clockRate * 1e3; // Hz (nominal; the chip may run lower)
  const double cyc = ms * 1e-3 * clk / wave_instr;
  printf("%-34s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (at the nominal %.2f GHz)\n", name, ms, cyc, clk / 1e9);
  return cyc;
}

int main()
{
  uint32_t* d_out;
  (void)hipMalloc((void**)&d_out, 4096);
  run<0>("v_xor_b32", 1, d_out);
  run<5>("v_bitop3_b32 (a^b^c)", 1, d_out);
  run<4>("v_alignbit_b32", 1, d_out);
  run<7>("v_bfe_u32", 1, d_out);
  run<8>("v_lshl_or_b32", 1, d_out);
  run<9>("v_lshl_add_u32", 1, d_out);
  run<10>("v_and_or_b32", 1, d_out);
  run<16>("v_bfi_b32", 1, d_out);
  run<15>("v_perm_b32", 1, d_out);
  run<19>("v_lshrrev_b32", 1, d_out);
  run<18>("v_mov_b32", 1, d_out);
  run<17>("v_cndmask_b32", 1, d_out);
  run<6>("v_add_co_u32 + v_addc_co_u32", 2, d_out);
  run<11>("v_cmp_lt_u64 + v_cndmask_b32", 2, d_out);
  run<12>("v_lshlrev_b64", 1, d_out);
  run<1>("v_mul_lo_u32", 1, d_out);
  run<2>("v_mul_hi_u32", 1, d_out);
  run<3>("v_mad_u64_u32", 1, d_out);
  run<13>("v_mul_u32_u24", 1, d_out);
  run<14>("v_mad_u32_u24", 1, d_out);
  return 0;
}
