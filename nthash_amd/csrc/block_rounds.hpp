// block_rounds.hpp -- placing what the tiles of a kernel emit into ONE dense output, without a second pass and without a
// wave that waits (round 4; used by minimizer_w_kernel.hpp and minimizer_fused_kernel.hpp).
//
// The tiles are dealt in rounds: tile = (round * blocks + block) * waves + wave.  A wave that has finished its tile of
// round rd knows how many items the tile emits; where they go depends on every tile before it.
//   arrive      the wave leaves its count in LDS; the wave of the block that arrives LAST adds the block's counts up and
//               publishes the sum at once (one 64-bit status word per block-round: the blocks behind need it);
//   look-back   over the earlier block-rounds, 256 per hop (4 words per lane): one hop reaches the round before, whose
//               INCLUSIVE counts are published.  It is done a round LATER, by the wave that arrives FIRST at round rd + 1:
//               everything it needs was published a tile's time ago, and the wave that does it is the one with time to
//               spare -- with the last wave of round rd looking back at once, the look-back's latency sat on the block's
//               critical path in every round (7.8 ms against 5.4 ms without any look-back: the minimizer kernel, 20 M reads);
//   offset_of   every wave's offset of round rd, left in LDS by the look-back; a wave asks for it one round later (its
//               items parked in LDS meanwhile) and finds it there.  A wave that cannot park its items claims the round's
//               look-back early (try_lead) and waits for it.
// What it replaced, measured on the minimizer kernel: a ticket per tile -- one device-scope atomic on one address --
// serialises at ~40 ns per tile (161 ms for 4 M tiles); a look-back per TILE with the waves waiting cost more than the
// hashing (every poll is a round trip over the fabric, 3072 waves polling).
// The look-back needs the blocks of a round to run together: the grid is at most one block per CU and goes out as a
// COOPERATIVE launch (hipLaunchCooperativeKernel: every block resident, or a launch error the host answers with another
// path -- round 5; a plain launch only promised that while nothing else held the device).  What is left is time: blocks that
// get their CUs late because another context's kernel sits there.  A leader that waits 50 ms sets *abort_flag, every later
// wait ends at once, the offsets are garbage from there on (every store must stay inside the caller's arrays) and the host
// repeats the call on another path (tests/test_gpu_minimizer_rounds.py forces both: NTHIP_TUNE_MZ_GRID oversizes the grid on
// a plain launch, NTHIP_TUNE_MZ_TIMEOUT_US shortens the wait).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace ntamd {

constexpr unsigned long long BR_FLAG_A = 1ull << 62; // a block-round's own count is known
constexpr unsigned long long BR_FLAG_P = 1ull << 63; // ... and the count of everything up to and including it
constexpr unsigned long long BR_VALUE = (1ull << 62) - 1ull;
// LDS words of a block: arrive[2], tag[2], asum_tag[2], claim[2], bsum[2], pad[2], woff[2][16] (u64), agg[2][16], wrel[2][16]
constexpr uint32_t BR_CTRL_DWORDS = 12 + 2 * 2 * 16 + 2 * 16 + 2 * 16;
#ifndef BR_ABL_NOLEAD
#define BR_ABL_NOLEAD 0 // ablation (wrong placement): no look-back
#endif

__device__ __forceinline__ uint32_t br_wave_incl_add32(uint32_t v)
{
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
  return v;
}

struct BlockRounds {
  uint32_t* arrive;            // [2] waves that finished the round's tile
  volatile uint32_t* tag;      // [2] round + 1 once the round's offsets are in woff
  volatile uint32_t* asum_tag; // [2] round + 1 once the block's count of the round is in bsum / wrel
  uint32_t* claim;             // [2] round + 1 once a wave has taken the round's look-back
  uint32_t* bsum;              // [2]
  uint64_t* woff;              // [2][16]
  uint32_t* agg;               // [2][16]
  uint32_t* wrel;              // [2][16]
  unsigned long long* status;  // global: [(n_rounds + 1) * blocks], zeroed by the host
  uint32_t* abort_flag;        // global, zeroed by the host
  uint64_t* grand_total;       // global: what everything emits (written by the leader of the last block-round)
  uint64_t* closing_offset;    // global or NULL: the same number once more (the closing entry of a CSR offsets array)
  uint32_t lane, wave, waves, n_rounds;
  uint64_t timeout_ticks;      // 100 MHz ticks a leader waits for a predecessor before it gives the launch up (5 000 000: 50 ms)

  __device__ __forceinline__ void init(uint32_t* ctrl, uint32_t lane_, uint32_t wave_, uint32_t waves_, uint32_t n_rounds_,
                                       unsigned long long* status_, uint32_t* abort_, uint64_t* total_, uint64_t* closing_,
                                       uint32_t timeout_us = 0)
  {
    timeout_ticks = timeout_us ? (uint64_t)timeout_us * 100ull : 5000000ull;
    arrive = ctrl;
    tag = ctrl + 2;
    asum_tag = ctrl + 4;
    claim = ctrl + 6;
    bsum = ctrl + 8;
    woff = (uint64_t*)(ctrl + 12);
    agg = ctrl + 12 + 64;
    wrel = agg + 32;
    status = status_;
    abort_flag = abort_;
    grand_total = total_;
    closing_offset = closing_;
    lane = lane_;
    wave = wave_;
    waves = waves_;
    n_rounds = n_rounds_;
  }

  // the wave that arrives LAST at round rd
  __device__ __forceinline__ void sum_round(uint32_t rd)
  {
    const uint32_t par = rd & 1u;
    const uint32_t v = lane < waves ? agg[par * 16u + lane] : 0u;
    const uint32_t incl = br_wave_incl_add32(v);
    const uint32_t sum = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    const uint64_t br = (uint64_t)rd * gridDim.x + blockIdx.x;
    if (lane == 0)
      __hip_atomic_store(status + br, (br == 0 ? BR_FLAG_P : BR_FLAG_A) | (unsigned long long)sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane < waves) wrel[par * 16u + lane] = incl - v;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    if (lane == 0) {
      bsum[par] = sum;
      arrive[par] = 0; // (next used two rounds on: a wave gets there only behind this round's offsets)
      asum_tag[par] = rd + 1u;
    }
  }

  // the look-back of round rd, by whichever wave claims it first
  __device__ __forceinline__ void try_lead(uint32_t rd)
  {
    const uint32_t par = rd & 1u;
    uint32_t before = 0;
    if (lane == 0) before = __hip_atomic_fetch_max(claim + par, rd + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    before = (uint32_t)__builtin_amdgcn_readfirstlane((int)before);
    if (before >= rd + 1u) return;
    while (asum_tag[par] != rd + 1u) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    const uint32_t sum = bsum[par];
    const uint64_t br = (uint64_t)rd * gridDim.x + blockIdx.x;
    uint64_t excl = 0;
    if (br != 0 && !BR_ABL_NOLEAD) {
      int64_t look = (int64_t)br - 1 - (int64_t)lane;
      bool done = false, aborted = false;
      while (!done && !aborted) {
        unsigned long long s[4];
        bool have[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
          have[j] = look - 64 * (int64_t)j >= 0;
          s[j] = 0;
        }
        uint64_t t_wait = 0;
        for (uint32_t spins = 0;; ++spins) {
#pragma unroll
          for (uint32_t j = 0; j < 4; ++j)
            if (have[j] && s[j] == 0ull) s[j] = __hip_atomic_load(status + (look - 64 * (int64_t)j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          // (only what lies nearer than the nearest inclusive count has to be there)
          uint32_t need = 4;
#pragma unroll
          for (uint32_t j = 4; j-- > 0;)
            if (__ballot(have[j] && (s[j] & BR_FLAG_P) != 0ull) != 0ull) need = j + 1u;
          bool missing = false;
#pragma unroll
          for (uint32_t j = 0; j < 4; ++j)
            if (j < need) missing = missing || (have[j] && s[j] == 0ull);
          if (__ballot(missing) == 0ull) break;
          __builtin_amdgcn_s_sleep(8);
          if ((spins & 63u) == 63u) { // (rare: a block is late, or is not running at all)
            const uint64_t now = __builtin_amdgcn_s_memrealtime(); // 100 MHz
            if (t_wait == 0) t_wait = now;
            const bool late = now - t_wait > timeout_ticks;
            if (late && lane == 0) __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (late || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
              aborted = true; // (the offsets are garbage from here on)
              break;
            }
          }
        }
        if (aborted) break;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
          if (done) break;
          const uint64_t pmask = __ballot(have[j] && (s[j] & BR_FLAG_P) != 0ull);
          const uint32_t first_p = pmask ? (uint32_t)__builtin_ctzll(pmask) : 64u;
          const uint32_t mine = have[j] && lane < first_p ? (uint32_t)(s[j] & BR_VALUE) : 0u;
          excl += (uint32_t)__builtin_amdgcn_readlane((int)br_wave_incl_add32(mine), 63);
          if (pmask) {
            const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)s[j], (int)first_p, 64);
            const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(s[j] >> 32), (int)first_p, 64);
            excl += (((uint64_t)hi << 32) | lo) & BR_VALUE;
            done = true;
          }
        }
        look -= 256;
      }
      if (lane == 0)
        __hip_atomic_store(status + br, BR_FLAG_P | (excl + sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane < waves) woff[par * 16u + lane] = excl + wrel[par * 16u + lane];
    if (rd == n_rounds - 1u && blockIdx.x == gridDim.x - 1u && lane == 0) {
      if (closing_offset) *closing_offset = excl + sum;
      *grand_total = excl + sum;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    if (lane == 0) tag[par] = rd + 1u;
  }

  // this wave's tile of round rd emits `count` items (0 for a wave without a tile; the round after the last: every wave, 0)
  __device__ __forceinline__ void arrive_round(uint32_t rd, uint32_t count)
  {
    const uint32_t par = rd & 1u;
    uint32_t old = 0;
    if (lane == 0) {
      agg[par * 16u + wave] = count;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      old = __hip_atomic_fetch_add(arrive + par, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
    if (old == waves - 1u) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
      sum_round(rd);
    }
    if (old == 0u && rd != 0u) try_lead(rd - 1u);
  }

  // where this wave's items of round rd go (waits for the round's look-back: there by the time a parked tile asks)
  __device__ __forceinline__ uint64_t offset_of(uint32_t rd)
  {
    const uint32_t par = rd & 1u;
    while (tag[par] != rd + 1u) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    return woff[par * 16u + wave];
  }
};

} // namespace ntamd
