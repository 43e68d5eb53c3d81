// lookback.hpp — decoupled look-back over per-tile {state, value} granules (single-pass prefix sums
// across the workgroups of ONE launch).
//
// Protocol (placement- and dispatch-order independent, MI355X_MICROARCH "inter-workgroup visibility",
// valid form R2 "the data IS the flag"):
//   * one naturally aligned granule per tile (8 bytes: 2-bit state + 62-bit value, or 4 bytes:
//     2-bit state + 30-bit value), zeroed by a hipMemsetAsync node before the launch;
//   * a tile publishes AGGREGATE(own total) as soon as it has counted, and INCLUSIVE(prefix + total)
//     once its prefix is known — each with ONE agent-scope relaxed store (global_store ... sc1);
//   * readers poll with agent-scope relaxed loads (global_load ... sc1, L1 bypassed);
//   * the grid is persistent and no larger than what is co-resident, tiles are walked in stride, so the
//     lowest unfinished tile always belongs to a running workgroup whose own earlier tiles are done:
//     every wait terminates.  Spins are time-bounded anyway; a timeout sets DBHIP_DEV_SPIN_TIMEOUT in
//     the workspace status word and the kernel still drains.
#pragma once
#include "dbhip_common.hpp"

namespace dbhip {

constexpr unsigned long long kLb64Shift = 62;
constexpr unsigned long long kLb64Aggregate = 1ull << kLb64Shift;
constexpr unsigned long long kLb64Inclusive = 2ull << kLb64Shift;
constexpr unsigned long long kLb64Value = (1ull << kLb64Shift) - 1;

constexpr unsigned kLb32Shift = 30;
constexpr unsigned kLb32Aggregate = 1u << kLb32Shift;
constexpr unsigned kLb32Inclusive = 2u << kLb32Shift;
constexpr unsigned kLb32Value = (1u << kLb32Shift) - 1;

constexpr unsigned long long kSpinLimitTicks = 200000000ull;  // 2 s of s_memrealtime (100 MHz)

// Whole-wave look-back: lane l inspects tile-1-l, the window slides back 64 tiles at a time until a
// tile with an INCLUSIVE prefix is met.  Returns the exclusive prefix of `tile` (same in all lanes).
// Must be called by all 64 lanes of one wave; tile >= 1.
__device__ __forceinline__ unsigned long long lookback_wave64(const unsigned long long *granules,
                                                              size_t tile, unsigned lane,
                                                              unsigned *status) {
  unsigned long long excl = 0;
  long long window_end = static_cast<long long>(tile) - 1;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (true) {
    const long long idx = window_end - static_cast<long long>(lane);
    // tiles below 0: a virtual predecessor with inclusive prefix 0
    const unsigned long long g = idx >= 0 ? ld_agent(granules + idx) : kLb64Inclusive;
    const unsigned state = static_cast<unsigned>(g >> kLb64Shift);
    const unsigned long long inc = __ballot(state == 2u);
    const unsigned long long invalid = __ballot(state == 0u);
    const int first_inc = inc ? __builtin_ctzll(inc) : kWave;
    const unsigned long long need = first_inc >= 63 ? ~0ull : ((2ull << first_inc) - 1ull);
    if (invalid & need) {
      if (__builtin_amdgcn_s_memrealtime() - t0 > kSpinLimitTicks) {
        if (lane == 0) atomicOr(status, DBHIP_DEV_SPIN_TIMEOUT);
        return excl;
      }
      __builtin_amdgcn_s_sleep(1);
      continue;
    }
    const unsigned long long mine = static_cast<int>(lane) <= first_inc ? (g & kLb64Value) : 0ull;
    excl += wave_reduce_add_u64(mine);
    if (first_inc < kWave) return excl;
    window_end -= kWave;
  }
}

// Per-thread serial look-back over 4-byte granules laid out [tile][stride]: thread `d` walks
// tile-1, tile-2, ... for its own column.  Used where every thread of a workgroup owns one column
// (radix digits), so neighbouring threads read neighbouring words.  tile >= 1.
__device__ __forceinline__ unsigned lookback_column32(const unsigned *granules, size_t tile,
                                                      unsigned stride, unsigned column,
                                                      unsigned *status) {
  unsigned excl = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  long long t = static_cast<long long>(tile) - 1;
  while (t >= 0) {
    const unsigned g = ld_agent(granules + static_cast<size_t>(t) * stride + column);
    const unsigned state = g >> kLb32Shift;
    if (state == 0u) {
      if (__builtin_amdgcn_s_memrealtime() - t0 > kSpinLimitTicks) {
        atomicOr(status, DBHIP_DEV_SPIN_TIMEOUT);
        return excl;
      }
      __builtin_amdgcn_s_sleep(1);
      continue;
    }
    excl += g & kLb32Value;
    if (state == 2u) break;
    --t;
  }
  return excl;
}

}  // namespace dbhip
