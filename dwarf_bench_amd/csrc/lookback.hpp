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
//   * a tile index must only ever be held by a RUNNING workgroup, so that every wait terminates whatever
//     else shares the GPU: scan.hip hands tiles out by ticket (a returning atomic); join.hip's counter
//     scan uses a persistent grid no larger than its guaranteed residency.  Spins are time-bounded
//     anyway; a timeout sets DBHIP_DEV_SPIN_TIMEOUT in the workspace status word and the kernel drains.
// Used only where the data is small or latency-tolerant: with the whole chip streaming, a poll costs
// 3.5-4.5 us and concentrated polling overloads single HBM channels (see scan.hip, radix.hip), which
// is why the large-input paths avoid look-back altogether.
#pragma once
#include "dbhip_common.hpp"

namespace dbhip {

constexpr unsigned long long kLb64Shift = 62;
constexpr unsigned long long kLb64Aggregate = 1ull << kLb64Shift;
constexpr unsigned long long kLb64Inclusive = 2ull << kLb64Shift;
constexpr unsigned long long kLb64Value = (1ull << kLb64Shift) - 1;

constexpr unsigned long long kSpinLimitTicks = 200000000ull;  // 2 s of s_memrealtime (100 MHz)
constexpr unsigned long long kSpinCheckTicks = 100000ull;     // after 1 ms also watch the status word

// One failed poll: returns true when the waiter must give up (its own 2 s limit, or another
// workgroup already timed out — so a broken launch drains in ~2 s instead of 2 s per tile).
__device__ __forceinline__ bool spin_should_abort(unsigned long long t0, unsigned *status) {
  const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - t0;
  if (dt > kSpinLimitTicks) return true;
  if (dt > kSpinCheckTicks && (ld_agent(status) & DBHIP_DEV_SPIN_TIMEOUT)) return true;
  __builtin_amdgcn_s_sleep(1);
  return false;
}

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
      if (spin_should_abort(t0, status)) {
        if (lane == 0) atomicOr(status, DBHIP_DEV_SPIN_TIMEOUT);
        return excl;
      }
      continue;
    }
    const unsigned long long mine = static_cast<int>(lane) <= first_inc ? (g & kLb64Value) : 0ull;
    excl += wave_reduce_add_u64(mine);
    if (first_inc < kWave) return excl;
    window_end -= kWave;
  }
}

}  // namespace dbhip
