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

// Wide variant: every lane inspects FOUR granules, so one poll covers the 256 tiles below `tile`.
// With at most 256 tiles in flight (one 16-wave workgroup per CU) a single window always reaches a
// tile of the previous round, whose prefix is INCLUSIVE: the look-back is one hop even when the
// whole grid advances in lockstep (each hop costs 2-3 us under streaming load).
__device__ __forceinline__ unsigned long long lookback_wave256(const unsigned long long *granules,
                                                               size_t tile, unsigned lane,
                                                               unsigned *status) {
  unsigned long long excl = 0;
  long long window_end = static_cast<long long>(tile) - 1;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (true) {
    unsigned long long g[4];
    unsigned st[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long idx = window_end - static_cast<long long>(4 * lane + j);
      g[j] = idx >= 0 ? ld_agent(granules + idx) : kLb64Inclusive;
      st[j] = static_cast<unsigned>(g[j] >> kLb64Shift);
    }
    const int fi = st[0] == 2u ? 0 : st[1] == 2u ? 1 : st[2] == 2u ? 2 : st[3] == 2u ? 3 : 4;
    const unsigned long long has_inc = __ballot(fi < 4);
    const int first_lane = has_inc ? __builtin_ctzll(has_inc) : kWave;
    const int take = static_cast<int>(lane) < first_lane ? 4 : (static_cast<int>(lane) == first_lane ? fi + 1 : 0);
    bool bad = false;
    unsigned long long sum = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < take) {
        bad |= st[j] == 0u;
        sum += g[j] & kLb64Value;
      }
    }
    if (__ballot(bad)) {
      if (spin_should_abort(t0, status)) {
        if (lane == 0) atomicOr(status, DBHIP_DEV_SPIN_TIMEOUT);
        return excl;
      }
      continue;
    }
    excl += wave_reduce_add_u64(sum);
    if (first_lane < kWave) return excl;
    window_end -= 4 * kWave;
  }
}

// Split form of the 256-tile window for software pipelining: issue the four polls early, evaluate
// them later.  try_resolve returns true when the window alone yields the exclusive prefix (every
// needed granule published and an INCLUSIVE one found); otherwise the caller falls back to
// lookback_wave256, which re-polls.
__device__ __forceinline__ void lookback256_issue(const unsigned long long *granules, size_t tile,
                                                  unsigned lane, unsigned long long (&g)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long idx = static_cast<long long>(tile) - 1 - static_cast<long long>(4 * lane + j);
    g[j] = idx >= 0 ? ld_agent(granules + idx) : kLb64Inclusive;
  }
}
__device__ __forceinline__ bool lookback256_try_resolve(const unsigned long long (&g)[4], unsigned lane,
                                                        unsigned long long *excl) {
  unsigned st[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) st[j] = static_cast<unsigned>(g[j] >> kLb64Shift);
  const int fi = st[0] == 2u ? 0 : st[1] == 2u ? 1 : st[2] == 2u ? 2 : st[3] == 2u ? 3 : 4;
  const unsigned long long has_inc = __ballot(fi < 4);
  if (!has_inc) return false;
  const int first_lane = __builtin_ctzll(has_inc);
  const int take = static_cast<int>(lane) < first_lane ? 4 : (static_cast<int>(lane) == first_lane ? fi + 1 : 0);
  bool bad = false;
  unsigned long long sum = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j < take) {
      bad |= st[j] == 0u;
      sum += g[j] & kLb64Value;
    }
  }
  if (__ballot(bad)) return false;
  *excl = wave_reduce_add_u64(sum);
  return true;
}

// ---------------------------------------------------------------------------------------------
// 1024-tile windows over 4-byte granules (2-bit state + 30-bit value, prefix sums < 2^30).
//
// Why so wide: with every CU streaming, a poll's round trip is 4-5 us (it queues behind the CU's
// own ~100 KiB of outstanding loads), and INCLUSIVE prefixes advance by at most one window per
// round trip.  A 256-tile window advances ~55 tiles/us — exactly the rate a 2^28-element scan
// produces 120 KiB tiles at 7 TB/s; 1024 tiles per poll gives a 4x margin.
//
// Layout: granule[kLbPad + tile]; the kLbPad entries in front are preset to INCLUSIVE|0 (tiles
// "before 0"), so the first window needs no special case.  Lane l holds the 16 granules
// [end-16-16l, end-1-16l] (end = index of the caller's own granule) as four 16-byte sc1 buffer
// loads; each 4-byte granule is written by one 4-byte sc1 store, so every dword is untorn.
// ---------------------------------------------------------------------------------------------
constexpr unsigned kLbPad = 1024;
constexpr unsigned kLbWindow = 1024;

struct LbWindow {
  u32x4 q[4];
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t lb_make_rsrc(const unsigned *granules,
                                                               unsigned num_granules) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(granules), 0,
                                           static_cast<int>(num_granules * 4u), 0x00020000);
}

// end = granule index one past the nearest granule to inspect
__device__ __forceinline__ void lb1024_issue(__amdgpu_buffer_rsrc_t rsrc, long long end, unsigned lane,
                                             LbWindow &w) {
  const long long lo = end - 16 - 16 * static_cast<long long>(lane);
  if (lo >= 0) {
    const unsigned off = static_cast<unsigned>(lo) * 4u;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      w.q[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16u * j, 0, /*aux: sc1*/ 16);
  } else {  // entirely in front of the padding: "tiles before 0", inclusive prefix 0
#pragma unroll
    for (int j = 0; j < 4; ++j) w.q[j] = u32x4{kLb32Inclusive, kLb32Inclusive, kLb32Inclusive, kLb32Inclusive};
  }
}

// Evaluates a window.  Returns 0 = resolved (*sum = prefix contribution up to and including the
// nearest INCLUSIVE granule), 1 = every granule valid but no INCLUSIVE one (*sum = sum of all 1024
// aggregates: continue with the next window), 2 = a needed granule is not published yet.
__device__ __forceinline__ int lb1024_eval(const LbWindow &w, unsigned lane, unsigned *sum_out) {
  unsigned g[16];
#pragma unroll
  for (int j = 0; j < 4; ++j) {  // g[0] = nearest
    g[4 * j + 0] = w.q[3 - j].w;
    g[4 * j + 1] = w.q[3 - j].z;
    g[4 * j + 2] = w.q[3 - j].y;
    g[4 * j + 3] = w.q[3 - j].x;
  }
  int fi = 16;
#pragma unroll
  for (int j = 15; j >= 0; --j) fi = (g[j] >> kLb32Shift) == 2u ? j : fi;
  const unsigned long long has_inc = __ballot(fi < 16);
  const int first_lane = has_inc ? __builtin_ctzll(has_inc) : kWave;
  const int take = static_cast<int>(lane) < first_lane ? 16 : (static_cast<int>(lane) == first_lane ? fi + 1 : 0);
  bool bad = false;
  unsigned sum = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (j < take) {
      bad |= (g[j] >> kLb32Shift) == 0u;
      sum += g[j] & kLb32Value;
    }
  }
  if (__ballot(bad)) return 2;
  *sum_out = wave_reduce_add(sum);
  return has_inc ? 0 : 1;
}

// Blocking look-back with 1024-tile windows.  `first` may hold a window issued earlier for
// end = own_index (pass have_first = false to start by polling).  Whole wave; returns the
// exclusive prefix of the caller's tile.
__device__ __forceinline__ unsigned lookback1024(__amdgpu_buffer_rsrc_t rsrc, long long own_index,
                                                 unsigned lane, LbWindow &first, bool have_first,
                                                 unsigned *status) {
  unsigned excl = 0;
  long long end = own_index;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  bool have = have_first;
  while (true) {
    if (!have) lb1024_issue(rsrc, end, lane, first);
    have = false;
    unsigned sum = 0;
    const int r = lb1024_eval(first, lane, &sum);
#ifdef DBHIP_SCAN_PROFILE
    if (lane == 0) atomicAdd(status + 4 + (r == 2 ? 16 : r == 1 ? 17 : 18), 1u);
#endif
    if (r == 2) {
      if (spin_should_abort(t0, status)) {
        if (lane == 0) atomicOr(status, DBHIP_DEV_SPIN_TIMEOUT);
        return excl;
      }
      continue;
    }
    excl += sum;
    if (r == 0) return excl;
    end -= kLbWindow;
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
      if (spin_should_abort(t0, status)) {
        atomicOr(status, DBHIP_DEV_SPIN_TIMEOUT);
        return excl;
      }
      continue;
    }
    excl += g & kLb32Value;
    if (state == 2u) break;
    --t;
  }
  return excl;
}

}  // namespace dbhip
