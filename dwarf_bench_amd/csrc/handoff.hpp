// handoff.hpp — the chunk-granular hand-off between workgroups used by the two single-launch prefix kernels of the
// library (the dense scan in scan.hip, the exclusive scan in xscan.hip): decoupled look-back over one granule per
// chunk.  A granule is an 8-byte {state, value} word written and read with agent-scope relaxed atomics; a chunk first
// publishes its AGGREGATE (its own total), looks back over its predecessors until it meets one that already knows its
// INCLUSIVE prefix, and then publishes its own inclusive prefix.  Chunks must be taken by ticket (at the moment the
// workgroup starts on them): a chunk's predecessors then always belong to workgroups that are already running and the
// wait terminates whatever else shares the GPU; it is time-bounded anyway (DBHIP_DEV_SPIN_TIMEOUT).
// Every granule has a 128-byte line of its own: with eight granules per line a poll read a line that seven other
// chunks were writing, and the hop grew with the number of chunks in flight.
#pragma once
#include "dbhip_common.hpp"

namespace dbhip {

constexpr unsigned long long kLbShift = 62, kLbAggregate = 1ull << kLbShift, kLbInclusive = 2ull << kLbShift,
                             kLbValue = (1ull << kLbShift) - 1;
constexpr unsigned long long kSpinLimitTicks = 200000000ull;  // 2 s of s_memrealtime (100 MHz)
constexpr size_t kGranuleStride = 16;  // in granules: every chunk's granule has a 128-byte line of its own (packed
                                       // granules: the line a poll reads is being written by eight other chunks)

__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(unsigned long long *p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Whole-wave look-back: lane l inspects chunk-1-l, the window slides back 64 chunks at a time until a chunk with an
// INCLUSIVE prefix is met.  Returns the exclusive prefix of `chunk` (same in all lanes); chunk >= 1.
__device__ __forceinline__ unsigned long long chunk_lookback(const unsigned long long *granules, size_t chunk, unsigned lane,
                                                          unsigned *status) {
  unsigned long long excl = 0;
  long long window_end = static_cast<long long>(chunk) - 1;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (true) {
    const long long idx = window_end - static_cast<long long>(lane);
    const unsigned long long g = idx >= 0 ? ld_agent(granules + idx * kGranuleStride) : kLbInclusive;  // below 0: inclusive prefix 0
    const unsigned state = static_cast<unsigned>(g >> kLbShift);
    const unsigned long long inc = __ballot(state == 2u);
    const unsigned long long invalid = __ballot(state == 0u);
    const int first_inc = inc ? __builtin_ctzll(inc) : kWave;
    const unsigned long long need = first_inc >= 63 ? ~0ull : ((2ull << first_inc) - 1ull);
    if (invalid & need) {  // a predecessor inside the window has not published yet
      if (__builtin_amdgcn_s_memrealtime() - t0 > kSpinLimitTicks) {
        if (lane == 0) atomicOr(status, DBHIP_DEV_SPIN_TIMEOUT);
        return excl;
      }
      __builtin_amdgcn_s_sleep(2);
      continue;
    }
    const unsigned long long mine = static_cast<int>(lane) <= first_inc ? (g & kLbValue) : 0ull;
    excl += wave_reduce_add_u64(mine);
    if (first_inc < kWave) return excl;
    window_end -= kWave;
  }
}

// one chunk's hand-off, by one whole wave: returns the exclusive prefix of `chunk` (values add modulo 2^62)
__device__ __forceinline__ unsigned long long chunk_handoff(unsigned long long *granules, size_t chunk,
                                                            unsigned long long total, unsigned lane, unsigned *status) {
  unsigned long long excl = 0;
  if (chunk == 0) {
    if (lane == 0) st_agent(granules, kLbInclusive | (total & kLbValue));
  } else {
    if (lane == 0) st_agent(granules + chunk * kGranuleStride, kLbAggregate | (total & kLbValue));
    excl = chunk_lookback(granules, chunk, lane, status);
    if (lane == 0) st_agent(granules + chunk * kGranuleStride, kLbInclusive | ((excl + total) & kLbValue));
  }
  return excl;
}

}  // namespace dbhip
