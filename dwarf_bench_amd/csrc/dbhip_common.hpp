// dbhip_common.hpp — shared device/host helpers for the gfx950 dwarf kernels.
// Wave = 64 lanes everywhere (CDNA4); nothing here is written for 32-wide warps.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/dbhip.h"

namespace dbhip {

constexpr int kWave = 64;

// native 16-byte vectors (clang ext_vector_type: usable with __builtin_nontemporal_load/store)
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));  // 16-byte access at 8-byte alignment
constexpr size_t kWsAlign = 256;   // workspace alignment required from callers
constexpr size_t kWsHeader = 256;  // [0] status word, rest reserved; cleared by every call

// ---------------------------------------------------------------------------------------------
// host-side helpers
// ---------------------------------------------------------------------------------------------
struct DeviceInfo {
  int cus = 0;
  int wave = 0;
  bool ok = false;
  bool gfx950 = false;  // the architecture this library is written and checked for (allow-list of the sort's LDS-atomic ranking)
};
const DeviceInfo &current_device_info();  // cached per device (dbhip_util.hip)

inline hipStream_t as_stream(dbhip_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline bool ws_ok(const void *ws, size_t have, size_t need) {
  return ws != nullptr && (reinterpret_cast<uintptr_t>(ws) % kWsAlign) == 0 && have >= need;
}
inline int launch_status() { return static_cast<int>(hipGetLastError()); }

// Fill `bytes` (a multiple of 4, `p` 4-byte aligned) with the byte `value`, asynchronously, by a kernel of this
// library (dbhip_util.hip).  Used instead of hipMemsetAsync everywhere: a captured hipMemsetAsync becomes a graph
// memset node, and replaying those left pointer-like garbage in the workspace header on ROCm 7.2
// (tests/test_gpu_graph.py); a plain kernel node replays correctly, and costs 2 us instead of the 4.5 us of
// __amd_rocclr_fillBufferAligned.
hipError_t fill_async(void *p, int value, size_t bytes, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt(0), i.e. it
// would wait for every global load the wave has in flight — exactly the prefetch a software
// pipeline wants to keep flying across the barrier.  Register uses of loaded values are still
// guarded by the compiler's own counted waits.
__device__ __forceinline__ void wg_barrier_lds_only() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// number of set bits of `mask` in lanes below the calling lane
__device__ __forceinline__ unsigned mbcnt(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(mask >> 32),
                                   __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(mask), 0u));
}

// Inclusive wave64 prefix sum in 6 DPP steps: row_shr 1/2/4/8 inside each 16-lane row, then
// row_bcast:15 into rows 1 and 3, then row_bcast:31 into rows 2 and 3 (gfx9/CDNA DPP controls).
__device__ __forceinline__ unsigned wave_inclusive_scan(unsigned v) {
  v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
  v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
  return v;
}

__device__ __forceinline__ unsigned wave_reduce_add(unsigned v) {
  return __builtin_amdgcn_readlane(wave_inclusive_scan(v), 63);
}

__device__ __forceinline__ unsigned long long wave_reduce_add_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}


// splitmix64-style counter hash shared with dbo_mix64 in oracle/dbo.c (must stay bit-identical)
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t seed, uint64_t i) {
  uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull;
  z ^= z >> 30;
  z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27;
  z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  return z;
}

// Murmur3 32-bit finaliser: the slot/partition hash of the join and group-by tables.
__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}

}  // namespace dbhip
