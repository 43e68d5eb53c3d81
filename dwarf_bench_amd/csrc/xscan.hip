// xscan.hip — exclusive prefix sum of a uint32 column (wrap-around), the scan primitive the reference uses in
// three places: prefix_local_test / prefix_sum_scalar (scan/scan.cl:44-66, tests/scan_tests.cpp:14-21, KAT
// {0,1,1,0,0,1,1} -> {0,0,1,2,2,2,3} at :46-51), oneDPL exclusive_scan behind DPLWrapper::exclusive_scan
// (common/dpcpp/dpl_wrapper/dpl_wrapper.hpp:18-25) and the count -> position step of the OmniSci table
// (common/dpcpp/omnisci_hashtable.hpp:252-254).  dst[0] = init, dst[i] = init + src[0] + ... + src[i-1].
//
// Reduce-then-scan over chunks, three launches, no workgroup waits on another one (same reasoning as scan.hip):
//   xs_sums    one workgroup per chunk: the chunk's sum (16-byte loads)
//   xs_offsets one workgroup: exclusive scan of the <= 4096 chunk sums (+ init)
//   xs_scan    one workgroup per chunk: lane-local prefix of 4 elements, DPP wave scan of the lane totals, wave
//              totals through LDS, the chunk's running offset in a register from tile to tile
// HBM bytes: 4n (sums) + 4n read + 4n written (scan).  dst may alias src (in place).
#include "dbhip_common.hpp"

namespace dbhip {
namespace {

constexpr int kXsThreads = 256;
constexpr int kXsWaves = kXsThreads / kWave;
constexpr size_t kXsTile = static_cast<size_t>(kXsThreads) * 4;  // 1024 elements per workgroup step
constexpr size_t kXsMaxChunks = 4096;

struct XsLayout {
  size_t chunk_elems, chunks, total;
};
inline XsLayout xs_layout(size_t n) {
  XsLayout L;
  size_t per = (n + kXsMaxChunks - 1) / kXsMaxChunks;
  per = (per + kXsTile - 1) / kXsTile * kXsTile;
  L.chunk_elems = per ? per : kXsTile;
  L.chunks = (n + L.chunk_elems - 1) / L.chunk_elems;
  if (L.chunks == 0) L.chunks = 1;
  L.total = align_up(kWsHeader + (L.chunks + 1) * sizeof(unsigned), kWsAlign);
  return L;
}

template <bool kAligned>
__device__ __forceinline__ u32x4 xs_load(const unsigned *__restrict__ src, size_t e, size_t hi) {
  if (kAligned && e + 4 <= hi) return *reinterpret_cast<const u32x4 *>(src + e);
  u32x4 v;
  v.x = e + 0 < hi ? src[e + 0] : 0u;
  v.y = e + 1 < hi ? src[e + 1] : 0u;
  v.z = e + 2 < hi ? src[e + 2] : 0u;
  v.w = e + 3 < hi ? src[e + 3] : 0u;
  return v;
}

template <bool kAligned>
__global__ __launch_bounds__(kXsThreads) void xs_sums_kernel(const unsigned *__restrict__ src, size_t n,
                                                             size_t chunk_elems, unsigned *__restrict__ sums) {
  __shared__ unsigned s_w[kXsWaves];
  const size_t lo = static_cast<size_t>(blockIdx.x) * chunk_elems;
  size_t hi = lo + chunk_elems;
  hi = hi < n ? hi : n;
  unsigned acc = 0;
  for (size_t e = lo + static_cast<size_t>(threadIdx.x) * 4; e < hi; e += kXsTile) {
    const u32x4 v = xs_load<kAligned>(src, e, hi);
    acc += v.x + v.y + v.z + v.w;
  }
  acc = wave_reduce_add(acc);
  if ((threadIdx.x & (kWave - 1)) == 0) s_w[threadIdx.x / kWave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned s = 0;
#pragma unroll
    for (int w = 0; w < kXsWaves; ++w) s += s_w[w];
    sums[blockIdx.x] = s;
  }
}

__global__ __launch_bounds__(1024) void xs_offsets_kernel(unsigned *sums, unsigned chunks, unsigned init) {
  __shared__ unsigned s_w[1024 / kWave];
  const unsigned tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  unsigned c[4], mine = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned i = tid * 4 + j;
    c[j] = i < chunks ? sums[i] : 0u;
    mine += c[j];
  }
  const unsigned incl = wave_inclusive_scan(mine);
  if (lane == kWave - 1) s_w[wave] = incl;
  __syncthreads();
  unsigned run = init + incl - mine;
  for (unsigned w = 0; w < wave; ++w) run += s_w[w];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned i = tid * 4 + j;
    if (i < chunks) sums[i] = run;
    run += c[j];
  }
}

template <bool kAligned>
__global__ __launch_bounds__(kXsThreads) void xs_scan_kernel(const unsigned *src, size_t n, size_t chunk_elems,
                                                             const unsigned *__restrict__ offsets, unsigned *dst) {
  __shared__ unsigned s_w[2][kXsWaves];
  const unsigned tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const size_t lo = static_cast<size_t>(blockIdx.x) * chunk_elems;
  size_t hi = lo + chunk_elems;
  hi = hi < n ? hi : n;
  unsigned running = offsets[blockIdx.x];
  unsigned par = 0;
  for (size_t base = lo; base < hi; base += kXsTile, par ^= 1u) {
    const size_t e = base + static_cast<size_t>(tid) * 4;
    const u32x4 v = xs_load<kAligned>(src, e, hi);
    const unsigned mine = v.x + v.y + v.z + v.w;
    const unsigned incl = wave_inclusive_scan(mine);
    if (lane == kWave - 1) s_w[par][wave] = incl;
    __syncthreads();  // two slots alternate with the tile parity: one barrier per tile
    unsigned excl = running + incl - mine, tile_total = 0;
#pragma unroll
    for (int w = 0; w < kXsWaves; ++w) {
      const unsigned t = s_w[par][w];
      excl += w < static_cast<int>(wave) ? t : 0u;
      tile_total += t;
    }
    const u32x4 o = u32x4{excl, excl + v.x, excl + v.x + v.y, excl + v.x + v.y + v.z};
    if (kAligned && e + 4 <= hi) {
      *reinterpret_cast<u32x4 *>(dst + e) = o;
    } else {
      if (e + 0 < hi) dst[e + 0] = o.x;
      if (e + 1 < hi) dst[e + 1] = o.y;
      if (e + 2 < hi) dst[e + 2] = o.z;
      if (e + 3 < hi) dst[e + 3] = o.w;
    }
    running += tile_total;
  }
}

template <bool kAligned>
int xs_launch(const unsigned *src, size_t n, unsigned init, unsigned *dst, void *workspace, hipStream_t s) {
  const XsLayout L = xs_layout(n);
  unsigned *sums = reinterpret_cast<unsigned *>(static_cast<char *>(workspace) + kWsHeader);
  hipLaunchKernelGGL(xs_sums_kernel<kAligned>, dim3(static_cast<unsigned>(L.chunks)), dim3(kXsThreads), 0, s, src, n,
                     L.chunk_elems, sums);
  hipLaunchKernelGGL(xs_offsets_kernel, dim3(1), dim3(1024), 0, s, sums, static_cast<unsigned>(L.chunks), init);
  hipLaunchKernelGGL(xs_scan_kernel<kAligned>, dim3(static_cast<unsigned>(L.chunks)), dim3(kXsThreads), 0, s, src, n,
                     L.chunk_elems, sums, dst);
  return launch_status();
}

}  // namespace
}  // namespace dbhip

using namespace dbhip;

extern "C" size_t dbhip_exclusive_scan_u32_workspace_bytes(size_t n) { return xs_layout(n ? n : 1).total; }

extern "C" int dbhip_exclusive_scan_u32(const uint32_t *src, size_t n, uint32_t init, uint32_t *dst, void *workspace,
                                        size_t workspace_bytes, dbhip_stream_t stream) {
  if (n && (!src || !dst)) return DBHIP_EINVAL;
  if (!ws_ok(workspace, workspace_bytes, dbhip_exclusive_scan_u32_workspace_bytes(n))) return DBHIP_EWORKSPACE;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  hipStream_t s = as_stream(stream);
  const hipError_t e = fill_async(workspace, 0, kWsHeader, s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (n == 0) return DBHIP_OK;
  const bool aligned = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0;
  return aligned ? xs_launch<true>(src, n, init, dst, workspace, s) : xs_launch<false>(src, n, init, dst, workspace, s);
}
