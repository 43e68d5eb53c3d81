// xscan.hip — exclusive prefix sum of a uint32 column (wrap-around), the scan primitive the reference uses in
// three places: prefix_local_test / prefix_sum_scalar (scan/scan.cl:44-66, tests/scan_tests.cpp:14-21, KAT
// {0,1,1,0,0,1,1} -> {0,0,1,2,2,2,3} at :46-51), oneDPL exclusive_scan behind DPLWrapper::exclusive_scan
// (common/dpcpp/dpl_wrapper/dpl_wrapper.hpp:18-25) and the count -> position step of the OmniSci table
// (common/dpcpp/omnisci_hashtable.hpp:252-254).  dst[0] = init, dst[i] = init + src[0] + ... + src[i-1].
//
// 16-byte aligned columns: ONE launch (xs_single_kernel), every element read once and written once (8n bytes).  An
// 8-wave workgroup takes a 128 KiB chunk by ticket and keeps it in registers (16 rows of 1 KiB per wave, 64 VGPRs)
// while the chunk's total goes through the chunk-granular hand-off of handoff.hpp (the dense scan's mechanism,
// scan.hip); then every wave scans its rows (lane-local prefix of 4, DPP wave scan of the lane totals) and writes
// them with 16-byte stores.
// Other alignments: reduce-then-scan over chunks, three launches, no workgroup waits on another one:
//   xs_sums    one workgroup per chunk: the chunk's sum (16-byte loads)
//   xs_offsets one workgroup: exclusive scan of the <= 4096 chunk sums (+ init)
//   xs_scan    one workgroup per chunk: lane-local prefix of 4 elements, DPP wave scan of the lane totals, wave
//              totals through LDS, the chunk's running offset in a register from tile to tile
//   HBM bytes: 4n (sums) + 4n read + 4n written (scan).
// dst may alias src (in place) on both paths.
#include "dbhip_common.hpp"
#include "handoff.hpp"

namespace dbhip {
namespace {

constexpr int kXsThreads = 256;
constexpr int kXsWaves = kXsThreads / kWave;
constexpr size_t kXsTile = static_cast<size_t>(kXsThreads) * 4;  // 1024 elements per workgroup step
constexpr size_t kXsMaxChunks = 4096;

// single-launch path: 8 waves x 16 rows x 256 elements
constexpr int kXs1Waves = 8, kXs1Rows = 16;
constexpr size_t kXs1WaveElems = static_cast<size_t>(kXs1Rows) * kWave * 4;  // 4096 contiguous elements per wave
constexpr size_t kXs1Chunk = kXs1Waves * kXs1WaveElems;                      // 32768 elements = 128 KiB
inline size_t xs1_chunks(size_t n) { return (n + kXs1Chunk - 1) / kXs1Chunk; }

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
  const size_t three_launch = kWsHeader + (L.chunks + 1) * sizeof(unsigned);
  const size_t single = kWsHeader + xs1_chunks(n) * kGranuleStride * sizeof(unsigned long long);
  L.total = align_up(three_launch > single ? three_launch : single, kWsAlign);
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

struct Xs1Header {
  unsigned status, pad0;
  unsigned long long ticket;
  unsigned pad[60];
};
static_assert(sizeof(Xs1Header) == kWsHeader, "workspace header size");

__global__ __launch_bounds__(kXs1Waves * kWave) void xs_single_kernel(const unsigned *src, size_t n, unsigned init,
                                                                      unsigned *dst, Xs1Header *ws,
                                                                      unsigned long long *granules, size_t num_chunks) {
  __shared__ unsigned s_sum[kXs1Waves];
  __shared__ unsigned long long s_chunk, s_excl;
  const unsigned lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  while (true) {
    __syncthreads();  // s_chunk / s_sum / s_excl of the previous chunk are no longer read
    if (threadIdx.x == 0) s_chunk = __hip_atomic_fetch_add(&ws->ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const size_t chunk = s_chunk;
    if (chunk >= num_chunks) return;  // uniform
    const size_t first = chunk * kXs1Chunk + static_cast<size_t>(wave) * kXs1WaveElems;
    const bool full = first + kXs1WaveElems <= n;  // one decision for the wave's whole slice
    u32x4 r[kXs1Rows];
    if (full) {
      const u32x4 *p = reinterpret_cast<const u32x4 *>(src + first) + lane;
#pragma unroll
      for (int k = 0; k < kXs1Rows; ++k) r[k] = __builtin_nontemporal_load(p + k * kWave);
    } else {
#pragma unroll
      for (int k = 0; k < kXs1Rows; ++k) {
        const size_t e = first + (static_cast<size_t>(k) * kWave + lane) * 4;
        r[k].x = e + 0 < n ? src[e + 0] : 0u;
        r[k].y = e + 1 < n ? src[e + 1] : 0u;
        r[k].z = e + 2 < n ? src[e + 2] : 0u;
        r[k].w = e + 3 < n ? src[e + 3] : 0u;
      }
    }
    unsigned mine = 0;
#pragma unroll
    for (int k = 0; k < kXs1Rows; ++k) mine += r[k].x + r[k].y + r[k].z + r[k].w;
    mine = wave_reduce_add(mine);
    if (lane == 0) s_sum[wave] = mine;
    wg_barrier_lds_only();
    unsigned wave_excl = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kXs1Waves; ++w) {
      const unsigned c = s_sum[w];
      wave_excl += w < static_cast<int>(wave) ? c : 0u;
      total += c;
    }
    if (wave == 0) {  // sums wrap modulo 2^32; the granules add modulo 2^62, of which 2^32 is a divisor
      const unsigned long long excl = chunk_handoff(granules, chunk, total, lane, &ws->status);
      if (lane == 0) s_excl = excl;
    }
    wg_barrier_lds_only();
    unsigned base = init + static_cast<unsigned>(s_excl) + wave_excl;
#pragma unroll
    for (int k = 0; k < kXs1Rows; ++k) {
      const unsigned lane_total = r[k].x + r[k].y + r[k].z + r[k].w;
      const unsigned incl = wave_inclusive_scan(lane_total);
      const unsigned excl = base + incl - lane_total;
      const u32x4 o = u32x4{excl, excl + r[k].x, excl + r[k].x + r[k].y, excl + r[k].x + r[k].y + r[k].z};
      const size_t e = first + (static_cast<size_t>(k) * kWave + lane) * 4;
      if (full) {
        __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(dst + e));
      } else {
        if (e + 0 < n) dst[e + 0] = o.x;
        if (e + 1 < n) dst[e + 1] = o.y;
        if (e + 2 < n) dst[e + 2] = o.z;
        if (e + 3 < n) dst[e + 3] = o.w;
      }
      base += __builtin_amdgcn_readlane(incl, 63);
    }
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
  const bool aligned = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0;
  if (n && aligned) {  // one launch (+ the fill that clears status, ticket and granules)
    const size_t chunks = xs1_chunks(n);
    const hipError_t e1 = fill_async(workspace, 0, kWsHeader + chunks * kGranuleStride * sizeof(unsigned long long), s);
    if (e1 != hipSuccess) return static_cast<int>(e1);
    char *base = static_cast<char *>(workspace);
    const size_t cap = static_cast<size_t>(dev.cus) * 2;  // two 8-wave workgroups per CU (64 data VGPRs per lane)
    hipLaunchKernelGGL(xs_single_kernel, dim3(static_cast<unsigned>(chunks < cap ? chunks : cap)), dim3(kXs1Waves * kWave), 0,
                       s, src, n, init, dst, reinterpret_cast<Xs1Header *>(base),
                       reinterpret_cast<unsigned long long *>(base + kWsHeader), chunks);
    return launch_status();
  }
  const hipError_t e = fill_async(workspace, 0, kWsHeader, s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (n == 0) return DBHIP_OK;
  return xs_launch<false>(src, n, init, dst, workspace, s);
}
