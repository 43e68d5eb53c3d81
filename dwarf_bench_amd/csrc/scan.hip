// scan.hip — dwarf 1: stable stream compaction out = [x in src : x < filter] for gfx950.
//
// Replaces the reference kernel simple_two_pass_scan (scan/scan.cl:3-42: chunked count -> serial
// prefix by work-item 0 -> chunked write, which reads src twice with uncoalesced per-lane chunks)
// and the oneDPL copy_if behind DPLScan (dpl_wrapper.hpp:27-33).  Same logical result
// (scan/scan.cpp:12-17 expected_out_lt), different machine mapping:
//
//   * ONE pass over src.  A persistent grid (<= resident capacity, so every workgroup is
//     co-resident) walks 32 KiB tiles in stride; each tile is read with 16-byte-per-lane
//     coalesced loads (each wave instruction = 1 KiB contiguous), kept in registers, counted,
//     and written once its global offset is known.
//   * The offset comes from a decoupled look-back over one 8-byte {state, value} granule per
//     tile.  Granules are written/read with agent-scope relaxed atomics (global_* ... sc1): the
//     flag and the value travel in one naturally aligned 8-byte store, so no fence is needed and
//     nothing depends on workgroup->XCD placement or dispatch order.
//   * In-wave ranks come from the compare masks themselves: v_cmp -> 64-bit ballot in SGPRs,
//     s_bcnt1 for totals, v_mbcnt_lo/hi for the lane-exclusive prefix.
//
// Algorithmic HBM bytes: 4*n read + 4*out_size written (+ 16 B of granule traffic per 32 KiB tile).
#include <climits>

#include "dbhip_common.hpp"
#include "lookback.hpp"

namespace dbhip {
namespace {

constexpr int kScanThreads = 256;
constexpr int kScanWaves = kScanThreads / kWave;
constexpr int kScanVpt = 8;                                       // int4 loads per lane per tile
constexpr int kScanWaveElems = kWave * kScanVpt * 4;              // 2048 contiguous elements per wave
constexpr int kScanTile = kScanWaveElems * kScanWaves;            // 8192 elements = 32 KiB

struct ScanWs {
  unsigned status;  // DBHIP_DEV_* bits
  unsigned pad[63];
  // followed by one 8-byte granule per tile
};
static_assert(sizeof(ScanWs) == kWsHeader, "workspace header size");

template <bool kAligned, bool kNontemporal>
__global__ __launch_bounds__(kScanThreads) void copy_if_lt_kernel(
    const int *__restrict__ src, size_t n, int filter, int *__restrict__ out,
    unsigned long long *__restrict__ out_size, ScanWs *ws, size_t num_tiles) {
  __shared__ unsigned s_wave_total[kScanWaves];
  __shared__ unsigned long long s_tile_excl;

  unsigned long long *granules = reinterpret_cast<unsigned long long *>(ws + 1);
  const unsigned tid = threadIdx.x;
  const unsigned lane = tid & (kWave - 1);
  const unsigned wave = tid / kWave;

  for (size_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    const size_t wave_base = tile * kScanTile + static_cast<size_t>(wave) * kScanWaveElems;
    i32x4 v[kScanVpt];
    if (kAligned && tile * kScanTile + kScanTile <= n) {
      const i32x4 *p = reinterpret_cast<const i32x4 *>(src + wave_base) + lane;
#pragma unroll
      for (int k = 0; k < kScanVpt; ++k)
        v[k] = kNontemporal ? __builtin_nontemporal_load(p + k * kWave) : p[k * kWave];
    } else {
#pragma unroll
      for (int k = 0; k < kScanVpt; ++k) {
        const size_t e = wave_base + (static_cast<size_t>(k) * kWave + lane) * 4;
        v[k].x = e + 0 < n ? src[e + 0] : INT_MAX;  // INT_MAX never satisfies x < filter
        v[k].y = e + 1 < n ? src[e + 1] : INT_MAX;
        v[k].z = e + 2 < n ? src[e + 2] : INT_MAX;
        v[k].w = e + 3 < n ? src[e + 3] : INT_MAX;
      }
    }

    // ---- count: the compare IS the ballot (v_cmp -> SGPR pair), totals are scalar popcounts
    unsigned wave_total = 0;
#pragma unroll
    for (int k = 0; k < kScanVpt; ++k) {
      wave_total += __builtin_popcountll(__ballot(v[k].x < filter));
      wave_total += __builtin_popcountll(__ballot(v[k].y < filter));
      wave_total += __builtin_popcountll(__ballot(v[k].z < filter));
      wave_total += __builtin_popcountll(__ballot(v[k].w < filter));
    }
    if (lane == 0) s_wave_total[wave] = wave_total;
    __syncthreads();

    unsigned wave_excl = 0, tile_total = 0;
#pragma unroll
    for (int w = 0; w < kScanWaves; ++w) {
      const unsigned t = s_wave_total[w];
      wave_excl += w < static_cast<int>(wave) ? t : 0u;
      tile_total += t;
    }

    if (wave == 0) {
      unsigned long long excl = 0;
      if (tile == 0) {
        if (lane == 0) st_agent(granules, kLb64Inclusive | tile_total);
      } else {
        if (lane == 0) st_agent(granules + tile, kLb64Aggregate | tile_total);
        excl = lookback_wave64(granules, tile, lane, &ws->status);
        if (lane == 0) st_agent(granules + tile, kLb64Inclusive | ((excl + tile_total) & kLb64Value));
      }
      if (lane == 0) {
        s_tile_excl = excl;
        if (tile == num_tiles - 1) *out_size = excl + tile_total;
      }
    }
    __syncthreads();

    if (wave_total != 0) {  // wave-uniform
      int *dst_wave = out + s_tile_excl + wave_excl;
      unsigned row_base = 0;
#pragma unroll
      for (int k = 0; k < kScanVpt; ++k) {
        const bool m0 = v[k].x < filter, m1 = v[k].y < filter, m2 = v[k].z < filter,
                   m3 = v[k].w < filter;
        const unsigned long long b0 = __ballot(m0), b1 = __ballot(m1), b2 = __ballot(m2),
                                 b3 = __ballot(m3);
        if ((b0 | b1 | b2 | b3) == 0) continue;  // scalar branch: most rows are empty at low selectivity
        int *dst = dst_wave + row_base + mbcnt(b0) + mbcnt(b1) + mbcnt(b2) + mbcnt(b3);
        if (m0) *dst++ = v[k].x;
        if (m1) *dst++ = v[k].y;
        if (m2) *dst++ = v[k].z;
        if (m3) *dst++ = v[k].w;
        row_base += __builtin_popcountll(b0) + __builtin_popcountll(b1) +
                    __builtin_popcountll(b2) + __builtin_popcountll(b3);
      }
    }
  }
}

inline int scan_blocks_per_cu() {
  static const int v = [] {
    const char *e = getenv("DBHIP_SCAN_BLOCKS_PER_CU");
    int x = e ? atoi(e) : 0;
    return (x >= 1 && x <= 8) ? x : 4;
  }();
  return v;
}

inline bool scan_nontemporal() {
  static const bool v = [] {
    const char *e = getenv("DBHIP_SCAN_NT");
    return e ? atoi(e) != 0 : true;
  }();
  return v;
}

}  // namespace
}  // namespace dbhip

using namespace dbhip;

extern "C" size_t dbhip_copy_if_lt_i32_workspace_bytes(size_t n) {
  const size_t tiles = (n + kScanTile - 1) / kScanTile;
  return align_up(kWsHeader + (tiles ? tiles : 1) * sizeof(unsigned long long), kWsAlign);
}

extern "C" int dbhip_copy_if_lt_i32(const int32_t *src, size_t n, int32_t filter_value,
                                    int32_t *out, uint64_t *out_size, void *workspace,
                                    size_t workspace_bytes, dbhip_stream_t stream) {
  if (!out_size || (n && (!src || !out))) return DBHIP_EINVAL;
  if (n >= (1ull << 61)) return DBHIP_EINVAL;
  const size_t need = dbhip_copy_if_lt_i32_workspace_bytes(n);
  if (!ws_ok(workspace, workspace_bytes, need)) return DBHIP_EWORKSPACE;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  hipStream_t s = as_stream(stream);

  // granules + status word must be zero before every launch (state 0 = not yet published)
  hipError_t e = hipMemsetAsync(workspace, 0, need, s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (n == 0) {
    e = hipMemsetAsync(out_size, 0, sizeof(uint64_t), s);
    return static_cast<int>(e);
  }
  const size_t tiles = (n + kScanTile - 1) / kScanTile;
  // persistent grid: never larger than what is co-resident (look-back waits on lower tiles)
  const bool aligned = (reinterpret_cast<uintptr_t>(src) & 15u) == 0;
  const int per_cu =
      !aligned ? resident_blocks_per_cu(copy_if_lt_kernel<false, false>, kScanThreads, 0, scan_blocks_per_cu())
      : scan_nontemporal()
          ? resident_blocks_per_cu(copy_if_lt_kernel<true, true>, kScanThreads, 0, scan_blocks_per_cu())
          : resident_blocks_per_cu(copy_if_lt_kernel<true, false>, kScanThreads, 0, scan_blocks_per_cu());
  const size_t cap = static_cast<size_t>(dev.cus) * per_cu;
  const unsigned grid = static_cast<unsigned>(tiles < cap ? tiles : cap);
  ScanWs *ws = static_cast<ScanWs *>(workspace);
  unsigned long long *osz = reinterpret_cast<unsigned long long *>(out_size);
  if (!aligned)
    hipLaunchKernelGGL((copy_if_lt_kernel<false, false>), dim3(grid), dim3(kScanThreads), 0, s, src,
                       n, filter_value, out, osz, ws, tiles);
  else if (scan_nontemporal())
    hipLaunchKernelGGL((copy_if_lt_kernel<true, true>), dim3(grid), dim3(kScanThreads), 0, s, src,
                       n, filter_value, out, osz, ws, tiles);
  else
    hipLaunchKernelGGL((copy_if_lt_kernel<true, false>), dim3(grid), dim3(kScanThreads), 0, s, src,
                       n, filter_value, out, osz, ws, tiles);
  return launch_status();
}
