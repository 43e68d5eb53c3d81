// scan.hip — dwarf 1: stable stream compaction out = [x in src : x < filter] for gfx950.
//
// Replaces the reference kernel simple_two_pass_scan (scan/scan.cl:3-42: chunked count -> serial
// prefix by work-item 0 -> chunked write, which reads src twice with uncoalesced per-lane chunks)
// and the oneDPL copy_if behind DPLScan (dpl_wrapper.hpp:27-33).  Same logical result
// (scan/scan.cpp:12-17 expected_out_lt), different machine mapping:
//
//   * ONE pass over src.  Every wave reads 2048 contiguous elements of a tile with eight
//     16-byte-per-lane non-temporal loads (each wave instruction = 1 KiB contiguous), keeps them in
//     registers, counts, and writes its matches at an offset inside the chunk's staging slot.
//   * In-wave ranks come from the compare masks themselves: v_cmp -> 64-bit ballot in SGPRs,
//     s_bcnt1 for totals, v_mbcnt_lo/hi for the lane-exclusive prefix.
//   * No workgroup ever waits on another one (see below): no tickets, no look-back, no spin, nothing to time
//     out; every call is two launches whatever the size.
//
// Algorithmic HBM bytes: 4*n read + 4*out_size written.
#include <climits>

#include "dbhip_common.hpp"

namespace dbhip {
namespace {

constexpr int kScanVpt = 8;                           // 16-byte loads per lane per tile
constexpr int kScanWaveElems = kWave * kScanVpt * 4;  // 2048 contiguous elements per wave

struct ScanWs {
  unsigned status;  // DBHIP_DEV_* bits (this dwarf has no device-side failure: always DBHIP_DEV_OK after a call)
  unsigned pad[63];
};
static_assert(sizeof(ScanWs) == kWsHeader, "workspace header size");

template <int WAVES, bool kAligned, bool kNontemporal>
__device__ __forceinline__ void load_wave_tile(i32x4 (&v)[kScanVpt], const int *__restrict__ src,
                                               size_t n, size_t tile, unsigned wave, unsigned lane) {
  constexpr size_t kTile = static_cast<size_t>(kScanWaveElems) * WAVES;
  const size_t wave_base = tile * kTile + static_cast<size_t>(wave) * kScanWaveElems;
  if (kAligned && tile * kTile + kTile <= n) {
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
}

// matches of one wave: the compare IS the ballot (v_cmp -> SGPR pair), totals are scalar popcounts
__device__ __forceinline__ unsigned count_wave_matches(const i32x4 (&v)[kScanVpt], int filter) {
  unsigned total = 0;
#pragma unroll
  for (int k = 0; k < kScanVpt; ++k) {
    total += __builtin_popcountll(__ballot(v[k].x < filter));
    total += __builtin_popcountll(__ballot(v[k].y < filter));
    total += __builtin_popcountll(__ballot(v[k].z < filter));
    total += __builtin_popcountll(__ballot(v[k].w < filter));
  }
  return total;
}

// writes the wave's matches, in source order, to dst_wave[0 .. wave_total)
// (dst_wave may point to global memory or to the LDS staging buffer)
__device__ __forceinline__ void emit_wave_matches(const i32x4 (&v)[kScanVpt], int filter,
                                                  int *dst_wave) {
  unsigned row_base = 0;
#pragma unroll
  for (int k = 0; k < kScanVpt; ++k) {
    const bool m0 = v[k].x < filter, m1 = v[k].y < filter, m2 = v[k].z < filter, m3 = v[k].w < filter;
    const unsigned long long b0 = __ballot(m0), b1 = __ballot(m1), b2 = __ballot(m2), b3 = __ballot(m3);
    if ((b0 | b1 | b2 | b3) == 0) continue;  // scalar branch: most rows are empty at low selectivity
    int *dst = dst_wave + row_base + mbcnt(b0) + mbcnt(b1) + mbcnt(b2) + mbcnt(b3);
    if (m0) *dst++ = v[k].x;
    if (m1) *dst++ = v[k].y;
    if (m2) *dst++ = v[k].z;
    if (m3) *dst++ = v[k].w;
    row_base += __builtin_popcountll(b0) + __builtin_popcountll(b1) + __builtin_popcountll(b2) +
                __builtin_popcountll(b3);
  }
}

// =================================================================================================
// Chunked path, two kernels, NO communication between workgroups while streaming.
//
// Why not a single-pass decoupled look-back (measured history in DESIGN.md 4.1): with all 256 CUs streaming,
// ~30 MiB of loads are in flight and any read that must come from memory — a look-back poll included — takes
// 3.5-4.5 us, as long as a CU needs for a whole 128 KiB tile; every variant built landed between 255 and 340 us
// at 2^28 against 160 us for the bare stream.  So the dependency is removed instead of hidden:
//   scan_chunk_kernel  chunks (1 MiB for large inputs) are dealt by blockIdx — nothing here waits on another
//                      workgroup, so no ticket and no zeroed header are needed: the call is two launches.
//                      A 16-wave workgroup streams its chunk tile by tile (128 KiB, double-buffered registers,
//                      next tile's loads issued right after the count, a barrier that does not drain vmcnt),
//                      ranks matches with ballots/mbcnt and writes them to the chunk's own slot of a staging
//                      buffer at chunk-local offsets; one count per chunk.
//   scan_move_kernel   one workgroup per chunk: prefix of the chunk counts (<= 4 KiB, read from L2),
//                      then a coalesced copy staging -> out at the global offset; writes out_size and the
//                      (always clean) status word.
// HBM bytes: 4n + 4*matches (stream) + 8*matches (move); at the reference's selectivity (4e-4) the
// move is ~0.1 % of the traffic.  Needs an n-element staging buffer in the workspace.
// =================================================================================================
constexpr int kChWaves = 16;
constexpr int kChThreads = kChWaves * kWave;
constexpr int kChTile = kChWaves * kScanWaveElems;  // 32768 elements = 128 KiB
constexpr int kChMaxTilesPerChunk = 8;  // 1 MiB chunks for large inputs; fewer tiles per chunk when
                                        // the input would otherwise give less than ~4 chunks per CU

struct ChunkLayout {
  size_t chunks, chunk_elems, counts_off, staging_off, total;
  unsigned tiles_per_chunk;
};
inline ChunkLayout chunk_layout(size_t n) {
  ChunkLayout L;
  const size_t tiles = (n + kChTile - 1) / kChTile;
  size_t tpc = tiles / 1024;  // aim at >= 1024 chunks (4 per CU on 256 CUs)
  tpc = tpc < 1 ? 1 : (tpc > kChMaxTilesPerChunk ? kChMaxTilesPerChunk : tpc);
  L.tiles_per_chunk = static_cast<unsigned>(tpc);
  L.chunk_elems = tpc * kChTile;
  L.chunks = (n + L.chunk_elems - 1) / L.chunk_elems;
  L.counts_off = kWsHeader;
  L.staging_off = align_up(L.counts_off + (L.chunks ? L.chunks : 1) * sizeof(unsigned), kWsAlign);
  L.total = align_up(L.staging_off + n * sizeof(int), kWsAlign);
  return L;
}

// one tile of the chunk: count `cur`, prefetch the next tile into `nxt`, emit matches
template <bool kAligned, bool kNontemporal>
__device__ __forceinline__ unsigned chunk_step(i32x4 (&cur)[kScanVpt], i32x4 (&nxt)[kScanVpt],
                                               const int *__restrict__ src, size_t n, int filter,
                                               int *__restrict__ dst_chunk, unsigned running,
                                               size_t tile, bool has_next, unsigned par,
                                               unsigned (*s_wave_total)[kChWaves], unsigned wave,
                                               unsigned lane) {
  const unsigned wave_total = count_wave_matches(cur, filter);
  if (lane == 0) s_wave_total[par][wave] = wave_total;
  // keep the CU streaming: the next tile's loads go out before anything else happens
  if (has_next) load_wave_tile<kChWaves, kAligned, kNontemporal>(nxt, src, n, tile + 1, wave, lane);
  wg_barrier_lds_only();  // NOT __syncthreads(): its vmcnt(0) would wait for the prefetch
  unsigned wave_excl = 0, tile_total = 0;
#pragma unroll
  for (int w = 0; w < kChWaves; ++w) {
    const unsigned t = s_wave_total[par][w];
    wave_excl += w < static_cast<int>(wave) ? t : 0u;
    tile_total += t;
  }
  if (wave_total != 0) emit_wave_matches(cur, filter, dst_chunk + running + wave_excl);
  return running + tile_total;
}

template <bool kAligned, bool kNontemporal>
__global__ __launch_bounds__(kChThreads) void scan_chunk_kernel(const int *__restrict__ src, size_t n,
                                                                int filter, int *__restrict__ staging,
                                                                unsigned *__restrict__ counts, size_t num_chunks,
                                                                unsigned tiles_per_chunk) {
  // two slots alternate with the tile parity: one barrier per tile is enough
  __shared__ unsigned s_wave_total[2][kChWaves];
  const unsigned lane = threadIdx.x & (kWave - 1);
  const unsigned wave = threadIdx.x / kWave;

  // Chunks are dealt by blockIdx, not by ticket: this kernel never waits on another workgroup, so residency does
  // not matter for progress, and without a ticket counter the call needs no zeroed header — one dependent launch
  // (the fill) less in front of the stream.
  for (size_t chunk = blockIdx.x; chunk < num_chunks; chunk += gridDim.x) {
    __syncthreads();  // the previous chunk's last tile may still be reading the s_wave_total slot tile 0 writes
    const size_t chunk_elems = static_cast<size_t>(tiles_per_chunk) * kChTile;
    const size_t first_tile = chunk * tiles_per_chunk;
    const size_t elems = n - chunk * chunk_elems < chunk_elems ? n - chunk * chunk_elems : chunk_elems;
    const unsigned tiles = static_cast<unsigned>((elems + kChTile - 1) / kChTile);
    int *dst_chunk = staging + chunk * chunk_elems;

    i32x4 a[kScanVpt], b[kScanVpt];
    load_wave_tile<kChWaves, kAligned, kNontemporal>(a, src, n, first_tile, wave, lane);
    unsigned running = 0;
    for (unsigned t = 0; t < tiles; t += 2) {
      running = chunk_step<kAligned, kNontemporal>(a, b, src, n, filter, dst_chunk, running, first_tile + t,
                                                   t + 1 < tiles, 0u, s_wave_total, wave, lane);
      if (t + 1 >= tiles) break;
      running = chunk_step<kAligned, kNontemporal>(b, a, src, n, filter, dst_chunk, running,
                                                   first_tile + t + 1, t + 2 < tiles, 1u, s_wave_total,
                                                   wave, lane);
    }
    if (threadIdx.x == 0) counts[chunk] = running;
  }
}

__global__ __launch_bounds__(256) void scan_move_kernel(const int *__restrict__ staging,
                                                        const unsigned *__restrict__ counts,
                                                        int *__restrict__ out,
                                                        unsigned long long *__restrict__ out_size,
                                                        size_t num_chunks, size_t chunk_elems, ScanWs *ws) {
  __shared__ unsigned long long s_part[256 / kWave];
  const size_t chunk = blockIdx.x;
  const unsigned tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  unsigned long long mine = 0;
  for (size_t i = tid; i < chunk; i += 256) mine += counts[i];
  mine = wave_reduce_add_u64(mine);
  if (lane == 0) s_part[wave] = mine;
  __syncthreads();
  unsigned long long prefix = 0;
#pragma unroll
  for (int w = 0; w < 256 / kWave; ++w) prefix += s_part[w];
  const unsigned m = counts[chunk];
  const int *src = staging + chunk * chunk_elems;  // 16-byte aligned: chunk_elems is a multiple of the tile
  int *dst = out + prefix;
  // 16 bytes per lane: the destination decides the alignment (elements up to its next 16-byte boundary go one by
  // one), the source is then read at whatever alignment that leaves (4-byte aligned 16-byte loads are legal on
  // global memory); at dense selectivities this kernel moves as many bytes as the stream itself
  const unsigned head0 = ((16u - (static_cast<unsigned>(reinterpret_cast<uintptr_t>(dst)) & 15u)) & 15u) / 4u;
  const unsigned head = head0 < m ? head0 : m;
  if (tid < head) dst[tid] = src[tid];
  const unsigned body = (m - head) / 4;
  typedef int i32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
  for (unsigned v = tid; v < body; v += 256) {
    const i32x4 x = *reinterpret_cast<const i32x4_a4 *>(src + head + 4 * v);
    __builtin_nontemporal_store(x, reinterpret_cast<i32x4 *>(dst + head + 4 * v));
  }
  const unsigned tail0 = head + 4 * body;
  if (tid < 4 && tail0 + tid < m) dst[tail0 + tid] = src[tail0 + tid];
  if (chunk == num_chunks - 1 && tid == 0) {
    *out_size = prefix + m;
    ws->status = DBHIP_DEV_OK;  // nothing on this path can fail on the device; the header is not cleared up front
  }
}

inline int env_int(const char *name, int lo, int hi, int dflt) {
  const char *e = getenv(name);
  if (!e) return dflt;
  const int x = atoi(e);
  return (x >= lo && x <= hi) ? x : dflt;
}
inline bool scan_nontemporal() {
  static const int v = env_int("DBHIP_SCAN_NT", 0, 1, 1);
  return v != 0;
}
template <bool kAligned, bool kNontemporal>
int launch_chunked(const int *src, size_t n, int filter, int *out, unsigned long long *osz, void *workspace,
                   const DeviceInfo &dev, hipStream_t s) {
  const ChunkLayout L = chunk_layout(n);
  char *base = static_cast<char *>(workspace);
  ScanWs *ws = reinterpret_cast<ScanWs *>(base);
  unsigned *counts = reinterpret_cast<unsigned *>(base + L.counts_off);
  int *staging = reinterpret_cast<int *>(base + L.staging_off);
  const size_t cap = static_cast<size_t>(dev.cus);  // one 16-wave workgroup per CU
  const unsigned grid = static_cast<unsigned>(L.chunks < cap ? L.chunks : cap);
  hipLaunchKernelGGL((scan_chunk_kernel<kAligned, kNontemporal>), dim3(grid), dim3(kChThreads), 0, s, src, n,
                     filter, staging, counts, L.chunks, L.tiles_per_chunk);
  hipLaunchKernelGGL(scan_move_kernel, dim3(static_cast<unsigned>(L.chunks)), dim3(256), 0, s, staging,
                     counts, out, osz, L.chunks, L.chunk_elems, ws);
  return launch_status();
}

}  // namespace
}  // namespace dbhip

using namespace dbhip;

extern "C" size_t dbhip_copy_if_lt_i32_workspace_bytes(size_t n) {
  return chunk_layout(n ? n : 1).total;  // header | counts[chunks] | staging[n]
}

extern "C" int dbhip_copy_if_lt_i32(const int32_t *src, size_t n, int32_t filter_value,
                                    int32_t *out, uint64_t *out_size, void *workspace,
                                    size_t workspace_bytes, dbhip_stream_t stream) {
  if (!out_size || (n && (!src || !out))) return DBHIP_EINVAL;
  if (n >= (1ull << 61)) return DBHIP_EINVAL;
  if (!ws_ok(workspace, workspace_bytes, dbhip_copy_if_lt_i32_workspace_bytes(n)))
    return DBHIP_EWORKSPACE;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  hipStream_t s = as_stream(stream);
  if (n == 0) {
    hipError_t e = fill_async(workspace, 0, kWsHeader, s);
    if (e == hipSuccess) e = fill_async(out_size, 0, sizeof(uint64_t), s);
    return static_cast<int>(e);
  }
  unsigned long long *osz = reinterpret_cast<unsigned long long *>(out_size);
  const bool aligned = (reinterpret_cast<uintptr_t>(src) & 15u) == 0;
  if (!aligned) return launch_chunked<false, false>(src, n, filter_value, out, osz, workspace, dev, s);
  if (scan_nontemporal()) return launch_chunked<true, true>(src, n, filter_value, out, osz, workspace, dev, s);
  return launch_chunked<true, false>(src, n, filter_value, out, osz, workspace, dev, s);
}
