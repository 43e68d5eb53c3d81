// reduce_nlj.hip — the two small dwarfs that complete the reference's taxonomy:
//
//   * sum-reduce of an int32 column (reduce/reduce.cpp:27-88: oneAPI reduction with plus<>, result an
//     int; here int32 wrap-around addition, which is associative, so any summation order is bit-exact);
//   * nested-loop equi-join (join/nested_join.cpp:10-100): the dense |A| x |B| cell matrix the reference
//     fills, cell (i, j) = (key, a_val[i], b_val[j]) when a_key[i] == b_key[j].  Unlike the reference,
//     which pre-fills the matrix on the host (key 0, values 0xFFFFFFFF, :30-32) and re-uploads it every
//     iteration, the kernel writes every cell itself, matches and misses alike, so no pre-fill pass exists
//     and the stores are full coalesced rows.  A small-n device oracle for the hash joins.
//
// Both are HBM-bound: reduce reads 4 B/row once (nontemporal 16-B loads, 8 in flight per lane),
// nested-loop join writes 12 B per cell.
#include <cstdlib>

#include "dbhip_common.hpp"

namespace dbhip {
namespace {

constexpr int kRedThreads = 512;
constexpr int kRedVecsPerLane = 8;  // 8 x 16 B in flight per lane
constexpr size_t kRedTileInts = static_cast<size_t>(kRedThreads) * kRedVecsPerLane * 4;  // 64 KiB tiles
#ifndef DBHIP_RED_CHUNK_TILES
#define DBHIP_RED_CHUNK_TILES 16
#endif
constexpr size_t kRedChunkTiles = DBHIP_RED_CHUNK_TILES;  // 1 MiB of consecutive tiles per workgroup turn

__global__ __launch_bounds__(kRedThreads) void reduce_sum_kernel(const int *__restrict__ src, size_t n, unsigned head,
                                                                 unsigned *__restrict__ out) {
  // `head` (0..3) elements in front of the first 16-byte boundary are added one by one by the first workgroup;
  // the vector loads start behind them
  __shared__ unsigned wave_sums[kRedThreads / kWave];
  unsigned acc = 0;
  if (blockIdx.x == 0 && threadIdx.x < head) acc += static_cast<unsigned>(src[threadIdx.x]);
  src += head;
  n -= head;
  const size_t full_tiles = n / kRedTileInts;
  const i32x4 *vsrc = reinterpret_cast<const i32x4 *>(src);
  // A workgroup takes kRedChunkTiles consecutive tiles at a time (1 MiB of contiguous addresses, like the scan's
  // chunks) and keeps the next tile's loads in flight while it adds the current one.
  auto load_tile = [&](size_t t, i32x4 (&v)[kRedVecsPerLane]) {
    const i32x4 *p = vsrc + t * (kRedTileInts / 4) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < kRedVecsPerLane; ++k) v[k] = __builtin_nontemporal_load(p + k * kRedThreads);
  };
  auto add_tile = [&](const i32x4 (&v)[kRedVecsPerLane]) {
#pragma unroll
    for (int k = 0; k < kRedVecsPerLane; ++k)
      acc += static_cast<unsigned>(v[k].x) + static_cast<unsigned>(v[k].y) + static_cast<unsigned>(v[k].z) +
             static_cast<unsigned>(v[k].w);
  };
  const size_t chunks = (full_tiles + kRedChunkTiles - 1) / kRedChunkTiles;
  for (size_t c = blockIdx.x; c < chunks; c += gridDim.x) {
    const size_t t0 = c * kRedChunkTiles;
    const size_t t1 = t0 + kRedChunkTiles < full_tiles ? t0 + kRedChunkTiles : full_tiles;
    i32x4 a[kRedVecsPerLane], b[kRedVecsPerLane];
    load_tile(t0, a);
    for (size_t t = t0; t < t1; t += 2) {
      if (t + 1 < t1) load_tile(t + 1, b);
      add_tile(a);
      if (t + 1 >= t1) break;
      if (t + 2 < t1) load_tile(t + 2, a);
      add_tile(b);
    }
  }
  // ragged tail (< one tile), shared by the grid
  const size_t tail0 = full_tiles * kRedTileInts;
  for (size_t i = tail0 + static_cast<size_t>(blockIdx.x) * kRedThreads + threadIdx.x; i < n;
       i += static_cast<size_t>(gridDim.x) * kRedThreads)
    acc += static_cast<unsigned>(src[i]);

  acc = wave_reduce_add(acc);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if (lane == kWave - 1) wave_sums[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned s = 0;
#pragma unroll
    for (int w = 0; w < kRedThreads / kWave; ++w) s += wave_sums[w];
    if (s) atomicAdd(out, s);
  }
}

// one workgroup = 256 B columns x kNljRows A rows; B keys/values live in registers, A rows are wave-uniform
constexpr int kNljThreads = 256;
constexpr int kNljRows = 16;
constexpr unsigned kNljMiss = 0xFFFFFFFFu;

__global__ __launch_bounds__(kNljThreads) void nested_join_kernel(
    const unsigned *__restrict__ a_keys, const unsigned *__restrict__ a_vals,
    const unsigned *__restrict__ b_keys, const unsigned *__restrict__ b_vals, size_t n_a, size_t n_b,
    unsigned *__restrict__ out_key, unsigned *__restrict__ out_val1, unsigned *__restrict__ out_val2) {
  const size_t j = static_cast<size_t>(blockIdx.x) * kNljThreads + threadIdx.x;
  const size_t i0 = static_cast<size_t>(blockIdx.y) * kNljRows;
  if (j >= n_b) return;
  const unsigned bk = b_keys[j], bv = b_vals[j];
  const size_t i1 = i0 + kNljRows < n_a ? i0 + kNljRows : n_a;
  for (size_t i = i0; i < i1; ++i) {
    const unsigned ak = a_keys[i], av = a_vals[i];  // uniform across the workgroup: scalar loads
    const bool hit = ak == bk;
    const size_t cell = i * n_b + j;
    out_key[cell] = hit ? ak : 0u;  // the reference's "no row here" markers (nested_join.cpp:30-32, :85)
    out_val1[cell] = hit ? av : kNljMiss;
    out_val2[cell] = hit ? bv : kNljMiss;
  }
}

}  // namespace
}  // namespace dbhip

using namespace dbhip;

extern "C" int dbhip_reduce_sum_i32(const int32_t *src, size_t n, int32_t *out, dbhip_stream_t stream) {
  if (!out || (n && !src)) return DBHIP_EINVAL;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  hipStream_t s = as_stream(stream);
  hipError_t e = fill_async(out, 0, sizeof(int32_t), s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (n == 0) return DBHIP_OK;
  const size_t tiles = (n + kRedTileInts - 1) / kRedTileInts;
  const size_t chunks = (tiles + kRedChunkTiles - 1) / kRedChunkTiles;
  static const int wgs_per_cu = [] {
    const char *e = std::getenv("DBHIP_RED_WGS");  // experiment knob: workgroups per CU of the persistent grid
    const int v = e ? std::atoi(e) : 0;
    return v >= 1 && v <= 16 ? v : 2;
  }();
  const size_t cap = static_cast<size_t>(dev.cus) * wgs_per_cu;
  const unsigned grid = static_cast<unsigned>(chunks < cap ? (chunks ? chunks : 1) : cap);
  size_t head = ((16 - (reinterpret_cast<uintptr_t>(src) & 15u)) & 15u) / 4;  // src is 4-byte aligned (int32)
  if (head > n) head = n;
  hipLaunchKernelGGL(reduce_sum_kernel, dim3(grid), dim3(kRedThreads), 0, s, src, n, static_cast<unsigned>(head),
                     reinterpret_cast<unsigned *>(out));
  return launch_status();
}

extern "C" int dbhip_nested_join_u32(const uint32_t *a_keys, const uint32_t *a_vals, const uint32_t *b_keys,
                                     const uint32_t *b_vals, size_t n_a, size_t n_b, uint32_t *out_key,
                                     uint32_t *out_val1, uint32_t *out_val2, dbhip_stream_t stream) {
  if (n_a == 0 || n_b == 0) return DBHIP_OK;
  if (!a_keys || !a_vals || !b_keys || !b_vals || !out_key || !out_val1 || !out_val2) return DBHIP_EINVAL;
  const size_t row_groups = (n_a + kNljRows - 1) / kNljRows;
  const size_t col_groups = (n_b + kNljThreads - 1) / kNljThreads;
  if (row_groups > 65535 || col_groups > 0x7FFFFFFFull) return DBHIP_EINVAL;  // n_a <= 1,048,560 rows
  hipLaunchKernelGGL(nested_join_kernel, dim3(static_cast<unsigned>(col_groups), static_cast<unsigned>(row_groups)),
                     dim3(kNljThreads), 0, as_stream(stream), a_keys, a_vals, b_keys, b_vals, n_a, n_b, out_key,
                     out_val1, out_val2);
  return launch_status();
}
