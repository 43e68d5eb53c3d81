// pjoin.hip — device pieces of the radix-partitioned multi-GPU hash join (no reference counterpart:
// the reference has no multi-device code; SURVEY 8(e) defines the path).
//
// Per GPU (one process per GPU, exchange done by the host side with RCCL all-to-all over xGMI):
//   dbhip_pjoin_partition_u32   splits a local column shard into `parts` destination buckets by the
//                               high bits of the mixed hash (Murmur3 finaliser, multiply-shift range
//                               reduction, so any part count balances) and tags every key with its
//                               GLOBAL row id.  Three launches: bucket histogram (LDS histogram per
//                               workgroup, one atomic per bucket per workgroup), bucket offsets, scatter
//                               (per 4096-key tile: LDS counts -> one global reservation per bucket ->
//                               LDS-ranked writes into the reserved run).  Order inside a bucket is not
//                               defined (the join does not need it).
//   dbhip_gather_u32            out[i] = table[idx[i]]: turns the local join's build-row indices into
//                               global row ids.
// The local join on the received (key, row id) pairs is dwarf 4a (join.hip).
// HBM bytes per partitioned row: 4 (histogram read) + 4 (scatter read) + 8 (key + row id written).
#include "dbhip_common.hpp"

namespace dbhip {
namespace {

constexpr int kPjThreads = 256;
constexpr int kPjKpt = 16;
constexpr int kPjTile = kPjThreads * kPjKpt;  // 4096 keys
constexpr unsigned kPjMaxParts = 256;

struct PjHeader {
  unsigned status;
  unsigned pad0;
  unsigned long long counts[kPjMaxParts];   // rows per bucket
  unsigned long long cursors[kPjMaxParts];  // next free slot of every bucket during the scatter
};
constexpr size_t kPjWsBytes = (sizeof(PjHeader) + kWsAlign - 1) / kWsAlign * kWsAlign;

__host__ __device__ __forceinline__ unsigned pj_dest(unsigned key, unsigned parts) {
  return static_cast<unsigned>((static_cast<unsigned long long>(fmix32(key)) * parts) >> 32);
}

__global__ __launch_bounds__(kPjThreads) void pj_histogram_kernel(const unsigned *__restrict__ keys, size_t n,
                                                                  unsigned parts, PjHeader *hdr) {
  __shared__ unsigned s_hist[kPjMaxParts];
  for (unsigned i = threadIdx.x; i < parts; i += kPjThreads) s_hist[i] = 0;
  __syncthreads();
  const size_t stride = static_cast<size_t>(gridDim.x) * kPjThreads;
  const size_t n4 = n / 4;
  const u32x4 *k4 = reinterpret_cast<const u32x4 *>(keys);
  for (size_t i = static_cast<size_t>(blockIdx.x) * kPjThreads + threadIdx.x; i < n4; i += stride) {
    const u32x4 v = k4[i];
    atomicAdd(&s_hist[pj_dest(v.x, parts)], 1u);
    atomicAdd(&s_hist[pj_dest(v.y, parts)], 1u);
    atomicAdd(&s_hist[pj_dest(v.z, parts)], 1u);
    atomicAdd(&s_hist[pj_dest(v.w, parts)], 1u);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) atomicAdd(&s_hist[pj_dest(keys[n4 * 4 + threadIdx.x], parts)], 1u);
  __syncthreads();
  for (unsigned i = threadIdx.x; i < parts; i += kPjThreads)
    if (s_hist[i]) atomicAdd(&hdr->counts[i], static_cast<unsigned long long>(s_hist[i]));
}

__global__ void pj_offsets_kernel(unsigned parts, PjHeader *hdr, unsigned long long *out_counts) {
  if (threadIdx.x == 0) {
    unsigned long long run = 0;
    for (unsigned d = 0; d < parts; ++d) {
      hdr->cursors[d] = run;
      out_counts[d] = hdr->counts[d];
      run += hdr->counts[d];
    }
  }
}

__global__ __launch_bounds__(kPjThreads) void pj_scatter_kernel(const unsigned *__restrict__ keys, size_t n,
                                                                unsigned long long first_row, unsigned parts,
                                                                unsigned *__restrict__ out_keys,
                                                                unsigned *__restrict__ out_rids,
                                                                PjHeader *hdr, size_t num_tiles) {
  __shared__ unsigned s_cnt[kPjMaxParts];
  __shared__ unsigned long long s_base[kPjMaxParts];
  for (size_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
    const size_t base = tile * kPjTile;
    for (unsigned i = threadIdx.x; i < parts; i += kPjThreads) s_cnt[i] = 0;
    __syncthreads();
    unsigned key[kPjKpt], rank[kPjKpt], dest[kPjKpt];
#pragma unroll
    for (int j = 0; j < kPjKpt; ++j) {
      const size_t idx = base + static_cast<size_t>(j) * kPjThreads + threadIdx.x;
      const bool valid = idx < n;
      key[j] = valid ? keys[idx] : 0u;
      dest[j] = valid ? pj_dest(key[j], parts) : parts;  // parts = "no bucket"
      rank[j] = valid ? atomicAdd(&s_cnt[dest[j]], 1u) : 0u;  // LDS atomic with return: rank in the tile
    }
    __syncthreads();
    for (unsigned i = threadIdx.x; i < parts; i += kPjThreads)  // one reservation per bucket per tile
      s_base[i] = s_cnt[i] ? atomicAdd(&hdr->cursors[i], static_cast<unsigned long long>(s_cnt[i])) : 0ull;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kPjKpt; ++j) {
      if (dest[j] < parts) {
        const size_t slot = s_base[dest[j]] + rank[j];
        const size_t idx = base + static_cast<size_t>(j) * kPjThreads + threadIdx.x;
        out_keys[slot] = key[j];
        out_rids[slot] = static_cast<unsigned>(first_row + idx);
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(kPjThreads) void gather_u32_kernel(const unsigned *__restrict__ table,
                                                                const unsigned *__restrict__ idx, size_t n,
                                                                unsigned *__restrict__ out) {
  const size_t stride = static_cast<size_t>(gridDim.x) * kPjThreads;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kPjThreads + threadIdx.x; i < n; i += stride)
    out[i] = table[idx[i]];
}

inline unsigned pj_grid(size_t work_items, const DeviceInfo &dev, int per_cu) {
  const size_t want = (work_items + kPjThreads - 1) / kPjThreads;
  const size_t cap = static_cast<size_t>(dev.cus) * per_cu;
  return static_cast<unsigned>(want < cap ? (want ? want : 1) : cap);
}

}  // namespace
}  // namespace dbhip

using namespace dbhip;

extern "C" size_t dbhip_pjoin_partition_workspace_bytes(size_t n, uint32_t parts) {
  (void)n;
  return parts >= 1 && parts <= kPjMaxParts ? kPjWsBytes : 0;
}

extern "C" int dbhip_pjoin_partition_u32(const uint32_t *keys, size_t n, uint64_t first_row_id, uint32_t parts,
                                         uint32_t *out_keys, uint32_t *out_row_ids, uint64_t *out_counts,
                                         void *workspace, size_t workspace_bytes, dbhip_stream_t stream) {
  if (parts < 1 || parts > kPjMaxParts || !out_counts) return DBHIP_EINVAL;
  if (n && (!keys || !out_keys || !out_row_ids)) return DBHIP_EINVAL;
  if (first_row_id + n > 0xFFFFFFFFull) return DBHIP_EINVAL;  // global row ids are 32-bit
  if (reinterpret_cast<uintptr_t>(keys) & 15u) return DBHIP_EINVAL;
  if (!ws_ok(workspace, workspace_bytes, kPjWsBytes)) return DBHIP_EWORKSPACE;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  hipStream_t s = as_stream(stream);
  hipError_t e = hipMemsetAsync(workspace, 0, kPjWsBytes, s);
  if (e != hipSuccess) return static_cast<int>(e);
  PjHeader *hdr = static_cast<PjHeader *>(workspace);
  if (n)
    hipLaunchKernelGGL(pj_histogram_kernel, dim3(pj_grid(n / 4 + 1, dev, 8)), dim3(kPjThreads), 0, s, keys, n, parts,
                       hdr);
  hipLaunchKernelGGL(pj_offsets_kernel, dim3(1), dim3(64), 0, s, parts, hdr,
                     reinterpret_cast<unsigned long long *>(out_counts));
  if (n) {
    const size_t tiles = (n + kPjTile - 1) / kPjTile;
    const size_t cap = static_cast<size_t>(dev.cus) * 8;
    hipLaunchKernelGGL(pj_scatter_kernel, dim3(static_cast<unsigned>(tiles < cap ? tiles : cap)), dim3(kPjThreads), 0,
                       s, keys, n, first_row_id, parts, out_keys, out_row_ids, hdr, tiles);
  }
  return launch_status();
}

extern "C" int dbhip_gather_u32(const uint32_t *table, const uint32_t *idx, size_t n, uint32_t *out,
                                dbhip_stream_t stream) {
  if (n == 0) return DBHIP_OK;
  if (!table || !idx || !out) return DBHIP_EINVAL;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  hipLaunchKernelGGL(gather_u32_kernel, dim3(pj_grid(n, dev, 8)), dim3(kPjThreads), 0, as_stream(stream), table, idx,
                     n, out);
  return launch_status();
}
