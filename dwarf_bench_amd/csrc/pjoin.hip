// pjoin.hip — device pieces of the radix-partitioned multi-GPU hash join (no reference counterpart:
// the reference has no multi-device code; SURVEY 8(e) defines the path).
//
// Per GPU (one process per GPU, exchange done by the host side with RCCL all-to-all over xGMI):
//   dbhip_pjoin_partition_u32   splits a local column shard into `parts` destination buckets by
//                               fmix32(key * 0x9E3779B1 + c) (Murmur3 finaliser of an affine image of the key,
//                               multiply-shift range reduction, so any part count balances) — a hash
//                               independent of the one the local join partitions by (jl_rank_of, join_lds.hip)
//                               — and tags every key with its GLOBAL row id.  It is the level-0 partition of the LDS join
//                               (join_lds.hip: per-group histogram, bucket/group cursors, LDS-staged
//                               scatter that writes runs) with bucket = destination rank.  Order
//                               inside a bucket is not defined (the join does not need it).
//   dbhip_gather_u32            out[i] = table[idx[i]].
// The local join on the received (key, row id) pairs is dbhip_join_build_pairs_u32 (join.hip), whose id
// buffer then holds GLOBAL row ids directly.
// HBM bytes per partitioned row: 4 (histogram read) + 4 (scatter read) + 8 (key + row id written).
#include "dbhip_common.hpp"
#include "join_common.hpp"

namespace dbhip {
namespace {

constexpr int kPjThreads = 256;
constexpr unsigned kPjMaxParts = 1024;

__global__ __launch_bounds__(kPjThreads) void gather_u32_kernel(const unsigned *__restrict__ table,
                                                                const unsigned *__restrict__ idx, size_t n,
                                                                unsigned *__restrict__ out) {
  const size_t stride = static_cast<size_t>(gridDim.x) * kPjThreads;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kPjThreads + threadIdx.x; i < n; i += stride)
    out[i] = table[idx[i]];
}

// (position, count) columns -> the reference's JoinOneToMany records {pointer into ids, size}
__global__ __launch_bounds__(kPjThreads) void join_answers_kernel(const unsigned *__restrict__ ids,
                                                                  const unsigned *__restrict__ pos,
                                                                  const unsigned *__restrict__ cnt, size_t n,
                                                                  dbhip_join_one_to_many *__restrict__ answers) {
  const size_t stride = static_cast<size_t>(gridDim.x) * kPjThreads;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kPjThreads + threadIdx.x; i < n; i += stride)
    answers[i] = dbhip_join_one_to_many{ids + pos[i], cnt[i]};
}

}  // namespace
}  // namespace dbhip

using namespace dbhip;

extern "C" int dbhip_join_answers_u32(const uint32_t *ids, const uint32_t *out_pos, const uint32_t *out_count,
                                      size_t n_probe, dbhip_join_one_to_many *answers, dbhip_stream_t stream) {
  if (n_probe == 0) return DBHIP_OK;
  if (!out_pos || !out_count || !answers) return DBHIP_EINVAL;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  const size_t want = (n_probe + kPjThreads - 1) / kPjThreads;
  const size_t cap = static_cast<size_t>(dev.cus) * 8;
  hipLaunchKernelGGL(join_answers_kernel, dim3(static_cast<unsigned>(want < cap ? want : cap)), dim3(kPjThreads), 0,
                     as_stream(stream), ids, out_pos, out_count, n_probe, answers);
  return launch_status();
}

extern "C" size_t dbhip_pjoin_partition_workspace_bytes(size_t n, uint32_t parts) {
  (void)n;
  return parts >= 1 && parts <= kPjMaxParts ? jl_partition_workspace_bytes(parts) : 0;
}

extern "C" int dbhip_pjoin_partition_u32(const uint32_t *keys, size_t n, uint64_t first_row_id, uint32_t parts,
                                         uint32_t *out_keys, uint32_t *out_row_ids, uint64_t *out_counts,
                                         void *workspace, size_t workspace_bytes, dbhip_stream_t stream) {
  if (parts < 1 || parts > kPjMaxParts || !out_counts) return DBHIP_EINVAL;
  if (n && (!keys || !out_keys || !out_row_ids)) return DBHIP_EINVAL;
  if (first_row_id + n > 0xFFFFFFFFull) return DBHIP_EINVAL;  // global row ids are 32-bit
  if (!ws_ok(workspace, workspace_bytes, jl_partition_workspace_bytes(parts))) return DBHIP_EWORKSPACE;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  return jl_partition(keys, n, first_row_id, parts, out_keys, out_row_ids,
                      reinterpret_cast<unsigned long long *>(out_counts), workspace, as_stream(stream), dev);
}

extern "C" int dbhip_check_pjoin_route_u32(const uint32_t *keys, size_t n, uint32_t parts, uint32_t rank, uint64_t *result,
                                           dbhip_stream_t stream) {
  if (!result || (n && !keys) || parts < 1 || parts > kPjMaxParts || rank >= parts) return DBHIP_EINVAL;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  return jl_route_check(keys, n, parts, rank, reinterpret_cast<unsigned long long *>(result), as_stream(stream), dev);
}

extern "C" int dbhip_gather_u32(const uint32_t *table, const uint32_t *idx, size_t n, uint32_t *out,
                                dbhip_stream_t stream) {
  if (n == 0) return DBHIP_OK;
  if (!table || !idx || !out) return DBHIP_EINVAL;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  const size_t want = (n + kPjThreads - 1) / kPjThreads;
  const size_t cap = static_cast<size_t>(dev.cus) * 8;
  hipLaunchKernelGGL(gather_u32_kernel, dim3(static_cast<unsigned>(want < cap ? want : cap)), dim3(kPjThreads), 0,
                     as_stream(stream), table, idx, n, out);
  return launch_status();
}
