// check.hip — device-side validators: what the `...Hip` dwarfs use to fill Result::valid at sizes where the
// reference's host-side checks (std::copy_if / std::sort / expected_GroupBy / the brute-force join oracle,
// scan/scan.cpp:157-164, sort/radix.cpp:46-52, groupby/groupby.cpp:95-103, join/join_omnisci.cpp:31-45) would
// dominate the run or not fit the host.  Every validator is an algorithm INDEPENDENT of the kernel it checks
// (no shared code with scan.hip / radix.hip / groupby.hip / join_lds.hip beyond the wave helpers):
//
//   dbhip_check_fingerprint_lt_i32   order-sensitive fingerprint + length of the subsequence x < filter:
//                                    H = sum (x_i + 1) * P^(L-1-i) mod 2^64, computed as an ordered monoid reduction
//                                    ((H1,L1)o(H2,L2) = (H1*P^L2 + H2, L1+L2)).  copy_if is right iff the fingerprint
//                                    of src under the filter equals the fingerprint of out[0..out_size).
//   dbhip_check_sorted_u32           number of descents key[i] > key[i+1] + a commutative multiset fingerprint
//                                    (sum of mix64(key), sum of keys): sorted and a permutation of the input.
//   dbhip_check_weighted_sum_u32     sum vals[i] * w(keys[i]) mod 2^32 for two weight functions: a group-by result
//                                    is right iff the weighted sum over (g, out[g]) equals the one over the rows.
//   dbhip_check_permutation_u32      ids[] is a permutation of 0..n-1 (bitmap + atomicOr).
//   dbhip_check_join_u32             per probe row: count == (upper - lower bound of the key in the SORTED build
//                                    column), the id range is inside the id buffer and its first / last / one
//                                    pseudo-random id carry the key.
//   dbhip_check_ujoin_u32            per probe row of the unique-key join: (key, build payload, probe payload) or
//                                    the three sentinels, the build side found by binary search.
//   dbhip_check_gen_uniform_u32      values[i] == lo + mix64(seed, index_i) % span: a column (or a received
//                                    (key, row id) pair of the partitioned join) is what the generator produced.
// All results are uint64 words in DEVICE memory, zeroed by the call itself.
#include "dbhip_common.hpp"

namespace dbhip {
namespace {

constexpr int kCkThreads = 256;
constexpr unsigned long long kFpMul = 0x9E3779B97F4A7C15ull;  // odd: invertible mod 2^64

inline unsigned ck_grid(size_t n, const DeviceInfo &dev, int per_cu = 8) {
  const size_t want = (n + kCkThreads - 1) / kCkThreads;
  const size_t cap = static_cast<size_t>(dev.cus) * per_cu;
  return static_cast<unsigned>(want < cap ? (want ? want : 1) : cap);
}

__device__ __forceinline__ unsigned long long fp_pow(unsigned long long e) {
  unsigned long long r = 1, b = kFpMul;
  while (e) {
    if (e & 1ull) r *= b;
    b *= b;
    e >>= 1;
  }
  return r;
}
struct Fp {
  unsigned long long h, len;
};
__device__ __forceinline__ Fp fp_combine(Fp left, Fp right) {
  return Fp{left.h * fp_pow(right.len) + right.h, left.len + right.len};
}

// block-level sum of a u64 into one atomic per block
__device__ __forceinline__ void block_add_u64(unsigned long long v, unsigned long long *dst) {
  __shared__ unsigned long long s_part[kCkThreads / kWave];
  v = wave_reduce_add_u64(v);
  if ((threadIdx.x & (kWave - 1)) == 0) s_part[threadIdx.x / kWave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long s = 0;
#pragma unroll
    for (int w = 0; w < kCkThreads / kWave; ++w) s += s_part[w];
    if (s) atomicAdd(dst, s);
  }
  __syncthreads();
}

// ---- ordered fingerprint -----------------------------------------------------------------------------------------
// Every thread walks its own CONTIGUOUS segment (order matters), lanes / waves / blocks are then combined left to
// right.  Lanes of a wave read 64 different lines per step, each line consumed over the next steps out of L1: slow
// next to a coalesced stream, which is fine for a check that runs once per measured size.
__global__ __launch_bounds__(kCkThreads) void fp_partial_kernel(const int *__restrict__ src, size_t n, int filter,
                                                                size_t seg, Fp *__restrict__ partial) {
  __shared__ Fp s_w[kCkThreads / kWave];
  const size_t t = static_cast<size_t>(blockIdx.x) * kCkThreads + threadIdx.x;
  size_t lo = t * seg, hi = lo + seg;
  lo = lo < n ? lo : n;
  hi = hi < n ? hi : n;
  Fp f{0ull, 0ull};
  for (size_t i = lo; i < hi; ++i) {
    const int x = src[i];
    if (x < filter) {
      f.h = f.h * kFpMul + (static_cast<unsigned long long>(static_cast<unsigned>(x)) + 1ull);
      ++f.len;
    }
  }
  // ordered tree over the 64 lanes: lane l (l % 2s == 0) absorbs lane l + s on its right
  const unsigned lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
  for (int s = 1; s < kWave; s <<= 1) {
    Fp r;
    r.h = __shfl_down(f.h, s, kWave);
    r.len = __shfl_down(f.len, s, kWave);
    if ((lane & (2 * s - 1)) == 0) f = fp_combine(f, r);
  }
  if (lane == 0) s_w[wave] = f;
  __syncthreads();
  if (threadIdx.x == 0) {
    Fp acc = s_w[0];
    for (int w = 1; w < kCkThreads / kWave; ++w) acc = fp_combine(acc, s_w[w]);
    partial[blockIdx.x] = acc;
  }
}

__global__ __launch_bounds__(kWave) void fp_final_kernel(const Fp *__restrict__ partial, unsigned blocks,
                                                         unsigned long long *__restrict__ result) {
  const unsigned lane = threadIdx.x;
  const unsigned per = (blocks + kWave - 1) / kWave;
  Fp f{0ull, 0ull};
  for (unsigned j = 0; j < per; ++j) {
    const unsigned b = lane * per + j;
    if (b < blocks) f = fp_combine(f, partial[b]);
  }
#pragma unroll
  for (int s = 1; s < kWave; s <<= 1) {
    Fp r;
    r.h = __shfl_down(f.h, s, kWave);
    r.len = __shfl_down(f.len, s, kWave);
    if ((lane & (2 * s - 1)) == 0) f = fp_combine(f, r);
  }
  if (lane == 0) {
    result[0] = f.h;
    result[1] = f.len;
  }
}

// ---- sortedness + multiset fingerprint -----------------------------------------------------------------------------
__global__ __launch_bounds__(kCkThreads) void sorted_kernel(const unsigned *__restrict__ keys, size_t n,
                                                            unsigned xor_mask, unsigned long long *result) {
  const size_t stride = static_cast<size_t>(gridDim.x) * kCkThreads;
  unsigned long long descents = 0, hsum = 0, ksum = 0;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kCkThreads + threadIdx.x; i < n; i += stride) {
    const unsigned k = keys[i];
    if (i + 1 < n && (k ^ xor_mask) > (keys[i + 1] ^ xor_mask)) ++descents;
    hsum += mix64(0x5bd1e995ull, k);
    ksum += k;
  }
  block_add_u64(descents, result + 0);
  block_add_u64(hsum, result + 1);
  block_add_u64(ksum, result + 2);
}

// ---- weighted sum (group-by) ---------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned wt0(unsigned k) { return fmix32(k) | 1u; }
__device__ __forceinline__ unsigned wt1(unsigned k) { return fmix32(k ^ 0x9E3779B9u) | 1u; }

__global__ __launch_bounds__(kCkThreads) void weighted_sum_kernel(const unsigned *__restrict__ keys,
                                                                  const unsigned *__restrict__ vals, size_t n,
                                                                  unsigned long long *result) {
  const size_t stride = static_cast<size_t>(gridDim.x) * kCkThreads;
  unsigned a0 = 0, a1 = 0;  // arithmetic mod 2^32: group sums wrap at 32 bits (groupby/groupby.cpp:8-19)
  for (size_t i = static_cast<size_t>(blockIdx.x) * kCkThreads + threadIdx.x; i < n; i += stride) {
    const unsigned k = keys ? keys[i] : static_cast<unsigned>(i);
    const unsigned v = vals[i];
    a0 += v * wt0(k);
    a1 += v * wt1(k);
  }
  a0 = wave_reduce_add(a0);
  a1 = wave_reduce_add(a1);
  if ((threadIdx.x & (kWave - 1)) == kWave - 1) {  // 32-bit atomics on the low words: the sums wrap at 2^32
    if (a0) atomicAdd(reinterpret_cast<unsigned *>(result + 0), a0);
    if (a1) atomicAdd(reinterpret_cast<unsigned *>(result + 1), a1);
  }
}

// ---- permutation ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kCkThreads) void permutation_kernel(const unsigned *__restrict__ ids, size_t n,
                                                                 unsigned *bitmap, unsigned long long *result) {
  const size_t stride = static_cast<size_t>(gridDim.x) * kCkThreads;
  unsigned long long bad = 0;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kCkThreads + threadIdx.x; i < n; i += stride) {
    const unsigned v = ids[i];
    if (v >= n) {
      ++bad;
    } else {
      const unsigned bit = 1u << (v & 31u);
      if (atomicOr(&bitmap[v >> 5], bit) & bit) ++bad;
    }
  }
  block_add_u64(bad, result);
}

// ---- joins ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ size_t lower_bound_u32(const unsigned *__restrict__ a, size_t n, unsigned key) {
  size_t lo = 0, hi = n;
  while (lo < hi) {
    const size_t mid = lo + (hi - lo) / 2;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ size_t upper_bound_u32(const unsigned *__restrict__ a, size_t lo, size_t n, unsigned key) {
  size_t hi = n;
  while (lo < hi) {
    const size_t mid = lo + (hi - lo) / 2;
    if (a[mid] <= key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

struct GenSpec {
  unsigned long long seed, span;
  unsigned lo;
};
__device__ __forceinline__ unsigned gen_value(const GenSpec &g, unsigned long long index) {
  return g.lo + static_cast<unsigned>(mix64(g.seed, index) % g.span);
}

__global__ __launch_bounds__(kCkThreads) void join_check_kernel(
    const unsigned *__restrict__ sorted_build, size_t n_build, const unsigned *__restrict__ probe, size_t n_probe,
    const unsigned *__restrict__ out_pos, const unsigned *__restrict__ out_cnt, const unsigned *__restrict__ ids,
    const unsigned *__restrict__ build_keys, GenSpec gen, unsigned long long *result) {
  const size_t stride = static_cast<size_t>(gridDim.x) * kCkThreads;
  unsigned long long bad = 0, total = 0;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kCkThreads + threadIdx.x; i < n_probe; i += stride) {
    const unsigned key = probe[i];
    const size_t lb = lower_bound_u32(sorted_build, n_build, key);
    const size_t ub = upper_bound_u32(sorted_build, lb, n_build, key);
    const unsigned cnt = out_cnt[i], pos = out_pos[i];
    total += cnt;
    bool ok = cnt == ub - lb;
    if (ok && cnt) {
      ok = static_cast<size_t>(pos) + cnt <= n_build;
      if (ok) {
        const size_t pick[3] = {pos, static_cast<size_t>(pos) + cnt - 1,
                                pos + static_cast<size_t>(mix64(7, i) % cnt)};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const unsigned id = ids[pick[j]];
          if (build_keys)
            ok = ok && id < n_build && build_keys[id] == key;
          else
            ok = ok && gen_value(gen, id) == key;
        }
      }
    }
    if (!ok) ++bad;
  }
  block_add_u64(bad, result + 0);
  block_add_u64(total, result + 1);
}

__global__ __launch_bounds__(kCkThreads) void ujoin_check_kernel(
    const unsigned *__restrict__ sorted_build, const unsigned *__restrict__ build_vals, size_t n_build,
    const unsigned *__restrict__ probe, const unsigned *__restrict__ probe_vals, size_t n_probe,
    const unsigned *__restrict__ out_key, const unsigned *__restrict__ out_bval, const unsigned *__restrict__ out_pval,
    unsigned long long *result) {
  const size_t stride = static_cast<size_t>(gridDim.x) * kCkThreads;
  unsigned long long bad = 0, hits = 0;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kCkThreads + threadIdx.x; i < n_probe; i += stride) {
    const unsigned key = probe[i];
    const size_t lb = lower_bound_u32(sorted_build, n_build, key);
    const bool found = lb < n_build && sorted_build[lb] == key;
    bool ok;
    if (found) {
      ok = out_key[i] == key && out_bval[i] == build_vals[lb] && out_pval[i] == probe_vals[i];
      ++hits;
    } else {
      ok = out_key[i] == 0xFFFFFFFFu && out_bval[i] == 0xFFFFFFFFu && out_pval[i] == 0xFFFFFFFFu;
    }
    if (!ok) ++bad;
  }
  block_add_u64(bad, result + 0);
  block_add_u64(hits, result + 1);
}

__global__ __launch_bounds__(kCkThreads) void gen_check_kernel(const unsigned *__restrict__ values,
                                                               const unsigned *__restrict__ indices, size_t n,
                                                               unsigned long long first, GenSpec gen,
                                                               unsigned long long *result) {
  const size_t stride = static_cast<size_t>(gridDim.x) * kCkThreads;
  unsigned long long bad = 0;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kCkThreads + threadIdx.x; i < n; i += stride) {
    const unsigned long long idx = indices ? indices[i] : first + i;
    if (values[i] != gen_value(gen, idx)) ++bad;
  }
  block_add_u64(bad, result);
}

}  // namespace
}  // namespace dbhip

using namespace dbhip;

namespace {
constexpr unsigned kFpMaxBlocks = 1024;
struct FpGeo {
  unsigned blocks;
  size_t seg;
};
inline FpGeo fp_geo(size_t n) {
  FpGeo g;
  const size_t threads_wanted = (n + 63) / 64;  // >= 64 elements per thread
  size_t blocks = (threads_wanted + kCkThreads - 1) / kCkThreads;
  blocks = blocks < 1 ? 1 : (blocks > kFpMaxBlocks ? kFpMaxBlocks : blocks);
  g.blocks = static_cast<unsigned>(blocks);
  const size_t threads = blocks * kCkThreads;
  g.seg = (n + threads - 1) / threads;
  if (g.seg == 0) g.seg = 1;
  return g;
}
}  // namespace

extern "C" size_t dbhip_check_fingerprint_workspace_bytes(size_t n) {
  (void)n;
  return align_up(kWsHeader + kFpMaxBlocks * 16, kWsAlign);
}

extern "C" int dbhip_check_fingerprint_lt_i32(const int32_t *src, size_t n, int32_t filter_value, uint64_t *result,
                                              void *workspace, size_t workspace_bytes, dbhip_stream_t stream) {
  if (!result || (n && !src)) return DBHIP_EINVAL;
  if (!ws_ok(workspace, workspace_bytes, dbhip_check_fingerprint_workspace_bytes(n))) return DBHIP_EWORKSPACE;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  hipStream_t s = as_stream(stream);
  const FpGeo g = fp_geo(n);
  Fp *partial = reinterpret_cast<Fp *>(static_cast<char *>(workspace) + kWsHeader);
  hipLaunchKernelGGL(fp_partial_kernel, dim3(g.blocks), dim3(kCkThreads), 0, s, src, n, filter_value, g.seg, partial);
  hipLaunchKernelGGL(fp_final_kernel, dim3(1), dim3(kWave), 0, s, partial, g.blocks,
                     reinterpret_cast<unsigned long long *>(result));
  return launch_status();
}

extern "C" int dbhip_check_sorted_u32(const uint32_t *keys, size_t n, int signed_order, uint64_t *result,
                                      dbhip_stream_t stream) {
  if (!result || (n && !keys)) return DBHIP_EINVAL;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  hipStream_t s = as_stream(stream);
  const hipError_t e = fill_async(result, 0, 3 * sizeof(uint64_t), s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (n == 0) return DBHIP_OK;
  hipLaunchKernelGGL(sorted_kernel, dim3(ck_grid(n, dev)), dim3(kCkThreads), 0, s, keys, n,
                     signed_order ? 0x80000000u : 0u, reinterpret_cast<unsigned long long *>(result));
  return launch_status();
}

extern "C" int dbhip_check_weighted_sum_u32(const uint32_t *keys, const uint32_t *vals, size_t n, uint64_t *result,
                                            dbhip_stream_t stream) {
  if (!result || (n && !vals)) return DBHIP_EINVAL;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  hipStream_t s = as_stream(stream);
  const hipError_t e = fill_async(result, 0, 2 * sizeof(uint64_t), s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (n == 0) return DBHIP_OK;
  hipLaunchKernelGGL(weighted_sum_kernel, dim3(ck_grid(n, dev)), dim3(kCkThreads), 0, s, keys, vals, n,
                     reinterpret_cast<unsigned long long *>(result));
  return launch_status();
}

extern "C" size_t dbhip_check_permutation_workspace_bytes(size_t n) {
  return align_up(kWsHeader + ((n + 31) / 32 + 1) * sizeof(unsigned), kWsAlign);
}

extern "C" int dbhip_check_permutation_u32(const uint32_t *ids, size_t n, uint64_t *result, void *workspace,
                                           size_t workspace_bytes, dbhip_stream_t stream) {
  if (!result || (n && !ids) || n > 0xFFFFFFFFull) return DBHIP_EINVAL;
  if (!ws_ok(workspace, workspace_bytes, dbhip_check_permutation_workspace_bytes(n))) return DBHIP_EWORKSPACE;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  hipStream_t s = as_stream(stream);
  hipError_t e = fill_async(result, 0, sizeof(uint64_t), s);
  if (e == hipSuccess) e = fill_async(workspace, 0, dbhip_check_permutation_workspace_bytes(n), s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (n == 0) return DBHIP_OK;
  unsigned *bitmap = reinterpret_cast<unsigned *>(static_cast<char *>(workspace) + kWsHeader);
  hipLaunchKernelGGL(permutation_kernel, dim3(ck_grid(n, dev)), dim3(kCkThreads), 0, s, ids, n, bitmap,
                     reinterpret_cast<unsigned long long *>(result));
  return launch_status();
}

extern "C" int dbhip_check_join_u32(const uint32_t *sorted_build_keys, size_t n_build, const uint32_t *probe_keys,
                                    size_t n_probe, const uint32_t *out_pos, const uint32_t *out_count,
                                    const uint32_t *ids, const uint32_t *build_keys, uint64_t gen_seed, uint32_t gen_lo,
                                    uint32_t gen_hi, uint64_t *result, dbhip_stream_t stream) {
  if (!result || (n_probe && (!probe_keys || !out_pos || !out_count)) || (n_build && (!sorted_build_keys || !ids)))
    return DBHIP_EINVAL;
  if (!build_keys && gen_hi < gen_lo) return DBHIP_EINVAL;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  hipStream_t s = as_stream(stream);
  const hipError_t e = fill_async(result, 0, 2 * sizeof(uint64_t), s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (n_probe == 0) return DBHIP_OK;
  const GenSpec gen{gen_seed, static_cast<unsigned long long>(gen_hi) - gen_lo + 1, gen_lo};
  hipLaunchKernelGGL(join_check_kernel, dim3(ck_grid(n_probe, dev)), dim3(kCkThreads), 0, s, sorted_build_keys, n_build,
                     probe_keys, n_probe, out_pos, out_count, ids, build_keys, gen,
                     reinterpret_cast<unsigned long long *>(result));
  return launch_status();
}

extern "C" int dbhip_check_ujoin_u32(const uint32_t *sorted_build_keys, const uint32_t *build_vals, size_t n_build,
                                     const uint32_t *probe_keys, const uint32_t *probe_vals, size_t n_probe,
                                     const uint32_t *out_key, const uint32_t *out_build_val,
                                     const uint32_t *out_probe_val, uint64_t *result, dbhip_stream_t stream) {
  if (!result || (n_probe && (!probe_keys || !probe_vals || !out_key || !out_build_val || !out_probe_val)) ||
      (n_build && (!sorted_build_keys || !build_vals)))
    return DBHIP_EINVAL;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  hipStream_t s = as_stream(stream);
  const hipError_t e = fill_async(result, 0, 2 * sizeof(uint64_t), s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (n_probe == 0) return DBHIP_OK;
  hipLaunchKernelGGL(ujoin_check_kernel, dim3(ck_grid(n_probe, dev)), dim3(kCkThreads), 0, s, sorted_build_keys,
                     build_vals, n_build, probe_keys, probe_vals, n_probe, out_key, out_build_val, out_probe_val,
                     reinterpret_cast<unsigned long long *>(result));
  return launch_status();
}

extern "C" int dbhip_check_gen_uniform_u32(const uint32_t *values, const uint32_t *indices, size_t n, uint64_t seed,
                                           uint64_t first_index, uint32_t lo, uint32_t hi, uint64_t *result,
                                           dbhip_stream_t stream) {
  if (!result || (n && !values) || hi < lo) return DBHIP_EINVAL;
  const DeviceInfo &dev = current_device_info();
  if (!dev.ok) return DBHIP_ENODEVICE;
  hipStream_t s = as_stream(stream);
  const hipError_t e = fill_async(result, 0, sizeof(uint64_t), s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (n == 0) return DBHIP_OK;
  const GenSpec gen{seed, static_cast<unsigned long long>(hi) - lo + 1, lo};
  hipLaunchKernelGGL(gen_check_kernel, dim3(ck_grid(n, dev)), dim3(kCkThreads), 0, s, values, indices, n,
                     static_cast<unsigned long long>(first_index), gen, reinterpret_cast<unsigned long long *>(result));
  return launch_status();
}
