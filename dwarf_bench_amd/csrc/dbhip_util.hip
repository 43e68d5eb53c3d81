// dbhip_util.hip — library/device queries, workspace status read-back and the deterministic
// counter-based data generators (device twins of dbo_gen_uniform_u32 / dbo_gen_unique_sorted_u32 in oracle/dbo.c).
#include <cstring>
#include <mutex>

#include "dbhip_common.hpp"

namespace dbhip {

const DeviceInfo &current_device_info() {
  static DeviceInfo cache[64];
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    static DeviceInfo none;
    return none;
  }
  std::lock_guard<std::mutex> lock(mu);
  DeviceInfo &d = cache[dev];
  if (!d.ok) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) == hipSuccess) {
      d.cus = p.multiProcessorCount;
      d.wave = p.warpSize;
      d.ok = d.cus > 0 && d.wave == kWave;
      d.gfx950 = std::strncmp(p.gcnArchName, "gfx950", 6) == 0;
    }
  }
  return d;
}

namespace {

constexpr int kFillThreads = 256;

// words [0, head) and [head + 4*vecs, words) one by one, the 16-byte aligned middle as u32x4 stores
__global__ __launch_bounds__(kFillThreads) void fill_kernel(unsigned *p, unsigned v, size_t words, size_t head,
                                                             size_t vecs) {
  const size_t stride = static_cast<size_t>(gridDim.x) * kFillThreads;
  const size_t t = static_cast<size_t>(blockIdx.x) * kFillThreads + threadIdx.x;
  u32x4 *mid = reinterpret_cast<u32x4 *>(p + head);
  for (size_t i = t; i < vecs; i += stride) mid[i] = u32x4{v, v, v, v};
  const size_t tail0 = head + 4 * vecs;
  for (size_t i = t; i < head; i += stride) p[i] = v;
  for (size_t i = tail0 + t; i < words; i += stride) p[i] = v;
}

constexpr int kGenThreads = 256;

__global__ __launch_bounds__(kGenThreads) void gen_uniform_u32_kernel(uint32_t *out, size_t n,
                                                                       uint64_t seed,
                                                                       uint64_t first, uint32_t lo,
                                                                       uint64_t span) {
  const size_t stride = static_cast<size_t>(gridDim.x) * kGenThreads;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kGenThreads + threadIdx.x; i < n; i += stride)
    out[i] = lo + static_cast<uint32_t>(mix64(seed, first + i) % span);
}

__global__ __launch_bounds__(kGenThreads) void gen_uniform_at_u32_kernel(uint32_t *out, const uint32_t *__restrict__ indices,
                                                                          size_t n, uint64_t seed, uint32_t lo, uint64_t span) {
  const size_t stride = static_cast<size_t>(gridDim.x) * kGenThreads;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kGenThreads + threadIdx.x; i < n; i += stride)
    out[i] = lo + static_cast<uint32_t>(mix64(seed, indices[i]) % span);
}

__global__ __launch_bounds__(kGenThreads) void gen_unique_sorted_u32_kernel(uint32_t *out, size_t n,
                                                                             uint64_t seed,
                                                                             uint64_t first) {
  const size_t stride = static_cast<size_t>(gridDim.x) * kGenThreads;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kGenThreads + threadIdx.x; i < n; i += stride)
    out[i] = static_cast<uint32_t>(10ull * (first + i) + mix64(seed, first + i) % 10ull);
}

inline unsigned gen_grid(size_t n) {
  const DeviceInfo &d = current_device_info();
  const size_t want = (n + kGenThreads - 1) / kGenThreads;
  const size_t cap = static_cast<size_t>(d.ok ? d.cus : 256) * 8;
  return static_cast<unsigned>(want < cap ? (want ? want : 1) : cap);
}

}  // namespace

hipError_t fill_async(void *p, int value, size_t bytes, hipStream_t s) {
  if (bytes == 0) return hipSuccess;
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  if (!p || (a & 3u) || (bytes & 3u)) return hipErrorInvalidValue;
  const size_t words = bytes / 4;
  size_t head = ((16 - (a & 15u)) & 15u) / 4;
  if (head > words) head = words;
  const size_t vecs = (words - head) / 4;
  const unsigned b = static_cast<unsigned>(value) & 0xFFu;
  const size_t want = (vecs + kFillThreads - 1) / kFillThreads;
  const DeviceInfo &d = current_device_info();
  const size_t cap = static_cast<size_t>(d.ok ? d.cus : 256) * 8;
  const unsigned grid = static_cast<unsigned>(want < cap ? (want ? want : 1) : cap);
  hipLaunchKernelGGL(fill_kernel, dim3(grid), dim3(kFillThreads), 0, s, static_cast<unsigned *>(p), b * 0x01010101u,
                     words, head, vecs);
  return hipGetLastError();
}

}  // namespace dbhip

using namespace dbhip;

extern "C" int dbhip_version(void) { return DBHIP_VERSION; }

extern "C" int dbhip_device_info(int device, char *name, size_t len, int *compute_units,
                                 int *wave_size) {
  hipDeviceProp_t p;
  hipError_t e = hipGetDeviceProperties(&p, device);
  if (e != hipSuccess) return DBHIP_ENODEVICE;
  if (name && len) {
    std::strncpy(name, p.gcnArchName, len - 1);
    name[len - 1] = 0;
  }
  if (compute_units) *compute_units = p.multiProcessorCount;
  if (wave_size) *wave_size = p.warpSize;
  return DBHIP_OK;
}

extern "C" int dbhip_workspace_status(const void *workspace, uint32_t *host_status,
                                      dbhip_stream_t stream) {
  if (!workspace || !host_status) return DBHIP_EINVAL;
  hipError_t e = hipMemcpyAsync(host_status, workspace, sizeof(uint32_t), hipMemcpyDeviceToHost,
                                as_stream(stream));
  if (e != hipSuccess) return static_cast<int>(e);
  return static_cast<int>(hipStreamSynchronize(as_stream(stream)));
}

extern "C" int dbhip_gen_uniform_u32(uint32_t *out, size_t n, uint64_t seed, uint64_t first_index,
                                     uint32_t lo, uint32_t hi, dbhip_stream_t stream) {
  if (n == 0) return DBHIP_OK;
  if (!out || hi < lo) return DBHIP_EINVAL;
  const uint64_t span = static_cast<uint64_t>(hi) - lo + 1;
  hipLaunchKernelGGL(gen_uniform_u32_kernel, dim3(gen_grid(n)), dim3(kGenThreads), 0,
                     as_stream(stream), out, n, seed, first_index, lo, span);
  return launch_status();
}

extern "C" int dbhip_gen_uniform_at_u32(uint32_t *out, const uint32_t *indices, size_t n, uint64_t seed, uint32_t lo,
                                        uint32_t hi, dbhip_stream_t stream) {
  if (n == 0) return DBHIP_OK;
  if (!out || !indices || hi < lo) return DBHIP_EINVAL;
  const uint64_t span = static_cast<uint64_t>(hi) - lo + 1;
  hipLaunchKernelGGL(gen_uniform_at_u32_kernel, dim3(gen_grid(n)), dim3(kGenThreads), 0, as_stream(stream), out, indices, n,
                     seed, lo, span);
  return launch_status();
}

extern "C" int dbhip_gen_unique_sorted_u32(uint32_t *out, size_t n, uint64_t seed,
                                           uint64_t first_index, dbhip_stream_t stream) {
  if (n == 0) return DBHIP_OK;
  if (!out || 10ull * (first_index + n) > 0xFFFFFFFFull) return DBHIP_EINVAL;
  hipLaunchKernelGGL(gen_unique_sorted_u32_kernel, dim3(gen_grid(n)), dim3(kGenThreads), 0,
                     as_stream(stream), out, n, seed, first_index);
  return launch_status();
}
