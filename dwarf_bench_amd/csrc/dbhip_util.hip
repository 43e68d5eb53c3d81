// dbhip_util.hip — library/device queries, workspace status read-back and the deterministic
// counter-based data generators (device twins of oracle/dbo_gen.c).
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
    }
  }
  return d;
}

namespace {

constexpr int kGenThreads = 256;

__global__ __launch_bounds__(kGenThreads) void gen_uniform_u32_kernel(uint32_t *out, size_t n,
                                                                       uint64_t seed,
                                                                       uint64_t first, uint32_t lo,
                                                                       uint64_t span) {
  const size_t stride = static_cast<size_t>(gridDim.x) * kGenThreads;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kGenThreads + threadIdx.x; i < n; i += stride)
    out[i] = lo + static_cast<uint32_t>(mix64(seed, first + i) % span);
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

extern "C" int dbhip_gen_unique_sorted_u32(uint32_t *out, size_t n, uint64_t seed,
                                           uint64_t first_index, dbhip_stream_t stream) {
  if (n == 0) return DBHIP_OK;
  if (!out || 10ull * (first_index + n) > 0xFFFFFFFFull) return DBHIP_EINVAL;
  hipLaunchKernelGGL(gen_unique_sorted_u32_kernel, dim3(gen_grid(n)), dim3(kGenThreads), 0,
                     as_stream(stream), out, n, seed, first_index);
  return launch_status();
}
