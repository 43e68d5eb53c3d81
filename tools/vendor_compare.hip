// vendor_compare.hip — development aid, not part of the product or its tests: times ROCm's own tuned
// primitives (rocPRIM) on the workloads of BASELINE.json, on the same box, so that DESIGN.md can say where
// the hand-written kernels stand next to the vendor library (context for the roofline fractions).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/vendor_compare.hip -o tools/_vendor_compare
#include <cstring>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <vector>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return 1;                                                                 \
    }                                                                           \
  } while (0)

__device__ __host__ inline uint64_t mix64(uint64_t seed, uint64_t i) {  // same generator as libdbhip / the oracle
  uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void gen(uint32_t *out, size_t n, uint64_t seed, uint32_t lo, uint64_t span) {
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
    out[i] = lo + uint32_t(mix64(seed, i) % span);
}

struct LessThan {
  int v;
  __device__ bool operator()(const int &x) const { return x < v; }
};

template <class F>
static float median_us(F &&f, int iters = 9) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  std::vector<float> t;
  for (int i = 0; i < iters + 2; ++i) {
    (void)hipEventRecord(a, nullptr);
    f();
    (void)hipEventRecord(b, nullptr);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    if (i >= 2) t.push_back(ms * 1000.f);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main() {
  {  // select (copy_if x < 5) on 2^28 int32, reference distribution
    const size_t n = size_t(1) << 28;
    int *src, *out;
    size_t *count;
    CK(hipMalloc(&src, n * 4));
    CK(hipMalloc(&out, n * 4));
    CK(hipMalloc(&count, 8));
    gen<<<4096, 256>>>(reinterpret_cast<uint32_t *>(src), n, 42, 1, 10000);
    for (int filt : {5, 5001}) {
      size_t tmp_bytes = 0;
      CK(rocprim::select(nullptr, tmp_bytes, src, out, count, n, LessThan{filt}));
      void *tmp;
      CK(hipMalloc(&tmp, tmp_bytes));
      const float us = median_us([&] { (void)rocprim::select(tmp, tmp_bytes, src, out, count, n, LessThan{filt}); });
      size_t c = 0;
      CK(hipMemcpy(&c, count, 8, hipMemcpyDeviceToHost));
      std::printf("rocprim::select n=2^28 filter=%d: %.1f us  %.3f TB/s (%.1f%% of 8 TB/s), %zu selected\n", filt, us,
                  (4.0 * n + 4.0 * c) / us / 1e6, (4.0 * n + 4.0 * c) / us / 1e6 / 8 * 100, c);
      CK(hipFree(tmp));
    }
    {  // reduce
      int *sum;
      CK(hipMalloc(&sum, 4));
      size_t tmp_bytes = 0;
      CK(rocprim::reduce(nullptr, tmp_bytes, src, sum, 0, n, rocprim::plus<int>()));
      void *tmp;
      CK(hipMalloc(&tmp, tmp_bytes));
      const float us = median_us([&] { (void)rocprim::reduce(tmp, tmp_bytes, src, sum, 0, n, rocprim::plus<int>()); });
      std::printf("rocprim::reduce n=2^28: %.1f us  %.3f TB/s (%.1f%%)\n", us, 4.0 * n / us / 1e6, 4.0 * n / us / 1e6 / 8 * 100);
      CK(hipFree(tmp));
      CK(hipFree(sum));
    }
    CK(hipFree(src));
    CK(hipFree(out));
    CK(hipFree(count));
  }
  for (int lg : {20, 24}) {  // radix sort of uint32 keys
    const size_t n = size_t(1) << lg;
    uint32_t *src, *a, *b;
    CK(hipMalloc(&src, n * 4));
    CK(hipMalloc(&a, n * 4));
    CK(hipMalloc(&b, n * 4));
    for (int full = 1; full >= 0; --full) {
      gen<<<4096, 256>>>(src, n, 42, full ? 0u : 1u, full ? (uint64_t(1) << 32) : 10000ull);
      size_t tmp_bytes = 0;
      CK(rocprim::radix_sort_keys(nullptr, tmp_bytes, a, b, n));
      void *tmp;
      CK(hipMalloc(&tmp, tmp_bytes));
      const float copy_us = median_us([&] { (void)hipMemcpyAsync(a, src, n * 4, hipMemcpyDeviceToDevice, nullptr); });
      const float us = median_us([&] {
        (void)hipMemcpyAsync(a, src, n * 4, hipMemcpyDeviceToDevice, nullptr);
        (void)rocprim::radix_sort_keys(tmp, tmp_bytes, a, b, n);
      });
      std::printf("rocprim::radix_sort_keys n=2^%d %s: %.1f us (copy %.1f excluded)  %.0f Mkeys/s\n", lg,
                  full ? "full-range" : "ref[1,10000]", us - copy_us, copy_us, n / (us - copy_us));
      CK(hipFree(tmp));
    }
    CK(hipFree(src));
    CK(hipFree(a));
    CK(hipFree(b));
  }
  return 0;
}
