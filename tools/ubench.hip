// ubench.hip — memory-system micro-benchmarks that size the dwarf kernels' designs on MI355X
// (stream read, random gather, random atomics, random scatter).  Standalone: hipcc tools/ubench.hip -o ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline uint64_t mix64(uint64_t seed, uint64_t i) {
  uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull;
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31; return z;
}

template <bool NT, int VPT>
__global__ __launch_bounds__(256) void k_read(const u32x4* __restrict__ p, size_t n4, unsigned* out) {
  unsigned acc = 0;
  const size_t stride = (size_t)gridDim.x * 256 * VPT;
  for (size_t base = (size_t)blockIdx.x * 256 * VPT + threadIdx.x; base < n4; base += stride) {
    u32x4 v[VPT];
#pragma unroll
    for (int k = 0; k < VPT; ++k) { size_t i = base + (size_t)k * 256; v[k] = i < n4 ? (NT ? __builtin_nontemporal_load(p + i) : p[i]) : u32x4{0,0,0,0}; }
#pragma unroll
    for (int k = 0; k < VPT; ++k) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_copy(const u32x4* __restrict__ p, u32x4* __restrict__ q, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) q[i] = p[i];
}
__global__ __launch_bounds__(256) void k_fill(unsigned* p, size_t n, unsigned v) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = v;
}
__global__ __launch_bounds__(256) void k_gen_idx(unsigned* p, size_t n, unsigned mask, uint64_t seed) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = (unsigned)mix64(seed, i) & mask;
}
// mode 0: gather 4B, 1: gather 16B (idx&~3), 2: atomicAdd, 3: atomicCAS, 4: scatter 4B store, 5: scatter 8B store
template <int MODE>
__global__ __launch_bounds__(256) void k_random(const unsigned* __restrict__ idx, size_t n, unsigned* table, unsigned* out) {
  unsigned acc = 0;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const unsigned j = idx[i];
    if (MODE == 0) acc += table[j];
    else if (MODE == 1) { u32x4 v = *reinterpret_cast<const u32x4*>(table + (j & ~3u)); acc += v.x + v.w; }
    else if (MODE == 2) atomicAdd(&table[j], 1u);
    else if (MODE == 3) acc += atomicCAS(&table[j], 0xFFFFFFFFu, (unsigned)i);
    else if (MODE == 4) table[j] = (unsigned)i;
    else { uint2 v = make_uint2((unsigned)i, j); *reinterpret_cast<uint2*>(table + (j & ~1u)) = v; }
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(1024) void k_lds_atomic(const unsigned* __restrict__ idx, size_t n, unsigned lmask, unsigned* out) {
  extern __shared__ unsigned s[];
  for (unsigned i = threadIdx.x; i <= lmask; i += 1024) s[i] = 0;
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * 1024;
  for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += stride) atomicAdd(&s[idx[i] & lmask], 1u);
  __syncthreads();
  if (s[threadIdx.x & lmask] == 0x12345678u) out[0] = 1;
}

template <typename F> float timeit(F f, int iters = 5) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int i = 0; i < iters; ++i) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); }
  std::sort(t.begin(), t.end()); return t[t.size() / 2];
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs %d\n", prop.gcnArchName, prop.multiProcessorCount);
  const int cus = prop.multiProcessorCount;
  unsigned* out; CK(hipMalloc(&out, 256));
  const size_t N = (size_t)1 << 28;  // 1 GiB of u32
  unsigned *a, *b; CK(hipMalloc(&a, N * 4)); CK(hipMalloc(&b, N * 4));
  hipLaunchKernelGGL(k_fill, dim3(cus * 8), dim3(256), 0, 0, a, N, 7u);
  CK(hipDeviceSynchronize());
  for (int per_cu : {2, 4, 8, 16}) {
    float t1 = timeit([&] { hipLaunchKernelGGL((k_read<false, 8>), dim3(cus * per_cu), dim3(256), 0, 0, (const u32x4*)a, N / 4, out); });
    float t2 = timeit([&] { hipLaunchKernelGGL((k_read<true, 8>), dim3(cus * per_cu), dim3(256), 0, 0, (const u32x4*)a, N / 4, out); });
    float t3 = timeit([&] { hipLaunchKernelGGL((k_read<true, 4>), dim3(cus * per_cu), dim3(256), 0, 0, (const u32x4*)a, N / 4, out); });
    float t4 = timeit([&] { hipLaunchKernelGGL((k_read<true, 16>), dim3(cus * per_cu), dim3(256), 0, 0, (const u32x4*)a, N / 4, out); });
    printf("read 1GiB blocks/CU=%2d: plain vpt8 %.1f us %.2f TB/s | nt vpt8 %.1f us %.2f TB/s | nt vpt4 %.2f TB/s | nt vpt16 %.2f TB/s\n", per_cu, t1 * 1e3, N * 4 / t1 / 1e9, t2 * 1e3, N * 4 / t2 / 1e9, N * 4 / t3 / 1e9, N * 4 / t4 / 1e9);
  }
  { float t = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(cus * 8), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, N / 4); });
    printf("copy 1GiB: %.1f us  %.2f TB/s (r+w)\n", t * 1e3, 2.0 * N * 4 / t / 1e9); }
  { float t = timeit([&] { CK(hipMemsetAsync(b, 0, N * 4, 0)); });
    printf("memset 1GiB: %.1f us  %.2f TB/s\n", t * 1e3, 1.0 * N * 4 / t / 1e9); }
  const size_t M = (size_t)1 << 26;  // accesses
  unsigned* idx; CK(hipMalloc(&idx, M * 4));
  for (int lg : {20, 24, 26, 27, 28}) {   // table of 2^lg u32: 4 MiB .. 1 GiB
    const unsigned mask = (1u << lg) - 1;
    hipLaunchKernelGGL(k_gen_idx, dim3(cus * 8), dim3(256), 0, 0, idx, M, mask, 1234ull + lg);
    hipLaunchKernelGGL(k_fill, dim3(cus * 8), dim3(256), 0, 0, b, (size_t)1 << lg, 0xFFFFFFFFu);
    CK(hipDeviceSynchronize());
    float g4 = timeit([&] { hipLaunchKernelGGL((k_random<0>), dim3(cus * 8), dim3(256), 0, 0, idx, M, b, out); });
    float g16 = timeit([&] { hipLaunchKernelGGL((k_random<1>), dim3(cus * 8), dim3(256), 0, 0, idx, M, b, out); });
    float aa = timeit([&] { hipLaunchKernelGGL((k_random<2>), dim3(cus * 8), dim3(256), 0, 0, idx, M, b, out); }, 3);
    hipLaunchKernelGGL(k_fill, dim3(cus * 8), dim3(256), 0, 0, b, (size_t)1 << lg, 0xFFFFFFFFu);
    float ac = timeit([&] { hipLaunchKernelGGL((k_random<3>), dim3(cus * 8), dim3(256), 0, 0, idx, M, b, out); }, 3);
    float s4 = timeit([&] { hipLaunchKernelGGL((k_random<4>), dim3(cus * 8), dim3(256), 0, 0, idx, M, b, out); });
    float s8 = timeit([&] { hipLaunchKernelGGL((k_random<5>), dim3(cus * 8), dim3(256), 0, 0, idx, M, b, out); });
    printf("random 2^26 ops, table %5zu MiB: gather4 %.0f us (%.1f G/s) gather16 %.0f us | atomicAdd %.0f us (%.1f G/s) atomicCAS %.0f us | scatter4 %.0f us (%.1f G/s) scatter8 %.0f us\n",
           ((size_t)4 << lg) >> 20, g4 * 1e3, M / g4 / 1e6, g16 * 1e3, aa * 1e3, M / aa / 1e6, ac * 1e3, s4 * 1e3, M / s4 / 1e6, s8 * 1e3);
  }
  for (unsigned lbits : {4u, 8u, 12u, 15u}) {
    hipLaunchKernelGGL(k_gen_idx, dim3(cus * 8), dim3(256), 0, 0, idx, M, 0xFFFFFFFFu, 99ull);
    CK(hipFuncSetAttribute((const void*)k_lds_atomic, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    float t = timeit([&] { hipLaunchKernelGGL(k_lds_atomic, dim3(cus), dim3(1024), (size_t)4 << lbits, 0, idx, M, (1u << lbits) - 1, out); });
    printf("LDS atomicAdd 2^26 ops into %u bins/WG (+4B idx read): %.0f us (%.1f G/s)\n", 1u << lbits, t * 1e3, M / t / 1e6);
  }
  return 0;
}
