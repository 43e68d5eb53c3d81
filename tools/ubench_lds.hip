// ubench_lds.hip — what LDS atomics cost on gfx950 when nothing else is in the way: indices come from a register
// hash (no index stream from memory), 16 waves or 32 waves per CU, random addresses over 2^bits bins.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_lds.hip -o tools/ubench_lds
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ unsigned mix(unsigned h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }

// MODE 0: ds_add (no return)  1: ds_add_rtn, result consumed  2: plain read then ds_cmpst on "empty"  3: ds_read only
template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void k(unsigned per_thread, unsigned lmask, unsigned *out) {
  extern __shared__ unsigned s[];
  for (unsigned i = threadIdx.x; i <= lmask; i += THREADS) s[i] = MODE == 2 ? 0xFFFFFFFFu : 0u;
  __syncthreads();
  unsigned acc = 0, x = blockIdx.x * THREADS + threadIdx.x;
  for (unsigned it = 0; it < per_thread; it += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      x = mix(x + 0x9E3779B9u);
      const unsigned j = x & lmask;
      if (MODE == 0) atomicAdd(&s[j], 1u);
      else if (MODE == 1) acc += atomicAdd(&s[j], 1u);
      else if (MODE == 2) { unsigned v = s[j]; if (v == 0xFFFFFFFFu) v = atomicCAS(&s[j], 0xFFFFFFFFu, x | 1u); acc += v; }
      else acc += s[j];
    }
  }
  __syncthreads();
  if (acc == 0x12345678u || s[threadIdx.x & lmask] == 0x87654321u) out[0] = acc;
}

template <typename F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  std::vector<float> t;
  for (int i = 0; i < 5; ++i) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); t.push_back(ms); }
  std::sort(t.begin(), t.end()); return t[2];
}

template <int MODE, int THREADS> void run(const char *what, int cus, int wg_per_cu, unsigned bits, unsigned *out) {
  const unsigned per_thread = 4096;
  const size_t lds = (size_t)4 << bits;
  hipFuncSetAttribute((const void *)k<MODE, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const float t = timeit([&] { hipLaunchKernelGGL((k<MODE, THREADS>), dim3(cus * wg_per_cu), dim3(THREADS), lds, 0, per_thread, (1u << bits) - 1, out); });
  const double ops = (double)cus * wg_per_cu * THREADS * per_thread;
  printf("%-34s %2d x %4d thr/CU, 2^%-2u bins: %7.1f us  %8.1f G ops/s  (%.2f per clk per CU at 2.1 GHz)\n", what, wg_per_cu, THREADS, bits, t * 1e3,
         ops / t / 1e6, ops / t / 1e6 / cus / 2.1);
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  unsigned *out; CK(hipMalloc(&out, 256));
  for (unsigned bits : {8u, 12u, 13u}) {
    run<0, 512>("ds_add (no return)", cus, 4, bits, out);
    run<1, 512>("ds_add returning, value used", cus, 4, bits, out);
    run<2, 512>("read, ds_cmpst on empty", cus, 4, bits, out);
    run<3, 512>("ds_read_b32 random", cus, 4, bits, out);
  }
  run<0, 1024>("ds_add (no return)", cus, 1, 15, out);
  run<1, 1024>("ds_add returning, value used", cus, 1, 15, out);
  run<0, 1024>("ds_add (no return)", cus, 2, 12, out);
  return 0;
}
