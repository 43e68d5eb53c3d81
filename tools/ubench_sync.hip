// ubench_sync.hip — what a dependent kernel launch costs against a grid barrier inside one resident kernel, on the data
// movement of an LSD sort pass ("the fused cooperative pass").
// A phase moves 2^24 keys: workgroup c reads chunk c (8192 keys) and writes it to chunk perm(c) of the other buffer —
// the next phase's reader of a chunk is another workgroup, most likely on another XCD, so every phase boundary has to
// make the whole 64 MiB visible across the chip, as a sort pass's does.  Timed, K phases each:
//   launches        one launch per phase (2048 workgroups)
//   launches x3     the same + two tiny dependent launches per phase (the histogram / scan kernels' boundaries)
//   fused           one cooperative launch of the resident grid, a grid barrier per phase
//   fused x3        the same + two more barriers per phase
// and the final buffer is checked (a barrier that does not publish the data would show).
// hipcc --offload-arch=gfx950 -O3 tools/ubench_sync.hip -o tools/ubench_sync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
namespace cg = cooperative_groups;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kThreads = 512, kChunkKeys = 8192, kChunks = 2048;  // 2^24 keys
constexpr unsigned kMul = 257;                                         // perm(c) = c * 257 mod 2048 (257 is odd: a permutation)

__device__ __forceinline__ void move_chunk(const unsigned *src, unsigned *dst, unsigned c) {
  const u32x4 *s = reinterpret_cast<const u32x4 *>(src + static_cast<size_t>(c) * kChunkKeys);
  u32x4 *d = reinterpret_cast<u32x4 *>(dst + static_cast<size_t>((c * kMul) % kChunks) * kChunkKeys);
  u32x4 v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = s[j * kThreads + threadIdx.x];
#pragma unroll
  for (int j = 0; j < 4; ++j) d[j * kThreads + threadIdx.x] = v[j];
}

__global__ __launch_bounds__(kThreads) void k_phase(const unsigned *src, unsigned *dst) { move_chunk(src, dst, blockIdx.x); }
__global__ __launch_bounds__(256) void k_tiny(unsigned *scratch) {  // a dependent launch with next to nothing in it
  scratch[blockIdx.x * 256 + threadIdx.x] += 1;
}

// a barrier of all workgroups of a resident grid: one arrival per workgroup on a counter that only grows.  ALLFENCE: every
// wave releases its own stores before the workgroup barrier (thread 0's fence alone waits for ITS wave's stores only; the
// runtime's grid.sync() is built like the thread-0 form).  The wait is bounded: a barrier that never completes ends the
// kernel with wrong data instead of hanging the device.
template <bool ALLFENCE>
__device__ __forceinline__ void grid_barrier(unsigned *counter, unsigned &epoch) {
  if (ALLFENCE) __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    epoch += gridDim.x;
    __threadfence();  // release: device scope (L2 write-back on a multi-XCD part)
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch && ++spins < (1u << 22))
      __builtin_amdgcn_s_sleep(2);
    __threadfence();  // acquire
  }
  __syncthreads();
}

template <int BARRIERS, int KIND>  // KIND 0: own barrier, thread 0 fences; 1: own barrier, every wave fences; 2: grid.sync()
__global__ __launch_bounds__(kThreads) void k_fused(unsigned *a, unsigned *b, unsigned phases, unsigned *counter) {
  unsigned epoch = 0;
  cg::grid_group grid = cg::this_grid();
  for (unsigned p = 0; p < phases; ++p) {
    const unsigned *src = (p & 1) ? b : a;
    unsigned *dst = (p & 1) ? a : b;
    for (unsigned c = blockIdx.x; c < kChunks; c += gridDim.x) move_chunk(src, dst, c);
    for (int k = 0; k < BARRIERS; ++k) {
      if (KIND == 2) grid.sync();
      else grid_barrier<KIND == 1>(counter, epoch);
    }
  }
}

static double ms_of(hipEvent_t a, hipEvent_t b) { float t; CK(hipEventElapsedTime(&t, a, b)); return t; }

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs %d cooperative launch %d\n", prop.gcnArchName, prop.multiProcessorCount, prop.cooperativeLaunch);
  const size_t n = static_cast<size_t>(kChunks) * kChunkKeys;
  unsigned *a, *b, *scratch, *counter;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&scratch, 256 * 256 * 4)); CK(hipMalloc(&counter, 256));
  CK(hipMemset(scratch, 0, 256 * 256 * 4));
  std::vector<unsigned> host(n);
  for (size_t i = 0; i < n; ++i) host[i] = static_cast<unsigned>(i * 2654435761u);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned K = 48;  // even: the data ends in `a`
  auto check = [&](const char *what) {
    std::vector<unsigned> got(n);
    CK(hipMemcpy(got.data(), a, n * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (unsigned c = 0; c < kChunks; ++c) {
      unsigned at = c;
      for (unsigned p = 0; p < K; ++p) at = (at * kMul) % kChunks;
      for (unsigned j = 0; j < kChunkKeys; j += 509)
        bad += got[static_cast<size_t>(at) * kChunkKeys + j] != host[static_cast<size_t>(c) * kChunkKeys + j];
    }
    printf("  %-34s data %s\n", what, bad ? "WRONG" : "ok");
  };
  auto reset = [&] { CK(hipMemcpy(a, host.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(counter, 0, 256)); };
  for (int rep = 0; rep < 2; ++rep) {
    // ---- one launch per phase
    for (int extra = 0; extra <= 2; extra += 2) {
      reset();
      for (int warm = 0; warm < 2; ++warm) {
        if (warm) CK(hipEventRecord(e0));
        for (unsigned p = 0; p < K; ++p) {
          hipLaunchKernelGGL(k_phase, dim3(kChunks), dim3(kThreads), 0, 0, (p & 1) ? b : a, (p & 1) ? a : b);
          for (int t = 0; t < extra; ++t) hipLaunchKernelGGL(k_tiny, dim3(256), dim3(256), 0, 0, scratch);
        }
        if (warm) CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
      }
      printf("launches, %d per phase: %7.2f us per phase\n", 1 + extra, ms_of(e0, e1) * 1e3 / K);
      if (rep == 0) { reset(); for (unsigned p = 0; p < K; ++p) hipLaunchKernelGGL(k_phase, dim3(kChunks), dim3(kThreads), 0, 0, (p & 1) ? b : a, (p & 1) ? a : b); CK(hipDeviceSynchronize()); check("launches"); }
    }
    // ---- one resident kernel, grid barriers
    auto fused = [&](const char *name, const void *fn, int barriers) {
      int per_cu = 0;
      CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kThreads, 0));
      for (int want : {1, 2, 4, 8}) {
        if (want > per_cu) continue;
        const unsigned grid = static_cast<unsigned>(prop.multiProcessorCount * want);
        unsigned phases = K;
        void *args[] = {&a, &b, &phases, &counter};
        reset();
        CK(hipLaunchCooperativeKernel(fn, dim3(grid), dim3(kThreads), args, 0, 0));
        CK(hipDeviceSynchronize());
        if (rep == 0) check(name);
        reset();
        CK(hipEventRecord(e0));
        CK(hipLaunchCooperativeKernel(fn, dim3(grid), dim3(kThreads), args, 0, 0));
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        printf("%-32s %d barrier(s) per phase, %d workgroups per CU: %7.2f us per phase\n", name, barriers, want, ms_of(e0, e1) * 1e3 / K);
      }
    };
    fused("fused, own barrier (t0 fence)", reinterpret_cast<const void *>(k_fused<1, 0>), 1);
    fused("fused, own barrier (t0 fence)", reinterpret_cast<const void *>(k_fused<3, 0>), 3);
    fused("fused, own barrier (all fence)", reinterpret_cast<const void *>(k_fused<1, 1>), 1);
    fused("fused, own barrier (all fence)", reinterpret_cast<const void *>(k_fused<3, 1>), 3);
    fused("fused, grid.sync()", reinterpret_cast<const void *>(k_fused<1, 2>), 1);
    fused("fused, grid.sync()", reinterpret_cast<const void *>(k_fused<3, 2>), 3);
  }
  return 0;
}
