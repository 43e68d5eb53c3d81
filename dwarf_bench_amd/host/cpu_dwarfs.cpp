// cpu_dwarfs.cpp — the two HOST dwarfs of the hot path, under the reference's own registry names, for --device=cpu:
//   TwoPassScan   scan/scan.cpp:22-195 with its kernel scan/scan.cl:3-42 (on the reference this is the OpenCL CPU device:
//                 BASELINE config 1, `dwarf_bench TwoPassScan --device=cpu --input_size=1024 --iterations=9`)
//   TBBSort       sort/tbbsort.cpp:15-48 (oneTBB parallel_sort: the CPU baseline of the Sort dwarf)
// Plain C++ over std::thread — no OpenCL CPU runtime and no oneTBB exist on a ROCm box, and none is needed: what these
// names owe a caller is the plugin contract (init / run / Result per iteration / `buf_size` / CSV) and the operation's
// result.  They touch no GPU and nothing under oracle/ (that is test infrastructure); a dwarf asked to run on another
// device than the CPU throws std::logic_error like the reference's device selection does for an unsupported type.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <iostream>
#include <numeric>
#include <stdexcept>
#include <thread>
#include <vector>

#include "cpu_dwarfs.hpp"

namespace {

using clk = std::chrono::steady_clock;

// the library's deterministic stand-in for helpers::make_random<int>(n) (common/common.hpp:31-40: uniform [1, 10000]):
// the same counter-based generator the device dwarfs fill their columns with (csrc/dbhip_common.hpp mix64), so a CPU
// run and a HIP run of one size see the same data
uint64_t mix64(uint64_t seed, uint64_t i) {
  uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull;
  z ^= z >> 30;
  z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27;
  z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  return z;
}
std::vector<int> make_column(size_t n, uint64_t seed) {
  std::vector<int> v(n);
  for (size_t i = 0; i < n; ++i) v[i] = 1 + static_cast<int>(mix64(seed, i) % 10000ull);
  return v;
}

unsigned host_threads(size_t n) {
  unsigned t = std::thread::hardware_concurrency();
  if (t == 0) t = 8;  // the reference's CPU threadnum (scan/scan.cpp:65-70)
  const size_t by_size = n / 4096 + 1;  // a thread per 4096 rows at least: 1024 rows run on one
  return static_cast<unsigned>(std::min<size_t>(t, by_size));
}

template <class F>
void parallel_chunks(unsigned threads, F &&body) {  // body(t) for t in [0, threads)
  if (threads <= 1) {
    body(0u);
    return;
  }
  std::vector<std::thread> pool;
  pool.reserve(threads - 1);
  for (unsigned t = 1; t < threads; ++t) pool.emplace_back([&body, t] { body(t); });
  body(0u);
  for (auto &th : pool) th.join();
}

void require_cpu(const RunOptions &opts, const char *who) {
  if (opts.device_ty != RunOptions::DeviceType::CPU && opts.device_ty != RunOptions::DeviceType::Default)
    throw std::logic_error(std::string(who) + " is the host dwarf: run it with --device=cpu (the HIP one is " + who + "Hip)");
}

}  // namespace

// =====================================================================================================
TwoPassScan::TwoPassScan() : Dwarf("TwoPassScan") {}
void TwoPassScan::init(const RunOptions &opts) {
  require_cpu(opts, "TwoPassScan");
  meter().set_opts(opts);
  meter().set_params({{"device_type", to_string(opts.device_ty)}});
}
void TwoPassScan::run(const RunOptions &opts) {
  for (auto size : opts.input_size) _run(size, meter());
}
// scan.cl:3-42 with T chunks on T threads: count per chunk, exclusive prefix of the counts, write per chunk at the
// chunk's offset.  Unlike the reference kernel the last chunk takes the n % T tail too (the reference would
// flag itself invalid there).  filter_value = 5 (scan/scan.cpp:73).
void TwoPassScan::_run(const size_t n, Meter &meter) {
  const RunOptions &opts = meter.opts();
  const int filter_value = 5;
  const std::vector<int> src = make_column(n, 42);
  std::vector<int> expected;  // scan/scan.cpp:12-17 expected_out_lt
  std::copy_if(src.begin(), src.end(), std::back_inserter(expected), [&](int x) { return x < filter_value; });
  const unsigned T = host_threads(n);
  std::cout << "Selected device: host (" << T << " thread" << (T == 1 ? "" : "s") << ") for TwoPassScan\n";
  for (size_t it = 0; it < opts.iterations; ++it) {
    std::vector<int> out(n);
    std::vector<size_t> prefix(T + 1, 0);
    auto result = std::make_unique<Result>();
    const auto t0 = clk::now();
    const size_t chunk = n / T;
    auto bounds = [&](unsigned t, size_t *lo, size_t *hi) {
      *lo = t * chunk;
      *hi = t + 1 == T ? n : *lo + chunk;
    };
    parallel_chunks(T, [&](unsigned t) {
      size_t lo, hi, c = 0;
      bounds(t, &lo, &hi);
      for (size_t i = lo; i < hi; ++i) c += src[i] < filter_value;
      prefix[t + 1] = c;
    });
    for (unsigned t = 0; t < T; ++t) prefix[t + 1] += prefix[t];  // (work-item 0's serial prefix, scan.cl:22-30)
    parallel_chunks(T, [&](unsigned t) {
      size_t lo, hi, at = prefix[t];
      bounds(t, &lo, &hi);
      for (size_t i = lo; i < hi; ++i)
        if (src[i] < filter_value) out[at++] = src[i];
    });
    const size_t out_size = prefix[T];
    const auto t1 = clk::now();
    result->host_time = t1 - t0;
    result->kernel_time = result->host_time;  // (no device: the two coincide)
    result->bytes = n * sizeof(int) + out_size * sizeof(int);
    if (out_size != expected.size() || !std::equal(expected.begin(), expected.end(), out.begin())) {
      std::cerr << "incorrect results" << std::endl;
      result->valid = false;
    }
    meter.add_result({{"buf_size", std::to_string(n)}}, std::move(result));
  }
}

// =====================================================================================================
TBBSort::TBBSort() : Dwarf("TBBSort") {}
void TBBSort::init(const RunOptions &opts) {
  require_cpu(opts, "TBBSort");
  meter().set_opts(opts);
  meter().set_params({{"device_type", to_string(opts.device_ty)}});
}
void TBBSort::run(const RunOptions &opts) {
  for (auto size : opts.input_size) _run(size, meter());
}
// sort/tbbsort.cpp:15-48: the column is generated once and sorted in place every iteration, so from the second
// iteration on the input is already sorted — kept, it is what the reference times.  parallel_sort restated: T sorted
// runs, then log2(T) rounds of pairwise std::inplace_merge on parallel threads.
void TBBSort::_run(const size_t n, Meter &meter) {
  const RunOptions &opts = meter.opts();
  std::vector<int> v = make_column(n, 42);
  std::vector<int> expected = v;  // sort/tbbsort.cpp:8-12 expected_out: std::sort
  std::sort(expected.begin(), expected.end());
  unsigned T = host_threads(n);
  while (T & (T - 1)) T &= T - 1;  // a power of two of runs
  std::cout << "Selected device: host (" << T << " thread" << (T == 1 ? "" : "s") << ") for TBBSort\n";
  for (size_t it = 0; it < opts.iterations; ++it) {
    auto result = std::make_unique<Result>();
    const auto t0 = clk::now();
    std::vector<size_t> cut(T + 1);
    for (unsigned t = 0; t <= T; ++t) cut[t] = n / T * t;
    cut[T] = n;
    parallel_chunks(T, [&](unsigned t) { std::sort(v.begin() + cut[t], v.begin() + cut[t + 1]); });
    for (unsigned width = 1; width < T; width *= 2)
      parallel_chunks(T / (2 * width), [&](unsigned p) {
        const unsigned a = p * 2 * width;
        std::inplace_merge(v.begin() + cut[a], v.begin() + cut[a + width], v.begin() + cut[a + 2 * width]);
      });
    const auto t1 = clk::now();
    result->host_time = t1 - t0;
    result->kernel_time = result->host_time;
    result->bytes = 2 * n * sizeof(int);
    if (v != expected) {
      std::cerr << "incorrect results" << std::endl;
      result->valid = false;
    }
    meter.add_result({{"buf_size", std::to_string(n)}}, std::move(result));
  }
}
