// hip_library_usage.cpp — the `dbench` library API with the HIP device: every dwarf of the public enum measured
// through DwarfBench::makeMeasurements on DeviceType::HIP and reported as one line per iteration plus a
// drop-the-slowest mean per dwarf (the statistic of the reference's report notebook).
//
//   hip_library_usage [rows] [iterations]        defaults: 1024 rows, 10 iterations
//
// The reference's own sample (example/bench_usage/main.cpp in kurapov-peter/dwarf_bench) also compiles unchanged
// against this build's <bench.hpp>; tests/test_host_layer.py checks that.  This file is the HIP-side counterpart.
#include <bench.hpp>

#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <numeric>
#include <utility>

int main(int argc, char **argv) {
  const size_t rows = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1024;
  const size_t iterations = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 10;
  const std::pair<DwarfBench::Dwarf, const char *> dwarfs[] = {{DwarfBench::Dwarf::Join, "Join"},
                                                               {DwarfBench::Dwarf::Sort, "Sort"},
                                                               {DwarfBench::Dwarf::Scan, "Scan"},
                                                               {DwarfBench::Dwarf::GroupBy, "GroupBy"}};
  DwarfBench::DwarfBench bench;
  for (const auto &[dwarf, label] : dwarfs) {
    DwarfBench::RunConfig config{};
    config.device = DwarfBench::DeviceType::HIP;
    config.inputSize = rows;
    config.iterations = iterations;
    config.dwarf = dwarf;
    const std::vector<DwarfBench::Measurement> ms = bench.makeMeasurements(config);
    std::vector<size_t> us;
    for (const DwarfBench::Measurement &m : ms) {
      std::cout << dwarf << ' ' << config.device << " RESULT: " << m.dataSize << ' ' << m.microseconds << std::endl;
      us.push_back(m.microseconds);
    }
    std::sort(us.begin(), us.end());
    if (us.size() > 1) us.pop_back();
    const double mean = us.empty() ? 0.0 : std::accumulate(us.begin(), us.end(), 0.0) / us.size();
    std::cout << label << " on HIP: " << rows << " rows, mean of " << us.size() << " iterations " << mean << " us"
              << std::endl;
  }
  return 0;
}
