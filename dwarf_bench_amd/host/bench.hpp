// bench.hpp — public library API of dwarf_bench (library `dbench`), MI355X edition.
//
// Source-compatible with the reference header (bench.hpp:13-96): the enums, Measurement, RunConfig,
// DwarfBench::makeMeasurements and DwarfBenchException keep their names, member order and meaning, so
// example/bench_usage/main.cpp:4-33 compiles unchanged.  Additions are purely additive:
//   DeviceType::HIP          third device type: the hand-written gfx950 kernels (…Hip dwarfs)
//   RunConfig::groups_count  / executors: defaulted trailing members (reference hard-codes 20 / 1024,
//                            bench.cpp:80), so designated / aggregate initialisers written for the
//                            reference still compile
// On a machine without the reference's SYCL runtimes CPU and GPU requests are served by the HIP
// dwarfs as well when DWARF_BENCH_HIP_FOR_ALL=1 is set; otherwise they raise DwarfBenchException
// (the reference asserts on an unknown dwarf, bench.cpp:84).
#pragma once

#include <cstddef>
#include <stdexcept>
#include <string>
#include <vector>

namespace DwarfBench {

enum Dwarf {
  Scan,
  Join,
  GroupBy,
  Sort,
};

enum DeviceType { CPU, GPU, HIP };

// dataSize: elements of the input column (the reference's "todo make bytes counting", bench.cpp:96)
// microseconds: host wall time of one iteration
struct Measurement {
  size_t dataSize;
  size_t microseconds;
};

struct RunConfig {
  DeviceType device;
  size_t inputSize;
  size_t iterations;
  Dwarf dwarf;
  size_t groups_count = 20;  // GroupBy only
  size_t executors = 1024;   // GroupByLocal-style dwarfs only
  size_t devices = 1;        // GPUs for multi-GPU dwarfs (PartitionedJoinHip)
};

class DwarfBench {
 public:
  DwarfBench() = default;
  // One Measurement per iteration, in order.
  std::vector<Measurement> makeMeasurements(const RunConfig &conf);

 private:
  enum DwarfImpl { DPLScan, GroupBy, Join, Radix, JoinOmnisci };
  DwarfImpl dwarfToImpl(Dwarf dwarf);
  std::string dwarfToString(DwarfImpl dwarf, DeviceType device);
};

class DwarfBenchException : public std::exception {
 public:
  explicit DwarfBenchException(const std::string &message);
  const char *what() const noexcept override;

 private:
  std::string message_;
};

}  // namespace DwarfBench
