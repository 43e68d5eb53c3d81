// dwarf_api.cpp — implementation of the plugin frame (see dwarf_api.hpp for the reference file:line map).
#include "dwarf_api.hpp"

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>

#include <unistd.h>

// ---- options -----------------------------------------------------------------------------------
std::istream &operator>>(std::istream &in, RunOptions::DeviceType &dt) {
  std::string word;
  in >> word;
  for (char &ch : word) ch = static_cast<char>(std::tolower(static_cast<unsigned char>(ch)));
  static const std::map<std::string, RunOptions::DeviceType> known = {
      {"cpu", RunOptions::CPU}, {"gpu", RunOptions::GPU}, {"igpu", RunOptions::iGPU}, {"hip", RunOptions::HIP}};
  const auto it = known.find(word);
  dt = it == known.end() ? RunOptions::Default : it->second;  // anything else: Default, as the reference
  return in;
}

std::string to_string(const RunOptions::DeviceType &dt) {
  switch (dt) {
    case RunOptions::CPU: return "CPU";
    case RunOptions::iGPU: return "iGPU";
    case RunOptions::HIP: return "HIP";
    case RunOptions::GPU:
    case RunOptions::Default: return "GPU";
  }
  throw std::logic_error("Unsupported device type!");
}

// ---- results -----------------------------------------------------------------------------------
std::ostream &operator<<(std::ostream &os, const Result &res) { return res.print_to_stream(os); }

std::ostream &Result::print_to_stream(std::ostream &os) const {
  // the reference divides the kernel time by 1000 and still labels it "us" (common/result.cpp:9-14)
  os << "Kernel duration: " << kernel_time.count() / 1000.0 << " us\n";
  os << "Host duration:   " << host_time.count() << " us\n";
  return os;
}
std::vector<Duration> Result::get_reported_timings_list() const { return {host_time, kernel_time}; }

std::ostream &HashJoinResult::print_to_stream(std::ostream &os) const {
  Result::print_to_stream(os);
  os << "Build time: " << build_time.count() << " us\n";
  os << "Probe time: " << probe_time.count() << " us\n";
  return os;
}

std::ostream &GroupByAggResult::print_to_stream(std::ostream &os) const {
  Result::print_to_stream(os);
  os << "Group stage time: " << group_by_time.count() << " us\n";
  os << "Reduce stage time: " << reduction_time.count() << " us\n";
  return os;
}
std::vector<Duration> GroupByAggResult::get_reported_timings_list() const {
  return {host_time, group_by_time, reduction_time};
}

void MeasureResults::add_result(DwarfParams params, std::unique_ptr<Result> result) {
  results_.push_back(DwarfRunResult{std::move(params), std::move(result)});
}

void MeasureResults::write_csv(const std::string &filename) const {
  const bool had_file = std::ifstream(filename).good();
  std::ofstream csv(filename, std::ios::app);
  if (!csv.is_open()) throw std::runtime_error("Could not open the file at " + filename);
  if (!had_file) csv << "device_type,buf_size_bytes," << header_ << "\n";
  for (const DwarfRunResult &run : results_) {
    const size_t bytes = static_cast<size_t>(std::stoll(run.params.at("buf_size"))) * sizeof(int);
    csv << run.params.at("device_type") << "," << bytes << ",";
    bool first = true;
    for (const Duration &d : run.result->get_reported_timings_list()) {
      // milliseconds with whole-microsecond resolution, default ostream formatting
      const auto us = std::chrono::duration_cast<std::chrono::microseconds>(d).count();
      csv << (first ? "" : ",") << us / 1000.0;
      first = false;
    }
    csv << "\n";
  }
}

// ---- meter / dwarf / registry ----------------------------------------------------------------------
void Meter::add_result(DwarfParams &&params, std::unique_ptr<Result> result) {
  DwarfParams merged = params_;  // stable params (device_type) first, per-run params do not override them
  merged.insert(params.begin(), params.end());
  result_.add_result(std::move(merged), std::move(result));
}

void Dwarf::report(const RunOptions &opts) {
  if (opts.report_path.empty()) {
    for (const DwarfRunResult &run : results_) std::cout << *run.result;
    return;
  }
  results_.set_report_header(reporting_header_);
  results_.write_csv(opts.report_path);
}

Registry *Registry::instance() {
  static std::unique_ptr<Registry> self(new Registry());
  return self.get();
}
void Registry::registerd(Dwarf *dw) {
  std::unique_ptr<Dwarf> owned(dw);
  dwarfs_.emplace(owned->name(), std::move(owned));  // duplicate name: the newcomer is dropped
}
Dwarf *Registry::find(const std::string &name) const {
  const auto it = dwarfs_.find(name);
  return it == dwarfs_.end() ? nullptr : it->second.get();
}

namespace helpers {
std::string get_kernels_root_env(const char *argv0) {
  if (const char *root = std::getenv("DWARF_BENCH_ROOT")) return root;
  char buf[4096];
  const ssize_t len = ::readlink("/proc/self/exe", buf, sizeof(buf) - 1);
  std::string exe = len > 0 ? std::string(buf, static_cast<size_t>(len)) : std::string(argv0 ? argv0 : ".");
  const size_t slash = exe.find_last_of('/');
  return slash == std::string::npos ? "." : exe.substr(0, slash);
}
}  // namespace helpers
