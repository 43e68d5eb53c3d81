// dwarf_api.hpp — the plugin frame of dwarf_bench, re-created for the MI355X backend.
//
// Same surface as the reference's L2 layer, so a `...Hip` dwarf is a drop-in Dwarf subclass and every
// caller (CLI, DwarfBench::makeMeasurements, tests) is unchanged:
//   RunOptions / GroupByRunOptions / DeviceType parsing+printing   common/options.hpp:6-25, options.cpp:3-33
//   Result / HashJoinResult / GroupByAggResult / MeasureResults     common/result.hpp:9-71, result.cpp:5-93
//   Meter                                                           common/meter.hpp:5-19, meter.cpp:3-16
//   Dwarf                                                           common/dwarf.hpp:6-40
//   Registry                                                        common/registry.hpp:7-25, registry.cpp:3-24
// One addition: RunOptions::DeviceType::HIP ("hip" on the command line, printed as "HIP").
// Behavioural details kept on purpose (tests/test_host_layer.py compares against the reference's own
// compiled result.cpp/options.cpp): CSV schema `device_type,buf_size_bytes,<header>` with
// buf_size_bytes = buf_size * sizeof(int) and times in ms truncated to whole microseconds, append with
// header-if-new; "Kernel duration" printed as kernel_time/1000 but labelled us (result.cpp:9-14);
// unknown device strings parse to Default, which prints as "GPU".
#pragma once
#include <chrono>
#include <iosfwd>
#include <map>
#include <memory>
#include <string>
#include <vector>

// ---- options -----------------------------------------------------------------------------------
struct RunOptions {
  enum DeviceType { CPU, GPU, iGPU, Default, HIP };
  DeviceType device_ty = DeviceType::Default;
  std::vector<size_t> input_size;
  size_t iterations = 1;
  std::string root_path;
  std::string report_path;
  size_t devices = 1;  // additive (no reference counterpart): GPUs a multi-GPU dwarf spreads over, CLI --gpus
};

struct GroupByRunOptions : public RunOptions {
  GroupByRunOptions(const RunOptions &opts, size_t groups, size_t execs)
      : RunOptions(opts), groups_count(groups), executors(execs) {}
  size_t groups_count;
  size_t executors;
};

std::istream &operator>>(std::istream &in, RunOptions::DeviceType &dt);
std::string to_string(const RunOptions::DeviceType &dt);

// ---- results -----------------------------------------------------------------------------------
using DwarfParams = std::map<std::string, std::string>;
using Duration = std::chrono::duration<double, std::micro>;

struct Result {
  size_t thread_x = 1, thread_y = 1, tread_z = 1;
  size_t group_size = 1;
  size_t bytes = 0;
  size_t iterations = 0;
  size_t bytes_per_iteration = 0;
  Duration kernel_time{};
  Duration host_time{};
  bool valid = true;

  virtual ~Result() = default;
  virtual std::vector<Duration> get_reported_timings_list() const;

 protected:
  virtual std::ostream &print_to_stream(std::ostream &os) const;
  friend std::ostream &operator<<(std::ostream &out, const Result &instance);
};

struct HashJoinResult : public Result {
  Duration probe_time{};
  Duration build_time{};
  std::ostream &print_to_stream(std::ostream &os) const override;
};

struct GroupByAggResult : public Result {
  Duration group_by_time{};
  Duration reduction_time{};
  std::vector<Duration> get_reported_timings_list() const override;
  std::ostream &print_to_stream(std::ostream &os) const override;
};

std::ostream &operator<<(std::ostream &os, const Result &res);

struct DwarfRunResult {
  DwarfParams params;
  std::unique_ptr<Result> result;
};

static constexpr auto default_report_header = "host_time_ms,kernel_time_ms";
using SingleRunResults = std::vector<DwarfRunResult>;

class MeasureResults {
 public:
  using const_iterator = SingleRunResults::const_iterator;
  explicit MeasureResults(const std::string &name) : name_(name), header_(default_report_header) {}

  void add_result(DwarfParams params, std::unique_ptr<Result> result);
  const_iterator begin() const { return results_.begin(); }
  const_iterator end() const { return results_.end(); }
  void set_report_header(const std::string &header) { header_ = header; }
  void write_csv(const std::string &filename) const;
  void clear() { results_.clear(); }

 private:
  SingleRunResults results_;
  const std::string name_;
  std::string header_;
};

// ---- meter -------------------------------------------------------------------------------------
class Meter {
 public:
  Meter(const std::string &dwarf_name, MeasureResults &result) : dwarf_name_(dwarf_name), result_(result) {}
  void add_result(DwarfParams &&params, std::unique_ptr<Result> result);
  void set_params(DwarfParams params) { params_ = std::move(params); }
  void set_opts(const RunOptions &opts) { opts_ = &opts; }  // caller keeps opts alive until run() returns
  const RunOptions &opts() const { return *opts_; }

 private:
  const std::string dwarf_name_;
  MeasureResults &result_;
  DwarfParams params_;
  RunOptions const *opts_ = nullptr;
};

// ---- the plugin base class -----------------------------------------------------------------------
class Dwarf {
 public:
  explicit Dwarf(const std::string &name)
      : reporting_header_(default_report_header), name_(name), results_(name), meter_(name, results_) {}
  virtual ~Dwarf() = default;

  const std::string &name() const { return name_; }
  virtual void run(const RunOptions &opts) = 0;
  virtual void init(const RunOptions &opts) = 0;
  void report(const RunOptions &opts);

  Meter &meter() { return meter_; }
  const MeasureResults &get_results() const { return results_; }
  void clear_results() { results_.clear(); }

 protected:
  std::string reporting_header_;

 private:
  std::string name_;
  MeasureResults results_;
  Meter meter_;
};

// ---- registry ------------------------------------------------------------------------------------
class Registry {
 public:
  using const_iterator = std::map<std::string, std::unique_ptr<Dwarf>>::const_iterator;
  static Registry *instance();
  void registerd(Dwarf *dw);  // takes ownership; a second dwarf of the same name is ignored
  Dwarf *find(const std::string &name) const;
  void set_root(const std::string &root) { root_path_ = root; }
  const_iterator begin() const { return dwarfs_.begin(); }
  const_iterator end() const { return dwarfs_.end(); }

 private:
  Registry() = default;
  std::map<std::string, std::unique_ptr<Dwarf>> dwarfs_;
  std::string root_path_;
};

void populate_registry();  // register_dwarfs.cpp:20-56: here it registers the ...Hip dwarfs

namespace helpers {
// $DWARF_BENCH_ROOT or the executable's directory (common/common.cpp:38-41, without Boost.DLL)
std::string get_kernels_root_env(const char *argv0);
}  // namespace helpers
