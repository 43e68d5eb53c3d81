// bench.cpp — DwarfBench::makeMeasurements for the MI355X backend (reference: bench.cpp:12-132).
// Same flow: populate the registry once, translate RunConfig into RunOptions wrapped in
// GroupByRunOptions for every dwarf (bench.cpp:80), look the dwarf up by the name the device suffix
// rule yields (CPU -> base name, GPU -> ...Cuda, HIP -> ...Hip), clear_results/init/run, and convert
// every iteration into a Measurement{elements, host microseconds}.
#include "bench.hpp"

#include <cstdlib>
#include <string>

#include "dwarf_api.hpp"

namespace DwarfBench {

namespace {
const char *base_name(int impl) {
  static const char *names[] = {"DPLScan", "GroupBy", "Join", "Radix", "JoinOmnisci"};
  return names[impl];
}
bool hip_serves_all() {
  const char *e = std::getenv("DWARF_BENCH_HIP_FOR_ALL");
  return e && *e && *e != '0';
}
}  // namespace

std::string DwarfBench::dwarfToString(DwarfImpl dwarf, DeviceType device) {
  std::string name = base_name(dwarf);
  switch (device) {
    case DeviceType::CPU: return name;
    case DeviceType::GPU: return name + "Cuda";
    case DeviceType::HIP: return name + "Hip";
  }
  return "Unknown Dwarf";
}

DwarfBench::DwarfImpl DwarfBench::dwarfToImpl(Dwarf dwarf) {
  switch (dwarf) {  // bench.cpp:107-120
    case Dwarf::Sort: return DwarfImpl::Radix;
    case Dwarf::Join: return DwarfImpl::JoinOmnisci;
    case Dwarf::GroupBy: return DwarfImpl::GroupBy;
    case Dwarf::Scan: return DwarfImpl::DPLScan;
  }
  throw DwarfBenchException("unknown Dwarf enumerator");
}

std::vector<Measurement> DwarfBench::makeMeasurements(const RunConfig &conf) {
  static Registry *reg = [] {
    populate_registry();
    return Registry::instance();
  }();

  RunOptions base;
  base.device_ty = conf.device == DeviceType::CPU   ? RunOptions::DeviceType::CPU
                   : conf.device == DeviceType::GPU ? RunOptions::DeviceType::GPU
                                                    : RunOptions::DeviceType::HIP;
  base.input_size = {conf.inputSize};
  base.iterations = conf.iterations;
  base.report_path = "";
  base.devices = conf.devices;
  GroupByRunOptions opts(base, conf.groups_count, conf.executors);

  const DwarfImpl impl = dwarfToImpl(conf.dwarf);
  std::string name = dwarfToString(impl, conf.device);
  ::Dwarf *dwarf = reg->find(name);
  if (!dwarf && hip_serves_all()) dwarf = reg->find(name = dwarfToString(impl, DeviceType::HIP));
  if (!dwarf)
    throw DwarfBenchException("dwarf '" + name + "' is not registered in this build (only the ...Hip dwarfs are; "
                              "set DWARF_BENCH_HIP_FOR_ALL=1 to serve CPU/GPU requests with them)");

  dwarf->clear_results();
  dwarf->init(opts);
  dwarf->run(opts);

  std::vector<Measurement> ms;
  for (const DwarfRunResult &res : dwarf->get_results())
    ms.push_back(Measurement{static_cast<size_t>(std::stoull(res.params.at("buf_size"))),
                             static_cast<size_t>(res.result->host_time.count())});
  return ms;
}

DwarfBenchException::DwarfBenchException(const std::string &message) : message_(message) {}
const char *DwarfBenchException::what() const noexcept { return message_.c_str(); }

}  // namespace DwarfBench
