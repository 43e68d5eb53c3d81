// host_selftest.cpp — test driver for the C++ host layer (dwarf_bench_amd/host/dwarf_api.*), CPU only.
//   host_selftest csv <path> <dwarf> <device_type> <header|-> <runs> {kind buf_size host_us kernel_us t2_us t3_us}...
//       builds Results through a Meter exactly like a dwarf would, prints them, writes the CSV
//   host_selftest devtype <string>     prints "<enum value> <to_string>"
//   host_selftest registry             registry semantics: ownership, duplicate names, lookup
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>

#include "dwarf_api.hpp"

namespace {
class Stub : public Dwarf {
 public:
  explicit Stub(const std::string &n) : Dwarf(n) {}
  void init(const RunOptions &opts) override {
    meter().set_opts(opts);
    meter().set_params({{"device_type", to_string(opts.device_ty)}});
  }
  void run(const RunOptions &) override {}
  void set_header(const std::string &h) { reporting_header_ = h; }
};
}  // namespace

int main(int argc, char **argv) {
  if (argc >= 3 && !std::strcmp(argv[1], "devtype")) {
    std::istringstream in(argv[2]);
    RunOptions::DeviceType dt;
    in >> dt;
    std::cout << static_cast<int>(dt) << " " << to_string(dt) << "\n";
    return 0;
  }
  if (argc >= 2 && !std::strcmp(argv[1], "registry")) {
    Registry *r = Registry::instance();
    r->registerd(new Stub("A"));
    r->registerd(new Stub("B"));
    Dwarf *first_a = r->find("A");
    r->registerd(new Stub("A"));  // duplicate: ignored (and freed)
    std::cout << (r->find("A") == first_a) << (r->find("B") != nullptr) << (r->find("C") == nullptr);
    int n = 0;
    for (auto it = r->begin(); it != r->end(); ++it) ++n;
    std::cout << " " << n << "\n";
    return 0;
  }
  if (argc >= 7 && !std::strcmp(argv[1], "csv")) {
    Stub d(argv[3]);
    RunOptions opts;
    std::istringstream dev(argv[4]);
    dev >> opts.device_ty;
    opts.report_path = argv[2];
    d.init(opts);
    if (std::strcmp(argv[5], "-")) d.set_header(argv[5]);
    const int runs = std::atoi(argv[6]);
    for (int i = 0; i < runs; ++i) {
      char **a = argv + 7 + 6 * i;
      const int kind = std::atoi(a[0]);
      std::unique_ptr<Result> r;
      if (kind == 1) {
        auto h = std::make_unique<HashJoinResult>();
        h->build_time = Duration(std::atof(a[4]));
        h->probe_time = Duration(std::atof(a[5]));
        r = std::move(h);
      } else if (kind == 2) {
        auto g = std::make_unique<GroupByAggResult>();
        g->group_by_time = Duration(std::atof(a[4]));
        g->reduction_time = Duration(std::atof(a[5]));
        r = std::move(g);
      } else {
        r = std::make_unique<Result>();
      }
      r->host_time = Duration(std::atof(a[2]));
      r->kernel_time = Duration(std::atof(a[3]));
      d.meter().add_result({{"buf_size", a[1]}}, std::move(r));
    }
    RunOptions to_stdout = opts;
    to_stdout.report_path = "";
    d.report(to_stdout);  // prints every result
    d.report(opts);       // appends to the CSV
    return 0;
  }
  std::cerr << "usage: see the header comment\n";
  return 2;
}
