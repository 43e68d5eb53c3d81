// main.cpp — the `dwarf_bench` CLI (reference: main.cpp:13-101) with the same positional argument and
// flags, parsed by hand (Boost.program_options is not available here):
//   dwarf_bench <Dwarf|list> [--input_size N [N ...]] [--iterations K] [--device cpu|gpu|igpu|hip]
//               [--report_path FILE] [--groups_count G] [--executors E] [--gpus P] [--help]
// Exit codes as in the reference: 1 for an unknown dwarf, 0 otherwise — also after a caught exception.
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "bench.hpp"
#include "dwarf_api.hpp"

namespace {
bool isGroupBy(const std::string &dwarfName) { return dwarfName.find("GroupBy") != std::string::npos; }

const char *kHelp =
    "Dwarf bench:\n"
    "  --help                 Show help message\n"
    "  --dwarf arg            Dwarf to run. List all with 'list' option.\n"
    "  --input_size arg       Data array size, ususally a column size in elements\n"
    "  --iterations arg       Number of iterations to run a bmark.\n"
    "  --device arg           Device to run on.\n"
    "  --report_path arg      Full/Relative path to a report file.\n"
    "  --groups_count arg     Number of unique keys for dwarfs with keys (groupby, hash build etc.).\n"
    "  --executors arg        Number of executors for GroupByLocal.\n"
    "  --gpus arg             Number of GPUs (ranks) for PartitionedJoinHip (default 1).\n";

bool is_flag(const std::string &s) { return s.rfind("--", 0) == 0; }
}  // namespace

int main(int argc, char *argv[]) {
  populate_registry();
  Registry *registry = Registry::instance();

  auto opts = std::make_unique<RunOptions>();
  size_t groups_count = 1, executors = 1;
  opts->root_path = helpers::get_kernels_root_env(argv[0]);
  std::cout << "DWARF_BENCH_ROOT is set to " << opts->root_path << std::endl
            << "You can change that with 'export DWARF_BENCH_ROOT=/your/path'\n";

  std::string dwarf_name;
  bool want_help = false;
  try {
    for (int i = 1; i < argc; ++i) {
      std::string arg = argv[i], value;
      const size_t eq = arg.find('=');
      const bool has_inline = is_flag(arg) && eq != std::string::npos;
      if (has_inline) {
        value = arg.substr(eq + 1);
        arg = arg.substr(0, eq);
      }
      auto next_value = [&]() -> std::string {
        if (has_inline) return value;
        if (i + 1 >= argc) throw std::invalid_argument("the required argument for option '" + arg + "' is missing");
        return argv[++i];
      };
      if (arg == "--help") {
        want_help = true;
      } else if (arg == "--dwarf") {
        dwarf_name = next_value();
      } else if (arg == "--input_size") {  // multitoken: every following non-flag token is a size
        opts->input_size.push_back(std::stoull(next_value()));
        while (i + 1 < argc && !is_flag(argv[i + 1])) opts->input_size.push_back(std::stoull(argv[++i]));
      } else if (arg == "--iterations") {
        opts->iterations = std::stoull(next_value());
      } else if (arg == "--device") {
        std::istringstream in(next_value());
        in >> opts->device_ty;
      } else if (arg == "--report_path") {
        opts->report_path = next_value();
      } else if (arg == "--groups_count") {
        groups_count = std::stoull(next_value());
      } else if (arg == "--executors") {
        executors = std::stoull(next_value());
      } else if (arg == "--gpus") {
        opts->devices = std::stoull(next_value());
      } else if (!is_flag(arg) && dwarf_name.empty()) {
        dwarf_name = arg;  // positional: the dwarf
      } else {
        throw std::invalid_argument("unrecognised option '" + arg + "'");
      }
    }

    if (dwarf_name == "list") {
      std::cout << "Supported dwarfs:\n";
      for (const auto &dw : *registry) std::cout << "\t" << dw.first << std::endl;
      return 0;
    }
    Dwarf *dwarf = registry->find(dwarf_name);
    if (want_help) {
      std::cout << kHelp;
      return 0;
    } else if (!dwarf) {
      std::cerr << "List supported dwarfs to run with '" << argv[0] << " list'" << std::endl;
      return 1;
    }
    if (opts->input_size.empty()) opts->input_size.push_back(1);

    if (isGroupBy(dwarf_name)) opts = std::make_unique<GroupByRunOptions>(*opts, groups_count, executors);

    dwarf->init(*opts);
    dwarf->run(*opts);
    dwarf->report(*opts);
  } catch (std::exception &e) {
    std::cerr << "Caught exception: " << e.what() << std::endl;
  }
  return 0;
}
