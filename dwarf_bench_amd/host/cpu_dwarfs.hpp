// cpu_dwarfs.hpp — the host dwarfs registered under the reference's own names (see cpu_dwarfs.cpp):
//   TwoPassScan   scan/scan.cpp:22-195 (BASELINE config 1: --device=cpu --input_size=1024 --iterations=9)
//   TBBSort       sort/tbbsort.cpp:15-48
#pragma once
#include "dwarf_api.hpp"

class TwoPassScan : public Dwarf {
 public:
  TwoPassScan();
  void run(const RunOptions &opts) override;
  void init(const RunOptions &opts) override;

 private:
  void _run(const size_t buf_size, Meter &meter);
};

class TBBSort : public Dwarf {
 public:
  TBBSort();
  void run(const RunOptions &opts) override;
  void init(const RunOptions &opts) override;

 private:
  void _run(const size_t buf_size, Meter &meter);
};
