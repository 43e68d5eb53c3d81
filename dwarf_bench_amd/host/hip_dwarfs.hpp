// hip_dwarfs.hpp — the `...Hip` Dwarf subclasses: C++ host code behind the reference's Dwarf::run() hook
// (common/dwarf.hpp:15-16) that drives the hand-written gfx950 kernels through the C ABI of
// include/dbhip.h.  One class per reference dwarf on the hot path:
//   TwoPassScanHip / DPLScanHip   scan/scan.cpp:22-195, scan/dplscan.cpp:22-92
//   RadixHip                      sort/radix.cpp:17-80
//   GroupByHip                    groupby/groupby.cpp:24-122
//   JoinOmnisciHip                join/join_omnisci.cpp:49-118
//   JoinHip                       join/join.cpp:8-154
#pragma once
#include "dwarf_api.hpp"

#define DBHIP_DECLARE_DWARF(cls)                 \
  class cls : public Dwarf {                     \
   public:                                       \
    cls();                                       \
    void run(const RunOptions &opts) override;   \
    void init(const RunOptions &opts) override;  \
                                                 \
   private:                                      \
    void _run(const size_t buf_size, Meter &meter); \
  }

DBHIP_DECLARE_DWARF(TwoPassScanHip);
DBHIP_DECLARE_DWARF(DPLScanHip);
DBHIP_DECLARE_DWARF(RadixHip);
DBHIP_DECLARE_DWARF(GroupByHip);
DBHIP_DECLARE_DWARF(JoinOmnisciHip);
DBHIP_DECLARE_DWARF(JoinHip);

#undef DBHIP_DECLARE_DWARF
