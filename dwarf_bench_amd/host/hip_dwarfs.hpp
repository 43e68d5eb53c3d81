// hip_dwarfs.hpp — the `...Hip` Dwarf subclasses: C++ host code behind the reference's Dwarf::run() hook
// (common/dwarf.hpp:15-16) that drives the hand-written gfx950 kernels through the C ABI of
// include/dbhip.h.  One class per reference dwarf on the hot path:
//   TwoPassScanHip / DPLScanHip   scan/scan.cpp:22-195, scan/dplscan.cpp:22-92
//   RadixHip                      sort/radix.cpp:17-80
//   GroupByHip                    groupby/groupby.cpp:24-122
//   JoinOmnisciHip                join/join_omnisci.cpp:49-118
//   JoinHip                       join/join.cpp:8-154
//   GroupByLocalHip, HashBuildHip, HashBuildNonBitmaskHip, ProbeHip: the "next" rows of SURVEY 8(f)
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
DBHIP_DECLARE_DWARF(PartitionedJoinHip);      // radix-partitioned join over --gpus ranks, RCCL exchange
// SURVEY 8(f) "next" rows
DBHIP_DECLARE_DWARF(GroupByLocalHip);         // groupby/groupby_local.cpp:24-142 (two-phase timings, --executors)
DBHIP_DECLARE_DWARF(HashBuildHip);            // hash/hash_build.cpp:8-98 (bitmask-claimed table, build only)
DBHIP_DECLARE_DWARF(HashBuildNonBitmaskHip);  // hash/hash_build_non_bitmask.cpp:7-91 (CAS table, build only)
DBHIP_DECLARE_DWARF(ProbeHip);                // probe/slab_probe.cpp:9-107 (table built untimed, lookups timed)
DBHIP_DECLARE_DWARF(ReduceHip);               // reduce/reduce.cpp:27-98 (int sum)
DBHIP_DECLARE_DWARF(NestedLoopJoinHip);       // join/nested_join.cpp:10-110 (dense cell matrix, small n)

#undef DBHIP_DECLARE_DWARF
