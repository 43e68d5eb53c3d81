// register_dwarfs.cpp — populate_registry() (reference: register_dwarfs.cpp:20-56).  The reference
// registers its dwarfs under EXPERIMENTAL / DPCPP_ENABLED / CUDA_ENABLED guards; this build has one
// guard, HIP_ENABLED, and registers the hand-written gfx950 dwarfs — plus, unguarded, the two HOST dwarfs of the hot
// path under the reference's own names (cpu_dwarfs.cpp: TwoPassScan for --device=cpu, TBBSort).
#include "cpu_dwarfs.hpp"
#include "dwarf_api.hpp"
#include "hip_dwarfs.hpp"

void populate_registry() {
  Registry *registry = Registry::instance();
  registry->registerd(new TwoPassScan());  // scan/scan.cpp:22-195 (register_dwarfs.cpp:24 in the reference)
  registry->registerd(new TBBSort());      // sort/tbbsort.cpp:15-48
#ifdef HIP_ENABLED
  registry->registerd(new TwoPassScanHip());
  registry->registerd(new DPLScanHip());
  registry->registerd(new RadixHip());
  registry->registerd(new GroupByHip());
  registry->registerd(new JoinOmnisciHip());
  registry->registerd(new JoinHip());
  registry->registerd(new PartitionedJoinHip());
  registry->registerd(new GroupByLocalHip());
  registry->registerd(new HashBuildHip());
  registry->registerd(new HashBuildNonBitmaskHip());
  registry->registerd(new ProbeHip());
  registry->registerd(new ReduceHip());
  registry->registerd(new NestedLoopJoinHip());
#endif
}
