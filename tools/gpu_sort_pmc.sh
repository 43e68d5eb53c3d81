#!/bin/bash
# development aid (GPU box): SQ counters of the sort kernels for library variants (tools/pmc_kernel_counters.sh per variant)
for v in "$@"; do
  lib="$(pwd)/dwarf_bench_amd/_lib/variants/libdbhip_$v.so"; [ "$v" = default ] && lib="$(pwd)/dwarf_bench_amd/_lib/libdbhip.so"
  rm -rf gpurun_out/pmc_${v}1 gpurun_out/pmc_${v}2
  DBHIP_LIB="$lib" SORT_BITS=8 bash tools/pmc_kernel_counters.sh sort-only "$v"
  echo "== $v"; python tools/prof_show.py counters "$v" rs_chunk_scatter
done
