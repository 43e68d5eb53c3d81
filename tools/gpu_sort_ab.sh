#!/bin/bash
# development aid (GPU box): A/B of sort library variants on ONE box: timing + per-kernel table for each
mkdir -p gpurun_out; repo="$(pwd)"; export TMPDIR=/tmp
for v in "$@"; do
  lib="$repo/dwarf_bench_amd/_lib/variants/libdbhip_$v.so"; [ "$v" = default ] && lib="$repo/dwarf_bench_amd/_lib/libdbhip.so"
  DBHIP_LIB="$lib" timeout -k 10 120 python tools/ab.py sort 24 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/sort_ab_variants.log
done
for v in "$@"; do
  lib="$repo/dwarf_bench_amd/_lib/variants/libdbhip_$v.so"; [ "$v" = default ] && lib="$repo/dwarf_bench_amd/_lib/libdbhip.so"
  rm -rf "gpurun_out/prof_sort_$v"; cd /tmp
  DBHIP_LIB="$lib" timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/gpurun_out/prof_sort_$v" -o s -- python3 "$repo/tools/ab.py" launch-sort 24 > "$repo/gpurun_out/prof_sort_$v.log" 2>&1
  cd "$repo"; echo "== $v"; python tools/prof_show.py stats "gpurun_out/prof_sort_$v" | grep "rs_" | sort
done
