#!/bin/bash
# development aid (GPU box): per-kernel table only, for several sort library variants on one box
mkdir -p gpurun_out; repo="$(pwd)"; export TMPDIR=/tmp
for v in "$@"; do
  lib="$repo/dwarf_bench_amd/_lib/variants/libdbhip_$v.so"; [ "$v" = default ] && lib="$repo/dwarf_bench_amd/_lib/libdbhip.so"
  rm -rf "gpurun_out/prof_sort_$v"; cd /tmp
  DBHIP_LIB="$lib" timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/gpurun_out/prof_sort_$v" -o s -- python3 "$repo/tools/ab.py" launch-sort 24 > "$repo/gpurun_out/prof_sort_$v.log" 2>&1
  cd "$repo"; echo "== $v"; python tools/prof_show.py stats "gpurun_out/prof_sort_$v" | grep "rs_chunk_scatter\|rs_chunk_hist_kernel<8\|upfront" | sort
done
