#!/bin/bash
# development aid (GPU box): tools/ab.py join <lg> for several library variants
lg="$1"; shift
for v in "$@"; do
  lib="$(pwd)/dwarf_bench_amd/_lib/variants/libdbhip_$v.so"; [ "$v" = default ] && lib="$(pwd)/dwarf_bench_amd/_lib/libdbhip.so"
  DBHIP_LIB="$lib" timeout -k 10 300 python tools/ab.py join "$lg" 2>&1 | grep -v amdgpu.ids
done
