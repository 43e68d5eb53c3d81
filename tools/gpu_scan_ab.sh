#!/bin/bash
# development aid (GPU box): tools/ab.py scan for several library variants
for v in "$@"; do
  lib="$(pwd)/dwarf_bench_amd/_lib/variants/libdbhip_$v.so"; [ "$v" = default ] && lib="$(pwd)/dwarf_bench_amd/_lib/libdbhip.so"
  DBHIP_LIB="$lib" timeout -k 10 300 python tools/ab.py scan 2>&1 | grep -v amdgpu.ids | grep "s=0.0004\|s=0.0100\|s=0.0500\|s=0.1000\|s=0.2500\|s=0.5000\|s=1.0000"
done
