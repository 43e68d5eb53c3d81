#!/bin/bash
set -uo pipefail
repo="$(pwd)"; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/exp.log; : > $out
for w in 2 4 6 8 16; do for v in default prev; do
  lib=$repo/dwarf_bench_amd/_lib/libdbhip.so; [ $v = prev ] && lib=$repo/dwarf_bench_amd/_lib/variants/libdbhip_prev.so
  echo "-- PROBE_WGS=$w $v" >> $out
  DBHIP_LIB=$lib DBHIP_JL_PROBE_WGS=$w timeout -k 10 300 python tools/ab.py join 26 2>&1 | grep -v amdgpu.ids | sed 's/| radix.*//' >> $out
done; done
cat $out
