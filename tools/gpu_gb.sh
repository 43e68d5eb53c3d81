#!/bin/bash
# development aid (GPU box): group-by tests and A/B of library variants (tools/ab.py groupby), each variant twice
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_groupby.py -x -q 2>&1 | tail -3
for round in 1 2; do for v in "$@"; do
  lib="$(pwd)/dwarf_bench_amd/_lib/variants/libdbhip_$v.so"; [ "$v" = default ] && lib="$(pwd)/dwarf_bench_amd/_lib/libdbhip.so"
  DBHIP_LIB="$lib" timeout -k 10 200 python tools/ab.py groupby 2>&1 | grep -v amdgpu.ids
done; done
