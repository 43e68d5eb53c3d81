#!/bin/bash
# development aid (GPU box): join tests, 2^27 join A/B (packed fused histogram vs two plain ones), scan selectivity sweep
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_join.py tests/test_gpu_pjoin.py -x -q > gpurun_out/t_join.log 2>&1
echo "pytest rc=$?" >> gpurun_out/t_join.log; tail -4 gpurun_out/t_join.log
for v in default nofused16; do
  lib="$(pwd)/dwarf_bench_amd/_lib/variants/libdbhip_$v.so"; [ "$v" = default ] && lib="$(pwd)/dwarf_bench_amd/_lib/libdbhip.so"
  DBHIP_LIB="$lib" timeout -k 10 300 python tools/ab.py join 27 2>&1 | grep -v amdgpu.ids
done
timeout -k 10 300 python tools/ab.py scan 2>&1 | grep -v amdgpu.ids
