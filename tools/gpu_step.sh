#!/bin/bash
# development aid (GPU box): group-by + sort tests, group-by A/B (packed table vs two ranges), sort timing and kernel table
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_groupby.py tests/test_gpu_sort.py -x -q > gpurun_out/t_step.log 2>&1
echo "pytest rc=$?" >> gpurun_out/t_step.log; tail -4 gpurun_out/t_step.log
for v in default nopacked; do
  lib="$(pwd)/dwarf_bench_amd/_lib/variants/libdbhip_$v.so"; [ "$v" = default ] && lib="$(pwd)/dwarf_bench_amd/_lib/libdbhip.so"
  DBHIP_LIB="$lib" timeout -k 10 200 python tools/ab.py groupby 2>&1 | grep -v amdgpu.ids
done
bash tools/gpu_sort_ab.sh r02sort default 2>&1 | grep "2^24\|rs_histogram\|rs_plan"
