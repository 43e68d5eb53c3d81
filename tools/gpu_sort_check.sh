#!/bin/bash
# development aid (GPU box): sort tests, sort timing, per-kernel times of the sort — one call, outputs under gpurun_out/
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_sort.py tests/test_gpu_graph.py -x -q > gpurun_out/t_sort.log 2>&1
echo "pytest rc=$?" >> gpurun_out/t_sort.log; tail -4 gpurun_out/t_sort.log
timeout -k 10 120 python tools/ab.py sort 24 2>&1 | tee gpurun_out/sort_ab.log
repo="$(pwd)"; cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/gpurun_out/prof_sort" -o s -- python3 "$repo/tools/ab.py" launch-sort 24 > "$repo/gpurun_out/prof_sort.log" 2>&1
cd "$repo"; python tools/prof_show.py stats gpurun_out/prof_sort | grep -v "elementwise\|fill_kernel\|gen_uniform" | head -30
