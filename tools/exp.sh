#!/bin/bash
set -uo pipefail
repo="$(pwd)"; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/exp.log; : > $out
timeout -k 10 900 python -m pytest tests/test_gpu_join.py tests/test_gpu_pjoin.py tests/test_gpu_graph.py -x -q -m gpu 2>&1 | tail -5 >> $out
for lg in 24 26 27; do
timeout -k 10 300 python tools/ab.py join $lg 2>&1 | grep -v amdgpu.ids >> $out
done
timeout -k 10 300 python tools/ab.py radix 30 2>&1 | grep -v amdgpu.ids >> $out
cat $out
