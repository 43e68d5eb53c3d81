#!/bin/bash
set -uo pipefail
repo="$(pwd)"; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/exp.log; : > $out
timeout -k 10 900 python -m pytest tests/test_gpu_join.py tests/test_gpu_pjoin.py -x -q -m gpu 2>&1 | tail -2 >> $out
for lg in 24 26 27 30; do
  timeout -k 10 300 python tools/ab.py radix $lg 2>&1 | grep -v amdgpu.ids >> $out
done
timeout -k 10 300 python tools/ab.py join 26 2>&1 | grep -v amdgpu.ids >> $out
rm -rf gpurun_out/prof_p26; cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/gpurun_out/prof_p26" -o s \
  -- python3 "$repo/tools/ab.py" radix 26 > "$repo/gpurun_out/prof_p26.log" 2>&1
cd "$repo"; python tools/prof_show.py stats gpurun_out/prof_p26 | grep jl_ | sort >> $out
cat $out
