#!/bin/bash
set -uo pipefail
repo="$(pwd)"; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/exp.log; : > $out
DBHIP_GB_DYNAMIC=1 timeout -k 10 900 python -m pytest tests/test_gpu_groupby.py -x -q -m gpu 2>&1 | tail -3 >> $out
for r in 1 2 3; do
echo "-- static" >> $out
timeout -k 10 300 python tools/ab.py groupby 2>&1 | grep -v amdgpu.ids >> $out
echo "-- DBHIP_GB_DYNAMIC=1" >> $out
DBHIP_GB_DYNAMIC=1 timeout -k 10 300 python tools/ab.py groupby 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
