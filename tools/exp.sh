#!/bin/bash
# scratch experiment driver for the GPU box (round 4): edited per experiment, output under gpurun_out/
set -uo pipefail
repo="$(pwd)"; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/exp.log; : > $out
echo "== correctness of the tile shapes (join tests)" >> $out
for t in "0 0" "1 1" "2 2"; do set -- $t
  echo "-- T0=$1 T1=$2" >> $out
  DBHIP_JL_T0=$1 DBHIP_JL_T1=$2 timeout -k 10 600 python -m pytest tests/test_gpu_join.py -x -q -m gpu 2>&1 | tail -3 >> $out || exit 1
done
echo "== tile shapes at 2^30" >> $out
for t in "0 0" "0 1" "0 2" "1 0" "2 0" "1 1" "2 2" "1 2" "2 1"; do set -- $t
  echo "-- T0=$1 T1=$2" >> $out
  DBHIP_JL_T0=$1 DBHIP_JL_T1=$2 timeout -k 10 300 python tools/ab.py radix 30 2>&1 | grep -v amdgpu.ids >> $out
done
echo "-- T0=0 T1=0 SC1_XCD=0" >> $out
DBHIP_JL_SC1_XCD=0 timeout -k 10 300 python tools/ab.py radix 30 2>&1 | grep -v amdgpu.ids >> $out
echo "== tile shapes at 2^26 / 2^27 / 2^28" >> $out
for lg in 26 27 28; do for t in "0 0" "1 1" "2 2" "0 2"; do set -- $t
  echo "-- lg=$lg T0=$1 T1=$2" >> $out
  DBHIP_JL_T0=$1 DBHIP_JL_T1=$2 timeout -k 10 300 python tools/ab.py radix $lg 2>&1 | grep -v amdgpu.ids >> $out
done; 
echo "-- lg=$lg T0=0 T1=0 SC1_XCD=0" >> $out
DBHIP_JL_SC1_XCD=0 timeout -k 10 300 python tools/ab.py radix $lg 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
