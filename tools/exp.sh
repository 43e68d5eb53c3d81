#!/bin/bash
set -uo pipefail
repo="$(pwd)"; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/exp.log; : > $out
timeout -k 10 900 python -m pytest tests/test_gpu_pjoin.py -x -q -m gpu 2>&1 | tail -3 >> $out
for h in 1 2 4; do
  echo "== DWARF_BENCH_PJOIN_SUBJOINS=$h, 8 virtual ranks, 2^30" >> $out
  DWARF_BENCH_PJOIN_SUBJOINS=$h DWARF_BENCH_VALIDATE_MAX=1 timeout -k 10 300 dwarf_bench_amd/_lib/dwarf_bench PartitionedJoinHip --device=hip --gpus 8 --iterations 2 --input_size 1073741824 2>&1 | grep -v "DWARF_BENCH_ROOT\|You can" | tail -9 >> $out
done
for h in 1 2; do
  echo "== DWARF_BENCH_PJOIN_SUBJOINS=$h, RCCL self exchange, 2^28" >> $out
  DWARF_BENCH_PJOIN_SUBJOINS=$h DWARF_BENCH_VALIDATE_MAX=1 timeout -k 10 300 dwarf_bench_amd/_lib/dwarf_bench PartitionedJoinHip --device=hip --gpus 1 --iterations 2 --input_size 268435456 2>&1 | grep -v "DWARF_BENCH_ROOT\|You can\|version\|Hostname\|Librccl" | tail -8 >> $out
done
echo "== small sizes with host validation, 2/3/4/8 ranks, sub-joins 1 2 4" >> $out
for h in 1 2 4; do for g in 2 3 8; do
  DWARF_BENCH_PJOIN_SUBJOINS=$h timeout -k 10 300 dwarf_bench_amd/_lib/dwarf_bench PartitionedJoinHip --device=hip --gpus $g --iterations 1 --input_size 1000003 2>&1 | grep -i "incorrect\|exception\|PartitionedJoinHip:" >> $out
done; done
cat $out
