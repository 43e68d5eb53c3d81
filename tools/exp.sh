#!/bin/bash
set -uo pipefail
repo="$(pwd)"; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/exp_piece2.log; : > $out
cli=dwarf_bench_amd/_lib/dwarf_bench
for piece in 268435457 402653184; do
  echo "== DWARF_BENCH_PJOIN_PIECE=$piece (one rank, all 2^30 pairs of both relations through its own ncclSend/ncclRecv group)" >> $out
  DWARF_BENCH_PJOIN_PIECE=$piece DWARF_BENCH_VALIDATE_MAX=1 timeout -k 10 300 $cli PartitionedJoinHip --device=hip --gpus 1 --iterations 2 --input_size 1073741824 >> $out 2>&1
  echo "exit code $?" >> $out
done
cat $out
