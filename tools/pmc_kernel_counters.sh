#!/bin/bash
# development aid: SQ counters (two passes of 8) for the kernels one small driver script launches
#   tools/pmc_kernel_counters.sh sort-only sort      (any mode of tools/ab.py)
set -euo pipefail
mode="$1"; tag="$2"; repo="$(pwd)"; script="$repo/tools/ab.py"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU \
  --output-format csv -d "$repo/gpurun_out/pmc_${tag}1" -o p -- python3 "$script" "$mode" > "$repo/gpurun_out/pmc_${tag}1.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM \
  --output-format csv -d "$repo/gpurun_out/pmc_${tag}2" -o p -- python3 "$script" "$mode" > "$repo/gpurun_out/pmc_${tag}2.log" 2>&1
echo "[pmc] done $tag"
