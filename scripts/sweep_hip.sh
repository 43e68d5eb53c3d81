#!/bin/bash
# Size sweeps of the HIP dwarfs through the dwarf_bench CLI, in the shape of the reference's
# scripts/benchmark_*.sh (same size ladders 256..65536 and 25600..134217728, --iterations=9, one CSV per
# dwarf appended across runs) but for --device=hip.  Reduce the CSVs with scripts/report.py.
#
#   scripts/sweep_hip.sh [small|large|all] [out_dir]        (default: all, ./reports)
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
bin="${DWARF_BENCH_BIN:-$here/../dwarf_bench_amd/_lib/dwarf_bench}"
which="${1:-all}"
out="${2:-reports}"
mkdir -p "$out"

small="256 512 1024 2048 4096 8192 16384 32768 65536"
large="25600 262144 524288 1048576 2097152 4194304 8388608 16777216 33554432 67108864 134217728"
case "$which" in
  small) sizes="$small" ;;
  large) sizes="$large" ;;
  all) sizes="$small $large" ;;
  *) echo "usage: $0 [small|large|all] [out_dir]" >&2; exit 2 ;;
esac

run() {  # run <dwarf> <csv> [extra args...]
  local dwarf="$1" csv="$2"; shift 2
  # shellcheck disable=SC2086
  "$bin" "$dwarf" --device=hip --iterations=9 --report_path="$out/$csv" "$@" --input_size $sizes > "$out/${csv%.csv}.log"
  echo "$dwarf -> $out/$csv"
}

run TwoPassScanHip report_scan_hip.csv
run DPLScanHip report_dpl_scan_hip.csv
run RadixHip report_radix_hip.csv
run GroupByHip report_groupby_hip.csv --groups_count 65536
run GroupByLocalHip report_groupby_local_hip.csv --groups_count 65536 --executors 1024
run JoinOmnisciHip report_join_omnisci_hip.csv
run JoinHip report_join_hip.csv
run HashBuildHip report_hash_build_hip.csv
run HashBuildNonBitmaskHip report_hash_build_non_bitmask_hip.csv
