#!/bin/bash
# development aid (GPU box, from the repo root): same-box A/B of library builds.  A variant is a tag of
# dwarf_bench_amd/_lib/variants/libdbhip_<tag>.so (tools/build_variant.sh) or `default` for the shipped library.
#   tools/gpu_ab.sh time <ab.py mode> [log2 size] -- <variant>...      tools/ab.py timings, one line per variant, two rounds
#   tools/gpu_ab.sh kernels <ab.py launch mode> [log2 size] -- <variant>...
#                                                                      rocprofv3 --kernel-trace --stats per variant, dbhip kernels
#   tools/gpu_ab.sh counters <ab.py mode> -- <variant>...              SQ counters (tools/pmc_kernel_counters.sh) per variant
# This is how the variant comparisons quoted in DESIGN.md and in the kernels' comments were taken.
set -uo pipefail
what="$1"; shift
args=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done
shift || true
repo="$(pwd)"; mkdir -p gpurun_out; export TMPDIR=/tmp
libof() { [ "$1" = default ] && echo "$repo/dwarf_bench_amd/_lib/libdbhip.so" || echo "$repo/dwarf_bench_amd/_lib/variants/libdbhip_$1.so"; }
case "$what" in
  time)
    for round in 1 2; do for v in "$@"; do
      DBHIP_LIB="$(libof "$v")" timeout -k 10 300 python tools/ab.py "${args[@]}" 2>&1 | grep -v amdgpu.ids
    done; done ;;
  kernels)
    for v in "$@"; do
      rm -rf "gpurun_out/prof_$v"; cd /tmp
      DBHIP_LIB="$(libof "$v")" timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/gpurun_out/prof_$v" -o s \
        -- python3 "$repo/tools/ab.py" "${args[@]}" > "$repo/gpurun_out/prof_$v.log" 2>&1
      cd "$repo"; echo "== $v"; python tools/prof_show.py stats "gpurun_out/prof_$v" | grep -v "elementwise\|fill_kernel\|gen_uniform\|rocclr" | sort
    done ;;
  counters)
    for v in "$@"; do
      rm -rf "gpurun_out/pmc_${v}1" "gpurun_out/pmc_${v}2"
      DBHIP_LIB="$(libof "$v")" bash tools/pmc_kernel_counters.sh "${args[0]}" "$v"
      echo "== $v"; python tools/prof_show.py counters "$v"
    done ;;
  *) echo "usage: see the header of $0" >&2; exit 2 ;;
esac
