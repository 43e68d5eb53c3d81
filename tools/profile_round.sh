#!/bin/bash
# Collect the rocprofv3 evidence bench.py's numbers are judged against (run on the GPU box, from the repo root):
#   1. --kernel-trace --stats of the bench command (no selectivity sweep, no CPU legs: every scan dispatch is the headline
#      configuration or the small result check)
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in their own passes (HBM traffic per launch; MI355X_MICROARCH.md)
# Raw output goes to gpurun_out/prof_<tag>/; tools/profile_summary.py condenses it into profiles/.
#   tools/profile_round.sh <tag> <git head the numbers belong to>
set -euo pipefail
tag="${1:-r04}"; head="${2:-unknown}"
repo="$(pwd)"
out="$repo/gpurun_out/prof_$tag"
mkdir -p "$out"
cd /tmp
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -o kt -- python3 "$repo/bench.py" --steps 20 --warmup 3 --no-cpu --no-sweep > "$out/bench_under_kt.log" 2>&1
echo "[profile] kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o pmc -- python3 "$repo/bench.py" --steps 10 --warmup 2 --no-cpu --no-pjoin --no-sweep > "$out/bench_under_pmc_fetch.log" 2>&1
echo "[profile] FETCH_SIZE done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o pmc -- python3 "$repo/bench.py" --steps 10 --warmup 2 --no-cpu --no-pjoin --no-sweep > "$out/bench_under_pmc_write.log" 2>&1
echo "[profile] WRITE_SIZE done"
cd "$repo"
# the device code these counters belong to (content hash; tools/profile_summary.py refuses a summary on another tree)
python3 -c "import sys; sys.path.insert(0, '.'); from dwarf_bench_amd.build import kernel_tree_sha256; print(kernel_tree_sha256())" > "$out/kernel_tree_sha256.txt"
python3 tools/profile_summary.py "$out" "$tag" "$head"
# the self-measured peaks bench.py divides by (RANDOM_GATHER_PEAK_G, LDS_ATOMIC_PEAK_G, the nt-read rate): raw output of
# the two micro-benchmarks, with the command and the git head, kept as profiles/<tag>_ubench.txt
for b in ubench ubench_lds; do
  if [ ! -x "tools/$b" ] || [ "tools/$b.hip" -nt "tools/$b" ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 "tools/$b.hip" -o "tools/$b"; fi
done
{ echo "# git head $head, tag $tag; commands: tools/ubench ; tools/ubench_lds (sources tools/ubench.hip, tools/ubench_lds.hip,"
  echo "# built with hipcc --offload-arch=gfx950 -O3), run on the GPU box by tools/profile_round.sh"
  echo "## tools/ubench"; ./tools/ubench; echo "## tools/ubench_lds"; ./tools/ubench_lds; } > "profiles/${tag}_ubench.txt" 2>&1
cp "profiles/${tag}_ubench.txt" "$out/"
# BASELINE config 5 on the one GPU of the box (the same three runs as tests/test_gpu_cli.py::
# test_partitioned_join_baseline_config_5, three iterations each): eight virtual ranks, the direct one-GPU join, and all
# pairs through the rank's own RCCL send/recv group
cli="dwarf_bench_amd/_lib/dwarf_bench"
run_pjoin() {  # <name> <gpus> [ENV=VALUE]
  local name="$1" gpus="$2" extra="${3:-DBENCH_UNUSED=1}"
  { echo "# git head $head, tag $tag; command: $extra DWARF_BENCH_VALIDATE_MAX=1 $cli PartitionedJoinHip --device=hip --gpus $gpus --iterations 3 --input_size 1073741824"
    env "$extra" DWARF_BENCH_VALIDATE_MAX=1 "$cli" PartitionedJoinHip --device=hip --gpus "$gpus" --iterations 3 --input_size 1073741824; } \
    > "profiles/${tag}_pjoin_2p30_${name}.txt" 2>&1
  cp "profiles/${tag}_pjoin_2p30_${name}.txt" "$out/"
}
run_pjoin 8_virtual_ranks 8
run_pjoin direct_one_gpu 1 DWARF_BENCH_PJOIN_DIRECT=1
run_pjoin rccl_self_exchange 1
echo "[profile] partitioned join logs done"
