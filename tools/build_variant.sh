#!/bin/bash
# development aid: an alternative build of libdbhip.so with extra -D knobs, for A/B timing through DBHIP_LIB
#   tools/build_variant.sh <tag> "<extra hipcc flags>" [files...]     -> dwarf_bench_amd/_lib/variants/libdbhip_<tag>.so
set -euo pipefail
tag="$1"; flags="$2"; shift 2
files="${*:-join_lds join pjoin}"
root="$(cd "$(dirname "$0")/.." && pwd)"
lib="$root/dwarf_bench_amd/_lib"; out="$lib/variants/$tag"
mkdir -p "$out"
objs=""
for f in $(ls "$root"/dwarf_bench_amd/csrc/*.hip); do
  b=$(basename "$f" .hip)
  if echo " $files " | grep -q " $b "; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function $flags -c "$f" -o "$out/$b.o"
    objs="$objs $out/$b.o"
  else
    objs="$objs $lib/obj/$b.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$lib/variants/libdbhip_$tag.so" $objs
echo "$lib/variants/libdbhip_$tag.so"
