#!/bin/bash
# A second build of the kernel library with one source compiled differently, for same-box A/B runs (ATRI_HIP_LIB=<out>):
#   tools/build_variant.sh <out.so> <source.hip> [extra hipcc flags, e.g. -DATRI_SPLIT_NOMIX=1]
# Every other object comes from the regular in-tree build (alphatriangle_amd/build/hip), which has to be up to date.
set -eo pipefail
out="$1"; src="$2"; shift 2
root="$(cd "$(dirname "$0")/.." && pwd)"
objs="$root/alphatriangle_amd/build/hip"
stem="$(basename "$src" .hip)"
tmp="$(mktemp -d)"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops \
  -Wno-unused-command-line-argument -Wno-unknown-warning-option "$@" -c "$src" -o "$tmp/$stem.o" 2> >(grep -v "recognized feature" >&2)
others=$(ls "$objs"/*.o | grep -v "/$stem.o$")
hipcc --offload-arch=gfx950 -shared -fPIC $others "$tmp/$stem.o" -o "$out"
rm -rf "$tmp"
echo "$out"
