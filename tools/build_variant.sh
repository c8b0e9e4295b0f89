#!/bin/bash
# A variant of libagdiff_hip.so that differs in ONE translation unit (the other objects are the tree's):
#   bash tools/build_variant.sh <file.hip> <name> [extra hipcc flags]   ->  _ab/lib_<name>.so     (AGDIFF_LIB=... selects it)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
src=$1; name=$2; shift 2
base=${src%.hip}
make -C "$root/agdiff_amd/csrc" > /dev/null
mkdir -p "$root/_ab/build_$name"
cd "$root/agdiff_amd/csrc"
extra=""
[ "$base" = nodeconv ] && extra="-fno-honor-nans -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp"
[ "$base" = edge ] && extra="-fno-honor-nans"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-function $extra "$@" -c $src -o "$root/_ab/build_$name/$base.o"
objs=$(ls _build/*.o | grep -v -F "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs "$root/_ab/build_$name/$base.o" -o "$root/_ab/lib_$name.so"
echo "built _ab/lib_$name.so"
