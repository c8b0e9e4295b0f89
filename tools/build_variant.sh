#!/bin/bash
# A variant of libagdiff_hip.so that differs in ONE translation unit (the other objects are the tree's):
#   bash tools/build_variant.sh <file.hip> <name> [extra hipcc flags]   ->  _ab/lib_<name>.so     (AGDIFF_LIB=... selects it)
# The timing-experiment branches (-DAG_NODE_ABL=..., -DAG_FRONT_ABL=..., -DAG_HEADP_ABL=..., -DAG_NODE_NOCOPY: kernels with a phase
# taken out, WRONG results, for finding what a kernel waits for) are not in the product sources: when one of these macros is among
# the flags, tools/ablations/timing_branches.patch is applied to a copy of the file first.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
src=$1; name=$2; shift 2
base=${src%.hip}
make -C "$root/agdiff_amd/csrc" > /dev/null
mkdir -p "$root/_ab/build_$name"
cd "$root/agdiff_amd/csrc"
case "$*" in
  *AG_NODE_ABL*|*AG_FRONT_ABL*|*AG_HEADP_ABL*|*AG_NODE_NOCOPY*)
    tmp="$root/_ab/build_$name/src"; rm -rf "$tmp"; mkdir -p "$tmp/agdiff_amd/csrc"
    cp *.hip *.hpp "$tmp/agdiff_amd/csrc/"
    (cd "$tmp" && patch -p1 -s < "$root/tools/ablations/timing_branches.patch")
    cd "$tmp/agdiff_amd/csrc"
    set -- "$@" -I"$root/include";;
esac
extra=""
[ "$base" = nodeconv ] && extra="-fno-honor-nans -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp"
[ "$base" = edge ] && extra="-fno-honor-nans"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-function $extra "$@" -c $src -o "$root/_ab/build_$name/$base.o"
cd "$root/agdiff_amd/csrc"
objs=$(ls _build/*.o | grep -v -F "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs "$root/_ab/build_$name/$base.o" -o "$root/_ab/lib_$name.so"
echo "built _ab/lib_$name.so"
