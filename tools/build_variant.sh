#!/bin/bash
# Build a variant of libagdiff_hip.so for same-box A/B runs:   bash tools/build_variant.sh <name> [extra hipcc flags]
# (flags in $NODECONV_FLAGS go to nodeconv.hip only -- unset: the Makefile's scheduling strategy, empty: none --, $NODE_FLAGS to
# node.hip, $EDGE_FLAGS to edge.hip)
# -> _ab/lib_<name>.so (objects under _ab/build_<name>/; _ab/ is git-ignored but travels to the GPU box).
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p "$root/_ab/build_$name"
cd "$root/agdiff_amd/csrc"
for f in graph front edge nodeconv node eval api; do
  flags=""
  [ $f = edge ] && flags="-fno-honor-nans $EDGE_FLAGS"
  [ $f = nodeconv ] && flags="-fno-honor-nans -fno-slp-vectorize ${NODECONV_FLAGS--mllvm -amdgpu-sched-strategy=max-ilp}"
  [ $f = node ] && flags="$NODE_FLAGS"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-function $flags "$@" -c $f.hip -o "$root/_ab/build_$name/$f.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$root"/_ab/build_$name/*.o -o "$root/_ab/lib_$name.so"
echo "built _ab/lib_$name.so"
