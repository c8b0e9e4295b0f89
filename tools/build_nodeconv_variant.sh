#!/bin/bash
# A variant of libagdiff_hip.so that differs in nodeconv.hip only (the other objects are the tree's):
#   bash tools/build_nodeconv_variant.sh <name> [extra hipcc flags for nodeconv.hip]   ->  _ab/lib_<name>.so
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
make -C "$root/agdiff_amd/csrc" > /dev/null
mkdir -p "$root/_ab/build_$name"
cd "$root/agdiff_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-function -fno-honor-nans -fno-slp-vectorize \
  ${NODECONV_FLAGS--mllvm -amdgpu-sched-strategy=max-ilp} "$@" -c nodeconv.hip -o "$root/_ab/build_$name/nodeconv.o"
objs=$(ls _build/*.o | grep -v nodeconv.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs "$root/_ab/build_$name/nodeconv.o" -o "$root/_ab/lib_$name.so"
echo "built _ab/lib_$name.so"
