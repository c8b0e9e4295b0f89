#!/bin/bash
# Interleaved stand-alone timing of k_cfconv_quad (tools/quad_ab.py --only quad) in the tree's library and every _ab/lib_*.so.
cd "$GRAFT_REPO_ROOT"
for r in $(seq ${1:-2}); do
  for lib in agdiff_amd/libagdiff_hip.so _ab/lib_*.so; do
    for v in quad per_target; do
    AGDIFF_LIB=$PWD/$lib python3 tools/quad_ab.py --only $v 2>/dev/null | tail -1 | sed "s|^|$lib $v |"
    done
  done
done | tee gpurun_out/quad_libs.txt
