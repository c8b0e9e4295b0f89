#!/bin/bash
# Interleaved timing of tools/nodeconv_time.py with every _ab/lib_*.so (and the tree's own library) on ONE box.
# (The FIRST timing of a process runs on a colder chip: compare like positions only.)
# Usage (GPU box): bash tools/nodeconv_ab.sh [reps] [harness args]
reps=${1:-2}; shift
cd "$GRAFT_REPO_ROOT"
for r in $(seq $reps); do
  for lib in agdiff_amd/libagdiff_hip.so _ab/lib_*.so; do
    AGDIFF_LIB=$PWD/$lib python3 tools/nodeconv_time.py "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-34s node %.4f radius-only %.4f  l2 %.4f' % ('$lib'.split('/')[-1], d['node_x6_ms'], d['node_radius_only_x6_ms'], d['node_typed_from_l2_x6_ms']))"
  done
done
