#!/bin/bash
# Interleaved stand-alone timings (tools/nodeconv_time.py) of the tree's library and every _ab/lib_*.so on ONE box:
# CFConv x6 and node stage x7.   Usage (GPU box): bash tools/lib_ab.sh [reps] [harness args]
reps=${1:-2}; shift
cd "$GRAFT_REPO_ROOT"
for r in $(seq $reps); do
  for lib in agdiff_amd/libagdiff_hip.so _ab/lib_*.so; do
    [ -e "$lib" ] || continue
    AGDIFF_LIB=$PWD/$lib python3 tools/nodeconv_time.py "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-28s N %d  cfconv_x6 %.4f  radius-only %.4f  node_stage_x7 %.4f' % ('$lib'.split('/')[-1], d['N'], d['node_x6_ms'], d['node_radius_only_x6_ms'], d['node_stage_x7_ms']))"
  done
done
