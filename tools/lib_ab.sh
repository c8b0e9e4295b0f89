#!/bin/bash
# The default job (bench.py, no extras) with the tree's library and every _ab/lib_*.so, interleaved on ONE box.
#   bash tools/lib_ab.sh [rounds]
cd "$GRAFT_REPO_ROOT"
for r in $(seq ${1:-2}); do
  for lib in agdiff_amd/libagdiff_hip.so _ab/lib_*.so; do
    AGDIFF_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --no-extra 2>>gpurun_out/lib_ab.err | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$lib', 'value %.2f  ms/step %.3f  cfconv in-step %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))" || exit 1
  done
done | tee gpurun_out/lib_ab.txt
