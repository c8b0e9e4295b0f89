#!/bin/bash
# The default job (bench.py, no extras) with each --tune setting given, interleaved on ONE box.
#   bash tools/job_ab.sh <rounds> <tune1> <tune2> ...     e.g.  bash tools/job_ab.sh 2 cfconv_quad_tiles=-1 cfconv_quad_tiles=0
cd "$GRAFT_REPO_ROOT"
rounds=$1; shift
for r in $(seq $rounds); do
  for t in "$@"; do
    python3 bench.py --no-cpu-baseline --no-extra --tune $t 2>>gpurun_out/job_ab.err | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$t', 'value %.2f  ms/step %.3f  cfconv in-step %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))" || exit 1
  done
done | tee gpurun_out/job_ab.txt
