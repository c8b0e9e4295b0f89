#!/bin/bash
# On the GPU box: stand-alone A/B of the quad-tile CFConv (tools/quad_ab.py), then the default job with and without it, interleaved.
#   bash tools/quad_ab.sh [job rounds]
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/quad_ab.txt
: > $out
for args in "--mols 36 --copies 128" "--mols 8 --copies 128"; do
  python3 tools/quad_ab.py $args 2>>gpurun_out/quad_ab.err | tail -1 >> $out || exit 1
done
for r in $(seq ${1:-2}); do
  for t in "cfconv_quad_tiles=-1" "cfconv_quad_tiles=0"; do
    python3 bench.py --no-cpu-baseline --no-extra --tune $t 2>>gpurun_out/quad_ab.err | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$t', 'value %.2f  ms/step %.3f  cfconv in-step %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))" >> $out || exit 1
  done
done
cat $out
