#!/bin/bash
# tools/ab_tree.sh's run step for the default job without extras:  bash tools/ab_run.sh [reps]   (after `ab_tree.sh prep <rev>`)
cd "$GRAFT_REPO_ROOT"
for rep in $(seq ${1:-2}); do for t in _ab .; do
  (cd $t && python3 bench.py --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$t', 'value %.2f' % d['value'], 'conv_ms',round(d['roofline']['avg_launch_ms'],4),'step',round(d['ms_per_step'],3))")
done; done | tee gpurun_out/ab_run.txt
