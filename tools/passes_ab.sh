#!/bin/bash
# Interleaved in-step A/B of the pass plans of the filter polynomials (agdiff_params_t.poly_plan: one pass for the high terms
# against three for every term) on ONE box: bench.py on one packed batch.
#   bash tools/passes_ab.sh [reps] [bench args]      (default args: --workload drugs --mols 36 --copies 128 --steps 60 --warmup 10)
reps=${1:-2}; shift
cd "$GRAFT_REPO_ROOT"
args=${@:---workload drugs --mols 36 --copies 128 --steps 60 --warmup 10}
for r in $(seq $reps); do
  for passes in auto full; do
    python3 bench.py $args --poly-passes $passes --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('poly_passes %-5s plan %d  ms/step %.4f  cfconv in-step %.4f' % ('$passes', d['config']['filter_polynomials']['pass_plan'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
  done
done
