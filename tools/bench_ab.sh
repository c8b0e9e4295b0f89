#!/bin/bash
# Interleaved in-step A/B of the tree's library and every _ab/lib_*.so on ONE box: bench.py on one packed batch.
#   bash tools/bench_ab.sh [reps] [bench args]      (default args: --workload drugs --mols 36 --copies 128 --steps 60 --warmup 10)
reps=${1:-2}; shift
cd "$GRAFT_REPO_ROOT"
args=${@:---workload drugs --mols 36 --copies 128 --steps 60 --warmup 10}
for r in $(seq $reps); do
  for lib in agdiff_amd/libagdiff_hip.so _ab/lib_*.so; do
    [ -e "$lib" ] || continue
    AGDIFF_LIB=$PWD/$lib python3 bench.py $args --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-24s ms/step %.4f  cfconv in-step %.4f' % ('$lib'.split('/')[-1], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
  done
done
