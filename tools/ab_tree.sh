#!/bin/bash
# A/B the working tree against another git revision ON THE SAME BOX (device clocks differ by ~10% between
# boxes, and an ABI change rules out swapping only the .so).  Prepare here:   bash tools/ab_tree.sh prep <rev>
# (exports <rev> to _ab/ and builds it), then on the GPU box:              bash tools/ab_tree.sh run [bench args]
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = prep ]; then
  rm -rf "$root/_ab" && mkdir -p "$root/_ab"
  git -C "$root" archive "$2" | tar -x -C "$root/_ab"
  make -C "$root/_ab/agdiff_amd/csrc" > /dev/null
  echo "prepared _ab/ at $(git -C "$root" rev-parse --short "$2")"
  exit 0
fi
shift
for rep in 1 2 3; do for t in "$root/_ab" "$root"; do
  (cd "$t" && python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-traj "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$t'.replace('$root','.') or '.', 'conv_ms',round(d['roofline']['avg_launch_ms'],4),'step',round(d['ms_per_step'],3))")
done; done
