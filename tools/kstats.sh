#!/bin/bash
# Per-kernel average durations (rocprofv3 --kernel-trace --stats) of a short bench run in tree $1 (default .).
# Usage: bash tools/kstats.sh <tree> <tag> [bench args]
tree=$(cd "${1:-.}" && pwd); tag=$2; shift 2
out=/tmp/kstats_$tag; rm -rf $out
cd /tmp && export TMPDIR=/tmp
(cd $tree && timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traj --no-extra "$@" > /dev/null 2>&1)
python3 - <<PY
import csv, glob
for f in glob.glob("$out/*/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:14]:
        print("%-14s %-60s calls %6s avg_us %9.2f total_ms %8.2f" % ("$tag", r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
