#!/bin/bash
# Counter passes for ONE kernel of a short bench run (on the GPU box): per-launch averages of the named counters.
#   bash tools/pmc_kernel.sh <kernel name substring> <out file> "<counter set 1>" ["<counter set 2>" ...] -- <bench args>
# Separate --pmc passes, only --kernel-trace beside them (MI355X_MICROARCH.md, rocprofv3 section).
k=$1; outf=$2; shift 2
sets=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do sets+=("$1"); shift; done
shift
tmp=$(mktemp -d /tmp/pmck.XXXXXX)
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "${sets[@]}"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $tmp/set$i -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traj --no-extra "$@" > /dev/null 2>&1
  echo "$(date +%T) pass $i ($set) rc $?"
done
python3 - <<PY > $root/$outf
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$tmp/set*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "$k" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("kernel *$k*: per-launch averages")
for c in sorted(agg):
    print("  %-28s %16.0f   (%d launches)" % (c, sum(agg[c]) / len(agg[c]), len(agg[c])))
PY
cat $root/$outf
rm -rf $tmp
