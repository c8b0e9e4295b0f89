#!/bin/bash
# SQ / TCP counter passes (separate --pmc runs) of the two radius-row layouts of agdiff_cfconv_node on tools/quad_ab.py.
# Usage (GPU box): bash tools/pmc_quad.sh <outfile> [harness args]
out=${1:-gpurun_out/pmc_quad.txt}; shift
tmp=$(mktemp -d /tmp/pmcq.XXXXXX)
cd /tmp && export TMPDIR=/tmp
i=0
for var in per_target quad; do
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INST_LEVEL_VMEM" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  echo "pass $i start: $var $set" >> $GRAFT_REPO_ROOT/gpurun_out/pmc_progress.txt
  timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $tmp/$var.set$i -- python3 $GRAFT_REPO_ROOT/tools/quad_ab.py --reps 2 --only $var "$@" > /dev/null 2>&1
done
done
python3 - <<PY > $GRAFT_REPO_ROOT/$out
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$tmp/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        for k in ("k_cfconv_node", "k_cfconv_quad"):
            if k in kn:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                break
names = sorted(set(c for v in agg.values() for c in v))
print("%-34s %18s %18s" % ("counter (mean per launch)", "k_cfconv_node", "k_cfconv_quad"))
for c in names:
    m = lambda k: (sum(agg[k][c]) / len(agg[k][c])) if agg[k].get(c) else float("nan")
    print("%-34s %18.0f %18.0f" % (c, m("k_cfconv_node"), m("k_cfconv_quad")))
PY
rm -rf $tmp
cat $GRAFT_REPO_ROOT/$out
