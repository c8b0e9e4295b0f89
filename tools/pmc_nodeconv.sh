#!/bin/bash
# SQ counter passes (separate --pmc runs, <= 4 counters each) for k_cfconv_node on the stand-alone harness tools/nodeconv_time.py.
# Usage (on the GPU box): bash tools/pmc_nodeconv.sh <outfile> [harness args]
out=${1:-gpurun_out/pmc_nodeconv.txt}; shift
tmp=$(mktemp -d /tmp/pmcnc.XXXXXX)
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INST_LEVEL_VMEM" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  echo "pass $i start: $set" | tee -a $GRAFT_REPO_ROOT/gpurun_out/pmc_progress.txt
  timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $tmp/set$i -- python3 $GRAFT_REPO_ROOT/tools/nodeconv_time.py --reps 2 "$@" > /dev/null 2>&1
  echo "pass $i done: $set" | tee -a $GRAFT_REPO_ROOT/gpurun_out/pmc_progress.txt
done
python3 - <<PY > $GRAFT_REPO_ROOT/$out
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$tmp/set*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        for k in ("k_cfconv_node", "k_schnet_node_stage", "k_rad_scales", "k_graph"):
            if k in kn:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                break
for k, v in agg.items():
    print(k, "launches", len(next(iter(v.values()))))
    for c in sorted(v):
        vals = v[c]
        print("    %-32s mean %16.0f   min %14.0f max %14.0f" % (c, sum(vals) / len(vals), min(vals), max(vals)))
PY
rm -rf $tmp
