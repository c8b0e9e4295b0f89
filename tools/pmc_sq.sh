#!/bin/bash
# SQ counter passes for the per-edge kernels (separate --pmc runs).  Usage: bash tools/pmc_sq.sh <outdir>
out=${1:-gpurun_out/pmc_sq}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
# (sets of <= 4 counters: larger sets made rocprofv3 crash on this image in round 2)
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_MFMA SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM" "SQ_INST_LEVEL_LDS SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmcsq/set$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traj --no-extra "${@:2}" > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmcsq/set*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "k_cfconv_radius" in kn:
            agg["k_cfconv_local (typed)" if "true>" in kn else "k_cfconv_radius"][r["Counter_Name"]].append(float(r["Counter_Value"]))
            continue
        for k in ("k_cfconv_fused", "k_edge_attr_poly", "k_edge_encoder", "k_pair_head_poly", "k_pair_head", "k_gin_layer", "k_gin_gather", "k_schnet_node_stage"):
            if k in kn:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                break
for k, v in agg.items():
    print(k, "launches", len(next(iter(v.values()))))
    for c in sorted(v):
        print("    %-28s %16.0f" % (c, sum(v[c]) / len(v[c])))
PY
