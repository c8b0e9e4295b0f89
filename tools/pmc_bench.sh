#!/bin/bash
# Counter passes of the DEFAULT bench command (python bench.py: configs[2], 96 packed batches), shortened to 1 warm-up +
# 1 timed step per batch -- the mix of launches is the one of the timed region, so per-launch averages are comparable.
# Separate --pmc passes, never combined with any trace domain other than --kernel-trace (MI355X_MICROARCH.md, HBM/rocprofv3
# section: FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts wide coalesced reads at half their size: x2).
#   bash tools/pmc_bench.sh <round tag, e.g. r03> [precision]     (on the GPU box; writes gpurun_out/profiles_new/)
R=${1:-r03}; PREC=${2:-bf16x3}
out=$GRAFT_REPO_ROOT/gpurun_out/profiles_new
mkdir -p $out
tmp=$(mktemp -d /tmp/pmcbench.XXXXXX)
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_INST_LEVEL_VMEM TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  echo "$(date +%T) pmc pass $i start: $set" | tee -a $GRAFT_REPO_ROOT/gpurun_out/pmc_progress.txt
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $tmp/set$i -- python3 $GRAFT_REPO_ROOT/bench.py --precision $PREC --steps 1 --warmup 1 --no-cpu-baseline --no-traj --no-extra > $tmp/bench$i.json 2>/dev/null
  echo "$(date +%T) pmc pass $i done (rc $?)" | tee -a $GRAFT_REPO_ROOT/gpurun_out/pmc_progress.txt
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
names = ("k_cfconv_quad", "k_cfconv_node", "k_cfconv_fused", "k_sampler_front", "k_schnet_node_stage", "k_pair_head_poly", "k_pair_head", "k_gin_layer",
         "k_gin_gather", "k_edge_attr_poly", "k_edge_encoder")
for f in glob.glob("$tmp/set*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        for k in names:
            if k in kn:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                break
edges = None
for f in sorted(glob.glob("$tmp/bench*.json")):
    try:
        edges = json.loads(open(f).read().strip().splitlines()[-1])["roofline"]["edges_per_launch"]
        break
    except Exception:
        pass
rec = {"command": "python3 bench.py --precision $PREC --steps 1 --warmup 1 --no-cpu-baseline --no-traj --no-extra (one --pmc pass per counter set)",
       "edges_per_launch": edges, "kernels": {},
       "definitions": {"hbm_bytes_per_launch": "2 * FETCH_SIZE KiB + WRITE_SIZE KiB (gfx950 correction for wide reads)",
                       "valu_issue_busy": "4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * SQ_BUSY_CYCLES / 32 SQ instances)",
                       "mfma_pipe_busy": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * SQ_BUSY_CYCLES / 32)",
                       "valu_per_tile": "SQ_INSTS_VALU (MFMAs excluded) per 16 directed edges of the launch (pad rows' instructions included)"}}
with open("$out/${R}_${PREC}_pmc.txt", "w") as txt:
    for k, v in agg.items():
        m = {c: sum(x) / len(x) for c, x in v.items()}
        txt.write("%s launches %d\n" % (k, len(next(iter(v.values())))))
        for c in sorted(m):
            txt.write("    %-28s %16.0f\n" % (c, m[c]))
        e = {"counters": m}
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            e.update(fetch_size_kib=m["FETCH_SIZE"], write_size_kib=m["WRITE_SIZE"], hbm_bytes_per_launch=2048 * m["FETCH_SIZE"] + 1024 * m["WRITE_SIZE"])
        if m.get("SQ_BUSY_CYCLES"):
            cyc = 1024 * m["SQ_BUSY_CYCLES"] / 32
            if "SQ_ACTIVE_INST_VALU" in m:
                e["valu_issue_busy"] = 4 * m["SQ_ACTIVE_INST_VALU"] / cyc
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                e["mfma_pipe_busy"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / cyc
        if k in ("k_cfconv_node", "k_cfconv_quad") and edges and "SQ_INSTS_VALU" in m:
            e["valu_per_tile"] = (m["SQ_INSTS_VALU"] - m.get("SQ_INSTS_MFMA", 0)) / (edges / 16)
            e["lds_per_tile"] = m.get("SQ_INSTS_LDS", 0) / (edges / 16)
        rec["kernels"][k] = e
json.dump(rec, open("$out/${R}_${PREC}_pmc.json", "w"), indent=1)
print(open("$out/${R}_${PREC}_pmc.txt").read())
PY
rm -rf $tmp
