#!/bin/bash
# HBM traffic of the dominant kernel (MI355X_MICROARCH.md §HBM: separate --pmc passes; FETCH_SIZE x2 on gfx950 for
# wide coalesced reads; both counters are in KiB).  Usage: bash tools/pmc_traffic.sh <outdir>
out=${1:-gpurun_out/pmc_traffic}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmctr/$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traj --no-extra > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/pmctr/%s/*/*counter_collection.csv" % c)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            kn = r["Kernel_Name"]
            # the split CFConv's two instantiations of one template: <.., false> = radius list, <.., true> = typed local list
            if "k_cfconv_radius" in kn:
                agg["k_cfconv_local" if "true>" in kn else "k_cfconv_radius"].append(float(r["Counter_Value"]))
                continue
            for k in ("k_cfconv_fused", "k_edge_attr_poly", "k_edge_encoder", "k_pair_head_poly", "k_pair_head", "k_gin_layer", "k_gin_gather", "k_schnet_node_stage", "k_graph"):
                if k in kn:
                    agg[k].append(float(r["Counter_Value"]))
                    break
    res[c] = {k: sum(v) / len(v) for k, v in agg.items()}
import json
rec = {"kernels": {}}
for k in res["FETCH_SIZE"]:
    fk, wk = res["FETCH_SIZE"][k], res["WRITE_SIZE"].get(k, 0)
    print("%-22s FETCH_SIZE %10.0f KiB (x2 = %8.1f MB)  WRITE_SIZE %10.0f KiB (%7.1f MB)" % (k, fk, 2 * fk * 1024 / 1e6, wk, wk * 1024 / 1e6))
    rec["kernels"][k] = {"fetch_size_kib": fk, "write_size_kib": wk, "hbm_bytes_per_launch": 2 * fk * 1024 + wk * 1024}
json.dump(rec, open("$GRAFT_REPO_ROOT/$out/traffic.json", "w"), indent=1)
PY
