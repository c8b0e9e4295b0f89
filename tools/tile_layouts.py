#!/usr/bin/env python3
"""CPU only: 16-row tiles one agdiff_cfconv_node launch walks on the default job's batches (bench.drugs200_job packed by
driver.plan_batches) while every molecule lies inside the cutoff (the saturated schedule: radius_graph keeps the first 33
candidates in index order), under the row layouts considered for the radius rows:
  per_target   ceil(rows / 16) tiles per target (k_cfconv_node)
  quad         max over a quad's targets of ceil(rows / 4) (k_cfconv_quad), with and without the radius column in the grouping
  dense4       rows padded to 4 per target, packed densely over the quad (a bound: needs hand-overs between targets)
  nopad        sum of rows / 16
   python tools/tile_layouts.py [--every 6]"""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from agdiff_amd import driver, topology

ap = argparse.ArgumentParser()
ap.add_argument("--every", type=int, default=6, help="every n-th batch of the job")
args = ap.parse_args()
mols, confs_of = bench.drugs200_job(2021)
batches = driver.plan_batches(mols, confs_of, 196608)
tot = {}
add = lambda k, v: tot.__setitem__(k, tot.get(k, 0) + int(v))
for bm in batches[::args.every]:
    b = driver.pack_batch(bm, confs_of)
    for col in (False, True):
        topo = topology.BatchTopology(b["atom_type"], b["bond_index"], b["bond_type"], b["batch"], b["num_graphs"], device="cpu",
                                      radius_column=col)
        N = topo.N
        gptr = topo.graph_ptr.numpy().astype(np.int64)
        ba = topo.batch64.numpy()
        n_of = np.diff(gptr)[ba]
        li = np.arange(N) - gptr[ba]
        m = np.minimum(n_of, 33)
        cand = np.where(li < m, m - 1, m)
        src, dst = topo.loc_src.numpy().astype(np.int64), topo.loc_dst.numpy().astype(np.int64)
        cnt = cand - np.bincount(dst[(src - gptr[ba[src]]) < m[dst]], minlength=N)
        qt = topo.quad_tgt.numpy().astype(np.int64).reshape(-1, 4)
        c4 = np.where(qt >= 0, cnt[np.maximum(qt, 0)], 0)
        tag = "_radius_column" if col else ""
        add("quad" + tag, ((c4 + 3) // 4).max(axis=1).sum())
        add("local" + tag, topo.T)
        if not col:
            add("per_target", ((cnt + 15) // 16).sum())
            add("dense4", ((((c4 + 3) // 4).sum(axis=1) + 3) // 4).sum())
            add("nopad", (cnt.sum() + 15) // 16)
            add("radius_rows", cnt.sum())
            add("local_rows", topo.L)
rows = tot["radius_rows"] + tot["local_rows"]
out = {"batches": len(batches[::args.every]), "radius_rows": tot["radius_rows"], "local_rows": tot["local_rows"], "layouts": {}}
for name, rt, lt in (("per_target", tot["per_target"], tot["local"]), ("quad", tot["quad"], tot["local"]),
                     ("quad_radius_column", tot["quad_radius_column"], tot["local_radius_column"]),
                     ("dense4", tot["dense4"], tot["local"]), ("nopad", tot["nopad"], tot["local"])):
    out["layouts"][name] = {"radius_tiles": rt, "local_tiles": lt, "radius_pad_frac": round(1 - tot["radius_rows"] / (16.0 * rt), 4),
                            "local_pad_frac": round(1 - tot["local_rows"] / (16.0 * lt), 4),
                            "executed_over_useful": round(16.0 * (rt + lt) / rows, 4)}
print(json.dumps(out, indent=1))
