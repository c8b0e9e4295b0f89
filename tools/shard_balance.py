#!/usr/bin/env python3
"""How well agdiff_amd.dist.shard_graphs balances an 8-rank run, measured on ONE GPU: for BASELINE
configs[3] (1000 Drugs-shaped molecules, the driver's plan at max_atoms x 8 per global batch) and configs[4] (2048 molecules
x 200 atoms), every rank's shard of a global batch is sampled alone -- W warm-up + K timed denoising steps -- and its time is
put next to the load the proxy predicted for it (sum over its graphs of n min(n - 1, 33) + local edges).  A rank's step time in
the real run is its shard's time (graphs are independent, weights replicated) plus the per-step all-gather.
   python tools/shard_balance.py [--world 8] [--steps 12] [--warmup 3] [--out profiles/r05_shard_balance.json]"""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from agdiff_amd import dist as adist, driver, get_model, synth

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--max-atoms", type=int, default=196608)
ap.add_argument("--out", default=None)
args = ap.parse_args()
dev = torch.device("cuda", 0)
world, W, K = args.world, args.warmup, args.steps


def model_for(kind):
    cfg = bench.make_cfg(kind, "saturated")
    m = get_model(cfg)
    m.load_state_dict(synth.synth_state_dict(m.state_dict()))
    return m.to(dev).eval(), cfg


def measure(name, packed, model, cfg):
    sizes, loc = adist.graph_weights(packed)
    w = sizes * np.minimum(sizes - 1, 33) + loc
    parts = adist.shard_graphs(sizes, loc, world)
    pred = np.array([w[g0:g1].sum() for g0, g1 in parts], dtype=np.float64)
    ms, atoms, edges = [], [], []
    for r in range(world):
        mine, (g0, g1), (lo, hi) = adist.shard_of(packed, r, world)
        el, run, _, _, _ = bench.timed_run(model, dev, mine, cfg, W, K, "saturated", True, False, 2021 + r, 0, False)
        ms.append(el / K * 1e3)
        atoms.append(int(hi - lo))
        edges.append(int(bench.live_edges(run)))
        del run
    ms = np.array(ms)
    rec = {"graphs": int(packed["num_graphs"]), "atoms": int(np.asarray(packed["atom_type"]).shape[0]),
           "per_rank": [{"rank": r, "graphs": int(parts[r][1] - parts[r][0]), "atoms": atoms[r], "edges_per_step": edges[r],
                         "predicted_load": float(pred[r]), "ms_per_step": float(ms[r])} for r in range(world)],
           "predicted_max_over_mean": float(pred.max() / pred.mean()), "measured_max_over_mean": float(ms.max() / ms.mean()),
           "measured_ms_per_unit_load_spread": float((ms / pred).max() / (ms / pred).min())}
    print(name, json.dumps({k: v for k, v in rec.items() if k != "per_rank"}), flush=True)
    return rec


out = {"world": world, "steps": K, "warmup": W,
       "what": "each rank's shard of one global batch sampled alone on one MI355X (saturated schedule, split-fp16): predicted load "
               "(dist.shard_graphs proxy) vs measured ms per denoising step"}
# configs[3]: the Drugs test set of 1000 molecules, G = 2 x U{50..500} conformers each
rng = np.random.default_rng(2021)
mols = []
for i in range(1000):
    at_, r_, c_, t_ = synth.random_molecule(rng, synth.sample_n_atoms(rng, "drugs"))
    mols.append(dict(atom_type=at_, edge_index=np.stack([r_, c_]), edge_type=t_, num_refs=int(rng.integers(50, 501)), name="m%d" % i, index=i))
confs_of = driver.num_confs("2x")
batches = driver.plan_batches(mols, confs_of, driver.sharded_capacity(args.max_atoms, world))
model, cfg = model_for("drugs")
order = sorted(range(len(batches)), key=lambda i: sum(len(m["atom_type"]) * confs_of(m["num_refs"]) for m in batches[i]))
out["configs3_drugs_1000"] = {"global_batches": len(batches), "max_atoms_per_global_batch": args.max_atoms * world, "measured": {}}
for tag, bi in (("largest_batch", order[-1]), ("median_batch", order[len(order) // 2])):
    out["configs3_drugs_1000"]["measured"][tag] = dict(batch_index=bi, **measure("configs3 " + tag, driver.pack_batch(batches[bi], confs_of), model, cfg))
del model
# configs[4]: 2048 molecules x ~200 atoms, one conformer each, ONE global batch over the 8 ranks
model, cfg = model_for("large")
out["configs4_large_2048x200"] = measure("configs4", bench.build_batch("large", 2048, 1, 2021), model, cfg)
if args.out:
    with open(os.path.join(ROOT, args.out), "w") as f:
        json.dump(out, f, indent=1)
print(json.dumps(out))
