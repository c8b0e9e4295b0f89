#!/usr/bin/env python3
"""The WHOLE default job (BASELINE configs[2]: 200 Drugs-shaped molecules x 2 U{50..500} conformers = 108,874 conformers, 5000 steps
each) through agdiff_amd.driver.run_job, wall clock: what bench.py extrapolates from timed steps, run for real once.
   python tools/full_job.py [--out profiles/r06_full_default_job.json] [--schedule saturated|default]   (about 11 minutes of GPU)"""
import argparse, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from agdiff_amd import driver, get_model, synth

ap = argparse.ArgumentParser()
ap.add_argument("--out", default=None)
ap.add_argument("--schedule", default="saturated")
ap.add_argument("--n-steps", type=int, default=5000)
ap.add_argument("--batches", type=int, default=0, help="only the first this many batches of the plan (0: all)")
args = ap.parse_args()
mols, confs_of = bench.drugs200_job(2021)
batches = driver.plan_batches(mols, confs_of, 196608)
if args.batches:
    mols = [m for bm in batches[:args.batches] for m in bm]
cfg = bench.make_cfg("drugs", args.schedule)
m = get_model(cfg)
fill = synth.restoring_state_dict if args.schedule == "default" else synth.synth_state_dict
m.load_state_dict(fill(m.state_dict()))
m = m.to("cuda:0").eval()
kw = dict(n_steps=args.n_steps, step_lr=1e-6, w_global=1.0, global_start_sigma=0.5, clip=1000.0)
confs = sum(confs_of(x["num_refs"]) for x in mols)
marks = []


def log(*a):
    marks.append(round(time.time() - t0, 1))
    print("%.1f s" % marks[-1], *a, flush=True)


with tempfile.TemporaryDirectory() as d:
    torch.cuda.synchronize()
    t0 = time.time()
    res = driver.run_job(m, mols, d, confs_of, 196608, kw, "cuda:0", log=log)
    torch.cuda.synchronize()
    wall = time.time() - t0
got = sum(v.shape[0] for k, v in res.items() if k.startswith("pos_gen_"))
finite = all(np.isfinite(v).all() for k, v in res.items() if k.startswith("pos_gen_"))
rec = {"what": "driver.run_job over the whole default job: plan, pack, topology (background thread), sample %d steps per batch, polls, "
               "per-batch .npz + merged file; trajectories not saved (scripts/test.py default)" % args.n_steps,
       "schedule": args.schedule, "molecules": len(mols), "conformers": int(confs), "conformers_written": int(got), "all_finite": bool(finite),
       "wall_s": wall, "conformers_per_s": confs / wall, "stats": dict(driver.SAMPLE_STATS), "batch_saved_at_s": marks}
print(json.dumps(rec))
if args.out:
    json.dump(rec, open(args.out, "w"), indent=1)
