#!/usr/bin/env python3
"""Wall clock of agdiff_amd.driver.run_job (plan, pack, topology, sample, save) on the first batches of the default job against
the time its GPU sampling alone takes, with the next batch prepared in a worker process (default), a background thread
(AGDIFF_PREPARE=thread) or inline (AGDIFF_PREPARE=inline).   python tools/job_wall.py [--batches 3] [--n-steps 600]"""
import argparse, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from agdiff_amd import driver, get_model, synth, topology

ap = argparse.ArgumentParser()
ap.add_argument("--batches", type=int, default=3)
ap.add_argument("--n-steps", type=int, default=600)
args = ap.parse_args()
mols, confs_of = bench.drugs200_job(2021)
batches = driver.plan_batches(mols, confs_of, 196608)
use = [m for bm in batches[:args.batches] for m in bm]
cfg = bench.make_cfg("drugs", "saturated")
m = get_model(cfg)
m.load_state_dict(synth.synth_state_dict(m.state_dict()))
m = m.to("cuda:0").eval()
kw = dict(n_steps=args.n_steps, step_lr=1e-6, w_global=1.0, global_start_sigma=0.5, clip=1000.0)
out = {}
# the floor: every batch packed and its topology built BEFORE the clock starts, then sampled and saved one after the other
prepared = [driver.prepare_batch(m, bm, confs_of) for bm in batches[:args.batches]]
for rep in range(2):
    with tempfile.TemporaryDirectory() as d:
        prepared = [driver.prepare_batch(m, bm, confs_of) for bm in batches[:args.batches]]      # (a topology is consumed by its run)
        torch.cuda.synchronize()
        t0 = time.time()
        for (packed, topo), bm in zip(prepared, batches[:args.batches]):
            pos, _, ok = driver.sample_batch(m, packed, "cuda:0", kw, log=lambda *_: None, topology=topo)
            driver._save_npz_atomic(driver._batch_path(d, bm), {"pos_gen_%d" % x["index"]: pos[off:off + n * g].numpy().reshape(g, n, 3)
                                                               for x, (off, n, g) in zip(bm, packed["spans"])})
        torch.cuda.synchronize()
        out.setdefault("sampling_and_saving_only", []).append(round(time.time() - t0, 2))
        print("sampling_and_saving_only", out["sampling_and_saving_only"], file=sys.stderr, flush=True)
for mode in ("warm-up", "inline", "thread", "process", "inline", "thread", "process"):
    os.environ["AGDIFF_PREPARE"] = "process" if mode == "warm-up" else mode
    topology._GROUP_ORDER_CACHE.clear()        # (every molecule of a real job is met once: no grouping is found in the cache)
    with tempfile.TemporaryDirectory() as d:
        torch.cuda.synchronize()
        t0 = time.time()
        res = driver.run_job(m, use, d, confs_of, 196608, kw, "cuda:0", log=lambda *_: None)
        torch.cuda.synchronize()
        out.setdefault(mode, []).append(round(time.time() - t0, 2))
        print(mode, out[mode], file=sys.stderr, flush=True)
    assert len([k for k in res if k.startswith("pos_gen_")]) == len(use)
out.pop("warm-up")
print(json.dumps({"batches": args.batches, "n_steps": args.n_steps, "molecules": len(use), "run_job_wall_s": out}))
