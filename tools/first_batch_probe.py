#!/usr/bin/env python3
"""Ad-hoc (GPU box): per-step wall time of the first batches of the default job, step by step -- is the first batch of a process
slower on every step (clocks after the idle set-up) or once (a lazy initialisation inside the timed region)?"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from agdiff_amd import driver, drugs_model_config, get_model, synth
dev = torch.device("cuda", 0)
cfg = drugs_model_config(beta_end=2e-5)
m = get_model(cfg); m.load_state_dict(synth.synth_state_dict(m.state_dict())); m = m.to(dev).eval()
mols, confs = bench.drugs200_job(2021)
batches = driver.plan_batches(mols, confs, 196608)
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
for bi_, bm in enumerate(batches[:3] + batches[:1]):
    b = driver.pack_batch(bm, confs)
    at, bi, bt, ba = T(b["atom_type"]), T(b["bond_index"]), T(b["bond_type"]), T(b["batch"])
    pos = torch.randn(at.shape[0], 3, generator=torch.Generator().manual_seed(1)).to(dev)
    run = m.begin_sampling(at, pos, bi, bt, ba, b["num_graphs"], False, n_steps=25, step_lr=1e-6, clip=1000.0, global_start_sigma=0.5, w_global=1.0)
    run.advance(5); torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); run.advance(1); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("batch", bi_, "atoms", at.shape[0], "ms per step:", " ".join("%.2f" % t for t in ts), flush=True)
    del run
