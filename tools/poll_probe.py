#!/usr/bin/env python3
"""Ad-hoc (GPU box): what the sampler's polls cost on a small batch (1 molecule x 100 conformers, 1044 steps, a poll every 64):
the trajectory landing / sending and the NaN + range poll of LangevinRun.check_nan, timed separately.   python tools/poll_probe.py"""
import sys, time, json
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from agdiff_amd import get_model, synth
dev = torch.device("cuda", 0)
cfg = bench.make_cfg("drugs", "saturated")
m = get_model(cfg); m.load_state_dict(synth.synth_state_dict(m.state_dict())); m = m.to(dev).eval()
b = bench.build_batch("drugs", 1, 100, 2021)
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
at, bi, bt, ba = T(b["atom_type"]), T(b["bond_index"]), T(b["bond_type"]), T(b["batch"])
pos = torch.randn(at.shape[0], 3).to(dev)
run = m.begin_sampling(at, pos, bi, bt, ba, b["num_graphs"], False, n_steps=1044, step_lr=1e-6, clip=1000.0, global_start_sigma=0.5,
                       w_global=1.0, save_traj=True, nan_check_every=64)
acc = {"land": 0.0, "send": 0.0, "poll": 0.0, "n": 0}
land0, send0, chk0 = run._traj_land, run._traj_send, run.check_nan
def land():
    t = time.perf_counter(); land0(); acc["land"] += time.perf_counter() - t
def send(u):
    t = time.perf_counter(); send0(u); acc["send"] += time.perf_counter() - t
def chk():
    t = time.perf_counter(); chk0(); acc["poll"] += time.perf_counter() - t; acc["n"] += 1
run._traj_land, run._traj_send, run.check_nan = land, send, chk
run.advance(20); torch.cuda.synchronize()
for k in acc: acc[k] = 0
t0 = time.perf_counter(); run.advance(1024); torch.cuda.synchronize(); t1 = time.perf_counter()
print(json.dumps({"ms_per_step": (t1 - t0) / 1024 * 1e3, "polls": acc["n"], "poll_ms_each": acc["poll"] / max(acc["n"], 1) * 1e3,
                  "land_ms_each": acc["land"] / max(acc["n"], 1) * 1e3, "send_ms_each": acc["send"] / max(acc["n"], 1) * 1e3}))
