#!/usr/bin/env python3
"""Stand-alone time of the seven SchNet node stages of a step (agdiff_schnet_node_stage) on a Drugs-shaped batch as a sampler
run left it.  With a timing build of node.hip (tools/build_variant.sh node.hip <name> -DAG_NODE_NOCOPY; AGDIFF_LIB=...) the
difference is what the weight staging costs.   python tools/node_time.py [--mols 36 --copies 128]"""
import argparse, ctypes, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from agdiff_amd import _lib, drugs_model_config, get_model, synth

ap = argparse.ArgumentParser()
ap.add_argument("--mols", type=int, default=36)
ap.add_argument("--copies", type=int, default=128)
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
lib = _lib.load()
dev = torch.device("cuda", 0)
cfg = drugs_model_config(beta_end=2e-5)
m = get_model(cfg)
m.load_state_dict(synth.synth_state_dict(m.state_dict()))
m = m.to(dev).eval()
b = synth.make_packed_batch("drugs", args.mols, args.copies, seed=2021)
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
at, bi, bt, ba = T(b["atom_type"]), T(b["bond_index"]), T(b["bond_type"]), T(b["batch"])
pos_init = torch.randn(at.shape[0], 3, generator=torch.Generator().manual_seed(2021)).to(dev)
run = m.begin_sampling(at, pos_init, bi, bt, ba, b["num_graphs"], False, n_steps=4, step_lr=1e-6, clip=1000.0,
                       global_start_sigma=0.5, w_global=1.0, save_traj=False, raise_on_nan=False)
run.advance(4)
torch.cuda.synchronize()
ws, topo, pk = run.ws, run.topo, run.pk
P, Tp, Wp, st = ctypes.byref(pk.struct), ctypes.byref(topo.struct), ctypes.byref(ws.struct), _lib.stream_ptr()


def stages():
    for k in range(cfg.num_convs + 1):
        _lib.check(lib.agdiff_schnet_node_stage(P, Tp, Wp, k, st), "agdiff_schnet_node_stage")


stages()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(args.reps):
    stages()
e1.record()
torch.cuda.synchronize()
print(json.dumps({"N": topo.N, "seven_stages_ms": e0.elapsed_time(e1) / args.reps}))
