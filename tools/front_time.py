#!/usr/bin/env python3
"""Stand-alone timing of agdiff_sampler_front (update + local lengths + radius graph) on the 8 x 128 Drugs batch as a sampler
run left it.   python tools/front_time.py [--mols 8 --copies 128]"""
import argparse, ctypes, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from agdiff_amd import _lib, drugs_model_config, get_model, synth
ap = argparse.ArgumentParser()
ap.add_argument("--mols", type=int, default=8)
ap.add_argument("--copies", type=int, default=128)
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
run = m.begin_sampling(at, pos_init, bi, bt, ba, b["num_graphs"], False, n_steps=8, step_lr=1e-6, clip=1000.0,
                       global_start_sigma=0.5, w_global=1.0, save_traj=False)
run.advance(8)
torch.cuda.synchronize()
ws, topo, pk = run.ws, run.topo, run.pk
P, Tp, Wp, st = ctypes.byref(pk.struct), ctypes.byref(topo.struct), ctypes.byref(ws.struct), _lib.stream_ptr()
cutoff = ctypes.c_float(float(cfg.cutoff))
out = {"N": topo.N}
for name, mode in (("update+local+graph", 1 | 2 | 4), ("local+graph", 2 | 4), ("graph", 2), ("update", 1), ("local", 4)):
    par = [0]

    def call():
        par[0] ^= 1
        assert lib.agdiff_sampler_front(P, Tp, Wp, ctypes.byref(run.args), mode | (par[0] << 4), cutoff, st) == 0
    ws.canon_counter.zero_()
    call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        call()
    e1.record()
    torch.cuda.synchronize()
    out[name + "_ms"] = e0.elapsed_time(e1) / 20
print(json.dumps(out))
