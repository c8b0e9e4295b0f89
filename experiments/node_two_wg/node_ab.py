#!/usr/bin/env python3
"""SchNet node stage: one 16-wave workgroup per CU with three weight phases (k_schnet_node_stage) against two 8-wave workgroups
per CU with six (k_schnet_node_stage2, tune node_two_wg) on ONE box, one process: outputs compared bit for bit, seven stages timed
interleaved.   python tools/node_ab.py [--mols 36] [--copies 128] [--precision f16x3]"""
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
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--precision", default="f16x3")
args = ap.parse_args()
lib = _lib.load()
dev = torch.device("cuda", 0)
cfg = drugs_model_config(beta_end=2e-5)
m = get_model(cfg)
m.precision = args.precision
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
nc = cfg.num_convs


def timeit(fn, reps=args.reps):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def stages(ks):
    for k in ks:
        _lib.check(lib.agdiff_schnet_node_stage_split(P, Tp, Wp, k, 1, st), "agdiff_schnet_node_stage_split")


out = {"N": topo.N, "tiles": (topo.N + 15) // 16}
h0, agg0 = ws.h.clone(), ws.agg.clone()
res = {}
for name, v in (("one_wg", -1), ("two_wg", 1)):
    pk.set_tuning(node_two_wg=v)
    ws.h.copy_(h0)
    ws.variant_log.zero_()
    stages([1])                       # finish block 0 + lin1 of block 1 (reads agg, h; writes h, xs)
    torch.cuda.synchronize()
    res[name] = (ws.h.clone(), ws.xs.clone())
    out["variant_" + name] = int(ws.variant_log.item())
out["h_bitwise_equal"] = bool(torch.equal(res["one_wg"][0], res["two_wg"][0]))
out["xs_bitwise_equal"] = bool(torch.equal(res["one_wg"][1], res["two_wg"][1]))
out["h_finite"] = bool(torch.isfinite(res["two_wg"][0]).all().item())
t = {"one_wg": [], "two_wg": []}
for _ in range(args.rounds):
    for name, v in (("one_wg", -1), ("two_wg", 1)):
        pk.set_tuning(node_two_wg=v)
        t[name].append(round(timeit(lambda: stages(range(nc + 1))), 4))
out["x%d_ms" % (nc + 1)] = t
pk.set_tuning(node_two_wg=0)
print(json.dumps(out))
