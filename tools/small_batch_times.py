#!/usr/bin/env python3
"""What a small batch's step is made of: stand-alone, back-to-back launches (one stream, each waits for the one before) of the
CFConv and node-stage kernels on a workspace a sampler run left, their chain as the step runs it, and the floor -- the same
number of launches of a one-word kernel.   python tools/small_batch_times.py [--mols 1 --copies 100] [--lds-sets K]"""
import argparse, ctypes, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from agdiff_amd import _lib, drugs_model_config, get_model, synth

ap = argparse.ArgumentParser()
ap.add_argument("--mols", type=int, default=1)
ap.add_argument("--copies", type=int, default=100)
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--lds-sets", type=int, default=0, help="tune_poly_lds_sets (0: as many as fit)")
ap.add_argument("--group", type=int, default=None)
args = ap.parse_args()
lib = _lib.load()
dev = torch.device("cuda", 0)
cfg = drugs_model_config(beta_end=2e-5)
m = get_model(cfg)
m.group_targets = args.group
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
if args.lds_sets:
    pk.struct.tune_poly_lds_sets = args.lds_sets
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
    return 1e3 * e0.elapsed_time(e1) / reps          # us


def ck(rc):
    assert rc == 0, rc


one = torch.zeros(1, device=dev)
out = {"atoms": topo.N, "graphs": int(b["num_graphs"]), "group_targets": topo.group_targets, "quads": int(topo.struct.num_quads),
       "variant_log": int(ws.variant_log.item())}
out["one_word_kernel_us"] = timeit(lambda: one.add_(1.0))
out["cfconv_us"] = [timeit(lambda k=k: ck(lib.agdiff_cfconv_node(P, Tp, Wp, k, st))) for k in (0, 1, nc - 1)]
out["node_stage_us"] = [timeit(lambda k=k: ck(lib.agdiff_schnet_node_stage(P, Tp, Wp, k, st))) for k in (0, 1, nc)]


def chain():
    for k in range(nc):
        ck(lib.agdiff_cfconv_node(P, Tp, Wp, k, st))
        ck(lib.agdiff_schnet_node_stage(P, Tp, Wp, k + 1, st))


cutoff = ctypes.c_float(float(cfg.cutoff))
par = [0]


def front():
    par[0] ^= 1
    ck(lib.agdiff_sampler_front(P, Tp, Wp, ctypes.byref(run.args), 1 | 2 | 4 | (par[0] << 4), cutoff, st))


ws.canon_counter.zero_()
out["front_us"] = timeit(front, 50)
out["head_poly_rows_us"] = timeit(lambda: ck(lib.agdiff_pair_head_poly_rows(P, Tp, Wp, par[0], st)), 50)
out["chain_of_%d_launches_us" % (2 * nc)] = timeit(chain, max(10, args.reps // 10))
out["floor_same_number_of_one_word_launches_us"] = timeit(lambda: [one.add_(1.0) for _ in range(2 * nc)], max(10, args.reps // 10))
print(json.dumps(out))
