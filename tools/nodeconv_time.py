#!/usr/bin/env python3
"""Stand-alone timing of agdiff_cfconv_node on the round-1/2 bench batch (8 Drugs-shaped molecules x 128 conformers), workspace as
a sampler run left it: all of it, radius rows only (tune_local_poly_off), typed sets from L2 (tune_poly_lds_sets = 1).
   python tools/nodeconv_time.py [--mols 8] [--copies 128] [--reps 20] [--kind drugs]"""
import argparse, ctypes, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from agdiff_amd import _lib, drugs_model_config, get_model, qm9_model_config, synth

ap = argparse.ArgumentParser()
ap.add_argument("--mols", type=int, default=8)
ap.add_argument("--copies", type=int, default=128)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--kind", default="drugs")
ap.add_argument("--precision", default="f16x3")
ap.add_argument("--group", type=int, default=None, help="targets per wave (BatchTopology group_targets): 4, 2, 1; default by batch size")
ap.add_argument("--only", default=None, choices=["node", "radius"], help="time one variant only (counter passes)")
args = ap.parse_args()
lib = _lib.load()
dev = torch.device("cuda", 0)
cfg = (qm9_model_config if args.kind == "qm9" else drugs_model_config)(beta_end=2e-5)
m = get_model(cfg)
m.precision = args.precision
m.group_targets = args.group
m.load_state_dict(synth.synth_state_dict(m.state_dict()))
m = m.to(dev).eval()
b = synth.make_packed_batch(args.kind, args.mols, args.copies, seed=2021)
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


cnt = ws.rad_cnt.cpu().numpy()
out = {"group_targets": topo.group_targets, "N": topo.N, "E": int(ws.num_edges.item()), "L": topo.L, "R": int(cnt.sum()), "local_tiles": topo.T,
       "radius_tiles": int(((cnt + 15) // 16).sum()), "radius_rows_padded": int((((cnt + 15) // 16) * 16).sum()),
       "rad_cnt_hist": np.bincount((cnt + 15) // 16, minlength=4).tolist()}
# how evenly the fixed pair -> wave assignment of k_cfconv_node spreads the work (cost model: every tile 1)
pt = topo.quad_tgt.cpu().numpy().reshape(-1, 4)
ltp = topo.lt_ptr.cpu().numpy()
tiles = (cnt + 15) // 16
cost = np.where(pt >= 0, tiles[np.maximum(pt, 0)], 0).sum(1) + 1.0 * (ltp[1:] - ltp[:-1])
W = 16 if pk.poly_kt == 1 else 12          # waves per workgroup of the instantiation (csrc/nodeconv.hip NodeConvShape)
Pn, wgs = cost.size, min(256, (cost.size + W - 1) // W)
per_wg = (Pn + wgs - 1) // wgs
wave_cost, wg_cost = np.zeros((wgs, W)), np.zeros(wgs)
for w in range(wgs):
    c = cost[w * per_wg:min((w + 1) * per_wg, Pn)]
    wg_cost[w] = c.sum()
    for k in range(W):
        wave_cost[w, k] = c[k::W].sum()
xcd = wg_cost.reshape(8, -1).sum(1) if wgs % 8 == 0 else wg_cost
out["balance"] = {"pair_cost_mean": float(cost.mean()), "pair_cost_std": float(cost.std()), "wave_max_over_mean": float(wave_cost.max() / wave_cost.mean()),
                  "wg_max_over_mean": float(wg_cost.max() / wg_cost.mean()), "xcd_contiguous_max_over_mean": float(xcd.max() / xcd.mean())}
if args.only:
    pk.set_tuning(local_poly_off=1 if args.only == "radius" else 0)
    out["%s_x%d_ms" % (args.only, nc)] = timeit(lambda: [lib.agdiff_cfconv_node(P, Tp, Wp, k, st) for k in range(nc)])
    print(json.dumps(out))
    sys.exit(0)
out["node_x%d_ms" % nc] = timeit(lambda: [lib.agdiff_cfconv_node(P, Tp, Wp, k, st) for k in range(nc)])
# all tiles / radius tiles only, interleaved (the first timing of a process runs on a colder chip)
ab = {"all": [], "radius_only": []}
for _ in range(3):
    for lp in (0, 1):
        pk.set_tuning(local_poly_off=lp)
        ab["radius_only" if lp else "all"].append(round(timeit(lambda: [lib.agdiff_cfconv_node(P, Tp, Wp, k, st) for k in range(nc)]), 4))
out["interleaved_x%d_ms" % nc] = ab
pk.set_tuning(local_poly_off=1)
out["node_radius_only_x%d_ms" % nc] = timeit(lambda: [lib.agdiff_cfconv_node(P, Tp, Wp, k, st) for k in range(nc)])
pk.set_tuning(local_poly_off=0, poly_lds_sets=1)
out["node_typed_from_l2_x%d_ms" % nc] = timeit(lambda: [lib.agdiff_cfconv_node(P, Tp, Wp, k, st) for k in range(nc)])
pk.set_tuning(poly_lds_sets=0)
out["node_stage_x%d_ms" % (nc + 1)] = timeit(lambda: [lib.agdiff_schnet_node_stage_split(P, Tp, Wp, k, 1, st) for k in range(nc + 1)])
out["scales_radius_ms"] = timeit(lambda: lib.agdiff_edge_scales_split(P, Tp, Wp, 0, st))
out["graph_build_ms"] = timeit(lambda: lib.agdiff_graph_build_ex(Tp, Wp, run.pos_p, ctypes.c_float(cfg.cutoff), 1, st))
if hasattr(lib, "agdiff_debug_node_stamps"):      # diagnostic build (-DAG_NODE_STAMPS): where a wave's time goes in a radius tile
    buf = (ctypes.c_uint64 * 8)()
    pk.set_tuning(local_poly_off=1)
    lib.agdiff_debug_node_stamps(None, 1)
    lib.agdiff_cfconv_node(P, Tp, Wp, 0, st)
    torch.cuda.synchronize()
    lib.agdiff_debug_node_stamps(buf, 1)
    v = list(buf)
    out["stamps_radius_only"] = {"cycles_per_tile_by_step": [x / max(v[5], 1) for x in v[:5]], "tiles": v[5],
                                 "clock_GHz": v[7] / max(v[6], 1) * 0.1}
    pk.set_tuning(local_poly_off=0)
print(json.dumps(out))
