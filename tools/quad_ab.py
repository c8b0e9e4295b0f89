#!/usr/bin/env python3
"""agdiff_cfconv_node with the radius rows in quad tiles (k_cfconv_quad, the default on quads) against every target's own radius
tiles (k_cfconv_node, tune cfconv_quad_tiles = -1) on ONE box, one process: aggregates compared, six launches timed interleaved.
   python tools/quad_ab.py [--mols 36] [--copies 128] [--reps 20] [--kind drugs] [--precision f16x3]"""
import argparse, ctypes, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from agdiff_amd import _lib, drugs_model_config, get_model, qm9_model_config, synth

ap = argparse.ArgumentParser()
ap.add_argument("--mols", type=int, default=36)
ap.add_argument("--copies", type=int, default=128)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--kind", default="drugs")
ap.add_argument("--precision", default="f16x3")
ap.add_argument("--radius-poly", default="auto")
ap.add_argument("--passes", default="auto", choices=["auto", "full"])
ap.add_argument("--only", default=None, choices=["quad", "per_target"], help="run this variant only (counter passes)")
ap.add_argument("--radius-only", action="store_true")
args = ap.parse_args()
lib = _lib.load()
dev = torch.device("cuda", 0)
cfg = (qm9_model_config if args.kind == "qm9" else drugs_model_config)(beta_end=2e-5)
m = get_model(cfg)
m.precision = args.precision
m.radius_poly = args.radius_poly
m.poly_passes = args.passes
m.group_targets = 4
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


def six():
    for k in range(nc):
        _lib.check(lib.agdiff_cfconv_node(P, Tp, Wp, k, st), "agdiff_cfconv_node")


if args.only:
    pk.set_tuning(cfconv_quad_tiles=0 if args.only == "quad" else -1, local_poly_off=1 if args.radius_only else 0)
    print(json.dumps({"only": args.only, "x%d_ms" % nc: timeit(six)}))
    sys.exit(0)
cnt = ws.rad_cnt.cpu().numpy().astype(np.int64)
qt = topo.quad_tgt.cpu().numpy().astype(np.int64).reshape(-1, 4)
c4 = np.where(qt >= 0, cnt[np.maximum(qt, 0)], 0)
out = {"N": topo.N, "group_targets": topo.group_targets, "radius_rows": int(cnt.sum()), "local_rows": topo.L, "local_tiles": topo.T,
       "radius_tiles_per_target": int(((cnt + 15) // 16).sum()), "radius_tiles_quad": int(((c4 + 3) // 4).max(axis=1).sum())}
agg = {}
for name, v in (("per_target", -1), ("quad", 0)):
    pk.set_tuning(cfconv_quad_tiles=v)
    ws.agg.fill_(float("nan"))
    ws.variant_log.zero_()
    _lib.check(lib.agdiff_cfconv_node(P, Tp, Wp, 0, st), "agdiff_cfconv_node")
    torch.cuda.synchronize()
    agg[name] = ws.agg.clone()
    out["variant_%s" % name] = int(ws.variant_log.item())
a, b_ = agg["per_target"], agg["quad"]
out["agg_finite"] = bool(torch.isfinite(b_).all().item())
out["agg_normwise_diff"] = float(((a - b_).abs().max() / a.abs().max()).item())
# bitwise run-to-run
pk.set_tuning(cfconv_quad_tiles=0)
_lib.check(lib.agdiff_cfconv_node(P, Tp, Wp, 0, st), "agdiff_cfconv_node")
torch.cuda.synchronize()
out["quad_bitwise_repeatable"] = bool(torch.equal(ws.agg, b_))
res = {"per_target": [], "quad": [], "per_target_radius_only": [], "quad_radius_only": []}
for _ in range(args.rounds):
    for name, v in (("per_target", -1), ("quad", 0)):
        for lp in (0, 1):
            pk.set_tuning(cfconv_quad_tiles=v, local_poly_off=lp)
            res[name + ("_radius_only" if lp else "")].append(round(timeit(six), 4))
pk.set_tuning(cfconv_quad_tiles=0, local_poly_off=0)
out["x%d_ms" % nc] = res
print(json.dumps(out))
