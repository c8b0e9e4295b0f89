#!/usr/bin/env python3
"""How the headline depends on the sharpness of the checkpoint's first encoder layer: the synthetic
checkpoint with edge_encoder_global.feature_expansion.weight scaled by 1, 2, 5, 8, 16, 40 -- per scale the number of
polynomial terms the host accepts for the radius edges (32 / 64 / refused -> filter MLPs), the fit error against the float64
networks, how many local edge types got a polynomial, parity of one forward against the oracle (small batch) and the
throughput of the saturated sampler on the round-1/2 bench batch (8 Drugs-shaped molecules x 128 conformers).
   python tools/sharpness_sweep.py [--out profiles/r04_sharpness_sweep.json]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from agdiff_amd import drugs_model_config, get_model, synth
from oracle import agdiff_oracle as O
from helpers import rel_err

ap = argparse.ArgumentParser()
ap.add_argument("--out", default=None)
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--scales", default="1,2,5,8,16,24,32,40")
ap.add_argument("--bounded", default="16,32,64,128", help="first layer (weight and bias) times s, its outputs' columns of the next layer divided by s: "
                "gelu(s u) / s -- kinks s times sharper at unchanged magnitudes (the plain first-layer scaling also blows the network's values up)")
ap.add_argument("--all-layers", default="1.5,2,3", help="every Linear weight of the edge encoder scaled by these factors")
ap.add_argument("--default-init-seeds", default="0,1,2", help="PyTorch-default-initialised models (torch.manual_seed) -- no synthetic checkpoint")
args = ap.parse_args()
dev = torch.device("cuda", 0)
T = lambda x: torch.from_numpy(np.ascontiguousarray(x))
cfg = drugs_model_config(beta_end=2e-5)
small = synth.make_packed_batch("drugs", 2, 2, seed=77)
big = synth.make_packed_batch("drugs", 8, 128, seed=2021)
pos_small = torch.randn(small["atom_type"].shape[0], 3, generator=torch.Generator().manual_seed(5)) * 2.0
rows = []
from agdiff_amd import _lib
cases = [("first_layer", float(x)) for x in args.scales.split(",") if x] + \
        [("first_layer_bounded", float(x)) for x in args.bounded.split(",") if x] + \
        [("all_encoder_layers", float(x)) for x in args.all_layers.split(",") if x] + \
        [("default_init", float(x)) for x in args.default_init_seeds.split(",") if x]
for kind_, scale in cases:
    def build(precision):
        if kind_ == "default_init":          # what nn.Linear / nn.Embedding give a freshly constructed model (the reference's own start)
            torch.manual_seed(int(scale))
        m_ = get_model(cfg)
        m_.precision = precision
        if kind_ == "default_init":
            sd_ = {k: v.clone() for k, v in m_.state_dict().items()}
        else:
            sd_ = synth.synth_state_dict(m_.state_dict())
            for k in sd_:
                ck = synth.canonical_key(k)
                if kind_ == "first_layer" and ck == "edge_encoder_global.feature_expansion.weight":
                    sd_[k] = sd_[k] * scale
                if kind_ == "first_layer_bounded" and ck in ("edge_encoder_global.feature_expansion.weight", "edge_encoder_global.feature_expansion.bias"):
                    sd_[k] = sd_[k] * scale
                if kind_ == "first_layer_bounded" and ck == "edge_encoder_global.edge_feature_mlp.0.weight":
                    w_ = sd_[k].clone()
                    w_[:, :cfg.hidden_dim] = w_[:, :cfg.hidden_dim] / scale          # (the columns that read the first layer's outputs)
                    sd_[k] = w_
                if kind_ == "all_encoder_layers" and ck.startswith("edge_encoder_global.") and ck.endswith(".weight") and "bond_emb" not in ck:
                    sd_[k] = sd_[k] * scale
            m_.load_state_dict(sd_)
        return m_.to(dev).eval(), sd_
    at, bi, bt, ba = [T(small[k]) for k in ("atom_type", "bond_index", "bond_type", "batch")]
    precision = "f16x3"
    m, sd = build(precision)
    ref = O.forward({k: v.clone() for k, v in sd.items()}, cfg, at, pos_small, bi, bt, ba, extend_order=False)
    try:
        got = m(at.to(dev), pos_small.to(dev), bi.to(dev), bt.to(dev), ba.to(dev), None, return_edges=True, extend_order=False)
    except _lib.AgdiffRangeError as e:         # node features beyond the split-fp16 range: the range-safe mode
        precision = "bf16x3"
        m, sd = build(precision)
        m.precision_local = "bf16x3"
        got = m(at.to(dev), pos_small.to(dev), bi.to(dev), bt.to(dev), ba.to(dev), None, return_edges=True, extend_order=False)
    pk = m.packed()
    rec = {"case": kind_, "first_layer_scale": scale if kind_ == "first_layer" else None, "scale_or_seed": scale, "precision": precision, "max_abs_reference_inv_g": float(ref[0].abs().max()),
           "poly_kt": int(pk.poly_kt), "terms": 32 * int(pk.poly_kt), "pass_plan": int(pk.poly_plan),
           "one_pass_bound_of_high_terms": max(pk.poly_high_bound.values()) if pk.poly_high_bound else None,
           "fit_errors": {str(k): float(v) for k, v in pk.poly_errors.items()},
           "local_type_slots": int(pk.struct.poly_num_slots), "refused_local_types": sorted(int(t_) for t_ in pk.poly_refused_types),
           "parity_inv_g": rel_err(got[0].cpu().numpy(), ref[0].numpy()), "parity_inv_l": rel_err(got[1].cpu().numpy(), ref[1].numpy())}
    at, bi, bt, ba = [T(big[k]).to(dev) for k in ("atom_type", "bond_index", "bond_type", "batch")]
    pos_init = torch.randn(at.shape[0], 3, generator=torch.Generator().manual_seed(2021)).to(dev)
    n = 10 + args.steps
    run = m.begin_sampling(at, pos_init, bi, bt, ba, big["num_graphs"], False, n_steps=n, step_lr=1e-6, clip=1000.0,
                           global_start_sigma=0.5, w_global=1.0, save_traj=False)
    try:
        run.advance(10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run.advance(args.steps)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        run.check_nan()
    except (FloatingPointError, _lib.AgdiffRangeError) as e:      # (the scaled network's own dynamics diverge: nothing to time)
        rec.update(ms_per_step_8x128=None, conformers_per_s_8x128=0.0, path="sampler diverged (%s)" % type(e).__name__)
        rows.append(rec)
        print(json.dumps(rec), flush=True)
        del run, m
        continue
    rec.update(ms_per_step_8x128=ms, conformers_per_s_8x128=big["num_graphs"] / (ms * 5000 / 1e3),
               path="filter polynomials, %d terms" % (32 * pk.poly_kt) if pk.poly_kt else "filter MLPs (fit refused at 32 .. 128 terms)")
    rows.append(rec)
    print(json.dumps(rec), flush=True)
    del run, m
base = rows[0]["conformers_per_s_8x128"]
for r in rows:
    r["fraction_of_scale_1"] = r["conformers_per_s_8x128"] / base          # (of the first row: the synthetic checkpoint as it is)
if args.out:
    json.dump({"workload": "8 Drugs-shaped molecules x 128 conformers, saturated schedule, %d timed steps; parity: one forward of a 2 x 2 batch against the oracle (normwise)" % args.steps,
               "rows": rows}, open(args.out, "w"), indent=1)
