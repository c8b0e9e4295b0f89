#!/usr/bin/env python3
"""One forward of a 2 x 2 Drugs-shaped batch against the oracle for first-layer scales between the 64- and 96-term bands, in the
three arithmetic modes: separates what the network's conditioning does to every mode from what one mode's range does.
   python tools/sharp_modes_probe.py [--scales 16,20,24,28]"""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from agdiff_amd import _lib, drugs_model_config, get_model, synth
from oracle import agdiff_oracle as O
from helpers import rel_err

ap = argparse.ArgumentParser()
ap.add_argument("--scales", default="16,20,24,28")
args = ap.parse_args()
dev = torch.device("cuda", 0)
T = lambda x: torch.from_numpy(np.ascontiguousarray(x))
cfg = drugs_model_config(beta_end=2e-5)
b = synth.make_packed_batch("drugs", 2, 2, seed=77)
at, bi, bt, ba = [T(b[k]) for k in ("atom_type", "bond_index", "bond_type", "batch")]
pos = torch.randn(at.shape[0], 3, generator=torch.Generator().manual_seed(5)) * 2.0
for scale in [float(x) for x in args.scales.split(",")]:
    m0 = get_model(cfg)
    sd = synth.synth_state_dict(m0.state_dict())
    for k in sd:
        if synth.canonical_key(k) == "edge_encoder_global.feature_expansion.weight":
            sd[k] = sd[k] * scale
    ref = O.forward({k: v.clone() for k, v in sd.items()}, cfg, at, pos, bi, bt, ba, extend_order=False)
    ref64 = O.forward({k: v.double() if v.is_floating_point() else v for k, v in sd.items()}, cfg, at, pos.double(), bi, bt, ba, extend_order=False) \
        if hasattr(O, "forward") else None
    rec = {"scale": scale, "max_abs_inv_g": float(ref[0].abs().max()),
           "oracle_fp32_vs_fp64": rel_err(ref[0].numpy(), ref64[0].float().numpy()) if ref64 is not None else None}
    for precision in ("f32", "bf16x3", "f16x3"):
        m = get_model(cfg)
        m.precision = precision
        m.load_state_dict(sd)
        m = m.to(dev).eval()
        try:
            got = m(at.to(dev), pos.to(dev), bi.to(dev), bt.to(dev), ba.to(dev), None, return_edges=True, extend_order=False)
            ws = m._batch_cache[2]
            rec[precision] = {"inv_g": rel_err(got[0].cpu().numpy(), ref[0].numpy()), "inv_l": rel_err(got[1].cpu().numpy(), ref[1].numpy()),
                              "vs_fp64": rel_err(got[0].cpu().numpy(), ref64[0].float().numpy()) if ref64 is not None else None,
                              "terms": 32 * m.packed().poly_kt, "max_h": float(ws.h.abs().max()), "max_xs": float(ws.xs.abs().max()),
                              "max_agg": float(ws.agg.abs().max())}
        except _lib.AgdiffRangeError as e:
            rec[precision] = {"range_error": str(e)[:120]}
    print(json.dumps(rec), flush=True)
