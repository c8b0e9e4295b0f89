#!/usr/bin/env python3
"""Where does the split-bf16 error of the LOCAL head on default-initialised weights come from?
Forward on a small Drugs-shaped batch vs the oracle with (a) everything bf16x3, (b) only the local head in exact fp32
(weights re-packed in mode 0, agdiff_head_params_t.precision = 0), (c) everything fp32."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from agdiff_amd import drugs_model_config, get_model, synth, packing
from oracle import agdiff_oracle as O
from helpers import rel_err, elem_err, t

torch.manual_seed(1234)
cfg = drugs_model_config()
m0 = get_model(cfg)
sd = {k: v.detach().clone() for k, v in m0.state_dict().items()}
b = synth.make_packed_batch("drugs", 2, 2, seed=77)
at, bi, bt, ba = t(b["atom_type"]), t(b["bond_index"]), t(b["bond_type"]), t(b["batch"])
pos = torch.randn(at.shape[0], 3, generator=torch.Generator().manual_seed(5)) * 2.0
ref = O.forward({k: v.clone() for k, v in sd.items()}, cfg, at, pos, bi, bt, ba, extend_order=False)


def run(precision, head_f32=False, poly="auto", local=None):
    m = get_model(cfg)
    m.precision, m.radius_poly, m.precision_local = precision, poly, local
    m.load_state_dict({k: v.clone() for k, v in sd.items()})
    m = m.to("cuda:0").eval()
    pk = m.packed()
    keep = []
    if head_f32:
        for name, p in (("head_local", "grad_local_dist_mlp"), ("head_global", "grad_global_dist_mlp")):
            hp = getattr(pk.struct, name)
            w1 = torch.from_numpy(packing.pack_blocks(sd[p + ".layers.0.weight"].numpy(), kouter=True, mode=0)).cuda()
            w2 = torch.from_numpy(packing.pack_blocks(sd[p + ".layers.1.weight"].numpy(), mode=0)).cuda()
            keep += [w1, w2]
            if name == "head_local":
                hp.w1_pk, hp.w2_pk, hp.precision = ctypes.c_void_p(w1.data_ptr()), ctypes.c_void_p(w2.data_ptr()), 0
    got = m(at.cuda(), pos.cuda(), bi.cuda(), bt.cuda(), ba.cuda(), None, return_edges=True, extend_order=False)
    torch.cuda.synchronize()
    return [(rel_err(got[i].cpu().numpy(), ref[i].numpy()), elem_err(got[i].cpu().numpy(), ref[i].numpy())) for i in (0, 1)]


for label, kw in (("bf16x3, local branch f16x3 (default)", dict(precision="bf16x3")),
                  ("all bf16x3", dict(precision="bf16x3", local="bf16x3")), ("bf16x3 + local head fp32", dict(precision="bf16x3", head_f32=True, local="bf16x3")),
                  ("bf16x3 global, local branch f32", dict(precision="bf16x3", local="f32")),
                  ("all fp32", dict(precision="f32")), ("bf16x3 poly off", dict(precision="bf16x3", poly="off", local="bf16x3")),
                  ("bf16x3 poly off, local f16x3", dict(precision="bf16x3", poly="off"))):
    r = run(**kw)
    print("%-36s inv_g normwise %.2e elem %.2e | inv_l normwise %.2e elem %.2e" % (label, r[0][0], r[0][1], r[1][0], r[1][1]))
