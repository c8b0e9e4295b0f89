#!/usr/bin/env python3
"""Ad-hoc check (GPU box): 400 denoising steps spread over the WHOLE reference schedule (dense graphs at low sigma, spread-out
molecules with long bonds and nearly empty radius lists at high sigma), twice: finite and bitwise reproducible; prints how
many local tiles the last step sent through the encoder MLP and how many radius edges were left.   python tools/longrun_check.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from agdiff_amd import drugs_model_config, qm9_model_config, get_model, synth
# two checkpoints: the restoring-force one in the default split-fp16 mode (molecules stay compact: what a trained model's job
# looks like), and the plain filler -- no attraction, atoms fly apart at high sigma, bonded lengths of 100+ A drive the GIN state
# past the split-fp16 range watch -- in split-bf16, which has fp32's range
def make(cfg, weights):
    m_ = get_model(cfg)
    if weights == "restoring":
        m_.load_state_dict(synth.restoring_state_dict(m_.state_dict()))
    else:
        m_.load_state_dict(synth.synth_state_dict(m_.state_dict()))
        m_.precision, m_.precision_local = "bf16x3", "bf16x3"
    return m_.cuda().eval()


for kind, cfgf, weights in (("drugs", drugs_model_config, "restoring"), ("qm9", qm9_model_config, "restoring"), ("drugs", drugs_model_config, "filler")):
    cfg = cfgf()
    m = make(cfg, weights)
    b = synth.make_packed_batch(kind, 6, 20, seed=77)
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    at, bi, bt, ba = T(b["atom_type"]), T(b["bond_index"]), T(b["bond_type"]), T(b["batch"])
    g = torch.Generator().manual_seed(5)
    pos_init = torch.randn(at.shape[0], 3, generator=g).cuda()
    noise = torch.randn(400, at.shape[0], 3, generator=g).cuda()
    idx = np.linspace(cfg.num_diffusion_timesteps - 1, 0, 400).round().astype(int).tolist()
    outs = []
    for rep in range(2):
        p, tr = m.langevin_dynamics_sample_diffusion(at, pos_init, bi, bt, ba, b["num_graphs"], False, n_steps=400, step_lr=1e-6,
                                                     clip=1000.0, global_start_sigma=0.5, w_global=1.0, step_indices=idx, noise=noise)
        outs.append((p.clone(), torch.stack(tr)))
    same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ws = m._batch_cache[2]
    print(kind, weights, m.precision, "400 steps over the whole schedule: finite", bool(torch.isfinite(outs[0][0]).all()), "bitwise reproducible", same,
          "flagged local tiles at the end", int(ws.enc_flags[0].item()), "radius edges", int(ws.rad_cnt.sum().item()))

# a COMPLETE 5000-step job at the reference's schedule (scripts/test.py defaults), twice: finite, bitwise reproducible, centred
cfg = drugs_model_config()
m = make(cfg, "restoring")
b = synth.make_packed_batch("drugs", 8, 16, seed=99)
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
at, bi, bt, ba = T(b["atom_type"]), T(b["bond_index"]), T(b["bond_type"]), T(b["batch"])
pos_init = torch.randn(at.shape[0], 3, generator=torch.Generator().manual_seed(7)).cuda()
res = []
for rep in range(2):
    torch.manual_seed(123)
    p, _ = m.langevin_dynamics_sample_diffusion(at, pos_init, bi, bt, ba, b["num_graphs"], False, n_steps=cfg.num_diffusion_timesteps,
                                                step_lr=1e-6, clip=1000.0, global_start_sigma=0.5, w_global=1.0, save_traj=False)
    res.append(p.clone())
com = torch.zeros(b["num_graphs"], 3, device="cuda").index_add_(0, ba, res[0]) / torch.bincount(ba).unsqueeze(1)
print("full %d-step job (restoring-force checkpoint, split-fp16), %d atoms: finite %s, bitwise reproducible %s, max |pos| %.2e, max |centre of mass| / max |pos| %.2e"
      % (cfg.num_diffusion_timesteps, at.shape[0], bool(torch.isfinite(res[0]).all()), torch.equal(res[0], res[1]),
         float(res[0].abs().max()), float(com.abs().max() / res[0].abs().max())))
