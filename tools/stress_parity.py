#!/usr/bin/env python3
"""Ad-hoc stress (GPU box): HIP path vs the oracle on many seeded batches -- forward (both heads, edge lists) and a short
sampler run, QM9- and Drugs-shaped, random molecule counts.   python tools/stress_parity.py [--seeds 12]"""
import argparse, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from agdiff_amd import drugs_model_config, get_model, qm9_model_config, synth   # noqa: E402
from oracle import agdiff_oracle as O   # noqa: E402
from helpers import elem_err, rel_err   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=12)
ap.add_argument("--precision", default="f16x3")
args = ap.parse_args()
t = lambda x: torch.from_numpy(np.ascontiguousarray(x))
worst = {}
for seed in range(100, 100 + args.seeds):
    for kind, cfgf, scale in (("qm9", qm9_model_config, 2.0), ("drugs", drugs_model_config, 1.2)):
        rng = np.random.default_rng(seed)
        cfg = cfgf(num_diffusion_timesteps=10)
        m = get_model(cfg)
        m.precision = args.precision
        sd = synth.synth_state_dict(m.state_dict(), seed=seed) if "seed" in synth.synth_state_dict.__code__.co_varnames else synth.synth_state_dict(m.state_dict())
        m.load_state_dict(sd)
        m = m.cuda().eval()
        b = synth.make_packed_batch(kind, int(rng.integers(1, 5)), int(rng.integers(1, 4)), seed=seed)
        at, bi, bt, ba = t(b["atom_type"]), t(b["bond_index"]), t(b["bond_type"]), t(b["batch"])
        gen = torch.Generator().manual_seed(seed)
        pos = torch.randn(at.shape[0], 3, generator=gen) * scale
        ref = O.forward(sd, cfg, at, pos, bi, bt, ba, extend_order=False)
        got = m(at.cuda(), pos.cuda(), bi.cuda(), bt.cuda(), ba.cuda(), None, return_edges=True, extend_order=False)
        assert np.array_equal(got[2].cpu().numpy(), ref[2].numpy()) and np.array_equal(got[3].cpu().numpy(), ref[3].numpy()), (seed, kind)
        noise = torch.randn(4, at.shape[0], 3, generator=gen)
        rpos, _ = O.langevin_dynamics_sample_diffusion(sd, cfg, at, pos, bi, bt, ba, b["num_graphs"], False, n_steps=4, noise=noise,
                                                       w_global=1.0, global_start_sigma=float("inf"))
        gpos, _ = m.langevin_dynamics_sample_diffusion(at.cuda(), pos.cuda(), bi.cuda(), bt.cuda(), ba.cuda(), b["num_graphs"], False,
                                                       n_steps=4, noise=noise.cuda(), w_global=1.0, global_start_sigma=float("inf"))
        for name, g_, r_ in (("inv_global", got[0], ref[0]), ("inv_local", got[1], ref[1]), ("pos", gpos, rpos)):
            e = (rel_err(g_.cpu().numpy(), r_.numpy()), elem_err(g_.cpu().numpy(), r_.numpy()))
            k = (kind, name)
            worst[k] = tuple(max(a, b_) for a, b_ in zip(worst.get(k, (0.0, 0.0)), e))
for k, v in sorted(worst.items()):
    print("%-6s %-11s worst normwise %.2e  element-wise %.2e" % (k[0], k[1], v[0], v[1]))
