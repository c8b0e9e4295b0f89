#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/*.npz from the REAL reference code.

Runs only in the build container: it imports /root/reference/src/agdiff (read-only, no
bytecode written) with tests/golden/ref_shims standing in for the third-party packages the
image lacks (torch_geometric / torch_scatter / torch_sparse / torch_cluster, rdkit via a
pre-seeded agdiff.utils.chem).  Nothing from the reference is copied; only inputs and the
reference's outputs are stored.  Re-run:  python tests/golden/make_golden.py

Weights are NOT stored: both the reference model here and the model under test are filled by
agdiff_amd.synth.synth_state_dict (closed-form, integer-hash based).
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, "ref_shims"))
sys.path.insert(0, "/root/reference/src")

import numpy as np
import torch

torch.manual_seed(0)
torch.set_num_threads(4)

import agdiff            # noqa: E402  (the reference package; empty __init__)
import agdiff.utils      # noqa: E402
chem = types.ModuleType("agdiff.utils.chem")
chem.BOND_TYPES = {i: i for i in range(22)}      # only len() is used on the model path
chem.BOND_NAMES = {i: str(i) for i in range(22)}
sys.modules["agdiff.utils.chem"] = chem
agdiff.utils.chem = chem

from agdiff.models.epsnet import get_model                         # noqa: E402
from agdiff.models.epsnet import dualenc as ref_dualenc            # noqa: E402
from agdiff.models import common as ref_common                     # noqa: E402
from agdiff.models import geometry as ref_geometry                 # noqa: E402

from agdiff_amd.config import qm9_model_config, drugs_model_config  # noqa: E402
from agdiff_amd import synth                                        # noqa: E402


def T(x, dtype=None):
    t = torch.from_numpy(np.asarray(x))
    return t if dtype is None else t.to(dtype)


def build_ref(cfg, head_scale=1e-3, weights="filler"):
    model = get_model(cfg)
    fill = synth.restoring_state_dict if weights == "restoring" else synth.synth_state_dict
    sd = fill(model.state_dict(), head_scale=head_scale)
    model.load_state_dict(sd)
    model.eval()
    return model


def small_batch(kind, seed, nmol, copies, pos_scale):
    b = synth.make_packed_batch(kind, nmol, copies, seed=seed)
    g = torch.Generator().manual_seed(seed)
    pos = torch.randn(b["atom_type"].shape[0], 3, generator=g) * pos_scale
    return b, pos


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %-28s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


# ------------------------------------------------------------------ G1 schedule, G7 keys
def g_schedule_and_keys():
    m = build_ref(qm9_model_config())
    sig = (1.0 - m.alphas).sqrt() / m.alphas.sqrt()
    idx = np.array([0, 1, 2499, 4998, 4999])
    save("g1_schedule", idx=idx, betas=m.betas[idx], alphas=m.alphas[idx], sigmas=sig[idx],
         n_below_half=int((sig < 0.5).sum()))
    with open(os.path.join(HERE, "g7_state_dict_keys.txt"), "w") as f:
        for k, v in m.state_dict().items():
            f.write("%s %s %s\n" % (k, "x".join(map(str, v.shape)) or "-", str(v.dtype).replace("torch.", "")))
    print("wrote g7_state_dict_keys.txt  (%d keys)" % len(m.state_dict()))
    for sched in ["quad", "linear", "const", "jsd"]:
        pass
    other = {}
    for sched in ["quad", "linear", "const", "jsd", "sigmoid"]:
        b = ref_dualenc.get_beta_schedule(sched, beta_start=1e-7, beta_end=2e-3, num_diffusion_timesteps=50)
        other[sched] = b
    save("g1_schedules_other", **other)


# ------------------------------------------------------------------ G2/G3/G4 forward pieces
def g_forward(name, cfg, kind, seed, nmol, copies, pos_scale, stages):
    m = build_ref(cfg)
    b, pos = small_batch(kind, seed, nmol, copies, pos_scale)
    at, bi, bt, ba = T(b["atom_type"]), T(b["bond_index"]), T(b["bond_type"]), T(b["batch"])
    emb_before = m.encoder_global.embedding.weight.detach().clone()
    with torch.no_grad():
        out = m(at, pos, bi, bt, ba, None, return_edges=True, extend_order=False, extend_radius=True)
    eg, el, ei, et, elen, lmask = out
    emb_after = m.encoder_global.embedding.weight.detach().clone()
    rec = dict(atom_type=at, pos=pos, bond_index=bi, bond_type=bt, batch=ba,
               edge_inv_global=eg, edge_inv_local=el, edge_index=ei, edge_type=et,
               edge_length=elen, local_edge_mask=lmask)
    with torch.no_grad():
        # G4: geometry helpers on these edges
        rec["eq_local"] = ref_geometry.eq_transform(el, pos, ei[:, lmask], elen[lmask])
        rec["eq_global"] = ref_geometry.eq_transform(eg * (1 - lmask.view(-1, 1).float()), pos, ei, elen)
        rec["clip_local_20"] = ref_dualenc.clip_norm(rec["eq_local"] * 1e4, limit=20.0)
        rec["center"] = ref_dualenc.center_pos(pos, ba)
        if stages:
            ea = m.edge_encoder_global(edge_length=elen, edge_type=et)
            rec["edge_attr"] = ea
            enc = m.encoder_global
            h0 = enc.embedding(at)
            rec["schnet_h0"] = h0
            blk = enc.interactions[0]
            rec["cfconv1_b0"] = blk.conv1(h0, ei, elen, ea)
            rec["cfconv2_b0"] = blk.conv2(h0, ei, elen, ea)
            ib = blk(h0, ei, elen, ea)
            rec["iblock_b0"] = ib
            rec["scaled_b0"] = enc.scaling_modules[0](ib.unsqueeze(-1)).squeeze(-1)
            rec["schnet_out"] = enc(at, ei, elen, ea)
            rec["gin_out"] = m.encoder_local(at, ei[:, lmask], ea[lmask])
            hp = ref_common.assemble_atom_pair_feature(rec["schnet_out"], ei, ea)
            rec["head_global_hidden1"] = torch.relu(m.grad_global_dist_mlp.layers[0](hp))
            rec["emb_rows_before"] = emb_before[:20]
            rec["emb_rows_after"] = emb_after[:20]
    save(name, **rec)
    return m, rec


# ------------------------------------------------------------------ G5 sampler w/ injected noise
class NoiseInjector:
    def __init__(self, noise):
        self.noise = noise
        self.k = 0
        self.orig = torch.randn_like

    def __enter__(self):
        def fake(x, *a, **kw):
            n = self.noise[self.k]
            self.k += 1
            assert n.shape == x.shape
            return n.clone()
        torch.randn_like = fake
        return self

    def __exit__(self, *a):
        torch.randn_like = self.orig


def g_sampler(name, cfg, kind, seed, nmol, copies, n_steps, head_scale=1e-3, weights="filler", **kw):
    m = build_ref(cfg, head_scale=head_scale, weights=weights)
    b, _ = small_batch(kind, seed, nmol, copies, 1.0)
    at, bi, bt, ba = T(b["atom_type"]), T(b["bond_index"]), T(b["bond_type"]), T(b["batch"])
    g = torch.Generator().manual_seed(seed + 7)
    n = at.shape[0]
    pos_init = torch.randn(n, 3, generator=g)
    noise = torch.randn(n_steps, n, 3, generator=g)
    ref_dualenc.tqdm = lambda it, **k: it
    with NoiseInjector(noise):
        pos, traj = m.langevin_dynamics_sample_diffusion(
            at, pos_init, bi, bt, ba, b["num_graphs"], extend_order=False, n_steps=n_steps, **kw)
    sig = (1.0 - m.alphas).sqrt() / m.alphas.sqrt()
    kwn = {("kw_" + k): (np.float64(v) if v is not None else np.float64("nan")) for k, v in kw.items()}
    save(name, atom_type=at, bond_index=bi, bond_type=bt, batch=ba, num_graphs=b["num_graphs"],
         pos_init=pos_init, noise=noise, pos_final=pos, traj=torch.stack(traj), n_steps=n_steps,
         sigmas=sig, head_scale=head_scale,
         cfg_T=cfg.num_diffusion_timesteps, cfg_beta_end=cfg.beta_end, cfg_smooth=int(cfg.smooth_conv), **kwn)


def g_restoring():
    """The synthetic checkpoint with a restoring force (agdiff_amd/synth.py: restoring_state_dict): one forward with all six
    outputs and a 14-step sampler run across the whole sigma range of a 14-step schedule (global branch on and off)."""
    cfg = drugs_model_config(num_diffusion_timesteps=14, beta_end=0.7)
    m = build_ref(cfg, weights="restoring")
    b, pos = small_batch("drugs", 41, 2, 2, 2.5)
    at, bi, bt, ba = T(b["atom_type"]), T(b["bond_index"]), T(b["bond_type"]), T(b["batch"])
    with torch.no_grad():
        eg, el, ei, et, elen, lmask = m(at, pos, bi, bt, ba, None, return_edges=True, extend_order=False, extend_radius=True)
    save("g14_forward_restoring", atom_type=at, pos=pos, bond_index=bi, bond_type=bt, batch=ba, edge_inv_global=eg,
         edge_inv_local=el, edge_index=ei, edge_type=et, edge_length=elen, local_edge_mask=lmask,
         cfg_T=cfg.num_diffusion_timesteps, cfg_beta_end=cfg.beta_end)
    g_sampler("g14_sampler_restoring", cfg, "drugs", 42, 2, 2, n_steps=14, weights="restoring",
              step_lr=1e-6, w_global=1.0, global_start_sigma=0.5, clip=1000.0)


def g_alanine():
    """BASELINE.json configs[0]: alanine dipeptide, 100 steps through the langevin_dynamics_sample wrapper as
    examples/test_alanine_dipeptide.py:303-320 calls it (qm9 config, w_global 1.0, global_start_sigma 0.5)."""
    m = build_ref(qm9_model_config())
    b = synth.alanine_dipeptide(3)
    at, bi, bt, ba = T(b["atom_type"]), T(b["bond_index"]), T(b["bond_type"]), T(b["batch"])
    g = torch.Generator().manual_seed(2021)
    n, n_steps = at.shape[0], 100
    pos_init = torch.randn(n, 3, generator=g)
    noise = torch.randn(n_steps, n, 3, generator=g)
    ref_dualenc.tqdm = lambda it, **k: it
    with NoiseInjector(noise):
        pos, traj = m.langevin_dynamics_sample(
            atom_type=at, pos_init=pos_init, bond_index=bi, bond_type=bt, batch=ba, num_graphs=3,
            extend_order=False, n_steps=n_steps, step_lr=1e-6, w_global=1.0, global_start_sigma=0.5, clip=1000.0,
            clip_local=None, sampling_type="ld", eta=1.0)
    save("g5_sampler_alanine", atom_type=at, bond_index=bi, bond_type=bt, batch=ba, num_graphs=3,
         pos_init=pos_init, noise=noise, pos_final=pos, traj=torch.stack(traj)[::10], n_steps=n_steps)


def g_loss(name, cfg, kind, seed, nmol, copies, pos_scale):
    """§8f-3: get_loss (dualenc.py:253-395) forward value; the two random draws are replaced by stored tensors."""
    m = build_ref(cfg, head_scale=1.0)
    b, pos = small_batch(kind, seed, nmol, copies, pos_scale)
    at, bi, bt, ba = T(b["atom_type"]), T(b["bond_index"]), T(b["bond_type"]), T(b["batch"])
    G = b["num_graphs"]
    g = torch.Generator().manual_seed(seed + 3)
    ts_half = torch.randint(0, m.num_timesteps, (G // 2 + 1,), generator=g)
    noise = torch.randn(at.shape[0], 3, generator=g)
    orig_randint, orig_normal = torch.randint, torch.Tensor.normal_
    torch.randint = lambda *a, **k: ts_half.clone()
    torch.Tensor.normal_ = lambda self, *a, **k: self.copy_(noise)
    try:
        with torch.no_grad():
            loss, lg, ll = m.get_loss(at, pos, bi, bt, ba, None, G, return_unreduced_loss=True, extend_order=False)
    finally:
        torch.randint, torch.Tensor.normal_ = orig_randint, orig_normal
    time_step = torch.cat([ts_half, m.num_timesteps - ts_half - 1], dim=0)[:G]
    save(name, atom_type=at, pos=pos, bond_index=bi, bond_type=bt, batch=ba, num_graphs=G, time_step=time_step,
         pos_noise=noise, loss=loss, loss_global=lg, loss_local=ll, cfg_smooth=int(cfg.smooth_conv))


def g_nan():
    """G6: NaN in positions -> FloatingPointError (dualenc.py:539-541)."""
    cfg = qm9_model_config(num_diffusion_timesteps=20)
    m = build_ref(cfg)
    b, _ = small_batch("qm9", 5, 2, 1, 1.0)
    at, bi, bt, ba = T(b["atom_type"]), T(b["bond_index"]), T(b["bond_type"]), T(b["batch"])
    pos_init = torch.randn(at.shape[0], 3)
    pos_init[3, 1] = float("nan")
    ref_dualenc.tqdm = lambda it, **k: it
    try:
        m.langevin_dynamics_sample_diffusion(at, pos_init, bi, bt, ba, b["num_graphs"],
                                             extend_order=False, n_steps=3)
        raised = False
    except FloatingPointError:
        raised = True
    assert raised
    print("G6 ok: reference raises FloatingPointError on NaN input")


def g_extend_order():
    """§8f-1: _extend_graph_order (common.py:135-205) on raw bond graphs."""
    rng = np.random.default_rng(11)
    recs = {}
    for i, n in enumerate([7, 19, 33]):
        pairs = set()
        for a in range(1, n):
            pairs.add((int(rng.integers(max(0, a - 3), a)), a))
        pairs = sorted(pairs)
        bt = rng.choice([1, 2, 12], size=len(pairs))
        src = np.array([p[0] for p in pairs] + [p[1] for p in pairs])
        dst = np.array([p[1] for p in pairs] + [p[0] for p in pairs])
        typ = np.concatenate([bt, bt])
        ei, et = ref_common._extend_graph_order(n, T(np.stack([src, dst])), T(typ), order=3)
        recs["n%d" % i] = n
        recs["bond_index%d" % i] = np.stack([src, dst])
        recs["bond_type%d" % i] = typ
        recs["ext_index%d" % i] = ei
        recs["ext_type%d" % i] = et
    save("g9_extend_order", **recs)


def g_gaussian():
    """Row a6b: edge_encoder='gaussian'.  The reference's edge.py uses GaussianSmearing without importing it
    (NameError at edge.py:24); the class exists in encoder/schnet.py:18-27, so it is put into edge's namespace
    here -- the one-line import the reference lacks -- and everything else runs unmodified."""
    from agdiff.models.encoder import edge as ref_edge, schnet as ref_schnet
    ref_edge.GaussianSmearing = ref_schnet.GaussianSmearing
    cfg = qm9_model_config(edge_encoder="gaussian")
    m, _ = g_forward("g3_forward_gaussian", cfg, "qm9", 5, 3, 1, 1.6, stages=True)
    with open(os.path.join(HERE, "g7_state_dict_keys_gaussian.txt"), "w") as f:
        for k, v in m.state_dict().items():
            f.write("%s %s %s\n" % (k, "x".join(map(str, v.shape)) or "-", str(v.dtype).replace("torch.", "")))
    print("wrote g7_state_dict_keys_gaussian.txt  (%d keys)" % len(m.state_dict()))
    g_forward("g3_forward_gaussian_drugs", drugs_model_config(edge_encoder="gaussian"), "drugs", 9, 2, 2, 1.5,
              stages=False)
    g_sampler("g5_sampler_gaussian", drugs_model_config(edge_encoder="gaussian", num_diffusion_timesteps=10),
              "qm9", 24, 2, 2, n_steps=10, step_lr=1e-6, w_global=1.0, global_start_sigma=0.5, clip=1000.0)


def g_forward_variants():
    """forward() argument variants of dualenc.py:142-178: extend_radius=False (bond graph only) and
    caller-supplied edge_index / edge_type / edge_length (shuffled order, lengths that are NOT |pos_i - pos_j|)."""
    cfg = qm9_model_config()
    m = build_ref(cfg)
    b, pos = small_batch("qm9", 41, 3, 2, 1.6)
    at, bi, bt, ba = T(b["atom_type"]), T(b["bond_index"]), T(b["bond_type"]), T(b["batch"])
    rec = dict(atom_type=at, pos=pos, bond_index=bi, bond_type=bt, batch=ba)
    with torch.no_grad():
        out = m(at, pos, bi, bt, ba, None, return_edges=True, extend_order=False, extend_radius=False)
        for k, v in zip(("inv_g", "inv_l", "edge_index", "edge_type", "edge_length", "mask"), out):
            rec["nr_" + k] = v
        full = m(at, pos, bi, bt, ba, None, return_edges=True, extend_order=False, extend_radius=True)
        g = torch.Generator().manual_seed(5)
        perm = torch.randperm(full[2].shape[1], generator=g)
        ei, et = full[2][:, perm], full[3][perm]
        el = full[4][perm] * (0.9 + 0.2 * torch.rand(perm.shape[0], 1, generator=g))
        out = m(at, pos, None, None, ba, None, edge_index=ei, edge_type=et, edge_length=el, return_edges=True)
        rec.update(given_edge_index=ei, given_edge_type=et, given_edge_length=el, given_inv_g=out[0],
                   given_inv_l=out[1])
        assert out[2] is ei and out[4] is el
    save("g11_forward_variants", **rec)


def g_mlp_act():
    """config.mlp_act other than relu (models/common.py:62-66 builds the heads' MultiLayerPerceptron with getattr(F, name)):
    one small forward per activation the HIP heads implement."""
    for act in ("gelu", "silu", "tanh", "sigmoid", "softplus", "leaky_relu", "elu", "celu", "relu6", "hardtanh", "selu", "mish", "hardswish",
                "hardsigmoid", "softsign", "logsigmoid", "hardshrink", "softshrink", "rrelu"):
        cfg = qm9_model_config(mlp_act=act)
        m = build_ref(cfg, head_scale=1.0)
        b, pos = small_batch("qm9", 51, 2, 2, 1.5)
        at, bi, bt, ba = T(b["atom_type"]), T(b["bond_index"]), T(b["bond_type"]), T(b["batch"])
        with torch.no_grad():
            out = m(at, pos, bi, bt, ba, None, return_edges=True, extend_order=False, extend_radius=True)
        save("g15_forward_act_" + act, atom_type=at, pos=pos, bond_index=bi, bond_type=bt, batch=ba,
             edge_inv_global=out[0], edge_inv_local=out[1], edge_index=out[2], edge_type=out[3])


def g_dsm():
    """config.type 'dsm' (dualenc.py:127-140): the module carries `sigmas` instead of betas / alphas; forward() is the same
    function of the weights (it never looks at the type).  Key list + one forward."""
    cfg = qm9_model_config(type="dsm", sigma_begin=10.0, sigma_end=0.01, num_noise_level=50)
    m = build_ref(cfg)
    with open(os.path.join(HERE, "g7_state_dict_keys_dsm.txt"), "w") as f:
        for k, v in m.state_dict().items():
            f.write("%s %s %s\n" % (k, "x".join(map(str, v.shape)) or "-", str(v.dtype).replace("torch.", "")))
    b, pos = small_batch("qm9", 61, 2, 2, 1.5)
    at, bi, bt, ba = T(b["atom_type"]), T(b["bond_index"]), T(b["bond_type"]), T(b["batch"])
    # the reference's forward does not run for this type: sigma_edge is only defined for 'diffusion' (dualenc.py:184-186, 210)
    try:
        with torch.no_grad():
            m(at, pos, bi, bt, ba, None, return_edges=True, extend_order=False, extend_radius=True)
        fwd = "ok"
    except UnboundLocalError as e:
        fwd = "UnboundLocalError"
    none = m.langevin_dynamics_sample(at, pos, bi, bt, ba, 4, False, n_steps=2)
    assert none is None
    save("g16_dsm", sigmas=m.sigmas, num_timesteps=m.num_timesteps, forward=np.str_(fwd))


def g_extend_order_forward():
    """§8f-1 on the model path: RAW bonds + extend_order=True (forward's default, dualenc.py:153,167-177 ->
    _extend_graph_order, common.py:135-205, applied to the whole batch) for one forward and one sampler run."""
    cfg = drugs_model_config(num_diffusion_timesteps=12)
    m = build_ref(cfg)
    b = synth.make_packed_batch("qm9", 3, 2, seed=51, raw_bonds=True)
    at, bi, bt, ba = T(b["atom_type"]), T(b["bond_index"]), T(b["bond_type"]), T(b["batch"])
    g = torch.Generator().manual_seed(51)
    pos = torch.randn(at.shape[0], 3, generator=g) * 1.7
    with torch.no_grad():
        out = m(at, pos, bi, bt, ba, None, return_edges=True, extend_order=True, extend_radius=True)
    rec = dict(atom_type=at, pos=pos, bond_index=bi, bond_type=bt, batch=ba, num_graphs=b["num_graphs"])
    for k, v in zip(("edge_inv_global", "edge_inv_local", "edge_index", "edge_type", "edge_length", "local_edge_mask"), out):
        rec[k] = v
    n_steps = 12
    pos_init = torch.randn(at.shape[0], 3, generator=g)
    noise = torch.randn(n_steps, at.shape[0], 3, generator=g)
    ref_dualenc.tqdm = lambda it, **k: it
    with NoiseInjector(noise):
        p, traj = m.langevin_dynamics_sample_diffusion(at, pos_init, bi, bt, ba, b["num_graphs"], extend_order=True,
                                                       n_steps=n_steps, step_lr=1e-6, w_global=1.0,
                                                       global_start_sigma=0.5, clip=1000.0)
    rec.update(pos_init=pos_init, noise=noise, pos_final=p, traj=torch.stack(traj), n_steps=n_steps)
    save("g12_extend_order_forward", **rec)


def g_covmat():
    """§8f-4: the COV / MAT reductions and the filtering of CovMatEvaluator.__call__ (utils/evaluation/covmat.py:
    104-165) on injected RMSD confusion matrices.  rdkit (GetBestRMS) is third party and absent: the matrices are
    inputs here, and only what the reference's own Python does with them is recorded."""
    for name in ("rdkit", "rdkit.Chem", "rdkit.Chem.rdForceFieldHelpers"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["rdkit"].Chem = sys.modules["rdkit.Chem"]
    sys.modules["rdkit.Chem.rdForceFieldHelpers"].MMFFOptimizeMolecule = lambda m: None
    ed = types.ModuleType("easydict")

    class EasyDict(dict):
        def __init__(self, d=None):
            super().__init__(d or {})
            self.__dict__ = self
    ed.EasyDict = EasyDict
    sys.modules.setdefault("easydict", ed)
    chem.get_best_rmsd = lambda a, b: 0.0
    chem.set_rdmol_positions = lambda m, p: m
    from agdiff.utils.evaluation import covmat as ref_covmat

    class Mol:
        def __init__(self, n):
            self.n = n

        def GetNumAtoms(self):
            return self.n
    rng = np.random.default_rng(13)
    mats, datas, recs = [], [], {}
    spec = [(9, 5, 10, "CCO"), (12, 7, 14, "c1ccccc1"), (6, 4, 9, "CC.O"), (8, 6, 11, "CCN"), (7, 3, 6, "C#N")]
    for i, (n, R, G, smi) in enumerate(spec):
        d = {"smiles": smi, "rdmol": Mol(n), "pos_ref": torch.randn(R * n, 3), "pos_gen": torch.randn(G * n, 3)}
        if i == 4:
            del d["pos_gen"]                       # skipped: nothing generated
        datas.append(d)
        recs["n%d" % i], recs["R%d" % i], recs["G%d" % i] = n, R, G
        recs["disconnected%d" % i] = int("." in smi)
        recs["has_gen%d" % i] = int(i != 4)
    # molecule 3 has G = 11 < ratio * R = 12 -> filtered; molecule 2 is disconnected -> filtered
    kept = [0, 1]
    for i in kept:
        R, G = spec[i][1], 2 * spec[i][1]
        m = np.abs(rng.normal(0.8, 0.5, size=(R, G)))
        mats.append(m)
        recs["confusion%d" % i] = m
    it = iter(mats)
    ref_covmat.get_rmsd_confusion_matrix = lambda data, useFF=False: next(it)
    ev = ref_covmat.CovMatEvaluator(num_workers=1, print_fn=lambda s: None)
    ev.pool.close(); ev.pool.join()

    class Serial:
        def imap(self, f, xs):
            return map(lambda x: ref_covmat.get_rmsd_confusion_matrix(x), xs)

        def close(self):
            pass

        def join(self):
            pass
    ev.pool = Serial()
    res = ev(datas)
    recs.update(thresholds=res.thresholds, CoverageR=res.CoverageR, MatchingR=res.MatchingR, CoverageP=res.CoverageP,
                MatchingP=res.MatchingP, kept=np.array(kept))
    lines = []
    df = ref_covmat.print_covmat_results(res, print_fn=lines.append)
    recs["df_values"] = df.values
    recs["df_columns"] = np.array(list(df.columns))
    recs["print_lines"] = np.array(lines)
    # evaluate_conf (covmat.py:38-41) on the first matrix
    it = iter([mats[0]])
    ref_covmat.get_rmsd_confusion_matrix = lambda data, useFF=False: next(it)
    cov, mat = ref_covmat.evaluate_conf(datas[0], threshold=0.5)
    recs["evaluate_conf"] = np.array([cov, mat])
    save("g13_covmat", **recs)


def g_losses():
    g_loss("g10_loss_qm9", qm9_model_config(), "qm9", 31, 3, 2, 1.6)
    g_loss("g10_loss_drugs", drugs_model_config(), "drugs", 32, 2, 2, 2.5)


if __name__ == "__main__":
    if sys.argv[1:] and all(a in ("gaussian", "alanine", "loss", "variants", "extend", "covmat", "restoring", "mlp_act", "dsm") for a in sys.argv[1:]):   # add without touching the others
        for a in sys.argv[1:]:
            {"gaussian": g_gaussian, "alanine": g_alanine, "loss": g_losses, "variants": g_forward_variants,
             "extend": g_extend_order_forward, "covmat": g_covmat, "restoring": g_restoring, "mlp_act": g_mlp_act, "dsm": g_dsm}[a]()
        sys.exit(0)
    g_schedule_and_keys()
    # G2+G3 uncapped QM9-shaped batch with per-stage outputs (small: 3 molecules x 1 copy)
    g_forward("g3_forward_qm9_small", qm9_model_config(), "qm9", 3, 3, 1, 1.6, stages=True)
    # same batch, smooth (cosine) cutoff, positions stretched so some pairs exceed the cutoff
    g_forward("g3_forward_smooth_sparse", drugs_model_config(), "qm9", 4, 3, 2, 4.5, stages=True)
    # G3 capped Drugs-shaped batch (compact positions -> 32-cap active), outputs only
    g_forward("g3_forward_drugs_capped", drugs_model_config(), "drugs", 8, 2, 2, 1.5, stages=False)
    # G5 samplers
    g_sampler("g5_sampler_top", qm9_model_config(), "qm9", 21, 2, 2, n_steps=8,
              step_lr=1e-6, w_global=1.0, global_start_sigma=0.5, clip=1000.0)
    g_sampler("g5_sampler_lowT_global", drugs_model_config(num_diffusion_timesteps=12, beta_end=2e-3),
              "drugs", 22, 1, 3, n_steps=12, step_lr=1e-6, w_global=1.0, global_start_sigma=0.5, clip=1000.0)
    g_sampler("g5_sampler_mixed_cliplocal", qm9_model_config(num_diffusion_timesteps=16, beta_end=0.05),
              "qm9", 23, 3, 1, n_steps=16, head_scale=1.0, step_lr=1e-6, w_global=0.3,
              global_start_sigma=0.5, clip=0.05, clip_local=0.02, clip_pos=30.0)
    g_nan()
    g_extend_order()
    g_gaussian()
    g_alanine()
    g_losses()
    g_forward_variants()
    g_extend_order_forward()
    g_covmat()
    g_restoring()
    g_mlp_act()
    g_dsm()
