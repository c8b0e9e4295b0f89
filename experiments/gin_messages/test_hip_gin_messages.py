"""GPU (MI355X): the GIN message sums with edge_attr recomputed from the per-type polynomials (csrc/node.hip k_gin_messages, over
the supertiles agdiff_topo_t.gn_*) against the row gather it replaces (k_gin_gather reading ws->l_attr_rows), the reference's
fixtures and the oracle: near sets, far sets, rows the polynomials do not cover (longer than the far range, a type without a set),
every arithmetic mode, inside the sampler.  Matches encoder/gin.py:38-69,112-148 and dualenc.py:214-239."""
import numpy as np
import pytest
import torch

from helpers import FORWARD_CASES, check_close, load_golden, rel_err, sampler_case_cfg, sampler_case_kwargs, t

pytestmark = pytest.mark.gpu
PRECISIONS = ["f32", "bf16x3", "f16x3"]


def _model(cfg, precision="f16x3", head_scale=1e-3, messages=True):
    from agdiff_amd import get_model
    from oracle import agdiff_oracle as O
    sd = O.synth_state_dict_for(cfg, head_scale=head_scale)
    m = get_model(cfg)
    m.precision = precision
    m.tuning["gin_msg_min_nodes"] = 1 if messages else -1
    m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    return m.to("cuda:0").eval(), sd


def _ran_messages(m):
    from agdiff_amd import _lib
    return bool(int(m._batch_cache[2].variant_log.item()) & _lib.DEFINES["AGDIFF_VAR_GIN_MESSAGES"])


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("case", ["g3_forward_qm9_small", "g3_forward_smooth_sparse", "g3_forward_drugs_capped"])
def test_forward_fixtures_with_recomputed_messages(case, precision):
    """The reference's forward fixtures (GIN output, local scores) with the message kernel forced onto these small batches; the
    same run with the row gather differs by the order of a node's additions only."""
    g = load_golden(case)
    args = [t(g[k]).cuda() for k in ("atom_type", "pos", "bond_index", "bond_type", "batch")]
    outs = {}
    for messages in (True, False):
        m, _ = _model(FORWARD_CASES[case](), precision, messages=messages)
        out = m(args[0], args[1], args[2], args[3], args[4], None, return_edges=True, extend_order=False)
        assert _ran_messages(m) == messages
        outs[messages] = (m._batch_cache[2].hl.view(-1, 128).cpu().numpy().copy(), out[1].cpu().numpy())
    if "gin_out" in g:
        check_close("gin_messages gin_out[%s]" % case, outs[True][0], g["gin_out"], precision)
    check_close("gin_messages inv_l[%s]" % case, outs[True][1], g["edge_inv_local"], precision)
    assert rel_err(outs[True][0], outs[False][0]) < 2e-6 and rel_err(outs[True][1], outs[False][1]) < 2e-5


@pytest.mark.parametrize("precision", PRECISIONS)
def test_far_rows_hard_rows_and_types_without_a_set(precision):
    """One batch, four molecules at four scales: compact (every local edge on its type's near set), spread over tens of
    Angstrom (far sets), spread beyond ten cutoffs (rows the encoder MLP wrote: read, not recomputed), and a mix; then the same
    with the 2-hop type's polynomials refused (the whole batch falls back to the row gather).  Against the oracle and the row gather."""
    from agdiff_amd import drugs_model_config, synth
    from oracle import agdiff_oracle as O
    cfg = drugs_model_config()
    b = synth.make_packed_batch("drugs", 4, 3, seed=41)
    at, bi, bt, ba = [t(b[k]) for k in ("atom_type", "bond_index", "bond_type", "batch")]
    gen = torch.Generator().manual_seed(2)
    pos = torch.randn(at.shape[0], 3, generator=gen)
    scale = torch.tensor([1.0, 9.0, 70.0, 4.0])[torch.from_numpy(b["mol_id"])[ba]]
    pos = pos * scale[:, None]
    for refuse in ((), (23,)):
        res = {}
        for messages in (True, False):
            m, sd = _model(cfg, precision, messages=messages)
            m.poly_refuse_types = refuse
            out = m(at.cuda(), pos.cuda(), bi.cuda(), bt.cuda(), ba.cuda(), None, return_edges=True, extend_order=False)
            # (a batch with a type whose polynomials were refused runs its local edges through the MLP kernels: the rows then come
            # from the encoder, not from agdiff_local_edge_rows' polynomial path, and the GIN layers gather them)
            assert _ran_messages(m) == (messages and not refuse)
            ws = m._batch_cache[2]
            res[messages] = (ws.hl.view(-1, 128).cpu().numpy().copy(), out[1].cpu().numpy())
            if messages and not refuse:
                flags = ws.enc_flags.cpu().numpy()
                lens = ws.lc_len[:m._batch_cache[1].Lc].cpu().numpy()
                assert flags[0] > 0 and (lens > 10 * cfg.cutoff).any() and ((lens > cfg.cutoff) & (lens < 10 * cfg.cutoff)).any() and (lens < cfg.cutoff).any()
        ref = O.forward(sd, cfg, at, pos, bi, bt, ba, extend_order=False)
        check_close("gin_messages far/hard inv_l refuse=%s" % (refuse,), res[True][1], ref[1].numpy(), precision)
        assert rel_err(res[True][0], res[False][0]) < 2e-6


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("case", ["g5_sampler_top", "g5_sampler_lowT_global", "g5_sampler_mixed_cliplocal"])
def test_sampler_fixtures_with_recomputed_messages(case, precision):
    """Inside the denoising loop (fused front: the lengths come from agdiff_sampler_front), against the reference's sampler
    fixtures; twice, bit for bit."""
    g = load_golden(case)
    m, _ = _model(sampler_case_cfg(g, case), precision, head_scale=float(g["head_scale"]))
    runs = []
    for _ in range(2):
        pos, traj = m.langevin_dynamics_sample_diffusion(
            t(g["atom_type"]).cuda(), t(g["pos_init"]).cuda(), t(g["bond_index"]).cuda(), t(g["bond_type"]).cuda(),
            t(g["batch"]).cuda(), int(g["num_graphs"]), extend_order=False, n_steps=int(g["n_steps"]),
            noise=t(g["noise"]).cuda(), **sampler_case_kwargs(g))
        assert _ran_messages(m)
        runs.append((pos.cpu(), torch.stack(traj)))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    check_close("gin_messages traj[%s]" % case, runs[0][1].numpy(), g["traj"], precision)
    check_close("gin_messages pos[%s]" % case, runs[0][0].numpy(), g["pos_final"], precision)


def test_molecule_of_300_atoms():
    """No size limit of its own: a 300-atom molecule (one wave's supertiles: 19) against the row gather."""
    from agdiff_amd import qm9_model_config, synth
    rng = np.random.default_rng(3)
    at, r, c, ty = synth.random_molecule(rng, 300)
    gen = torch.Generator().manual_seed(1)
    pos = torch.randn(300, 3, generator=gen) * 3.0
    hl = {}
    for messages in (True, False):
        m, _ = _model(qm9_model_config(), messages=messages)
        out = m(t(at).cuda(), pos.cuda(), t(np.stack([r, c])).cuda(), t(ty).cuda(), torch.zeros(300, dtype=torch.long).cuda(), None,
                extend_order=False)
        assert _ran_messages(m) == messages and torch.isfinite(out[1]).all()
        hl[messages] = m._batch_cache[2].hl.view(-1, 128).cpu().numpy().copy()
    assert rel_err(hl[True], hl[False]) < 2e-6
