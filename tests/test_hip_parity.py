"""GPU (MI355X): the HIP path, called through the C ABI (include/agdiff_hip.h), against the oracle
and the reference-generated golden fixtures.  Floating point: BASELINE.json's north_star asks for 1e-4 relative
fp32; the gates here are per arithmetic mode and about ten times tighter (helpers.TOL_NORM: exact-fp32 MFMA 1e-5,
split-bf16 5e-5 normwise = max|a-b| / max|b|) plus an element-wise figure with an absolute floor
(helpers.TOL_ELEM); every comparison prints and records both figures (gpurun_out/parity_errors.json).
Bit-exact for every index / integer output."""
import ctypes

import numpy as np
import pytest
import torch

from helpers import (FORWARD_CASES, SAMPLER_CASES, STAGE_CASES, check_close, load_golden, rel_err, sampler_case_cfg,
                     sampler_case_kwargs, t)

pytestmark = pytest.mark.gpu


PRECISIONS = ["f32", "bf16x3", "f16x3"]


def _gpu_model(cfg, head_scale=1e-3, precision="f16x3"):
    from agdiff_amd import get_model
    from oracle import agdiff_oracle as O
    sd = O.synth_state_dict_for(cfg, head_scale=head_scale)
    m = get_model(cfg)
    m.precision = precision
    m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    return m.to("cuda:0").eval(), sd


def unfrag(frag, n, precision):
    """operand-form edge attrs (csrc/common.hpp: [tile][t][u][edge column][quarter] 16-byte units) -> row-major
    [n][128] fp32."""
    tiles = frag.numel() // (16 * 128)
    if precision == "f32":      # [tile][t][u][edge][q][r], feature = 32t + 16u + 4q + r
        x = frag.view(tiles, 4, 2, 16, 4, 4).permute(0, 3, 1, 2, 4, 5).reshape(tiles * 16, 128)
        return x[:n]
    # bf16x3 / f16x3: [tile][t][part][edge][q][u][r] 16-bit floats, feature = 32t + 16u + 4q + r, value = hi + lo
    if precision == "f16x3":
        val = frag.view(torch.float16).view(tiles, 4, 2, 16, 4, 2, 4).to(torch.float32)
    else:
        u = frag.view(torch.int16).view(tiles, 4, 2, 16, 4, 2, 4).to(torch.int32)
        val = ((u & 0xFFFF) << 16).view(torch.float32)
    val = val[:, :, 0] + val[:, :, 1]                           # [tile][t][edge][q][u][r]
    x = val.permute(0, 2, 1, 4, 3, 5).reshape(tiles * 16, 128)
    return x[:n]


def test_dense_layer_orientations_on_asymmetric_weights():
    """MFMA lane maps: the edge encoder's last layer with identity-like inputs is covered end to end by
    test_edge_encoder; here the stand-alone aggregate op checks the CSR / gather convention."""
    from agdiff_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    for F in (128, 64):
        n, deg = 257, None
        degs = torch.randint(0, 40, (n,), generator=g)
        ptr = torch.cat([torch.zeros(1, dtype=torch.long), degs.cumsum(0)])
        E = int(ptr[-1])
        src = torch.randint(0, n, (E,), generator=g)
        x = torch.randn(n, F, generator=g)
        W = torch.randn(E, F, generator=g)
        dst = torch.repeat_interleave(torch.arange(n), degs)
        ref = torch.zeros(n, F).index_add_(0, dst, x[src] * W)
        xd, Wd = x.cuda(), W.cuda()
        pd, sd_ = ptr.int().cuda(), src.int().cuda()
        out = torch.empty(n, F, device="cuda")
        rc = lib.agdiff_cfconv_aggregate(_lib.ptr(xd), _lib.ptr(Wd), _lib.ptr(pd), _lib.ptr(sd_), ctypes.c_int64(n), F,
                                         _lib.ptr(out), _lib.stream_ptr())
        assert rc == 0
        torch.cuda.synchronize()
        assert rel_err(out.cpu().numpy(), ref.numpy()) < 1e-5


@pytest.mark.parametrize("case", list(FORWARD_CASES))
def test_graph_build_bit_exact(case):
    """radius_graph + coalesce order + types: bit-exact vs the reference fixture (common.py:208-233)."""
    from agdiff_amd import _lib
    from agdiff_amd.topology import BatchTopology, Workspace
    g = load_golden(case)
    lib = _lib.load()
    topo = BatchTopology(g["atom_type"], g["bond_index"], g["bond_type"], g["batch"], device="cuda")
    ws = Workspace(topo)
    pos = t(g["pos"]).cuda().contiguous()
    rc = lib.agdiff_graph_build(ctypes.byref(topo.struct), ctypes.byref(ws.struct), _lib.ptr(pos),
                                ctypes.c_float(10.0), _lib.stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    E = int(ws.num_edges.item())
    assert E == g["edge_index"].shape[1]
    perm = ws.ref2dst[:E].long()
    assert np.array_equal(np.sort(perm.cpu().numpy()), np.arange(E))
    ei = torch.stack([ws.e_src[:E][perm], ws.e_dst[:E][perm]]).cpu().numpy()
    assert np.array_equal(ei, g["edge_index"])
    assert np.array_equal(ws.e_type[:E][perm].cpu().numpy(), g["edge_type"])
    assert rel_err(ws.e_len[:E][perm].cpu().numpy(), g["edge_length"][:, 0]) < 1e-6
    # destination-sorted CSR invariants
    dst = ws.e_dst[:E].cpu().numpy(); src = ws.e_src[:E].cpu().numpy(); ip = ws.in_ptr.cpu().numpy()
    assert np.all(np.diff(dst) >= 0) and ip[-1] == E
    assert np.array_equal(np.bincount(dst, minlength=topo.N), np.diff(ip))
    same = np.diff(dst) == 0
    assert np.all(np.diff(src)[same] > 0)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("case", STAGE_CASES)
def test_edge_encoder_and_stages(case, precision):
    from agdiff_amd import _lib
    g = load_golden(case)
    cfg = FORWARD_CASES[case]()
    m, sd = _gpu_model(cfg, precision=precision)
    out = m(t(g["atom_type"]).cuda(), t(g["pos"]).cuda(), t(g["bond_index"]).cuda(), t(g["bond_type"]).cuda(),
            t(g["batch"]).cuda(), None, return_edges=True, extend_order=False)
    torch.cuda.synchronize()
    topo, ws = m._batch_cache[1], m._batch_cache[2]
    E = int(ws.num_edges.item())
    perm = ws.ref2dst[:E].long()
    ea = unfrag(ws.e_attr, E, precision)[perm].cpu().numpy()
    check_close("edge_encoder_and_stages ea[%s]" % case, ea, g["edge_attr"], precision)
    check_close("edge_encoder_and_stages ws.h.view1128[%s]" % case, ws.h.view(-1, 128).cpu().numpy(), g["schnet_out"], precision)
    check_close("edge_encoder_and_stages ws.hl.view1128[%s]" % case, ws.hl.view(-1, 128).cpu().numpy(), g["gin_out"], precision)
    # embedding renorm side effect (G8)
    assert rel_err(m.encoder_global.embedding.weight[:20].detach().cpu().numpy(), g["emb_rows_after"]) < 1e-6


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("case", list(FORWARD_CASES))
def test_forward_matches_reference_golden(case, precision):
    g = load_golden(case)
    cfg = FORWARD_CASES[case]()
    m, sd = _gpu_model(cfg, precision=precision)
    out = m(t(g["atom_type"]).cuda(), t(g["pos"]).cuda(), t(g["bond_index"]).cuda(), t(g["bond_type"]).cuda(),
            t(g["batch"]).cuda(), None, return_edges=True, extend_order=False)
    inv_g, inv_l, ei, et, elen, lm = [o.cpu().numpy() for o in out]
    assert ei.dtype == np.int64 and et.dtype == np.int64 and lm.dtype == np.bool_
    assert np.array_equal(ei, g["edge_index"])
    assert np.array_equal(et, g["edge_type"])
    assert np.array_equal(lm, g["local_edge_mask"])
    assert inv_g.shape == g["edge_inv_global"].shape and inv_l.shape == g["edge_inv_local"].shape
    assert rel_err(elen, g["edge_length"]) < 1e-6
    check_close("forward_matches_reference_golden inv_g[%s]" % case, inv_g, g["edge_inv_global"], precision)
    check_close("forward_matches_reference_golden inv_l[%s]" % case, inv_l, g["edge_inv_local"], precision)
    two = m(t(g["atom_type"]).cuda(), t(g["pos"]).cuda(), t(g["bond_index"]).cuda(), t(g["bond_type"]).cuda(),
            t(g["batch"]).cuda(), None, extend_order=False)
    assert len(two) == 2 and np.array_equal(two[0].cpu().numpy(), inv_g)


def test_trajectory_copied_while_sampling_equals_the_copy_at_the_end():
    """LangevinRun sends the trajectory to the host at every poll (pinned staging on a side stream) when a poll interval's chunk
    is >= 16 MiB; forced onto a small fixture (traj_overlap_min_bytes = 0, polls every 3 steps, 14 steps: full and partial chunks,
    both staging buffers) it must return the same bits as the single copy at the end, entry by entry."""
    g = load_golden(SAMPLER_CASES[0])
    cfg = sampler_case_cfg(g, SAMPLER_CASES[0])
    m, _ = _gpu_model(cfg, head_scale=float(g["head_scale"]))
    kw = sampler_case_kwargs(g)
    args = (t(g["atom_type"]).cuda(), t(g["pos_init"]).cuda(), t(g["bond_index"]).cuda(), t(g["bond_type"]).cuda(),
            t(g["batch"]).cuda(), int(g["num_graphs"]))
    n = int(g["n_steps"])
    outs = []
    for min_bytes in (1 << 40, 0):
        pos, traj = m.langevin_dynamics_sample_diffusion(*args, extend_order=False, n_steps=n, noise=t(g["noise"]).cuda(),
                                                         nan_check_every=3, traj_overlap_min_bytes=min_bytes, **kw)
        assert len(traj) == n and all(not x.is_cuda for x in traj)
        outs.append((pos.cpu(), torch.stack(traj)))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[1][1][-1], outs[1][0])


MLP_ACTS = ["gelu", "silu", "tanh", "sigmoid", "softplus", "leaky_relu", "elu", "celu", "relu6", "hardtanh", "selu", "mish", "hardswish",
            "hardsigmoid", "softsign", "logsigmoid", "hardshrink", "softshrink", "rrelu"]


# (round 5's seven activations in all three modes; round 6's nine in the two split modes -- the exact-fp32 mode shares the
# activation code and is the one that needs it least; the suite's time)
_ACT_CASES = [(a_, p_) for i, a_ in enumerate(MLP_ACTS) for p_ in PRECISIONS if i < 7 or p_ != "f32"]


@pytest.mark.parametrize("act,precision", _ACT_CASES)
def test_forward_other_mlp_act_matches_reference_golden(act, precision):
    """config.mlp_act is any torch.nn.functional name in the reference (models/common.py:62-66); both heads
    (k_pair_head, and k_pair_head_poly inside the sampler) switch on agdiff_head_params_t.act.  Forward against the reference's
    fixture per activation, and two sampler steps (polynomial head) against the oracle."""
    from agdiff_amd import qm9_model_config
    from oracle import agdiff_oracle as O
    g = load_golden("g15_forward_act_" + act)
    cfg = qm9_model_config(mlp_act=act)
    m, sd = _gpu_model(cfg, head_scale=1.0, precision=precision)
    at, bi, bt, ba = [t(g[k]) for k in ("atom_type", "bond_index", "bond_type", "batch")]
    out = m(at.cuda(), t(g["pos"]).cuda(), bi.cuda(), bt.cuda(), ba.cuda(), None, return_edges=True, extend_order=False)
    assert np.array_equal(out[2].cpu().numpy(), g["edge_index"]) and np.array_equal(out[3].cpu().numpy(), g["edge_type"])
    # sigmoid / softplus outputs do not centre at 0 (y ~ 0.5 resp. ~ 0.7): the last layer's dot product cancels (|sum w y| well
    # below sum |w y|), so the split-bf16 rounding of y (2^-16 of y, not of the result) shows amplified in the normwise figure:
    # 2.95e-5 / 2.26e-5 measured (profiles/r05_parity_errors.json) against 0.5..1.9e-5 for the centred activations.  The kernels
    # are bitwise reproducible, so this is a property of the fixture, not noise; the gate for these two in split-bf16 is 6e-5
    # (north_star: 1e-4), everything else keeps 3e-5.
    wide = 2.0 if (precision == "bf16x3" and act in ("sigmoid", "softplus", "hardsigmoid", "logsigmoid")) else 1.0      # (hardsigmoid, logsigmoid: not centred either)
    check_close("mlp_act %s inv_g" % act, out[0].cpu().numpy(), g["edge_inv_global"], precision, scale=wide)
    check_close("mlp_act %s inv_l" % act, out[1].cpu().numpy(), g["edge_inv_local"], precision)
    cfg2 = qm9_model_config(mlp_act=act, num_diffusion_timesteps=6, beta_end=2e-3)
    m2, sd2 = _gpu_model(cfg2, head_scale=1e-2, precision=precision)
    gen = torch.Generator().manual_seed(9)
    pos_init, noise = torch.randn(at.shape[0], 3, generator=gen), torch.randn(2, at.shape[0], 3, generator=gen)
    kw = dict(n_steps=2, step_lr=1e-6, w_global=1.0, global_start_sigma=float("inf"), clip=1000.0)
    pos, _ = m2.langevin_dynamics_sample_diffusion(at.cuda(), pos_init.cuda(), bi.cuda(), bt.cuda(), ba.cuda(), int(ba.max()) + 1,
                                                   extend_order=False, noise=noise.cuda(), **kw)
    ref, _ = O.langevin_dynamics_sample_diffusion(sd2, cfg2, at, pos_init, bi, bt, ba, int(ba.max()) + 1, False, noise=noise, **kw)
    check_close("mlp_act %s sampler pos" % act, pos.cpu().numpy(), ref.numpy(), precision)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("case", SAMPLER_CASES)
def test_sampler_matches_reference_golden(case, precision):
    g = load_golden(case)
    cfg = sampler_case_cfg(g, case)
    m, sd = _gpu_model(cfg, head_scale=float(g["head_scale"]), precision=precision)
    kw = sampler_case_kwargs(g)
    pos, traj = m.langevin_dynamics_sample_diffusion(
        t(g["atom_type"]).cuda(), t(g["pos_init"]).cuda(), t(g["bond_index"]).cuda(), t(g["bond_type"]).cuda(),
        t(g["batch"]).cuda(), int(g["num_graphs"]), extend_order=False, n_steps=int(g["n_steps"]),
        noise=t(g["noise"]).cuda(), **kw)
    assert pos.is_cuda and len(traj) == int(g["n_steps"]) and not traj[0].is_cuda
    check_close("sampler_matches_reference_golden traj[%s]" % case, torch.stack(traj).numpy(), g["traj"], precision)
    check_close("sampler_matches_reference_golden pos[%s]" % case, pos.cpu().numpy(), g["pos_final"], precision)
    # running the global encoder on the steps whose result is discarded changes nothing
    pos2, _ = m.langevin_dynamics_sample_diffusion(
        t(g["atom_type"]).cuda(), t(g["pos_init"]).cuda(), t(g["bond_index"]).cuda(), t(g["bond_type"]).cuda(),
        t(g["batch"]).cuda(), int(g["num_graphs"]), extend_order=False, n_steps=int(g["n_steps"]),
        noise=t(g["noise"]).cuda(), skip_discarded_global=False, save_traj=False, **kw)
    assert torch.equal(pos2, pos)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_alanine_dipeptide_config0(precision):
    """BASELINE.json configs[0] through the langevin_dynamics_sample wrapper, as
    examples/test_alanine_dipeptide.py:303-320 calls it, against the reference's own trajectory."""
    from agdiff_amd import qm9_model_config
    g = load_golden("g5_sampler_alanine")
    m, _ = _gpu_model(qm9_model_config(), precision=precision)
    pos, traj = m.langevin_dynamics_sample(
        atom_type=t(g["atom_type"]).cuda(), pos_init=t(g["pos_init"]).cuda(), bond_index=t(g["bond_index"]).cuda(),
        bond_type=t(g["bond_type"]).cuda(), batch=t(g["batch"]).cuda(), num_graphs=3, extend_order=False,
        n_steps=100, step_lr=1e-6, w_global=1.0, global_start_sigma=0.5, clip=1000.0, clip_local=None,
        sampling_type="ld", eta=1.0, noise=t(g["noise"]).cuda())
    assert len(traj) == 100
    check_close("alanine_dipeptide_config0 traj[10]", torch.stack(traj)[::10].numpy(), g["traj"], precision)
    check_close("alanine_dipeptide_config0 pos", pos.cpu().numpy(), g["pos_final"], precision)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("case", ["g10_loss_qm9", "g10_loss_drugs"])
def test_loss_matches_reference_golden(case, precision):
    """get_loss (dualenc.py:253-395) forward value, as scripts/train.py:160-170 `validate` calls it."""
    from agdiff_amd import drugs_model_config, qm9_model_config
    g = load_golden(case)
    cfg = (drugs_model_config if int(g["cfg_smooth"]) else qm9_model_config)()
    m, _ = _gpu_model(cfg, head_scale=1.0, precision=precision)
    args = (t(g["atom_type"]).cuda(), t(g["pos"]).cuda(), t(g["bond_index"]).cuda(), t(g["bond_type"]).cuda(),
            t(g["batch"]).cuda(), None, int(g["num_graphs"]))
    kw = dict(extend_order=False, time_step=t(g["time_step"]).cuda(), pos_noise=t(g["pos_noise"]).cuda())
    loss, lg, ll = m.get_loss(*args, return_unreduced_loss=True, **kw)
    assert loss.shape == lg.shape == ll.shape == (g["atom_type"].shape[0], 1) and loss.is_cuda
    check_close("loss_matches_reference_golden lg[%s]" % case, lg.cpu().numpy(), g["loss_global"], precision)
    check_close("loss_matches_reference_golden ll[%s]" % case, ll.cpu().numpy(), g["loss_local"], precision)
    check_close("loss_matches_reference_golden loss[%s]" % case, loss.cpu().numpy(), g["loss"], precision)
    only = m.get_loss(*args, **kw)
    assert torch.equal(only, loss)
    assert m.get_loss(*args, return_unreduced_edge_loss=True, **kw) is None       # dualenc.py:390-391 falls through
    # own random draws: finite, right shape, different each call
    r1 = m.get_loss(*args, extend_order=False)
    r2 = m.get_loss(*args, extend_order=False)
    assert r1.shape == loss.shape and torch.isfinite(r1).all() and not torch.equal(r1, r2)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_forward_variants_match_reference_golden(precision):
    """forward(extend_radius=False) and forward(edge_index=, edge_type=, edge_length=) -- dualenc.py:165-178."""
    from agdiff_amd import qm9_model_config
    g = load_golden("g11_forward_variants")
    m, _ = _gpu_model(qm9_model_config(), precision=precision)
    a = (t(g["atom_type"]).cuda(), t(g["pos"]).cuda(), t(g["bond_index"]).cuda(), t(g["bond_type"]).cuda(),
         t(g["batch"]).cuda(), None)
    out = m(*a, return_edges=True, extend_order=False, extend_radius=False)
    assert np.array_equal(out[2].cpu().numpy(), g["nr_edge_index"])
    assert np.array_equal(out[3].cpu().numpy(), g["nr_edge_type"]) and bool(out[5].all())
    assert rel_err(out[4].cpu().numpy(), g["nr_edge_length"]) < 1e-6
    check_close("forward_variants_match_reference_golden out[0]", out[0].cpu().numpy(), g["nr_inv_g"], precision)
    check_close("forward_variants_match_reference_golden out[1]", out[1].cpu().numpy(), g["nr_inv_l"], precision)
    ei, et, el = t(g["given_edge_index"]).cuda(), t(g["given_edge_type"]).cuda(), t(g["given_edge_length"]).cuda()
    out = m(a[0], a[1], None, None, a[4], None, edge_index=ei, edge_type=et, edge_length=el, return_edges=True)
    assert out[2] is ei and out[3] is et and out[4] is el
    assert np.array_equal(out[5].cpu().numpy(), g["given_edge_type"] > 0)
    check_close("forward_variants_match_reference_golden out[0]", out[0].cpu().numpy(), g["given_inv_g"], precision)
    check_close("forward_variants_match_reference_golden out[1]", out[1].cpu().numpy(), g["given_inv_l"], precision)
    # the sampler with extend_radius=False keeps working (local edges only; nothing for the global term to act on)
    pos, _ = m.langevin_dynamics_sample_diffusion(a[0], a[1], a[2], a[3], a[4], int(g["batch"].max()) + 1, False,
                                                  extend_radius=False, n_steps=3, w_global=1.0, global_start_sigma=1e9)
    assert torch.isfinite(pos).all()


@pytest.mark.parametrize("precision", PRECISIONS)
def test_extend_order_true_matches_reference_golden(precision):
    """SURVEY §8 f1 on the HIP path: RAW bonds, extend_order=True (forward's default, dualenc.py:153,167-177;
    _extend_graph_order, common.py:135-205) -- forward with edges and a 12-step sampler run against the reference."""
    from agdiff_amd import drugs_model_config
    g = load_golden("g12_extend_order_forward")
    m, _ = _gpu_model(drugs_model_config(num_diffusion_timesteps=12), precision=precision)
    a = (t(g["atom_type"]).cuda(), t(g["pos"]).cuda(), t(g["bond_index"]).cuda(), t(g["bond_type"]).cuda(),
         t(g["batch"]).cuda(), None)
    out = m(*a, return_edges=True)                                   # extend_order defaults to True
    assert np.array_equal(out[2].cpu().numpy(), g["edge_index"])
    assert np.array_equal(out[3].cpu().numpy(), g["edge_type"]) and (g["edge_type"] >= 23).any()
    assert np.array_equal(out[5].cpu().numpy(), g["local_edge_mask"])
    assert rel_err(out[4].cpu().numpy(), g["edge_length"]) < 1e-6
    check_close("extend_order inv_g", out[0], g["edge_inv_global"], precision)
    check_close("extend_order inv_l", out[1], g["edge_inv_local"], precision)
    pos, traj = m.langevin_dynamics_sample_diffusion(
        a[0], t(g["pos_init"]).cuda(), a[2], a[3], a[4], int(g["num_graphs"]), extend_order=True,
        n_steps=int(g["n_steps"]), step_lr=1e-6, w_global=1.0, global_start_sigma=0.5, clip=1000.0,
        noise=t(g["noise"]).cuda())
    check_close("extend_order traj", torch.stack(traj), g["traj"], precision)
    check_close("extend_order pos", pos, g["pos_final"], precision)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_node_kernel_variants(precision):
    """The SchNet node stage has three weight-delivery variants (csrc/node.hip): four waves per tile with LDS hand-offs
    for small batches, one wave per tile streaming from L2, and workgroup-shared LDS copies for large batches (the
    GIN layer has the last two).  Streaming and LDS-shared run the same MFMA order: bitwise equal in fp32 mode (the
    split-bf16 mode differs only by FMA contraction around the hi/lo split of two template instantiations); the
    four-wave variant sums the gate's dot product in another order.  All must match the reference fixture.  The
    variants are selected through agdiff_params_t.tune_* (model.tuning) and confirmed by agdiff_ws_t.variant_log."""
    from agdiff_amd import _lib
    V = _lib.DEFINES
    case = "g3_forward_drugs_capped"
    g = load_golden(case)
    m, _ = _gpu_model(FORWARD_CASES[case](), precision=precision)
    a = (t(g["atom_type"]).cuda(), t(g["pos"]).cuda(), t(g["bond_index"]).cuda(), t(g["bond_type"]).cuda(),
         t(g["batch"]).cuda(), None)

    def run(want, **tuning):
        m.tuning = dict(tuning)
        out = m(*a, return_edges=True, extend_order=False)
        var = int(m._batch_cache[2].variant_log.item())
        for name in ("NODE_SPLIT4", "NODE_STREAM", "NODE_LDSW", "GIN_LDSW", "SHARE_ROWS"):
            assert bool(var & V["AGDIFF_VAR_" + name]) == (name in want), (name, want, hex(var))
        return out
    split = run({"NODE_SPLIT4"})                                         # default for a batch this small
    stream = run({"NODE_STREAM"}, node_split_max_tiles=-1)
    shared = run({"NODE_LDSW", "GIN_LDSW"}, node_split_max_tiles=-1, node_ldsw_min_tiles=1)
    # the local edges' attribute rows: from the local branch's own encoder pass (batches this small) or written by the
    # global encoder pass through e_loc (large batches) -- the same encoder on the same lengths
    rows_shared = run({"NODE_SPLIT4", "SHARE_ROWS"} if m.packed().poly_kt == 0 else {"NODE_SPLIT4"}, share_rows_min_nodes=1)
    m.tuning = {}
    assert torch.equal(rows_shared[0], split[0]) and torch.equal(rows_shared[1], split[1])
    if precision == "f32":
        assert torch.equal(stream[0], shared[0]) and torch.equal(stream[1], shared[1])
    assert torch.equal(stream[1], split[1])                               # the local branch does not use the node stage
    assert not torch.equal(stream[0], split[0]) or precision == "f32"
    for name, v in (("split", split), ("stream", stream), ("shared", shared)):
        # (two split-bf16 evaluations against each other: each ~1.5e-5 from the exact value)
        assert rel_err(v[0].cpu().numpy(), stream[0].cpu().numpy()) < (2e-6 if precision == "f32" else 2e-5), name
        check_close("node_kernel_variants %s inv_g" % name, v[0], g["edge_inv_global"], precision)
        check_close("node_kernel_variants %s inv_l" % name, v[1], g["edge_inv_local"], precision)


def test_nan_raises_floating_point_error():
    from agdiff_amd import qm9_model_config, synth
    cfg = qm9_model_config(num_diffusion_timesteps=20)
    m, _ = _gpu_model(cfg)
    b = synth.make_packed_batch("qm9", 2, 1, seed=5)
    pos = torch.randn(b["atom_type"].shape[0], 3)
    pos[3, 1] = float("nan")
    with pytest.raises(FloatingPointError):
        m.langevin_dynamics_sample_diffusion(t(b["atom_type"]).cuda(), pos.cuda(), t(b["bond_index"]).cuda(),
                                             t(b["bond_type"]).cuda(), t(b["batch"]).cuda(), b["num_graphs"],
                                             extend_order=False, n_steps=3)


def test_oracle_parity_on_seeded_batches_and_wrapper_defaults():
    """Fresh seeded inputs (not fixtures): HIP vs oracle, QM9- and Drugs-shaped, forward and the
    langevin_dynamics_sample wrapper with its own defaults (dualenc.py:397-439).  (extend_order=True has its own
    reference fixture: test_extend_order_true_matches_reference_golden.)"""
    from agdiff_amd import drugs_model_config, qm9_model_config, synth
    from oracle import agdiff_oracle as O
    for kind, cfgf, seed, scale in (("qm9", qm9_model_config, 31, 2.0), ("drugs", drugs_model_config, 32, 1.2)):
        cfg = cfgf(num_diffusion_timesteps=10)
        m, sd = _gpu_model(cfg)
        b = synth.make_packed_batch(kind, 3, 2, seed=seed)
        at, bi, bt, ba = t(b["atom_type"]), t(b["bond_index"]), t(b["bond_type"]), t(b["batch"])
        gen = torch.Generator().manual_seed(seed)
        pos = torch.randn(at.shape[0], 3, generator=gen) * scale
        ref = O.forward(sd, cfg, at, pos, bi, bt, ba, extend_order=False)
        got = m(at.cuda(), pos.cuda(), bi.cuda(), bt.cuda(), ba.cuda(), None, return_edges=True, extend_order=False)
        assert np.array_equal(got[2].cpu().numpy(), ref[2].numpy())
        assert np.array_equal(got[3].cpu().numpy(), ref[3].numpy())
        check_close("oracle_parity_on_seeded_batches_and_wrapper_defaults got[0]", got[0].cpu().numpy(), ref[0].numpy(), "f16x3")
        check_close("oracle_parity_on_seeded_batches_and_wrapper_defaults got[1]", got[1].cpu().numpy(), ref[1].numpy(), "f16x3")
        noise = torch.randn(5, at.shape[0], 3, generator=gen)
        rpos, _ = O.langevin_dynamics_sample_diffusion(sd, cfg, at, pos, bi, bt, ba, b["num_graphs"], False,
                                                       n_steps=5, noise=noise)
        gpos, gtraj = m.langevin_dynamics_sample(at.cuda(), pos.cuda(), bi.cuda(), bt.cuda(), ba.cuda(),
                                                 b["num_graphs"], False, n_steps=5, noise=noise.cuda())
        check_close("oracle_parity_on_seeded_batches_and_wrapper_defaults gpos", gpos.cpu().numpy(), rpos.numpy(), "f16x3")
        assert len(gtraj) == 5


_BENCH_SIZE = {}


def _bench_size_oracle():
    """The oracle on a Drugs-shaped batch large enough for every large-batch kernel variant to be the NATURAL choice
    (8 molecules x 80 conformers ~ 29 k atoms, ~1 M edges at the 32-neighbour cap: >= 1,536 node tiles, >= 8,192 atoms):
    one forward(return_edges=True) at the sampler's first positions and two denoising steps with injected noise.  Computed
    once per session (about a minute of host time), shared by both precisions and both filter modes."""
    if _BENCH_SIZE:
        return _BENCH_SIZE
    import os
    from agdiff_amd import drugs_model_config, synth
    from oracle import agdiff_oracle as O
    cfg = drugs_model_config(num_diffusion_timesteps=40, beta_end=2e-5)          # sigma < 0.5 on every step: global branch on
    sd = O.synth_state_dict_for(cfg)
    b = synth.make_packed_batch("drugs", 8, 80, seed=2021)
    at, bi, bt, ba = [t(b[k]) for k in ("atom_type", "bond_index", "bond_type", "batch")]
    gen = torch.Generator().manual_seed(11)
    pos_init = torch.randn(at.shape[0], 3, generator=gen)
    noise = torch.randn(2, at.shape[0], 3, generator=gen)
    nthr = torch.get_num_threads()
    torch.set_num_threads(max(nthr, min(32, os.cpu_count() or 1)))
    try:
        with torch.no_grad():
            sig_T = O.schedule_tensors(cfg)[2][-1]
            pos0 = pos_init * sig_T
            fwd = O.forward(sd, cfg, at, pos0, bi, bt, ba, extend_order=False)
            kw = dict(extend_order=False, n_steps=2, w_global=1.0, global_start_sigma=0.5, clip=1000.0)
            ref_pos, ref_traj = O.langevin_dynamics_sample_diffusion(sd, cfg, at, pos_init, bi, bt, ba, b["num_graphs"],
                                                                     noise=noise, **kw)
    finally:
        torch.set_num_threads(nthr)
    _BENCH_SIZE.update(cfg=cfg, sd=sd, b=b, inputs=(at, bi, bt, ba), pos_init=pos_init, pos0=pos0, noise=noise, kw=kw,
                       fwd=[x.numpy() for x in fwd], ref_pos=ref_pos.numpy(), ref_traj=torch.stack(ref_traj).numpy())
    return _BENCH_SIZE


@pytest.mark.parametrize("mode", ["auto", "off"])
@pytest.mark.parametrize("precision", PRECISIONS)
def test_oracle_parity_at_natural_thresholds(precision, mode):
    """the kernels the bench times, against the ORACLE, at a size where they are chosen by the
    library's own thresholds -- no tuning override, no environment variable -- with agdiff_ws_t.variant_log asserting which
    variants ran: the per-target polynomial CFConv with its local quad tiles (`auto`) or the one-list MLP CFConv with the
    local rows shared from the global encoder pass (`off`), the LDS-shared node stage and GIN layer, the side stream.
    Matches dualenc.py:142-251 (forward) and :478-545 (two denoising steps)."""
    import os
    from agdiff_amd import _lib, get_model
    assert not [k for k in os.environ if k.startswith("AGDIFF_") and k not in ("AGDIFF_PARITY_GATE_SCALE",)], "no AGDIFF_* overrides here"
    o = _bench_size_oracle()
    cfg, b = o["cfg"], o["b"]
    m = get_model(cfg)
    m.precision, m.radius_poly = precision, mode
    m.load_state_dict({k: v.clone() for k, v in o["sd"].items()}, strict=True)
    m = m.to("cuda:0").eval()
    assert m.tuning == {}
    at, bi, bt, ba = [x.cuda() for x in o["inputs"]]
    V = _lib.DEFINES
    out = m(at, o["pos0"].cuda(), bi, bt, ba, None, return_edges=True, extend_order=False)
    topo, ws = m._batch_cache[1], m._batch_cache[2]
    assert topo.N >= 25000 and (topo.N + 15) // 16 >= 1536
    var = int(ws.variant_log.item())
    want = {"NODE_LDSW", "GIN_LDSW", "SIDE_STREAM"} | ({"CFCONV_NODE", "CFCONV_NODE_LOCAL", "ATTR_POLY"} if mode == "auto"
                                                       else {"CFCONV_FUSED", "SHARE_ROWS"})
    for name in ("CFCONV_NODE", "CFCONV_NODE_LOCAL", "CFCONV_LOCAL_MLP", "CFCONV_FUSED", "NODE_LDSW", "NODE_STREAM", "NODE_SPLIT4",
                 "GIN_LDSW", "SHARE_ROWS", "ATTR_POLY", "SIDE_STREAM", "POLY_L2_SETS"):
        assert bool(var & V["AGDIFF_VAR_" + name]) == (name in want), (name, hex(var))
    inv_g, inv_l, ei, et, el, lm = o["fwd"]
    assert np.array_equal(out[2].cpu().numpy(), ei) and np.array_equal(out[3].cpu().numpy(), et)     # bit-exact graph
    assert np.array_equal(out[5].cpu().numpy(), lm)
    assert np.diff(np.bincount(ei[1], minlength=topo.N)).size and np.bincount(ei[1], minlength=topo.N).mean() > 30      # at the cap
    tag = "natural[%s]" % mode
    check_close(tag + " edge_length", out[4], el, precision)
    # (the normwise figure is a maximum over 1.0 M edges here against a few thousand on the fixtures the gates were set on:
    # the split-bf16 error's tail reaches 3.2e-5 on the MLP path at this size, still 3x inside north_star's 1e-4)
    check_close(tag + " inv_g", out[0], inv_g, precision, scale=2.0)
    check_close(tag + " inv_l", out[1], inv_l, precision, scale=2.0)
    pos, traj = m.langevin_dynamics_sample_diffusion(at, o["pos_init"].cuda(), bi, bt, ba, b["num_graphs"],
                                                     noise=o["noise"].cuda(), **o["kw"])
    var = int(m._batch_cache[2].variant_log.item())
    assert bool(var & V["AGDIFF_VAR_HEAD_POLY"]) == (mode == "auto") and var & V["AGDIFF_VAR_NODE_LDSW"]
    check_close(tag + " traj", torch.stack(traj), o["ref_traj"], precision)
    check_close(tag + " pos", pos, o["ref_pos"], precision)


def _with_triple_bonds(rng, n):
    """synth.random_molecule with one SINGLE bond turned into a TRIPLE one (both directions) when it has a single bond between
    two atoms that sit on no other multiple bond: a sixth local edge type (1, 2, 3, 12 + the 2- / 3-hop types 23, 24)."""
    from agdiff_amd import synth
    at, r, c, ty = synth.random_molecule(rng, n)
    one = np.flatnonzero((ty == 1) & (r < c))
    if one.size:
        k = int(one[rng.integers(0, one.size)])
        ty = ty.copy()
        ty[(r == r[k]) & (c == c[k])] = 3
        ty[(r == c[k]) & (c == r[k])] = 3
    return at, r, c, ty


def test_oracle_parity_of_molecules_sliced_out_of_a_product_shape_batch():
    """ONE 196,608-atom-class Drugs-shaped batch at the library's own thresholds -- the 16-wave
    k_cfconv_quad with host-cut workgroup ranges (quad_wg_ptr), six local edge types so that one coefficient set is read from
    L2, LDS-shared node stage / GIN layer, side stream, fused front, polynomial head -- for forward(return_edges=True) and two
    denoising steps with injected noise; graphs are independent (dualenc.py:142-251 touches no other graph's rows), so the
    oracle is run on the conformers of five molecules SLICED OUT of the batch (first, middle, last, largest, smallest; first
    and last conformer of each) and gated with check_close like every fixture.  Matches dualenc.py:142-251, 478-545."""
    import os
    from agdiff_amd import _lib, drugs_model_config, get_model, synth
    from oracle import agdiff_oracle as O
    assert not [k for k in os.environ if k.startswith("AGDIFF_")], "no AGDIFF_* overrides here"
    cfg = drugs_model_config(num_diffusion_timesteps=40, beta_end=2e-5)
    sd = O.synth_state_dict_for(cfg)
    rng = np.random.default_rng(606)
    copies, parts, node_off, g_off, spans = 128, [], 0, 0, []
    while node_off < 196608:
        n = synth.sample_n_atoms(rng, "drugs")
        at1, r, c, ty = _with_triple_bonds(rng, n)
        parts.append(synth.repeat_molecule(at1, r, c, ty, copies, node_off, g_off))
        spans.append((node_off, n, g_off))
        node_off += n * copies
        g_off += copies
    at = np.concatenate([p[0] for p in parts]); bi = np.stack([np.concatenate([p[1] for p in parts]), np.concatenate([p[2] for p in parts])])
    bt = np.concatenate([p[3] for p in parts]); ba = np.concatenate([p[4] for p in parts])
    N, G = at.shape[0], g_off
    assert N >= 196608 and set(np.unique(bt)) >= {1, 2, 3, 12, 23, 24}
    gen = torch.Generator().manual_seed(17)
    pos_init = torch.randn(N, 3, generator=gen)
    noise = torch.randn(2, N, 3, generator=gen)
    pos0 = pos_init * O.schedule_tensors(cfg)[2][-1]

    m = get_model(cfg)
    m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    m = m.to("cuda:0").eval()
    assert m.tuning == {} and m.precision == "f16x3" and m.group_targets is None
    dv = [t(x).cuda() for x in (at, bi, bt, ba)]
    out = m(dv[0], pos0.cuda(), dv[1], dv[2], dv[3], None, return_edges=True, extend_order=False)
    topo, ws = m._batch_cache[1], m._batch_cache[2]
    V = _lib.DEFINES
    var = int(ws.variant_log.item())
    assert topo.Q >= 8192 and topo.quad_wg_ptr is not None and topo.group_targets == 4
    for name in ("CFCONV_NODE", "CFCONV_NODE_LOCAL", "CFCONV_NODE_QUAD", "CFCONV_NODE_FOUR", "POLY_L2_SETS", "NODE_LDSW", "GIN_LDSW",
                 "SIDE_STREAM", "ATTR_POLY"):
        assert var & V["AGDIFF_VAR_" + name], (name, hex(var))
    assert not var & (V["AGDIFF_VAR_CFCONV_FUSED"] | V["AGDIFF_VAR_CFCONV_LOCAL_MLP"])
    inv_g, inv_l, ei, et, el, lm = [x.cpu() for x in out]
    kw = dict(extend_order=False, n_steps=2, w_global=1.0, global_start_sigma=0.5, clip=1000.0)
    pos, traj = m.langevin_dynamics_sample_diffusion(dv[0], pos_init.cuda(), dv[1], dv[2], dv[3], G, noise=noise.cuda(), **kw)
    var = int(m._batch_cache[2].variant_log.item())
    for name in ("CFCONV_NODE_QUAD", "CFCONV_NODE_FOUR", "POLY_L2_SETS", "HEAD_POLY", "FUSED_FRONT", "NODE_LDSW", "GIN_LDSW", "SIDE_STREAM"):
        assert var & V["AGDIFF_VAR_" + name], (name, hex(var))
    pos, traj = pos.cpu(), torch.stack(traj)

    sizes = np.array([n for _, n, _ in spans])
    picks = sorted({0, len(spans) // 2, len(spans) - 1, int(sizes.argmax()), int(sizes.argmin())})
    gsel = np.array(sorted(g0 + k for _, _, g0 in (spans[i] for i in picks) for k in (0, copies - 1)))
    node_sel = np.isin(ba, gsel)
    new_id = np.cumsum(node_sel) - 1
    esel = node_sel[bi[0]]
    s_at, s_ba = t(at[node_sel]), t(np.searchsorted(gsel, ba[node_sel]))
    s_bi, s_bt = t(new_id[bi[:, esel]]), t(bt[esel])
    with torch.no_grad():
        ref = O.forward(sd, cfg, s_at, pos0[node_sel], s_bi, s_bt, s_ba, extend_order=False)
        rpos, rtraj = O.langevin_dynamics_sample_diffusion(sd, cfg, s_at, pos_init[node_sel], s_bi, s_bt, s_ba, len(gsel),
                                                           noise=noise[:, node_sel], **kw)
    keep = torch.from_numpy(node_sel)[ei[0]]                       # the batch's edges that belong to the sliced graphs, in order
    got_ei = torch.from_numpy(new_id)[ei[:, keep]]
    assert torch.equal(got_ei, ref[2]) and torch.equal(et[keep], ref[3]) and torch.equal(lm[keep], ref[5])     # bit-exact graph
    assert float(np.bincount(ref[2][1].numpy()).mean()) > 30                                                  # at the 32-cap
    tag = "product_shape"
    check_close(tag + " edge_length", el[keep], ref[4], "f16x3")
    check_close(tag + " inv_g", inv_g[keep], ref[0], "f16x3")
    check_close(tag + " inv_l", inv_l[keep[lm]], ref[1], "f16x3")
    check_close(tag + " traj", traj[:, node_sel], torch.stack(rtraj), "f16x3")
    check_close(tag + " pos", pos[node_sel], rpos, "f16x3")


def test_full_size_properties():
    """At a BASELINE-sized batch (Drugs-shaped, ~9k atoms, 32-cap active) the oracle is too slow to be the
    checker, so check size-independent properties: run-to-run bitwise determinism, momentum conservation
    of eq_transform (sum over a molecule of the score is ~0 before centring -> positions stay centred),
    CSR invariants and the in-degree cap."""
    from agdiff_amd import drugs_model_config, synth
    cfg = drugs_model_config(num_diffusion_timesteps=50, beta_end=2e-5)
    m, _ = _gpu_model(cfg)
    b = synth.make_packed_batch("drugs", 8, 25, seed=77)
    at, bi, bt, ba = [t(b[k]).cuda() for k in ("atom_type", "bond_index", "bond_type", "batch")]
    gen = torch.Generator().manual_seed(1)
    pos_init = torch.randn(at.shape[0], 3, generator=gen).cuda()
    noise = torch.randn(4, at.shape[0], 3, generator=gen).cuda()
    kw = dict(extend_order=False, n_steps=4, w_global=1.0, global_start_sigma=0.5, clip=1000.0, noise=noise)
    p1, tr1 = m.langevin_dynamics_sample_diffusion(at, pos_init, bi, bt, ba, b["num_graphs"], **kw)
    p2, tr2 = m.langevin_dynamics_sample_diffusion(at, pos_init, bi, bt, ba, b["num_graphs"], **kw)
    assert torch.equal(p1, p2) and torch.equal(torch.stack(tr1), torch.stack(tr2))
    m(at, p1, bi, bt, ba, None, extend_order=False)        # (the loop itself keeps no full edge list: forward() builds one)
    ws, topo = m._batch_cache[2], m._batch_cache[1]
    E = int(ws.num_edges.item())
    indeg = np.diff(ws.in_ptr.cpu().numpy())
    assert indeg.max() <= topo.max_in_degree and E <= topo.max_edges and E == indeg.sum()
    assert indeg.mean() > 30           # cap is active on Drugs-shaped compact molecules
    cen = torch.zeros(b["num_graphs"], 3, device="cuda").index_add_(0, ba, p1)
    assert float(cen.abs().max()) < 1e-3
    assert torch.isfinite(p1).all()


def test_driver_end_to_end(tmp_path):
    """scripts/test.py counterpart on a reference-layout checkpoint file and an .npz test set."""
    from agdiff_amd import Config, driver, qm9_model_config, synth
    from oracle import agdiff_oracle as O
    cfg = qm9_model_config(num_diffusion_timesteps=30)
    ckpt = str(tmp_path / "ckpt.pt")
    torch.save({"config": Config(model=cfg), "model": O.synth_state_dict_for(cfg), "iteration": 0}, ckpt)
    rng = np.random.default_rng(5)
    mols = []
    for i in range(4):
        at, r, c, ty = synth.random_molecule(rng, int(rng.integers(10, 25)))
        mols.append(dict(atom_type=at, edge_index=np.stack([r, c]), edge_type=ty, num_refs=2 + i, name="mol%d" % i))
    ts = str(tmp_path / "test.npz")
    driver.save_testset(ts, mols)
    out = str(tmp_path / "out")
    driver.main(["--ckpt", ckpt, "--testset", ts, "--out", out, "--n-steps", "6", "--max-atoms", "120", "--save-traj"])
    z = np.load(out + "/samples_all.npz")
    for i, m in enumerate(mols):
        p = z["pos_gen_%d" % i]
        assert p.shape == (2 * m["num_refs"], m["atom_type"].shape[0], 3) and np.isfinite(p).all()
        assert np.abs(p.mean(axis=1)).max() < 1e-3            # every conformer is centred (dualenc.py:542)
        assert z["traj_%d" % i].shape == (6,) + p.shape
    driver.main(["--ckpt", ckpt, "--testset", ts, "--out", out, "--n-steps", "6", "--resume"])   # nothing left to do


@pytest.mark.parametrize("case", ["g3_forward_qm9_small", "g3_forward_drugs_capped", "big"])
def test_canonical_edge_list(case):
    """agdiff_ws_t.c_*: every directed edge is either canonical or the mirror of exactly one canonical edge; a
    mirror pair has swapped end points, equal type and bitwise equal length; unpaired edges (asymmetric 32-cap) are
    canonical; the canonical list keeps the destination-sorted order."""
    from agdiff_amd import _lib, synth
    from agdiff_amd.topology import BatchTopology, Workspace
    lib = _lib.load()
    if case == "big":
        b = synth.make_packed_batch("large", 1, 1, seed=5)
        s2 = synth.make_packed_batch("drugs", 3, 2, seed=6)
        n0 = b["atom_type"].shape[0]
        at = np.concatenate([b["atom_type"], s2["atom_type"]])
        bi = np.concatenate([b["bond_index"], s2["bond_index"] + n0], axis=1)
        bt = np.concatenate([b["bond_type"], s2["bond_type"]])
        ba = np.concatenate([b["batch"], s2["batch"] + 1])
        pos = torch.randn(at.shape[0], 3, generator=torch.Generator().manual_seed(3)) * 2.5
    else:
        g = load_golden(case)
        at, bi, bt, ba, pos = g["atom_type"], g["bond_index"], g["bond_type"], g["batch"], t(g["pos"])
    topo = BatchTopology(at, bi, bt, ba, device="cuda")
    ws = Workspace(topo)
    posd = pos.cuda().contiguous()
    assert lib.agdiff_graph_build(ctypes.byref(topo.struct), ctypes.byref(ws.struct), _lib.ptr(posd),
                                  ctypes.c_float(10.0), _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    E, C = int(ws.num_edges.item()), int(ws.num_canon.item())
    src, dst = ws.e_src[:E].cpu().numpy(), ws.e_dst[:E].cpu().numpy()
    ty, ln = ws.e_type[:E].cpu().numpy(), ws.e_len[:E].cpu().numpy()
    cp, cm = ws.c_pos[:C].cpu().numpy(), ws.c_mir[:C].cpu().numpy()
    assert 0 < C <= E and np.all(np.diff(cp) > 0)
    assert np.array_equal(ws.c_src[:C].cpu().numpy(), src[cp]) and np.array_equal(ws.c_dst[:C].cpu().numpy(), dst[cp])
    assert np.array_equal(ws.c_type[:C].cpu().numpy(), ty[cp]) and np.array_equal(ws.c_len[:C].cpu().numpy(), ln[cp])
    has = cm >= 0
    m = cm[has]
    assert np.array_equal(src[m], dst[cp[has]]) and np.array_equal(dst[m], src[cp[has]])
    assert np.array_equal(ty[m], ty[cp[has]]) and np.array_equal(ln[m].view(np.int32), ln[cp[has]].view(np.int32))
    assert np.all(src[cp[has]] < dst[cp[has]])
    cover = np.zeros(E, dtype=np.int64)
    np.add.at(cover, cp, 1)
    np.add.at(cover, m, 1)
    assert np.all(cover == 1)
    # an edge without mirror really has none: its reverse is absent or carries another type
    key = {(int(a_), int(b_)): int(t_) for a_, b_, t_ in zip(src, dst, ty)}
    for e in cp[~has][:2000]:
        assert key.get((int(dst[e]), int(src[e])), -1) != int(ty[e])
    if case != "g3_forward_qm9_small":
        assert (~has).any()                # the 32-cap leaves unpaired edges
    else:
        assert C * 2 == E                  # uncapped, symmetric: exactly half


def test_graph_build_molecules_beyond_one_wave():
    """Molecules of 200 atoms (four 64-source chunks per target in csrc/graph.hip) next to small ones: edge set,
    order, types and lengths bit-exact against the oracle's radius graph, cap active (compact coordinates)."""
    from agdiff_amd import _lib, synth
    from agdiff_amd.topology import BatchTopology, Workspace
    from oracle import agdiff_oracle as O
    lib = _lib.load()
    big = synth.make_packed_batch("large", 2, 1, seed=3)
    small = synth.make_packed_batch("drugs", 2, 2, seed=4)
    n_big = big["atom_type"].shape[0]
    at = np.concatenate([big["atom_type"], small["atom_type"]])
    bi = np.concatenate([big["bond_index"], small["bond_index"] + n_big], axis=1)
    bt = np.concatenate([big["bond_type"], small["bond_type"]])
    ba = np.concatenate([big["batch"], small["batch"] + big["num_graphs"]])
    gen = torch.Generator().manual_seed(12)
    pos = torch.randn(at.shape[0], 3, generator=gen) * 3.0
    ei, et = O.extend_graph_order_radius(at.shape[0], pos, t(bi), t(bt), t(ba), cutoff=10.0, extend_order=False)
    elen = O.get_distance(pos, ei)
    topo = BatchTopology(at, bi, bt, ba, device="cuda")
    ws = Workspace(topo)
    posd = pos.cuda().contiguous()
    assert lib.agdiff_graph_build(ctypes.byref(topo.struct), ctypes.byref(ws.struct), _lib.ptr(posd),
                                  ctypes.c_float(10.0), _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    E = int(ws.num_edges.item())
    assert E == ei.shape[1] and E <= topo.max_edges
    perm = ws.ref2dst[:E].long()
    got = torch.stack([ws.e_src[:E][perm], ws.e_dst[:E][perm]]).cpu().numpy()
    assert np.array_equal(got, ei.numpy())
    assert np.array_equal(ws.e_type[:E][perm].cpu().numpy(), et.numpy())
    assert rel_err(ws.e_len[:E][perm].cpu().numpy(), elen.numpy()) < 1e-6
    indeg = np.diff(ws.in_ptr.cpu().numpy())
    assert indeg.max() > 33            # cap + bonded neighbours beyond it


@pytest.mark.parametrize("precision", PRECISIONS)
def test_default_initialised_weights(precision):
    """torch's own parameter initialisation instead of the synthetic filler (other magnitudes: SchNet embedding rows
    of norm ~11 > max_norm, kaiming-uniform linears, BatchNorm at identity): one forward against the oracle."""
    from agdiff_amd import drugs_model_config, get_model, synth
    from oracle import agdiff_oracle as O
    torch.manual_seed(1234)
    cfg = drugs_model_config()
    m = get_model(cfg)
    m.precision = precision
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to("cuda:0").eval()
    b = synth.make_packed_batch("drugs", 2, 2, seed=77)
    at, bi, bt, ba = t(b["atom_type"]), t(b["bond_index"]), t(b["bond_type"]), t(b["batch"])
    pos = torch.randn(at.shape[0], 3, generator=torch.Generator().manual_seed(5)) * 2.0
    ref = O.forward(sd, cfg, at, pos, bi, bt, ba, extend_order=False)
    got = m(at.cuda(), pos.cuda(), bi.cuda(), bt.cuda(), ba.cuda(), None, return_edges=True, extend_order=False)
    assert np.array_equal(got[2].cpu().numpy(), ref[2].numpy())
    # kaiming-uniform heads without the synthetic filler's damping: the 64 -> 1 output layer cancels to ~1 % of its terms'
    # size.  Rounds 2-3 needed scale = 3 here in split-bf16 (7.1e-5 on the local head's output); since round 4 the local
    # branch computes in split-fp16 in both split modes (1.1e-5): the plain gates hold
    check_close("default_initialised_weights got[0]", got[0].cpu().numpy(), ref[0].numpy(), precision)
    check_close("default_initialised_weights got[1]", got[1].cpu().numpy(), ref[1].numpy(), precision)
    # the max_norm renormalisation touched the module's own embedding exactly as the oracle's copy
    assert rel_err(m.encoder_global.embedding.weight.detach().cpu().numpy(), sd["encoder_global.embedding.weight"].numpy()) < 1e-6


def _scaled_relu_chains(sd, f):
    """The synthetic checkpoint with both heads' hidden layers times f and, in every GIN layer, the first layer times f and
    the second matrix divided by f (both names of every tensor: attribute path and ModuleList alias)."""
    from agdiff_amd import synth
    out = {}
    for k, v in sd.items():
        c = synth.canonical_key(k)
        if c.endswith((".layers.0.weight", ".layers.1.weight")) and "dist_mlp" in c:
            v = v * f
        elif "encoder_local.convs" in c and c.endswith(("nn.layers.0.weight", "nn.layers.0.bias")):
            v = v * f                  # (weight AND bias: with the second matrix divided by f the layer computes what it did)
        elif "encoder_local.convs" in c and c.endswith("nn.layers.1.weight"):
            v = v / f
        out[k] = v.clone()
    return out


@pytest.mark.parametrize("f", [1e-3, 1e2])
def test_head_and_gin_weights_far_from_unit_scale_keep_split_fp16_accuracy(f):
    """split-fp16 parts below 6e-5 lose bits and parts below 6e-8 vanish, so a matrix of size 1e-3 packed
    as it is carries ~3e-5 per operand, and hidden activations 1e4 x larger than usual leave fp16's range.  Both heads and the
    GIN MLPs are ReLU chains: the host normalises their matrices by exact powers of two (packing.pow2_norm; the fp32 output layer /
    the second matrix takes the factor out again).  Forward against the oracle at the plain split-fp16 gates, no fallback."""
    from agdiff_amd import drugs_model_config, get_model, synth
    from oracle import agdiff_oracle as O
    cfg = drugs_model_config()
    sd = _scaled_relu_chains(O.synth_state_dict_for(cfg), f)
    m = get_model(cfg)
    m.precision = "f16x3"
    m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    m = m.to("cuda:0").eval()
    b = synth.make_packed_batch("drugs", 3, 2, seed=78)
    at, bi, bt, ba = t(b["atom_type"]), t(b["bond_index"]), t(b["bond_type"]), t(b["batch"])
    pos = torch.randn(at.shape[0], 3, generator=torch.Generator().manual_seed(6)) * 2.0
    ref = O.forward(sd, cfg, at, pos, bi, bt, ba, extend_order=False)
    got = m(at.cuda(), pos.cuda(), bi.cuda(), bt.cuda(), ba.cuda(), None, return_edges=True, extend_order=False)
    assert (m.effective_precision, m.effective_precision_local) == ("f16x3", "f16x3")
    rep = m.packed().split_fp16_report
    assert max(rep["global"]["err"], rep["local"]["err"]) <= 2.0 ** -19
    check_close("relu_chains_x%g got[0]" % f, got[0].cpu().numpy(), ref[0].numpy(), "f16x3")
    check_close("relu_chains_x%g got[1]" % f, got[1].cpu().numpy(), ref[1].numpy(), "f16x3")


def test_weights_that_do_not_fit_split_fp16_run_in_split_bf16():
    """A matrix outside the ReLU chains whose magnitude split-fp16 cannot hold (InteractionBlock.lin times 1e-4, its BatchNorm'd
    successor... none: the block's output simply shrinks): the host measures what packing costs (packing.split_fp16_error), warns
    and packs that branch in split-bf16; the forward then meets the split-bf16 gates against the oracle."""
    from agdiff_amd import drugs_model_config, get_model, synth
    from oracle import agdiff_oracle as O
    cfg = drugs_model_config()
    sd = O.synth_state_dict_for(cfg)
    for k in list(sd):
        if synth.canonical_key(k).endswith("interactions.2.lin.weight"):
            sd[k] = sd[k] * 1e-4
    m = get_model(cfg)
    m.precision = "f16x3"
    m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    m = m.to("cuda:0").eval()
    b = synth.make_packed_batch("drugs", 3, 2, seed=79)
    at, bi, bt, ba = t(b["atom_type"]), t(b["bond_index"]), t(b["bond_type"]), t(b["batch"])
    pos = torch.randn(at.shape[0], 3, generator=torch.Generator().manual_seed(7)) * 2.0
    ref = O.forward(sd, cfg, at, pos, bi, bt, ba, extend_order=False)
    with pytest.warns(UserWarning, match="split-bf16"):
        got = m(at.cuda(), pos.cuda(), bi.cuda(), bt.cuda(), ba.cuda(), None, return_edges=True, extend_order=False)
    assert (m.effective_precision, m.effective_precision_local) == ("bf16x3", "f16x3")
    check_close("lin_x1e-4 got[0]", got[0].cpu().numpy(), ref[0].numpy(), "bf16x3")
    check_close("lin_x1e-4 got[1]", got[1].cpu().numpy(), ref[1].numpy(), "f16x3")


def test_molecule_larger_than_a_workgroup():
    """A 300-atom molecule (more atoms than the 256 threads of the per-molecule kernels: graph build, Langevin
    update, loss) next to a small one: forward and three sampler steps against the oracle."""
    from agdiff_amd import drugs_model_config, synth
    from oracle import agdiff_oracle as O
    cfg = drugs_model_config(num_diffusion_timesteps=10)
    m, sd = _gpu_model(cfg)
    rng = np.random.default_rng(21)
    a1, r1, c1, t1 = synth.random_molecule(rng, 300)
    a2, r2, c2, t2 = synth.random_molecule(rng, 23)
    at = t(np.concatenate([a1, a2]))
    bi = t(np.concatenate([np.stack([r1, c1]), np.stack([r2, c2]) + 300], axis=1))
    bt = t(np.concatenate([t1, t2]))
    ba = t(np.concatenate([np.zeros(300, dtype=np.int64), np.ones(23, dtype=np.int64)]))
    gen = torch.Generator().manual_seed(8)
    pos = torch.randn(323, 3, generator=gen) * 4.0
    ref = O.forward(sd, cfg, at, pos, bi, bt, ba, extend_order=False)
    got = m(at.cuda(), pos.cuda(), bi.cuda(), bt.cuda(), ba.cuda(), None, return_edges=True, extend_order=False)
    assert np.array_equal(got[2].cpu().numpy(), ref[2].numpy()) and np.array_equal(got[3].cpu().numpy(), ref[3].numpy())
    check_close("molecule_larger_than_a_workgroup got[0]", got[0].cpu().numpy(), ref[0].numpy(), "f16x3")
    check_close("molecule_larger_than_a_workgroup got[1]", got[1].cpu().numpy(), ref[1].numpy(), "f16x3")
    noise = torch.randn(3, 323, 3, generator=gen)
    kw = dict(n_steps=3, w_global=1.0, global_start_sigma=float("inf"), clip=1000.0)
    rpos, _ = O.langevin_dynamics_sample_diffusion(sd, cfg, at, pos, bi, bt, ba, 2, False, noise=noise, **kw)
    gpos, _ = m.langevin_dynamics_sample_diffusion(at.cuda(), pos.cuda(), bi.cuda(), bt.cuda(), ba.cuda(), 2, False,
                                                   noise=noise.cuda(), **kw)
    check_close("molecule_larger_than_a_workgroup gpos", gpos.cpu().numpy(), rpos.numpy(), "f16x3")
    ts = torch.tensor([3, 7])
    pn = torch.randn(323, 3, generator=gen)
    rl = O.get_loss_diffusion(sd, cfg, at, pos, bi, bt, ba, 2, ts, pn, extend_order=False)
    gl = m.get_loss(at.cuda(), pos.cuda(), bi.cuda(), bt.cuda(), ba.cuda(), None, 2, return_unreduced_loss=True,
                    extend_order=False, time_step=ts.cuda(), pos_noise=pn.cuda())
    for a_, b_ in zip(gl, rl):
        check_close("molecule_larger_than_a_workgroup a_", a_.cpu().numpy(), b_.numpy(), "f16x3")


@pytest.mark.parametrize("precision", PRECISIONS)
def test_ragged_and_degenerate_graphs(precision):
    """Edge cases the path must survive: a single-atom graph (no edges at all), a two-atom graph, isolated atoms far
    beyond the cutoff (zero in-degree nodes between lists), a graph larger than the 32-neighbour cap, and a
    batch whose edge count is not a multiple of the tile size."""
    from agdiff_amd import drugs_model_config
    from oracle import agdiff_oracle as O
    cfg = drugs_model_config(num_diffusion_timesteps=10)
    m, sd = _gpu_model(cfg, precision=precision)
    rng = np.random.default_rng(9)
    from agdiff_amd import synth
    at3, r3, c3, t3 = synth.random_molecule(rng, 47)
    atoms = [np.array([6]), np.array([6, 8]), at3, np.array([1, 1, 1])]
    bonds = [(np.zeros((2, 0), dtype=np.int64), np.zeros(0, dtype=np.int64)),
             (np.array([[0, 1], [1, 0]]), np.array([1, 1])),
             (np.stack([r3, c3]), t3),
             (np.array([[0, 1], [1, 0]]), np.array([2, 2]))]          # third atom of the last graph has no bond
    at_l, bi_l, bt_l, ba_l, off = [], [], [], [], 0
    for g, (a, (bi, bt)) in enumerate(zip(atoms, bonds)):
        at_l.append(a); bi_l.append(bi + off); bt_l.append(bt); ba_l.append(np.full(a.shape[0], g)); off += a.shape[0]
    at, bi, bt, ba = (t(np.concatenate(at_l)), t(np.concatenate(bi_l, axis=1)), t(np.concatenate(bt_l)),
                      t(np.concatenate(ba_l)))
    gen = torch.Generator().manual_seed(4)
    pos = torch.randn(at.shape[0], 3, generator=gen) * 1.3
    pos[-1] += 40.0                                   # isolated atom: beyond the cutoff of its graph mates
    ref = O.forward(sd, cfg, at, pos, bi, bt, ba, extend_order=False)
    got = m(at.cuda(), pos.cuda(), bi.cuda(), bt.cuda(), ba.cuda(), None, return_edges=True, extend_order=False)
    assert np.array_equal(got[2].cpu().numpy(), ref[2].numpy())
    assert np.array_equal(got[3].cpu().numpy(), ref[3].numpy())
    check_close("ragged_and_degenerate_graphs got[0]", got[0].cpu().numpy(), ref[0].numpy(), precision)
    check_close("ragged_and_degenerate_graphs got[1]", got[1].cpu().numpy(), ref[1].numpy(), precision)
    noise = torch.randn(3, at.shape[0], 3, generator=gen)
    rpos, _ = O.langevin_dynamics_sample_diffusion(sd, cfg, at, pos, bi, bt, ba, 4, False, n_steps=3, noise=noise,
                                                   w_global=1.0, global_start_sigma=float("inf"))
    gpos, _ = m.langevin_dynamics_sample_diffusion(at.cuda(), pos.cuda(), bi.cuda(), bt.cuda(), ba.cuda(), 4, False,
                                                   n_steps=3, noise=noise.cuda(), w_global=1.0,
                                                   global_start_sigma=float("inf"))
    check_close("ragged_and_degenerate_graphs gpos", gpos.cpu().numpy(), rpos.numpy(), precision)


# ---------------------------------------------------------------------------------------------------- restoring checkpoint
def _restoring_model(cfg, precision="f16x3"):
    from agdiff_amd import get_model, synth
    m = get_model(cfg)
    m.precision = precision
    m.load_state_dict(synth.restoring_state_dict(m.state_dict()), strict=True)
    return m.to("cuda:0").eval()


@pytest.mark.parametrize("precision", PRECISIONS)
def test_restoring_checkpoint_matches_reference_golden(precision):
    """Forward and sampler on the synthetic checkpoint with a restoring force (agdiff_amd/synth.py: restoring_state_dict)
    against the REFERENCE's outputs on the same weights (tests/golden/make_golden.py:g_restoring)."""
    from agdiff_amd import drugs_model_config
    g = load_golden("g14_forward_restoring")
    cfg = drugs_model_config(num_diffusion_timesteps=int(g["cfg_T"]), beta_end=float(g["cfg_beta_end"]))
    m = _restoring_model(cfg, precision)
    out = m(t(g["atom_type"]).cuda(), t(g["pos"]).cuda(), t(g["bond_index"]).cuda(), t(g["bond_type"]).cuda(),
            t(g["batch"]).cuda(), None, return_edges=True, extend_order=False)
    assert np.array_equal(out[2].cpu().numpy(), g["edge_index"]) and np.array_equal(out[3].cpu().numpy(), g["edge_type"])
    check_close("restoring forward inv_g", out[0].cpu().numpy(), g["edge_inv_global"], precision)
    check_close("restoring forward inv_l", out[1].cpu().numpy(), g["edge_inv_local"], precision)
    gs = load_golden("g14_sampler_restoring")
    pos, traj = m.langevin_dynamics_sample_diffusion(
        t(gs["atom_type"]).cuda(), t(gs["pos_init"]).cuda(), t(gs["bond_index"]).cuda(), t(gs["bond_type"]).cuda(),
        t(gs["batch"]).cuda(), int(gs["num_graphs"]), extend_order=False, n_steps=int(gs["n_steps"]),
        noise=t(gs["noise"]).cuda(), **sampler_case_kwargs(gs))
    check_close("restoring sampler traj", torch.stack(traj).numpy(), gs["traj"], precision)
    check_close("restoring sampler pos", pos.cpu().numpy(), gs["pos_final"], precision)


def test_restoring_checkpoint_keeps_molecules_compact_over_the_reference_schedule():
    """What the restoring checkpoint is for: over the reference's own schedule (sigma 12.2 -> 0.002; 500 of
    its 5000 steps, evenly spaced) Drugs-shaped molecules stay compact -- the filler alone lets them random-walk apart -- and
    end with bonded atoms a few Angstrom apart (the springs of the random topologies are frustrated: bonds want 1.5, the 2- and
    3-hop edges across them 2.5 / 3.5, and settle in between), so that the radius graph the global branch sees below
    sigma = 0.5 is the dense one a trained model would see."""
    from agdiff_amd import drugs_model_config, get_model, synth
    cfg = drugs_model_config()
    b = synth.make_packed_batch("drugs", 4, 3, seed=11)
    at, bi, bt, ba = [t(b[k]).cuda() for k in ("atom_type", "bond_index", "bond_type", "batch")]
    pos_init = torch.randn(at.shape[0], 3, generator=torch.Generator().manual_seed(4)).cuda()
    idx = np.linspace(cfg.num_diffusion_timesteps - 1, 0, 500).round().astype(int).tolist()
    kw = dict(extend_order=False, n_steps=500, step_indices=idx, w_global=1.0, global_start_sigma=0.5, clip=1000.0, save_traj=False)
    torch.manual_seed(7)
    m = _restoring_model(cfg)
    pos, _ = m.langevin_dynamics_sample_diffusion(at, pos_init, bi, bt, ba, b["num_graphs"], **kw)
    assert bool(torch.isfinite(pos).all())
    p = pos.cpu().numpy()
    bonds = b["bond_type"] < 22                            # true bonds (2-/3-hop edges carry 23 / 24)
    i, j = b["bond_index"][0][bonds], b["bond_index"][1][bonds]
    d = np.linalg.norm(p[i] - p[j], axis=1)
    assert np.abs(p).max() < 25.0, np.abs(p).max()         # compact: a 44-atom chain of 1.5 A bonds spans < 2 x 25 A
    assert 1.0 < np.median(d) < 3.2 and d.max() < 10.0, (np.median(d), d.max())
    rad_cnt = m._batch_cache[2].rad_cnt.cpu().numpy()
    assert rad_cnt.mean() > 10                              # the radius graph of the last steps is dense
    # the plain filler on the same job: atoms far apart, (almost) no radius edges left
    m2 = get_model(cfg)
    m2.load_state_dict(synth.synth_state_dict(m2.state_dict()))
    m2 = m2.to("cuda:0").eval()
    torch.manual_seed(7)
    pos2, _ = m2.langevin_dynamics_sample_diffusion(at, pos_init, bi, bt, ba, b["num_graphs"], **kw)
    assert np.abs(pos2.cpu().numpy()).max() > 4 * np.abs(p).max()


def test_split_fp16_range_watch_raises_instead_of_returning_saturated_results():
    """fp16 operands saturate at 65504.  A checkpoint whose node features explode (the filler with a 40 x sharper first encoder
    layer: |h| ~ 1e3, head outputs ~ 6e3 in the reference, exact-fp32 mode itself only good to 1e-4 there) must not come back
    with finite-but-wrong scores in the split-fp16 mode: forward raises AgdiffRangeError (an ArithmeticError) and names the
    way out; the same weights in split-bf16 (fp32's range) run."""
    from agdiff_amd import _lib, drugs_model_config, get_model, synth
    cfg = drugs_model_config(beta_end=2e-5)
    b = synth.make_packed_batch("drugs", 2, 2, seed=77)
    at, bi, bt, ba = [t(b[k]).cuda() for k in ("atom_type", "bond_index", "bond_type", "batch")]
    pos = (torch.randn(at.shape[0], 3, generator=torch.Generator().manual_seed(5)) * 2.0).cuda()

    def model(precision):
        m = get_model(cfg)
        m.precision = precision
        sd = synth.synth_state_dict(m.state_dict())
        for k in sd:
            if synth.canonical_key(k) == "edge_encoder_global.feature_expansion.weight":
                sd[k] = sd[k] * 40.0
        m.load_state_dict(sd)
        return m.to("cuda:0").eval()
    with pytest.raises(_lib.AgdiffRangeError) as e:
        model("f16x3")(at, pos, bi, bt, ba, None, return_edges=True, extend_order=False)
    assert isinstance(e.value, ArithmeticError) and "bf16x3" in str(e.value)
    m = model("bf16x3")
    m.precision_local = "bf16x3"
    out = m(at, pos, bi, bt, ba, None, return_edges=True, extend_order=False)
    assert bool(torch.isfinite(out[0]).all()) and float(out[0].abs().max()) > 100.0


def test_hidden_activations_beyond_fp16_range_are_reported_not_saturated():
    """Found in round 6 with tools/sharp_modes_probe.py: the same filler with a first encoder layer 24 x sharper keeps every WATCHED
    node tensor inside the split-fp16 range (max |h| = 203 <= 255) while the global head's hidden layer -- 128 products h_i h_j of
    up to 4e4 each -- passes 65504 and saturates as an operand of the next layer: the split-fp16 scores came back finite and 4 %
    off (exact-fp32 mode: 6e-6, split-bf16: 2e-4 against the oracle).  The kernels that convert state-dependent hidden activations
    (node stage, GIN layers, pair heads) now flag the node (agdiff_ws_t.range_rows) and forward raises AgdiffRangeError; the flags
    are cleared by the poll; split-bf16 on the same weights agrees with the exact-fp32 mode."""
    from agdiff_amd import _lib, drugs_model_config, get_model, synth
    cfg = drugs_model_config(beta_end=2e-5)
    b = synth.make_packed_batch("drugs", 2, 2, seed=77)
    at, bi, bt, ba = [t(b[k]).cuda() for k in ("atom_type", "bond_index", "bond_type", "batch")]
    pos = (torch.randn(at.shape[0], 3, generator=torch.Generator().manual_seed(5)) * 2.0).cuda()

    def model(precision, scale):
        m = get_model(cfg)
        m.precision = precision
        sd = synth.synth_state_dict(m.state_dict())
        for k in sd:
            if synth.canonical_key(k) == "edge_encoder_global.feature_expansion.weight":
                sd[k] = sd[k] * scale
        m.load_state_dict(sd)
        return m.to("cuda:0").eval()
    m = model("f16x3", 24.0)
    with pytest.raises(_lib.AgdiffRangeError) as e:
        m(at, pos, bi, bt, ba, None, return_edges=True, extend_order=False)
    assert e.value.tensor == "hidden activations" and "hidden activation" in str(e.value) and "bf16x3" in str(e.value)
    ws = m._batch_cache[2]
    assert float(ws.h.abs().max()) < 255.0 and float(ws.agg.abs().max()) < 60000.0        # (no watched tensor shows it)
    assert int(ws.range_rows.sum()) == 0 and m.range_report(ws, ba) is None                # (cleared by the poll that reported them)
    assert sorted(e.value.graphs) == sorted(set(e.value.graphs)) and set(e.value.graphs) <= set(range(int(b["num_graphs"])))
    ref = model("f32", 24.0)(at, pos, bi, bt, ba, None, return_edges=True, extend_order=False)
    mb = model("bf16x3", 24.0)
    got = mb(at, pos, bi, bt, ba, None, return_edges=True, extend_order=False)
    err = float((got[0] - ref[0]).abs().max() / ref[0].abs().max())
    assert err < 1e-3, err
    # (no false alarms: every other split-fp16 test of the suite runs with the flags live and would raise)
