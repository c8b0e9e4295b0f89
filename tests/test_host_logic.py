"""CPU: host-side logic of the product (module tree / state_dict layout, weight packing and
folding, topology building, C-ABI loading).  No GPU compute calls."""
import ctypes
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, load_golden, rel_err, t
from agdiff_amd import _lib, get_model, qm9_model_config, drugs_model_config, synth
from agdiff_amd.config import Config
from agdiff_amd.packing import PackedParams, pack_blocks, unpack_blocks, fold_bn
from agdiff_amd.topology import BatchTopology
from oracle import agdiff_oracle as O


@pytest.mark.parametrize("enc,fname,nkeys", [("mlp", "g7_state_dict_keys.txt", 854),
                                             ("gaussian", "g7_state_dict_keys_gaussian.txt", 802)])
def test_state_dict_layout_matches_reference_g7(enc, fname, nkeys):
    m = get_model(qm9_model_config(edge_encoder=enc))
    sd = m.state_dict()
    ref = [l.split() for l in open(os.path.join(GOLDEN, fname))]
    assert len(sd) == len(ref) == nkeys
    for (k, v), (rk, rshape, rdt) in zip(sd.items(), ref):
        assert k == rk
        assert ("x".join(map(str, v.shape)) or "-") == rshape, k
        assert str(v.dtype).replace("torch.", "") == rdt, k
    # aliases share storage (dualenc.py:103-108)
    assert sd["model_global.1.embedding.weight"].data_ptr() == sd["encoder_global.embedding.weight"].data_ptr()
    # strict load of a reference-layout state_dict
    m.load_state_dict(O.synth_state_dict_for(qm9_model_config(edge_encoder=enc)), strict=True)
    if enc == "gaussian":      # GaussianSmearing buffer + coeff (schnet.py:21-23)
        assert torch.equal(sd["edge_encoder_global.rbf.offset"], torch.linspace(0.0, 20.0, 64))
        assert abs(m.edge_encoder_global.rbf.coeff - (-4.9612)) < 1e-3


def test_dsm_model_type_layout_g16():
    """config.type 'dsm' (dualenc.py:127-140): the module carries `sigmas` (exp of a log-linear grid) instead of betas / alphas --
    853 keys in the reference's order --, langevin_dynamics_sample returns None for it (dualenc.py:418: only 'diffusion' is
    handled), and the reference's own forward() does not run for the type (UnboundLocalError, dualenc.py:184-186,210: recorded in
    the fixture), so forward() here refuses with a message instead of inventing semantics."""
    cfg = qm9_model_config(type="dsm", sigma_begin=10.0, sigma_end=0.01, num_noise_level=50)
    m = get_model(cfg)
    sd = m.state_dict()
    ref = [l.split() for l in open(os.path.join(GOLDEN, "g7_state_dict_keys_dsm.txt"))]
    assert len(sd) == len(ref) == 853
    for (k, v), (rk, rshape, rdt) in zip(sd.items(), ref):
        assert k == rk and ("x".join(map(str, v.shape)) or "-") == rshape and str(v.dtype).replace("torch.", "") == rdt, k
    g = np.load(os.path.join(GOLDEN, "g16_dsm.npz"))
    assert m.num_timesteps == int(g["num_timesteps"]) and np.array_equal(m.sigmas.detach().numpy(), g["sigmas"])
    assert str(g["forward"]) == "UnboundLocalError"
    z = torch.zeros(4, dtype=torch.long)
    assert m.langevin_dynamics_sample(z, torch.zeros(4, 3), torch.zeros(2, 0, dtype=torch.long), z[:0], z, 1, False) is None
    with pytest.raises(NotImplementedError, match="dsm"):
        m(z, torch.zeros(4, 3), torch.zeros(2, 0, dtype=torch.long), z[:0], z, None)


def test_factory_errors():
    with pytest.raises(NotImplementedError):
        get_model(qm9_model_config(network="nope"))
    with pytest.raises(NotImplementedError):
        get_model(qm9_model_config(edge_encoder="nope"))
    with pytest.raises(NotImplementedError):
        get_model(qm9_model_config(beta_schedule="nope"))


def test_schedule_matches_golden():
    m = get_model(qm9_model_config())
    g = load_golden("g1_schedule")
    sig = (1.0 - m.alphas).sqrt() / m.alphas.sqrt()
    assert np.array_equal(m.betas.detach().numpy()[g["idx"]], g["betas"])
    assert rel_err(sig.detach().numpy()[g["idx"]], g["sigmas"]) < 1e-6


def test_no_cpu_fallback():
    m = get_model(qm9_model_config())
    b = synth.make_packed_batch("qm9", 1, 1, seed=1)
    with pytest.raises(_lib.AgdiffHipError):
        m(t(b["atom_type"]), torch.randn(b["atom_type"].shape[0], 3), t(b["bond_index"]), t(b["bond_type"]),
          t(b["batch"]), None, extend_order=False)


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    assert lib.agdiff_abi_version() == _lib.DEFINES["AGDIFF_ABI_VERSION"]
    assert len(_lib.EXPORTS) >= 13
    for name in _lib.EXPORTS:
        assert hasattr(lib, name), name
    # argument validation happens on the host before any launch
    assert lib.agdiff_graph_build(None, None, None, ctypes.c_float(10.0), None) == -1
    assert lib.agdiff_cfconv_aggregate(None, None, None, None, ctypes.c_int64(0), 128, None, None) == -1


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("shape,kouter", [((128, 128), False), ((192, 128), True), ((8, 128), False),
                                          ((128, 8), False), ((64, 64), False), ((128, 256), True)])
def test_pack_blocks_roundtrip(shape, kouter, mode):
    rng = np.random.default_rng(0)
    W = rng.standard_normal(shape).astype(np.float32)
    flat = pack_blocks(W, kouter=kouter, mode=mode)
    assert flat.dtype == np.float32 and flat.size == ((shape[0] + 15) // 16) * ((shape[1] + 31) // 32) * 512
    R = unpack_blocks(flat, *shape, kouter=kouter, mode=mode)
    if mode == 0:
        assert np.array_equal(R, W)
    else:
        assert np.abs(R - W).max() < 2.0 ** -15 * np.abs(W).max()      # hi + lo keeps ~16 bits


def test_pack_blocks_lane_map():
    """Spot-check the documented lane map (include/agdiff_hip.h) independently of unpack_blocks."""
    W = np.arange(32 * 64, dtype=np.float32).reshape(32, 64)
    flat = pack_blocks(W, mode=0).reshape(2, 2, 2, 64, 4)             # [ot][t][u][lane][4]
    for (ot, t, u, lane, r) in [(0, 0, 0, 0, 0), (1, 1, 1, 37, 2), (0, 1, 0, 63, 3), (1, 0, 1, 16, 1)]:
        q = lane >> 4
        assert flat[ot, t, u, lane, r] == W[16 * ot + (lane & 15), 32 * t + 16 * u + 4 * q + r]


def test_folded_edge_encoder_equals_oracle():
    """The table/fold form of MLPEdgeEncoder the kernels evaluate == edge.py:84-103 as executed."""
    cfg = qm9_model_config()
    sd = O.synth_state_dict_for(cfg)
    pk = PackedParams(sd, cfg, "cpu")
    g = load_golden("g3_forward_qm9_small")
    d = g["edge_length"].astype(np.float64)[:, 0]
    ty = g["edge_type"]
    V = lambda n: pk.view(n).numpy().astype(np.float64)
    W = lambda n, o, i: unpack_blocks(pk.view(n).numpy(), o, i).astype(np.float64)
    from scipy.special import erf
    gelu = lambda x: 0.5 * x * (1 + erf(x / np.sqrt(2)))
    x0 = gelu(d[:, None] * V("ee_fe_w")[None] + V("ee_fe_b")[None])
    h1 = gelu(x0 @ W("ee_w1_pk", 128, 128).T + V("ee_t1").reshape(100, 128)[ty])
    h2 = gelu(h1 @ W("ee_w23_pk", 128, 128).T + V("ee_t3").reshape(100, 128)[ty])
    a = h2 @ W("ee_w4_pk", 128, 128).T + V("ee_b4")[None]
    assert rel_err(a, g["edge_attr"]) < 5e-6


def test_bn_fold_equals_eval_batchnorm():
    cfg = qm9_model_config()
    sd = O.synth_state_dict_for(cfg)
    p = "encoder_global.interactions.2.conv1"
    x = torch.randn(17, 128)
    ref = O._bn_eval(sd, p + ".norm1", O._lin(sd, p + ".lin1", x))
    W, b = fold_bn(sd[p + ".lin1.weight"].double().numpy(), sd[p + ".lin1.bias"].double().numpy(), sd, p + ".norm1")
    got = x.double().numpy() @ W.T + b
    assert rel_err(got, ref.numpy()) < 1e-6


def test_gelu_polynomial_mirror():
    """numpy mirror of ag_gelu (csrc/common.hpp: one-range erfc fit) against torch's erf-form gelu."""
    f = np.float32
    cp = [f(c) for c in (-1.627925070e+00, -9.181654693e-01, -1.496994283e-01, 3.089617305e-02, -3.664264106e-03,
                         1.420383199e-04)]
    x = np.concatenate([np.linspace(-12, 12, 400001), np.random.default_rng(0).standard_normal(50000) * 3]).astype(f)
    tt = np.minimum(np.abs(x) * f(0.70710678118654752440), f(4.1))
    p = np.full_like(tt, cp[5])
    for k in range(4, -1, -1):
        p = (p.astype(np.float64) * tt + cp[k]).astype(f)
    p = (p.astype(np.float64) * tt - 1.0).astype(f)
    q = np.exp2(p.astype(np.float64)).astype(f)
    # max(x, 0) - |x| q as one fma (== x * (x >= 0 ? 1 - q : q))
    got = (np.maximum(x, 0).astype(np.float64) - np.abs(x).astype(np.float64) * q.astype(np.float64)).astype(f).astype(np.float64)
    ref = torch.nn.functional.gelu(torch.from_numpy(x).double()).numpy()
    err = np.abs(got - ref)
    assert err.max() < 6e-7 and (err / np.maximum(np.abs(x), 1e-3)).max() < 2e-7


def test_topology_matches_reference_local_edges():
    g = load_golden("g3_forward_drugs_capped")
    topo = BatchTopology(g["atom_type"], g["bond_index"], g["bond_type"], g["batch"], device="cpu")
    lm = g["local_edge_mask"]
    assert np.array_equal(topo.loc_index64.numpy(), g["edge_index"][:, lm])
    assert np.array_equal(topo.loc_type64.numpy(), g["edge_type"][lm])
    assert topo.max_edges >= g["edge_index"].shape[1]
    indeg = np.bincount(g["edge_index"][1], minlength=topo.N)
    assert topo.max_in_degree >= indeg.max()
    # CSR views
    src, dst = topo.loc_src.numpy(), topo.loc_dst.numpy()
    ip, ie = topo.loc_in_ptr.numpy(), topo.loc_in_eid.numpy()
    for i in (0, 5, topo.N - 1):
        ids = ie[ip[i]:ip[i + 1]]
        assert np.all(dst[ids] == i) and np.all(np.diff(src[ids]) > 0)
    op = topo.loc_out_ptr.numpy()
    assert np.all(src[op[7]:op[8]] == 7)


def test_topology_extend_order_and_errors():
    g = load_golden("g9_extend_order")
    n = int(g["n1"])
    topo = BatchTopology(np.ones(n, dtype=np.int64), g["bond_index1"], g["bond_type1"], np.zeros(n, dtype=np.int64),
                         extend_order=True, device="cpu")
    assert np.array_equal(topo.loc_index64.numpy(), g["ext_index1"])
    assert np.array_equal(topo.loc_type64.numpy(), g["ext_type1"])
    with pytest.raises(ValueError):
        BatchTopology(np.ones(4, dtype=np.int64), np.zeros((2, 0), dtype=np.int64), np.zeros(0, dtype=np.int64),
                      np.array([0, 1, 0, 1]), device="cpu")
    with pytest.raises(ValueError):
        BatchTopology(np.ones(4, dtype=np.int64), np.array([[0], [3]]), np.array([1]), np.array([0, 0, 1, 1]), device="cpu")


def test_canonical_local_edge_list():
    """agdiff_topo_t.lc_*: every local edge is canonical or the mirror of exactly one canonical edge; a mirror has the
    swapped end points and the same type; an edge whose reverse is missing or carries another type stays canonical."""
    b = synth.make_packed_batch("drugs", 3, 2, seed=12)
    bi, bt = b["bond_index"].copy(), b["bond_type"].copy()
    # make the list asymmetric in two places: drop one direction of an edge, change the type of another's reverse
    drop = 5
    r0, c0 = bi[0][drop], bi[1][drop]
    keep = np.ones(bt.shape[0], dtype=bool)
    keep[drop] = False
    other = np.nonzero((bi[0] == bi[1][40]) & (bi[1] == bi[0][40]))[0][0]
    bt[other] = 7 if bt[other] != 7 else 8
    topo = BatchTopology(b["atom_type"], bi[:, keep], bt[keep], b["batch"], device="cpu")
    src, dst, typ = topo.loc_src.numpy(), topo.loc_dst.numpy(), topo.loc_type.numpy()
    cp, cm = topo.lc_pos.numpy(), topo.lc_mir.numpy()
    assert topo.Lc == cp.shape[0] and np.array_equal(topo.lc_src.numpy(), src[cp]) and np.array_equal(topo.lc_type.numpy(), typ[cp])
    has = cm >= 0
    assert np.array_equal(src[cm[has]], dst[cp[has]]) and np.array_equal(dst[cm[has]], src[cp[has]])
    assert np.array_equal(typ[cm[has]], typ[cp[has]]) and np.all(src[cp[has]] < dst[cp[has]])
    cover = np.zeros(topo.L, dtype=np.int64)
    np.add.at(cover, cp, 1)
    np.add.at(cover, cm[has], 1)
    assert np.all(cover == 1)
    row = topo.loc_row.numpy()                      # the l_attr_rows row of every local edge: its canonical index
    assert np.array_equal(row[cp], np.arange(topo.Lc)) and np.array_equal(row[cm[has]], np.nonzero(has)[0])
    key = {(int(a), int(c)): int(t_) for a, c, t_ in zip(src, dst, typ)}
    unpaired = cp[~has]
    assert len(unpaired) >= 3                       # the dropped edge's reverse and both directions of the retyped pair
    for e in unpaired:
        assert key.get((int(dst[e]), int(src[e])), -1) != int(typ[e])
    assert (int(c0), int(r0)) in {(int(src[e]), int(dst[e])) for e in unpaired}


# ---------------------------------------------------------------------------------- filter polynomials (host side)
def test_poly_basis_and_feature_order():
    from agdiff_amd import packing
    x = np.linspace(-1, 1, 513)
    for K in (32, 64):
        ph = packing.poly_features(x, K)
        Tn = np.polynomial.chebyshev.chebvander(x, K - 1)
        assert np.abs(ph - Tn @ packing.poly_basis_matrix(K)).max() < 1e-12          # phi = T_{8g} T_j, product rule
        assert np.linalg.cond(packing.poly_basis_matrix(K)) < 20
        order = packing.poly_feature_order(K // 32)
        assert sorted(order.tolist()) == list(range(K))
    o = packing.poly_feature_order(1)      # element j of quarter q sits at natural column 16 (j >> 2) + 4 q + (j & 3)
    assert o[0] == 0 and o[3] == 3 and o[4] == 8 and o[16] == 4 and o[31] == 31


@pytest.mark.parametrize("which", ["synth", "default_init"])
def test_filter_polynomials_reproduce_the_networks(which):
    """The accepted fits against the ORACLE's encoder + filter network evaluated in float64 at lengths that are not fit
    nodes, for a radius type and the local types of the synthetic molecules."""
    import torch
    from agdiff_amd import drugs_model_config, get_model, packing
    from oracle import agdiff_oracle as O
    cfg = drugs_model_config()
    if which == "synth":
        sd = O.synth_state_dict_for(cfg)
    else:
        torch.manual_seed(1)
        sd = get_model(cfg).state_dict()
    kt, mats, errs = packing.radius_polynomials(sd, cfg)
    assert kt == 1 and errs[1] < 1e-8
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    d = torch.rand(257, 1, dtype=torch.float64, generator=torch.Generator().manual_seed(0)) * cfg.cutoff
    inv = np.argsort(packing.poly_feature_order(1))        # phi index -> natural packed column
    ph = packing.poly_features(2.0 * d[:, 0].numpy() / cfg.cutoff - 1.0, 32)
    for typ in (0, 1, 12, 24):
        m, err = packing.fit_type(sd, cfg, typ, 1, typ == 0)
        assert err < 1e-6
        a = O.mlp_edge_encoder(sd64, "edge_encoder_global", d, torch.full((257,), typ, dtype=torch.long))
        for k in (0, cfg.num_convs - 1):
            ws = []
            for conv in ("conv1", "conv2"):
                p = "encoder_global.interactions.%d.%s" % (k, conv)
                ws.append(O._lin(sd64, p + ".nn.2", O._ssp(sd64[p + ".nn.1.beta"], O._lin(sd64, p + ".nn.0", a))))
            ref = torch.cat(ws, 1).numpy()
            got = ph @ m["conv%d.filt_poly_pk" % k][:, inv].T
            assert np.abs(got - ref).max() < 2e-6 * np.abs(ref).max()
        if typ == 0:
            ref = (a @ sd64["grad_global_dist_mlp.layers.0.weight"][:, 128:].T).numpy()
            assert np.abs(ph @ m["head_global.attr_poly_pk"][:, inv].T - ref).max() < 2e-6 * np.abs(ref).max()
        else:
            # ... and edge_attr itself BEYOND the cutoff (packing.fit_attr_far: the rows the GIN layers and the local head read for
            # bonded atoms far apart at high sigma) against the ORACLE's encoder at lengths that are not fit nodes
            cf, errf = packing.fit_attr_far(sd, cfg, typ)
            assert errf < 1e-6
            hi = packing.ATTR_FAR_FACTOR * cfg.cutoff
            dfar = cfg.cutoff + torch.rand(257, 1, dtype=torch.float64, generator=torch.Generator().manual_seed(1)) * (hi - cfg.cutoff)
            afar = O.mlp_edge_encoder(sd64, "edge_encoder_global", dfar, torch.full((257,), typ, dtype=torch.long)).numpy()
            phf = packing.poly_features(2.0 * (dfar[:, 0].numpy() - cfg.cutoff) / (hi - cfg.cutoff) - 1.0, 32)
            assert np.abs(phf @ cf[:, inv].T - afar).max() < 2e-6 * np.abs(afar).max()


def _emulate_poly_mfma(c_nat, d, cutoff, plan, np16):
    """The filter values agdiff_cfconv_node's MFMAs produce for lengths d in split-fp16, operand by operand (csrc/common.hpp
    ag_poly_features / ag_cvt_pair_mixed, csrc/nodeconv.hip mma_tiles; include/agdiff_hip.h poly_plan), kt = 1: lane
    (row r, quarter q) element j is term f = 8 q + j; products in float64, i.e. the fp32 accumulation error left out."""
    from agdiff_amd import packing
    def rtz(x):                                  # v_cvt_pkrtz_f16_f32: toward zero
        h = x.astype(np.float16)
        over = np.abs(h.astype(np.float64)) > np.abs(x)
        return np.where(over, np.nextafter(h, np.float16(0)), h).astype(np.float16)
    blk = packing.pack_blocks(c_nat, mode=2)
    blk = (packing.mix_units(blk, 1) if plan == 1 else blk).view(np.float16).reshape(-1, 2, 64, 8).astype(np.float64)
    ph = packing.poly_features(2.0 * d / cutoff - 1.0, 32).astype(np.float32).astype(np.float64)      # [M, f]
    hi = rtz(ph)
    lo = rtz(ph - hi.astype(np.float64))
    hi, lo = hi.astype(np.float64), lo.astype(np.float64)
    M, OT = d.shape[0], blk.shape[0]
    out = np.zeros((M, 16 * OT))
    for ot in range(OT):
        for q in range(4):
            f = 8 * q + np.arange(8)
            u0, u1 = blk[ot, 0, 16 * q:16 * q + 16], blk[ot, 1, 16 * q:16 * q + 16]          # [o, j]
            if plan == 0:
                acc = hi[:, f] @ u0.T + lo[:, f] @ u0.T + hi[:, f] @ u1.T
            else:
                a1 = lo[:, f] if q < 2 else hi[:, f - 16]
                acc = hi[:, f] @ u0.T + a1 @ u1.T
            out[:, 16 * ot:16 * ot + 16] += acc
    return out


@pytest.mark.parametrize("which", ["synth", "default_init"])
def test_one_pass_plan_for_the_high_terms_is_bounded(which):
    """agdiff_params_t.poly_plan 1: the host's bound (PackedParams.poly_pass_plan) against an operand-level emulation of
    the two-MFMA scheme -- the mixed unit of the packed blocks and the mixed feature operand reproduce the float64
    polynomial as closely as the three-pass scheme does, and the difference of the two stays below the bound."""
    import torch
    from agdiff_amd import drugs_model_config, get_model, packing
    from oracle import agdiff_oracle as O
    cfg = drugs_model_config()
    if which == "synth":
        sd = O.synth_state_dict_for(cfg)
    else:
        torch.manual_seed(1)
        sd = get_model(cfg).state_dict()
    pk = packing.PackedParams(sd, cfg, "cpu", "f16x3")
    assert pk.poly_kt == 1 and pk.poly_plan == 1 and pk.struct.poly_plan == 1
    assert pk.poly_errors[1] + pk.poly_high_bound["radius"] <= packing.POLY_TOL
    full = packing.PackedParams(sd, cfg, "cpu", "f16x3", poly_passes="full")
    assert full.poly_plan == 0 and full.struct.poly_plan == 0
    assert packing.PackedParams(sd, cfg, "cpu", "f32").poly_plan == 0
    d = np.random.default_rng(3).uniform(0.0, cfg.cutoff, 193)
    inv = np.argsort(packing.poly_feature_order(1))
    for k in (0, cfg.num_convs - 1):
        c = pk._poly["conv%d.filt_poly_pk" % k]
        exact = packing.poly_features(2.0 * d / cfg.cutoff - 1.0, 32) @ c[:, inv].T
        scale = np.abs(exact).max()
        three, two = (_emulate_poly_mfma(c, d, cfg.cutoff, plan, np.float16) for plan in (0, 1))
        assert np.abs(three - exact).max() < 2e-6 * scale     # (the split itself: fp32 features, fp16 lo parts down to 6e-8)
        assert np.abs(two - exact).max() < 2e-6 * scale
        assert np.abs(two - three).max() <= pk.poly_high_bound["radius"] * scale + 1e-9 * scale
        # what is packed: 2^S times the coefficients (conv[k].filt_poly_unscale = 2^-S) -- the lo parts leave the subnormals
        up = pk.filt_poly_upscale[k]
        assert 64.0 <= np.abs(c).max() * up <= 128.0 and pk.struct.conv[k].filt_poly_unscale == 1.0 / up
        assert full.struct.conv[k].filt_poly_unscale == 1.0 / up
        assert np.abs(_emulate_poly_mfma(c * up, d, cfg.cutoff, 1, np.float16) / up - exact).max() < 3e-7 * scale
    # plan and scale are the MODEL's (decided on the radius set + packing.POLY_PLAN_TYPES at pack time): slots that arrive later
    # change neither them nor a bit of the sets already packed (numerics must not depend on the batches seen before)
    rad0, up0 = pk.rad_poly_flat.clone(), list(pk.filt_poly_upscale)
    pk.ensure_local_types([1, 2])
    assert pk.poly_plan == 1 and set(pk.local_slots) == {1, 2}
    typed0 = pk.typed_flat.clone()
    pk.ensure_local_types([23, 24, 12])
    assert pk.poly_plan == 1 and pk.filt_poly_upscale == up0 and torch.equal(pk.rad_poly_flat, rad0)
    per_conv = typed0.numel() // cfg.num_convs
    new_per_conv = pk.typed_flat.numel() // cfg.num_convs
    for k in range(cfg.num_convs):
        assert torch.equal(pk.typed_flat[k * new_per_conv:k * new_per_conv + per_conv], typed0[k * per_conv:(k + 1) * per_conv])
    # a type whose high terms are too heavy for the model's plan keeps the filter MLPs for ITS edges; the plan stays
    pk.poly_high_bound["type3"] = 1.0
    assert pk.ensure_local_types([3]) is False and 3 in pk.poly_refused_types and pk.poly_plan == 1


def test_sharp_networks_need_more_terms_or_are_refused():
    from agdiff_amd import packing, qm9_model_config
    from oracle import agdiff_oracle as O
    cfg = qm9_model_config()
    sd = O.synth_state_dict_for(cfg)
    base = sd["edge_encoder_global.feature_expansion.weight"].clone()
    seen = set()
    for scale in (1.0, 12.0, 40.0, 100.0, 400.0):
        sd["edge_encoder_global.feature_expansion.weight"] = base * scale
        kt, mats, errs = packing.radius_polynomials(sd, cfg)
        assert (kt == 0) == (min(errs.values()) > packing.POLY_TOL)
        assert kt == 0 or (errs[kt] <= packing.POLY_TOL and all(errs[k] > packing.POLY_TOL for k in range(1, kt)))
        assert kt == 0 or all(m.shape[1] == 32 * kt for m in mats.values())
        seen.add(kt)
    # smooth weights: 32 terms; 12 x: 64; 40 x: 96; 100 x: 128 (the two rungs of round 6); 400 x: refused -> the filter MLPs
    assert seen == {1, 2, 3, 4, 0}, seen


def test_polynomial_features_beyond_64_terms_span_the_chebyshev_basis():
    """poly_features for K = 96, 128 (T_64 .. T_120 by the recurrence in steps of eight, as csrc/common.hpp ag_poly_features) are
    the products T_{8 g} T_j they claim to be, and poly_basis_matrix maps them onto T_0 .. T_{K-1}."""
    from agdiff_amd import packing
    x = np.cos(np.linspace(0.0, np.pi, 257))
    for K in (96, 128):
        phi = packing.poly_features(x, K)
        th = np.arccos(np.clip(x, -1, 1))
        for f in (0, 7, 63, 64, 71, 95, K - 8, K - 1):
            g, j = divmod(f, 8)
            assert np.abs(phi[:, f] - np.cos(8 * g * th) * np.cos(j * th)).max() < 1e-9, (K, f)
        cheb = np.stack([np.cos(n * th) for n in range(K)], axis=-1)
        assert np.abs(cheb @ packing.poly_basis_matrix(K) - phi).max() < 1e-9
        order = packing.poly_feature_order(K // 32)
        assert sorted(order.tolist()) == list(range(K))


def test_group_order_is_a_permutation_and_beats_the_plain_sort():
    """agdiff_group_order (host function of the library behind BatchTopology's target groups): a permutation of the molecule's
    atoms, deterministic, never more tiles than cutting the lexicographically sorted need vectors into runs of GT, and fewer on
    the synthetic Drugs-shaped molecules; GT = 1 and molecules of at most GT atoms keep the sorted order."""
    from agdiff_amd import synth, topology
    from agdiff_amd.topology import BatchTopology
    rng = np.random.default_rng(11)
    tot_sort = tot_new = 0
    for trial in range(12):
        at_, r_, c_, t_ = synth.random_molecule(rng, synth.sample_n_atoms(rng, "drugs"))
        b = dict(atom_type=at_, bond_index=np.stack([r_, c_]), bond_type=t_, batch=np.zeros(at_.shape[0], dtype=np.int64))
        tp = BatchTopology(b["atom_type"], b["bond_index"], b["bond_type"], b["batch"], device="cpu", group_targets=4,
                           radius_column=False)
        dst, typ, N = tp.loc_dst.numpy(), tp.loc_type.numpy(), tp.N
        types = np.unique(typ)
        cnt = np.stack([np.bincount(dst[typ == ty], minlength=N) for ty in types], 1)
        for gt in (4, 2, 1):
            need = ((cnt + 16 // gt - 1) // (16 // gt)).astype(np.int32)
            cost = lambda order: sum(need[order[j:j + gt]].max(axis=0).sum() for j in range(0, N, gt))
            plain = np.lexsort(tuple(need[:, k] for k in range(need.shape[1] - 1, -1, -1)))
            topology._GROUP_ORDER_CACHE.clear()
            o1 = topology._group_order(need, gt)
            topology._GROUP_ORDER_CACHE.clear()
            o2 = topology._group_order(need, gt)
            assert np.array_equal(o1, o2) and sorted(o1.tolist()) == list(range(N))
            assert cost(o1) <= cost(plain)
            if gt == 1:
                assert cost(o1) == cost(plain)
            if gt == 4:
                tot_sort += cost(plain); tot_new += cost(o1)
                assert cost(o1) == tp.T          # (what the topology built)
    assert tot_new < 0.97 * tot_sort, (tot_new, tot_sort)
    # with the radius column (the default on quads: k_cfconv_quad walks max over a quad's targets of ceil(radius rows / 4) tiles)
    # the grouping trades a few local tiles for fewer radius tiles: local + radius tiles of a compact molecule do not grow
    rng = np.random.default_rng(12)
    tot = {False: 0, True: 0}
    for trial in range(8):
        n = synth.sample_n_atoms(rng, "drugs")
        at_, r_, c_, t_ = synth.random_molecule(rng, n)
        for col in (False, True):
            tp = BatchTopology(at_, np.stack([r_, c_]), t_, np.zeros(n, dtype=np.int64), device="cpu", group_targets=4, radius_column=col)
            m = min(n, 33)
            cand = np.where(np.arange(n) < m, m - 1, m)
            src, dst = tp.loc_src.numpy(), tp.loc_dst.numpy()
            rad = cand - np.bincount(dst[src < m], minlength=n)
            qt = tp.quad_tgt.numpy().reshape(-1, 4)
            assert sorted(qt[qt >= 0].tolist()) == list(range(n))
            tot[col] += tp.T + int(((np.where(qt >= 0, rad[np.maximum(qt, 0)], 0) + 3) // 4).max(axis=1).sum())
    assert tot[True] < tot[False], tot
    tiny = np.array([[1, 0], [0, 2], [1, 1]], dtype=np.int32)
    assert sorted(topology._group_order(tiny, 4).tolist()) == [0, 1, 2]


def test_padded_local_list():
    """agdiff_topo_t.lp_*: every target's local list padded to at least 8 entries, real entries first and in in-slot order."""
    from agdiff_amd import synth
    from agdiff_amd.topology import BatchTopology
    b = synth.make_packed_batch("drugs", 3, 2, seed=5)
    tp = BatchTopology(b["atom_type"], b["bond_index"], b["bond_type"], b["batch"], device="cpu")
    lp = tp.lp_ptr.numpy(); src, dst, row = tp.lp_src.numpy(), tp.lp_dst.numpy(), tp.lp_row.numpy()
    ip, isrc, irow = tp.loc_in_ptr.numpy(), tp.loc_in_src.numpy(), tp.loc_in_row.numpy()
    assert lp[-1] == tp.Lp and tp.struct.num_local_padded == tp.Lp
    for i in range(tp.N):
        dg = ip[i + 1] - ip[i]
        assert lp[i + 1] - lp[i] == (max(dg, 8) if dg else 0)
        assert np.array_equal(src[lp[i]:lp[i] + dg], isrc[ip[i]:ip[i + 1]]) and np.all(dst[lp[i]:lp[i + 1]] == i)
        assert np.array_equal(row[lp[i]:lp[i] + dg], irow[ip[i]:ip[i + 1]]) and np.all(row[lp[i] + dg:lp[i + 1]] == -1)
        assert np.all(src[lp[i] + dg:lp[i + 1]] == i)
    pp, pm = tp.lc_ppos.numpy(), tp.lc_pmir.numpy()
    assert np.array_equal(src[pp], tp.lc_src.numpy()) and np.array_equal(dst[pp], tp.lc_dst.numpy())
    mk = pm >= 0
    assert np.array_equal(src[pm[mk]], tp.lc_dst.numpy()[mk]) and np.array_equal(dst[pm[mk]], tp.lc_src.numpy()[mk])
    cover = np.zeros(tp.Lp, bool); cover[pp] = True; cover[pm[mk]] = True
    assert np.array_equal(cover, row >= 0)


@pytest.mark.parametrize("gt", [4, 2, 1])
@pytest.mark.parametrize("kind,mols,copies", [("drugs", 3, 2), ("qm9", 5, 1)])
def test_local_quad_tiles(kind, mols, copies, gt):
    """agdiff_topo_t.quad_tgt / lt_* (agdiff_cfconv_node), for GT = 4, 2, 1 targets per group (RT = 16 / GT rows per target and
    tile): every atom sits in exactly one group, the atoms of a group belong to one molecule (unused entries are -1); a tile
    holds rows of ONE edge type, rows RT k .. RT k + RT - 1 of tile u of that type are in-edges [RT u, RT u + RT) -- of that
    type, in source order -- of the group's k-th target; a group has max over its targets of ceil(in-edges of the type / RT)
    tiles per type, types ascending; pad rows point at the target itself (a missing target: the group's first) and carry
    the tile's type; every local edge sits in exactly one row.  The default group size follows the batch size."""
    from agdiff_amd import synth
    from agdiff_amd.topology import BatchTopology
    b = synth.make_packed_batch(kind, mols, copies, seed=5)
    tp = BatchTopology(b["atom_type"], b["bond_index"], b["bond_type"], b["batch"], device="cpu", group_targets=gt)
    rt = 16 // gt
    assert tp.struct.group_targets == gt
    auto = BatchTopology(b["atom_type"], b["bond_index"], b["bond_type"], b["batch"], device="cpu")
    assert auto.group_targets == (4 if tp.N > 6144 else 2 if tp.N > 3072 else 1)
    qt, ltp, src, typ = tp.quad_tgt.numpy().reshape(-1, 4), tp.lt_ptr.numpy(), tp.lt_src.numpy(), tp.lt_type.numpy()
    lsrc, ldst, ltyp, ba = tp.loc_src.numpy(), tp.loc_dst.numpy(), tp.loc_type.numpy(), b["batch"]
    Q = tp.Q
    assert qt.shape[0] == Q == tp.struct.num_quads and ltp.shape[0] == Q + 1 and ltp[-1] == tp.T == tp.struct.num_local_tiles
    assert sorted(qt[qt >= 0].tolist()) == list(range(tp.N)) and np.all(qt[:, 0] >= 0)
    sizes = np.bincount(ba)
    assert np.all(qt[:, gt:] < 0) and int((qt[:, :gt] < 0).sum()) == int(((gt - sizes % gt) % gt).sum())
    seen = np.zeros(tp.L, int)
    tiles_per_target = []
    for p_ in range(Q):
        tg = qt[p_]
        live = tg[tg >= 0]
        assert np.all(ba[live] == ba[live[0]]) and np.all(np.diff((tg >= 0).astype(int)) <= 0)      # one molecule; -1 only at the end
        want_tiles = []
        types_here = np.unique(ltyp[np.isin(ldst, live)])
        for ty in types_here:
            per = [int(((ldst == t_) & (ltyp == ty)).sum()) for t_ in live]
            want_tiles += [(int(ty), u) for u in range((max(per) + rt - 1) // rt)]
        assert ltp[p_ + 1] - ltp[p_] == len(want_tiles)
        for tl, (ty, u) in zip(range(ltp[p_], ltp[p_ + 1]), want_tiles):
            assert np.all(typ[16 * tl:16 * tl + 16] == ty)
            for k in range(gt):
                rows = np.arange(16 * tl + rt * k, 16 * tl + rt * k + rt)
                tgt = int(tg[k])
                if tgt < 0:
                    assert np.all(src[rows] == tg[0]) and not tp.lt_real[rows].any()
                    continue
                e = np.nonzero((ldst == tgt) & (ltyp == ty))[0]
                e = e[np.argsort(lsrc[e], kind="stable")][rt * u:rt * u + rt]
                n = e.size
                assert np.array_equal(tp.lt_eid[rows[:n]], e) and np.array_equal(src[rows[:n]], lsrc[e]) and tp.lt_real[rows[:n]].all()
                assert np.all(src[rows[n:]] == tgt) and not tp.lt_real[rows[n:]].any() and np.all(tp.lt_eid[rows[n:]] == -1)
                seen[e] += 1
        tiles_per_target.append((ltp[p_ + 1] - ltp[p_]) / max(live.size, 1))
    assert np.all(seen == 1)
    # the mapping the per-step kernels write lengths and scales through
    lc_pos, lc_mir = tp.lc_pos.numpy(), tp.lc_mir.numpy()
    tpos, tmir = tp.lc_tpos.numpy(), tp.lc_tmir.numpy()
    assert np.array_equal(tp.lt_eid[tpos], lc_pos)
    assert np.array_equal(tp.lt_eid[tmir[lc_mir >= 0]], lc_mir[lc_mir >= 0]) and np.all(tmir[lc_mir < 0] == -1)
    # grouping by needs keeps the padding moderate: about one tile per target on these molecules (pair tiles took ~2 type rounds)
    assert gt != 4 or tp.T / tp.N < 1.35


def test_distance_weighting_segments():
    """packing.dist_segments (agdiff_conv_params_t.dist_seg): the piecewise-linear form of DistanceWeightingNetwork before its
    sigmoid equals layer2(relu(layer1(d))) (schnet.py:83-100) at random lengths and at the kinks themselves, also with
    dead hidden units (w1 = 0) and against the oracle's cfconv weights."""
    from agdiff_amd.packing import dist_segments
    rng = np.random.default_rng(0)
    for trial in range(6):
        w1, b1, w2 = rng.uniform(-1, 1, 32), rng.uniform(-1, 1, 32), rng.uniform(-1, 1, 32)
        b2 = float(rng.uniform(-1, 1))
        if trial == 3:
            w1[:5] = 0.0
        if trial == 4:
            w1[:] = np.abs(w1); b1[:] = np.abs(b1)          # no kink inside d >= 0
        seg = dist_segments(w1, b1, w2, b2)
        bp, al, be = seg[:32], seg[32:65], seg[65:98]
        assert np.all(np.diff(bp[np.isfinite(bp)]) >= 0)
        d = np.concatenate([rng.uniform(0, 12, 2000), bp[np.isfinite(bp)], [0.0]])
        s = (bp[None, :] <= d[:, None]).sum(1)               # the kernel's binary search computes this count
        got = al[s] * d + be[s]
        ref = (w2[None, :] * np.maximum(w1[None, :] * d[:, None] + b1[None, :], 0)).sum(1) + b2
        assert np.abs(got - ref).max() < 1e-12
    cfg = drugs_model_config()
    sd = O.synth_state_dict_for(cfg)
    p = "encoder_global.interactions.2.conv2.distance_weighting"
    seg = dist_segments(sd[p + ".layer1.weight"][:, 0].numpy(), sd[p + ".layer1.bias"].numpy(), sd[p + ".layer2.weight"][0].numpy(),
                        float(sd[p + ".layer2.bias"][0]))
    d = torch.linspace(0, 11, 501, dtype=torch.float64)
    ref = torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(d.view(-1, 1), sd[p + ".layer1.weight"].double(),
                                                                             sd[p + ".layer1.bias"].double())),
                                     sd[p + ".layer2.weight"].double(), sd[p + ".layer2.bias"].double())[:, 0].numpy()
    s = (seg[None, :32] <= d.numpy()[:, None]).sum(1)
    assert np.abs(seg[32:65][s] * d.numpy() + seg[65:98][s] - ref).max() < 1e-12


def test_dist_union_table_selects_the_lines_of_the_per_conv_tables():
    """agdiff_params_t.dist_union (packing.dist_union_table): for any length d in [0, cutoff] the union segment u = number of
    union kinks <= d carries, for every conv, exactly the (alpha, beta) floats that conv's own table (dist_segments: s = number
    of its kinks <= d) selects -- so the fused front's ONE search gives bit-identical scales.  The union holds the kinks in
    (0, cutoff] only: K slots (a power of two, +inf padded), S = their number + 1 segments."""
    from agdiff_amd.packing import dist_segments, dist_union_table
    rng = np.random.default_rng(3)
    cutoff = 10.0
    segs = []
    for k in range(6):
        two = []
        for h in range(2):
            w1 = rng.normal(size=32)
            w1[rng.integers(0, 32, 3)] = 0.0                      # (units without a kink)
            b1 = rng.normal(size=32) * 3.0
            if k == 2:
                b1[:4] = -w1[:4] * 2.5                             # (coinciding kinks, also across convs below)
            if k == 4:
                b1[5], b1[6] = 0.0, -w1[6] * cutoff                # (a kink at exactly 0 and one at exactly the cutoff)
            two.append(dist_segments(w1, b1, rng.normal(size=32), float(rng.normal())))
        segs.append(np.concatenate(two))
    segs[3][:32] = segs[2][:32]                                    # conv1 of block 3 shares every kink with block 2's
    tab, K, S = dist_union_table(segs, cutoff)
    n = 12
    kinks = tab[:K]
    fin = kinks[np.isfinite(kinks)]
    assert K & (K - 1) == 0 and 2 <= K <= 512 and S == fin.size + 1 <= K and tab.shape[0] == K + S * 2 * n
    assert np.all(np.diff(fin) > 0) and fin.min() > 0 and fin.max() <= np.float32(cutoff)
    rows = [np.asarray(segs[k], np.float64).astype(np.float32).reshape(2, 100)[h] for k in range(6) for h in (0, 1)]
    every = np.concatenate([r[:32][np.isfinite(r[:32])] for r in rows])
    assert fin.size == np.unique(every[(every > 0) & (every <= np.float32(cutoff))]).size < np.unique(every).size
    ds = np.concatenate([rng.uniform(0.0, cutoff, 4000).astype(np.float32), fin, np.nextafter(fin, np.float32(-np.inf)),
                         np.float32([0.0, cutoff])])
    for d in ds:
        # the kernel's search: steps K/2 .. 1 over the +inf padded slots
        u, step = 0, K >> 1
        while step >= 1:
            u += step if kinks[u + step - 1] <= d else 0
            step >>= 1
        assert u == int(np.count_nonzero(kinks <= d)) < S
        for cc, r in enumerate(rows):
            s_ = int(np.count_nonzero(r[:32] <= d))
            assert tab[K + (u * n + cc) * 2] == r[32 + s_] and tab[K + (u * n + cc) * 2 + 1] == r[65 + s_], (d, cc)
    # beyond the cutoff the search still lands on a stored (finite) line: the envelope zeroes the scale there
    for d in np.float32([cutoff * 1.0001, 50.0, 1e6]):
        u = int(np.count_nonzero(kinks <= d))
        assert u == S - 1 and np.isfinite(tab[K + u * n * 2: K + (u + 1) * n * 2]).all()


def test_static_local_adjacency_masks():
    """agdiff_topo_t.loc_bits: bit (src - first atom of the molecule) of row dst for every local edge, rows of
    2 ceil(max atoms per molecule / 64) words -- what agdiff_sampler_front copies into LDS instead of walking the in-lists."""
    from agdiff_amd import synth, topology
    b = synth.make_packed_batch("drugs", 3, 4, seed=5)
    tp = topology.BatchTopology(b["atom_type"], b["bond_index"], b["bond_type"], b["batch"], b["num_graphs"], device="cpu")
    W = 2 * ((tp.max_atoms + 63) // 64)
    bits = tp.loc_bits.numpy().view(np.uint32).reshape(tp.N, W)
    src, dst, gp, ba = tp.loc_src.numpy(), tp.loc_dst.numpy(), tp.graph_ptr.numpy(), tp.batch64.numpy()
    ref = np.zeros((tp.N, W), np.uint32)
    for s_, d_ in zip(src, dst):
        j = int(s_ - gp[ba[s_]])
        ref[d_, j >> 5] |= np.uint32(1 << (j & 31))
    assert np.array_equal(bits, ref) and int(np.unpackbits(bits.view(np.uint8)).sum()) == tp.L


def test_workgroup_ranges_of_like_tile_counts():
    """agdiff_topo_t.quad_wg_ptr: 256 contiguous quad ranges covering every quad once, cut so that the tiles a workgroup of
    k_cfconv_quad walks (local tiles + the radius tiles of a compact molecule) are balanced better than by equal quad counts."""
    from agdiff_amd import synth, topology
    b = synth.make_packed_batch("drugs", 12, 40, seed=7)
    tp = topology.BatchTopology(b["atom_type"], b["bond_index"], b["bond_type"], b["batch"], b["num_graphs"], device="cpu", group_targets=4)
    assert tp.Q >= 256 and tp.quad_wg_ptr is not None
    w = tp.quad_wg_ptr.numpy().astype(np.int64)
    assert w.shape == (257,) and w[0] == 0 and w[-1] == tp.Q and np.all(np.diff(w) >= 0)
    # estimated tiles per quad as the topology counts them
    N, gp, ba = tp.N, tp.graph_ptr.numpy().astype(np.int64), tp.batch64.numpy()
    n_of = np.diff(gp)[ba]
    li = np.arange(N) - gp[ba]
    m = np.minimum(n_of, 33)
    cand = np.where(li < m, m - 1, m)
    src, dst = tp.loc_src.numpy().astype(np.int64), tp.loc_dst.numpy().astype(np.int64)
    cnt = cand - np.bincount(dst[(src - gp[ba[src]]) < m[dst]], minlength=N)
    qt = tp.quad_tgt.numpy().astype(np.int64).reshape(-1, 4)
    tiles = np.diff(tp.lt_ptr.numpy().astype(np.int64)) + ((np.where(qt >= 0, cnt[np.maximum(qt, 0)], 0).max(axis=1) + 3) // 4)
    cs = np.concatenate([[0], np.cumsum(tiles)])
    by_tiles = cs[w[1:]] - cs[w[:-1]]
    per = (tp.Q + 255) // 256
    eq = np.minimum(np.arange(257) * per, tp.Q)
    by_quads = cs[eq[1:]] - cs[eq[:-1]]
    # (a range ends at a quad boundary: within one quad's tiles of the mean, and never worse than equal quad counts)
    assert by_tiles.max() - by_tiles.mean() <= tiles.max() and by_tiles.min() >= by_tiles.mean() - tiles.max()
    assert by_tiles.max() <= by_quads.max()
    small = topology.BatchTopology(b["atom_type"][:60], np.zeros((2, 0), np.int64), np.zeros(0, np.int64), np.zeros(60, np.int64), 1, device="cpu")
    assert small.quad_wg_ptr is None


def test_topology_struct_points_at_its_own_tensors():
    """Every pointer field of agdiff_topo_t holds the address of the BatchTopology tensor of the same name (BatchTopology.to
    moves the tensors and re-points the struct: a field it forgot would keep pointing at the old device's memory)."""
    import ctypes
    from agdiff_amd import _lib, synth, topology
    b = synth.make_packed_batch("drugs", 12, 40, seed=7)
    tp = topology.BatchTopology(b["atom_type"], b["bond_index"], b["bond_type"], b["batch"], b["num_graphs"], device="cpu", group_targets=4)
    checked = 0
    for name, ctype in _lib.Topo._fields_:
        if ctype is not ctypes.c_void_p:
            continue
        val = getattr(tp.struct, name)
        t = getattr(tp, name, None)
        assert t is not None or not val, name
        if t is not None:
            assert val == t.data_ptr(), name
            checked += 1
    assert checked >= 30 and tp.struct.quad_wg_ptr and tp.struct.loc_bits
    assert tp.to("cpu") is tp                      # (same device: nothing moves)


def test_prepared_topology_is_tied_to_its_batch():
    """a BatchTopology prepared ahead is only accepted for the batch it was built from -- same atoms, graphs, bonds
    (as passed), extend_order -- not for any batch with the same atom count.  The fingerprint is the same number from numpy
    arrays and from torch tensors."""
    import torch
    from agdiff_amd import synth, topology
    b = synth.make_packed_batch("drugs", 3, 4, seed=1)
    tp = topology.BatchTopology(b["atom_type"], b["bond_index"], b["bond_type"], b["batch"], b["num_graphs"], device="cpu")
    tt = lambda x: torch.from_numpy(x)
    fp = topology.batch_fingerprint(tt(b["atom_type"]), tt(b["bond_index"]), tt(b["bond_type"]), tt(b["batch"]), b["num_graphs"], False)
    assert fp == tp.fingerprint
    other = b["bond_type"].copy()
    other[5] = 2 if other[5] != 2 else 1
    swapped = b["bond_index"][:, ::-1].copy()                   # the same set of bonds in another order: another input
    moved = b["batch"].copy()
    edge = int(np.flatnonzero(np.diff(moved))[0]) + 1           # first atom of graph 1 ...
    moved[edge] = moved[edge - 1]                               # ... now belongs to graph 0
    for args in ((b["atom_type"], b["bond_index"], other, b["batch"], b["num_graphs"], False),
                 (b["atom_type"], swapped, b["bond_type"], b["batch"], b["num_graphs"], False),
                 (b["atom_type"], b["bond_index"], b["bond_type"], b["batch"], b["num_graphs"], True),
                 (b["atom_type"], b["bond_index"], b["bond_type"], moved, b["num_graphs"], False)):
        assert topology.batch_fingerprint(*args) != tp.fingerprint
    from agdiff_amd import get_model, qm9_model_config
    m = get_model(qm9_model_config())
    with pytest.raises(ValueError, match="another batch"):
        m._batch(tt(b["atom_type"]), tt(b["bond_index"]), tt(other), tt(b["batch"]), b["num_graphs"], False, topology=tp)
    m.group_targets = 4 if tp.group_targets != 4 else 2
    with pytest.raises(ValueError, match="targets per wave"):
        m._batch(tt(b["atom_type"]), tt(b["bond_index"]), tt(b["bond_type"]), tt(b["batch"]), b["num_graphs"], False, topology=tp)


def test_pass_plan_of_the_96_and_128_term_rungs():
    """Three and four k-tiles (sharp first layers): plan 1 gives every term from 32 on one pass, plans 2 and 3 (new with these
    rungs) the terms from 64 / 96 on; the host takes the first whose bound holds for the radius set and the common local types, and what it records
    is that plan's bound.  Checked on the bounded-sharpness family (gelu(s u) / s: the first layer's weight and bias times s, the
    next layer's columns that read it divided by s)."""
    from agdiff_amd import drugs_model_config, packing
    from oracle import agdiff_oracle as O
    cfg = drugs_model_config()
    seen = {}
    for precision, scale in (("f16x3", 32.0), ("f16x3", 64.0), ("bf16x3", 32.0), ("bf16x3", 64.0)):
        sd = O.synth_state_dict_for(cfg)
        for e in ("edge_encoder_global", "model_global.0"):
            n = sd[e + ".feature_expansion.weight"].shape[0]
            sd[e + ".feature_expansion.weight"] = sd[e + ".feature_expansion.weight"] * scale
            sd[e + ".feature_expansion.bias"] = sd[e + ".feature_expansion.bias"] * scale
            w = sd[e + ".edge_feature_mlp.0.weight"].clone()
            w[:, :n] = w[:, :n] / scale
            sd[e + ".edge_feature_mlp.0.weight"] = w
        pk = packing.PackedParams(sd, cfg, "cpu", precision)
        kt = pk.poly_kt
        assert kt == {32.0: 3, 64.0: 4}[scale], pk.poly_errors
        bounds = {p: pk._bound(pk._poly, p) for p in range(1, kt)}
        assert all(bounds[p + 1] < bounds[p] for p in range(1, kt - 1))                 # (fewer terms in one pass)
        plan = pk.poly_plan
        assert 0 <= plan < kt and pk.struct.poly_plan == plan
        if plan:
            assert pk.poly_errors[kt] + bounds[plan] <= packing.POLY_TOL and pk.poly_high_bound["radius"] == bounds[plan]

        def fits(p):        # (the radius set and every common local type under plan p)
            sets = [(pk.poly_errors[kt], pk._poly)] + [packing.fit_type(sd, cfg, t, kt, False)[::-1] for t in packing.POLY_PLAN_TYPES]
            return all(err > packing.POLY_TOL or err + pk._bound(m, p) <= packing.POLY_TOL for err, m in sets)
        assert not any(fits(p) for p in range(1, plan or kt))                          # (no earlier plan would have done)
        seen[precision, scale] = pk.poly_plan
        assert packing.PackedParams(sd, cfg, "cpu", precision, poly_passes="full").poly_plan == 0
    # (split-bf16 rounds an operand to 8 bits instead of 11: the terms 32..63, and at 128 terms 64..95 too, are then too heavy for one pass)
    assert seen == {("f16x3", 32.0): 1, ("f16x3", 64.0): 2, ("bf16x3", 32.0): 2, ("bf16x3", 64.0): 3}, seen


def test_length_only_activations_beyond_fp16_range_send_the_branch_to_split_bf16():
    """The MLP edge encoder's and the filter networks' activations depend on an edge's length and type alone: their range is a
    property of the checkpoint.  When one of them would saturate as a split-fp16 operand (65504) the pack-time report says so --
    DualEncoderEpsNetwork.packed() then packs the global branch in split-bf16, as for weights the mode cannot hold -- although every
    MATRIX of the checkpoint fits the mode."""
    from agdiff_amd import drugs_model_config, packing
    from oracle import agdiff_oracle as O
    cfg = drugs_model_config()
    sd = O.synth_state_dict_for(cfg)
    pk = packing.PackedParams(sd, cfg, "cpu", "f16x3")
    rep = pk.split_fp16_report["global"]
    assert pk.encoder_activation_max < 1e3 and not rep["clipped"] and rep["err"] <= packing.SPLIT_FP16_MAX_ERR
    assert abs(packing.encoder_activation_max(sd, cfg) - pk.encoder_activation_max) < 1e-9
    sd["edge_encoder_global.feature_expansion.weight"] = sd["edge_encoder_global.feature_expansion.weight"] * 3000.0
    pk = packing.PackedParams(sd, cfg, "cpu", "f16x3", radius_poly="off")
    rep = pk.split_fp16_report["global"]
    assert pk.encoder_activation_max > packing.SPLIT_FP16_ACT_LIMIT and rep["clipped"] and rep["activations"] == pk.encoder_activation_max
    assert rep["err"] <= packing.SPLIT_FP16_MAX_ERR            # (no matrix is the problem)
    # split-bf16 global branch next to a split-fp16 local one (the default pairing after a fallback): the local kernels read the
    # encoder's rows as operands too -- that branch is reported; with both in split-bf16 nothing is checked
    pk = packing.PackedParams(sd, cfg, "cpu", "bf16x3", radius_poly="off")
    assert pk.precision_local == "f16x3" and pk.split_fp16_report["local"]["clipped"] and not pk.split_fp16_report["global"]["clipped"]
    assert not hasattr(packing.PackedParams(sd, cfg, "cpu", "bf16x3", radius_poly="off", precision_local="bf16x3"), "encoder_activation_max")
