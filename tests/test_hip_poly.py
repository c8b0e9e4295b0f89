"""GPU (MI355X): the filter-polynomial kernels (csrc/nodeconv.hip k_cfconv_node: one wave per quad of targets, radius rows +
local quad tiles; k_pair_head_poly; k_edge_attr_poly) and every fallback around them, against the reference fixtures and
against the one-list kernels that evaluate the filter MLPs for every edge.  `radius_poly` (agdiff_amd/packing.py):
  auto    radius edges and every local edge type from d-polynomials (the default the other test files run)
  radius  polynomials for the radius edges only, local edges through the filter MLPs on the padded local list
  kt2     64-term expansions (two k-tiles) for radius edges and local types alike
  kt3/kt4 96 / 128 terms: the rungs a sharp checkpoint takes before the filter MLPs
  off     one list, every edge through the encoder + filter MLPs (rounds 1-2a product path)
plus `auto-l2`: as auto with every local type's coefficient set read from L2 instead of LDS (tune_poly_lds_sets = 1: the
path types beyond the LDS-resident sets take), and `auto-full` / `kt2-full`: three MFMA passes for every polynomial term
(model.poly_passes = "full"; the other split-mode runs take one pass for the high terms: agdiff_params_t.poly_plan 1; `-from64` / `-from96`:
plans 2 / 3, one pass from term 64 / 96 on)."""
import ctypes

import numpy as np
import pytest
import torch

from helpers import FORWARD_CASES, check_close, load_golden, sampler_case_cfg, sampler_case_kwargs, t

pytestmark = pytest.mark.gpu
MODES = ["auto", "radius", "kt2", "kt3", "kt4", "off"]


def _model(cfg, mode, head_scale=1e-3, precision="f16x3"):
    from agdiff_amd import get_model
    from oracle import agdiff_oracle as O
    sd = O.synth_state_dict_for(cfg, head_scale=head_scale)
    m = get_model(cfg)
    m.precision, m.radius_poly = precision, mode
    m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    return m.to("cuda:0").eval()


def _one_pass(ptr, agg, first, chunk_edges, N):
    """agg[node] plus the partial sums the node's later chunks kept in `first` (csrc/node.hip ag_agg_slice)."""
    ip = ptr.cpu().numpy().astype(np.int64)
    a = agg.view(-1, 192).cpu().double().numpy().copy()
    f = first.view(-1, 192).cpu().double().numpy()
    for i in range(N):
        lo, hi = ip[i], ip[i + 1]
        if hi <= lo:
            a[i] = 0.0
            continue
        for c in range(lo // chunk_edges + 1, (hi - 1) // chunk_edges + 1):
            a[i] += f[c]
    return a


def _aggregates(ws, topo, lib):
    """What the node stage reads after agdiff_cfconv_fused over the full list."""
    from agdiff_amd import _lib
    ct = lambda n: _lib.TILE * lib.agdiff_conv_chunk_tiles(ctypes.c_int64(n))
    return _one_pass(ws.in_ptr, ws.agg, ws.agg_first, ct(topo.max_edges), topo.N)


def _variants(ws):
    return int(ws.variant_log.item())


def _expect(pk, mode):
    assert pk.poly_kt == {"auto": 1, "radius": 1, "kt2": 2, "kt3": 3, "kt4": 4, "off": 0}[mode]
    want_slots = mode == "auto" or mode.startswith("kt")
    assert (pk.struct.poly_num_slots > 0) == want_slots, (mode, pk.struct.poly_num_slots, pk.poly_errors)


# (the pass-plan variants of the 96 / 128-term rungs on the Drugs-shaped fixture only: the suite's time)
_FORWARD_MODES = [(c, m) for c in ("g3_forward_qm9_small", "g3_forward_drugs_capped") for m in MODES + ["auto-full", "kt2-full"]] + \
    [("g3_forward_drugs_capped", m) for m in ("kt3-full", "kt4-full", "kt3-from64", "kt4-from64", "kt4-from96")]


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x3"])
@pytest.mark.parametrize("case,mode", _FORWARD_MODES)
def test_forward_every_filter_mode(case, mode, precision):
    g = load_golden(case)
    full, later = mode.endswith("-full"), {"from64": 2, "from96": 3}.get(mode.split("-")[-1], 0)
    passes = mode.split("-")[-1] if (full or later) else "auto"
    mode = mode.split("-")[0]
    m = _model(FORWARD_CASES[case](), mode, precision=precision)
    m.poly_passes = passes
    out = m(t(g["atom_type"]).cuda(), t(g["pos"]).cuda(), t(g["bond_index"]).cuda(), t(g["bond_type"]).cuda(),
            t(g["batch"]).cuda(), None, return_edges=True, extend_order=False)
    _expect(m.packed(), mode)
    # (plans 2, 3 -- one pass from term 64 / 96 on -- are what a sharp network takes at three / four k-tiles; forced here on the smooth one)
    assert m.packed().struct.poly_plan == ((later or 1) if (mode != "off" and precision != "f32" and not full) else 0), m.packed().poly_high_bound
    ws = m._batch_cache[2]
    assert np.array_equal(out[2].cpu().numpy(), g["edge_index"]) and np.array_equal(out[3].cpu().numpy(), g["edge_type"])
    if "schnet_out" in g:
        check_close("poly[%s] schnet_out[%s]" % (mode, case), ws.h.view(-1, 128).cpu().numpy(), g["schnet_out"], precision)
    check_close("poly[%s] inv_g[%s]" % (mode, case), out[0].cpu().numpy(), g["edge_inv_global"], precision)
    check_close("poly[%s] inv_l[%s]" % (mode, case), out[1].cpu().numpy(), g["edge_inv_local"], precision)


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x3"])
@pytest.mark.parametrize("mode", MODES + ["auto-l2", "auto-mixed"])
@pytest.mark.parametrize("case", ["g5_sampler_lowT_global", "g5_sampler_mixed_cliplocal"])
def test_sampler_every_filter_mode(case, mode, precision):
    """The denoising loop (polynomial global head on the radius edges in every mode but `off`)."""
    from agdiff_amd import _lib
    l2, mixed = mode == "auto-l2", mode == "auto-mixed"
    mode = "auto" if (l2 or mixed) else mode
    g = load_golden(case)
    m = _model(sampler_case_cfg(g, case), mode, head_scale=float(g["head_scale"]), precision=precision)
    if l2:
        m.tuning["poly_lds_sets"] = 1
    if mixed:      # the 2-hop edges (type 23) as if their fit had been refused: THEY go through the filter MLPs, the other local
        m.poly_refuse_types = (23,)      # types and the radius edges keep their polynomials (a "mixed" batch)
    pos, traj = m.langevin_dynamics_sample_diffusion(
        t(g["atom_type"]).cuda(), t(g["pos_init"]).cuda(), t(g["bond_index"]).cuda(), t(g["bond_type"]).cuda(),
        t(g["batch"]).cuda(), int(g["num_graphs"]), extend_order=False, n_steps=int(g["n_steps"]),
        noise=t(g["noise"]).cuda(), **sampler_case_kwargs(g))
    _expect(m.packed(), mode)
    V, var = _lib.DEFINES, _variants(m._batch_cache[2])
    if "lowT" in case:        # the global branch ran: which CFConv did?
        assert bool(var & V["AGDIFF_VAR_CFCONV_NODE"]) == (mode != "off") and bool(var & V["AGDIFF_VAR_CFCONV_FUSED"]) == (mode == "off")
        assert bool(var & V["AGDIFF_VAR_CFCONV_NODE_LOCAL"]) == (mode == "auto" or mode.startswith("kt"))
        assert bool(var & V["AGDIFF_VAR_CFCONV_LOCAL_MLP"]) == (mode == "radius" or mixed)
        if mixed:
            assert 23 in m.packed().poly_refused_types and 23 not in m.packed().local_slots and m.packed().struct.poly_num_slots > 0
        if not mode.startswith("kt"):     # (two-k-tile sets are 48 KiB: two typed ones fit next to the radius edges', the rest come from L2)
            assert bool(var & V["AGDIFF_VAR_POLY_L2_SETS"]) == l2
    check_close("poly[%s] traj[%s]" % (mode, case), torch.stack(traj).numpy(), g["traj"], precision)
    check_close("poly[%s] pos[%s]" % (mode, case), pos.cpu().numpy(), g["pos_final"], precision)


@pytest.mark.parametrize("gt", [4, 2, 1])
@pytest.mark.parametrize("kind,mols,copies", [("drugs", 5, 4), ("qm9", 7, 5)])
def test_node_cfconv_equals_one_list_kernel(kind, mols, copies, gt):
    """Kernel level, through the C ABI: on the same graph, node inputs and block, agdiff_cfconv_node (polynomial filters;
    radius rows + local quad tiles, one complete row of agg per node) equals what agdiff_cfconv_fused (filter MLPs on
    every edge of the full list) aggregates; the radius rows of a target are the type-0 subsequence of its list, the pad
    rows that complete its last tile contribute exactly nothing; and the same with the local edges through the filter MLPs
    (agdiff_cfconv_local's second aggregate) and with the typed sets read from L2; for four, two and one target per wave."""
    from agdiff_amd import _lib, drugs_model_config, qm9_model_config, synth
    lib = _lib.load()
    RS = _lib.DEFINES["AGDIFF_RAD_STRIDE"]
    cfg = (qm9_model_config if kind == "qm9" else drugs_model_config)(num_diffusion_timesteps=50, beta_end=2e-5)
    for precision in ("f32", "bf16x3", "f16x3"):
        m = _model(cfg, "auto", precision=precision)
        m.group_targets = gt                    # targets per wave: 4 (what large batches take), 2, 1
        b = synth.make_packed_batch(kind, mols, copies, seed=17)
        at, bi, bt, ba = [t(b[k]).cuda() for k in ("atom_type", "bond_index", "bond_type", "batch")]
        pos = (torch.randn(at.shape[0], 3, generator=torch.Generator().manual_seed(2)) * 2.0).cuda()
        m(at, pos, bi, bt, ba, None, return_edges=True, extend_order=False)     # full forward: e_attr, scales of every list
        topo, ws, pk = m._batch_cache[1], m._batch_cache[2], m.packed()
        assert topo.group_targets == gt == topo.struct.group_targets
        assert pk.struct.poly_num_slots > 0 and lib.agdiff_local_poly_enabled(
            ctypes.byref(pk.struct), ctypes.byref(topo.struct), ctypes.byref(ws.struct)) == 1
        P, T, W, st = ctypes.byref(pk.struct), ctypes.byref(topo.struct), ctypes.byref(ws.struct), _lib.stream_ptr()
        E, N = int(ws.num_edges.item()), topo.N
        ety, edst = ws.e_type[:E].cpu().numpy(), ws.e_dst[:E].cpu().numpy()
        esrc, elen = ws.e_src[:E].cpu().numpy(), ws.e_len[:E].cpu().numpy()
        cnt = ws.rad_cnt.cpu().numpy()
        rsrc, rlen = ws.rad_src.view(N, RS).cpu().numpy(), ws.rad_len.view(N, RS).cpu().numpy()
        rsc = ws.r_scale.view(-1, N, RS)[: 2 * cfg.num_convs].cpu().numpy()
        assert np.array_equal(cnt, np.bincount(edst[ety == 0], minlength=N)) and cnt.max() <= _lib.RADIUS_CAP
        ip = ws.in_ptr.cpu().numpy()
        for i in range(N):
            sel = slice(ip[i], ip[i + 1])
            r0 = ety[sel] == 0
            c, cp = cnt[i], (cnt[i] + 15) // 16 * 16
            assert np.array_equal(rsrc[i, :c], esrc[sel][r0]) and np.array_equal(rlen[i, :c], elen[sel][r0])
            assert np.all(rsrc[i, c:cp] == i) and np.all(rlen[i, c:cp] == 0) and np.all(rsc[:, i, c:cp] == 0)
        # the scales the graph build's fill pass wrote are those of the stand-alone launch, bit for bit
        fused = ws.r_scale.clone()
        ws.r_scale.fill_(float("nan"))
        assert lib.agdiff_edge_scales_split(P, T, W, 0, st) == 0
        torch.cuda.synchronize()
        used = torch.zeros(N, RS, dtype=torch.bool)
        for i in range(N):
            used[i, :(cnt[i] + 15) // 16 * 16] = True
        used = used.cuda().view(-1)
        for c in range(2 * cfg.num_convs):
            assert torch.equal(fused.view(-1, N * RS)[c][used], ws.r_scale.view(-1, N * RS)[c][used]), c
        assert lib.agdiff_edge_scales(P, T, W, 1, st) == 0
        sc = 3.0 if precision == "bf16x3" else 1.0
        # ws.xs holds lin1 outputs of the last block after the forward: any block's filters may be applied to them
        for k in (0, cfg.num_convs - 1):
            assert lib.agdiff_cfconv_fused(P, T, W, k, st) == 0
            torch.cuda.synchronize()
            ref = _aggregates(ws, topo, lib)
            ws.agg.fill_(float("nan"))
            assert lib.agdiff_cfconv_node(P, T, W, k, st) == 0
            torch.cuda.synchronize()
            got = ws.agg.view(-1, 192).cpu().double().numpy()
            # (both sides carry their own split-bf16 rounding -- polynomial vs MLP chain --, each ~1.5e-5 from the exact
            # value: the gate is for a difference of two such figures)
            check_close("node_cfconv[%s] block %d" % (kind, k), got, ref, precision, scale=sc)
            again = ws.agg.clone()
            ws.agg.fill_(float("nan"))
            assert lib.agdiff_cfconv_node(P, T, W, k, st) == 0
            torch.cuda.synchronize()
            assert torch.equal(again, ws.agg)                                  # fixed summation order: bitwise reproducible
            # every local type's set from L2 instead of LDS: the same arithmetic in the same order
            pk.set_tuning(poly_lds_sets=1)
            ws.agg.fill_(float("nan"))
            assert lib.agdiff_cfconv_node(P, T, W, k, st) == 0
            torch.cuda.synchronize()
            assert torch.equal(again, ws.agg)
            pk.set_tuning(poly_lds_sets=0)
            # local edges through the filter MLPs: agdiff_cfconv_node (radius rows only) + agdiff_cfconv_local
            pk.set_tuning(local_poly_off=1)
            ctl = (topo.Lc + 15) // 16
            assert lib.agdiff_edge_encoder(P, _lib.ptr(ws.num_local_canon), ctl, _lib.ptr(ws.lc_len), _lib.ptr(topo.lc_type),
                                           _lib.ptr(ws.l_attr_frag), _lib.ptr(ws.l_attr_rows), _lib.ptr(topo.lp_row),
                                           _lib.ptr(topo.lc_ppos), _lib.ptr(topo.lc_pmir), st) == 0
            assert lib.agdiff_edge_scales_split(P, T, W, 1, st) == 0
            ws.agg.fill_(float("nan"))
            assert lib.agdiff_cfconv_node(P, T, W, k, st) == 0 and lib.agdiff_cfconv_local(P, T, W, k, st) == 0
            torch.cuda.synchronize()
            ct = lambda n: _lib.TILE * lib.agdiff_conv_chunk_tiles(ctypes.c_int64(n))
            got2 = ws.agg.view(-1, 192).cpu().double().numpy() + _one_pass(topo.lp_ptr, ws.agg_loc, ws.agg_first_loc, ct(topo.Lp), N)
            check_close("node_cfconv+local_mlp[%s] block %d" % (kind, k), got2, ref, precision, scale=sc)
            pk.set_tuning(local_poly_off=0)


@pytest.mark.parametrize("passes", ["auto", "full"])
@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x3"])
@pytest.mark.parametrize("kind,mols,copies,mode", [("drugs", 3, 6, "auto"), ("qm9", 5, 9, "auto"), ("drugs", 2, 5, "kt2"),
                                                   ("drugs", 2, 5, "kt3")])
def test_cfconv_node_shapes_agree_bitwise(kind, mols, copies, mode, precision, passes):
    """agdiff_cfconv_node has two shapes at one k-tile (csrc/nodeconv.hip NodeConvShape): 12-wave workgroups with groups of three
    channel tiles, and -- from tune_cfconv_four_min_quads quads on -- 16-wave workgroups at 128 VGPRs with groups of two (two
    k-tiles keep the first shape).  A target's sums are taken by ONE wave in the same tile and row
    order in both, so the aggregates, and everything downstream, must be identical bit for bit; variant_log says which shape
    ran.  The same holds inside each of the two row layouts on quads (radius rows in quad tiles: k_cfconv_quad; every target its
    own radius tiles: k_cfconv_node); BETWEEN the layouts the order of a target's additions differs, so they agree to rounding."""
    from agdiff_amd import _lib, drugs_model_config, qm9_model_config, synth
    if mode == "kt3" and passes == "full":
        pytest.skip("three k-tiles: one shape; its three-pass build is covered by test_forward_every_filter_mode")
    cfg = (drugs_model_config if kind == "drugs" else qm9_model_config)(num_diffusion_timesteps=20)
    b = synth.make_packed_batch(kind, mols, copies, seed=23)
    at, bi, bt, ba = [t(b[k]).cuda() for k in ("atom_type", "bond_index", "bond_type", "batch")]
    pos = (torch.randn(at.shape[0], 3, generator=torch.Generator().manual_seed(3)) * 1.5).cuda()
    by_layout = {}
    for group, quad in ((None, 0), (4, 0), (4, -1)):
        outs = {}
        for four in (1, -1):
            m = _model(cfg, mode, precision=precision)
            m.poly_passes = passes
            m.group_targets = group
            m.tuning["cfconv_four_min_quads"] = four
            m.tuning["cfconv_quad_tiles"] = quad
            out = m(at, pos, bi, bt, ba, None, return_edges=True, extend_order=False)
            ws, var = m._batch_cache[2], _variants(m._batch_cache[2])
            has_four = mode == "auto"                    # (two k-tiles: one shape only; the tuning field must not matter)
            assert bool(var & _lib.DEFINES["AGDIFF_VAR_CFCONV_NODE_FOUR"]) == (four == 1 and has_four), (four, var)
            assert var & _lib.DEFINES["AGDIFF_VAR_CFCONV_NODE"]
            assert bool(var & _lib.DEFINES["AGDIFF_VAR_CFCONV_NODE_QUAD"]) == (group == 4 and quad == 0), (group, quad, var)
            outs[four] = (out[0].clone(), out[1].clone(), ws.h.clone(), ws.agg.clone())
        for a_, b_ in zip(outs[1], outs[-1]):
            assert torch.equal(a_, b_)
        by_layout[(group, quad)] = outs[1]
    for name, a_, b_ in zip(("edge_inv_global", "edge_inv_local", "h", "agg"), by_layout[(4, 0)], by_layout[(4, -1)]):
        assert torch.isfinite(a_).all()
        err = float((a_ - b_).abs().max() / b_.abs().max().clamp_min(1e-30))
        # (two chains of the same arithmetic whose aggregates differ in the last fp32 bits: six blocks and a head later the
        # operand roundings of the mode have amplified that -- the figure tests/helpers.py allows two split chains)
        assert err <= {"f32": 2e-6, "f16x3": 6e-6, "bf16x3": 6e-5}[precision], (name, err)


_ORACLE_RESULTS = {}        # oracle outputs shared by the parametrisations of one test (same inputs, same weights)


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x3"])
def test_sharper_first_layer_takes_64_terms_at_bench_scale(precision):
    """a feature_expansion layer 8 x sharper than the synthetic checkpoint's is not a 32-term polynomial
    at 1e-6 but a 64-term one: in mode `auto` the radius edges AND every local type run on two-k-tile sets (48 KiB each:
    two typed sets LDS-resident, the rest from L2), no MLP filter anywhere, on a batch past every small-batch threshold
    (>= 12,288 atoms); three denoising steps against the oracle."""
    from agdiff_amd import _lib, drugs_model_config, get_model, synth
    from oracle import agdiff_oracle as O
    cfg = drugs_model_config(num_diffusion_timesteps=30, beta_end=2e-5)
    sd = O.synth_state_dict_for(cfg)
    for k in ("edge_encoder_global.feature_expansion.weight", "model_global.0.feature_expansion.weight"):
        sd[k] = sd[k] * 8.0             # (Drugs config: 5 x still passes at 32 terms with 8e-7; 8 x gives 7e-6 / 4e-10)
    m = get_model(cfg)
    m.precision = precision
    m.load_state_dict({k: v.clone() for k, v in sd.items()})
    m = m.to("cuda:0").eval()
    b = synth.make_packed_batch("drugs", 6, 56, seed=41)
    at, bi, bt, ba = [t(b[k]) for k in ("atom_type", "bond_index", "bond_type", "batch")]
    assert at.shape[0] >= 12288
    g = torch.Generator().manual_seed(13)
    pos_init, noise = torch.randn(at.shape[0], 3, generator=g), torch.randn(3, at.shape[0], 3, generator=g)
    kw = dict(extend_order=False, n_steps=3, w_global=1.0, global_start_sigma=0.5, clip=1000.0)
    if "sharper_kt2" not in _ORACLE_RESULTS:          # (the oracle's three steps on 14.8 k atoms take ~35 s of CPU: once for the three modes)
        nthr = torch.get_num_threads()
        torch.set_num_threads(max(nthr, 16))
        try:
            _ORACLE_RESULTS["sharper_kt2"] = O.langevin_dynamics_sample_diffusion(sd, cfg, at, pos_init, bi, bt, ba, b["num_graphs"],
                                                                                  noise=noise, **kw)[0]
        finally:
            torch.set_num_threads(nthr)
    ref = _ORACLE_RESULTS["sharper_kt2"]
    got, _ = m.langevin_dynamics_sample_diffusion(at.cuda(), pos_init.cuda(), bi.cuda(), bt.cuda(), ba.cuda(),
                                                  b["num_graphs"], noise=noise.cuda(), **kw)
    pk, var, V = m.packed(), _variants(m._batch_cache[2]), _lib.DEFINES
    assert pk.poly_kt == 2 and pk.poly_errors[1] > 1e-6 >= pk.poly_errors[2], pk.poly_errors
    assert pk.struct.poly_num_slots == len(m._batch_cache[1].local_types) and not pk.poly_refused_types
    assert var & V["AGDIFF_VAR_CFCONV_NODE_LOCAL"] and var & V["AGDIFF_VAR_HEAD_POLY"] and var & V["AGDIFF_VAR_POLY_L2_SETS"]
    assert not var & (V["AGDIFF_VAR_CFCONV_LOCAL_MLP"] | V["AGDIFF_VAR_CFCONV_FUSED"])
    check_close("sharper_first_layer_kt2 sampler", got.cpu().numpy(), ref.numpy(), precision)


def _sharpen(sd, scale, bounded):
    """The encoder's first layer `scale` times sharper.  Plain: its weight times `scale` (the network's values grow with it).
    Bounded: weight and bias times `scale`, the next layer's columns that read its outputs divided by it -- gelu(s u) / s: kinks
    `scale` times sharper at unchanged magnitudes, the shape a trained network's sharp features would have."""
    for e in ("edge_encoder_global", "model_global.0"):
        sd[e + ".feature_expansion.weight"] = sd[e + ".feature_expansion.weight"] * scale
        if bounded:
            n = sd[e + ".feature_expansion.weight"].shape[0]
            sd[e + ".feature_expansion.bias"] = sd[e + ".feature_expansion.bias"] * scale
            w = sd[e + ".edge_feature_mlp.0.weight"].clone()
            w[:, :n] = w[:, :n] / scale
            sd[e + ".edge_feature_mlp.0.weight"] = w
    return sd


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x3"])
@pytest.mark.parametrize("kind,scale,terms", [("qm9", 100.0, 128), ("drugs-bounded", 32.0, 96), ("drugs-bounded", 64.0, 128)])
def test_sharp_first_layer_takes_the_96_and_128_term_rungs(kind, scale, terms, precision):
    """between the 64-term sets (first layer up to ~24 x the synthetic checkpoint's) and the filter MLPs there are
    two more rungs -- 96 and 128 terms (k-tiles 2, 3: T_64 .. T_120 by the recurrence in steps of eight; 8-wave workgroups at 256
    registers; 72- / 96-KiB sets, so one typed set or none stays in LDS and the others come from L2).  A first layer 32 .. 100 x
    sharper is accepted there in mode `auto`, for the radius edges and every local type, and four denoising steps match the
    oracle -- with the plain scaling (whose values grow with the scale) and with the bounded one (_sharpen)."""
    from agdiff_amd import _lib, drugs_model_config, get_model, qm9_model_config, synth
    from oracle import agdiff_oracle as O
    bounded, kind = kind.endswith("-bounded"), kind.split("-")[0]
    cfg = (qm9_model_config if kind == "qm9" else drugs_model_config)(num_diffusion_timesteps=30, beta_end=2e-5)
    sd = _sharpen(O.synth_state_dict_for(cfg), scale, bounded)
    m = get_model(cfg)
    m.precision = precision
    m.load_state_dict({k: v.clone() for k, v in sd.items()})
    m = m.to("cuda:0").eval()
    b = synth.make_packed_batch(kind, 2, 4, seed=17)
    at, bi, bt, ba = [t(b[k]) for k in ("atom_type", "bond_index", "bond_type", "batch")]
    g = torch.Generator().manual_seed(5)
    pos_init, noise = torch.randn(at.shape[0], 3, generator=g), torch.randn(4, at.shape[0], 3, generator=g)
    kw = dict(extend_order=False, n_steps=4, w_global=1.0, global_start_sigma=0.5, clip=1000.0)
    key = ("rungs", kind, scale, bounded)
    if key not in _ORACLE_RESULTS:
        _ORACLE_RESULTS[key] = O.langevin_dynamics_sample_diffusion(sd, cfg, at, pos_init, bi, bt, ba, b["num_graphs"], noise=noise, **kw)[0]
    ref = _ORACLE_RESULTS[key]
    got, _ = m.langevin_dynamics_sample_diffusion(at.cuda(), pos_init.cuda(), bi.cuda(), bt.cuda(), ba.cuda(),
                                                  b["num_graphs"], noise=noise.cuda(), **kw)
    pk, var, V = m.packed(), _variants(m._batch_cache[2]), _lib.DEFINES
    kt = terms // 32
    assert pk.poly_kt == kt and pk.poly_errors[kt - 1] > 1e-6 >= pk.poly_errors[kt], pk.poly_errors
    assert pk.struct.poly_num_slots == len(m._batch_cache[1].local_types) and not pk.poly_refused_types
    assert var & V["AGDIFF_VAR_CFCONV_NODE_LOCAL"] and var & V["AGDIFF_VAR_HEAD_POLY"] and var & V["AGDIFF_VAR_POLY_L2_SETS"]
    assert not var & (V["AGDIFF_VAR_CFCONV_LOCAL_MLP"] | V["AGDIFF_VAR_CFCONV_FUSED"])
    check_close("sharp_first_layer_%d_terms[%s] sampler" % (terms, kind), got.cpu().numpy(), ref.numpy(), precision)


def test_rejected_fit_falls_back_to_the_mlps():
    """A first layer too sharp for 128 terms at 1e-6: the polynomials are refused at load time and every edge goes through
    the encoder + filter MLPs; results still match the oracle."""
    from agdiff_amd import get_model, qm9_model_config, synth
    from oracle import agdiff_oracle as O
    cfg = qm9_model_config(num_diffusion_timesteps=30, beta_end=2e-5)
    sd = O.synth_state_dict_for(cfg)
    for k in ("edge_encoder_global.feature_expansion.weight", "model_global.0.feature_expansion.weight"):
        sd[k] = sd[k] * 400.0
    m = get_model(cfg)
    m.load_state_dict({k: v.clone() for k, v in sd.items()})
    m = m.to("cuda:0").eval()
    b = synth.make_packed_batch("qm9", 2, 2, seed=3)
    at, bi, bt, ba = [t(b[k]) for k in ("atom_type", "bond_index", "bond_type", "batch")]
    g = torch.Generator().manual_seed(5)
    pos_init, noise = torch.randn(at.shape[0], 3, generator=g), torch.randn(4, at.shape[0], 3, generator=g)
    kw = dict(extend_order=False, n_steps=4, w_global=1.0, global_start_sigma=0.5, clip=1000.0)
    ref, _ = O.langevin_dynamics_sample_diffusion(sd, cfg, at, pos_init, bi, bt, ba, b["num_graphs"], noise=noise, **kw)
    got, _ = m.langevin_dynamics_sample_diffusion(at.cuda(), pos_init.cuda(), bi.cuda(), bt.cuda(), ba.cuda(),
                                                  b["num_graphs"], noise=noise.cuda(), **kw)
    pk = m.packed()
    assert pk.poly_kt == 0 and pk.struct.poly_num_slots == 0 and min(pk.poly_errors.values()) > 1e-6, pk.poly_errors
    check_close("rejected_fit sampler", got.cpu().numpy(), ref.numpy(), "f16x3")


def _chain_with_bond_types(n_types):
    n = n_types + 1                                         # a chain with a different bond type on every link
    src = np.arange(n - 1); dst = src + 1
    bi = torch.from_numpy(np.concatenate([np.stack([src, dst]), np.stack([dst, src])], axis=1)).long()
    bt = torch.from_numpy(np.concatenate([np.arange(1, n), np.arange(1, n)])).long()        # types 1..n_types
    return torch.full((n,), 6, dtype=torch.long), bi, bt, torch.zeros(n, dtype=torch.long)


@pytest.mark.parametrize("n_types,slots_kept", [(8, True), (17, False)])
def test_many_local_edge_types(n_types, slots_kept):
    """Eight distinct bond types in one batch: more than the five typed coefficient sets that fit in LDS next to the radius
    edges' one -- the later types' sets are read from L2 by the tiles that meet them; seventeen: one more than
    AGDIFF_POLY_MAX_SLOTS -- the seventeenth type's edges go through the filter MLPs (a "mixed" batch), the sixteen
    slotted types and the radius edges keep their polynomials.  Both against the oracle."""
    from agdiff_amd import _lib, get_model, qm9_model_config
    from oracle import agdiff_oracle as O
    cfg = qm9_model_config(num_diffusion_timesteps=30, beta_end=2e-5)
    sd = O.synth_state_dict_for(cfg)
    m = get_model(cfg)
    m.load_state_dict({k: v.clone() for k, v in sd.items()})
    m = m.to("cuda:0").eval()
    at, bi, bt, ba = _chain_with_bond_types(n_types)
    n = at.shape[0]
    g = torch.Generator().manual_seed(9)
    pos_init, noise = torch.randn(n, 3, generator=g), torch.randn(3, n, 3, generator=g)
    kw = dict(extend_order=False, n_steps=3, w_global=1.0, global_start_sigma=0.5, clip=1000.0)
    ref, _ = O.langevin_dynamics_sample_diffusion(sd, cfg, at, pos_init, bi, bt, ba, 1, noise=noise, **kw)
    got, _ = m.langevin_dynamics_sample_diffusion(at.cuda(), pos_init.cuda(), bi.cuda(), bt.cuda(), ba.cuda(), 1,
                                                  noise=noise.cuda(), **kw)
    pk, var, V = m.packed(), _variants(m._batch_cache[2]), _lib.DEFINES
    assert pk.poly_kt == 1 and pk.struct.poly_num_slots == min(n_types, V["AGDIFF_POLY_MAX_SLOTS"])
    assert pk.poly_refused_types == (set() if slots_kept else {17})
    assert var & V["AGDIFF_VAR_POLY_L2_SETS"] and var & V["AGDIFF_VAR_CFCONV_NODE_LOCAL"]
    assert bool(var & V["AGDIFF_VAR_CFCONV_LOCAL_MLP"]) == (not slots_kept)      # the seventeenth type alone takes the MLPs
    check_close("many_local_types[%d] sampler" % n_types, got.cpu().numpy(), ref.numpy(), "f16x3")


@pytest.mark.parametrize("kind,mols,copies", [("drugs", 4, 3), ("qm9", 6, 4)])
def test_radius_only_canonical_list_of_the_sampler(kind, mols, copies):
    """agdiff_graph_build_ex(canon_radius_only=1), as the denoising loop calls it: the canonical list holds one entry per
    mirror pair of RADIUS edges (plus every unpaired radius edge); an entry and its mirror cover every radius edge of the full
    list exactly once."""
    from agdiff_amd import drugs_model_config, qm9_model_config, synth
    cfg = (qm9_model_config if kind == "qm9" else drugs_model_config)(num_diffusion_timesteps=20, beta_end=2e-5)
    m = _model(cfg, "auto")
    m.fused_front = False                # (the unfused loop: agdiff_langevin_update + agdiff_graph_build_scaled per step)
    b = synth.make_packed_batch(kind, mols, copies, seed=23)
    at, bi, bt, ba = [t(b[k]).cuda() for k in ("atom_type", "bond_index", "bond_type", "batch")]
    pos_init = torch.randn(at.shape[0], 3, generator=torch.Generator().manual_seed(6)).cuda()
    m.langevin_dynamics_sample_diffusion(at, pos_init, bi, bt, ba, b["num_graphs"], extend_order=False, n_steps=2,
                                         w_global=1.0, global_start_sigma=0.5, clip=1000.0)
    ws = m._batch_cache[2]
    E, C = int(ws.num_edges.item()), int(ws.num_canon.item())
    R = int(ws.rad_cnt.sum().item())
    g = lambda x, n: x[:n].cpu().numpy()
    ety, esrc, edst = g(ws.e_type, E), g(ws.e_src, E), g(ws.e_dst, E)
    cs, cd, cp, cm = [g(x, C) for x in (ws.c_src, ws.c_dst, ws.c_pos, ws.c_mir)]
    elen = g(ws.e_len, E)
    assert np.all(g(ws.c_type, C) == 0) and np.all(ety[cp] == 0)
    assert np.array_equal(esrc[cp], cs) and np.array_equal(edst[cp], cd)
    assert np.array_equal(elen[cp], g(ws.c_len, C))
    mk = cm >= 0
    assert np.array_equal(esrc[cm[mk]], cd[mk]) and np.array_equal(edst[cm[mk]], cs[mk]) and np.all(ety[cm[mk]] == 0)
    assert np.array_equal(elen[cm[mk]], elen[cp[mk]])                    # a mirror pair has ONE length, bit for bit
    assert np.all(cs[mk] < cd[mk])                                        # the canonical one of a pair is src < dst
    cover = np.bincount(np.concatenate([cp, cm[mk]]), minlength=E)
    assert R == int((ety == 0).sum()) and np.array_equal(cover, (ety == 0).astype(cover.dtype))


@pytest.mark.parametrize("far", [False, True])
@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x3"])
def test_local_edge_rows_polynomial_and_flagged_tiles(precision, far):
    """agdiff_local_edge_rows against agdiff_edge_encoder (the MLP) on the same canonical local list, through the C ABI:
    compact molecules (every length inside the cutoff: all rows by polynomial, none flagged), stretched ones (bonded atoms
    far apart, as at high sigma) and a mix.  far = False: every row longer than the cutoff is flagged and goes through the MLP;
    far = True (the default build): rows in (cutoff, 10 cutoff] of a type with a far set come from that polynomial too
    (packing.fit_attr_far: edge_attr beyond the cutoff is a 32-term polynomial to ~1e-9), only rows beyond it -- the
    "exploded" case -- or of a type without a far set (the kernel holds 9 sets: here type 12 has none) keep the MLP."""
    from agdiff_amd import _lib, drugs_model_config, synth
    lib = _lib.load()
    cfg = drugs_model_config(num_diffusion_timesteps=20, beta_end=2e-5)
    m = _model(cfg, "auto", precision=precision)
    m.attr_far_rows = far
    b = synth.make_packed_batch("drugs", 5, 3, seed=31)
    at, bi, bt, ba = [t(b[k]).cuda() for k in ("atom_type", "bond_index", "bond_type", "batch")]
    gen = torch.Generator().manual_seed(8)
    base = torch.randn(at.shape[0], 3, generator=gen)
    G = b["num_graphs"]
    stretch = torch.ones(G)
    stretch[G // 2:] = 9.0                       # the second half of the molecules blown up: lengths far beyond 10 A
    cases = [("compact", base * 1.5), ("stretched", base * 12.0), ("mixed", base * 1.5 * stretch[t(b["batch"])].unsqueeze(1)),
             ("exploded", base * 60.0)]
    for name, pos in cases:
        m(at, pos.cuda(), bi, bt, ba, None, extend_order=False)
        topo, ws, pk = m._batch_cache[1], m._batch_cache[2], m.packed()
        assert pk.struct.poly_num_slots > 0 and (pk.struct.attr_poly_far_slots > 0) == far
        P, T, W, st = ctypes.byref(pk.struct), ctypes.byref(topo.struct), ctypes.byref(ws.struct), _lib.stream_ptr()
        ct = (topo.Lc + _lib.TILE - 1) // _lib.TILE
        assert lib.agdiff_edge_encoder(P, _lib.ptr(ws.num_local_canon), ct, _lib.ptr(ws.lc_len), _lib.ptr(topo.lc_type), None,
                                       _lib.ptr(ws.l_attr_rows), None, None, None, st) == 0
        torch.cuda.synchronize()
        ref = ws.l_attr_rows.view(-1, 128)[:topo.Lc].clone()
        ws.l_attr_rows.zero_()
        assert lib.agdiff_local_edge_rows(P, T, W, st) == 0
        torch.cuda.synchronize()
        got = ws.l_attr_rows.view(-1, 128)[:topo.Lc]
        flags = ws.enc_flags.cpu().numpy()
        lens = ws.lc_len[:topo.Lc].cpu().numpy()
        types = topo.lc_type[:topo.Lc].cpu().numpy()
        has_far = (pk.attr_far_table.cpu().numpy()[types] >= 0) if far else np.zeros(topo.Lc, dtype=bool)
        hard = (lens > cfg.cutoff) & ~(has_far & (lens <= float(pk.struct.attr_poly_far_hi)))
        want = np.array([hard[16 * k:16 * k + 16].any() for k in range(ct)])
        masks = np.array([sum(1 << r for r in range(16) if 16 * k + r < topo.Lc and hard[16 * k + r]) for k in range(ct)])
        assert np.array_equal(flags[1:1 + ct], masks) and flags[0] == want.sum(), name
        long_rows = int((lens > cfg.cutoff).sum())
        if far:
            assert {"compact": hard.sum() == 0, "stretched": 0 < hard.sum() < long_rows // 4, "mixed": 0 < hard.sum() < long_rows,
                    "exploded": hard.sum() > topo.Lc // 2}[name], (name, int(hard.sum()), long_rows)
        else:
            assert {"compact": want.sum() == 0, "stretched": want.sum() > ct // 2, "mixed": 0 < want.sum() < ct,
                    "exploded": want.sum() > ct // 2}[name]
        check_close("local_edge_rows[%s, far %d]" % (name, far), got.cpu().numpy(), ref.cpu().numpy(), precision,
                    scale=3.0 if precision == "bf16x3" else 1.0)       # (two split-bf16 evaluations against each other)
        # rows the polynomials do not cover are the MLP's own output, bit for bit -- and ONLY they: what a row holds depends
        # on its own edge, not on the tile it shares (a neighbour molecule that stretches must not change this one's bits)
        rows = torch.from_numpy(hard)
        assert torch.equal(got[rows], ref[rows])
        assert bool(rows.all()) or not torch.equal(got[~rows], ref[~rows])


@pytest.mark.parametrize("kind,mols,copies", [("drugs", 4, 3), ("qm9", 6, 4)])
@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x3"])
def test_fused_front_equals_the_unfused_loop(kind, mols, copies, precision):
    """agdiff_sampler_front (update of step t + radius graph of step t + 1 in one launch, scores by radius row) against the
    unfused loop (agdiff_langevin_update + agdiff_graph_build_scaled, scores by position in the full list): the graph phase
    alone writes the radius rows, scales and pad rows of agdiff_graph_build_scaled bit for bit and the same canonical
    entries (molecule by molecule, in the order the workgroups claim their ranges); six denoising steps agree to fp32 rounding (the update sums a node's terms over another
    lane partition)."""
    from agdiff_amd import _lib, drugs_model_config, qm9_model_config, synth
    lib = _lib.load()
    RS = _lib.DEFINES["AGDIFF_RAD_STRIDE"]
    cfg = (qm9_model_config if kind == "qm9" else drugs_model_config)(num_diffusion_timesteps=20, beta_end=2e-5)
    m = _model(cfg, "auto", precision=precision)
    b = synth.make_packed_batch(kind, mols, copies, seed=29)
    at, bi, bt, ba = [t(b[k]).cuda() for k in ("atom_type", "bond_index", "bond_type", "batch")]
    gen = torch.Generator().manual_seed(7)
    pos_init = torch.randn(at.shape[0], 3, generator=gen)
    noise = torch.randn(6, at.shape[0], 3, generator=gen)
    kw = dict(extend_order=False, n_steps=6, w_global=1.0, global_start_sigma=0.5, clip=1000.0, noise=noise.cuda())
    m.fused_front = True
    pf, tf = m.langevin_dynamics_sample_diffusion(at, pos_init.cuda(), bi, bt, ba, b["num_graphs"], **kw)
    assert _variants(m._batch_cache[2]) & _lib.DEFINES["AGDIFF_VAR_FUSED_FRONT"]
    m.fused_front = False
    pu, tu = m.langevin_dynamics_sample_diffusion(at, pos_init.cuda(), bi, bt, ba, b["num_graphs"], **kw)
    assert not _variants(m._batch_cache[2]) & _lib.DEFINES["AGDIFF_VAR_FUSED_FRONT"]
    assert float((torch.stack(tf) - torch.stack(tu)).abs().max()) < 2e-6 * float(torch.stack(tu).abs().max())
    # structure: the graph phase alone, on the unfused run's final positions, next to agdiff_graph_build_scaled
    topo, ws, pk = m._batch_cache[1], m._batch_cache[2], m.packed()
    P, T, W, st = ctypes.byref(pk.struct), ctypes.byref(topo.struct), ctypes.byref(ws.struct), _lib.stream_ptr()
    pos = pu.contiguous()
    assert lib.agdiff_graph_build_scaled(P, T, W, _lib.ptr(pos), ctypes.c_float(cfg.cutoff), 1, st) == 0
    torch.cuda.synchronize()
    N = topo.N
    ref = {k: getattr(ws, k).clone() for k in ("rad_cnt", "rad_src", "rad_len", "r_scale")}
    C = int(ws.num_canon.item())
    full_pos = ws.c_pos[:C].long()
    ref_c = set(zip(ws.c_src[:C].tolist(), ws.c_dst[:C].tolist(), ws.c_len[:C].tolist(), (ws.c_mir[:C] >= 0).tolist()))
    for k in ("rad_cnt", "rad_src", "rad_len", "r_scale"):
        getattr(ws, k).fill_(0)
    sa = _lib.StepArgs()
    sa.pos_in = _lib.ptr(pos)
    ws.canon_counter.zero_()
    assert lib.agdiff_sampler_front(P, T, W, ctypes.byref(sa), 2, ctypes.c_float(cfg.cutoff), st) == 0        # (parity 0)
    torch.cuda.synchronize()
    cnt = ws.rad_cnt.cpu().numpy()
    assert torch.equal(ws.rad_cnt, ref["rad_cnt"])
    used = torch.zeros(N, RS, dtype=torch.bool)
    for i in range(N):
        used[i, :(cnt[i] + 15) // 16 * 16] = True
    u = used.view(-1).cuda()
    assert torch.equal(ws.rad_src[u], ref["rad_src"][u]) and torch.equal(ws.rad_len[u], ref["rad_len"][u])
    for c in range(2 * cfg.num_convs):
        assert torch.equal(ws.r_scale.view(-1, N * RS)[c][u], ref["r_scale"].view(-1, N * RS)[c][u])
    live = np.arange(int(ws.canon_counter[0].item()))
    got_c = set(zip(ws.c_src.cpu().numpy()[live].tolist(), ws.c_dst.cpu().numpy()[live].tolist(),
                    ws.c_len.cpu().numpy()[live].tolist(), (ws.c_mir.cpu().numpy()[live] >= 0).tolist()))
    assert live.size == C and got_c == ref_c
    # the local in-adjacency masks: copied from topo->loc_bits (host-built) or, without them, built in the kernel per step
    assert topo.struct.loc_bits
    with_bits = {k: getattr(ws, k).clone() for k in ("rad_cnt", "rad_src", "rad_len", "r_scale")}
    keep, topo.struct.loc_bits = topo.struct.loc_bits, None
    try:
        for k in with_bits:
            getattr(ws, k).fill_(0)
        ws.canon_counter.zero_()
        assert lib.agdiff_sampler_front(P, T, W, ctypes.byref(sa), 2, ctypes.c_float(cfg.cutoff), st) == 0
        torch.cuda.synchronize()
        for k in with_bits:
            assert torch.equal(getattr(ws, k), with_bits[k]), k
    finally:
        topo.struct.loc_bits = keep
