"""GPU (MI355X): kernel-level parity of the SchNet block kernels through their own C-ABI entry points, against the
per-module reference fixtures (G2: CFConv F=128 / F=64, InteractionBlock + AdaptiveScaling of block 0,
tests/golden/make_golden.py:g_forward(stages=True)) -- so that rows a8-a10 of SURVEY.md §8 do not rest on the
six-block end-to-end `schnet_out` comparison alone.

  agdiff_schnet_node_stage(k=0)  -> xs  = LeakyReLU(BN(lin1(h0)))  of conv1 | conv2          (schnet.py:153-155)
  agdiff_cfconv_fused(k=0)       -> agg = sum_e x[src] * W_e       of conv1 | conv2          (schnet.py:138-162)
       checked as BN(lin2(agg)) against the reference's CFConv.forward outputs (the fixtures hold those)
  agdiff_schnet_node_stage(k=1)  -> h   = h0 + AdaptiveScaling(InteractionBlock(h0))         (schnet.py:201-234, 280)
       fed with a reference aggregate, checked against h0 + the reference's scaling_modules[0](interactions[0](h0))
  agdiff_graph_build_scaled + agdiff_cfconv_node(k=0)  -> the same aggregate from the kernel the sampler (and bench.py)
       runs: radius rows by target + local quad tiles, filters from d-polynomials; checked like agdiff_cfconv_fused
  agdiff_langevin_update / agdiff_sampler_front  -> eq_transform, clip_norm, center_pos against the G4 fixtures
       (geometry.py:9-17, dualenc.py:581-589), fed with the reference's own edge scores
"""
import ctypes
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import FORWARD_CASES, STAGE_CASES, check_close, load_golden, rel_err, t

pytestmark = pytest.mark.gpu
PRECISIONS = ["f32", "bf16x3", "f16x3"]


def _setup(case, precision, group_targets=None, tuning=None):
    from agdiff_amd import _lib, get_model
    from oracle import agdiff_oracle as O
    g = load_golden(case)
    cfg = FORWARD_CASES[case]()
    sd = O.synth_state_dict_for(cfg)
    m = get_model(cfg)
    m.precision = precision
    m.group_targets = group_targets
    m.tuning.update(tuning or {})
    m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    m = m.to("cuda:0").eval()
    lib = _lib.load()
    at = t(g["atom_type"]).cuda()
    with torch.no_grad():
        pk = m._renorm_embedding(at)
        topo, ws = m._batch(at, t(g["bond_index"]), t(g["bond_type"]), t(g["batch"]), None, False)
    pos = t(g["pos"]).cuda().contiguous()
    st = _lib.stream_ptr()
    P, Tp, Wp = ctypes.byref(pk.struct), ctypes.byref(topo.struct), ctypes.byref(ws.struct)
    etiles = (topo.max_edges + _lib.TILE - 1) // _lib.TILE
    assert lib.agdiff_graph_build(Tp, Wp, _lib.ptr(pos), ctypes.c_float(cfg.cutoff), st) == 0
    assert lib.agdiff_edge_scales(P, Tp, Wp, 1, st) == 0
    assert lib.agdiff_edge_encoder(P, _lib.ptr(ws.num_canon), etiles, _lib.ptr(ws.c_len), _lib.ptr(ws.c_type),
                                   _lib.ptr(ws.e_attr), None, None, _lib.ptr(ws.c_pos), _lib.ptr(ws.c_mir), st) == 0
    torch.cuda.synchronize()
    # the reference's own renormalised embedding rows (O.synth_state_dict_for copy, renormed like the module's)
    O.embedding_renorm_(sd["encoder_global.embedding.weight"], t(g["atom_type"]))
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    return g, cfg, sd64, m, lib, topo, ws, (P, Tp, Wp, st)


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, 1e-5)


def _xs_ref(sd, blk, h):
    """LeakyReLU(BN(lin1(h))) of conv1 | conv2 in float64 (schnet.py:153-155)."""
    return torch.cat([F.leaky_relu(_bn(sd, "%s.conv%d.norm1" % (blk, c), _lin(sd, "%s.conv%d.lin1" % (blk, c), h)), 0.2)
                      for c in (1, 2)], dim=1)


def _agg_ref(sd, cfg, blk, g, xs):
    """sum_e x[src] * W_e for conv1 | conv2 in float64, from the fixture's edges / lengths / edge_attr."""
    ei, d, ea = t(g["edge_index"]), t(g["edge_length"]).double().view(-1), t(g["edge_attr"]).double()
    out = []
    for c, lo, hi in ((1, 0, 128), (2, 128, 192)):
        p = "%s.conv%d" % (blk, c)
        lw = torch.sigmoid(_lin(sd, p + ".distance_weighting.layer2",
                                F.relu(_lin(sd, p + ".distance_weighting.layer1", d.view(-1, 1))))).view(-1)
        if cfg.smooth_conv:
            C = 0.5 * (torch.cos(d * math.pi / cfg.cutoff) + 1.0)
        else:
            C = torch.exp(-((d - cfg.cutoff) ** 2) / (2 * cfg.cutoff ** 2))
        C = C * (d <= cfg.cutoff) * (d >= 0.0)
        W = _lin(sd, p + ".nn.2", F.softplus(sd[p + ".nn.1.beta"] * _lin(sd, p + ".nn.0", ea)) - math.log(2.0))
        W = W * (lw * C).view(-1, 1)
        x = xs[:, lo:hi]
        out.append(torch.zeros_like(x).index_add_(0, ei[1], x[ei[0]] * W))
    return torch.cat(out, dim=1)


def _device_agg(lib, topo, ws):
    """agg[node] plus the partial sums its later chunks kept in agg_first (csrc/edge.hip, k_cfconv_fused; the node
    stage adds them the same way, csrc/node.hip)."""
    from agdiff_amd import _lib
    ce = _lib.TILE * lib.agdiff_conv_chunk_tiles(ctypes.c_int64(topo.max_edges))
    ip = ws.in_ptr.cpu().numpy().astype(np.int64)
    agg = ws.agg.view(-1, 192).cpu().double().numpy().copy()
    first = ws.agg_first.view(-1, 192).cpu().double().numpy()
    for i in range(topo.N):
        lo, hi = ip[i], ip[i + 1]
        if hi <= lo:
            agg[i] = 0.0
            continue
        for c in range(lo // ce + 1, (hi - 1) // ce + 1):
            agg[i] += first[c]
    return torch.from_numpy(agg)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("case", STAGE_CASES[:2])
def test_cfconv_fused_block0_vs_reference_modules(case, precision):
    g, cfg, sd, m, lib, topo, ws, (P, Tp, Wp, st) = _setup(case, precision)
    blk = "encoder_global.interactions.0"
    h0 = t(g["schnet_h0"]).double()
    assert lib.agdiff_schnet_node_stage(P, Tp, Wp, 0, st) == 0
    torch.cuda.synchronize()
    check_close("node_stage0 h[%s]" % case, ws.h.view(-1, 128), g["schnet_h0"], precision)
    xs_ref = _xs_ref(sd, blk, h0)
    check_close("node_stage0 xs[%s]" % case, ws.xs.view(-1, 192), xs_ref.float(), precision)
    assert lib.agdiff_cfconv_fused(P, Tp, Wp, 0, st) == 0
    torch.cuda.synchronize()
    agg = _device_agg(lib, topo, ws)
    # the host decomposition used as the aggregate's reference reproduces the reference module outputs
    agg_ref = _agg_ref(sd, cfg, blk, g, xs_ref)
    for c, lo, hi, key in ((1, 0, 128, "cfconv1_b0"), (2, 128, 192, "cfconv2_b0")):
        p = "%s.conv%d" % (blk, c)
        post = lambda a: _bn(sd, p + ".norm2", _lin(sd, p + ".lin2", a[:, lo:hi])).float().numpy()
        assert rel_err(post(agg_ref), g[key]) < 5e-6
        check_close("cfconv_fused conv%d out[%s]" % (c, case), post(agg), g[key], precision)
    check_close("cfconv_fused agg[%s]" % case, agg.float().numpy(), agg_ref.float().numpy(), precision)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("case", STAGE_CASES[:2])
def test_node_stage_block0_vs_reference_modules(case, precision):
    """InteractionBlock (lin2/BN, ssp, lin, gate) + AdaptiveScaling + residual of block 0 from a reference aggregate."""
    g, cfg, sd, m, lib, topo, ws, (P, Tp, Wp, st) = _setup(case, precision)
    blk = "encoder_global.interactions.0"
    h0 = t(g["schnet_h0"]).double()
    assert lib.agdiff_schnet_node_stage(P, Tp, Wp, 0, st) == 0
    agg_ref = _agg_ref(sd, cfg, blk, g, _xs_ref(sd, blk, h0))
    ws.agg.view(-1, 192)[: topo.N].copy_(agg_ref.float().cuda())
    ws.agg_first.zero_()
    assert lib.agdiff_schnet_node_stage(P, Tp, Wp, 1, st) == 0
    torch.cuda.synchronize()
    h1_ref = g["schnet_h0"].astype(np.float64) + g["scaled_b0"].astype(np.float64)
    check_close("node_stage1 h[%s]" % case, ws.h.view(-1, 128), h1_ref.astype(np.float32), precision)
    # and the part of it that is new (h1 - h0 = AdaptiveScaling(InteractionBlock(h0))) on its own scale
    check_close("node_stage1 h-h0[%s]" % case, ws.h.view(-1, 128).cpu().double().numpy() - g["schnet_h0"],
                g["scaled_b0"], precision, scale=4.0)      # difference of two O(1) fp32 numbers, |result| ~ 0.3
    xs1 = _xs_ref(sd, "encoder_global.interactions.1", torch.from_numpy(h1_ref))
    check_close("node_stage1 xs[%s]" % case, ws.xs.view(-1, 192), xs1.float(), precision)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_node_stage_flags_the_rows_it_takes_out_of_the_split_fp16_range(precision):
    """agdiff_ws_t.range_rows (round 6): in the split-fp16 mode the node stage flags a node whose hidden activations reach 65000
    or whose new state |h| passes 255 (the heads multiply two states) WHERE it computes them -- sticky until the host's poll clears
    it, whatever the tensors show by then.  Aggregates blown up for two nodes: exactly those two are flagged; the other modes
    (fp32's exponent range) write no flag; a second, tame launch leaves the flags standing."""
    case = STAGE_CASES[0]
    g, cfg, sd, m, lib, topo, ws, (P, Tp, Wp, st) = _setup(case, precision)
    blk = "encoder_global.interactions.0"
    h0 = t(g["schnet_h0"]).double()
    assert lib.agdiff_schnet_node_stage(P, Tp, Wp, 0, st) == 0
    agg_ref = _agg_ref(sd, cfg, blk, g, _xs_ref(sd, blk, h0)).float().cuda()
    ws.agg_first.zero_()
    ws.range_rows.zero_()
    ws.agg.view(-1, 192)[: topo.N].copy_(agg_ref)
    assert lib.agdiff_schnet_node_stage(P, Tp, Wp, 1, st) == 0
    torch.cuda.synchronize()
    assert int(ws.range_rows.sum()) == 0                                  # (the reference's aggregates: nothing near the range)
    hot = [1, topo.N - 2]
    big = agg_ref.clone()
    big[hot] *= 1e7
    ws.agg.view(-1, 192)[: topo.N].copy_(big)
    assert lib.agdiff_schnet_node_stage(P, Tp, Wp, 1, st) == 0
    torch.cuda.synchronize()
    flagged = torch.nonzero(ws.range_rows[: topo.N]).flatten().cpu().tolist()
    assert flagged == (hot if precision == "f16x3" else []), (precision, flagged)
    ws.agg.view(-1, 192)[: topo.N].copy_(agg_ref)                         # a tame launch afterwards: the flags stay
    assert lib.agdiff_schnet_node_stage(P, Tp, Wp, 1, st) == 0
    torch.cuda.synchronize()
    assert torch.nonzero(ws.range_rows[: topo.N]).flatten().cpu().tolist() == flagged
    if precision == "f16x3":                                              # ... until the poll reports the owning graphs and clears them
        ba = t(g["batch"]).cuda()
        rep = m.range_report(ws, ba)
        assert rep is not None and sorted(rep[3]) == sorted(set(int(ba[i]) for i in hot)) and int(ws.range_rows.sum()) == 0


@pytest.mark.parametrize("layout", ["by_batch_size", "quad_tiles", "quads_per_target"])
@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("case", STAGE_CASES[:2])
def test_cfconv_node_block0_vs_reference_modules(case, precision, layout):
    """The CFConv kernels bench.py times (polynomial filters; k_cfconv_quad: radius rows and local rows in quad tiles, what every
    batch above 6,144 atoms runs -- forced onto the small fixture by group_targets = 4; k_cfconv_node: every target its own
    radius tiles + local quad tiles, what small batches and tune_cfconv_quad_tiles = -1 run) against the reference's CFConv
    modules of block 0 -- not only against the MLP kernel (tests/test_hip_poly.py)."""
    from agdiff_amd import _lib
    gt = None if layout == "by_batch_size" else 4
    g, cfg, sd, m, lib, topo, ws, (P, Tp, Wp, st) = _setup(case, precision, group_targets=gt,
                                                             tuning={"cfconv_quad_tiles": -1} if layout == "quads_per_target" else None)
    if "cfconv1_b0" not in g:
        pytest.skip("fixture without per-module outputs")
    pk = m.packed()
    assert pk.poly_kt >= 1 and lib.agdiff_local_poly_enabled(P, Tp, Wp) == 1
    ws.variant_log.zero_()
    blk = "encoder_global.interactions.0"
    pos = t(g["pos"]).cuda().contiguous()
    assert lib.agdiff_graph_build_scaled(P, Tp, Wp, _lib.ptr(pos), ctypes.c_float(cfg.cutoff), 0, st) == 0
    assert lib.agdiff_local_lengths(Tp, Wp, _lib.ptr(pos), st) == 0
    assert lib.agdiff_edge_scales_split(P, Tp, Wp, 2, st) == 0
    assert lib.agdiff_schnet_node_stage(P, Tp, Wp, 0, st) == 0
    ws.agg.fill_(float("nan"))
    assert lib.agdiff_cfconv_node(P, Tp, Wp, 0, st) == 0
    torch.cuda.synchronize()
    assert bool(int(ws.variant_log.item()) & _lib.DEFINES["AGDIFF_VAR_CFCONV_NODE_QUAD"]) == (layout == "quad_tiles")
    agg = ws.agg.view(-1, 192)[: topo.N].cpu().double()
    xs_ref = _xs_ref(sd, blk, t(g["schnet_h0"]).double())
    agg_ref = _agg_ref(sd, cfg, blk, g, xs_ref)
    for c, lo, hi, key in ((1, 0, 128, "cfconv1_b0"), (2, 128, 192, "cfconv2_b0")):
        p = "%s.conv%d" % (blk, c)
        post = lambda a: _bn(sd, p + ".norm2", _lin(sd, p + ".lin2", a[:, lo:hi])).float().numpy()
        check_close("cfconv_node conv%d out[%s]" % (c, case), post(agg), g[key], precision)
    check_close("cfconv_node agg[%s]" % case, agg.float().numpy(), agg_ref.float().numpy(), precision)
    # ... and the node stage fed with THIS aggregate (split bit 1: one complete row per node) closes block 0
    assert lib.agdiff_schnet_node_stage_split(P, Tp, Wp, 1, 1, st) == 0
    torch.cuda.synchronize()
    h1_ref = g["schnet_h0"].astype(np.float64) + g["scaled_b0"].astype(np.float64)
    check_close("cfconv_node -> node_stage1 h[%s]" % case, ws.h.view(-1, 128), h1_ref.astype(np.float32), precision)


# ---------------------------------------------------------------------------------------------------- G4 fixtures
def _center_np(x, batch):
    out = x.copy()
    for gidx in np.unique(batch):
        sel = batch == gidx
        out[sel] -= x[sel].mean(axis=0, keepdims=True)
    return out


def _step_args(pos, out, scratch, noise, *, step_size, use_global, clip, clip_local, w_global=1.0):
    from agdiff_amd import _lib
    sa = _lib.StepArgs()
    sa.pos_in, sa.pos_out, sa.scratch, sa.noise = _lib.ptr(pos), _lib.ptr(out), _lib.ptr(scratch), _lib.ptr(noise)
    sa.traj_out = ctypes.c_void_p(0)
    sa.sigma, sa.step_size, sa.noise_scale = 1.0, float(step_size), 0.0
    sa.w_global, sa.clip, sa.clip_local, sa.clip_pos = float(w_global), float(clip), float(clip_local), -1.0
    sa.use_global = int(use_global)
    return sa


@pytest.mark.parametrize("front", ["unfused", "fused"])
@pytest.mark.parametrize("case", ["g3_forward_qm9_small", "g3_forward_drugs_capped"])
def test_update_kernels_vs_reference_geometry_fixtures(case, front):
    """agdiff_langevin_update (unfused) and the update phase of agdiff_sampler_front (fused) against the reference's own
    eq_transform / clip_norm / center_pos outputs (G4: `eq_local`, `eq_global`, `clip_local_20`, `center` of the forward
    fixtures, tests/golden/make_golden.py:113-116), fed with the REFERENCE's edge scores -- not only through sampler
    trajectories.  With sigma = 1, zero noise and step size S the kernels return center_pos(pos + S * term): S = 2^16
    makes the term the whole result (S * |term| >> |pos|), S = 0 leaves center_pos alone."""
    from agdiff_amd import _lib
    g, cfg, sd, m, lib, topo, ws, (P, Tp, Wp, st) = _setup(case, "f32")
    RS = _lib.DEFINES["AGDIFF_RAD_STRIDE"]
    N, batch = topo.N, g["batch"]
    pos = t(g["pos"]).cuda().contiguous()
    E = int(ws.num_edges.item())
    assert E == g["edge_type"].shape[0]
    inv_g_ref = t(g["edge_inv_global"]).view(-1).cuda()
    inv_l_ref = t(g["edge_inv_local"]).view(-1).cuda()
    assert lib.agdiff_local_lengths(Tp, Wp, _lib.ptr(pos), st) == 0
    if front == "fused":
        sa0 = _lib.StepArgs()
        sa0.pos_in = _lib.ptr(pos)
        ws.canon_counter.zero_()
        assert lib.agdiff_sampler_front(P, Tp, Wp, ctypes.byref(sa0), 2 | 4, ctypes.c_float(cfg.cutoff), st) == 0
        torch.cuda.synchronize()
        # the reference's global scores by radius row: row (i, k) is the k-th radius (type 0) in-edge of target i
        where = {(int(s_), int(d_)): q_ for q_, (s_, d_) in enumerate(zip(g["edge_index"][0], g["edge_index"][1]))}
        cnt, rsrc = ws.rad_cnt.cpu().numpy(), ws.rad_src.view(N, RS).cpu().numpy()
        inv_r = np.zeros((N, RS), dtype=np.float32)
        for i in range(N):
            for k in range(cnt[i]):
                q_ = where[(int(rsrc[i, k]), i)]
                assert g["edge_type"][q_] == 0
                inv_r[i, k] = g["edge_inv_global"][q_, 0]
        assert int(cnt.sum()) == int((g["edge_type"] == 0).sum())
        ws.inv_r.view(N, RS).copy_(t(inv_r).cuda())
    else:
        ws.e_inv_global.zero_()
        ws.e_inv_global[ws.ref2dst[:E].long()] = inv_g_ref           # reference (row, col) order -> destination-sorted
    out, scratch, zero = torch.empty_like(pos), torch.empty_like(pos), torch.zeros_like(pos)

    def run(l_scores, S, use_global, clip_local):
        ws.l_inv[: topo.L].copy_(l_scores)
        ws.nan_flag.zero_()
        sa = _step_args(pos, out, scratch, zero, step_size=S, use_global=use_global, clip=1e30, clip_local=clip_local)
        if front == "fused":
            assert lib.agdiff_sampler_front(P, Tp, Wp, ctypes.byref(sa), 1, ctypes.c_float(cfg.cutoff), st) == 0
        else:
            assert lib.agdiff_langevin_update(Tp, Wp, ctypes.byref(sa), st) == 0
        torch.cuda.synchronize()
        assert int(ws.nan_flag[0].item()) == 0
        return out.cpu().double().numpy()

    S = 65536.0
    p64 = g["pos"].astype(np.float64)
    expect = lambda term: _center_np(p64 + S * term.astype(np.float64), batch)
    check_close("%s update eq_local[%s]" % (front, case), run(inv_l_ref, S, 0, -1.0), expect(g["eq_local"]), "f32")
    check_close("%s update eq_global[%s]" % (front, case), run(torch.zeros_like(inv_l_ref), S, 1, -1.0), expect(g["eq_global"]), "f32")
    # clip_norm(eq_local * 1e4, 20): eq_transform is linear in the scores
    check_close("%s update clip_local_20[%s]" % (front, case), run(inv_l_ref * 1e4, S, 0, 20.0), expect(g["clip_local_20"]), "f32")
    check_close("%s update center[%s]" % (front, case), run(inv_l_ref, 0.0, 0, -1.0), g["center"], "f32")


def test_noise_mode_per_step_consumes_the_generator_like_the_reference():
    """noise_mode="per_step": one torch.randn_like(pos) per denoising step, the reference's own call (dualenc.py:529) --
    so a run equals, bit for bit, the run with those same draws injected, and the generator ends where the reference's
    would (missing item 3)."""
    from agdiff_amd import get_model, qm9_model_config, synth
    cfg = qm9_model_config(num_diffusion_timesteps=12)
    m = get_model(cfg)
    m.load_state_dict(synth.synth_state_dict(m.state_dict()))
    m = m.to("cuda:0").eval()
    b = synth.make_packed_batch("qm9", 3, 2, seed=7)
    at, bi, bt, ba = [t(b[k]).cuda() for k in ("atom_type", "bond_index", "bond_type", "batch")]
    pos_init = torch.randn(at.shape[0], 3, generator=torch.Generator().manual_seed(3)).cuda()
    kw = dict(extend_order=False, n_steps=5, w_global=1.0, global_start_sigma=0.5, clip=1000.0)
    torch.manual_seed(1234)
    p1, t1 = m.langevin_dynamics_sample_diffusion(at, pos_init, bi, bt, ba, b["num_graphs"], noise_mode="per_step", **kw)
    after1 = torch.randn(4, device="cuda:0")
    torch.manual_seed(1234)
    draws = torch.stack([torch.randn_like(pos_init) for _ in range(5)])
    after2 = torch.randn(4, device="cuda:0")
    p2, t2 = m.langevin_dynamics_sample_diffusion(at, pos_init, bi, bt, ba, b["num_graphs"], noise=draws, **kw)
    assert torch.equal(p1, p2) and all(torch.equal(a, b_) for a, b_ in zip(t1, t2))
    assert torch.equal(after1, after2)
