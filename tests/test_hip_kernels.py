"""GPU (MI355X): kernel-level parity of the SchNet block kernels through their own C-ABI entry points, against the
per-module reference fixtures (G2: CFConv F=128 / F=64, InteractionBlock + AdaptiveScaling of block 0,
tests/golden/make_golden.py:g_forward(stages=True)) -- so that rows a8-a10 of SURVEY.md §8 do not rest on the
six-block end-to-end `schnet_out` comparison alone.

  agdiff_schnet_node_stage(k=0)  -> xs  = LeakyReLU(BN(lin1(h0)))  of conv1 | conv2          (schnet.py:153-155)
  agdiff_cfconv_fused(k=0)       -> agg = sum_e x[src] * W_e       of conv1 | conv2          (schnet.py:138-162)
       checked as BN(lin2(agg)) against the reference's CFConv.forward outputs (the fixtures hold those)
  agdiff_schnet_node_stage(k=1)  -> h   = h0 + AdaptiveScaling(InteractionBlock(h0))         (schnet.py:201-234, 280)
       fed with a reference aggregate, checked against h0 + the reference's scaling_modules[0](interactions[0](h0))
"""
import ctypes
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import FORWARD_CASES, STAGE_CASES, check_close, load_golden, rel_err, t

pytestmark = pytest.mark.gpu
PRECISIONS = ["f32", "bf16x3"]


def _setup(case, precision):
    from agdiff_amd import _lib, get_model
    from oracle import agdiff_oracle as O
    g = load_golden(case)
    cfg = FORWARD_CASES[case]()
    sd = O.synth_state_dict_for(cfg)
    m = get_model(cfg)
    m.precision = precision
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
