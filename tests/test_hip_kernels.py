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


@pytest.mark.parametrize("precision", PRECISIONS)
def test_mirror_sharing_cfconv_prototype_matches_product_kernel(precision):
    """csrc/pairs.hip (experiment, DESIGN.md §8.2: filter evaluated once per mirror pair, mirror sums accumulated in
    registers through a 0/1 selection MFMA): its aggregates equal the product kernel's on the capped Drugs-shaped
    fixture -- the direct sums and the per-slot mirror sums add up to agg[node] of agdiff_cfconv_fused."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from proto_pairs import build_pair_sweeps, wave_partition
    from agdiff_amd import _lib
    g, cfg, sd, m, lib, topo, ws, (P, Tp, Wp, st) = _setup("g3_forward_drugs_capped", precision)
    dev = "cuda:0"
    C = int(ws.num_canon.item())
    r = build_pair_sweeps(ws.c_src[:C].cpu().numpy(), ws.c_dst[:C].cpu().numpy(), ws.c_mir[:C].cpu().numpy(),
                          topo.graph_ptr.cpu().numpy(), max_tiles=3)
    R, S, I = r["rows"], len(r["seg_dst"]), len(r["item_tiles"])
    assert (r["p_slot"] >= 0).any() and (r["p_slot"][r["p_can"] >= 0] < 0).any()      # mirrored and unpaired rows
    i32 = lambda x: torch.from_numpy(np.ascontiguousarray(x).astype(np.int32)).to(dev)
    p_src, p_dst, p_slot, p_seg = i32(r["p_src"]), i32(r["p_dst"]), i32(r["p_slot"]), i32(r["p_seg"])
    seg_ptr, item_row0, item_tiles = i32(r["seg_ptr"]), i32(r["item_row0"]), i32(r["item_tiles"])
    num_waves = 24
    wave_ptr = i32(wave_partition(r["item_tiles"], num_waves))
    p_can = torch.from_numpy(r["p_can"]).to(dev)
    valid = p_can >= 0
    row_of_can = torch.empty(C, dtype=torch.int32, device=dev)
    row_of_can[p_can[valid]] = torch.nonzero(valid).flatten().to(torch.int32)
    e_attr2 = torch.zeros((R // 16) * 2048, dtype=torch.float32, device=dev)
    nomir = torch.full((C,), -1, dtype=torch.int32, device=dev)
    etiles = (topo.max_edges + 15) // 16
    assert lib.agdiff_edge_encoder(P, _lib.ptr(ws.num_canon), etiles, _lib.ptr(ws.c_len), _lib.ptr(ws.c_type),
                                   _lib.ptr(e_attr2), None, None, _lib.ptr(row_of_can), _lib.ptr(nomir), st) == 0
    epad, c_pos, scales = etiles * 16, ws.c_pos[:C].long(), []
    for c in (0, 1):
        srow = torch.zeros(R, dtype=torch.float32, device=dev)
        srow[valid] = ws.e_scale[c * epad:(c + 1) * epad][c_pos[p_can[valid]]]
        scales.append(srow)
    assert lib.agdiff_schnet_node_stage(P, Tp, Wp, 0, st) == 0
    assert lib.agdiff_cfconv_fused(P, Tp, Wp, 0, st) == 0
    fnf = lib.agdiff_proto_cfconv_pairs_fused
    fnf.restype = ctypes.c_int
    fnf.argtypes = [ctypes.c_void_p, ctypes.c_int32] + [ctypes.c_void_p] * 14 + [ctypes.c_int32, ctypes.c_void_p]
    agg_seg = torch.zeros(S * 192, dtype=torch.float32, device=dev)
    mir_rows = torch.zeros(I * 16 * 192, dtype=torch.float32, device=dev)
    assert fnf(P, 0, _lib.ptr(p_src), _lib.ptr(p_dst), _lib.ptr(p_slot), _lib.ptr(p_seg), _lib.ptr(seg_ptr),
               _lib.ptr(scales[0]), _lib.ptr(scales[1]), _lib.ptr(e_attr2), _lib.ptr(ws.xs), _lib.ptr(agg_seg),
               _lib.ptr(mir_rows), _lib.ptr(item_row0), _lib.ptr(item_tiles), _lib.ptr(wave_ptr), num_waves, st) == 0
    torch.cuda.synchronize()
    got = torch.zeros(topo.N, 192, device=dev, dtype=torch.float64)
    got.index_add_(0, torch.from_numpy(r["seg_dst"]).to(dev), agg_seg.view(S, 192).double())
    slots_atom = (torch.from_numpy(r["item_j0"]).to(dev)[:, None] + torch.arange(16, device=dev)[None, :]).reshape(-1)
    ok = slots_atom < topo.N
    got.index_add_(0, slots_atom[ok], mir_rows.view(I * 16, 192)[ok].double())
    ref = _device_agg(lib, topo, ws)
    # same arithmetic per edge in both kernels; only the order of the fp32 additions differs
    check_close("pairs prototype agg", got.float(), ref.float(), "f32")


@pytest.mark.gpu
@pytest.mark.parametrize("workload,mols,copies", [("drugs", 4, 3), ("qm9", 6, 4)])
def test_pair_tile_cfconv_prototype_matches_product_kernel(workload, mols, copies):
    """csrc/pairs4.hip (experiment, DESIGN.md §8.2, second design: 4 x 4 pair tiles, direct sums inside a lane, mirror
    sums by one reduce-scatter over the quarters): direct + mirror partial rows add up to agg[node] of
    agdiff_cfconv_fused for every molecule of the batch.  The harness (tools/proto_run4.py) builds the tables."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "proto_run4.py"), "--workload", workload, "--mols", str(mols),
                          "--copies", str(copies), "--max-atoms", "512", "--reps", "2"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["E_dense"] == rec["E"] and rec["pair_tiles"] > 0
    assert rec["rel_err"] < 1e-5, rec
