#!/usr/bin/env python3
"""Per-launch times (ms, events on the launch stream) of the split CFConv's kernels on the bench batch, next to the
one-list kernel on the same graph: python tools/split_time.py [mols copies [kind]]"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from agdiff_amd import _lib, drugs_model_config, qm9_model_config, get_model, synth  # noqa: E402


def main():
    mols = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    copies = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    kind = sys.argv[3] if len(sys.argv) > 3 else "drugs"
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    cfg = (qm9_model_config if kind == "qm9" else drugs_model_config)(beta_end=2e-5)
    m = get_model(cfg)
    m.load_state_dict(synth.synth_state_dict(m.state_dict()))
    m = m.to(dev).eval()
    b = synth.make_packed_batch(kind, mols, copies, seed=2021)
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    at, bi, bt, ba = T(b["atom_type"]), T(b["bond_index"]), T(b["bond_type"]), T(b["batch"])
    pos_init = torch.randn(at.shape[0], 3, generator=torch.Generator().manual_seed(1)).to(dev)
    run = m.begin_sampling(at, pos_init, bi, bt, ba, b["num_graphs"], False, n_steps=30, step_lr=1e-6, clip=1000.0,
                           global_start_sigma=0.5, w_global=1.0, save_traj=False, nan_check_every=10 ** 9)
    run.advance(30)
    torch.cuda.synchronize()
    ws, topo, pk = run.ws, run.topo, run.pk
    P, Tp, Wp = ctypes.byref(pk.struct), ctypes.byref(topo.struct), ctypes.byref(ws.struct)
    st = _lib.stream_ptr()
    ops = {"poly_kt": pk.poly_kt, "N": topo.N, "E": int(ws.num_edges.item()), "R": int(ws.num_rad.item()), "L": topo.L,
           "canon": int(ws.num_canon.item())}

    def timeit(name, fn, reps=10):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ops[name] = e0.elapsed_time(e1) / reps
    nc = cfg.num_convs
    et = (topo.max_edges + _lib.TILE - 1) // _lib.TILE
    ct = (topo.Lc + _lib.TILE - 1) // _lib.TILE
    timeit("graph_build", lambda: lib.agdiff_graph_build(Tp, Wp, run.pos_p, ctypes.c_float(cfg.cutoff), st))
    timeit("scales_radius", lambda: lib.agdiff_edge_scales_split(P, Tp, Wp, 0, st))
    timeit("scales_local", lambda: lib.agdiff_edge_scales_split(P, Tp, Wp, 1, st))
    timeit("local_encoder_frag_rows", lambda: lib.agdiff_edge_encoder(
        P, _lib.ptr(ws.num_local_canon), ct, _lib.ptr(ws.lc_len), _lib.ptr(topo.lc_type), _lib.ptr(ws.l_attr_frag),
        _lib.ptr(ws.l_attr_rows), _lib.ptr(topo.lp_row), _lib.ptr(topo.lc_ppos), _lib.ptr(topo.lc_pmir), st))
    timeit("local_edge_rows", lambda: lib.agdiff_local_edge_rows(P, Tp, Wp, st))
    timeit("cfconv_radius_x%d" % nc, lambda: [lib.agdiff_cfconv_radius(P, Tp, Wp, k, st) for k in range(nc)])
    timeit("cfconv_local_x%d" % nc, lambda: [lib.agdiff_cfconv_local(P, Tp, Wp, k, st) for k in range(nc)])
    timeit("node_stage_split_x%d" % (nc + 1), lambda: [lib.agdiff_schnet_node_stage_split(P, Tp, Wp, k, 1, st) for k in range(nc + 1)])
    timeit("head_poly", lambda: lib.agdiff_pair_head_poly(P, _lib.ptr(ws.num_canon), et, _lib.ptr(ws.c_src), _lib.ptr(ws.c_dst),
                                                         _lib.ptr(ws.c_len), _lib.ptr(ws.h), _lib.ptr(ws.c_pos), _lib.ptr(ws.c_mir),
                                                         _lib.ptr(ws.e_inv_global), st))
    timeit("local_branch", lambda: lib.agdiff_score_forward(P, Tp, Wp, run.pos_p, 0, st))
    timeit("score_forward_global_sampler", lambda: lib.agdiff_score_forward(P, Tp, Wp, run.pos_p, 1 | 8, st))
    timeit("score_forward_global_full_head", lambda: lib.agdiff_score_forward(P, Tp, Wp, run.pos_p, 1, st))
    # the one-list path on the same graph (the full forward above has left e_attr / e_scale behind)
    lib.agdiff_edge_scales(P, Tp, Wp, 1, st)
    timeit("edge_encoder_all", lambda: lib.agdiff_edge_encoder(P, _lib.ptr(ws.num_canon), et, _lib.ptr(ws.c_len), _lib.ptr(ws.c_type),
                                                              _lib.ptr(ws.e_attr), None, None, _lib.ptr(ws.c_pos), _lib.ptr(ws.c_mir), st))
    timeit("cfconv_fused_x%d" % nc, lambda: [lib.agdiff_cfconv_fused(P, Tp, Wp, k, st) for k in range(nc)])
    timeit("head_global_exact", lambda: lib.agdiff_pair_head(ctypes.byref(pk.struct.head_global), _lib.ptr(ws.num_canon), et,
                                                            _lib.ptr(ws.c_src), _lib.ptr(ws.c_dst), _lib.ptr(ws.h), _lib.ptr(ws.e_attr),
                                                            None, _lib.ptr(ws.c_pos), _lib.ptr(ws.c_mir), _lib.ptr(ws.e_inv_global), st))
    print(json.dumps(ops, indent=1))


if __name__ == "__main__":
    main()
