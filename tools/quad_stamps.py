#!/usr/bin/env python3
"""Start / end clocks of the 256 workgroups of one k_cfconv_quad launch (a debug build of nodeconv.hip:
bash tools/build_variant.sh nodeconv.hip stamps -DAG_QUAD_STAMPS; AGDIFF_LIB=$PWD/_ab/lib_stamps.so): how far apart the
workgroups finish, i.e. what a chunk queue across workgroups could still recover.   python tools/quad_stamps.py [--mols 36 --copies 128]"""
import argparse, ctypes, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from agdiff_amd import _lib, drugs_model_config, get_model, synth

ap = argparse.ArgumentParser()
ap.add_argument("--mols", type=int, default=36)
ap.add_argument("--copies", type=int, default=128)
args = ap.parse_args()
lib = _lib.load()
raw = ctypes.CDLL(os.environ["AGDIFF_LIB"])
dev = torch.device("cuda", 0)
cfg = drugs_model_config(beta_end=2e-5)
m = get_model(cfg)
m.load_state_dict(synth.synth_state_dict(m.state_dict()))
m = m.to(dev).eval()
b = synth.make_packed_batch("drugs", args.mols, args.copies, seed=2021)
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
at, bi, bt, ba = T(b["atom_type"]), T(b["bond_index"]), T(b["bond_type"]), T(b["batch"])
pos_init = torch.randn(at.shape[0], 3, generator=torch.Generator().manual_seed(2021)).to(dev)
run = m.begin_sampling(at, pos_init, bi, bt, ba, b["num_graphs"], False, n_steps=6, step_lr=1e-6, clip=1000.0,
                       global_start_sigma=0.5, w_global=1.0, save_traj=False)
run.advance(6)
torch.cuda.synchronize()
ws, topo, pk = run.ws, run.topo, run.pk
P, Tp, Wp, st = ctypes.byref(pk.struct), ctypes.byref(topo.struct), ctypes.byref(ws.struct), _lib.stream_ptr()
out = []
buf = (ctypes.c_ulonglong * 512)()
zero = (ctypes.c_ulonglong * 512)()
for k in range(cfg.num_convs):
    for rep in range(3):
        raw.agdiff_debug_quad_stamps(buf)            # (clears)
        _lib.check(lib.agdiff_cfconv_node(P, Tp, Wp, k, st), "agdiff_cfconv_node")
        torch.cuda.synchronize()
        assert raw.agdiff_debug_quad_stamps(buf) == 0
        a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 2).astype(np.float64)
        start, end = a[:, 0], a[:, 1]
        dur = end - start
        span = end.max() - start.min()
        if rep == 2:
            xcd = np.arange(256) % 8
            per_xcd = [round(float(dur[xcd == x].mean() / dur.mean()), 3) for x in range(8)]
            within = round(float(np.mean([dur[xcd == x].max() / dur[xcd == x].mean() for x in range(8)])), 3)
            # local share of the range's tiles vs duration
            w = topo.quad_wg_ptr.cpu().numpy().astype(np.int64)
            ltp = topo.lt_ptr.cpu().numpy().astype(np.int64)
            wgid = (np.arange(256) % 8) * 32 + np.arange(256) // 8          # the range a block owns (XCD remap)
            loc_tiles = ltp[w[wgid + 1]] - ltp[w[wgid]]
            corr = round(float(np.corrcoef(loc_tiles, dur)[0, 1]), 3)
            # radius tiles of a range (every molecule inside the cutoff: rows = min(n, 33) - 1 - local in-edges among the candidates)
            cnt = ws.rad_cnt.cpu().numpy().astype(np.int64)
            qt = topo.quad_tgt.cpu().numpy().astype(np.int64).reshape(-1, 4)
            rq = ((np.where(qt >= 0, cnt[np.maximum(qt, 0)], 0).max(axis=1) + 3) // 4)
            crq = np.concatenate([[0], np.cumsum(rq)])
            rad_tiles = crq[w[wgid + 1]] - crq[w[wgid]]
            A = np.stack([loc_tiles, rad_tiles], axis=1).astype(np.float64)
            coef, *_ = np.linalg.lstsq(A, dur, rcond=None)
            fit = {"ticks_per_local_tile": round(float(coef[0]), 3), "ticks_per_radius_tile": round(float(coef[1]), 3),
                   "local_over_radius": round(float(coef[0] / coef[1]), 3)}
            out.append({"mean_by_xcd": per_xcd, "slowest_over_mean_within_xcd": within, "corr_local_tiles_vs_duration": corr, "fit": fit,"block": k, "span_ticks": span, "mean_wg_ticks": float(dur.mean()), "max_wg_ticks": float(dur.max()),
                        "slowest_over_mean": float(dur.max() / dur.mean()), "span_over_mean": float(span / dur.mean()),
                        "start_spread_over_mean": float((start.max() - start.min()) / dur.mean())})
print(json.dumps({"N": topo.N, "launches": out}, indent=1))
