#!/usr/bin/env python3
"""Prototype harness for the mirror-sharing CFConv (csrc/pairs.hip): builds the pair-sweep order on the host from the
device graph of the bench batch, runs the new kernels next to agdiff_cfconv_fused on the same inputs, compares the
aggregates and times both.   python tools/proto_run.py [--workload drugs|qm9] [--mols 8] [--copies 128] [--max-tiles 8]"""
import argparse, ctypes, json, os, sys, time
import numpy as np
os.environ["AGDIFF_RADIUS_POLY"] = "off"    # this experiment compares against the one-list product kernel (agdiff_cfconv_fused)
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from proto_pairs import build_pair_sweeps, wave_partition   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="drugs")
ap.add_argument("--mols", type=int, default=8)
ap.add_argument("--copies", type=int, default=128)
ap.add_argument("--max-tiles", type=int, default=8)
ap.add_argument("--precision", default="bf16x3")
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()

from agdiff_amd import _lib, drugs_model_config, get_model, qm9_model_config, synth   # noqa: E402
lib = _lib.load()
proto = ctypes.CDLL(os.path.join(HERE, "libagdiff_proto.so"))       # make -C experiments/mirror_pairs
dev = torch.device("cuda", 0)
cfg = (qm9_model_config if args.workload == "qm9" else drugs_model_config)(beta_end=2e-5)
model = get_model(cfg)
model.precision = args.precision
model.load_state_dict(synth.synth_state_dict(model.state_dict()))
model = model.to(dev).eval()
b = synth.make_packed_batch(args.workload, args.mols, args.copies, seed=2021)
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
at, bi, bt, ba = T(b["atom_type"]), T(b["bond_index"]), T(b["bond_type"]), T(b["batch"])
g = torch.Generator().manual_seed(2021)
pos_init = torch.randn(at.shape[0], 3, generator=g).to(dev)
run = model.begin_sampling(at, pos_init, bi, bt, ba, b["num_graphs"], False, n_steps=10, w_global=1.0,
                           global_start_sigma=0.5, save_traj=False)
run.advance(10)
torch.cuda.synchronize()
ws, topo, pk = run.ws, run.topo, run.pk
P, Tp, Wp = ctypes.byref(pk.struct), ctypes.byref(topo.struct), ctypes.byref(ws.struct)
st = _lib.stream_ptr()
E, C = int(ws.num_edges.item()), int(ws.num_canon.item())
print("N %d  E %d  canonical %d (%.1f %%)" % (topo.N, E, C, 100.0 * C / E))

t0 = time.time()
r = build_pair_sweeps(ws.c_src[:C].cpu().numpy(), ws.c_dst[:C].cpu().numpy(), ws.c_mir[:C].cpu().numpy(),
                      topo.graph_ptr.cpu().numpy(), max_tiles=args.max_tiles)
R, S, I = r["rows"], len(r["seg_dst"]), len(r["item_tiles"])
print("host order: %.1f s; rows %d (%.1f %% padding), tiles %d (now %d), segments %d, items %d"
      % (time.time() - t0, R, 100.0 * (R - C) / R, R // 16, (E + 15) // 16, S, I))
i32 = lambda x: torch.from_numpy(np.ascontiguousarray(x).astype(np.int32)).to(dev)
p_src, p_dst, p_slot, p_seg = i32(r["p_src"]), i32(r["p_dst"]), i32(r["p_slot"]), i32(r["p_seg"])
seg_ptr, item_row0, item_tiles = i32(r["seg_ptr"]), i32(r["item_row0"]), i32(r["item_tiles"])
num_waves = 256 * 8
wave_ptr = i32(wave_partition(r["item_tiles"], num_waves))
p_can = torch.from_numpy(r["p_can"]).to(dev)
valid = p_can >= 0
# e_attr in row order: the encoder writes canonical edge e to row pos_index[e]
row_of_can = torch.empty(C, dtype=torch.int32, device=dev)
row_of_can[p_can[valid]] = torch.nonzero(valid).flatten().to(torch.int32)
e_attr2 = torch.zeros((R // 16) * 2048, dtype=torch.float32, device=dev)
nomir = torch.full((C,), -1, dtype=torch.int32, device=dev)
etiles = (topo.max_edges + 15) // 16
rc = lib.agdiff_edge_encoder(P, _lib.ptr(ws.num_canon), etiles, _lib.ptr(ws.c_len), _lib.ptr(ws.c_type), _lib.ptr(e_attr2),
                             None, None, _lib.ptr(row_of_can), _lib.ptr(nomir), st)
assert rc == 0
epad = etiles * 16
K = 0                                                         # block
c_pos = ws.c_pos[:C].long()
scales = []
for c in (0, 1):
    sfull = ws.e_scale[(2 * K + c) * epad:(2 * K + c + 1) * epad]
    srow = torch.zeros(R, dtype=torch.float32, device=dev)
    srow[valid] = sfull[c_pos[p_can[valid]]]
    scales.append(srow)
assert lib.agdiff_schnet_node_stage(P, Tp, Wp, 0, st) == 0
torch.cuda.synchronize()

agg_seg = torch.zeros(S * 192, dtype=torch.float32, device=dev)
mir_rows = torch.zeros(I * 16 * 192, dtype=torch.float32, device=dev)


fnf = proto.agdiff_proto_cfconv_pairs_fused
fnf.restype = ctypes.c_int
fnf.argtypes = [ctypes.c_void_p, ctypes.c_int32] + [ctypes.c_void_p] * 14 + [ctypes.c_int32, ctypes.c_void_p]


def newf():
    rc = fnf(P, K, _lib.ptr(p_src), _lib.ptr(p_dst), _lib.ptr(p_slot), _lib.ptr(p_seg), _lib.ptr(seg_ptr),
             _lib.ptr(scales[0]), _lib.ptr(scales[1]), _lib.ptr(e_attr2), _lib.ptr(ws.xs), _lib.ptr(agg_seg),
             _lib.ptr(mir_rows), _lib.ptr(item_row0), _lib.ptr(item_tiles), _lib.ptr(wave_ptr), num_waves, st)
    assert rc == 0, rc


def old():
    assert lib.agdiff_cfconv_fused(P, Tp, Wp, K, st) == 0


old(); newf()
torch.cuda.synchronize()
# reference aggregate: agg + agg_first partials (as the node stage adds them)
ce = 16 * lib.agdiff_conv_chunk_tiles(ctypes.c_int64(topo.max_edges))
ip = ws.in_ptr.cpu().numpy().astype(np.int64)
ref = ws.agg.view(-1, 192)[: topo.N].clone()
first = ws.agg_first.view(-1, 192)
lo, hi = ip[:-1], ip[1:]
has = hi > lo
clo, chi = lo // ce, np.where(has, (hi - 1) // ce, lo // ce)
ref[torch.from_numpy(~has).to(dev)] = 0
for i in np.nonzero(chi > clo)[0]:
    for c in range(clo[i] + 1, chi[i] + 1):
        ref[i] += first[c]
slots_atom = (torch.from_numpy(r["item_j0"]).to(dev)[:, None] + torch.arange(16, device=dev)[None, :]).reshape(-1)
okslot = slots_atom < topo.N


def assemble():
    got = torch.zeros(topo.N, 192, device=dev)
    got.index_add_(0, torch.from_numpy(r["seg_dst"]).to(dev), agg_seg.view(S, 192))
    got.index_add_(0, slots_atom[okslot], mir_rows.view(I * 16, 192)[okslot])
    return got


errf = float((assemble() - ref).abs().max() / ref.abs().max())
print("aggregate: max|pairs - product| / max|product| = %.3e   (max|product| %.3e)" % (errf, float(ref.abs().max())))


def timeit(f, reps):
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# the OLD kernel's code on the NEW row order (segments as pseudo-targets; direct sums only, no mirror): separates what
# the order / data layout costs from what the new tile body costs
import copy
topo2, ws2 = _lib.Topo(), _lib.Workspace()
ctypes.memmove(ctypes.byref(topo2), ctypes.byref(topo.struct), ctypes.sizeof(topo2))
ctypes.memmove(ctypes.byref(ws2), ctypes.byref(ws.struct), ctypes.sizeof(ws2))
topo2.max_edges = R
ct2 = lib.agdiff_conv_chunk_tiles(ctypes.c_int64(R))
nR = torch.tensor([R], dtype=torch.int32, device=dev)
esc2 = torch.zeros(2 * _lib.DEFINES["AGDIFF_MAX_CONVS"] * R, dtype=torch.float32, device=dev)
esc2[(2 * K) * R:(2 * K + 1) * R] = scales[0]
esc2[(2 * K + 1) * R:(2 * K + 2) * R] = scales[1]
agg2 = torch.zeros(S * 192, dtype=torch.float32, device=dev)
first2 = torch.zeros(((R // 16 + ct2 - 1) // ct2 + 1) * 192, dtype=torch.float32, device=dev)
for f, tns in (("num_edges", nR), ("in_ptr", seg_ptr), ("e_src", p_src), ("e_dst", p_seg), ("e_scale", esc2),
               ("e_attr", e_attr2), ("agg", agg2), ("agg_first", first2)):
    setattr(ws2, f, _lib.ptr(tns))


def old_on_new():
    assert lib.agdiff_cfconv_fused(P, ctypes.byref(topo2), ctypes.byref(ws2), K, st) == 0


old_on_new()
torch.cuda.synchronize()
d = float((agg2.view(S, 192) - agg_seg.view(S, 192)).abs().max() / agg_seg.abs().max())
print("old code on the new order: direct sums vs new kernel %.2e (chunk tiles %d)" % (d, ct2))

# the product kernel on its own order, cut to the same number of rows (fixed launch costs vs per-tile costs)
topo3, ws3 = _lib.Topo(), _lib.Workspace()
ctypes.memmove(ctypes.byref(topo3), ctypes.byref(topo.struct), ctypes.sizeof(topo3))
ctypes.memmove(ctypes.byref(ws3), ctypes.byref(ws.struct), ctypes.sizeof(ws3))
topo3.max_edges = R
esc3 = torch.zeros(2 * _lib.DEFINES["AGDIFF_MAX_CONVS"] * R, dtype=torch.float32, device=dev)
for c in (0, 1):
    esc3[(2 * K + c) * R:(2 * K + c + 1) * R] = ws.e_scale[(2 * K + c) * epad:(2 * K + c) * epad + R]
first3 = torch.zeros_like(first2)
setattr(ws3, "num_edges", _lib.ptr(nR)); setattr(ws3, "e_scale", _lib.ptr(esc3)); setattr(ws3, "agg_first", _lib.ptr(first3))


def old_cut():
    assert lib.agdiff_cfconv_fused(P, ctypes.byref(topo3), ctypes.byref(ws3), K, st) == 0


for rep in range(3):
    print("product kernel on its own order, first %d rows only: %.4f ms" % (R, timeit(old_cut, args.reps)))
    print("product kernel code on the pair order (direct sums only): %.4f ms" % timeit(old_on_new, args.reps))
    to = timeit(old, args.reps)
    tf = timeit(newf, args.reps)
    print("product k_cfconv_fused %.4f ms   pairs kernel %.4f ms (%.1f %%)" % (to, tf, 100.0 * tf / to))
print(json.dumps({"workload": args.workload, "E": E, "C": C, "rows": R, "items": I, "segments": S, "old_ms": to,
                  "pairs_ms": tf, "rel_err": errf, "max_tiles": args.max_tiles}))
