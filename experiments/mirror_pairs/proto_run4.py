#!/usr/bin/env python3
"""Prototype harness for the 4 x 4 pair-tile CFConv (csrc/pairs4.hip): builds the static tile tables and the per-step
row data on the host from the device graph of the bench batch, runs the new kernel for the molecules up to --max-atoms
next to agdiff_cfconv_fused on the same inputs, compares the aggregates and times both.
   python tools/proto_run4.py [--workload drugs|qm9] [--mols 8] [--copies 128] [--max-atoms 56]"""
import argparse, ctypes, json, os, sys, time
import numpy as np
os.environ["AGDIFF_RADIUS_POLY"] = "off"    # this experiment compares against the one-list product kernel (agdiff_cfconv_fused)
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from proto_pairs4 import build_pair_tiles, pair_rows   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="drugs")
ap.add_argument("--mols", type=int, default=8)
ap.add_argument("--copies", type=int, default=128)
ap.add_argument("--max-atoms", type=int, default=56)
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
N = topo.N
E, C = int(ws.num_edges.item()), int(ws.num_canon.item())
gp = topo.graph_ptr.cpu().numpy().astype(np.int64)
sizes = np.diff(gp)
dense = np.nonzero(sizes <= args.max_atoms)[0]
e_src, e_dst = ws.e_src[:E].cpu().numpy().astype(np.int64), ws.e_dst[:E].cpu().numpy().astype(np.int64)
mol_of_edge = np.searchsorted(gp, e_dst, side="right") - 1
is_dense_mol = np.zeros(len(sizes), dtype=bool)
is_dense_mol[dense] = True
E_dense = int(is_dense_mol[mol_of_edge].sum())
print("N %d  E %d  canonical %d (%.1f %%);  molecules <= %d atoms: %d of %d, %d edges = %d tiles of the product kernel"
      % (N, E, C, 100.0 * C / E, args.max_atoms, len(dense), len(sizes), E_dense, (E_dense + 15) // 16))

num_waves = 256 * 8
t0 = time.time()
tabs = build_pair_tiles(gp, dense, num_waves)
TL, R = tabs["tiles"], tabs["tiles"] * 16
rt, rs, rdiag = pair_rows(tabs)
print("host tables: %.1f s; %d pair tiles (%.1f %% of the product's), %d groups, %d mirror row sets"
      % (time.time() - t0, TL, 100.0 * TL / ((E_dense + 15) // 16), tabs["n_groups"], tabs["n_sets"]))
# directed edge (src -> dst) -> position in the destination-sorted list (keys ascending by construction)
keys = e_dst * N + e_src
assert np.all(np.diff(keys) > 0)


def lookup(src, dst):
    ok = (src >= 0) & (dst >= 0) & (src != dst)
    k = np.where(ok, dst * N + src, 0)
    p = np.searchsorted(keys, k)
    p = np.minimum(p, E - 1)
    return np.where(ok & (keys[p] == k), p, -1)


posd = lookup(rs, rt)
posm = np.where(rdiag, -1, lookup(rt, rs))
used = (posd >= 0).sum() + (posm >= 0).sum()
assert used == E_dense, (used, E_dense)
print("rows with an edge in at least one direction: %.1f %%" % (100.0 * ((posd >= 0) | (posm >= 0)).mean()))
K = 0
etiles = (topo.max_edges + 15) // 16
epad = etiles * 16
i64 = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def row_scale(c, pos):
    sfull = ws.e_scale[(2 * K + c) * epad:(2 * K + c + 1) * epad]
    p = i64(pos)
    return torch.where(p >= 0, sfull[p.clamp(min=0)], torch.zeros((), device=dev)).contiguous()


sd1, sm1, sd2, sm2 = row_scale(0, posd), row_scale(0, posm), row_scale(1, posd), row_scale(1, posm)
# attributes in row order: canonical edge -> its row (and, in a diagonal tile, the row of the opposite direction)
c_pos, c_mir = ws.c_pos[:C].cpu().numpy().astype(np.int64), ws.c_mir[:C].cpu().numpy().astype(np.int64)
can_of_pos = np.full(E, -1, dtype=np.int64)
can_of_pos[c_pos] = np.arange(C)
can_of_pos[c_mir[c_mir >= 0]] = np.nonzero(c_mir >= 0)[0]
pos_any = np.where(posd >= 0, posd, posm)
rows = np.nonzero(pos_any >= 0)[0]
can = can_of_pos[pos_any[rows]]
order = np.argsort(can, kind="stable")
can_s, rows_s = can[order], rows[order]
firsts = np.ones(len(can_s), dtype=bool)
firsts[1:] = can_s[1:] != can_s[:-1]
row1 = np.full(C, -1, dtype=np.int64)
row2 = np.full(C, -1, dtype=np.int64)
row1[can_s[firsts]] = rows_s[firsts]
row2[can_s[~firsts]] = rows_s[~firsts]
assert (~firsts).sum() == len(np.unique(can_s[~firsts]))          # at most two rows per canonical edge
i32 = lambda x: torch.from_numpy(np.ascontiguousarray(x).astype(np.int32)).to(dev)
e_attr2 = torch.zeros(TL * 2048, dtype=torch.float32, device=dev)
row1_d, row2_d = i32(row1), i32(row2)          # (named: the launch is asynchronous, the buffers must outlive it)
rc = lib.agdiff_edge_encoder(P, _lib.ptr(ws.num_canon), etiles, _lib.ptr(ws.c_len), _lib.ptr(ws.c_type), _lib.ptr(e_attr2),
                             None, None, _lib.ptr(row1_d), _lib.ptr(row2_d), st)
assert rc == 0
assert lib.agdiff_schnet_node_stage(P, Tp, Wp, 0, st) == 0
torch.cuda.synchronize()

# check: the row-ordered attributes equal the product's (destination-sorted) ones
ua = ws.e_attr.view(-1, 2, 64, 4)          # [(tile*4+t)][u][slot][4]
ub = e_attr2.view(-1, 2, 64, 4)
rr = torch.from_numpy(rows[:4096]).to(dev)
pp = torch.from_numpy(pos_any[rows[:4096]]).to(dev)
mx = 0.0
for t in range(4):
    for qq in range(4):
        x1 = ua[(pp >> 4) * 4 + t, :, (pp & 15) * 4 + qq]
        x2 = ub[(rr >> 4) * 4 + t, :, (rr & 15) * 4 + qq]
        mx = max(mx, float((x1 - x2).abs().max()))
print("attr rows vs product attrs: max abs diff %.3e" % mx)
pt_atoms, pt_info = i32(tabs["pt_atoms"].reshape(-1)), i32(tabs["pt_info"].reshape(-1))
wave_ptr = i32(tabs["wave_tile_ptr"])
dbuf = torch.zeros(tabs["n_groups"] * 4 * 192, dtype=torch.float32, device=dev)
mbuf = torch.zeros(tabs["n_sets"] * 16 * 192, dtype=torch.float32, device=dev)
fn = proto.agdiff_proto_cfconv_pairs4
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int32] + [ctypes.c_void_p] * 11 + [ctypes.c_int32, ctypes.c_void_p]


def new():
    rc = fn(P, K, _lib.ptr(pt_atoms), _lib.ptr(pt_info), _lib.ptr(sd1), _lib.ptr(sm1), _lib.ptr(sd2), _lib.ptr(sm2),
            _lib.ptr(e_attr2), _lib.ptr(ws.xs), _lib.ptr(dbuf), _lib.ptr(mbuf), _lib.ptr(wave_ptr), num_waves, st)
    assert rc == 0, rc


def old():
    assert lib.agdiff_cfconv_fused(P, Tp, Wp, K, st) == 0


old(); new()
torch.cuda.synchronize()
# reference aggregate of the product kernel: agg + agg_first partials (as the node stage adds them)
ce = 16 * lib.agdiff_conv_chunk_tiles(ctypes.c_int64(topo.max_edges))
ip = ws.in_ptr.cpu().numpy().astype(np.int64)
ref = ws.agg.view(-1, 192)[:N].clone()
first = ws.agg_first.view(-1, 192)
lo, hi = ip[:-1], ip[1:]
has = hi > lo
clo, chi = lo // ce, np.where(has, (hi - 1) // ce, lo // ce)
ref[torch.from_numpy(~has).to(dev)] = 0
for i in np.nonzero(chi > clo)[0]:
    for c in range(clo[i] + 1, chi[i] + 1):
        ref[i] += first[c]
got = torch.zeros(N, 192, device=dev)
da, ma = i64(tabs["d_atom"]), i64(tabs["m_atom"])
got.index_add_(0, da[da >= 0], dbuf.view(-1, 192)[da >= 0])
got.index_add_(0, ma[ma >= 0], mbuf.view(-1, 192)[ma >= 0])
dense_atom = torch.from_numpy(is_dense_mol[np.searchsorted(gp, np.arange(N), side="right") - 1]).to(dev)
diff = (got - ref)[dense_atom].abs()
err = float(diff.max() / ref.abs().max())
if err > 1e-4:
    bad = torch.nonzero(diff.max(dim=1).values > 1e-4 * ref.abs().max()).flatten()
    print("atoms off: %d of %d; first %s; channels off of the first: %s" % (len(bad), int(dense_atom.sum()), bad[:12].tolist(),
          torch.nonzero(diff[bad[0]] > 1e-4 * ref.abs().max()).flatten().tolist()[:24]))
print("aggregate of the dense molecules: max|pairs4 - product| / max|product| = %.3e   (max|product| %.3e)"
      % (err, float(ref.abs().max())))


if err > 1e-4:
    # localise: product kernel with the scales of the mirror (resp. direct) positions zeroed vs the kernel's dbuf (mbuf)
    keep = ws.e_scale.clone()
    for name, pz, buf, atoms in (("direct", posm, dbuf, da), ("mirror", posd, mbuf, ma)):
        ws.e_scale.copy_(keep)
        pzt = i64(pz[pz >= 0])
        for c in (0, 1):
            ws.e_scale[(2 * K + c) * epad + pzt] = 0.0
        old()
        torch.cuda.synchronize()
        r2 = ws.agg.view(-1, 192)[:N].clone()
        r2[torch.from_numpy(~has).to(dev)] = 0
        for i in np.nonzero(chi > clo)[0]:
            for c in range(clo[i] + 1, chi[i] + 1):
                r2[i] += first[c]
        g2 = torch.zeros(N, 192, device=dev)
        g2.index_add_(0, atoms[atoms >= 0], buf.view(-1, 192)[atoms >= 0])
        d2 = (g2 - r2)[dense_atom].abs()
        print(name, "part: max err %.3e; per-channel-tile max err" % float(d2.max() / r2.abs().max()),
              [round(float(d2[:, 16 * c:16 * c + 16].max() / r2.abs().max()), 4) for c in range(12)])
        print("   sample got", g2[dense_atom][5, :4].tolist(), "ref", r2[dense_atom][5, :4].tolist())
    ws.e_scale.copy_(keep)


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


# the product kernel restricted to the edges of the remaining (larger) molecules: they are a suffix / subset of the list;
# estimate its time as (their share of the tiles) x (the full launch)
share_sparse = 1.0 - E_dense / E
for rep in range(3):
    to = timeit(old, args.reps)
    tn = timeit(new, args.reps)
    print("product k_cfconv_fused (all %d tiles) %.4f ms   pairs4 kernel (%d tiles) %.4f ms   + remaining molecules ~%.4f ms  => %.4f ms (%.1f %%)"
          % ((E + 15) // 16, to, TL, tn, share_sparse * to, tn + share_sparse * to, 100.0 * (tn + share_sparse * to) / to))
print(json.dumps({"workload": args.workload, "E": E, "C": C, "E_dense": E_dense, "pair_tiles": TL, "old_ms": to, "pairs4_ms": tn,
                  "rel_err": err, "max_atoms": args.max_atoms}))
