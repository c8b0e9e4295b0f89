"""Host-side static topology of one packed batch and the device workspace for it.

The positions change every denoising step, the bond / 2-hop / 3-hop ("local", type > 0) edges never
do (dualenc.py:566-567; utils/transforms.py:12-71).  Everything position-independent is built once
per batch here with numpy and uploaded: graph offsets, the coalesced local edge list in the
reference's (src, dst) order (models/common.py:215-231), its CSR views, and exact capacity bounds
for the per-step radius graph.
"""
import numpy as np

from . import _lib
from .synth import extend_graph_order_np


def _to_np(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x)


def batch_fingerprint(atom_type, bond_index, bond_type, batch, num_graphs, extend_order):
    """What a prepared BatchTopology is checked against when a sampler is handed one (epsnet._batch): atom count, graph count,
    extend_order and two order-sensitive 64-bit sums over (atom_type, batch) and the bond list AS PASSED (wrapping int64
    arithmetic: the same number from numpy arrays and from torch tensors on any device -- one small reduction, no copy of
    the lists)."""
    def wsum(cols, mults):
        if hasattr(cols[0], "detach"):                       # torch (possibly on the GPU)
            import torch
            n = int(cols[0].reshape(-1).shape[0])
            if n == 0:
                return 0
            acc = torch.zeros(n, dtype=torch.int64, device=cols[0].device)
            for c, m in zip(cols, mults):
                acc += c.reshape(-1).to(torch.int64) * m
            acc = (acc + 12345) * torch.arange(1, n + 1, dtype=torch.int64, device=acc.device)
            return int(acc.sum().item())
        cols = [np.asarray(c).reshape(-1).astype(np.int64) for c in cols]
        n = cols[0].shape[0]
        if n == 0:
            return 0
        acc = np.zeros(n, dtype=np.int64)
        with np.errstate(over="ignore"):
            for c, m in zip(cols, mults):
                acc += c * np.int64(m)
            acc = (acc + np.int64(12345)) * np.arange(1, n + 1, dtype=np.int64)
            return int(acc.sum(dtype=np.int64))
    bi = bond_index.reshape(2, -1)
    G = None if num_graphs is None else int(num_graphs)
    return (int(atom_type.shape[0]), G, bool(extend_order), wsum([atom_type, batch], [1000003, 998244353]),
            wsum([bi[0], bi[1], bond_type], [1000003, 998244353, 7919]))


_GROUP_ORDER_CACHE = {}


def _group_order(need, GT):
    """agdiff_group_order (include/agdiff_hip.h; a host function of the library): the order of a molecule's atoms (local
    indices) in which consecutive runs of GT atoms form the groups agdiff_cfconv_node's waves own -- atoms with like per-type
    tile needs together.  On the bench job: 6 % fewer local tiles than sorting by the need vectors alone.  Cached per need
    matrix: the conformers of a molecule, and a molecule met again, cost one look-up."""
    need = np.ascontiguousarray(need, dtype=np.int32)
    key = (GT, need.shape, need.tobytes())
    hit = _GROUP_ORDER_CACHE.get(key)
    if hit is None:
        import ctypes
        hit = np.zeros(need.shape[0], dtype=np.int32)
        _lib.check(_lib.load().agdiff_group_order(need.ctypes.data_as(ctypes.c_void_p), need.shape[0], need.shape[1], GT,
                                                  hit.ctypes.data_as(ctypes.c_void_p)), "agdiff_group_order")
        hit = hit.astype(np.int64)
        if len(_GROUP_ORDER_CACHE) > 4096:
            _GROUP_ORDER_CACHE.clear()
        _GROUP_ORDER_CACHE[key] = hit
    return hit


class BatchTopology:
    # agdiff_cfconv_node gives a wave a GROUP of targets (quad_tgt): four for batches that fill the chip that way (one
    # 16-row local tile then serves four targets: fewest tiles), two or one for small batches, where there are more wave
    # slots (3,072 on an MI355X) than groups and the wave's chain of tiles is what a launch lasts
    GROUP_MIN_NODES = ((6144, 4), (3072, 2))        # N > 6144: 4 targets per group, N > 3072: 2, else 1

    def __init__(self, atom_type, bond_index, bond_type, batch, num_graphs=None, extend_order=False,
                 order=3, device="cuda", group_targets=None, radius_column=True):
        import torch
        at = _to_np(atom_type).astype(np.int64)
        bi = _to_np(bond_index).astype(np.int64).reshape(2, -1)
        bt = _to_np(bond_type).astype(np.int64).reshape(-1)
        ba = _to_np(batch).astype(np.int64)
        self.fingerprint = batch_fingerprint(at, bi, bt, ba, num_graphs, extend_order)
        N = at.shape[0]
        if ba.shape[0] != N:
            raise ValueError("batch and atom_type disagree on the number of nodes")
        if N == 0:
            raise ValueError("empty batch")
        if np.any(np.diff(ba) < 0):
            raise ValueError("batch must be sorted (graph-contiguous nodes, PyG Batch.from_data_list)")
        G = int(ba[-1]) + 1 if num_graphs is None else int(num_graphs)
        counts = np.bincount(ba, minlength=G)
        gptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        if at.min() < 0 or at.max() >= 100:
            raise ValueError("atom_type out of the embedding range [0, 100)")
        if bi.size and (bi.min() < 0 or bi.max() >= N):
            raise ValueError("bond_index out of range")
        if bi.size and np.any(ba[bi[0]] != ba[bi[1]]):
            raise ValueError("bond between different graphs")

        if extend_order:
            # _extend_graph_order (common.py:135-205), per graph instead of one dense N x N matrix
            rs, cs, ts = [], [], []
            order_src = np.argsort(ba[bi[0]], kind="stable") if bi.size else np.zeros(0, dtype=np.int64)
            gb = np.searchsorted(ba[bi[0]][order_src], np.arange(G + 1)) if bi.size else np.zeros(G + 1, dtype=np.int64)
            for g in range(G):
                sel = order_src[gb[g]:gb[g + 1]]
                n = int(counts[g])
                if n == 0:
                    continue
                r, c, t = extend_graph_order_np(n, bi[0][sel] - gptr[g], bi[1][sel] - gptr[g], bt[sel], order=order)
                rs.append(r + gptr[g]); cs.append(c + gptr[g]); ts.append(t)
            bi = np.stack([np.concatenate(rs), np.concatenate(cs)]) if rs else np.zeros((2, 0), dtype=np.int64)
            bt = np.concatenate(ts) if ts else np.zeros(0, dtype=np.int64)

        # coalesce: sort by (row, col), sum duplicate values (common.py:215,226)
        key = bi[0] * N + bi[1]
        self.bond_list_coalesced = bool(np.all(np.diff(key) > 0))     # already (row, col)-sorted and unique
        uniq, inv = np.unique(key, return_inverse=True)
        self.loc_pos_of_input = inv.reshape(-1)                        # input bond edge -> index in the coalesced list
        typ = np.zeros(uniq.shape[0], dtype=np.int64)
        np.add.at(typ, inv, bt)
        src, dst = uniq // N, uniq % N
        if np.any(typ <= 0) or np.any(typ >= 100):
            raise NotImplementedError("bond/edge types must lie in 1..99 after coalescing (type 0 marks radius edges)")
        if np.any(src == dst):
            raise NotImplementedError("self-loop bond edges are not supported")
        L = int(uniq.shape[0])
        out_ptr = np.concatenate([[0], np.cumsum(np.bincount(src, minlength=N))])
        in_order = np.argsort(dst, kind="stable")               # grouped by dst, src ascending
        in_ptr = np.concatenate([[0], np.cumsum(np.bincount(dst, minlength=N))])
        locdeg = np.diff(in_ptr)
        # canonical local edges (agdiff_topo_t.lc_*): one of j -> i / i -> j when both exist with the same type
        # (uniq is sorted, so the mirror of an edge is found by binary search on its swapped key)
        mkey = dst * N + src
        mpos = np.searchsorted(uniq, mkey)
        mpos_c = np.minimum(mpos, max(L - 1, 0))
        has_m = (mpos < L) & (uniq[mpos_c] == mkey) & (typ[mpos_c] == typ) if L else np.zeros(0, dtype=bool)
        canon = ~has_m | (src < dst)
        lc_pos = np.nonzero(canon)[0]
        # ... in order of (molecule, type, source): a 16-entry tile of the list then mostly holds ONE type (agdiff_local_edge_rows
        # runs one masked MFMA round per type present in a tile), and a molecule's entries stay contiguous (lcm_ptr)
        lc_pos = lc_pos[np.lexsort((src[lc_pos], typ[lc_pos], ba[src[lc_pos]]))]
        lc_mir = np.where(has_m[lc_pos], mpos_c[lc_pos], -1)
        loc_row = np.full(L, -1, dtype=np.int64)          # canonical index (row of l_attr_rows) of every local edge
        loc_row[lc_pos] = np.arange(lc_pos.shape[0])
        loc_row[lc_mir[lc_mir >= 0]] = np.nonzero(lc_mir >= 0)[0]
        assert L == 0 or loc_row.min() >= 0

        n_of_node = counts[ba]
        cap = np.minimum(n_of_node - 1, np.minimum(_lib.RADIUS_CAP, n_of_node - 1) + locdeg)
        cap = np.maximum(cap, 0)
        self.N, self.G, self.L = N, G, L
        self.max_edges = int(cap.sum())
        self.max_in_degree = int(cap.max()) if N else 0
        self.max_atoms = int(counts.max())
        if self.max_atoms > _lib.MAX_ATOMS_PER_GRAPH:
            raise NotImplementedError("graphs with more than %d atoms are not supported" % _lib.MAX_ATOMS_PER_GRAPH)
        if N * 192 * 4 >= 2 ** 32 or self.max_edges >= 2 ** 31 - 64:
            raise NotImplementedError("batch too large for 32-bit offsets; split it")

        i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a).astype(np.int32)).to(device)
        self.device = device
        self.graph_ptr = i32(gptr)
        self.atom_type = i32(at)
        self.loc_src, self.loc_dst, self.loc_type = i32(src), i32(dst), i32(typ)
        self.loc_out_ptr, self.loc_in_ptr, self.loc_in_eid = i32(out_ptr), i32(in_ptr), i32(in_order)
        self.Lc = int(lc_pos.shape[0])
        self.lc_src, self.lc_dst, self.lc_type = i32(src[lc_pos]), i32(dst[lc_pos]), i32(typ[lc_pos])
        self.lc_pos, self.lc_mir, self.loc_row = i32(lc_pos), i32(lc_mir), i32(loc_row)
        self.loc_in_src, self.loc_in_row = i32(src[in_order]), i32(loc_row[in_order])
        # the local list as a destination-sorted edge list of its own for the split CFConv, every target's list padded to
        # at least 8 entries (agdiff_topo_t.lp_*: a 16-edge tile then holds at most three targets, the middle one whole)
        pdeg = np.where(locdeg > 0, np.maximum(locdeg, 8), 0)
        lp_ptr = np.concatenate([[0], np.cumsum(pdeg)])
        Lp = int(lp_ptr[-1])
        tgt = np.repeat(np.arange(N), pdeg)                               # target of every padded entry
        k_in = np.arange(Lp) - lp_ptr[tgt]                                # its index inside the target's list
        real = k_in < locdeg[tgt]
        slot = np.minimum(in_ptr[tgt] + k_in, max(L - 1, 0))              # in-slot of a real entry
        first = in_order[np.minimum(in_ptr[tgt], max(L - 1, 0))] if L else np.zeros(Lp, dtype=np.int64)
        eid = np.where(real, in_order[slot], first) if L else np.zeros(0, dtype=np.int64)
        ppos = np.empty(L, dtype=np.int64)                                 # local edge id -> padded position
        ppos[eid[real]] = np.nonzero(real)[0]
        self.Lp = Lp
        self.lp_ptr = i32(lp_ptr)
        self.lp_src = i32(np.where(real, src[eid], tgt) if L else tgt)
        self.lp_dst = i32(tgt)
        self.lp_type = i32(typ[eid] if L else np.zeros(Lp))
        self.lp_row = i32(np.where(real, loc_row[eid], -1) if L else np.zeros(Lp))
        self.lc_ppos = i32(ppos[lc_pos])
        self.lc_pmir = i32(np.where(lc_mir >= 0, ppos[np.maximum(lc_mir, 0)], -1))
        # ... and as QUAD TILES for agdiff_cfconv_node (agdiff_topo_t.quad_tgt, lt_*): one wave owns GT = 4 (2, 1) targets of one
        # molecule, and a 16-row local tile holds rows of ONE edge type: rows RT k .. RT k + RT - 1 (RT = 16 / GT) are in-edges of
        # the group's k-th target (so the four rows a lane quarter holds belong to one target and the tile needs no masks and
        # ONE filter set).  Per group and type: max over its targets of ceil(in-edges of that type / RT) tiles; atoms are
        # grouped by those needs so that a group's targets need like tiles.
        ltypes = np.unique(typ) if L else np.zeros(0, dtype=np.int64)
        cnt_tt = np.zeros((N, max(ltypes.size, 1)), dtype=np.int64)
        for k, ty in enumerate(ltypes):
            cnt_tt[:, k] = np.bincount(dst[typ == ty], minlength=N)
        if group_targets is None:
            group_targets = next((g for n, g in self.GROUP_MIN_NODES if N > n), 1)
        if group_targets not in (1, 2, 4):
            raise ValueError("group_targets must be 1, 2 or 4")
        GT, RT = int(group_targets), 16 // int(group_targets)               # targets per group, rows per target in a tile
        need = (cnt_tt + RT - 1) // RT
        # On quads the radius rows go in quad tiles too (csrc/nodeconv.hip k_cfconv_quad: a quad walks max over its targets of
        # ceil(radius rows / 4) tiles), so the grouping also counts a radius column: the rows a target has while its molecule
        # lies inside the cutoff -- radius_graph keeps the first 33 candidates in index order, self included and then dropped
        # (models/common.py:217; AGDIFF_RADIUS_CAP), minus those that are local edges.  Only an ordering heuristic: the kernels
        # take the row counts of the graph that was built.
        need_order = need
        rad_est = None
        if GT == 4:
            li = np.arange(N) - gptr[ba]
            m = np.minimum(n_of_node, _lib.RADIUS_CAP)
            cand = np.where(li < m, m - 1, m)
            loc_in_cand = np.bincount(dst[(src - gptr[ba[src]]) < m[dst]], minlength=N) if L else np.zeros(N, dtype=np.int64)
            rad_est = np.maximum(cand - loc_in_cand, 0)
            if radius_column:
                need_order = np.concatenate([need, ((rad_est + 3) // 4)[:, None]], axis=1)
        # Consecutive graphs with the same need matrix -- the conformers of one molecule -- share one grouping: one look-up per
        # run of such graphs, their quads written at once (a 200 k-atom batch has ~5,000 graphs of ~10 molecules).
        quad_tgt = []
        size_change = np.flatnonzero(np.diff(counts)) + 1
        for ga, gb in zip(np.concatenate([[0], size_change]), np.concatenate([size_change, [G]])):
            n = int(counts[ga])
            if n == 0:
                continue
            block = need_order[gptr[ga]:gptr[gb]].reshape(gb - ga, n, need_order.shape[1])
            starts = np.concatenate([[0], np.flatnonzero(~(block[1:] == block[:-1]).all(axis=(1, 2))) + 1, [gb - ga]])
            for sa, sb in zip(starts[:-1], starts[1:]):
                idx = _group_order(block[sa], GT)
                if idx.size % GT:
                    idx = np.concatenate([idx, np.full(GT - idx.size % GT, -1, dtype=idx.dtype)])
                grp = np.full((idx.size // GT, 4), -1, dtype=np.int64)          # (always four entries per group: -1 = none)
                grp[:, :GT] = idx.reshape(-1, GT)
                base = gptr[ga + sa:ga + sb].astype(np.int64)[:, None, None]
                quad_tgt.append(np.where(grp[None] >= 0, grp[None] + base, -1).reshape(-1))
        quad_tgt = np.concatenate(quad_tgt).astype(np.int64) if quad_tgt else np.zeros(0, dtype=np.int64)
        Q = quad_tgt.size // 4
        qt = quad_tgt.reshape(Q, 4)
        qneed = np.where(qt[:, :, None] >= 0, need[np.maximum(qt, 0)], 0).max(axis=1)        # [Q, types]: tiles of each type
        nt_quad = qneed.sum(axis=1)
        lt_ptr = np.concatenate([[0], np.cumsum(nt_quad)])
        T = int(lt_ptr[-1])
        # agdiff_topo_t.quad_wg_ptr: the quads cut into 256 contiguous ranges of like TILE cost (local tiles + the radius tiles a
        # quad has while its molecule lies inside the cutoff) for k_cfconv_quad's 256 persistent workgroups: with equal QUAD counts
        # the busiest workgroup of a default-job batch walked 4..12 % more tiles than the average one
        if GT == 4 and Q >= 256:
            rq = np.where(qt >= 0, rad_est[np.maximum(qt, 0)], 0).max(axis=1)
            # (a local tile takes about 1.2 x the time of a radius tile: tools/quad_stamps.py)
            cost = np.cumsum(6 * nt_quad + 5 * ((rq + 3) // 4))
            cuts = np.searchsorted(cost, cost[-1] * np.arange(1, 256) / 256.0, side="left") + 1
            wg_ptr = np.concatenate([[0], np.minimum(cuts, Q), [Q]])
            wg_ptr = np.maximum.accumulate(wg_ptr)
        else:
            wg_ptr = np.zeros(0, dtype=np.int64)
        # a target's in-edges of one type in order of source: position of every local edge inside its (target, type) list
        by_tts = np.lexsort((src, typ, dst)) if L else np.zeros(0, dtype=np.int64)
        rank_in = np.zeros(L, dtype=np.int64)
        if L:
            key = dst[by_tts] * 256 + typ[by_tts]
            start = np.concatenate([[True], key[1:] != key[:-1]])
            first_of = np.maximum.accumulate(np.where(start, np.arange(L), 0))
            rank_in[by_tts] = np.arange(L) - first_of
        # tiles of a quad: type ascending, then the tile index u inside the type
        tile_quad = np.repeat(np.arange(Q), nt_quad)
        tile_type_k = np.zeros(T, dtype=np.int64)
        tile_u = np.zeros(T, dtype=np.int64)
        if T:
            flat = qneed.reshape(-1)                                          # (quad, type) -> tiles
            kq = np.repeat(np.arange(flat.size), flat)                        # (quad, type) of every tile, in tile order
            tile_type_k = kq % qneed.shape[1]
            tile_u = np.arange(T) - np.concatenate([[0], np.cumsum(flat)])[kq]
        tile_of_qk = np.concatenate([[0], np.cumsum(qneed.reshape(-1))])     # first tile of (quad, type)
        quad_of = np.zeros(N, dtype=np.int64)
        slot_of = np.zeros(N, dtype=np.int64)
        valid = quad_tgt >= 0
        quad_of[quad_tgt[valid]] = np.nonzero(valid)[0] // 4
        slot_of[quad_tgt[valid]] = np.nonzero(valid)[0] % 4
        tpos = np.zeros(L, dtype=np.int64)
        if L:
            k_of = np.searchsorted(ltypes, typ)
            tile_e = tile_of_qk[quad_of[dst] * qneed.shape[1] + k_of] + rank_in // RT
            tpos = tile_e * 16 + slot_of[dst] * RT + rank_in % RT
        real_t = np.zeros(16 * T, dtype=bool)
        eid_t = np.full(16 * T, -1, dtype=np.int64)
        real_t[tpos] = True
        eid_t[tpos] = np.arange(L)
        assert int(real_t.sum()) == L                                        # every local edge has its own row
        trow = np.arange(16 * T)
        tgt_t = qt[tile_quad[trow // 16], (trow % 16) // RT] if T else np.zeros(0, dtype=np.int64)
        first_t = qt[tile_quad[trow // 16], 0] if T else tgt_t
        tgt_c = np.where(tgt_t >= 0, tgt_t, first_t)                         # pad rows: src = the target itself (a missing one: the first)
        self.T, self.Q, self.group_targets = T, int(Q), GT
        self.quad_tgt = i32(quad_tgt)
        self.lt_ptr = i32(lt_ptr)
        self.lt_src = i32(np.where(real_t, src[np.maximum(eid_t, 0)], tgt_c) if L else np.zeros(16 * T))
        self.lt_type = i32(ltypes[tile_type_k][trow // 16] if T else np.zeros(0))     # every row of a tile carries the tile's type
        self.lt_real = real_t
        self.quad_wg_ptr = i32(wg_ptr) if wg_ptr.size else None
        # static local in-adjacency masks (agdiff_topo_t.loc_bits): bit (src - first atom of the molecule) of row dst
        W = 2 * ((self.max_atoms + 63) // 64)
        bits = np.zeros(N * W, dtype=np.uint32)
        if L:
            jl = src - gptr[ba[src]]
            # (the local edges are unique, so every (row, word) sums DISTINCT powers of two: exact in float64, and a bincount)
            bits = np.bincount(dst * W + (jl >> 5), weights=np.ldexp(1.0, (jl & 31).astype(np.int32)), minlength=N * W).astype(np.uint32)
        self.loc_bits = torch.from_numpy(bits.view(np.int32)).to(device)
        self.lt_eid = eid_t
        self.lc_tpos = i32(tpos[lc_pos])
        self.lc_tmir = i32(np.where(lc_mir >= 0, tpos[np.maximum(lc_mir, 0)], -1))
        self.lcm_ptr = i32(np.searchsorted(ba[src[lc_pos]], np.arange(G + 1)) if L else np.zeros(G + 1))
        self.local_types = np.unique(typ)                 # PackedParams.ensure_local_types (per-type filter polynomials)
        # int64 copies of the local edges for the API results (forward() returns int64 indices)
        self.loc_index64 = torch.from_numpy(np.stack([src, dst])).to(device)
        self.loc_type64 = torch.from_numpy(typ).to(device)
        self.batch64 = torch.from_numpy(ba).to(device)
        self.graph_sizes = counts

        t = _lib.Topo()
        t.num_nodes, t.num_graphs, t.num_local = N, G, L
        t.max_edges, t.max_atoms_per_graph, t.max_in_degree = self.max_edges, self.max_atoms, self.max_in_degree
        t.num_local_canon = self.Lc
        t.num_local_padded = self.Lp
        t.num_local_tiles = self.T
        t.num_quads = self.Q
        t.group_targets = self.group_targets
        tm = [0, 0]
        for ty in self.local_types:
            tm[int(ty) >> 6] |= 1 << (int(ty) & 63)
        for w in (0, 1):
            t.local_type_mask[w] = tm[w] - (1 << 64) if tm[w] >= (1 << 63) else tm[w]
        self.struct = t
        self._set_pointers()

    _POINTER_FIELDS = ("graph_ptr", "atom_type", "loc_src", "loc_dst", "loc_type", "loc_out_ptr", "loc_in_ptr", "loc_in_eid",
                       "lc_src", "lc_dst", "lc_type", "lc_pos", "lc_mir", "loc_row", "loc_in_src",
                       "loc_in_row", "lp_ptr", "lp_src", "lp_dst", "lp_type", "lp_row", "lc_ppos", "lc_pmir",
                       "quad_tgt", "lt_ptr", "lt_src", "lt_type", "lc_tpos", "lc_tmir", "lcm_ptr", "loc_bits")

    def _set_pointers(self):
        for f in self._POINTER_FIELDS:
            setattr(self.struct, f, _lib.ptr(getattr(self, f)))
        self.struct.quad_wg_ptr = _lib.ptr(self.quad_wg_ptr) if self.quad_wg_ptr is not None else None

    # A topology prepared in a worker PROCESS (agdiff_amd/driver.py) comes back pickled: the ctypes struct (raw addresses) stays
    # behind; its scalar fields travel as a dict and the pointers are taken again from the tensors on arrival.
    def __getstate__(self):
        import ctypes
        st = dict(self.__dict__)
        t = st.pop("struct")
        scalars = {}
        for name, ctype in _lib.Topo._fields_:
            if ctype is ctypes.c_void_p:
                continue
            v = getattr(t, name)
            scalars[name] = list(v) if hasattr(v, "__len__") else v
        st["_struct_scalars"] = scalars
        return st

    def __setstate__(self, st):
        scalars = st.pop("_struct_scalars")
        self.__dict__.update(st)
        t = _lib.Topo()
        for name, v in scalars.items():
            if isinstance(v, list):
                for k, x in enumerate(v):
                    getattr(t, name)[k] = x
            else:
                setattr(t, name, v)
        self.struct = t
        self._set_pointers()

    def to(self, device):
        """Move the index arrays to `device` (in place; returns self).  A topology is host work only -- numpy sorts and the
        quad grouping -- so the driver builds the next batch's on the CPU in a background thread while the GPU samples the
        current one (agdiff_amd/driver.py) and moves it over when its turn comes."""
        import torch
        device = torch.device(device)
        if torch.device(self.device) == device:
            return self
        for name, v in list(vars(self).items()):
            if isinstance(v, torch.Tensor):
                setattr(self, name, v.to(device))
        self.device = device
        self._set_pointers()
        return self


class Workspace:
    """Device buffers for one BatchTopology (include/agdiff_hip.h: agdiff_ws_t)."""

    def __init__(self, topo, max_edges=None):
        import torch
        dev = topo.device
        if max_edges is not None:          # caller-supplied graph (forward(edge_index=...)): capacity = its size
            topo.max_edges = int(max_edges)
            topo.struct.max_edges = int(max_edges)
        N, G, L, E = topo.N, topo.G, topo.L, topo.max_edges
        TW = _lib.TILE
        etiles = (E + TW - 1) // TW
        ltiles = (L + TW - 1) // TW
        chunk_tiles = _lib.load().agdiff_conv_chunk_tiles(E)
        chunks = (etiles + chunk_tiles - 1) // chunk_tiles
        i32 = lambda n: torch.zeros(max(int(n), 1), dtype=torch.int32, device=dev)
        f32 = lambda n: torch.zeros(max(int(n), 1), dtype=torch.float32, device=dev)
        self.num_edges = i32(1)
        self.num_local = torch.tensor([L], dtype=torch.int32, device=dev)
        self.graph_edge_cnt, self.graph_edge_ptr = i32(G), i32(G + 1)
        self.in_ptr, self.out_ptr = i32(N + 1), i32(N + 1)
        self.e_src, self.e_dst, self.e_type, self.ref2dst = i32(etiles * TW), i32(etiles * TW), i32(etiles * TW), i32(etiles * TW)
        self.e_loc = i32(etiles * TW)
        self.num_canon = i32(1)
        self.graph_canon_cnt, self.graph_canon_ptr = i32(G), i32(G + 1)
        cn = etiles * TW
        self.c_len = f32(cn)
        self.c_type, self.c_src, self.c_dst = i32(cn), i32(cn), i32(cn)
        self.c_pos, self.c_mir = i32(cn), i32(cn)
        self.e_len = f32(etiles * TW)
        self.e_attr = f32(etiles * TW * 128)
        self.e_inv_global = f32(etiles * TW)
        self.e_scale = f32(2 * _lib.DEFINES["AGDIFF_MAX_CONVS"] * etiles * TW)
        self.l_len, self.l_inv = f32(ltiles * TW), f32(ltiles * TW)
        self.lc_len = f32(ltiles * TW)
        self.num_local_canon = torch.tensor([topo.Lc], dtype=torch.int32, device=dev)
        self.l_attr_rows = f32(ltiles * TW * 128)
        self.h, self.xs, self.agg = f32(N * 128), f32(N * 192), f32(N * 192)
        self.agg_first = f32(chunks * 192)
        self.hl, self.hl2 = f32(N * 128), f32(N * 128)
        self.nan_flag = i32(1 + G)
        self.range_rows = i32(N)           # agdiff_ws_t.range_rows: per-node flags of the split-fp16 kernels' hidden activations
        self.scratch = f32(N * 3)
        # CFConv by filter polynomials: radius rows by target (agdiff_ws_t.rad_*), local quad tiles (lt_*), padded local list
        RS = _lib.DEFINES["AGDIFF_RAD_STRIDE"]
        Lp = topo.Lp
        ptiles = (Lp + TW - 1) // TW
        lchunk = _lib.load().agdiff_conv_chunk_tiles(Lp)
        self.rad_cnt = i32(N)
        # every row of target i starts out naming i itself as its source: k_cfconv_quad runs the rows between a target's count and
        # its quad's tile end with scale 0 but still gathers x[src], and agdiff_sampler_front does not write them -- a row that was
        # never written must not point at an atom of ANOTHER molecule (0 x a non-finite x of a diverging molecule is NaN)
        self.rad_src = torch.arange(N, dtype=torch.int32, device=dev).repeat_interleave(RS)
        self.rad_len = f32(N * RS)
        self.r_scale = f32(2 * _lib.DEFINES["AGDIFF_MAX_CONVS"] * N * RS)
        self.lt_len = f32(TW * topo.T)
        self.lt_scale = f32(2 * _lib.DEFINES["AGDIFF_MAX_CONVS"] * TW * topo.T)
        self.inv_r = f32(N * RS)
        self.canon_counter = i32(2)
        self.variant_log = torch.zeros(1, dtype=torch.int64)          # host word (include/agdiff_hip.h: AGDIFF_VAR_*)
        self.num_local_padded = torch.tensor([Lp], dtype=torch.int32, device=dev)
        self.l_scale = f32(2 * _lib.DEFINES["AGDIFF_MAX_CONVS"] * ptiles * TW)
        self.l_attr_frag = f32(ptiles * TW * 128)
        self.l_len_p = f32(ptiles * TW)
        self.h0, self.xs0 = f32(N * 128), f32(N * 192)
        self.enc_flags = i32(1 + (topo.Lc + TW - 1) // TW)
        self.g_inbits = i32(N * 2 * ((topo.max_atoms + 63) // 64))
        self.g_deg, self.g_cdeg = i32(N), i32(N)
        self.agg_loc = f32(N * 192)
        self.agg_first_loc = f32((ptiles + lchunk - 1) // lchunk * 192)
        w = _lib.Workspace()
        for f, _ in _lib.Workspace._fields_:
            setattr(w, f, _lib.ptr(getattr(self, f)))
        self.struct = w
        self.topo = topo

    def bytes(self):
        import torch
        return sum(v.numel() * v.element_size() for v in vars(self).values() if isinstance(v, torch.Tensor))
