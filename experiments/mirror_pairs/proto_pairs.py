"""Host-side (numpy) builder of the PAIR-SWEEP order for the mirror-sharing CFConv prototype.

Input: the canonical edge list the graph build emits (one entry per mirror pair + every unpaired edge; destination
sorted).  Output: the same entries re-ordered by (molecule, source block b = local source index // 16, destination,
source) and cut into work ITEMS of at most `max_tiles` 16-row tiles at destination-group boundaries, each item padded
to whole tiles:
  row arrays [R]   p_src, p_dst (global atom ids), p_slot (source slot 0..15 in its block, -1: no mirror / padding),
                   p_can (index into the canonical list, -1: padding), p_seg (segment id)
  segments   [S]   one per (item, destination) group: seg_ptr[S+1] first row, seg_dst[S] destination atom
  items      [I]   item_row0 (multiple of 16), item_tiles, item_j0 (global atom id of the block's first source)
The prototype kernel writes agg_seg[S][192] (direct sums per segment) and mir_rows[I][16][192] (mirror sums per item
and source slot); agg[i] = sum of its segments + sum over the items of its block of mir_rows[item][slot(i)].
"""
import numpy as np


def build_pair_sweeps(c_src, c_dst, c_mir, graph_ptr, max_tiles=8, tile=16):
    c_src, c_dst, c_mir = [np.asarray(x, dtype=np.int64) for x in (c_src, c_dst, c_mir)]
    gp = np.asarray(graph_ptr, dtype=np.int64)
    C = c_src.shape[0]
    g = np.searchsorted(gp, c_dst, side="right") - 1
    blk = (c_src - gp[g]) // tile
    order = np.lexsort((c_src, c_dst, blk, g))
    g_o, b_o, d_o, s_o = g[order], blk[order], c_dst[order], c_src[order]
    # groups = runs of equal (g, b, dst); sweeps = runs of equal (g, b)
    new_grp = np.ones(C, dtype=bool)
    new_grp[1:] = (g_o[1:] != g_o[:-1]) | (b_o[1:] != b_o[:-1]) | (d_o[1:] != d_o[:-1])
    new_swp = np.ones(C, dtype=bool)
    new_swp[1:] = (g_o[1:] != g_o[:-1]) | (b_o[1:] != b_o[:-1])
    grp_start = np.nonzero(new_grp)[0]
    grp_len = np.diff(np.concatenate([grp_start, [C]]))
    swp_first_grp = np.nonzero(new_swp[grp_start])[0]
    swp_end_grp = np.concatenate([swp_first_grp[1:], [grp_start.shape[0]]])
    p_src, p_dst, p_slot, p_can, p_seg = [], [], [], [], []
    seg_ptr, seg_dst, item_row0, item_tiles, item_j0 = [], [], [], [], []
    rows = 0
    for ga, gb in zip(swp_first_grp, swp_end_grp):
        # best-fit decreasing of the sweep's groups (each <= 16 rows) into 16-row tiles: whole groups per tile, so a
        # tile's direct sums are complete when the tile ends (no carry between tiles)
        by_len = [[] for _ in range(tile + 1)]
        for k in range(ga, gb):
            by_len[grp_len[k]].append(k)
        left = gb - ga
        tiles = []
        while left:
            room, members = tile, []
            while room:
                L = room
                while L > 0 and not by_len[L]:
                    L -= 1
                if L == 0:
                    break
                members.append(by_len[L].pop())
                room -= L
                left -= 1
            tiles.append(members)
        a0 = grp_start[ga]
        j0 = gp[g_o[a0]] + b_o[a0] * tile
        for ti, members in enumerate(tiles):
            if ti % max_tiles == 0:
                item_row0.append(rows)
                item_j0.append(j0)
                item_tiles.append(min(max_tiles, len(tiles) - ti))
            used = 0
            for k in members:
                a, n = grp_start[k], grp_len[k]
                sl = slice(a, a + n)
                seg_ptr.append(rows + used)
                seg_dst.append(d_o[a])
                p_src.append(s_o[sl]); p_dst.append(d_o[sl])
                p_slot.append(np.where(c_mir[order[sl]] >= 0, s_o[sl] - j0, -1))
                p_can.append(order[sl]); p_seg.append(np.full(n, len(seg_dst) - 1))
                used += n
            pad = tile - used
            if pad:
                p_src.append(np.full(pad, p_src[-1][-1])); p_dst.append(np.full(pad, p_dst[-1][-1]))
                p_slot.append(np.full(pad, -1)); p_can.append(np.full(pad, -1)); p_seg.append(np.full(pad, len(seg_dst) - 1))
            rows += tile
    seg_ptr.append(rows)
    cat = lambda x: np.concatenate(x) if x else np.zeros(0, dtype=np.int64)
    return dict(p_src=cat(p_src), p_dst=cat(p_dst), p_slot=cat(p_slot), p_can=cat(p_can), p_seg=cat(p_seg),
                seg_ptr=np.asarray(seg_ptr), seg_dst=np.asarray(seg_dst), item_row0=np.asarray(item_row0),
                item_tiles=np.asarray(item_tiles), item_j0=np.asarray(item_j0), rows=rows)


def wave_partition(item_tiles, num_waves):
    """Contiguous item ranges per wave with balanced tile counts: wave w takes items [ptr[w], ptr[w+1])."""
    cum = np.concatenate([[0], np.cumsum(item_tiles)])
    tgt = cum[-1] * np.arange(num_waves + 1) / num_waves
    ptr = np.searchsorted(cum, tgt, side="left")
    ptr[0], ptr[-1] = 0, len(item_tiles)
    return np.maximum.accumulate(ptr)


if __name__ == "__main__":
    # self-check on a synthetic dense molecule pair list
    rng = np.random.default_rng(0)
    gp = np.array([0, 44, 62])
    src, dst, mir = [], [], []
    for g0, g1 in zip(gp[:-1], gp[1:]):
        for i in range(g0, g1):
            for j in range(g0, i):
                if rng.random() < 0.95:
                    src.append(j); dst.append(i); mir.append(0 if rng.random() < 0.8 else -1)
    r = build_pair_sweeps(src, dst, mir, gp, max_tiles=4)
    C = len(src)
    assert sorted(r["p_can"][r["p_can"] >= 0].tolist()) == list(range(C))
    assert np.all(r["item_row0"] % 16 == 0) and r["rows"] == r["item_tiles"].sum() * 16
    ok = r["p_can"] >= 0
    assert np.array_equal(np.asarray(src)[r["p_can"][ok]], r["p_src"][ok])
    s = r["p_slot"][ok]
    assert s.max() <= 15 and np.all((s >= 0) == (np.asarray(mir)[r["p_can"][ok]] >= 0))
    print("items", len(r["item_tiles"]), "tiles", r["item_tiles"].sum(), "dense tiles", (C + 15) // 16, "segments", len(r["seg_dst"]))
