"""Host-side (numpy) tables for the 4 x 4 pair-tile CFConv prototype (csrc/pairs4.hip).

A molecule's atoms are cut into blocks of four.  A TILE is the 4 x 4 block of pairs (target block tb, source block sg)
with sg <= tb; row 4 a + b is the pair (T_a, S_b).  sg < tb: every unordered pair of the two blocks once, carrying the
direct edge S_b -> T_a and the mirror edge T_a -> S_b.  sg == tb (diagonal): both directions as separate rows (a, b) and
(b, a), no mirror part.  Tiles are ordered by (molecule, 16-source sweep J = sg // 4, tb, sg): the tiles of one (J, tb)
form a GROUP (same four targets, up to four source blocks: the direct sums stay in registers over it), the groups of
one (molecule, J) a SWEEP (mirror sums of its 16 sources stay in registers).  Waves take contiguous ranges of whole
groups with balanced tile counts; a sweep cut by a wave boundary yields one mirror row set per part.

Only static data here (atom ids, flags, output rows); which pairs are edges, their scales and attributes change per step.
"""
import numpy as np


def build_pair_tiles(graph_ptr, mol_ids, num_waves):
    """-> dict(pt_atoms [T,8], pt_info [T,4], wave_tile_ptr [W+1], n_groups, n_sets,
               d_atom [n_groups*4] (atom of every direct row, -1: none), m_atom [n_sets*16] (atom of every mirror row))"""
    gp = np.asarray(graph_ptr, dtype=np.int64)
    atoms, tb_l, sg_l, mol_l, grp_first = [], [], [], [], []
    for m in mol_ids:
        g0, n = gp[m], gp[m + 1] - gp[m]
        nb4 = (n + 3) // 4
        for J in range((nb4 + 3) // 4):
            for tb in range(4 * J, nb4):
                for sg in range(4 * J, min(tb, 4 * J + 3) + 1):
                    row = np.full(8, -1, dtype=np.int64)
                    t = np.arange(4 * tb, min(4 * tb + 4, n))
                    s = np.arange(4 * sg, min(4 * sg + 4, n))
                    row[: t.shape[0]] = g0 + t
                    row[4: 4 + s.shape[0]] = g0 + s
                    atoms.append(row)
                    tb_l.append(tb); sg_l.append(sg); mol_l.append(m)
                    grp_first.append(sg == 4 * J)
    T = len(atoms)
    pt_atoms = np.stack(atoms) if T else np.zeros((0, 8), dtype=np.int64)
    tb_a, sg_a, mol_a = np.asarray(tb_l), np.asarray(sg_l), np.asarray(mol_l)
    first = np.asarray(grp_first, dtype=bool)
    gid = np.cumsum(first) - 1                                   # group of every tile
    n_groups = int(gid[-1]) + 1 if T else 0
    g_start = np.nonzero(first)[0]
    g_len = np.diff(np.concatenate([g_start, [T]]))
    # waves: contiguous ranges of whole groups, balanced by tiles
    cum = np.concatenate([[0], np.cumsum(g_len)])
    tgt = cum[-1] * np.arange(num_waves + 1) / num_waves
    gptr = np.searchsorted(cum, tgt, side="left")
    gptr[0], gptr[-1] = 0, n_groups
    gptr = np.maximum.accumulate(gptr)
    wave_tile_ptr = cum[gptr]
    wave_of_tile = np.searchsorted(wave_tile_ptr, np.arange(T), side="right") - 1
    sweep_key = mol_a * 64 + sg_a // 4
    new_set = np.ones(T, dtype=bool)
    new_set[1:] = (sweep_key[1:] != sweep_key[:-1]) | (wave_of_tile[1:] != wave_of_tile[:-1])
    sid = np.cumsum(new_set) - 1
    n_sets = int(sid[-1]) + 1 if T else 0
    last_of_group = np.ones(T, dtype=bool)
    last_of_group[:-1] = first[1:]
    flush_m = np.ones(T, dtype=bool)
    flush_m[:-1] = new_set[1:]
    pt_info = np.zeros((T, 4), dtype=np.int64)
    pt_info[:, 0] = (sg_a & 3) | (last_of_group.astype(np.int64) << 2) | (flush_m.astype(np.int64) << 3)
    pt_info[:, 1] = gid
    pt_info[:, 2] = sid
    # output rows -> atoms
    d_atom = np.full(n_groups * 4, -1, dtype=np.int64)
    d_atom.reshape(-1, 4)[gid[first]] = pt_atoms[first][:, :4]
    m_atom = np.full(n_sets * 16, -1, dtype=np.int64)
    ma = m_atom.reshape(-1, 4, 4)
    ma[sid, sg_a & 3] = pt_atoms[:, 4:]
    diag = tb_a == sg_a
    return dict(pt_atoms=pt_atoms, pt_info=pt_info, wave_tile_ptr=wave_tile_ptr, n_groups=n_groups, n_sets=n_sets,
                d_atom=d_atom, m_atom=m_atom, diag=diag, tiles=T)


def pair_rows(tabs):
    """(target atom, source atom, is-diagonal-tile) of every row (-1 atoms: none)."""
    pa = tabs["pt_atoms"]
    t = np.repeat(pa[:, :4], 4, axis=1).reshape(-1)              # row 4a+b -> T_a
    s = np.tile(pa[:, 4:], (1, 4)).reshape(-1)                   # row 4a+b -> S_b
    d = np.repeat(tabs["diag"], 16)
    return t, s, d


if __name__ == "__main__":
    gp = np.array([0, 45, 80, 153])
    tb = build_pair_tiles(gp, [0, 1], 7)
    print("tiles", tb["tiles"], "groups", tb["n_groups"], "sets", tb["n_sets"], tb["wave_tile_ptr"])
    t, s, d = pair_rows(tb)
    ok = (t >= 0) & (s >= 0) & (t != s)
    # every ordered pair (t, s), t != s, of a molecule is covered exactly once as direct-or-mirror
    cover = {}
    for ti, si, di in zip(t[ok], s[ok], d[ok]):
        cover[(ti, si)] = cover.get((ti, si), 0) + 1
        if not di:
            cover[(si, ti)] = cover.get((si, ti), 0) + 1
    n_pairs = 45 * 44 + 35 * 34
    assert len(cover) == n_pairs and set(cover.values()) == {1}, (len(cover), n_pairs)
    assert tb["tiles"] == 78 + 45
    print("ok")
