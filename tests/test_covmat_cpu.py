"""CPU: the COV / MAT evaluation.  The reductions and the evaluator's filtering are checked against
what the reference's own CovMatEvaluator / evaluate_conf / print_covmat_results produced on injected confusion matrices
(tests/golden/g13_covmat.npz); the Kabsch RMSD of the oracle against closed-form cases (rdkit is absent: unpinned)."""
import numpy as np
import pytest

from helpers import load_golden
from oracle import covmat_oracle as CO


def _items(g):
    items = []
    rng = np.random.default_rng(0)
    for i in range(5):
        n, R, G = int(g["n%d" % i]), int(g["R%d" % i]), int(g["G%d" % i])
        d = {"smiles": "C.C" if int(g["disconnected%d" % i]) else "CC", "num_atoms": n,
             "atom_type": np.full(n, 6), "pos_ref": rng.normal(size=(R * n, 3))}
        if int(g["has_gen%d" % i]):
            d["pos_gen"] = rng.normal(size=(G * n, 3))
        items.append(d)
    return items


def test_oracle_reductions_and_filtering_match_reference_g13():
    g = load_golden("g13_covmat")
    items = _items(g)
    kept = CO.filter_items(items, ratio=2, filter_disconnected=True)
    assert [k for k, _ in kept] == g["kept"].tolist()
    assert [u for _, u in kept] == [2 * int(g["R%d" % k]) for k in g["kept"]]
    for row, k in enumerate(g["kept"]):
        covr, matr, covp, matp = CO.covmat_scores(g["confusion%d" % k], g["thresholds"])
        assert np.array_equal(covr, g["CoverageR"][row]) and np.array_equal(covp, g["CoverageP"][row])
        assert matr == g["MatchingR"][row] and matp == g["MatchingP"][row]
    cov, mat = CO.evaluate_conf(g["confusion0"], threshold=0.5)
    assert cov == g["evaluate_conf"][0] and mat == g["evaluate_conf"][1]


def test_evaluator_host_logic_matches_reference_g13():
    """agdiff_amd.evaluation.CovMatEvaluator with injected confusion matrices (no GPU work)."""
    from agdiff_amd.evaluation import CovMatEvaluator, print_covmat_results
    g = load_golden("g13_covmat")
    items = _items(g)
    mats = iter([g["confusion%d" % k] for k in g["kept"]])
    seen = []

    def confusion(data):
        seen.append((data["pos_ref"].shape[0], data["pos_gen"].shape[0]))
        return next(mats)
    logs = []
    ev = CovMatEvaluator(print_fn=logs.append, confusion_fn=confusion)
    res = ev(items)
    assert logs == ["Filtered: 2 / 5"]
    assert seen == [(int(g["R%d" % k]), 2 * int(g["R%d" % k])) for k in g["kept"]]     # pos_gen cut to ratio x references
    assert np.array_equal(res.thresholds, g["thresholds"]) and len(res.thresholds) == 60
    for key in ("CoverageR", "MatchingR", "CoverageP", "MatchingP"):
        assert np.array_equal(res[key], g[key]), key
    lines = []
    cols = print_covmat_results(res, print_fn=lines.append)
    assert list(cols) == g["df_columns"].tolist()
    assert np.allclose(np.stack([cols[c] for c in cols], axis=1), g["df_values"], rtol=0, atol=1e-15)
    assert lines[1:] == g["print_lines"][1:].tolist()                  # the two MAT lines, character for character
    with pytest.raises(NotImplementedError):
        CovMatEvaluator(use_force_field=True)


def test_oracle_kabsch_known_answers():
    rng = np.random.default_rng(5)
    x = rng.normal(size=(17, 3))
    assert CO.kabsch_rmsd(x, x) < 1e-7
    # proper rotation + translation: zero
    a, b, c = 0.3, -1.1, 2.0
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(c), -np.sin(c)], [0, np.sin(c), np.cos(c)]])
    y = x @ (Rz @ Ry @ Rx).T + np.array([3.0, -2.0, 0.5])
    assert CO.kabsch_rmsd(x, y) < 1e-6
    # a mirror image of a chiral set cannot be superposed by a proper rotation
    assert CO.kabsch_rmsd(x, x * np.array([1.0, 1.0, -1.0])) > 0.1
    # two points at distance d1 vs d2 on a line: rmsd = |d1 - d2| / 2
    p = np.array([[0.0, 0, 0], [2.0, 0, 0]]); r = np.array([[0.0, 0, 0], [0, 3.0, 0]])
    assert abs(CO.kabsch_rmsd(p, r) - 0.5) < 1e-12
    # symmetric labelling: swapping two equivalent atoms is undone by the matching permutation only
    idx = np.arange(17)
    swapped = x.copy(); swapped[[3, 4]] = swapped[[4, 3]]
    perm = idx.copy(); perm[[3, 4]] = perm[[4, 3]]
    assert CO.best_rmsd(swapped, x, idx, perms=[idx, perm]) < 1e-7 < CO.best_rmsd(swapped, x, idx)
    # hydrogens (not in atom_idx) do not count
    moved = x.copy(); moved[0] += 5.0
    assert CO.best_rmsd(moved, x, idx[1:]) < 1e-7


def _mol(atoms, bonds):
    """atoms: atomic numbers; bonds: (i, j, type) undirected -> both directions, as the reference's edge lists."""
    bi = np.array([[i, j] for i, j, _ in bonds] + [[j, i] for i, j, _ in bonds]).T
    bt = np.array([t for _, _, t in bonds] * 2)
    return np.array(atoms), bi, bt


def _check_group(perms, atoms, bi, bt):
    heavy = np.nonzero(atoms != 1)[0]
    m = heavy.size
    new_id = -np.ones(atoms.size, int); new_id[heavy] = np.arange(m)
    A = np.zeros((m, m), int)
    for (u, v), t in zip(bi.T, bt):
        if new_id[u] >= 0 and new_id[v] >= 0 and t < 22:
            A[new_id[u], new_id[v]] = t
    assert (perms[0] == np.arange(m)).all() and len({tuple(p) for p in perms}) == len(perms)
    for p in perms:
        assert sorted(p) == list(range(m)) and (atoms[heavy][p] == atoms[heavy]).all()
        assert (A[np.ix_(p, p)] == A).all()                       # bonds and bond types preserved
    S = {tuple(p) for p in perms}
    for p in perms[:8]:
        for q in perms[:8]:
            assert tuple(np.asarray(p)[q]) in S                    # closed under composition


def test_heavy_atom_automorphisms_hand_derivable_cases():
    """the symmetry mappings GetBestRMS enumerates (covmat.py:16-35, chem.py:133-137), on molecules whose
    heavy-atom symmetry group is known by hand."""
    from agdiff_amd.evaluation import heavy_atom_automorphisms as auto
    ring = lambda n, t: [(i, (i + 1) % n, t) for i in range(n)]
    cases = {
        "benzene (D6h on the ring: 12)": (_mol([6] * 6 + [1] * 6, ring(6, 12) + [(i, 6 + i, 1) for i in range(6)]), 12),
        "cyclohexane": (_mol([6] * 6, ring(6, 1)), 12),
        "toluene (mirror of the ring: 2)": (_mol([6] * 7, ring(6, 12) + [(0, 6, 1)]), 2),
        "tert-butanol (three methyls: 6)": (_mol([6, 6, 6, 6, 8], [(0, 1, 1), (0, 2, 1), (0, 3, 1), (0, 4, 1)]), 6),
        "neopentane (S4: 24)": (_mol([6] * 5, [(0, k, 1) for k in range(1, 5)]), 24),
        "pyridine (2)": (_mol([7] + [6] * 5, ring(6, 12)), 2),
        "acetic acid, C=O vs C-O (1)": (_mol([6, 6, 8, 8], [(0, 1, 1), (1, 2, 2), (1, 3, 1)]), 1),
        "acetate-like, two equal C-O (2)": (_mol([6, 6, 8, 8], [(0, 1, 1), (1, 2, 12), (1, 3, 12)]), 2),
        "biphenyl (2 x 2 x 2 = 8)": (_mol([6] * 12, ring(6, 12) + [(6 + i, 6 + (i + 1) % 6, 12) for i in range(6)] + [(0, 6, 1)]), 8),
        "naphthalene (4)": (_mol([6] * 10, [(0, 1, 12), (1, 2, 12), (2, 3, 12), (3, 4, 12), (4, 5, 12), (5, 0, 12),
                                           (4, 6, 12), (6, 7, 12), (7, 8, 12), (8, 9, 12), (9, 5, 12)]), 4),
        "two fragments: ethane + ethane, no swap of labels across... (2 x 2 x 2 = 8)": (_mol([6] * 4, [(0, 1, 1), (2, 3, 1)]), 8),
        "CF3 on a ring (2 x 6 = 12)": (_mol([6] * 7 + [9] * 3, ring(6, 12) + [(0, 6, 1), (6, 7, 1), (6, 8, 1), (6, 9, 1)]), 12),
    }
    for name, ((atoms, bi, bt), order) in cases.items():
        perms = auto(atoms, bi, bt)
        assert perms.shape == (order, int((atoms != 1).sum())), (name, perms.shape)
        _check_group(perms, atoms, bi, bt)
    # higher-order edges (types 23 / 24 of AddHigherOrderEdges) are not bonds; the cap raises
    atoms, bi, bt = _mol([6] * 6, ring(6, 1) + [(0, 2, 23), (0, 3, 24)])
    assert auto(atoms, bi, bt).shape[0] == 12
    with pytest.raises(ValueError):
        auto(*_mol([6] * 9, [(0, k, 1) for k in range(1, 9)]), max_perms=1000)        # 8! = 40320 > 1000
