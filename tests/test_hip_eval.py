"""GPU (MI355X): the RMSD confusion matrix and its minima (csrc/eval.hip) through agdiff_amd.evaluation against the
oracle (oracle/covmat_oracle.py: Kabsch by SVD in float64).  Tolerance: 2e-5 absolute in Angstrom on values of
order 1 (positions are fp32; the kernel accumulates and solves in fp64)."""
import numpy as np
import pytest
import torch

from oracle import covmat_oracle as CO

pytestmark = pytest.mark.gpu
ATOL = 2e-5


def _mol(rng, n, R, G, frac_h=0.5):
    at = np.where(rng.random(n) < frac_h, 1, rng.choice([6, 7, 8], size=n))
    at[0] = 6
    base = rng.normal(size=(n, 3)) * 1.5
    ref = base[None] + 0.3 * rng.normal(size=(R, n, 3))
    gen = base[None] + 0.5 * rng.normal(size=(G, n, 3))
    return at, ref.astype(np.float32), gen.astype(np.float32)


@pytest.mark.parametrize("n,R,G,with_perms", [(23, 5, 10, False), (61, 17, 33, True), (9, 1, 1, True), (200, 3, 40, False)])
def test_rmsd_matrix_matches_oracle(n, R, G, with_perms):
    from agdiff_amd.evaluation import get_rmsd_confusion_matrix, matrix_minima
    rng = np.random.default_rng(n)
    at, ref, gen = _mol(rng, n, R, G)
    heavy = np.nonzero(at != 1)[0]
    m = heavy.size
    data = {"atom_type": at, "pos_ref": ref.reshape(-1, 3), "pos_gen": gen.reshape(-1, 3)}
    perms = None
    if with_perms:
        perms = [np.arange(m)]
        for _ in range(4):
            p = np.arange(m)
            a, b = rng.choice(m, size=2, replace=False)
            p[[a, b]] = p[[b, a]]
            perms.append(p)
        data["perms"] = np.stack(perms)
        # make one generated conformer the relabelled copy of a reference: only the matching mapping gives ~0
        g0 = ref[0].copy()
        g0[heavy] = ref[0][heavy][perms[2]]
        gen[0] = g0
        data["pos_gen"] = gen.reshape(-1, 3)
    got = get_rmsd_confusion_matrix(data)
    assert got.shape == (R, G) and got.is_cuda
    want = CO.get_rmsd_confusion_matrix(ref, gen, heavy, perms)
    assert np.abs(got.cpu().numpy() - want).max() < ATOL
    if with_perms:
        assert got[0, 0] < 1e-4
    rmin, gmin = matrix_minima(got)
    assert torch.equal(rmin, got.min(dim=1).values) and torch.equal(gmin, got.min(dim=0).values)


def test_rmsd_degenerate_geometries():
    """identical conformers, a rigidly moved copy, a mirror image, planar and collinear molecules"""
    from agdiff_amd.evaluation import get_rmsd_confusion_matrix
    rng = np.random.default_rng(3)
    n = 12
    at = np.full(n, 6)
    x = rng.normal(size=(n, 3)).astype(np.float32)
    a = 0.7
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], dtype=np.float32)
    moved = x @ Rz.T + np.array([4.0, -1.0, 2.0], dtype=np.float32)
    mirror = x * np.array([1.0, 1.0, -1.0], dtype=np.float32)
    planar = x.copy(); planar[:, 2] = 0.0
    planar2 = planar @ Rz.T
    line = np.zeros((n, 3), dtype=np.float32); line[:, 0] = np.arange(n)
    line2 = np.zeros((n, 3), dtype=np.float32); line2[:, 1] = 1.1 * np.arange(n)
    ref = np.stack([x, planar, line])
    gen = np.stack([x, moved, mirror, planar2, line2])
    got = get_rmsd_confusion_matrix({"atom_type": at, "pos_ref": ref, "pos_gen": gen}).cpu().numpy()
    want = CO.get_rmsd_confusion_matrix(ref, gen, np.arange(n))
    assert np.isfinite(got).all() and np.abs(got - want).max() < ATOL
    assert got[0, 0] < 1e-5 and got[0, 1] < 1e-5 and got[0, 2] > 0.1 and got[1, 3] < 1e-5


def test_covmat_evaluator_end_to_end_and_limits():
    from agdiff_amd import _lib
    from agdiff_amd.evaluation import CovMatEvaluator, evaluate_conf, get_rmsd_confusion_matrix
    rng = np.random.default_rng(8)
    items, want = [], []
    for k, (n, R) in enumerate([(14, 4), (30, 6), (11, 3)]):
        at, ref, gen = _mol(rng, n, R, 2 * R + k)
        items.append({"atom_type": at, "pos_ref": torch.from_numpy(ref).reshape(-1, 3), "smiles": "CC" if k != 2 else "C.C",
                      "pos_gen": torch.from_numpy(gen).reshape(-1, 3)})
        if k != 2:
            cm = CO.get_rmsd_confusion_matrix(ref, gen[:2 * R], np.nonzero(at != 1)[0])
            want.append(CO.covmat_scores(cm, np.arange(0.05, 3.05, 0.05)))
    res = CovMatEvaluator(print_fn=lambda s: None)(items)
    assert res.CoverageR.shape == (2, 60)
    for row, (covr, matr, covp, matp) in enumerate(want):
        assert np.array_equal(res.CoverageR[row], covr) and np.array_equal(res.CoverageP[row], covp)
        assert abs(res.MatchingR[row] - matr) < ATOL and abs(res.MatchingP[row] - matp) < ATOL
    cov, mat = evaluate_conf(items[0], threshold=0.5)
    cm0 = CO.get_rmsd_confusion_matrix(items[0]["pos_ref"].numpy().reshape(4, 14, 3), items[0]["pos_gen"].numpy().reshape(8, 14, 3),
                                       np.nonzero(items[0]["atom_type"] != 1)[0])
    assert (cov, round(mat, 4)) == (CO.evaluate_conf(cm0)[0], round(CO.evaluate_conf(cm0)[1], 4))
    big = {"atom_type": np.full(300, 6), "pos_ref": np.zeros((1, 300, 3), np.float32), "pos_gen": np.zeros((1, 300, 3), np.float32)}
    with pytest.raises(_lib.AgdiffLimitError):
        get_rmsd_confusion_matrix(big)


def test_rmsd_is_symmetry_aware_when_the_item_carries_its_bonds():
    """GetBestRMS aligns over every self-match of the heavy-atom graph (chem.py:133-137): a generated conformer that is a
    reference with its ring atoms relabelled by a symmetry of the molecule (here: toluene's mirror and a cyclohexane
    rotation) scores ~0 when the item carries bond_index / bond_type, and stays an upper bound without them."""
    from agdiff_amd.evaluation import get_rmsd_confusion_matrix, heavy_atom_automorphisms
    rng = np.random.default_rng(11)
    ring = [(i, (i + 1) % 6, 12) for i in range(6)]
    for name, atoms, bonds, sigma in (
            ("toluene", [6] * 7 + [1] * 8, ring + [(0, 6, 1)] + [(1 + k, 7 + k, 1) for k in range(5)] + [(6, 12 + k, 1) for k in range(3)],
             [0, 5, 4, 3, 2, 1, 6]),                                # mirror through C0 .. C3 (+ the methyl carbon)
            ("cyclohexane", [6] * 6, [(i, (i + 1) % 6, 1) for i in range(6)], [2, 3, 4, 5, 0, 1])):
        at = np.array(atoms)
        n, heavy = at.size, np.nonzero(at != 1)[0]
        bi = np.array([[i, j] for i, j, _ in bonds] + [[j, i] for i, j, _ in bonds]).T
        bt = np.array([t for _, _, t in bonds] * 2)
        ref = (rng.normal(size=(3, n, 3)) * 1.4).astype(np.float32)
        gen = (rng.normal(size=(4, n, 3)) * 1.4).astype(np.float32)
        gen[1] = ref[2]
        gen[1][heavy] = ref[2][heavy][np.array(sigma)]                # heavy atom k sits where sigma(k) sat
        item = {"atom_type": at, "pos_ref": ref, "pos_gen": gen}
        plain = get_rmsd_confusion_matrix(item).cpu().numpy()
        sym = get_rmsd_confusion_matrix(dict(item, bond_index=bi, bond_type=bt)).cpu().numpy()
        perms = heavy_atom_automorphisms(at, bi, bt)
        want = CO.get_rmsd_confusion_matrix(ref, gen, heavy, list(perms))
        assert np.abs(sym - want).max() < ATOL, name
        assert sym[2, 1] < 1e-4 < plain[2, 1] and (sym <= plain + ATOL).all(), name
