"""CPU ORACLE for the COV / MAT evaluation -- test infrastructure, NOT product code.

numpy restatement of /root/reference/src/agdiff/utils/evaluation/covmat.py:16-165.  Only tests/ may import this.

Parity status
  * PINNED for everything the reference's own Python does with a confusion matrix -- minima, thresholds, COV / MAT
    means, the evaluator's filtering (missing pos_gen, disconnected SMILES, fewer than ratio x references generated,
    truncation to ratio x references) and the summary table: tests/golden/g13_covmat.npz was produced by the reference's
    CovMatEvaluator / evaluate_conf / print_covmat_results on injected matrices (tests/golden/make_golden.py:g_covmat).
  * UNPINNED for the RMSD itself: get_best_rmsd (utils/chem.py:133-137) is rdkit's RemoveHs + rdMolAlign.GetBestRMS
    (third party, unpinned, absent from this image and from /root/reference; no reference test holds a value).  Restated
    from its published definition: the minimum, over the molecule's self-matches (atom mappings), of the RMSD after the
    optimal proper rotation and translation (AlignMol, uniform weights, no reflection).  The caller supplies the
    mappings (`perms`); with the identity only the value is an upper bound of GetBestRMS.
"""
import numpy as np


def kabsch_rmsd(x, y):
    """RMSD of x [m,3] onto y [m,3] after the optimal proper rotation + translation (Kabsch via SVD, float64)."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    x = x - x.mean(0)
    y = y - y.mean(0)
    H = x.T @ y
    U, sv, Vt = np.linalg.svd(H)
    d = np.sign(np.linalg.det(U) * np.linalg.det(Vt))
    sv = sv.copy()
    if d < 0:
        sv[-1] = -sv[-1]                         # proper rotations only: flip the smallest singular value
    msd = ((x * x).sum() + (y * y).sum() - 2.0 * sv.sum()) / x.shape[0]
    return float(np.sqrt(max(msd, 0.0)))


def best_rmsd(gen, ref, atom_idx, perms=None):
    """get_best_rmsd(gen_mol, ref_mol), utils/chem.py:133-137: atoms `atom_idx` (heavy atoms), min over `perms`
    (perms[p][k] = which selected reference atom the k-th selected generated atom is paired with)."""
    g = np.asarray(gen)[atom_idx]
    r = np.asarray(ref)[atom_idx]
    if perms is None:
        return kabsch_rmsd(g, r)
    return min(kabsch_rmsd(g, r[np.asarray(p)]) for p in perms)


def get_rmsd_confusion_matrix(pos_ref, pos_gen, atom_idx, perms=None):
    """covmat.py:16-35: [num_ref, num_gen], entry (j, i) = best RMSD of generated i against reference j."""
    R, G = pos_ref.shape[0], pos_gen.shape[0]
    out = -1.0 * np.ones([R, G], dtype=float)
    for i in range(G):
        for j in range(R):
            out[j, i] = best_rmsd(pos_gen[i], pos_ref[j], atom_idx, perms)
    return out


def evaluate_conf(confusion, threshold=0.5):
    """covmat.py:38-41."""
    ref_min = confusion.min(-1)
    return (ref_min <= threshold).mean(), ref_min.mean()


def covmat_scores(confusion, thresholds):
    """covmat.py:134-153 for one molecule: (COV-R [T], MAT-R, COV-P [T], MAT-P)."""
    thresholds = np.asarray(thresholds).flatten()
    ref_min = confusion.min(-1)
    gen_min = confusion.min(0)
    covr = (ref_min.reshape(-1, 1) <= thresholds.reshape(1, -1)).mean(0)
    covp = (gen_min.reshape(-1, 1) <= thresholds.reshape(1, -1)).mean(0)
    return covr, ref_min.mean(), covp, gen_min.mean()


def filter_items(items, ratio=2, filter_disconnected=True):
    """covmat.py:106-127: indices of the items the evaluator keeps, and how many generated conformers it uses."""
    kept = []
    for k, d in enumerate(items):
        if "pos_gen" not in d or "pos_ref" not in d:
            continue
        if filter_disconnected and ("." in d["smiles"]):
            continue
        n = d["num_atoms"]
        R = np.asarray(d["pos_ref"]).reshape(-1, n, 3).shape[0]
        G = np.asarray(d["pos_gen"]).reshape(-1, n, 3).shape[0]
        if G < R * ratio:
            continue
        kept.append((k, R * ratio))
    return kept
