"""COV / MAT evaluation of generated conformers -- the step after the sampling path, mirroring
the reference's utils/evaluation/covmat.py:16-165 (get_rmsd_confusion_matrix, evaluate_conf, CovMatEvaluator,
print_covmat_results) without rdkit / PyG / easydict.

The RMSD confusion matrix [references x generated] is computed on the GPU (agdiff_rmsd_matrix, csrc/eval.hip:
Kabsch / Horn RMSD with proper rotations, minimum over atom mappings); so are its row / column minima.
There is no CPU fallback.

Data items are plain dicts (the reference uses PyG Data objects with an rdkit molecule):
    pos_ref [R*n, 3] or [R, n, 3], pos_gen [G*n, 3] or [G, n, 3], atom_type [n] (atomic numbers; hydrogens = 1 are
    removed like utils/chem.py:133-137 does with RemoveHs), smiles (optional; "." marks a disconnected molecule),
    perms [P, m] (optional): the molecule's heavy-atom self-matches, e.g. from rdkit
    `RemoveHs(mol).GetSubstructMatches(RemoveHs(mol), uniquify=False)`.  With them the matrix is rdkit's GetBestRMS;
    without (identity only) symmetric molecules get an UPPER BOUND of it -- COV is then a lower bound, MAT an upper
    bound, and the numbers are indicative only.
Force-field relaxation (use_force_field=True -> rdkit MMFF) is not available.
"""
import ctypes

import numpy as np

from . import _lib
from .config import Config


def _as_conformers(pos, n):
    import torch
    t = pos if hasattr(pos, "is_cuda") else torch.as_tensor(np.asarray(pos))
    return t.reshape(-1, n, 3).to(torch.float32)


def get_rmsd_confusion_matrix(data, useFF=False, device="cuda"):
    """covmat.py:16-35.  Returns a float32 torch tensor [num_ref, num_gen] on `device`."""
    import torch
    if useFF:
        raise NotImplementedError("MMFF relaxation needs rdkit (covmat.py:27-29); not available here")
    lib = _lib.load()
    at = np.asarray(data["atom_type"]).reshape(-1)
    n = at.shape[0]
    ref = _as_conformers(data["pos_ref"], n).to(device).contiguous()
    gen = _as_conformers(data["pos_gen"], n).to(device).contiguous()
    heavy = np.nonzero(at != 1)[0].astype(np.int32)
    if heavy.size == 0:
        raise ValueError("molecule without heavy atoms")
    m = int(heavy.size)
    idx = torch.from_numpy(heavy).to(device)
    perms = data.get("perms") if isinstance(data, dict) else getattr(data, "perms", None)
    P, pt = 0, None
    if perms is not None:
        pa = np.ascontiguousarray(np.asarray(perms, dtype=np.int32).reshape(-1, m))
        if pa.min() < 0 or pa.max() >= m or not all(np.array_equal(np.sort(r), np.arange(m)) for r in pa):
            raise ValueError("perms must hold permutations of the %d heavy atoms" % m)
        P, pt = pa.shape[0], torch.from_numpy(pa).to(device)
    R, G = ref.shape[0], gen.shape[0]
    out = torch.empty((R, G), dtype=torch.float32, device=device)
    scratch = torch.empty((R + G) * (3 * m + 1), dtype=torch.float32, device=device)
    with torch.cuda.device(out.device):
        _lib.check(lib.agdiff_rmsd_matrix(_lib.ptr(ref), _lib.ptr(gen), _lib.ptr(idx), _lib.ptr(pt), R, G, n, m, P,
                                          _lib.ptr(scratch), _lib.ptr(out), _lib.stream_ptr()), "agdiff_rmsd_matrix")
    return out


def matrix_minima(confusion):
    """(rmsd_ref_min [R], rmsd_gen_min [G]) of a confusion matrix on the GPU (covmat.py:135-136)."""
    import torch
    lib = _lib.load()
    c = confusion.contiguous()
    R, G = c.shape
    rmin = torch.empty(R, dtype=torch.float32, device=c.device)
    gmin = torch.empty(G, dtype=torch.float32, device=c.device)
    with torch.cuda.device(c.device):
        _lib.check(lib.agdiff_matrix_minima(_lib.ptr(c), R, G, _lib.ptr(rmin), _lib.ptr(gmin), _lib.stream_ptr()),
                   "agdiff_matrix_minima")
    return rmin, gmin


def evaluate_conf(data, useFF=False, threshold=0.5):
    """covmat.py:38-41: (coverage at `threshold`, mean of the references' smallest RMSD)."""
    rmin, _ = matrix_minima(get_rmsd_confusion_matrix(data, useFF=useFF))
    rmin = rmin.cpu().numpy().astype(np.float64)
    return (rmin <= threshold).mean(), rmin.mean()


def scores_from_minima(ref_min, gen_min, thresholds):
    """covmat.py:137-153 for one molecule."""
    thresholds = np.asarray(thresholds).flatten()
    ref_min, gen_min = np.asarray(ref_min, dtype=np.float64), np.asarray(gen_min, dtype=np.float64)
    covr = (ref_min.reshape(-1, 1) <= thresholds.reshape(1, -1)).mean(0, keepdims=True)
    covp = (gen_min.reshape(-1, 1) <= thresholds.reshape(1, -1)).mean(0, keepdims=True)
    return covr, ref_min.mean(), covp, gen_min.mean()


def print_covmat_results(results, print_fn=print):
    """covmat.py:44-74 (a dict of columns instead of a pandas DataFrame; same numbers, same MAT lines)."""
    cols = {
        "COV-R_mean": np.mean(results.CoverageR, 0), "COV-R_median": np.median(results.CoverageR, 0),
        "COV-R_std": np.std(results.CoverageR, 0),
        "COV-P_mean": np.mean(results.CoverageP, 0), "COV-P_median": np.median(results.CoverageP, 0),
        "COV-P_std": np.std(results.CoverageP, 0),
    }
    head = "%9s " % "" + " ".join("%12s" % c for c in cols)
    rows = ["%9.2f " % t + " ".join("%12.6f" % cols[c][k] for c in cols) for k, t in enumerate(results.thresholds)]
    print_fn("\n" + "\n".join([head] + rows))
    print_fn("MAT-R_mean: %.4f | MAT-R_median: %.4f | MAT-R_std %.4f"
             % (np.mean(results.MatchingR), np.median(results.MatchingR), np.std(results.MatchingR)))
    print_fn("MAT-P_mean: %.4f | MAT-P_median: %.4f | MAT-P_std %.4f"
             % (np.mean(results.MatchingP), np.median(results.MatchingP), np.std(results.MatchingP)))
    return cols


class CovMatEvaluator(object):
    """covmat.py:77-165.  `num_workers` is accepted and ignored: the confusion matrices come from the GPU."""

    def __init__(self, num_workers=8, use_force_field=False, thresholds=np.arange(0.05, 3.05, 0.05), ratio=2,
                 filter_disconnected=True, print_fn=print, confusion_fn=None):
        if use_force_field:
            raise NotImplementedError("MMFF relaxation needs rdkit; not available here")
        self.num_workers = num_workers
        self.use_force_field = use_force_field
        self.thresholds = np.array(thresholds).flatten()
        self.ratio = ratio
        self.filter_disconnected = filter_disconnected
        self.print_fn = print_fn
        # hook for callers that already hold the matrices (and for CPU tests of the filtering / reductions)
        self.confusion_fn = confusion_fn

    def __call__(self, packed_data_list, start_idx=0):
        filtered = []
        for data in packed_data_list:
            if "pos_gen" not in data or "pos_ref" not in data:
                continue
            if self.filter_disconnected and ("." in data.get("smiles", "")):
                continue
            n = int(np.asarray(data["atom_type"]).reshape(-1).shape[0])
            ref = _as_conformers(data["pos_ref"], n)
            gen = _as_conformers(data["pos_gen"], n)
            num_gen = ref.shape[0] * self.ratio
            if gen.shape[0] < num_gen:
                continue
            filtered.append(dict(data, pos_ref=ref, pos_gen=gen[:num_gen]))
        filtered = filtered[start_idx:]
        self.print_fn("Filtered: %d / %d" % (len(filtered), len(packed_data_list)))
        covr_scores, matr_scores, covp_scores, matp_scores = [], [], [], []
        for data in filtered:
            if self.confusion_fn is not None:
                cm = np.asarray(self.confusion_fn(data))
                ref_min, gen_min = cm.min(-1), cm.min(0)
            else:
                rmin, gmin = matrix_minima(get_rmsd_confusion_matrix(data))
                ref_min, gen_min = rmin.cpu().numpy(), gmin.cpu().numpy()
            covr, matr, covp, matp = scores_from_minima(ref_min, gen_min, self.thresholds)
            covr_scores.append(covr); matr_scores.append(matr); covp_scores.append(covp); matp_scores.append(matp)
        return Config({
            "CoverageR": np.vstack(covr_scores) if covr_scores else np.zeros((0, self.thresholds.shape[0])),
            "MatchingR": np.array(matr_scores),
            "thresholds": self.thresholds,
            "CoverageP": np.vstack(covp_scores) if covp_scores else np.zeros((0, self.thresholds.shape[0])),
            "MatchingP": np.array(matp_scores),
        })

    def close(self):
        pass


def main(argv=None):
    """python -m agdiff_amd.evaluation --samples samples_all.npz --refs refs.npz
    samples: `pos_gen_<i>` [G, n, 3] (agdiff_amd.driver output); refs: `pos_ref_<i>` [R, n, 3], `atom_type_<i>` [n],
    optional `smiles_<i>`, `perms_<i>` [P, m].  Prints the COV / MAT table of the reference's eval_covmat.py."""
    import argparse
    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("--samples", required=True)
    ap.add_argument("--refs", required=True)
    ap.add_argument("--ratio", type=int, default=2)
    args = ap.parse_args(argv)
    zs, zr = np.load(args.samples), np.load(args.refs)
    items = []
    for key in zr.files:
        if not key.startswith("pos_ref_"):
            continue
        i = key[len("pos_ref_"):]
        d = {"pos_ref": zr[key], "atom_type": zr["atom_type_" + i]}
        if "smiles_" + i in zr.files:
            d["smiles"] = str(zr["smiles_" + i])
        if "perms_" + i in zr.files:
            d["perms"] = zr["perms_" + i]
        if "pos_gen_" + i in zs.files:
            d["pos_gen"] = zs["pos_gen_" + i]
        items.append(d)
    res = CovMatEvaluator(ratio=args.ratio)(items)
    print_covmat_results(res)
    return res


if __name__ == "__main__":
    main()
