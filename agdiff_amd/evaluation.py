"""COV / MAT evaluation of generated conformers -- the step after the sampling path, mirroring
the reference's utils/evaluation/covmat.py:16-165 (get_rmsd_confusion_matrix, evaluate_conf, CovMatEvaluator,
print_covmat_results) without rdkit / PyG / easydict.

The RMSD confusion matrix [references x generated] is computed on the GPU (agdiff_rmsd_matrix, csrc/eval.hip:
Kabsch / Horn RMSD with proper rotations, minimum over atom mappings); so are its row / column minima.
There is no CPU fallback.

Data items are plain dicts (the reference uses PyG Data objects with an rdkit molecule):
    pos_ref [R*n, 3] or [R, n, 3], pos_gen [G*n, 3] or [G, n, 3], atom_type [n] (atomic numbers; hydrogens = 1 are
    removed like utils/chem.py:133-137 does with RemoveHs), smiles (optional; "." marks a disconnected molecule),
    bond_index [2, e] + bond_type [e] (or edge_index / edge_type; 2-/3-hop entries are ignored): the molecule's bonds, from
    which the heavy-atom self-matches GetBestRMS minimises over are enumerated here (heavy_atom_automorphisms) -- or
    perms [P, m]: those matches given directly.  With neither only the identity mapping is tried and symmetric molecules
    get an UPPER BOUND of GetBestRMS (COV a lower bound, MAT an upper bound).
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


def heavy_atom_automorphisms(atom_type, bond_index, bond_type, max_perms=65536):
    """The atom mappings rdkit's GetBestRMS minimises over (utils/chem.py:133-137, covmat.py:16-35): every match of the
    hydrogen-free molecule onto itself, i.e. every automorphism of its heavy-atom graph that keeps atomic numbers and bond
    types (rdkit: `RemoveHs(mol).GetSubstructMatches(RemoveHs(mol), uniquify=False)`; a molecule used as a query matches
    atoms by atomic number and bonds by type, aromatic with aromatic).  Returns int32 [P, m] over the heavy atoms in
    ascending index order (row 0 is the identity): perms[p][k] = image of heavy atom k.
      atom_type  [n] atomic numbers (hydrogens = 1 are dropped)
      bond_index [2, e], bond_type [e]  directed or undirected bond list; entries with type >= 22 (the 2-/3-hop edges of
                 utils/transforms.py:12-71) and bonds to hydrogens are ignored
    Colour refinement (1-WL over atom type and typed neighbourhoods) partitions the atoms first; a backtracking search then
    maps atoms class by class, checking every bond to the atoms already placed.  Raises if the group has more than
    `max_perms` elements (rdkit's own cap, maxMatches, is 1e6)."""
    at = np.asarray(atom_type).reshape(-1).astype(np.int64)
    heavy = np.nonzero(at != 1)[0]
    m = int(heavy.size)
    if m == 0:
        raise ValueError("molecule without heavy atoms")
    new_id = np.full(at.shape[0], -1, dtype=np.int64)
    new_id[heavy] = np.arange(m)
    bi = np.asarray(bond_index).reshape(2, -1).astype(np.int64)
    bt = np.asarray(bond_type).reshape(-1).astype(np.int64)
    adj = [dict() for _ in range(m)]                 # neighbour -> bond type
    for (u, v), ty in zip(bi.T, bt):
        if ty <= 0 or ty >= 22 or u == v:
            continue
        a, b = new_id[u], new_id[v]
        if a < 0 or b < 0:
            continue
        adj[a][int(b)] = int(ty)
        adj[b][int(a)] = int(ty)
    colour = [int(at[h]) for h in heavy]
    for _ in range(m):                               # 1-WL refinement to a fixed point
        sig = [(colour[i], tuple(sorted((colour[j], ty) for j, ty in adj[i].items()))) for i in range(m)]
        ids = {s_: k for k, s_ in enumerate(sorted(set(sig)))}
        new = [ids[s_] for s_ in sig]
        if len(set(new)) == len(set(colour)):
            colour = new
            break
        colour = new
    # visiting order: breadth first from the rarest colour class, so that every atom after the first of its component has
    # a placed neighbour to be checked against
    count = {c: colour.count(c) for c in set(colour)}
    order, seen = [], [False] * m
    for start in sorted(range(m), key=lambda i: (count[colour[i]], i)):
        if seen[start]:
            continue
        queue = [start]
        seen[start] = True
        while queue:
            i = queue.pop(0)
            order.append(i)
            for j in sorted(adj[i], key=lambda j: (count[colour[j]], j)):
                if not seen[j]:
                    seen[j] = True
                    queue.append(j)
    placed_nbrs = []                                  # for order[k]: its neighbours among order[:k]
    pos_in_order = {a: k for k, a in enumerate(order)}
    for k, i in enumerate(order):
        placed_nbrs.append([(j, ty) for j, ty in adj[i].items() if pos_in_order[j] < k])
    by_colour = {}
    for i in range(m):
        by_colour.setdefault(colour[i], []).append(i)
    perms, image, used = [], [-1] * m, [False] * m

    def place(k):
        if k == m:
            perms.append(list(image))
            if len(perms) > max_perms:
                raise ValueError("more than %d heavy-atom self-matches; pass a larger max_perms" % max_perms)
            return
        i = order[k]
        nb = placed_nbrs[k]
        if nb:          # candidates: the neighbours of an already placed neighbour's image
            j0, ty0 = nb[0]
            cands = [c for c, ty in adj[image[j0]].items() if ty == ty0]
        else:
            cands = by_colour[colour[i]]
        for c in sorted(cands):
            if used[c] or colour[c] != colour[i] or len(adj[c]) != len(adj[i]):
                continue
            if any(adj[c].get(image[j]) != ty for j, ty in nb):
                continue
            image[i], used[c] = c, True
            place(k + 1)
            image[i], used[c] = -1, False

    import sys
    lim = sys.getrecursionlimit()
    if lim < m + 100:
        sys.setrecursionlimit(m + 100)
    try:
        place(0)
    finally:
        sys.setrecursionlimit(lim)
    out = np.asarray(perms, dtype=np.int32).reshape(-1, m)
    ident = np.nonzero((out == np.arange(m)[None, :]).all(1))[0]
    if ident.size and ident[0] != 0:                  # identity first
        out[[0, ident[0]]] = out[[ident[0], 0]]
    return out


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
    get = (lambda k: data.get(k)) if isinstance(data, dict) else (lambda k: getattr(data, k, None))
    perms = get("perms")
    if perms is None:       # the molecule's own symmetry, as GetBestRMS finds it, when the item carries its bonds
        b_idx = get("bond_index") if get("bond_index") is not None else get("edge_index")
        b_typ = get("bond_type") if get("bond_type") is not None else get("edge_type")
        if b_idx is not None and b_typ is not None:
            perms = heavy_atom_automorphisms(at, b_idx, b_typ)
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
    optional `smiles_<i>`, `bond_index_<i>` + `bond_type_<i>` (symmetry-aware RMSD) or `perms_<i>` [P, m].  Prints the COV / MAT table of the reference's eval_covmat.py."""
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
        for k in ("bond_index", "bond_type", "edge_index", "edge_type"):
            if "%s_%s" % (k, i) in zr.files:
                d[k] = zr["%s_%s" % (k, i)]
        if "pos_gen_" + i in zs.files:
            d["pos_gen"] = zs["pos_gen_" + i]
        items.append(d)
    res = CovMatEvaluator(ratio=args.ratio)(items)
    print_covmat_results(res)
    return res


if __name__ == "__main__":
    main()
