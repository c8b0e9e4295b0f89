"""Synthetic checkpoints and GEOM-shaped synthetic molecules.

There is no pretrained checkpoint and no GEOM data in the reference tree or on the GPU box
(README.md:74 of the reference points at Google Drive), so tests and bench.py use
  * a closed-form deterministic weight filler, applied identically to the reference model
    (golden generation, this container only), to the oracle and to the HIP path;
  * a seeded molecule generator that reproduces the caller-side data contract of
    scripts/test.py:96-141 (bond graph extended to order 3, replicated `num_samples` times).
Everything here is integer-hash based (splitmix64) so the values are bit-reproducible on any
machine; no libm call is involved.
"""
import zlib

import numpy as np

_ALIAS = {
    "model_global.0.": "edge_encoder_global.",
    "model_global.1.": "encoder_global.",
    "model_global.2.": "grad_global_dist_mlp.",
    "model_local.0.": "edge_encoder_local.",
    "model_local.1.": "encoder_local.",
    "model_local.2.": "grad_local_dist_mlp.",
}


def canonical_key(key):
    """dualenc.py:103-108 registers every sub-module twice (attribute + ModuleList alias)."""
    for a, c in _ALIAS.items():
        if key.startswith(a):
            return c + key[len(a):]
    return key


def _splitmix_uniform(n, seed):
    """n uniform doubles in [0,1) from splitmix64(seed + k), k = 0..n-1 (exact integer math)."""
    x = (np.arange(n, dtype=np.uint64) + np.uint64(seed)) * np.uint64(0x9E3779B97F4A7C15)
    x ^= x >> np.uint64(30)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27)
    x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _u(key, shape, lo, hi, salt=0):
    n = int(np.prod(shape)) if len(shape) else 1
    seed = (zlib.crc32(key.encode()) + 0x51ED270B * salt) & 0xFFFFFFFF
    with np.errstate(over="ignore"):
        v = _splitmix_uniform(n, seed << 20)
    return (lo + (hi - lo) * v).reshape(shape).astype(np.float32)


def synth_tensor(key, shape, head_scale=1e-3):
    """Closed-form value for one state_dict entry, or None to keep the module's own value
    (betas / alphas / num_batches_tracked / GIN eps / rbf.offset)."""
    key = canonical_key(key)
    shape = tuple(shape)
    leaf = key.split(".")[-1]
    if key in ("betas", "alphas") or leaf in ("num_batches_tracked", "eps", "offset"):
        return None
    if leaf == "running_var":
        return _u(key, shape, 0.6, 1.4)
    if leaf == "running_mean":
        return _u(key, shape, -0.1, 0.1)
    if leaf == "beta":                         # ShiftedSoftplus learnable scalar (schnet.py:74)
        return _u(key, shape, 0.9, 1.1)
    if leaf == "attention_weights":            # dead parameter (schnet.py:106,126)
        return _u(key, shape, -1.0, 1.0)
    is_bn = (".norm1." in key or ".norm2." in key or ".batch_norms." in key)
    if is_bn and leaf == "weight":
        return _u(key, shape, 0.9, 1.1)
    if is_bn and leaf == "bias":
        return _u(key, shape, -0.05, 0.05)
    if leaf == "weight" and ("embedding" in key or "_emb" in key):
        w = _u(key, shape, -1.0, 1.0)
        if key.endswith("encoder_global.embedding.weight"):
            # rows get norms ~5.9 .. 12.5 so that max_norm=10 renorm (schnet.py:254) is active
            amp = 0.9 + 0.25 * (np.arange(shape[0]) % 5)
            w = w * amp[:, None].astype(np.float32)
        return w.astype(np.float32)
    if leaf == "weight" and len(shape) == 2:
        fan_in = shape[1]
        a = 1.7 / np.sqrt(fan_in)
        if key.endswith("feature_expansion.weight"):
            a = 0.35
        if key.endswith("distance_weighting.layer1.weight"):
            a = 0.5
        if "encoder_local.convs." in key and ".layers.0." in key:
            a = 0.3 / np.sqrt(fan_in)          # GIN sums ~8 messages per node: keep the gain < 1
        w = _u(key, shape, -a, a)
        if ".layers.2." in key and "dist_mlp" in key:
            w = w * np.float32(head_scale)
        return w
    if leaf == "bias":
        b = _u(key, shape, -0.1, 0.1)
        if ".layers.2." in key and "dist_mlp" in key:
            b = b * np.float32(head_scale)
        return b
    raise KeyError("no synthetic rule for %s %s" % (key, shape))


def synth_state_dict(template, head_scale=1e-3):
    """template: mapping key -> tensor (a freshly constructed model's state_dict). Returns a
    new dict of torch tensors with every fillable entry replaced."""
    import torch
    out = {}
    for k, t in template.items():
        v = synth_tensor(k, tuple(t.shape), head_scale)
        out[k] = t.clone() if v is None else torch.from_numpy(v.copy()).to(t.dtype).reshape(t.shape)
    return out


# ----------------------------------------------------------------------------- molecules
NUM_BOND_TYPES = 22          # len(BOND_TYPES), utils/chem.py:17 / edge.py:21
_ATOM_CHOICES = np.array([1, 6, 7, 8, 9, 16, 17])
_ATOM_P = np.array([0.50, 0.35, 0.07, 0.07, 0.004, 0.003, 0.003])
_BOND_CHOICES = np.array([1, 2, 12])          # SINGLE, DOUBLE, AROMATIC
_BOND_P = np.array([0.85, 0.10, 0.05])


def extend_graph_order_np(n, bond_src, bond_dst, bond_type, order=3, num_types=NUM_BOND_TYPES):
    """Host restatement of AddHigherOrderEdges / _extend_graph_order
    (utils/transforms.py:12-71, models/common.py:135-205): binarised adjacency powers up to
    `order`; k-hop pairs (k>1) get type num_types + k - 1; result sorted by (row, col)."""
    adj = np.zeros((n, n), dtype=np.int64)
    np.add.at(adj, (bond_src, bond_dst), 1)
    tmat = np.zeros((n, n), dtype=np.int64)
    np.add.at(tmat, (bond_src, bond_dst), bond_type)
    eye = np.eye(n, dtype=np.int64)
    mats = [eye, ((adj + eye) > 0).astype(np.int64)]
    for i in range(2, order + 1):
        mats.append(((mats[i - 1] @ mats[1]) > 0).astype(np.int64))
    order_mat = np.zeros_like(adj)
    for i in range(1, order + 1):
        order_mat += (mats[i] - mats[i - 1]) * i
    thigh = np.where(order_mat > 1, num_types + order_mat - 1, 0)
    assert (tmat * thigh == 0).all()
    tnew = tmat + thigh
    r, c = np.nonzero(tnew)                    # row-major == sorted by (row, col)
    return r.astype(np.int64), c.astype(np.int64), tnew[r, c].astype(np.int64)


def random_bonds(rng, n):
    """Random tree over n atoms (parent uniform among the previous 3 indices) plus n//10 ring
    closures between atoms <= 6 apart. Returns atom_type[n] and the raw symmetric
    bond list (row, col, type), both directions of every bond, in generation order."""
    atom_type = rng.choice(_ATOM_CHOICES, size=n, p=_ATOM_P).astype(np.int64)
    pairs = set()
    for a in range(1, n):
        p = int(rng.integers(max(0, a - 3), a))
        pairs.add((p, a))
    for _ in range(n // 10):
        a = int(rng.integers(0, n))
        b = a + int(rng.integers(2, 7))
        if b < n and (a, b) not in pairs:
            pairs.add((a, b))
    pairs = sorted(pairs)
    bt = rng.choice(_BOND_CHOICES, size=len(pairs), p=_BOND_P).astype(np.int64)
    src = np.array([p[0] for p in pairs] + [p[1] for p in pairs], dtype=np.int64)
    dst = np.array([p[1] for p in pairs] + [p[0] for p in pairs], dtype=np.int64)
    typ = np.concatenate([bt, bt])
    return atom_type, src, dst, typ


def random_molecule(rng, n, raw_bonds=False):
    """random_bonds + the order-3 extension AddHigherOrderEdges applies to every data set item
    (utils/transforms.py:12-71): atom_type[n] and the symmetric edge list (row, col, type) sorted by
    (row, col).  raw_bonds=True returns the bond list as it is (for forward(extend_order=True))."""
    atom_type, src, dst, typ = random_bonds(rng, n)
    if raw_bonds:
        return atom_type, src, dst, typ
    r, c, t = extend_graph_order_np(n, src, dst, typ, order=3)
    return atom_type, r, c, t


def repeat_molecule(atom_type, row, col, typ, num_samples, node_offset=0, graph_offset=0):
    """repeat_data (utils/misc.py:88-90) = PyG Batch.from_data_list of `num_samples` clones:
    node offsets k*n, batch = repeat_interleave(arange(G), n)."""
    n = atom_type.shape[0]
    k = np.arange(num_samples, dtype=np.int64)
    at = np.tile(atom_type, num_samples)
    off = (k * n + node_offset)[:, None]
    r = (row[None, :] + off).reshape(-1)
    c = (col[None, :] + off).reshape(-1)
    t = np.tile(typ, num_samples)
    batch = np.repeat(k + graph_offset, n)
    return at, r, c, t, batch


def alanine_dipeptide(num_samples=250):
    """BASELINE.json configs[0]: the molecule of examples/alanine_dipeptide.pdb (ACE-ALA-NME, 22 atoms in file
    order) with its 21 covalent bonds by standard peptide topology (both C=O double, the rest single; rdkit's
    own perception is unpinned), extended to order 3 and replicated `num_samples` times the way
    examples/test_alanine_dipeptide.py:292-298 does (250 copies).  Same dict as make_packed_batch."""
    elements = "H C H H C O N H C H C H H H C O N H C H H H".split()
    z = {"H": 1, "C": 6, "N": 7, "O": 8}
    atom_type = np.array([z[e] for e in elements], dtype=np.int64)
    # 1-based serials of the pdb: (i, j, bond type)
    bonds = [(2, 1, 1), (2, 3, 1), (2, 4, 1), (2, 5, 1), (5, 6, 2), (5, 7, 1),              # ACE
             (7, 8, 1), (7, 9, 1), (9, 10, 1), (9, 11, 1), (11, 12, 1), (11, 13, 1), (11, 14, 1),
             (9, 15, 1), (15, 16, 2), (15, 17, 1),                                            # ALA
             (17, 18, 1), (17, 19, 1), (19, 20, 1), (19, 21, 1), (19, 22, 1)]                 # NME
    i = np.array([b[0] - 1 for b in bonds]); j = np.array([b[1] - 1 for b in bonds]); t = np.array([b[2] for b in bonds])
    r, c, ty = extend_graph_order_np(22, np.concatenate([i, j]), np.concatenate([j, i]), np.concatenate([t, t]), order=3)
    at, r2, c2, t2, b2 = repeat_molecule(atom_type, r, c, ty, num_samples)
    return dict(atom_type=at, bond_index=np.stack([r2, c2]), bond_type=t2, batch=b2, num_graphs=int(num_samples),
                mol_id=np.zeros(num_samples, dtype=np.int64))


def sample_n_atoms(rng, kind):
    if kind == "qm9":
        return int(rng.integers(10, 30))
    if kind == "drugs":
        return int(np.clip(np.rint(rng.normal(44.0, 11.0)), 20, 181))
    if kind == "large":
        return 200
    raise ValueError(kind)


def make_packed_batch(kind, num_molecules, copies, seed=2021, raw_bonds=False):
    """Pack `num_molecules` distinct synthetic molecules, each replicated copies(rng) times,
    into one batch the way scripts/test.py builds a per-molecule batch (but many molecules per
    batch so that one GPU is filled). `copies` is an int or a callable rng -> int.
    Returns dict of int64 numpy arrays: atom_type, bond_index[2,E], bond_type, batch,
    num_graphs, mol_id[G] (which distinct molecule each graph is a copy of)."""
    rng = np.random.default_rng(seed)
    ats, rs, cs, ts, bs, mids = [], [], [], [], [], []
    node_off, g_off = 0, 0
    for m in range(num_molecules):
        n = sample_n_atoms(rng, kind)
        at, r, c, t = random_molecule(rng, n, raw_bonds=raw_bonds)
        g = copies(rng) if callable(copies) else int(copies)
        a2, r2, c2, t2, b2 = repeat_molecule(at, r, c, t, g, node_off, g_off)
        ats.append(a2); rs.append(r2); cs.append(c2); ts.append(t2); bs.append(b2)
        mids.append(np.full(g, m, dtype=np.int64))
        node_off += n * g
        g_off += g
    return dict(
        atom_type=np.concatenate(ats), bond_index=np.stack([np.concatenate(rs), np.concatenate(cs)]),
        bond_type=np.concatenate(ts), batch=np.concatenate(bs), num_graphs=g_off,
        mol_id=np.concatenate(mids))
