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
    (betas / alphas / the dsm sigmas / num_batches_tracked / GIN eps / rbf.offset)."""
    key = canonical_key(key)
    shape = tuple(shape)
    leaf = key.split(".")[-1]
    if key in ("betas", "alphas", "sigmas") or leaf in ("num_batches_tracked", "eps", "offset"):
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


# ----------------------------------------------------------------------------- a checkpoint with a restoring force
# The closed-form filler above gives a score network without any attraction between bonded atoms: over a full reference
# schedule (sigma up to 12) the atoms random-walk apart and the radius graph empties -- not the work a trained model does.
# restoring_state_dict() carves ONE channel through the same 854-key layout so that the local head returns, for every local
# edge (bond / 2-hop / 3-hop), a spring score  s_e = -kappa (d_e - d0[type_e])  on top of the filler's (small) output:
#   feature_expansion unit 0           x0 = gelu(d + B) = d + B           (B = 20: gelu is the identity to fp32 there)
#   edge_feature_mlp.0 unit 0          gelu(x0 + bond_emb[type][0]) = d - d0[type] + B        (bond_emb[:, 0] = -d0)
#   edge_feature_mlp.2 / combination_mlp.0 / .2 unit 0   pass it on:  edge_attr[0] = d - d0[type] + B
#   grad_local_dist_mlp.layers.0 units 0 / 1   relu(+-(edge_attr[0] - B)),  layers.1 units 0 / 1 pass them on,
#   layers.2                           -kappa z0 + kappa z1 = -kappa (d - d0)
# and takes channel 0 out of everything else that reads edge_attr or the spring units (filter networks, GIN layers, global
# head, the other head units), so that the rest of the network computes what the filler alone would on 127 channels.  With
# eq_transform (geometry.py:9-17) s_e < 0 pulls the two atoms together: every local edge is a spring of rest length d0.
# kappa = 0.1 keeps the explicit Langevin update stable at sigma_max (step / sigma * 4 kappa * degree = 0.12 * 0.4 * 8 < 2).
RESTORING_D0 = {"bond": 1.5, 23: 2.5, 24: 3.5}      # rest lengths (Angstrom) by edge type: bonds, 2-hop (22 + 2 - 1), 3-hop
RESTORING_B = 20.0


def restoring_state_dict(template, kappa=0.1, head_scale=1e-3):
    """synth_state_dict(template) with the spring channel described above (same keys, shapes, dtypes)."""
    return apply_restoring(synth_state_dict(template, head_scale), kappa)


def apply_restoring(sd, kappa=0.1):
    """The spring channel, written into a filler state_dict in place (aliased entries that share storage are edited once
    more with the same values: harmless)."""
    B = RESTORING_B

    def edit(ckey, fn):                       # apply fn to the tensor under its attribute name AND its ModuleList alias
        for k in sd:
            if canonical_key(k) == ckey:
                fn(sd[k])

    e = "edge_encoder_global."

    def fe_w(t):
        t[0, 0] = 1.0
    def fe_b(t):
        t[0] = B
    edit(e + "feature_expansion.weight", fe_w)
    edit(e + "feature_expansion.bias", fe_b)

    def emb(t):                               # column 0: minus the rest length of the type (type 0 = radius edges: unused)
        t[:, 0] = -RESTORING_D0["bond"]
        t[0, 0] = 0.0
        for ty in (23, 24):
            t[ty, 0] = -RESTORING_D0[ty]
    edit(e + "bond_emb.weight", emb)

    def first_of_pair(t):                     # Linear(256 -> 128) on [x || bond_emb]: unit 0 = x0 + emb0; nobody else reads them
        t[:, 0] = 0.0
        t[:, 128] = 0.0
        t[0, :] = 0.0
        t[0, 0] = 1.0
        t[0, 128] = 1.0
    def pass_on(t):                           # Linear(128 -> 128): unit 0 <- unit 0 only
        t[:, 0] = 0.0
        t[0, :] = 0.0
        t[0, 0] = 1.0
    def zero0(t):
        t[0] = 0.0
    edit(e + "edge_feature_mlp.0.weight", first_of_pair)
    edit(e + "edge_feature_mlp.0.bias", zero0)
    edit(e + "edge_feature_mlp.2.weight", pass_on)
    edit(e + "edge_feature_mlp.2.bias", zero0)

    def comb0(t):                             # Linear(256 -> 128) on [p || bond_emb]: unit 0 = p0 (emb0 already applied)
        t[:, 0] = 0.0
        t[:, 128] = 0.0
        t[0, :] = 0.0
        t[0, 0] = 1.0
    edit(e + "combination_mlp.0.weight", comb0)
    edit(e + "combination_mlp.0.bias", zero0)
    edit(e + "combination_mlp.2.weight", pass_on)
    edit(e + "combination_mlp.2.bias", zero0)
    # consumers of edge_attr other than the local head's spring units: channel 0 out
    def col0(t):
        t[:, 0] = 0.0
    def col128(t):
        t[:, 128] = 0.0
    for k in list(sd):
        ck = canonical_key(k)
        if ck.startswith("encoder_global.interactions.") and ck.endswith(".nn.0.weight"):
            col0(sd[k])
        if ck.startswith("encoder_local.convs.") and ck.endswith(".nn.layers.0.weight"):
            col0(sd[k])                       # (GIN adds edge_attr to h_j: channel 0 of the messages is not read)
        if ck == "grad_global_dist_mlp.layers.0.weight":
            col128(sd[k])

    h = "grad_local_dist_mlp."

    def l0w(t):                               # Linear(256 -> 128): units 0 / 1 = +-(edge_attr[0] - B); the others ignore it
        t[:, 128] = 0.0
        t[0, :] = 0.0
        t[1, :] = 0.0
        t[0, 128] = 1.0
        t[1, 128] = -1.0
    def l0b(t):
        t[0] = -B
        t[1] = B
    def l1w(t):                               # Linear(128 -> 64): units 0 / 1 pass the spring units on
        t[:, 0] = 0.0
        t[:, 1] = 0.0
        t[0, :] = 0.0
        t[1, :] = 0.0
        t[0, 0] = 1.0
        t[1, 1] = 1.0
    def l1b(t):
        t[0] = 0.0
        t[1] = 0.0
    def l2w(t):
        t[0, 0] = -kappa
        t[0, 1] = kappa
    edit(h + "layers.0.weight", l0w)
    edit(h + "layers.0.bias", l0b)
    edit(h + "layers.1.weight", l1w)
    edit(h + "layers.1.bias", l1b)
    edit(h + "layers.2.weight", l2w)
    return sd


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
