"""Weight pre-packing: reference state_dict (SURVEY.md §8b key layout) -> the device buffers and the
agdiff_params_t struct the HIP kernels read (include/agdiff_hip.h).

Output-preserving transformations applied here (each checked against the oracle in tests/):
  * eval-mode BatchNorm folded into the adjacent Linear (schnet.py:153-158, gin.py:131-132);
  * MLPEdgeEncoder: the bond-embedding halves of the two 256->128 layers become per-edge-type
    tables; edge_feature_mlp.2 is folded into combination_mlp.0; the size-1 softmax attention
    (== 1.0) is dropped (edge.py:84-103);
  * conv1/conv2 first filter layers and lin1 layers are fused along the output dimension.
All folding is done in float64 and rounded once to float32.

Since round 2b also the *filter polynomials* (DESIGN.md §4a): MLPEdgeEncoder followed by a CFConv's filter network is a
function of the edge length alone for a given edge type, fitted here in float64 at Chebyshev nodes and handed to the
kernels only when it reproduces the networks to POLY_TOL (otherwise those edges keep the MLP kernels).
"""
import ctypes

import numpy as np

from . import _lib

H = 128


def _np(sd, key):
    return sd[key].detach().cpu().double().numpy()


def bf16_round(x32):
    """fp32 array -> (bf16 bits as uint16, the rounded value as fp32), round-to-nearest-even."""
    u = np.ascontiguousarray(x32, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32)
    return r.astype(np.uint16), (r << 16).astype(np.uint32).view(np.float32)


def _block_columns():
    """Input-feature offsets (within a 32-feature k-tile) held by each lane of a weight block:
    cols[lane][j], j = 0..7: {4q..4q+3} then {16+4q..16+4q+3}, q = lane >> 4 (csrc/common.hpp)."""
    lane = np.arange(64)
    j = np.arange(8)
    return 16 * (j >> 2)[None, :] + 4 * (lane >> 4)[:, None] + (j & 3)[None, :]      # [64, 8]


def pack_blocks(W, kouter=False, mode=0):
    """Linear weight W[out, in] -> MFMA-operand-major 2-KiB blocks of 16 outputs x 32 inputs
    (include/agdiff_hip.h): lane l of block (ot, t) holds W[16*ot + (l & 15)][32*t + cols[l][0..7]].
      mode 0 (fp32):   unit u = fp32 of elements 4u..4u+3                      -> [2][64][4] floats
      mode 1 (bf16x3): unit 0 = bf16(w) of the 8 elements, unit 1 = bf16(w - hi) -> [2][64][8] bf16
      mode 2 (f16x3):  the same with fp16 (local branch, agdiff_params_t.precision_local)
    Blocks are ordered [OT][KT] ("pk") or [KT][OT] ("pkk", kouter).  Returned as a float32 array
    (bf16 bit patterns viewed as float32 in mode 1)."""
    W = np.asarray(W, dtype=np.float64)
    out, inn = W.shape
    OT, KT = (out + 15) // 16, (inn + 31) // 32
    Wp = np.zeros((OT * 16, KT * 32), dtype=np.float64)
    Wp[:out, :inn] = W
    rows = (np.arange(64) & 15)[:, None]                   # [64, 1]
    cols = _block_columns()                                # [64, 8]
    vals = np.empty((OT, KT, 64, 8), dtype=np.float32)
    for ot in range(OT):
        for t in range(KT):
            vals[ot, t] = Wp[16 * ot + rows, 32 * t + cols].astype(np.float32)
    if kouter:
        vals = vals.transpose(1, 0, 2, 3)
    vals = np.ascontiguousarray(vals)
    nb = OT * KT
    if mode == 0:
        # [block][lane][u][4] -> [block][u][lane][4]
        return np.ascontiguousarray(vals.reshape(nb, 64, 2, 4).transpose(0, 2, 1, 3)).reshape(-1)
    if mode == 2:            # split fp16: hi = fp16(w), lo = fp16(w - hi) (values beyond fp16's range saturate)
        v = np.clip(vals.astype(np.float64), -65504.0, 65504.0)
        hi = v.astype(np.float16)
        lo = (v - hi.astype(np.float64)).astype(np.float16)
        blk = np.stack([hi.reshape(nb, 64, 8), lo.reshape(nb, 64, 8)], axis=1)
        return np.ascontiguousarray(blk).reshape(-1).view(np.float32)
    hb, hv = bf16_round(vals)
    lb, _ = bf16_round(vals - hv)
    blk = np.stack([hb.reshape(nb, 64, 8), lb.reshape(nb, 64, 8)], axis=1)      # [block][part][lane][8]
    return np.ascontiguousarray(blk).reshape(-1).view(np.float32)


def unpack_blocks(flat, out, inn, kouter=False, mode=0):
    """Inverse of pack_blocks (tests): fp32 W[out, in] (hi + lo in mode 1)."""
    OT, KT = (out + 15) // 16, (inn + 31) // 32
    nb = OT * KT
    if mode == 0:
        vals = np.asarray(flat, dtype=np.float32).reshape(nb, 2, 64, 4).transpose(0, 2, 1, 3).reshape(nb, 64, 8)
    elif mode == 2:
        h = np.ascontiguousarray(flat).view(np.float16).reshape(nb, 2, 64, 8).astype(np.float32)
        vals = h[:, 0] + h[:, 1]
    else:
        u = np.ascontiguousarray(flat).view(np.uint16).reshape(nb, 2, 64, 8).astype(np.uint32)
        v = (u << 16).view(np.float32)
        vals = v[:, 0] + v[:, 1]
    vals = vals.reshape((KT, OT, 64, 8) if kouter else (OT, KT, 64, 8))
    if kouter:
        vals = vals.transpose(1, 0, 2, 3)
    W = np.zeros((OT * 16, KT * 32), dtype=np.float32)
    rows = (np.arange(64) & 15)[:, None]
    cols = _block_columns()
    for ot in range(OT):
        for t in range(KT):
            W[16 * ot + rows, 32 * t + cols] = vals[ot, t]
    return W[:out, :inn]


def pow2_norm(W):
    """The power of two n with max |n W| in (0.5, 1] (1 for an all-zero matrix): exact to apply and to undo."""
    m = float(np.abs(np.asarray(W, dtype=np.float64)).max())
    return 1.0 if not (m > 0.0) or not np.isfinite(m) else 2.0 ** -int(np.ceil(np.log2(m)))


def split_fp16_error(W):
    """(normwise error of the split-fp16 image of W: max |w - (hi + lo)| / max |w|, whether a value was clipped at fp16's range):
    what packing a matrix in mode 2 costs.  11 + 11 mantissa bits give 2^-22 while the lo parts are normal fp16 numbers; a lo part
    below 6e-5 has fewer bits and one below 6e-8 is lost, an ABSOLUTE floor -- so the figure grows as max |w| falls below ~0.1."""
    W = np.asarray(W, dtype=np.float64)
    m = float(np.abs(W).max()) if W.size else 0.0
    if not (m > 0.0):
        return 0.0, False
    v = np.clip(W, -65504.0, 65504.0)
    hi = v.astype(np.float32).astype(np.float16)
    lo = (v - hi.astype(np.float64)).astype(np.float32).astype(np.float16)
    return float(np.abs(W - hi.astype(np.float64) - lo.astype(np.float64)).max() / m), bool(m > 65504.0)


# split-fp16 is taken for a branch only while every matrix it packs keeps this normwise accuracy (2^-19: an eighth of what the
# mode gives O(1) weights, ~10 x better than split-bf16's 2^-16) and nothing is clipped; otherwise the branch runs in split-bf16
SPLIT_FP16_MAX_ERR = 2.0 ** -19
SPLIT_FP16_ACT_LIMIT = 60000.0       # activations that become split-fp16 operands (encoder_activation_max; fp16 saturates at 65504)


def fold_bn(W, b, sd, p, eps=1e-5):
    """Linear(W, b) followed by eval BatchNorm1d `p` -> one Linear."""
    s = _np(sd, p + ".weight") / np.sqrt(_np(sd, p + ".running_var") + eps)
    return W * s[:, None], (b - _np(sd, p + ".running_mean")) * s + _np(sd, p + ".bias")


# ---------------------------------------------------------------------------------- radius-edge polynomials
# A radius edge (type 0, d < cutoff by construction: common.py:217) has NO input but its length: edge_attr = MLPEdgeEncoder(d,
# type 0) (edge.py:84-103), the CFConv filter nn(edge_attr) (schnet.py:169-179) and the edge_attr half of the global head's
# first layer (common.py:106-109) are therefore functions of ONE scalar on [0, cutoff] -- compositions of Linear, GELU
# and softplus, i.e. analytic -- and a Chebyshev expansion converges geometrically.  They are fitted here in float64
# (interpolation at Chebyshev nodes) and ACCEPTED ONLY IF they reproduce the networks on a dense grid to `POLY_TOL` of the
# largest value; otherwise the kernels evaluate the MLPs for every edge as before (poly_kt = 0).
POLY_TOL = 1e-6
POLY_MAX_KT = 4          # include/agdiff_hip.h AGDIFF_POLY_MAX_KT: 32 / 64 terms for smooth checkpoints, 96 / 128 before the filter MLPs


def poly_feature_order(kt):
    """Natural column n of a packed K = 32 kt matrix -> index f of the product-basis function phi_f it multiplies
    (include/agdiff_hip.h, agdiff_params_t.poly_kt): operand element j of lane quarter q in k-tile t sits at natural
    column 32 t + 16 (j >> 2) + 4 q + (j & 3) (pack_blocks / csrc/common.hpp) and holds phi[8 (4 t + q) + j]."""
    order = np.empty(32 * kt, dtype=np.int64)
    for t in range(kt):
        for q in range(4):
            for j in range(8):
                order[32 * t + 16 * (j >> 2) + 4 * q + (j & 3)] = 8 * (4 * t + q) + j
    return order


def poly_basis_matrix(K):
    """M[n, f]: coefficient of T_n in phi_f = T_{8 g} T_j (f = 8 g + j): T_a T_b = (T_{a+b} + T_{|a-b|}) / 2."""
    M = np.zeros((K, K))
    for f in range(K):
        g, j = divmod(f, 8)
        a = 8 * g
        if a == 0 or j == 0:
            M[a + j, f] = 1.0
        else:
            M[a + j, f] += 0.5
            M[a - j, f] += 0.5
    return M


def poly_features(x, K):
    """phi_f(x), f < K, evaluated the way the kernels do (csrc/edge.hip: ag_poly_features): float64 here."""
    x = np.asarray(x, dtype=np.float64)
    T = [np.ones_like(x), x]
    for n in range(2, 9):
        T.append(2 * x * T[-1] - T[-2])
    T8 = T[8]
    G = [np.ones_like(x), T8]
    G.append(2 * T8 * T8 - 1)                 # T16
    G.append(2 * G[2] * T8 - T8)              # T24
    G.append(2 * G[2] * G[2] - 1)             # T32
    G.append(2 * G[4] * T8 - G[3])            # T40
    G.append(2 * G[3] * G[3] - 1)             # T48
    G.append(2 * G[6] * T8 - G[5])            # T56
    while len(G) < K // 8:                    # T64 .. T120 (k-tiles 2, 3): the recurrence in steps of eight
        G.append(2 * T8 * G[-1] - G[-2])
    return np.stack([G[f // 8] * T[f % 8] for f in range(K)], axis=-1)


def fit_poly(fn, cutoff, K, lo=0.0):
    """Coefficients C[out, K] (product basis) of the degree-(K-1) interpolant of fn: d[M] -> [M, out] at the K Chebyshev
    nodes of [lo, cutoff] (x = 2 (d - lo) / (cutoff - lo) - 1), and its largest error on a dense grid relative to the
    largest |fn|."""
    k = np.arange(K)
    xn = np.cos(np.pi * (k + 0.5) / K)
    half = (cutoff - lo) / 2.0
    fv = fn(lo + (xn + 1.0) * half)                                         # [K, out]
    b = np.polynomial.chebyshev.chebfit(xn, fv, K - 1)                      # [K, out] Chebyshev coefficients
    c = np.linalg.solve(poly_basis_matrix(K), b)                            # product-basis coefficients
    dg = np.linspace(lo, cutoff, 4097)
    ref = fn(dg)
    err = np.abs(poly_features((dg - lo) / half - 1.0, K) @ c - ref).max() / max(np.abs(ref).max(), 1e-300)
    return c.T, float(err)


# Local edges LONGER than the cutoff (bonded atoms far apart at high sigma: every local edge of the local-only steps early in
# the schedule) contribute nothing to the CFConvs (envelope 0, schnet.py:140-146) but the GIN layers and the local head read
# their edge_attr rows at any length (dualenc.py:214-239).  Beyond the cutoff the encoder's GELUs are saturated and edge_attr(d)
# is very smooth: a 32-term fit on [cutoff, 10 cutoff] reproduces it to ~1e-9 (synthetic and default-initialised weights), so
# those rows come from a second, "far" coefficient set per type instead of the encoder MLP (agdiff_params_t.attr_poly_far_slots).
ATTR_FAR_FACTOR = 10.0
ATTR_POLY_MAX_SETS = 9           # csrc/edge.hip AG_ATTRP_MAX_SLOTS: near + far sets the kernel holds in LDS


def fit_attr_far(sd, cfg, typ):
    """(coefficients [128, 32] in natural packed column order, error) of edge_attr(d, typ) on [cutoff, ATTR_FAR_FACTOR cutoff]."""
    key = (_poly_weights_digest(sd, cfg), float(cfg.cutoff), int(typ), "attr_far")
    if key not in _FIT_CACHE:
        fn = _edge_attr_fn(sd, "edge_encoder_global", typ)
        c, err = fit_poly(lambda d: fn(d).numpy(), ATTR_FAR_FACTOR * float(cfg.cutoff), 32, lo=float(cfg.cutoff))
        _FIT_CACHE[key] = (c[:, poly_feature_order(1)], err)
    return _FIT_CACHE[key]


def _edge_attr_fn(sd, e, typ):
    """d[M] (float64 numpy) -> edge_attr [M, 128] of edges of type `typ`, MLPEdgeEncoder.forward (edge.py:84-103) in
    float64; the trailing attention factor is a softmax over a size-1 axis, i.e. 1."""
    import torch
    import torch.nn.functional as F
    g = lambda k: sd[e + k].detach().cpu().double()
    few, feb = g(".feature_expansion.weight"), g(".feature_expansion.bias")
    W0, b0, W2, b2 = g(".edge_feature_mlp.0.weight"), g(".edge_feature_mlp.0.bias"), g(".edge_feature_mlp.2.weight"), g(".edge_feature_mlp.2.bias")
    C0, c0, C2, c2 = g(".combination_mlp.0.weight"), g(".combination_mlp.0.bias"), g(".combination_mlp.2.weight"), g(".combination_mlp.2.bias")
    emb = g(".bond_emb.weight")[int(typ)]

    def fn(d):
        d = torch.from_numpy(np.ascontiguousarray(d, dtype=np.float64)).view(-1, 1)
        x = F.gelu(F.linear(d, few, feb))
        b = emb.expand(d.shape[0], -1)
        h = F.linear(F.gelu(F.linear(torch.cat([x, b], 1), W0, b0)), W2, b2)
        return F.linear(F.gelu(F.linear(torch.cat([h, b], 1), C0, c0)), C2, c2)
    return fn


def encoder_activation_max(sd, cfg, types=(0, 1, 2, 3, 12, 23, 24), far=10.0):
    """Largest magnitude any activation of the MLP edge encoder (edge.py:84-103) and of the CFConv filter networks behind it
    (schnet.py:169-179) takes as an MFMA operand, over lengths in [0, far x cutoff] and the common edge types: these depend on the
    length and the type alone, so their range is a property of the checkpoint and is checked when it is packed (the split-fp16 mode
    saturates operands at 65504; the state-dependent activations are flagged by the kernels: agdiff_ws_t.range_rows).  float64."""
    import torch
    import torch.nn.functional as F
    e = "edge_encoder_global"
    g = lambda k: sd[k].detach().cpu().double()
    few, feb = g(e + ".feature_expansion.weight"), g(e + ".feature_expansion.bias")
    W0, b0, W2, b2 = g(e + ".edge_feature_mlp.0.weight"), g(e + ".edge_feature_mlp.0.bias"), g(e + ".edge_feature_mlp.2.weight"), g(e + ".edge_feature_mlp.2.bias")
    C0, c0, C2, c2 = g(e + ".combination_mlp.0.weight"), g(e + ".combination_mlp.0.bias"), g(e + ".combination_mlp.2.weight"), g(e + ".combination_mlp.2.bias")
    d = torch.linspace(0.0, float(far) * float(cfg.cutoff), 1025, dtype=torch.float64).view(-1, 1)
    x = F.gelu(F.linear(d, few, feb))
    worst = float(x.abs().max())
    LN2 = float(np.log(2.0))
    for typ in types:
        emb = g(e + ".bond_emb.weight")[int(typ)].expand(d.shape[0], -1)
        h1 = F.gelu(F.linear(torch.cat([x, emb], 1), W0, b0))
        h2 = F.linear(h1, W2, b2)
        h3 = F.gelu(F.linear(torch.cat([h2, emb], 1), C0, c0))
        attr = F.linear(h3, C2, c2)
        worst = max(worst, float(h1.abs().max()), float(h3.abs().max()), float(attr.abs().max()))
        for k in range(cfg.num_convs):
            for conv in ("conv1", "conv2"):
                p = "encoder_global.interactions.%d.%s" % (k, conv)
                sp = F.softplus(g(p + ".nn.1.beta") * F.linear(attr, g(p + ".nn.0.weight"), g(p + ".nn.0.bias"))) - LN2
                worst = max(worst, float(sp.abs().max()))
    return worst


def _poly_targets(sd, cfg, typ, with_head):
    """name -> function d[M] -> [M, out] (float64 numpy) of everything the kernels replace by a polynomial for edges of
    type `typ`: the two CFConv filters of every block (schnet.py:169-179) and, for radius edges, the edge_attr half of the
    global head's first layer (common.py:106-109)."""
    import torch
    import torch.nn.functional as F
    attr = _edge_attr_fn(sd, "edge_encoder_global", typ)
    g = lambda k: sd[k].detach().cpu().double()
    LN2 = float(np.log(2.0))
    fns = {}
    for k in range(cfg.num_convs):
        convs = []          # (nn.0 weight, bias, beta, nn.2 weight, bias) of conv1 / conv2, copied to the host once
        for conv in ("conv1", "conv2"):
            p = "encoder_global.interactions.%d.%s" % (k, conv)
            convs.append((g(p + ".nn.0.weight"), g(p + ".nn.0.bias"), g(p + ".nn.1.beta"), g(p + ".nn.2.weight"), g(p + ".nn.2.bias")))

        def filt(d, convs=convs):
            a = attr(d)
            outs = []
            for W0, b0, beta, W2, b2 in convs:
                s = F.softplus(beta * F.linear(a, W0, b0)) - LN2                           # schnet.py:71-80
                outs.append(F.linear(s, W2, b2))
            return torch.cat(outs, 1).numpy()
        fns["conv%d.filt_poly_pk" % k] = filt
    if with_head:
        Wb = g("grad_global_dist_mlp.layers.0.weight")[:, H:]
        fns["head_global.attr_poly_pk"] = lambda d: F.linear(attr(d), Wb).numpy()
    else:               # local types: edge_attr itself (the GIN layers and the local head read it, agdiff_local_edge_rows)
        fns["edge_attr_poly_pk"] = lambda d: attr(d).numpy()
    return fns


_FIT_CACHE = {}      # (weights digest, cutoff, type, kt, with_head) -> fit_type result; a few MB at most


def _poly_weights_digest(sd, cfg):
    """Digest of every tensor the fitted functions depend on (edge encoder, filter networks, the head's first layer)."""
    import hashlib
    h = hashlib.sha1()
    for k in sorted(sd):
        if k.startswith("edge_encoder_global.") or ".nn." in k and k.startswith("encoder_global.") or \
                k == "grad_global_dist_mlp.layers.0.weight":
            h.update(k.encode())
            h.update(np.ascontiguousarray(sd[k].detach().cpu().numpy()).tobytes())
    return h.hexdigest()


def fit_type(sd, cfg, typ, kt, with_head, digest=None):
    """({name: coefficient matrix in natural packed column order}, worst relative error) for edges of type `typ`.
    Results are cached per process by the digest of the weights they depend on."""
    key = (digest or _poly_weights_digest(sd, cfg), float(cfg.cutoff), int(cfg.num_convs), int(typ), int(kt), bool(with_head))
    if key not in _FIT_CACHE:
        if len(_FIT_CACHE) > 256:
            _FIT_CACHE.clear()
        _FIT_CACHE[key] = _fit_type(sd, cfg, typ, kt, with_head)
    return _FIT_CACHE[key]


def _fit_type(sd, cfg, typ, kt, with_head):
    order = poly_feature_order(kt)
    mats, worst = {}, 0.0
    for name, fn in _poly_targets(sd, cfg, typ, with_head).items():
        c, err = fit_poly(fn, float(cfg.cutoff), 32 * kt)
        mats[name] = c[:, order]
        worst = max(worst, err)
    return mats, worst


def radius_polynomials(sd, cfg, tol=POLY_TOL, max_kt=POLY_MAX_KT, min_kt=1):
    """(poly_kt, {name: coefficient matrix}, errors) or (0, {}, errors) when no degree up to 32 max_kt - 1 meets `tol`
    (or the edge encoder is not the MLP one).  min_kt > 1 skips the shorter expansions (tests of the two-k-tile kernels)."""
    if cfg.edge_encoder != "mlp":
        return 0, {}, {}
    errors = {}
    digest = _poly_weights_digest(sd, cfg)
    for kt in range(min_kt, max_kt + 1):
        mats, worst = fit_type(sd, cfg, 0, kt, True, digest)
        errors[kt] = worst
        if worst <= tol:
            return kt, mats, errors
    return 0, {}, errors


# Passes of the split arithmetic over a filter polynomial's terms (include/agdiff_hip.h: agdiff_params_t.poly_plan).  The lo
# parts of the split operands buy 2^-21 per product; a term whose coefficient is small against the filter's size does not
# need them.  The Chebyshev coefficients of an analytic function decay geometrically, so the HIGH half of the accepted
# expansion (f >= 16 at 32 terms, f >= 32 at 64) usually carries 1e-4 .. 1e-7 of the weight, and one hi x hi pass over it
# costs POLY_EPS1 times that -- the two operand roundings of a single product: fp16 has 11 significant bits, so features rounded
# toward zero (v_cvt_pkrtz) are off by < 2^-10 and coefficients rounded to nearest by <= 2^-11; bf16 has 8: 2^-8 + 2^-8
# (the earlier constants were a factor 2 too small).  Plan 1 is taken only when fit error + that bound stays within
# POLY_TOL for the sets the plan is decided on (PackedParams.poly_pass_plan).
POLY_EPS1 = {1: 2.0 ** -7, 2: 1.5 * 2.0 ** -10}
# edge types every batch brings (bonds of order 1..3 and aromatic ones, 2- and 3-hop edges at edge_order 3): the pass plan and
# the coefficient scale are decided on the radius set + these ONCE per model, so that a process's numerics do not depend on
# which batches it has seen; a later type that does not fit that decision keeps the filter MLPs for its edges
POLY_PLAN_TYPES = (1, 2, 3, 12, 23, 24)
POLY_COEF_TARGET = 128.0        # split-fp16: coefficients are stored times 2^S with max |c| 2^S in (64, 128] over the radius set
POLY_COEF_LIMIT = 32768.0       # ... and a typed set must stay below this at the same S


def poly_high_weight(c_nat, kt, high=None):
    """max over outputs of sum_{f >= high} |c[out][f]| relative to the largest value the polynomials take on [0, cutoff]
    (|phi_f| <= 1): c_nat [out, 32 kt] in natural packed column order (fit_type).  `high`: the first term plan 1 gives one pass
    (default: 16 at one k-tile, 32 else); 64 / 96 for plans 2 / 3."""
    K = 32 * kt
    c_f = np.empty_like(np.asarray(c_nat, dtype=np.float64))
    c_f[:, poly_feature_order(kt)] = c_nat
    scale = np.abs(poly_features(np.linspace(-1.0, 1.0, 1025), K) @ c_f.T).max()
    if high is None:
        high = 16 if kt == 1 else 32          # (the terms plan 1 gives one pass: f >= 16 at one k-tile, every k-tile but the first else)
    return float(np.abs(c_f[:, high:]).sum(1).max() / max(scale, 1e-300))


def mix_units(packed, kt):
    """pack_blocks output (split modes, pk [OT][kt]) -> the layout of agdiff_params_t.poly_plan 1 at kt 1: unit 1 of every
    block = [unit 0 of lanes 0..31 | the LO elements of lanes 0..31] (lane 32 + l holds lo of lane l: same row, quarter - 2).
    kt 2 keeps the layout (the kernel does not read unit 1 of the k-tile-1 blocks)."""
    if kt != 1:
        return packed
    b = np.ascontiguousarray(packed).view(np.uint16).reshape(-1, 2, 64, 8).copy()
    hi, lo = b[:, 0].copy(), b[:, 1].copy()
    b[:, 1, :32] = hi[:, :32]
    b[:, 1, 32:] = lo[:, :32]
    return b.reshape(-1).view(np.float32)


def dist_segments(w1, b1, w2, b2):
    """DistanceWeightingNetwork before its sigmoid, layer2(relu(layer1(d))) (schnet.py:83-100), as a piecewise linear
    function of d (agdiff_conv_params_t.dist_seg): kinks bp[32] ascending (+inf padded), alpha[33], beta[33], 2 unused.
    Segment s = number of kinks <= d; its line is the sum over the hidden units active there, in float64."""
    w1, b1, w2 = (np.asarray(x, dtype=np.float64) for x in (w1, b1, w2))
    H_ = w1.shape[0]
    assert H_ == 32
    nz = w1 != 0
    t = np.sort(-b1[nz] / w1[nz])
    bp = np.full(32, np.inf)
    bp[:t.size] = t
    alpha, beta = np.zeros(33), np.zeros(33)
    for s_ in range(33):
        if s_ > t.size:                      # (segments beyond the last kink repeat the last one: never selected)
            alpha[s_], beta[s_] = alpha[t.size], beta[t.size]
            continue
        lo = t[s_ - 1] if s_ > 0 else None
        hi = t[s_] if s_ < t.size else None
        if lo is None and hi is None:
            x = 0.0
        elif lo is None:
            x = hi - 1.0
        elif hi is None:
            x = lo + 1.0
        else:
            x = 0.5 * (lo + hi)
        act = (w1 * x + b1) > 0              # (on an empty segment lo == hi any choice gives the same value at its one point)
        alpha[s_] = float((w2 * w1)[act].sum())
        beta[s_] = float((w2 * b1)[act].sum()) + float(b2)
    return np.concatenate([bp, alpha, beta, np.zeros(2)])


def dist_union_table(segs, cutoff):
    """(agdiff_params_t.dist_union, dist_union_kinks K, dist_union_segments S) from the per-conv tables (dist_segments, two per
    InteractionBlock, rounded to float32 as the kernels read them): the union of all kinks that lie in (0, cutoff], ascending, in
    [0..K-1] (K a power of two > their number, +inf padded) and, for union segment u = number of those kinks <= d (S = their
    number + 1 segments), the line every conv's own table selects for a d in [0, cutoff] of that segment -- conv kinks <= d are
    the conv's kinks <= 0 plus its kinks among the first u union kinks -- at [K + (u * n + cc) * 2], cc = 2 k + (0 | 1),
    n = 2 num_convs.  Kinks outside (0, cutoff] cannot separate two lengths in [0, cutoff]; beyond the cutoff every scale is
    multiplied by an envelope of exactly 0 (schnet.py:140-146), so any finite line serves there.  (The table sits in the LDS of
    every workgroup of agdiff_sampler_front: 17 KiB instead of 39 for default-initialised networks, i.e. twice the workgroups
    per CU.)"""
    tabs = [np.asarray(t, dtype=np.float64).astype(np.float32).reshape(2, 100) for t in segs]
    rows = [tabs[k][h] for k in range(len(tabs)) for h in (0, 1)]
    n = len(rows)
    rc = np.float32(cutoff)
    allk = np.concatenate([r[:32][np.isfinite(r[:32])] for r in rows]) if n else np.zeros(0, np.float32)
    kinks = np.unique(allk[(allk > 0) & (allk <= rc)])
    U = int(kinks.size)
    assert U <= 384
    K = 2
    while K <= U:
        K *= 2
    S = U + 1
    out = np.zeros(K + S * 2 * n, dtype=np.float32)
    out[:K] = np.inf
    out[:U] = kinks
    for u in range(S):
        upto = np.float32(0.0) if u == 0 else kinks[u - 1]
        for cc, r in enumerate(rows):
            s_ = int(np.count_nonzero(r[:32] <= upto))
            out[K + (u * n + cc) * 2] = r[32 + s_]
            out[K + (u * n + cc) * 2 + 1] = r[65 + s_]
    return out, K, S


PRECISIONS = {"f32": 0, "bf16x3": 1, "f16x3": 2}
# config.mlp_act (models/common.py:62-66: getattr(F, name)) -> agdiff_head_params_t.act (include/agdiff_hip.h AGDIFF_ACT_*)
HEAD_ACTS = {"relu": 0, "gelu": 1, "silu": 2, "tanh": 3, "sigmoid": 4, "softplus": 5, "leaky_relu": 6, "elu": 7, "celu": 7, "relu6": 8,
             "hardtanh": 9, "selu": 10, "mish": 11, "hardswish": 12, "hardsigmoid": 13, "softsign": 14, "logsigmoid": 15, "hardshrink": 16,
             "softshrink": 17, "rrelu": 18}
LOCAL_PRECISIONS = {"f32": 0, "bf16x3": 1, "f16x3": 2}      # agdiff_params_t.precision_local
EDGE_ENCODERS = {"mlp": 0, "gaussian": 1}       # agdiff_params_t.edge_encoder


class PackedParams:
    """Owns the device copies of all packed weights and the agdiff_params_t that points at them."""

    def __init__(self, sd, cfg, device, precision="f32", radius_poly="auto", refuse_types=(), precision_local=None,
                 poly_passes="auto", attr_far=True):
        """radius_poly: "auto" -- radius edges take their filters from d-polynomials when the fit is accepted
        (radius_polynomials above), and so do the local edges, per type (ensure_local_types); "off" -- every edge goes
        through the encoder + filter MLPs; "radius" -- polynomials for the radius edges only; "kt2" / "kt3" / "kt4" -- as
        "auto" but starting at 64 / 96 / 128 terms (the variants exist for tests and A/B runs).
        poly_passes: "auto" -- one pass over the high terms of the filter polynomials when their coefficients allow it
        (poly_pass_plan); "full" -- three passes for every term; "from64" / "from96" -- as "auto" without plan 1 / plans 1, 2 (tests and
        A/B runs of plans 2 and 3)."""
        import torch
        self.device = device
        self.attr_far = bool(attr_far)  # False keeps the encoder MLP for local edges beyond the cutoff (tests, A/B runs)
        if poly_passes not in ("auto", "full", "from64", "from96"):
            raise ValueError("poly_passes must be 'auto', 'full', 'from64' or 'from96'")
        self.poly_passes = poly_passes
        if radius_poly not in ("auto", "off", "radius", "kt2", "kt3", "kt4"):
            raise ValueError("radius_poly must be one of 'auto', 'off', 'radius', 'kt2', 'kt3', 'kt4'")
        self.poly_kt, self._poly, self.poly_errors = (0, {}, {}) if radius_poly == "off" else \
            radius_polynomials(sd, cfg, min_kt=int(radius_poly[2]) if radius_poly.startswith("kt") else 1)
        # local edge types with filter polynomials (ensure_local_types): type -> slot, grown as batches bring new types
        self._sd, self._cfg, self._mode = sd, cfg, PRECISIONS.get(precision, 0)
        self.local_slots, self._typed_mats = {}, {}
        self._typed_ok = self.poly_kt >= 1 and radius_poly != "radius"
        # local edge types without a slot (fit refused / slots full): filter MLPs for them.  `refuse_types` seeds the set
        # (tests of the mixed path: a type is treated as if its fit had missed POLY_TOL)
        self.poly_refused_types = set(int(t) for t in refuse_types)
        self.typed_flat = self.slot_table = None
        if precision not in PRECISIONS:
            raise ValueError("precision must be one of %s" % (list(PRECISIONS),))
        self.precision = precision
        mode = PRECISIONS[precision]
        # the local branch's MFMA kernels (GIN layers, local head, local edge_attr rows by polynomial): split-fp16 next to a
        # split-bf16 global branch by default (same rate, 11 + 11 instead of 8 + 8 mantissa bits per operand)
        if precision_local is None:
            precision_local = "f16x3" if precision in ("bf16x3", "f16x3") else precision
        if precision_local not in LOCAL_PRECISIONS:
            raise ValueError("precision_local must be one of %s" % (list(LOCAL_PRECISIONS),))
        self.precision_local = precision_local
        self._mode_local = lmode = LOCAL_PRECISIONS[precision_local]
        _pack = globals()["pack_blocks"]
        # what split-fp16 costs the matrices of each branch (split_fp16_error): worst normwise error, clipping, which matrix
        self.split_fp16_report = {"global": {"err": 0.0, "clipped": False, "worst": None, "matrices": 0},
                                  "local": {"err": 0.0, "clipped": False, "worst": None, "matrices": 0}}

        def note(branch, W):
            e, c = split_fp16_error(W)
            r = self.split_fp16_report[branch]
            r["matrices"] += 1
            r["clipped"] = r["clipped"] or c
            if e > r["err"]:
                r["err"], r["worst"] = e, "matrix #%d %s, max |w| = %.3g" % (r["matrices"], tuple(np.shape(W)), float(np.abs(W).max()))

        def pack_local(W, kouter=False):
            if lmode == 2:
                note("local", W)
            return _pack(W, kouter=kouter, mode=lmode)
        self._pack_local = pack_local

        def pack_blocks(W, kouter=False):      # every matrix of this model is packed in the chosen mode
            if mode == 2:
                note("global", W)
            return _pack(W, kouter=kouter, mode=mode)

        if cfg.hidden_dim != H:
            raise NotImplementedError("hidden_dim must be 128 (InteractionBlock.lin is Linear(256, hidden), schnet.py:190)")
        if cfg.edge_encoder not in EDGE_ENCODERS:
            raise NotImplementedError("Unknown edge encoder: %s" % cfg.edge_encoder)
        if cfg.mlp_act not in HEAD_ACTS:
            raise NotImplementedError("mlp_act=%s: the HIP heads implement %s (models/common.py:62-66 takes any torch.nn.functional "
                                      "name; configs/*.yml:8: relu)" % (cfg.mlp_act, ", ".join(sorted(HEAD_ACTS))))
        if cfg.num_convs > _lib.DEFINES["AGDIFF_MAX_CONVS"] or cfg.num_convs_local > _lib.DEFINES["AGDIFF_MAX_CONVS_LOCAL"]:
            raise NotImplementedError("too many conv layers for this build")
        arrays = {}      # name -> np.float32 array
        scalars = {}

        # ---------------- edge encoder (edge.py:84-103)
        e = "edge_encoder_global"
        scalars["ge_coeff"] = 0.0
        if cfg.edge_encoder == "gaussian":     # edge.py:17-42; schnet.py:18-27
            off = _np(sd, e + ".rbf.offset")
            arrays["ge_offset"] = off
            arrays["ge_emb"] = _np(sd, e + ".bond_emb.weight")          # [100,64]
            if off.shape != (H // 2,) or arrays["ge_emb"].shape != (100, H // 2):
                raise ValueError("gaussian edge encoder: expected 64 offsets and a [100,64] bond_emb")
            scalars["ge_coeff"] = -0.5 / float(off[1] - off[0]) ** 2    # (offset[1]-offset[0]).item() ** 2
        else:
            self._pack_mlp_edge_encoder(sd, e, arrays, pack_blocks)
            if mode == 2 or lmode == 2:       # (activations that depend on the length and the type alone: their range is the checkpoint's;
                # the local branch's kernels read the encoder's rows as operands too)
                key = (_poly_weights_digest(sd, cfg), float(cfg.cutoff), int(cfg.num_convs), "activation_max")
                if key not in _FIT_CACHE:
                    _FIT_CACHE[key] = encoder_activation_max(sd, cfg)
                amax = _FIT_CACHE[key]
                self.encoder_activation_max = amax
                if not amax < SPLIT_FP16_ACT_LIMIT:
                    for branch, on in (("global", mode == 2), ("local", lmode == 2)):
                        if on:
                            self.split_fp16_report[branch]["clipped"] = True
                            self.split_fp16_report[branch]["activations"] = amax
        arrays["schnet_emb"] = _np(sd, "encoder_global.embedding.weight")
        arrays["gin_emb"] = _np(sd, "encoder_local.node_emb.weight")
        self._pack_rest(sd, cfg, device, mode, arrays, scalars, pack_blocks)

    @staticmethod
    def _pack_mlp_edge_encoder(sd, e, arrays, pack_blocks):
        emb = _np(sd, e + ".bond_emb.weight")                           # [100,128]
        W0, b0 = _np(sd, e + ".edge_feature_mlp.0.weight"), _np(sd, e + ".edge_feature_mlp.0.bias")
        W2, b2 = _np(sd, e + ".edge_feature_mlp.2.weight"), _np(sd, e + ".edge_feature_mlp.2.bias")
        C0, c0 = _np(sd, e + ".combination_mlp.0.weight"), _np(sd, e + ".combination_mlp.0.bias")
        C2, c2 = _np(sd, e + ".combination_mlp.2.weight"), _np(sd, e + ".combination_mlp.2.bias")
        arrays["ee_fe_w"] = _np(sd, e + ".feature_expansion.weight")[:, 0]
        arrays["ee_fe_b"] = _np(sd, e + ".feature_expansion.bias")
        arrays["ee_t1"] = emb @ W0[:, H:].T + b0[None, :]
        arrays["ee_w1_pk"] = pack_blocks(W0[:, :H])
        arrays["ee_t3"] = emb @ C0[:, H:].T + (c0 + C0[:, :H] @ b2)[None, :]
        arrays["ee_w23_pk"] = pack_blocks(C0[:, :H] @ W2)
        arrays["ee_w4_pk"] = pack_blocks(C2)
        arrays["ee_b4"] = c2

    def _pack_rest(self, sd, cfg, device, mode, arrays, scalars, pack_blocks):
        import torch
        # ---------------- SchNet blocks (schnet.py:113-234)
        for k in range(cfg.num_convs):
            p = "encoder_global.interactions.%d" % k
            c1, c2_ = p + ".conv1", p + ".conv2"
            n = "conv%d." % k
            # ShiftedSoftplus in base 2, constants folded into the linear layers (include/agdiff_hip.h)
            LN2 = np.log(2.0)
            k1 = float(_np(sd, c1 + ".nn.1.beta")) / LN2
            k2 = float(_np(sd, c2_ + ".nn.1.beta")) / LN2
            f64 = lambda key: _np(sd, key).astype(np.float64)
            arrays[n + "filt_w1_pk"] = pack_blocks(
                np.concatenate([k1 * f64(c1 + ".nn.0.weight"), k2 * f64(c2_ + ".nn.0.weight")], 0), kouter=True)
            arrays[n + "filt_b1"] = np.concatenate([k1 * f64(c1 + ".nn.0.bias"), k2 * f64(c2_ + ".nn.0.bias")])
            arrays[n + "filt_w2a_pk"] = pack_blocks(LN2 * f64(c1 + ".nn.2.weight"))
            arrays[n + "filt_w2b_pk"] = pack_blocks(LN2 * f64(c2_ + ".nn.2.weight"))
            arrays[n + "filt_b2"] = np.concatenate([f64(c1 + ".nn.2.bias") - LN2 * f64(c1 + ".nn.2.weight").sum(1),
                                                    f64(c2_ + ".nn.2.bias") - LN2 * f64(c2_ + ".nn.2.weight").sum(1)])
            arrays[n + "dist_seg"] = np.concatenate([dist_segments(_np(sd, c + ".distance_weighting.layer1.weight")[:, 0],
                                                                   _np(sd, c + ".distance_weighting.layer1.bias"),
                                                                   _np(sd, c + ".distance_weighting.layer2.weight")[0],
                                                                   float(_np(sd, c + ".distance_weighting.layer2.bias")[0]))
                                                     for c in (c1, c2_)])
            W1a, b1a = fold_bn(_np(sd, c1 + ".lin1.weight"), _np(sd, c1 + ".lin1.bias"), sd, c1 + ".norm1")
            W1b, b1b = fold_bn(_np(sd, c2_ + ".lin1.weight"), _np(sd, c2_ + ".lin1.bias"), sd, c2_ + ".norm1")
            arrays[n + "lin1_pk"] = pack_blocks(np.concatenate([W1a, W1b], 0))
            arrays[n + "lin1_b"] = np.concatenate([b1a, b1b])
            W2a, b2a = fold_bn(_np(sd, c1 + ".lin2.weight"), _np(sd, c1 + ".lin2.bias"), sd, c1 + ".norm2")
            W2b, b2b = fold_bn(_np(sd, c2_ + ".lin2.weight"), _np(sd, c2_ + ".lin2.bias"), sd, c2_ + ".norm2")
            # InteractionBlock.act (ShiftedSoftplus, schnet.py:71-80,206) in base 2 like the filter networks' one: with
            # kb = act.beta * log2(e) the (BN-folded) lin2 layers yield u = kb (W2 agg + b2); the kernel forms
            # s = max(u, log2(1 + 2^u)); softplus(beta x) - ln 2 = ln 2 (s - 1), so `lin` takes ln2 * W and b - ln2 * W 1
            kb = float(_np(sd, p + ".act.beta")) / LN2
            Wl, bl = f64(p + ".lin.weight"), f64(p + ".lin.bias")
            arrays[n + "lin2a_pk"] = pack_blocks(kb * W2a.astype(np.float64), kouter=True)
            arrays[n + "lin2b_pk"] = pack_blocks(kb * W2b.astype(np.float64), kouter=True)
            arrays[n + "lin2_b"] = kb * np.concatenate([b2a, b2b]).astype(np.float64)
            arrays[n + "lin_pk"] = pack_blocks(LN2 * Wl)
            arrays[n + "lin_b"] = bl - LN2 * Wl.sum(1)
            arrays[n + "gate1_pk"] = pack_blocks(_np(sd, p + ".attention.0.weight"))
            arrays[n + "gate1_b"] = _np(sd, p + ".attention.0.bias")
            arrays[n + "gate2_w"] = _np(sd, p + ".attention.2.weight")[0]
            scalars[n + "gate2_b"] = float(_np(sd, p + ".attention.2.bias")[0])
            scalars[n + "act_beta"] = float(_np(sd, p + ".act.beta"))
            s = "encoder_global.scaling_modules.%d" % k
            arrays[n + "scale1_pk"] = pack_blocks(_np(sd, s + ".fc.0.weight"))
            arrays[n + "scale2_pk"] = pack_blocks(_np(sd, s + ".fc.2.weight"))

        # the DistanceWeightingNetworks of all CFConvs over their common segments (agdiff_params_t.dist_union)
        arrays["dist_union"], self._union_kinks, self._union_segments = dist_union_table(
            [arrays["conv%d.dist_seg" % k] for k in range(cfg.num_convs)], float(cfg.cutoff))

        # ---------------- GIN (gin.py:38-69, 112-148)
        for k in range(cfg.num_convs_local):
            p = "encoder_local.convs.%d" % k
            n = "gin%d." % k
            W1, b1 = _np(sd, p + ".nn.layers.0.weight"), _np(sd, p + ".nn.layers.0.bias")
            W, b = fold_bn(_np(sd, p + ".nn.layers.1.weight"), _np(sd, p + ".nn.layers.1.bias"), sd,
                           "encoder_local.batch_norms.%d" % k)
            # relu(n x) = n relu(x) for n > 0: the hidden layer may carry any power of two n (W1, b1 times n; W2 by n), exact in
            # fp32 and free for the kernel.  n balances the two matrices' magnitudes, so that neither's split-fp16 lo parts sit in
            # fp16's subnormals (split_fp16_error) and the hidden activations stay in fp16's range when a checkpoint's layers
            # are far from O(1); the layer's output keeps its scale (it is added to h, gin.py:143-147).
            nb = 2.0 ** int(np.round(0.5 * (np.log2(pow2_norm(W1)) - np.log2(pow2_norm(W))))) if self._mode_local != 0 else 1.0
            arrays[n + "w1_pk"] = self._pack_local(W1.astype(np.float64) * nb)
            arrays[n + "b1"] = b1.astype(np.float64) * nb
            arrays[n + "w2_pk"] = self._pack_local(W.astype(np.float64) / nb)
            arrays[n + "b2"] = b
            scalars[n + "one_plus_eps"] = 1.0 + float(_np(sd, p + ".eps")[0])

        # ---------------- heads (common.py:44-103; dualenc.py:88-98)
        for name, p in (("head_global", "grad_global_dist_mlp"), ("head_local", "grad_local_dist_mlp")):
            n = name + "."
            pb = self._pack_local if name == "head_local" else pack_blocks
            W0, W1 = _np(sd, p + ".layers.0.weight").astype(np.float64), _np(sd, p + ".layers.1.weight").astype(np.float64)
            # The heads are ReLU chains (common.py:62-66 with mlp_act = relu) ending in an fp32 dot product: every hidden layer may
            # carry a power of two (relu(n x) = n relu(x)), which the last layer's fp32 weights take out again.  Both matrices
            # are normalised to max |w| in (0.5, 1] -- exact, free for the kernels -- so that a checkpoint whose head weights
            # are ~1e-3 (lo parts in fp16's subnormals) or ~1e+2 (hidden activations beyond fp16's range) runs split-fp16
            # at its nominal accuracy.  Other activations are not homogeneous: no normalisation (n = 1).
            n0, n1 = (pow2_norm(W0), pow2_norm(W1)) if (cfg.mlp_act in ("relu", "leaky_relu") and (self._mode_local if name == "head_local" else mode) != 0) else (1.0, 1.0)
            self.head_norm = getattr(self, "head_norm", {})
            self.head_norm[name] = (n0, n1)
            arrays[n + "w1_pk"] = pb(W0 * n0, kouter=True)
            arrays[n + "b1"] = _np(sd, p + ".layers.0.bias").astype(np.float64) * n0
            arrays[n + "w2_pk"] = pb(W1 * n1)
            arrays[n + "b2"] = _np(sd, p + ".layers.1.bias").astype(np.float64) * (n0 * n1)
            arrays[n + "w3"] = _np(sd, p + ".layers.2.weight")[0].astype(np.float64) / (n0 * n1)
            scalars[n + "b3"] = float(_np(sd, p + ".layers.2.bias")[0])

        # ---------------- radius-edge polynomials (include/agdiff_hip.h: agdiff_params_t.poly_kt)
        for name, c in self._poly.items():
            if name.startswith("head_"):        # (the CFConv filter sets are packed by _pack_filter_sets: their layout
                # depends on the pass plan); the edge_attr half of the global head's first layer carries that layer's n0
                arrays[name] = pack_blocks(np.asarray(c, dtype=np.float64) * self.head_norm["head_global"][0], kouter=True)

        # one flat device buffer, every section 256-byte aligned
        offs, total = {}, 0
        for k, v in arrays.items():
            v = np.asarray(v)
            if v.dtype != np.float32:          # packed matrices are float32 already (bf16 bit patterns in mode 1)
                v = v.astype(np.float64).astype(np.float32)
            v = np.ascontiguousarray(v.reshape(-1))
            arrays[k] = v
            offs[k] = total
            total += (v.size + 63) // 64 * 64
        flat = np.zeros(total, dtype=np.float32)
        for k, v in arrays.items():
            flat[offs[k]:offs[k] + v.size] = v
        self.offsets = offs
        self.sizes = {k: v.size for k, v in arrays.items()}
        self.flat = torch.from_numpy(flat).to(device)
        base = self.flat.data_ptr()

        def P(name):
            return ctypes.c_void_p(base + 4 * offs[name])

        prm = _lib.Params()
        for f in ("ee_fe_w", "ee_fe_b", "ee_t1", "ee_w1_pk", "ee_t3", "ee_w23_pk", "ee_w4_pk", "ee_b4",
                  "ge_offset", "ge_emb", "schnet_emb", "gin_emb", "dist_union"):
            if f in offs:
                setattr(prm, f, P(f))
        prm.dist_union_kinks, prm.dist_union_segments = self._union_kinks, self._union_segments
        prm.edge_encoder = EDGE_ENCODERS[cfg.edge_encoder]
        prm.ge_coeff = scalars["ge_coeff"]
        for k in range(cfg.num_convs):
            cp, n = prm.conv[k], "conv%d." % k
            for f, _ in _lib.ConvParams._fields_:
                if (n + f) in offs:
                    setattr(cp, f, P(n + f))
                elif f not in ("filt_poly_pk", "filt_poly_typed_pk", "filt_poly_unscale", "pad0"):   # (set by _pack_filter_sets / null)
                    setattr(cp, f, scalars[n + f])
        for k in range(cfg.num_convs_local):
            gp, n = prm.gin[k], "gin%d." % k
            for f in ("w1_pk", "b1", "w2_pk", "b2"):
                setattr(gp, f, P(n + f))
            gp.one_plus_eps = scalars[n + "one_plus_eps"]
            gp.relu_out = 1 if k < cfg.num_convs_local - 1 else 0
        for name in ("head_global", "head_local"):
            hp, n = getattr(prm, name), name + "."
            for f in ("w1_pk", "b1", "w2_pk", "b2", "w3"):
                setattr(hp, f, P(n + f))
            if (n + "attr_poly_pk") in offs:
                hp.attr_poly_pk = P(n + "attr_poly_pk")
            hp.b3 = scalars[n + "b3"]
            hp.act = HEAD_ACTS[cfg.mlp_act]
            hp.precision = self._mode_local if name == "head_local" else mode
        prm.num_convs = cfg.num_convs
        prm.num_convs_local = cfg.num_convs_local
        prm.cutoff = float(cfg.cutoff)
        prm.smooth = 1 if cfg.smooth_conv else 0
        prm.precision = mode
        prm.precision_local = self._mode_local
        prm.poly_kt = self.poly_kt
        prm.poly_num_slots = 0
        self.struct = prm
        self.poly_plan, self.poly_high_bound, self.rad_poly_flat = 0, {}, None
        self._plan_fixed, self.filt_poly_upscale = None, None
        for k in range(cfg.num_convs):
            prm.conv[k].filt_poly_unscale = 1.0
        if self.poly_kt >= 1:
            self._pack_filter_sets()

    def _bound(self, mats, plan):
        """POLY_EPS1 x the weight of the terms `plan` gives one pass (1: f >= 16 at one k-tile, f >= 32 else; 2: f >= 64; 3: f >= 96)."""
        high = 32 * plan if plan >= 2 else None
        return POLY_EPS1[self._mode] * max(poly_high_weight(mats["conv%d.filt_poly_pk" % k], self.poly_kt, high)
                                           for k in range(self._cfg.num_convs))

    def _set_bound(self, name, mats):
        """The one-pass bound of a set under the model's plan (kept in poly_high_bound)."""
        if name not in self.poly_high_bound:
            self.poly_high_bound[name] = self._bound(mats, max(self.poly_pass_plan(), 1))
        return self.poly_high_bound[name]

    def poly_pass_plan(self):
        """agdiff_params_t.poly_plan, decided ONCE per model: the first of 1, 2, 3 (p < poly_kt: one pass from k-tile p on) for which the radius
        set and each of POLY_PLAN_TYPES whose fit is accepted at all satisfy fit error + POLY_EPS1[mode] * (weight of the terms that
        plan gives one pass) <= POLY_TOL, else 0; self.poly_high_bound records the second summand per set under the plan taken (under
        plan 1 when none is).  (Until round 4 the plan followed the types met so far and could flip mid-process.)"""
        if self._plan_fixed is not None:
            return self._plan_fixed
        plan = 0
        if self._mode in POLY_EPS1 and self.poly_kt >= 1 and self.poly_passes != "full":
            first = None
            cands = tuple(range(1, max(self.poly_kt, 2)))            # (plan p: one pass from k-tile p on -- p < poly_kt)
            for cand in cands[{"from64": 1, "from96": 2}.get(self.poly_passes, 0):]:
                bounds = {"radius": self._bound(self._poly, cand)}
                ok = self.poly_errors[self.poly_kt] + bounds["radius"] <= POLY_TOL
                if ok and self._typed_ok:
                    for t in POLY_PLAN_TYPES:
                        if t in self.poly_refused_types:
                            continue
                        mats, err = fit_type(self._sd, self._cfg, t, self.poly_kt, False)
                        if err <= POLY_TOL:
                            bounds["type%d" % t] = self._bound(mats, cand)
                            ok = ok and err + bounds["type%d" % t] <= POLY_TOL
                first = first or bounds
                if ok:
                    plan, first = cand, bounds
                    break
            self.poly_high_bound.update(first or {})
        self._plan_fixed = plan
        return plan

    def _type_fits_plan(self, t, mats, err):
        """Whether a local type's accepted fit can join the sets already packed: under plan 1 its own one-pass bound must hold,
        and in split-fp16 its coefficients must stay in range at the model's scale 2^S."""
        if self.poly_pass_plan() >= 1 and err + self._set_bound("type%d" % t, mats) > POLY_TOL:
            return False
        if self._mode == 2:
            for k in range(self._cfg.num_convs):
                if np.abs(mats["conv%d.filt_poly_pk" % k]).max() * self.filt_poly_upscale[k] > POLY_COEF_LIMIT:
                    return False
        return True

    def _pack_filter_sets(self):
        """(Re)pack the CFConv filter polynomials -- the radius edges' set and the slotted local types' -- and point the conv
        structs at the new buffers (the old ones may still be in use by enqueued launches: torch's allocator keeps them alive
        in stream order).  Layout (pass plan) and scale are the model's, fixed at the first call: a new slot adds a set, it
        never changes the others' bits."""
        import torch
        prm, nc, kt = self.struct, self._cfg.num_convs, self.poly_kt
        plan = self.poly_pass_plan()

        by_slot = sorted(self.local_slots, key=self.local_slots.get)
        # split-fp16: coefficients times 2^S per conv (conv[k].filt_poly_unscale = 2^-S): their lo parts leave fp16's subnormal
        # range, whose quantum of 6e-8 otherwise costs up to 1e-6 of a filter of size 0.2.  S comes from the RADIUS set alone
        # (max |c| 2^S in (64, 128]: typed sets up to 256 x larger still fit, _type_fits_plan).  bf16 parts have fp32's exponent
        # range and fp32 operands no parts: S = 0
        if self.filt_poly_upscale is None:
            up = []
            for k in range(nc):
                cmax = np.abs(self._poly["conv%d.filt_poly_pk" % k]).max()
                S = int(np.clip(np.floor(np.log2(POLY_COEF_TARGET / max(cmax, 1e-30))), 0, 24)) if self._mode == 2 else 0
                up.append(2.0 ** S)
            self.filt_poly_upscale = up
        up = self.filt_poly_upscale
        for k in range(nc):
            prm.conv[k].filt_poly_unscale = 1.0 / up[k]

        def pk(c, k):
            v = pack_blocks(np.asarray(c, dtype=np.float64) * up[k], mode=self._mode)
            return mix_units(v, kt) if plan == 1 else v
        if self.rad_poly_flat is None:
            rad = [pk(self._poly["conv%d.filt_poly_pk" % k], k) for k in range(nc)]
            self.rad_poly_flat = torch.from_numpy(np.concatenate(rad)).to(self.device)
            for k in range(nc):
                prm.conv[k].filt_poly_pk = ctypes.c_void_p(self.rad_poly_flat.data_ptr() + 4 * rad[0].size * k)
        if by_slot:
            per_conv = [np.concatenate([pk(self._typed_mats[t]["conv%d.filt_poly_pk" % k], k) for t in by_slot]) for k in range(nc)]
            self.typed_flat = torch.from_numpy(np.concatenate(per_conv)).to(self.device)
            for k in range(nc):
                prm.conv[k].filt_poly_typed_pk = ctypes.c_void_p(self.typed_flat.data_ptr() + 4 * per_conv[0].size * k)
        self.poly_plan = prm.poly_plan = plan

    def ensure_local_types(self, types):
        """Give every local edge type of a batch (BatchTopology.local_types) a filter-polynomial slot, so that
        agdiff_cfconv_node needs no edge_attr for the local edges either (include/agdiff_hip.h: poly_num_slots).  Slots are
        kept in order of first appearance, the types of one batch by expected row count (the first ones that fit stay
        LDS-resident in the kernel, later ones are read from L2), fitted with the radius edges' number of terms (poly_kt).  A type whose fit misses POLY_TOL, or a type beyond
        AGDIFF_POLY_MAX_SLOTS, gets no slot: ITS edges go through the filter MLPs (agdiff_cfconv_local, in a "mixed" batch
        next to agdiff_cfconv_node's polynomial tiles for the slotted types); everything else keeps its polynomials.
        Returns True when every type of `types` has a slot."""
        import torch
        if not self._typed_ok:
            return False
        new = [int(t) for t in types if int(t) not in self.local_slots and int(t) not in self.poly_refused_types]
        if not new:
            return not any(int(t) in self.poly_refused_types for t in types)
        # (the kernel keeps the FIRST slots' sets in LDS -- five at 32 terms -- and reads later ones from L2: the types with the most
        # rows first -- 3-hop, 2-hop, single, aromatic, double bonds -- so that with all six of GEOM-Drugs' types in one batch it is
        # the triple bonds' set that comes from L2, not the 3-hop edges')
        rank = {24: 0, 23: 1, 1: 2, 12: 3, 2: 4}
        new.sort(key=lambda t: (rank.get(t, 5), t))
        prm = self.struct
        max_slots = _lib.DEFINES["AGDIFF_POLY_MAX_SLOTS"]
        kt = self.poly_kt
        added = False
        for t in new:
            mats, err = fit_type(self._sd, self._cfg, t, kt, False)
            self.poly_errors["type%d" % t] = err
            if err > POLY_TOL or len(self.local_slots) >= max_slots or not self._type_fits_plan(t, mats, err):
                # THIS type keeps the filter MLPs (agdiff_cfconv_local in a mixed batch); the slotted ones keep their polynomials
                self.poly_refused_types.add(t)
                continue
            self.local_slots[t] = len(self.local_slots)
            self._typed_mats[t] = mats
            added = True
        if not added:
            return not any(int(t) in self.poly_refused_types for t in types)
        by_slot = sorted(self.local_slots, key=self.local_slots.get)
        self._pack_filter_sets()           # (the typed sets are packed again with the new one; the radius set, the plan and 2^S stay)
        # edge_attr itself per type (agdiff_local_edge_rows): pk [8][kt] -- the kernel takes one k-tile, so only with kt == 1
        if kt == 1:
            sets = [pack_blocks(self._typed_mats[t]["edge_attr_poly_pk"], mode=self._mode_local) for t in by_slot]
            # far sets (edge_attr on [cutoff, 10 cutoff]): as many as the kernel's LDS holds next to the near sets, the types
            # that carry most local edges first (3-hop, 2-hop, single bonds, then the rest); a type whose far fit misses
            # POLY_TOL keeps the encoder MLP beyond the cutoff
            far_table = np.full(100, -1, dtype=np.int32)
            room = max(0, ATTR_POLY_MAX_SETS - len(by_slot)) if self.attr_far else 0
            prio = {24: 0, 23: 1, 1: 2}
            for t in sorted(by_slot, key=lambda ty: (prio.get(ty, 3), ty)):
                if room <= 0:
                    break
                c, err = fit_attr_far(self._sd, self._cfg, t)
                self.poly_errors["type%d_far" % t] = err
                if err > POLY_TOL:
                    continue
                far_table[t] = len(sets)
                sets.append(pack_blocks(c, mode=self._mode_local))
                room -= 1
            attr_sets = np.concatenate(sets)
            self.typed_attr_flat = torch.from_numpy(attr_sets).to(self.device)
            prm.attr_poly_typed_pk = ctypes.c_void_p(self.typed_attr_flat.data_ptr())
            self.attr_far_table = torch.from_numpy(far_table).to(self.device)
            prm.attr_poly_far_set = ctypes.c_void_p(self.attr_far_table.data_ptr())
            prm.attr_poly_far_slots = int((far_table >= 0).sum())
            prm.attr_poly_far_hi = ATTR_FAR_FACTOR * float(self._cfg.cutoff)
        table = np.full(100, -1, dtype=np.int32)
        for t, sl in self.local_slots.items():
            table[t] = sl
        self.slot_table = torch.from_numpy(table).to(self.device)
        prm.poly_type_slot = ctypes.c_void_p(self.slot_table.data_ptr())
        prm.poly_num_slots = len(by_slot)
        sm = [0, 0]
        for t in self.local_slots:
            sm[t >> 6] |= 1 << (t & 63)
        for w in (0, 1):
            prm.poly_slot_mask[w] = sm[w] - (1 << 64) if sm[w] >= (1 << 63) else sm[w]
        return not any(int(t) in self.poly_refused_types for t in types)

    TUNING = ("share_rows_min_nodes", "node_ldsw_min_tiles", "node_split_max_tiles", "serial_branches", "local_poly_off",
              "attr_poly_off", "poly_lds_sets", "cfconv_four_min_quads", "cfconv_quad_tiles")

    def set_tuning(self, **kw):
        """Kernel-variant thresholds (include/agdiff_hip.h: agdiff_params_t.tune_*; 0 = library default)."""
        for k, v in kw.items():
            if k not in self.TUNING:
                raise KeyError("unknown tuning field %r (known: %s)" % (k, ", ".join(self.TUNING)))
            setattr(self.struct, "tune_" + k, int(v))

    def view(self, name):
        o = self.offsets[name]
        return self.flat[o:o + self.sizes[name]]

    def update_schnet_embedding(self, weight):
        """Re-upload encoder_global.embedding after the max_norm renorm touched it (schnet.py:254)."""
        self.view("schnet_emb").copy_(weight.detach().reshape(-1).to(self.flat.dtype))
