"""Weight pre-packing: reference state_dict (SURVEY.md §8b key layout) -> the device buffers and the
agdiff_params_t struct the HIP kernels read (include/agdiff_hip.h).

Output-preserving transformations applied here (each checked against the oracle in tests/):
  * eval-mode BatchNorm folded into the adjacent Linear (schnet.py:153-158, gin.py:131-132);
  * MLPEdgeEncoder: the bond-embedding halves of the two 256->128 layers become per-edge-type
    tables; edge_feature_mlp.2 is folded into combination_mlp.0; the size-1 softmax attention
    (== 1.0) is dropped (edge.py:84-103);
  * conv1/conv2 first filter layers and lin1 layers are fused along the output dimension.
All folding is done in float64 and rounded once to float32.
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


def fold_bn(W, b, sd, p, eps=1e-5):
    """Linear(W, b) followed by eval BatchNorm1d `p` -> one Linear."""
    s = _np(sd, p + ".weight") / np.sqrt(_np(sd, p + ".running_var") + eps)
    return W * s[:, None], (b - _np(sd, p + ".running_mean")) * s + _np(sd, p + ".bias")


PRECISIONS = {"f32": 0, "bf16x3": 1}
EDGE_ENCODERS = {"mlp": 0, "gaussian": 1}       # agdiff_params_t.edge_encoder


class PackedParams:
    """Owns the device copies of all packed weights and the agdiff_params_t that points at them."""

    def __init__(self, sd, cfg, device, precision="f32"):
        import torch
        self.device = device
        if precision not in PRECISIONS:
            raise ValueError("precision must be one of %s" % (list(PRECISIONS),))
        self.precision = precision
        mode = PRECISIONS[precision]
        _pack = globals()["pack_blocks"]

        def pack_blocks(W, kouter=False):      # every matrix of this model is packed in the chosen mode
            return _pack(W, kouter=kouter, mode=mode)

        if cfg.hidden_dim != H:
            raise NotImplementedError("hidden_dim must be 128 (InteractionBlock.lin is Linear(256, hidden), schnet.py:190)")
        if cfg.edge_encoder not in EDGE_ENCODERS:
            raise NotImplementedError("Unknown edge encoder: %s" % cfg.edge_encoder)
        if cfg.mlp_act != "relu":
            raise NotImplementedError("mlp_act=%s (HIP heads implement relu, configs/*.yml:8)" % cfg.mlp_act)
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
            dws = []
            for c in (c1, c2_):
                d = c + ".distance_weighting"
                dws.append(np.concatenate([_np(sd, d + ".layer1.weight")[:, 0], _np(sd, d + ".layer1.bias"),
                                           _np(sd, d + ".layer2.weight")[0], _np(sd, d + ".layer2.bias")]))
            arrays[n + "dist_w"] = np.concatenate(dws)
            W1a, b1a = fold_bn(_np(sd, c1 + ".lin1.weight"), _np(sd, c1 + ".lin1.bias"), sd, c1 + ".norm1")
            W1b, b1b = fold_bn(_np(sd, c2_ + ".lin1.weight"), _np(sd, c2_ + ".lin1.bias"), sd, c2_ + ".norm1")
            arrays[n + "lin1_pk"] = pack_blocks(np.concatenate([W1a, W1b], 0))
            arrays[n + "lin1_b"] = np.concatenate([b1a, b1b])
            W2a, b2a = fold_bn(_np(sd, c1 + ".lin2.weight"), _np(sd, c1 + ".lin2.bias"), sd, c1 + ".norm2")
            W2b, b2b = fold_bn(_np(sd, c2_ + ".lin2.weight"), _np(sd, c2_ + ".lin2.bias"), sd, c2_ + ".norm2")
            arrays[n + "lin2a_pk"] = pack_blocks(W2a, kouter=True)
            arrays[n + "lin2b_pk"] = pack_blocks(W2b, kouter=True)
            arrays[n + "lin2_b"] = np.concatenate([b2a, b2b])
            arrays[n + "lin_pk"] = pack_blocks(_np(sd, p + ".lin.weight"))
            arrays[n + "lin_b"] = _np(sd, p + ".lin.bias")
            arrays[n + "gate1_pk"] = pack_blocks(_np(sd, p + ".attention.0.weight"))
            arrays[n + "gate1_b"] = _np(sd, p + ".attention.0.bias")
            arrays[n + "gate2_w"] = _np(sd, p + ".attention.2.weight")[0]
            scalars[n + "gate2_b"] = float(_np(sd, p + ".attention.2.bias")[0])
            scalars[n + "act_beta"] = float(_np(sd, p + ".act.beta"))
            s = "encoder_global.scaling_modules.%d" % k
            arrays[n + "scale1_pk"] = pack_blocks(_np(sd, s + ".fc.0.weight"))
            arrays[n + "scale2_pk"] = pack_blocks(_np(sd, s + ".fc.2.weight"))

        # ---------------- GIN (gin.py:38-69, 112-148)
        for k in range(cfg.num_convs_local):
            p = "encoder_local.convs.%d" % k
            n = "gin%d." % k
            arrays[n + "w1_pk"] = pack_blocks(_np(sd, p + ".nn.layers.0.weight"))
            arrays[n + "b1"] = _np(sd, p + ".nn.layers.0.bias")
            W, b = fold_bn(_np(sd, p + ".nn.layers.1.weight"), _np(sd, p + ".nn.layers.1.bias"), sd,
                           "encoder_local.batch_norms.%d" % k)
            arrays[n + "w2_pk"] = pack_blocks(W)
            arrays[n + "b2"] = b
            scalars[n + "one_plus_eps"] = 1.0 + float(_np(sd, p + ".eps")[0])

        # ---------------- heads (common.py:44-103; dualenc.py:88-98)
        for name, p in (("head_global", "grad_global_dist_mlp"), ("head_local", "grad_local_dist_mlp")):
            n = name + "."
            arrays[n + "w1_pk"] = pack_blocks(_np(sd, p + ".layers.0.weight"), kouter=True)
            arrays[n + "b1"] = _np(sd, p + ".layers.0.bias")
            arrays[n + "w2_pk"] = pack_blocks(_np(sd, p + ".layers.1.weight"))
            arrays[n + "b2"] = _np(sd, p + ".layers.1.bias")
            arrays[n + "w3"] = _np(sd, p + ".layers.2.weight")[0]
            scalars[n + "b3"] = float(_np(sd, p + ".layers.2.bias")[0])

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
                  "ge_offset", "ge_emb", "schnet_emb", "gin_emb"):
            if f in offs:
                setattr(prm, f, P(f))
        prm.edge_encoder = EDGE_ENCODERS[cfg.edge_encoder]
        prm.ge_coeff = scalars["ge_coeff"]
        for k in range(cfg.num_convs):
            cp, n = prm.conv[k], "conv%d." % k
            for f, _ in _lib.ConvParams._fields_:
                if (n + f) in offs:
                    setattr(cp, f, P(n + f))
                else:
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
            hp.b3 = scalars[n + "b3"]
            hp.act = 0
            hp.precision = mode
        prm.num_convs = cfg.num_convs
        prm.num_convs_local = cfg.num_convs_local
        prm.cutoff = float(cfg.cutoff)
        prm.smooth = 1 if cfg.smooth_conv else 0
        prm.precision = mode
        self.struct = prm

    def view(self, name):
        o = self.offsets[name]
        return self.flat[o:o + self.sizes[name]]

    def update_schnet_embedding(self, weight):
        """Re-upload encoder_global.embedding after the max_norm renorm touched it (schnet.py:254)."""
        self.view("schnet_emb").copy_(weight.detach().reshape(-1).to(self.flat.dtype))
