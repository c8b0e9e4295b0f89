"""Host-side mirror of the reference's module API for the diffusion-sampling hot path.

`get_model(config)` / `DualEncoderEpsNetwork` keep the reference's constructor argument, module
tree, 854-key state_dict layout, `forward`, `langevin_dynamics_sample` and
`langevin_dynamics_sample_diffusion` signatures (epsnet/__init__.py:7-11, epsnet/dualenc.py:54-547),
so `scripts/test.py:111-164` runs against it unchanged.  The sub-modules below are parameter
containers only: all arithmetic happens in libagdiff_hip.so (hand-written gfx950 kernels) through
the C ABI of include/agdiff_hip.h.  There is no CPU / PyTorch fallback: calling the network
without a GPU and the built library raises.
"""
import ctypes
import math
import os

import numpy as np
import torch
from torch import nn

from . import _lib
from .packing import SPLIT_FP16_MAX_ERR, PackedParams
from .topology import BatchTopology, Workspace, batch_fingerprint

H_FIXED = 128


# ----------------------------------------------------------------------------------- schedule
def get_beta_schedule(beta_schedule, *, beta_start, beta_end, num_diffusion_timesteps):
    """epsnet/dualenc.py:21-51 (float64; the sigmoid branch's linspace has no dtype -> float64)."""
    T = num_diffusion_timesteps
    if beta_schedule == "quad":
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, T, dtype=np.float64) ** 2
    elif beta_schedule == "linear":
        betas = np.linspace(beta_start, beta_end, T, dtype=np.float64)
    elif beta_schedule == "const":
        betas = beta_end * np.ones(T, dtype=np.float64)
    elif beta_schedule == "jsd":
        betas = 1.0 / np.linspace(T, 1, T, dtype=np.float64)
    elif beta_schedule == "sigmoid":
        betas = np.linspace(-6, 6, T)
        betas = 1 / (np.exp(-betas) + 1) * (beta_end - beta_start) + beta_start
    else:
        raise NotImplementedError(beta_schedule)
    assert betas.shape == (T,)
    return betas


# ----------------------------------------------------------------------------------- containers
class _ShiftedSoftplus(nn.Module):          # schnet.py:71-75
    def __init__(self):
        super().__init__()
        self.beta = nn.Parameter(torch.tensor(1.0))


class _FilterAttention(nn.Module):          # schnet.py:103-106 (constructed, never called)
    def __init__(self, f):
        super().__init__()
        self.attention_weights = nn.Parameter(torch.randn(f))


class _DistanceWeighting(nn.Module):        # schnet.py:83-88
    def __init__(self, hidden=32):
        super().__init__()
        self.layer1 = nn.Linear(1, hidden)
        self.layer2 = nn.Linear(hidden, 1)


class _CFConv(nn.Module):                   # schnet.py:113-134
    def __init__(self, in_ch, out_ch, nf, filt):
        super().__init__()
        self.lin1 = nn.Linear(in_ch, nf, bias=True)
        self.norm1 = nn.BatchNorm1d(nf)
        self.lin2 = nn.Linear(nf, out_ch)
        self.norm2 = nn.BatchNorm1d(out_ch)
        self.nn = filt
        self.attention = _FilterAttention(nf)
        self.distance_weighting = _DistanceWeighting(32)
        nn.init.xavier_uniform_(self.lin1.weight)
        self.lin1.bias.data.fill_(0)
        nn.init.xavier_uniform_(self.lin2.weight)
        self.lin2.bias.data.fill_(0)


class _InteractionBlock(nn.Module):         # schnet.py:165-199
    def __init__(self, hidden, edge_ch, nf):
        super().__init__()
        mlp1 = nn.Sequential(nn.Linear(edge_ch, nf), _ShiftedSoftplus(), nn.Linear(nf, nf))
        mlp2 = nn.Sequential(nn.Linear(edge_ch, nf // 2), _ShiftedSoftplus(), nn.Linear(nf // 2, nf // 2))
        self.conv1 = _CFConv(hidden, hidden, nf, mlp1)
        self.conv2 = _CFConv(hidden, hidden, nf // 2, mlp2)
        self.act = _ShiftedSoftplus()
        self.lin = nn.Linear(256, hidden)
        self.attention = nn.Sequential(nn.Linear(hidden, hidden // 2), nn.ReLU(inplace=True),
                                       nn.Linear(hidden // 2, 1), nn.Sigmoid())


class _AdaptiveScaling(nn.Module):          # schnet.py:219-228
    def __init__(self, ch, reduction=16):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(ch, ch // reduction, bias=False), nn.ReLU(inplace=True),
                                nn.Linear(ch // reduction, ch, bias=False), nn.Sigmoid())


class _SchNetEncoder(nn.Module):            # schnet.py:237-266
    def __init__(self, hidden, nf, n_inter, edge_ch):
        super().__init__()
        self.embedding = nn.Embedding(100, hidden, max_norm=10.0)
        self.interactions = nn.ModuleList([_InteractionBlock(hidden, edge_ch, nf) for _ in range(n_inter)])
        self.scaling_modules = nn.ModuleList([_AdaptiveScaling(hidden) for _ in range(n_inter)])


class _GaussianSmearing(nn.Module):         # schnet.py:18-27
    def __init__(self, start=0.0, stop=5.0, num_gaussians=50):
        super().__init__()
        offset = torch.linspace(start, stop, num_gaussians)
        self.coeff = -0.5 / (offset[1] - offset[0]).item() ** 2
        self.register_buffer("offset", offset)


class _GaussianSmearingEdgeEncoder(nn.Module):     # edge.py:17-42
    def __init__(self, num_gaussians=64, cutoff=10.0):
        super().__init__()
        self.num_gaussians = num_gaussians
        self.cutoff = cutoff
        self.rbf = _GaussianSmearing(start=0.0, stop=cutoff * 2, num_gaussians=num_gaussians)
        self.bond_emb = nn.Embedding(100, embedding_dim=num_gaussians)

    @property
    def out_channels(self):
        return self.num_gaussians * 2


class _MLPEdgeEncoder(nn.Module):           # edge.py:45-82
    def __init__(self, hidden):
        super().__init__()
        self.hidden_dim = hidden
        self.bond_emb = nn.Embedding(100, embedding_dim=hidden)
        self.feature_expansion = nn.Linear(1, hidden)
        self.edge_feature_mlp = nn.Sequential(nn.Linear(hidden * 2, hidden), nn.GELU(), nn.Linear(hidden, hidden))
        self.combination_mlp = nn.Sequential(nn.Linear(hidden * 2, hidden), nn.GELU(), nn.Linear(hidden, hidden))
        self.attention = nn.Sequential(nn.Linear(hidden, hidden), nn.Tanh(), nn.Linear(hidden, 1), nn.Softmax(dim=1))

    @property
    def out_channels(self):
        return self.hidden_dim


class _MLP(nn.Module):                      # common.py:44-84
    def __init__(self, input_dim, hidden_dims):
        super().__init__()
        dims = [input_dim] + list(hidden_dims)
        self.layers = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])
        self.attention_layers = nn.ModuleList()


class _GINEConv(nn.Module):                 # gin.py:14-36
    def __init__(self, mlp):
        super().__init__()
        self.nn = mlp
        self.register_buffer("eps", torch.Tensor([0.0]))


class _GINEncoder(nn.Module):               # gin.py:75-110
    def __init__(self, hidden, num_convs):
        super().__init__()
        self.node_emb = nn.Embedding(100, hidden)
        self.convs = nn.ModuleList([_GINEConv(_MLP(hidden, [hidden, hidden])) for _ in range(num_convs)])
        self.batch_norms = nn.ModuleList([nn.BatchNorm1d(hidden) for _ in range(num_convs)])


def get_edge_encoder(cfg):                  # edge.py:106-116
    if cfg.edge_encoder == "mlp":
        return _MLPEdgeEncoder(cfg.hidden_dim)
    elif cfg.edge_encoder == "gaussian":
        # (the reference as shipped raises NameError here -- edge.py:24 uses GaussianSmearing without importing
        # it from schnet.py; this is the encoder it would build with that import in place)
        return _GaussianSmearingEdgeEncoder(num_gaussians=cfg.hidden_dim // 2, cutoff=cfg.cutoff)
    else:
        raise NotImplementedError(f"Unknown edge encoder: {cfg.edge_encoder}")


# ----------------------------------------------------------------------------------- the network
class DualEncoderEpsNetwork(nn.Module):
    """Drop-in for agdiff.models.epsnet.dualenc.DualEncoderEpsNetwork (sampling path only)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        if config.hidden_dim != H_FIXED:
            raise NotImplementedError("hidden_dim must be 128: InteractionBlock.lin = Linear(256, hidden) (schnet.py:190)")
        self.edge_encoder_global = get_edge_encoder(config)
        self.edge_encoder_local = get_edge_encoder(config)
        self.encoder_global = _SchNetEncoder(config.hidden_dim, config.hidden_dim, config.num_convs,
                                             self.edge_encoder_global.out_channels)
        self.encoder_local = _GINEncoder(config.hidden_dim, config.num_convs_local)
        self.grad_global_dist_mlp = _MLP(2 * config.hidden_dim, [config.hidden_dim, config.hidden_dim // 2, 1])
        self.grad_local_dist_mlp = _MLP(2 * config.hidden_dim, [config.hidden_dim, config.hidden_dim // 2, 1])
        self.model_global = nn.ModuleList([self.edge_encoder_global, self.encoder_global, self.grad_global_dist_mlp])
        self.model_local = nn.ModuleList([self.edge_encoder_local, self.encoder_local, self.grad_local_dist_mlp])
        self.model_type = config.type
        if self.model_type == "diffusion":
            betas = get_beta_schedule(beta_schedule=config.beta_schedule, beta_start=config.beta_start,
                                      beta_end=config.beta_end,
                                      num_diffusion_timesteps=config.num_diffusion_timesteps)
            betas = torch.from_numpy(betas).float()
            self.betas = nn.Parameter(betas, requires_grad=False)
            alphas = (1.0 - betas).cumprod(dim=0)
            self.alphas = nn.Parameter(alphas, requires_grad=False)
            self.num_timesteps = self.betas.size(0)
        elif self.model_type == "dsm":
            # denoising score matching (dualenc.py:127-140): the noise levels as a parameter; forward() does not depend on the type,
            # and the reference's samplers handle 'diffusion' only (langevin_dynamics_sample returns None for 'dsm', dualenc.py:418)
            sigmas = torch.tensor(np.exp(np.linspace(np.log(config.sigma_begin), np.log(config.sigma_end), config.num_noise_level)),
                                  dtype=torch.float32)
            self.sigmas = nn.Parameter(sigmas, requires_grad=False)
            self.num_timesteps = self.sigmas.size(0)
        # (any other type: the reference builds the module without a schedule, dualenc.py:110-140; so does this)
        # arithmetic of the HIP kernels: "f32" = exact fp32 MFMA; "bf16x3" = split-bf16 MFMA (hi+lo operands,
        # three passes, fp32 accumulation, ~2^-16 relative per product).  Both meet the 1e-4 parity bar.
        # (config field or attribute; nothing is read from the environment)
        # "f16x3" (default) = the same three-pass scheme on split-fp16 operands (v_mfma_f32_16x16x32_f16, same rate): 11 + 11
        # instead of 8 + 8 mantissa bits per operand -- measured ~10 x closer to the reference than "bf16x3" on every
        # fixture (tests/helpers.py gates); operands beyond fp16's range (65504) saturate, bf16x3 keeps fp32's range.
        self.precision = getattr(config, "precision", None) or "f16x3"
        # "auto": radius edges take their CFConv filters / head inputs from d-polynomials when packing.py accepts the
        # fit for these weights (<= 1e-6 of the networks they replace), "off": every edge through the MLPs
        self.radius_poly = getattr(config, "radius_poly", None) or "auto"
        # arithmetic of the local branch's MFMA kernels: None = "f16x3" (split-fp16) next to a split-bf16 global branch, else
        # the global mode; "bf16x3" / "f32" / "f16x3" force one (packing.LOCAL_PRECISIONS)
        self.precision_local = getattr(config, "precision_local", None)
        # "auto": the high terms of the filter polynomials take one MFMA pass instead of three when the host can bound what
        # that costs by the fit tolerance (packing.PackedParams.poly_pass_plan); "full": three passes for every term
        self.poly_passes = getattr(config, "poly_passes", None) or "auto"
        # kernel-variant thresholds (PackedParams.set_tuning; include/agdiff_hip.h: agdiff_params_t.tune_*), e.g.
        # model.tuning["node_ldsw_min_tiles"] = 1 -- tests reach every variant on small fixtures this way
        self.tuning = {}
        self.group_targets = None            # targets per wave of agdiff_cfconv_node: None = by batch size (BatchTopology.GROUP_MIN_NODES)
        self.poly_refuse_types = ()          # (tests: local edge types to treat as if their polynomial fit had been refused)
        # local edges LONGER than the cutoff take their edge_attr rows from a second polynomial per type on [cutoff, 10 cutoff]
        # (packing.fit_attr_far) instead of the encoder MLP; False keeps the MLP for them
        self.attr_far_rows = True
        # The denoising loop's steps as replayed HIP graphs (include/agdiff_hip.h: agdiff_step_graph_capture): True, False, or "auto" (for
        # batches of up to STEP_GRAPH_MAX_NODES atoms).  Bit-identical to the launch-by-launch loop (tests/test_hip_step_graphs.py) and
        # NOT faster on this stack: one molecule x 100 conformers 0.327 / 0.335 ms per step replayed against 0.318 / 0.317 launched,
        # alanine dipeptide 0.35 against 0.28 (profiles/r06_step_graphs.txt) -- a small batch's step is a chain of ~14 dependent
        # dispatches at ~20 us each on the GPU side, which a replay issues one by one just the same; the host was never the bound.
        # Hence off by default.
        self.step_graphs = False
        self._packed = None
        self._packed_key = None
        self._batch_cache = None

    # ------------------------------------------------------------------ plumbing
    def _device(self):
        return self.betas.device

    def _require_gpu(self):
        dev = self._device()
        if dev.type != "cuda":
            raise _lib.AgdiffHipError(
                "agdiff_amd.DualEncoderEpsNetwork computes on an MI355X only (module is on %s); "
                "there is no CPU fallback -- call .to('cuda')" % dev)
        return _lib.load()

    def _weights_key(self):
        return (str(self._device()), self.precision, self.radius_poly, tuple(self.poly_refuse_types), getattr(self, "precision_local", None),
                getattr(self, "poly_passes", "auto"), getattr(self, "attr_far_rows", True)) + tuple(int(p._version) for p in self.state_dict(keep_vars=True).values())

    def packed(self):
        """Packed device weights, rebuilt when any parameter / buffer changed (load_state_dict, .to).  A branch asked to run
        in split-fp16 whose matrices that mode cannot hold -- packing.split_fp16_error above SPLIT_FP16_MAX_ERR (weights far
        below O(1): lo parts in fp16's subnormals) or a value beyond 65504 -- is packed in split-bf16 instead (fp32's exponent
        range, same speed, ~2^-16 per product), with a warning; `effective_precision` / `effective_precision_local` say what
        runs.  (The ReLU chains -- both heads, the GIN MLPs -- are normalised by exact powers of two first, packing.pow2_norm.)"""
        key = self._weights_key()
        if self._packed is None or self._packed_key != key:
            sd = {k: v for k, v in self.state_dict().items()}
            prec, prec_l = self.precision, getattr(self, "precision_local", None)
            for _ in range(3):
                pk = PackedParams(sd, self.config, self._device(), prec, self.radius_poly,
                                  refuse_types=self.poly_refuse_types, precision_local=prec_l,
                                  poly_passes=getattr(self, "poly_passes", "auto"), attr_far=getattr(self, "attr_far_rows", True))
                rep, again = pk.split_fp16_report, False
                for branch in ("global", "local"):
                    r = rep[branch]
                    if r["err"] > SPLIT_FP16_MAX_ERR or r["clipped"]:
                        import warnings
                        why = ("activations of the edge encoder / filter networks reach %.3g" % r["activations"]) if r.get("activations") else \
                            "%s: normwise error %.2g%s" % (r["worst"], r["err"], ", clipped" if r["clipped"] else "")
                        warnings.warn("agdiff_amd: the %s branch's weights do not fit split-fp16 (%s); "
                                      "that branch runs in split-bf16" % (branch, why))
                        if branch == "global":
                            prec, prec_l = "bf16x3", (prec_l or pk.precision_local)
                        else:
                            prec_l = "bf16x3"
                        again = True
                if not again:
                    break
            self._packed, self._packed_key = pk, key
            self.effective_precision, self.effective_precision_local = pk.precision, pk.precision_local
        # (keys of self.tuning that are not library thresholds are host-side choices: "group_radius_column")
        self._packed.set_tuning(**{k: self.tuning.get(k, 0) for k in PackedParams.TUNING})
        return self._packed

    # Range watch of the split-fp16 modes.  fp16 operands saturate at 65504 (a saturated operand gives a finite but wrong
    # product), and the pair heads multiply two node features, so the node tensors every MFMA chain starts from are watched:
    # |h|, |hl| (SchNet / GIN node states) <= 255 keeps h_i * h_j inside the range, |agg|, |xs| <= 60000 themselves.  A trained
    # AGDIFF (max_norm-10 embedding, BatchNorm'd CFConvs and GIN layers) sits two to three orders of magnitude below; what
    # trips the watch is a diverging run or a degenerate checkpoint, for which precision="bf16x3" has fp32's range.
    RANGE_LIMITS = (("h", 255.0), ("hl", 255.0), ("agg", 60000.0), ("xs", 60000.0))
    STEP_GRAPH_MAX_NODES = 32768

    def _uses_split_fp16(self):
        p = self._packed
        prec, prec_l = (p.precision, p.precision_local) if p is not None else (self.precision, getattr(self, "precision_local", None) or "f16x3")
        return prec == "f16x3" or (prec == "bf16x3" and prec_l == "f16x3")

    def range_report(self, ws, batch=None):
        """(name, max |x|, limit, graphs) when a watched node tensor of workspace `ws` has left the split-fp16 range -- or a kernel
        has flagged a node whose hidden activations did, or whose state / CFConv input did when it was stored (ws.range_rows: node
        stage, GIN layers, pair heads; sticky between two polls, cleared here) --, else None: `name` / `max |x|` / `limit` describe the first such tensor, `graphs` = the graphs (ids of `batch` [N])
        that own an offending row of ANY of them (None without `batch`) -- so that one poll quarantines them all.  NaNs are masked
        (they are the NaN flag's business, dualenc.py:539-541: a quarantined molecule must not blind the watch for the others):
        the comparisons below are false for a NaN.  Byte-sized temporaries and one host synchronisation: called where the NaN
        flag is polled."""
        if not self._uses_split_fp16():
            return None
        glob = (self._packed.precision if self._packed is not None else self.precision) == "f16x3"
        names = [n for n, _ in self.RANGE_LIMITS if glob or n == "hl"]
        lims = dict(self.RANGE_LIMITS)
        rows = None if batch is None else int(batch.shape[0])
        over = []
        for n in names:
            x = getattr(ws, n)
            o = (x > lims[n]) | (x < -lims[n])
            over.append(o.view(rows, -1).any(dim=1) if rows else o.any().reshape(1))
        flagged = ws.range_rows != 0
        names, over = names + ["hidden activations"], over + [flagged if rows else flagged.any().reshape(1)]
        hit = torch.stack([o.any() for o in over]).cpu().tolist()
        if not any(hit):
            return None
        n = names[hit.index(True)]
        if hit[-1]:
            ws.range_rows.zero_()
        v = float(torch.nan_to_num(getattr(ws, n), nan=0.0).abs().max()) if n in lims else 65000.0
        graphs = None
        if rows:
            bad_rows = over[0]
            for o in over[1:]:
                bad_rows = bad_rows | o
            graphs = torch.unique(batch[bad_rows]).cpu().tolist()
        return n, v, lims.get(n, 65000.0), graphs

    def check_range(self, ws, batch=None):
        """Raise AgdiffRangeError (with .tensor, .value, .limit, .graphs) when range_report finds a violation."""
        rep = self.range_report(ws, batch)
        if rep is not None:
            n, v, lim, graphs = rep
            what = ("max |%s| = %.3g exceeds %.0f" % (n, v, lim)) if n in dict(self.RANGE_LIMITS) else \
                "a hidden activation of the node stage, the GIN layers or the pair heads reached %.0f" % lim
            e = _lib.AgdiffRangeError(
                "%s: outside the range in which the split-fp16 arithmetic mode is valid (fp16 "
                "operands saturate at 65504; the pair heads multiply two node features) -- set model.precision = 'bf16x3' "
                "(agdiff_amd.driver re-samples the affected molecules that way by itself)" % what)
            e.tensor, e.value, e.limit, e.graphs = n, v, lim, graphs
            raise e

    def arithmetic(self, precision=None, precision_local=None):
        """Context manager: run with another arithmetic mode (the driver's bf16x3 retry of molecules that left the split-fp16
        range); the packed weights of the mode left behind are rebuilt on the next call."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            old = (self.precision, self.precision_local)
            self.precision, self.precision_local = precision or old[0], precision_local if precision_local is not None else old[1]
            try:
                yield self
            finally:
                self.precision, self.precision_local = old
        return cm()

    def _renorm_embedding(self, atom_type):
        # nn.Embedding(max_norm=10) renormalises looked-up rows in place on every forward, also in
        # eval mode (schnet.py:254,271); same ATen op, same side effect on the state_dict.
        # embedding_renorm_ bumps the weight's version counter even when no row changes, so the packed weights are
        # looked up BEFORE it (a validation loop calling forward() per batch must not repack 854 tensors every
        # call); the renormed rows are then pushed into the packed copy and the key refreshed.
        pk = self.packed()
        w = self.encoder_global.embedding.weight
        with torch.no_grad():
            torch.embedding_renorm_(w, atom_type.reshape(-1).long().to(w.device), 10.0, 2.0)
        pk.update_schnet_embedding(w)
        self._packed_key = self._weights_key()
        return pk

    def prepare_topology(self, atom_type, bond_index, bond_type, batch, num_graphs, extend_order=False, device="cpu"):
        """The static topology of a batch (BatchTopology: coalesced local edges, canonical list, quad tiles), built on the
        host: what begin_sampling(..., topology=...) takes instead of building it itself.  No GPU call is made for
        device="cpu", so the driver prepares the NEXT batch in a background thread while this one samples."""
        return BatchTopology(atom_type, bond_index, bond_type, batch, num_graphs=num_graphs,
                             extend_order=extend_order, order=self.config.edge_order, device=device,
                             group_targets=getattr(self, "group_targets", None),
                             radius_column=bool(self.tuning.get("group_radius_column", 1)))

    def _batch(self, atom_type, bond_index, bond_type, batch, num_graphs, extend_order, topology=None):
        """Static topology + workspace of this call's batch (kept on the module afterwards so that
        tests and tools can inspect the device buffers; never reused across calls)."""
        key = None
        if topology is not None:
            if topology.N != int(atom_type.shape[0]):
                raise ValueError("topology was prepared for %d atoms, the batch has %d" % (topology.N, int(atom_type.shape[0])))
            # ... and for THIS batch: same graphs, same bonds, same way of extending them, same grouping (a topology of another
            # batch -- or of another rank's range -- with the same atom count would run with the wrong bonds and quads)
            fp = batch_fingerprint(atom_type, bond_index, bond_type, batch, num_graphs, extend_order)
            if fp != topology.fingerprint:
                raise ValueError("topology was prepared for another batch (fingerprint %r, this batch %r)" % (topology.fingerprint, fp))
            gt = getattr(self, "group_targets", None)
            if gt is not None and int(gt) != topology.group_targets:
                raise ValueError("topology was grouped for %d targets per wave, the model asks for %d" % (topology.group_targets, int(gt)))
            topo = topology.to(self._device())
        else:
            topo = self.prepare_topology(atom_type, bond_index, bond_type, batch, num_graphs, extend_order, device=self._device())
        ws = Workspace(topo)
        if self._packed is not None and topo.L:
            self._packed.ensure_local_types(topo.local_types)     # filter polynomials for this batch's local edge types
        self._batch_cache = (key, topo, ws)
        return topo, ws

    # ------------------------------------------------------------------ forward (dualenc.py:142-251)
    def forward(self, atom_type, pos, bond_index, bond_type, batch, time_step, edge_index=None, edge_type=None,
                edge_length=None, return_edges=False, extend_order=True, extend_radius=True):
        if self.model_type != "diffusion":
            # dualenc.py:184-186,210: sigma_edge is only bound for 'diffusion'; the reference's forward raises UnboundLocalError
            raise NotImplementedError("forward() of a %r model: the reference's own forward fails for this type "
                                      "(sigma_edge undefined, dualenc.py:184-210); only 'diffusion' computes" % (self.model_type,))
        lib = self._require_gpu()
        if edge_index is not None and edge_type is not None and edge_length is not None:
            return self._forward_given_graph(lib, atom_type, pos, batch, edge_index, edge_type, edge_length, return_edges)
        with torch.no_grad():
            pk = self._renorm_embedding(atom_type)
            topo, ws = self._batch(atom_type, bond_index, bond_type, batch, None, extend_order)
            posc = pos.detach().to(torch.float32).contiguous()
            _lib.check(lib.agdiff_score_forward(ctypes.byref(pk.struct), ctypes.byref(topo.struct),
                                                ctypes.byref(ws.struct), _lib.ptr(posc),
                                                self._fwd_flags(topo, extend_radius), _lib.stream_ptr()),
                       "agdiff_score_forward")
            E = int(ws.num_edges.item())
            self.check_range(ws, topo.batch64)
            perm = ws.ref2dst[:E].long()                     # reference (row, col)-sorted order
            inv_g = ws.e_inv_global[:E][perm].unsqueeze(-1)
            inv_l = ws.l_inv[:topo.L].clone().unsqueeze(-1)
            if not return_edges:
                return inv_g, inv_l
            e_index = torch.stack([ws.e_src[:E][perm].long(), ws.e_dst[:E][perm].long()], dim=0)
            e_type = ws.e_type[:E][perm].long()
            e_len = ws.e_len[:E][perm].unsqueeze(-1)
            return inv_g, inv_l, e_index, e_type, e_len, e_type > 0

    @staticmethod
    def _fwd_flags(topo, extend_radius, with_global=True):
        flags = _lib.DEFINES["AGDIFF_FWD_GLOBAL"] if with_global else 0
        if not extend_radius:
            # dualenc.py:166-176 with extend_radius=False leaves the bond list as it was passed; the device graph
            # is always in coalesced (row, col) order, so the two agree only for an already-coalesced list
            if not topo.bond_list_coalesced:
                raise NotImplementedError("extend_radius=False needs a (row, col)-sorted duplicate-free bond list")
            flags |= _lib.DEFINES["AGDIFF_FWD_NO_RADIUS"]
        return flags

    def _forward_given_graph(self, lib, atom_type, pos, batch, edge_index, edge_type, edge_length, return_edges):
        """forward(..., edge_index, edge_type, edge_length) (dualenc.py:165 skips the graph construction): the
        caller's edges, in the caller's order.  The host sorts them by destination once and uploads them -- one
        device-to-host copy and a numpy sort per call (a host synchronisation): this variant is for callers that hold
        an edge list already, not for inner loops; the sampler never takes it."""
        import numpy as np
        dev = self._device()
        ei = edge_index.detach().cpu().numpy().astype(np.int64).reshape(2, -1)
        et = edge_type.detach().cpu().numpy().astype(np.int64).reshape(-1)
        el = edge_length.detach().to(torch.float32).cpu().numpy().reshape(-1)
        E = et.shape[0]
        if ei.shape[1] != E or el.shape[0] != E:
            raise ValueError("edge_index, edge_type and edge_length disagree on the number of edges")
        if E and (et.min() < 0 or et.max() >= 100):
            raise ValueError("edge_type out of the embedding range [0, 100)")
        loc = np.nonzero(et > 0)[0]                                    # is_local_edge, dualenc.py:566-567
        with torch.no_grad():
            pk = self._renorm_embedding(atom_type)
            topo = BatchTopology(atom_type, ei[:, loc], et[loc], batch, num_graphs=None, extend_order=False,
                                 device=dev, group_targets=getattr(self, "group_targets", None))
            if topo.L != loc.shape[0]:
                raise NotImplementedError("duplicate local edges in a caller-supplied edge list")
            ws = Workspace(topo, max_edges=E)
            self._batch_cache = (None, topo, ws)
            order = np.lexsort((ei[0], ei[1]))                          # by destination, sources ascending
            where = np.empty(E, dtype=np.int64)
            where[order] = np.arange(E)
            by_src = np.lexsort((ei[1], ei[0]))
            N = topo.N
            up_i = lambda a: torch.from_numpy(np.ascontiguousarray(a).astype(np.int32)).to(dev)
            ws.e_src[:E].copy_(up_i(ei[0][order])); ws.e_dst[:E].copy_(up_i(ei[1][order]))
            ws.e_type[:E].copy_(up_i(et[order]))
            ws.e_len[:E].copy_(torch.from_numpy(np.ascontiguousarray(el[order])).to(dev))
            ws.in_ptr.copy_(up_i(np.concatenate([[0], np.cumsum(np.bincount(ei[1], minlength=N))])))
            ws.out_ptr.copy_(up_i(np.concatenate([[0], np.cumsum(np.bincount(ei[0], minlength=N))])))
            ws.ref2dst[:E].copy_(up_i(where[by_src]))
            ws.num_edges.fill_(E)
            l_len = np.empty(topo.L, dtype=np.float32)
            l_len[topo.loc_pos_of_input] = el[loc]
            ws.l_len[:topo.L].copy_(torch.from_numpy(l_len).to(dev))
            posc = pos.detach().to(dev, torch.float32).contiguous()
            flags = _lib.DEFINES["AGDIFF_FWD_GLOBAL"] | _lib.DEFINES["AGDIFF_FWD_GRAPH_GIVEN"]
            _lib.check(lib.agdiff_score_forward(ctypes.byref(pk.struct), ctypes.byref(topo.struct),
                                                ctypes.byref(ws.struct), _lib.ptr(posc), flags, _lib.stream_ptr()),
                       "agdiff_score_forward")
            inv_g = ws.e_inv_global[:E][torch.from_numpy(where).to(dev)].unsqueeze(-1)
            inv_l = ws.l_inv[:topo.L][torch.from_numpy(topo.loc_pos_of_input).to(dev)].unsqueeze(-1)
            if not return_edges:
                return inv_g, inv_l
            return inv_g, inv_l, edge_index, edge_type, edge_length, edge_type > 0       # dualenc.py:241-249

    # ------------------------------------------------------------------ loss value (dualenc.py:253-395)
    def get_loss(self, atom_type, pos, bond_index, bond_type, batch, num_nodes_per_graph, num_graphs,
                 anneal_power=2.0, return_unreduced_loss=False, return_unreduced_edge_loss=False,
                 extend_order=True, extend_radius=True, **kwargs):
        if self.model_type == "diffusion":
            return self.get_loss_diffusion(atom_type, pos, bond_index, bond_type, batch, num_nodes_per_graph,
                                           num_graphs, anneal_power, return_unreduced_loss,
                                           return_unreduced_edge_loss, extend_order, extend_radius, **kwargs)

    def get_loss_diffusion(self, atom_type, pos, bond_index, bond_type, batch, num_nodes_per_graph, num_graphs,
                           anneal_power=2.0, return_unreduced_loss=False, return_unreduced_edge_loss=False,
                           extend_order=True, extend_radius=True, *, time_step=None, pos_noise=None):
        """Forward VALUE of the training loss (what scripts/train.py:160-170 `validate` computes under no_grad):
        per-atom `loss [N,1]`, or `(loss, loss_global, loss_local)` with return_unreduced_loss.  No autograd graph
        is built -- backward kernels are out of scope.  Keyword-only `time_step [G]` / `pos_noise
        [N,3]` replace the torch.randint / normal_ draws of dualenc.py:299-311 (parity tests)."""
        lib = self._require_gpu()
        dev = self._device()
        with torch.no_grad():
            if time_step is None:
                ts = torch.randint(0, self.num_timesteps, size=(num_graphs // 2 + 1,), device=dev)
                time_step = torch.cat([ts, self.num_timesteps - ts - 1], dim=0)[:num_graphs]
            time_step = time_step.to(dev).long()
            a = self.alphas.index_select(0, time_step).to(torch.float32).contiguous()          # (G,)
            posc = pos.detach().to(dev, torch.float32).contiguous()
            if pos_noise is None:
                pos_noise = torch.zeros(size=posc.size(), device=dev)
                pos_noise.normal_()
            noise = pos_noise.detach().to(dev, torch.float32).contiguous()
            pk = self._renorm_embedding(atom_type)
            topo, ws = self._batch(atom_type, bond_index, bond_type, batch, num_graphs, extend_order)
            if a.numel() != topo.G:
                raise ValueError("time_step must have one entry per graph")
            pert = torch.empty_like(posc)
            loss = torch.empty(3, topo.N, device=dev, dtype=torch.float32)
            st = _lib.stream_ptr()
            _lib.check(lib.agdiff_perturb_positions(ctypes.byref(topo.struct), _lib.ptr(posc), _lib.ptr(noise),
                                                    _lib.ptr(a), _lib.ptr(pert), st), "agdiff_perturb_positions")
            _lib.check(lib.agdiff_score_forward(ctypes.byref(pk.struct), ctypes.byref(topo.struct),
                                                ctypes.byref(ws.struct), _lib.ptr(pert),
                                                self._fwd_flags(topo, extend_radius), st), "agdiff_score_forward")
            _lib.check(lib.agdiff_diffusion_loss(ctypes.byref(pk.struct), ctypes.byref(topo.struct),
                                                 ctypes.byref(ws.struct), _lib.ptr(posc), _lib.ptr(pert), _lib.ptr(a),
                                                 _lib.ptr(loss), st), "agdiff_diffusion_loss")
            if return_unreduced_edge_loss:
                return None                      # the reference's branch is `pass` and falls off the end (dualenc.py:390-391)
            if return_unreduced_loss:
                return loss[0].unsqueeze(-1), loss[1].unsqueeze(-1), loss[2].unsqueeze(-1)
            return loss[0].unsqueeze(-1)

    # ------------------------------------------------------------------ samplers (dualenc.py:397-547)
    def langevin_dynamics_sample(self, atom_type, pos_init, bond_index, bond_type, batch, num_graphs, extend_order,
                                 extend_radius=True, n_steps=5000, step_lr=0.0000010, clip=1000, clip_local=None,
                                 clip_pos=None, min_sigma=0, global_start_sigma=float("inf"), w_global=0.2,
                                 w_reg=1.0, **kwargs):
        if self.model_type == "diffusion":
            kwargs.setdefault("sampling_type", "ddpm_noisy")
            kwargs.setdefault("eta", 1.0)
            return self.langevin_dynamics_sample_diffusion(
                atom_type, pos_init, bond_index, bond_type, batch, num_graphs, extend_order, extend_radius, n_steps,
                step_lr, clip, clip_local, clip_pos, min_sigma, global_start_sigma, w_global, w_reg, **kwargs)

    def langevin_dynamics_sample_diffusion(self, atom_type, pos_init, bond_index, bond_type, batch, num_graphs,
                                           extend_order, extend_radius=True, n_steps=5000, step_lr=0.0000010,
                                           clip=1000, clip_local=None, clip_pos=None, min_sigma=0,
                                           global_start_sigma=float("inf"), w_global=0.2, w_reg=1.0, **kwargs):
        """Same contract as the reference; extra keyword-only hooks (all optional):
          noise            [n_steps, N, 3] tensor replacing torch.randn_like (parity tests)
          save_traj        False -> return an empty pos_traj and skip the per-step copy
          skip_discarded_global  False -> also run the global encoder on steps whose result the
                           reference discards (sigma >= global_start_sigma, dualenc.py:523-524)
          nan_check_every  host polls the device NaN flag every this many steps (default 64)
          step_indices     explicit list of schedule indices to visit instead of the last n_steps
          on_step          callback(k, i, pos) after each step is enqueued (data-parallel gather)
          raise_on_nan     False -> never raise FloatingPointError; the caller reads LangevinRun.nan_graphs()
                           (graphs are independent, so the others' results stay valid: agdiff_amd/driver.py)
          noise_mode       "chunked" (default): without `noise`, standard normals come from torch.randn on the module's
                           device generator in chunks of 128 steps ([128, N, 3] per call) -- same distribution as the
                           reference's draws but another consumption of the Philox stream;
                           "per_step": one torch.randn_like(pos) per step, exactly the reference's call (dualenc.py:529):
                           with equal seeds and equal generator state the SAME normals as the reference draws on this
                           device, hence seed-for-seed comparable runs (one small launch more per step)
          traj_overlap_min_bytes  the trajectory goes to the host while the run samples when a poll interval's chunk
                           (nan_check_every x N x 12 B) is at least this large (default 16 MiB); else one copy at the end
        """
        run = self.begin_sampling(atom_type, pos_init, bond_index, bond_type, batch, num_graphs, extend_order,
                                  extend_radius, n_steps, step_lr, clip, clip_local, clip_pos, global_start_sigma,
                                  w_global, **kwargs)
        run.advance(run.remaining())
        return run.finish()

    def begin_sampling(self, atom_type, pos_init, bond_index, bond_type, batch, num_graphs, extend_order,
                       extend_radius=True, n_steps=5000, step_lr=0.0000010, clip=1000, clip_local=None,
                       clip_pos=None, global_start_sigma=float("inf"), w_global=0.2, **kwargs):
        """Set up one sampling job (dualenc.py:468-476) and return a LangevinRun that enqueues steps."""
        self._require_gpu()
        return LangevinRun(self, atom_type, pos_init, bond_index, bond_type, batch, num_graphs, extend_order,
                           n_steps, step_lr, clip, clip_local, clip_pos, global_start_sigma, w_global,
                           extend_radius=extend_radius, **kwargs)


class LangevinRun:
    """The denoising loop of dualenc.py:478-545 as an object: `advance(m)` enqueues the next m steps on
    the current stream with no host synchronisation except the periodic NaN-flag poll."""

    def __init__(self, model, atom_type, pos_init, bond_index, bond_type, batch, num_graphs, extend_order,
                 n_steps, step_lr, clip, clip_local, clip_pos, global_start_sigma, w_global, noise=None,
                 save_traj=True, skip_discarded_global=True, nan_check_every=64, step_indices=None, on_step=None,
                 extend_radius=True, raise_on_nan=True, noise_mode="chunked", traj_overlap_min_bytes=16 << 20, topology=None,
                 **_ignored):
        self.model, self.lib = model, _lib.load()
        dev = model._device()
        self.sigmas = ((1.0 - model.alphas).sqrt() / model.alphas.sqrt()).detach().cpu()
        model.eval()
        with torch.no_grad():
            self.pk = model._renorm_embedding(atom_type)
            self.topo, self.ws = model._batch(atom_type, bond_index, bond_type, batch, num_graphs, extend_order, topology=topology)
            self.radius_flags = model._fwd_flags(self.topo, extend_radius, with_global=False)
            self._sampler_flag = _lib.DEFINES["AGDIFF_FWD_SAMPLER"]    # only radius edges' global scores are used (dualenc.py:516-518)
            self._stage0_done = False
            T = model.num_timesteps
            self.steps = list(step_indices) if step_indices is not None else list(reversed(range(T - n_steps, T)))
            self.pos = (pos_init.detach().to(dev, torch.float32) * self.sigmas[-1].to(dev)).contiguous()
        N = self.topo.N
        self.traj = torch.empty((len(self.steps), N, 3), dtype=torch.float32, device=dev) if save_traj else None
        # The trajectory goes to the host WHILE the run samples (the reference copies every step: pos_traj.append(pos.clone().cpu()),
        # dualenc.py:545): at every poll the steps finished since the last one are copied device -> pinned staging buffer on a side
        # stream and from there into the host tensor finish() returns -- one 11.8 GB pageable copy at the end of a 196 k-atom job
        # took 1.1 s (4 % of the job; profiles/r05_f16x3_bench.json: extra.full_jobs)
        # (only where it pays: a poll interval's chunk of >= 16 MiB, i.e. batches of ~22 k atoms or more at 64 steps per poll.  On a
        # 4,300-atom batch -- the reference driver's one-molecule calls, a launch-bound 0.32 ms step -- the side-stream copies made
        # every poll 13 ms longer (tools/poll_probe.py), while its whole trajectory is a 0.1 s copy at the end)
        self._traj_host, self._traj_sent, self._traj_pending, self._traj_stage, self._traj_stream = None, 0, None, None, None
        self._traj_ready = 0                   # trajectory rows whose writing launch has been enqueued
        self._traj_overlap = bool(save_traj) and max(int(nan_check_every), 1) * N * 12 >= int(traj_overlap_min_bytes)
        self.noise, self.on_step = noise, on_step
        self.step_lr, self.global_start_sigma = step_lr, global_start_sigma
        self.skip_discarded, self.nan_every = bool(skip_discarded_global), int(nan_check_every)
        self.raise_on_nan = bool(raise_on_nan)
        if noise_mode not in ("chunked", "per_step"):
            raise ValueError("noise_mode must be 'chunked' or 'per_step'")
        self.noise_mode = noise_mode
        self.k = 0
        self.range_graphs = set()              # graphs taken out of the run because they left the split-fp16 range (check_nan)
        self.ws.nan_flag.zero_()
        self.ws.range_rows.zero_()
        self._quarantine_non_finite_input()
        self.pos_p = _lib.ptr(self.pos)
        a = _lib.StepArgs()
        a.pos_in = self.pos_p
        a.pos_out = self.pos_p
        a.scratch = _lib.ptr(self.ws.scratch)
        a.w_global = float(w_global)
        a.clip = float(clip)
        a.clip_local = -1.0 if clip_local is None else float(clip_local)
        a.clip_pos = -1.0 if clip_pos is None else float(clip_pos)
        self.args = a
        self._nz, self._nz_base = None, 0
        self.global_steps = 0
        # steps as HIP graphs (model.step_graphs): one graph per (global branch on / off, step parity); the front kernel of a replay
        # reads its step from the device table below, the graph's last node moves the index on
        sg = getattr(model, "step_graphs", "auto")
        self._use_graphs = bool((sg is True or (sg == "auto" and N <= model.STEP_GRAPH_MAX_NODES)) and on_step is None and
                                noise_mode == "chunked" and self.pk.poly_kt > 0 and getattr(model, "fused_front", True) and
                                not getattr(model, "front_split_graph", False))
        self._graphs, self._step_table, self._step_index, self._dev_index = {}, None, None, None
        # (a capture needs a stream of its own: the legacy default stream, which torch hands out as the current one, cannot be captured)
        self._gstream = torch.cuda.Stream(device=dev) if self._use_graphs else None
        self.graph_steps = 0                   # steps that ran as graph replays (tests, bench)
        if self._use_graphs and self.noise is not None:
            self.noise = self.noise.to(dev, torch.float32).contiguous()       # (rows at fixed addresses for the step table)
        # per-step scalars of dualenc.py:515,532,536 for every step of the run, evaluated once in the reference's
        # fp32 tensor arithmetic (the loop then only touches Python floats)
        sig = self.sigmas[torch.as_tensor(self.steps, dtype=torch.long)] if len(self.steps) else self.sigmas[:0]
        step = self.step_lr * (sig / 0.01) ** 2
        self._sched = list(zip(sig.tolist(), step.tolist(), torch.sqrt(step * 2).tolist(),
                               (sig < self.global_start_sigma).tolist()))
        # the range watch once on the fresh workspace: what it polls later (check_nan) is then loaded and sized before the
        # first step (its first use in a process otherwise lands ~20 ms of lazy kernel loading inside the run's last step)
        model.check_range(self.ws)

    def _quarantine_non_finite_input(self):
        """Graphs whose INITIAL positions hold a NaN / inf are flagged like graphs that diverge later (k_langevin_update) and
        get the same finite placeholder geometry, so that no non-finite value ever enters a forward (device ops only: no
        host synchronisation; the reference would raise at the end of its first step, dualenc.py:539-541)."""
        topo, G, N = self.topo, self.topo.G, self.topo.N
        bad_node = ~torch.isfinite(self.pos).all(dim=1)
        bad_graph = torch.zeros(G, dtype=torch.int32, device=self.pos.device).index_add_(0, topo.batch64, bad_node.to(torch.int32)) > 0
        self.ws.nan_flag[1:1 + G] |= bad_graph.to(torch.int32)
        self.ws.nan_flag[0:1] |= bad_graph.any().to(torch.int32).reshape(1)
        gp = topo.graph_ptr.long()
        n_of = (gp[1:] - gp[:-1])[topo.batch64].to(torch.float32)
        li = (torch.arange(N, device=self.pos.device) - gp[:-1][topo.batch64]).to(torch.float32)
        chain = torch.zeros_like(self.pos)
        chain[:, 0] = (li - 0.5 * (n_of - 1.0)) * 1.5
        self.pos.copy_(torch.where(bad_graph[topo.batch64].unsqueeze(1), chain, self.pos))

    def remaining(self):
        return len(self.steps) - self.k

    def _noise_for(self, k, dev, N):
        if self.noise is not None:
            return self.noise[k].to(dev, torch.float32).contiguous()
        if self.noise_mode == "per_step":          # dualenc.py:529: noise = torch.randn_like(pos), one draw per step
            return torch.randn_like(self.pos)
        chunk = 128
        if self._use_graphs:
            # ONE buffer refilled in place (fixed row addresses for the step table).  Safe where _fill_args calls this: every launch
            # that reads the rows of the chunk before is already enqueued on this stream
            if self._nz is None:
                self._nz = torch.empty((chunk, N, 3), dtype=torch.float32, device=dev)
                self._nz_base, self._nz_origin = -chunk, k
            if k >= self._nz_base + chunk:
                self._nz_base = k
                self._nz[:min(chunk, len(self.steps) - k)].normal_()
            return self._nz[k - self._nz_base]
        if self._nz is None or k >= self._nz_base + self._nz.shape[0]:
            self._nz = torch.randn((min(chunk, len(self.steps) - k), N, 3), dtype=torch.float32, device=dev)
            self._nz_base = k
        return self._nz[k - self._nz_base]

    def _fused_front(self):
        """The polynomial path keeps the serial front of a step in one launch (agdiff_sampler_front: update of step t +
        radius graph of step t + 1): whenever the radius edges have their polynomials and the graph is built here."""
        return self.pk.poly_kt > 0 and getattr(self.model, "fused_front", True)

    def _fill_args(self, a, k, dev, N):
        sig, step_size, noise_scale, use_global = self._sched[k]
        cur = self._noise_for(k, dev, N)
        a.noise = _lib.ptr(cur)
        a.traj_out = ctypes.c_void_p(self.traj[k].data_ptr()) if self.traj is not None else ctypes.c_void_p(0)
        a.sigma = sig
        a.step_size = step_size
        a.noise_scale = noise_scale
        a.use_global = 1 if use_global else 0
        return cur, bool(use_global or not self.skip_discarded)

    def _build_step_table(self, first):
        """The update's arguments of EVERY step of the run as a device array of agdiff_step_args_t (what _fill_args would put into
        the launch arguments step by step), + the device index the graphs' front kernels read it at."""
        a, N, dev = self.args, self.topo.N, self.pos.device
        n = len(self.steps)
        if self.noise is None:
            self._noise_for(first, dev, N)                 # (allocates the buffer)
        tab = (_lib.StepArgs * n)()
        nz0 = self._nz.data_ptr() if self.noise is None else self.noise.data_ptr()
        tr0 = self.traj.data_ptr() if self.traj is not None else 0
        for k in range(n):
            sig, step_size, noise_scale, use_global = self._sched[k]
            e = tab[k]
            e.pos_in, e.pos_out, e.scratch = a.pos_in, a.pos_out, a.scratch
            row = ((k - self._nz_origin) % 128) if self.noise is None else k
            e.noise = nz0 + row * N * 12
            e.traj_out = (tr0 + k * N * 12) if tr0 else None
            e.sigma, e.step_size, e.noise_scale = sig, step_size, noise_scale
            e.w_global, e.clip, e.clip_local, e.clip_pos = a.w_global, a.clip, a.clip_local, a.clip_pos
            e.use_global = 1 if use_global else 0
        raw = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8)
        self._step_table = raw.to(dev)
        self._step_index = torch.zeros(1, dtype=torch.int32, device=dev)
        self._dev_index = None
        self._table_first = first

    def _step_graph(self, run_global, par, cached, cutoff):
        """The replayable graph of one steady-state iteration [update of the step before | radius graph | local edges | forward]
        for this (global branch, parity), captured on first use; None when the library refuses (then the run goes on without)."""
        key = (run_global, par)
        g = self._graphs.get(key)
        if g is None:
            V = _lib.DEFINES
            mode = 1 | (2 if run_global else 0) | 4 | (par << 4)
            flags = (run_global | self.radius_flags | self._sampler_flag | cached | V["AGDIFF_FWD_GRAPH_READY"] |
                     (V["AGDIFF_FWD_PARITY"] if par else 0))
            out = ctypes.c_void_p(0)
            rc = self.lib.agdiff_step_graph_capture(ctypes.byref(self.pk.struct), ctypes.byref(self.topo.struct), ctypes.byref(self.ws.struct),
                                                    ctypes.byref(self.args), _lib.ptr(self._step_table), _lib.ptr(self._step_index),
                                                    mode, cutoff, flags, _lib.stream_ptr(), ctypes.byref(out))
            if rc != 0 or not out.value:
                self._use_graphs = False
                return None
            g = self._graphs[key] = out.value
        return g

    def __del__(self):
        try:
            for g in getattr(self, "_graphs", {}).values():
                self.lib.agdiff_step_graph_destroy(ctypes.c_void_p(g))
        except Exception:
            pass

    def _advance_fused(self, end):
        """Steps k .. end - 1 with ONE launch between a step's global head and the next step's first CFConv: iteration k
        enqueues [update of step k - 1 | radius graph of step k] (agdiff_sampler_front), then the forward on the graph it
        left (AGDIFF_FWD_GRAPH_READY); a last update-only launch closes the chunk."""
        lib, a, pk, topo, ws = self.lib, self.args, self.pk, self.topo, self.ws
        dev, N = self.pos.device, topo.N
        stream = _lib.stream_ptr()
        V = _lib.DEFINES
        cutoff = ctypes.c_float(0.0 if (self.radius_flags & V["AGDIFF_FWD_NO_RADIUS"]) else float(self.model.config.cutoff))
        P, T, W = ctypes.byref(pk.struct), ctypes.byref(topo.struct), ctypes.byref(ws.struct)
        first, keep = self.k, []
        split_graph = bool(getattr(self.model, "front_split_graph", False))
        ws.canon_counter.zero_()
        self._graph_parity = 0
        while self.k < end:
            k, i = self.k, self.steps[self.k]
            run_global = 1 if (self._sched[k][3] or not self.skip_discarded) else 0
            if run_global:                   # the canonical list's length counter alternates with every graph build
                self._graph_parity ^= 1
            par = self._graph_parity
            pending = bool(run_global) and split_graph
            # update(k - 1) | radius graph(k) | local edges(k)   (a carries step k - 1 when an update is due; its noise row
            # stays alive in `keep`).  With `front_split_graph` the graph phase is launched by the forward, after the fork.
            mode = (1 if k > first else 0) | (2 if (run_global and not pending) else 0) | 4 | (par << 4)
            cached = V["AGDIFF_FWD_STAGE0_CACHED"] if (run_global and self._stage0_done) else 0
            # steady state as ONE graph replay: the update of step k - 1 is due, node stage 0 is cached (every kernel has run once)
            graph = None
            if self._use_graphs and k > first + 1 and (cached or not run_global):
                if self._step_table is None:
                    self._build_step_table(first)
                graph = self._step_graph(run_global, par, cached, cutoff)
            if graph is not None:
                if self._dev_index != k - 1:                # (the front kernel reads the table at the step whose update it does)
                    self._step_index.fill_(k - 1)
                _lib.check(lib.agdiff_step_graph_launch(ctypes.c_void_p(graph), stream), "agdiff_step_graph_launch")
                self._dev_index = k
                self.graph_steps += 1
            else:
                _lib.check(lib.agdiff_sampler_front(P, T, W, ctypes.byref(a), mode, cutoff, stream), "agdiff_sampler_front")
            if k > first:
                self._traj_ready = k          # (the update of step k - 1 -- the launch above -- writes trajectory row k - 1)
            if k > first and self.on_step is not None:
                self.on_step(k - 1, self.steps[k - 1], self.pos)
            self.global_steps += run_global
            if graph is None:
                _lib.check(lib.agdiff_score_forward(P, T, W, self.pos_p, run_global | self.radius_flags | self._sampler_flag | cached |
                                                    V["AGDIFF_FWD_GRAPH_READY"] | (V["AGDIFF_FWD_PARITY"] if par else 0) |
                                                    (V["AGDIFF_FWD_GRAPH_PENDING"] if pending else 0), stream), "agdiff_score_forward")
            self._stage0_done = self._stage0_done or bool(run_global)
            keep = [self._fill_args(a, k, dev, N)[0]]
            a.use_global = 1 if self._sched[k][3] else 0
            self.k += 1
            if self.k % self.nan_every == 0 and self.k < end:
                self.check_nan()
        if end > first:     # the chunk's last update
            _lib.check(lib.agdiff_sampler_front(P, T, W, ctypes.byref(a), 1, cutoff, stream), "agdiff_sampler_front")
            self._traj_ready = end
            if self.on_step is not None:
                self.on_step(end - 1, self.steps[end - 1], self.pos)
            if self.k % self.nan_every == 0 or self.k == len(self.steps):
                self.check_nan()
        del keep

    def advance(self, m):
        lib, a, pk, topo, ws = self.lib, self.args, self.pk, self.topo, self.ws
        dev, N = self.pos.device, topo.N
        end = min(self.k + int(m), len(self.steps))
        if self._gstream is not None and self._use_graphs and self._fused_front():
            # the run's own stream (ordered after what the caller has enqueued; the caller's stream waits for it at the end)
            outer = torch.cuda.current_stream(dev)
            self._gstream.wait_stream(outer)
            try:
                with torch.cuda.stream(self._gstream), torch.no_grad():
                    return self._advance_fused(end)
            finally:
                outer.wait_stream(self._gstream)
        stream = _lib.stream_ptr()
        with torch.no_grad():
            if self._fused_front():
                return self._advance_fused(end)
            while self.k < end:
                k, i = self.k, self.steps[self.k]
                cur = self._noise_for(k, dev, N)
                sig, step_size, noise_scale, use_global = self._sched[k]
                a.noise = _lib.ptr(cur)
                a.traj_out = ctypes.c_void_p(self.traj[k].data_ptr()) if self.traj is not None else ctypes.c_void_p(0)
                a.sigma = sig
                a.step_size = step_size
                a.noise_scale = noise_scale
                a.use_global = 1 if use_global else 0
                run_global = 1 if (use_global or not self.skip_discarded) else 0
                self.global_steps += run_global
                # node stage 0 of the SchNet encoder does not depend on the positions: from the second global step of
                # the run on its cached outputs are used (ws.h0 / ws.xs0)
                cached = _lib.DEFINES["AGDIFF_FWD_STAGE0_CACHED"] if (run_global and self._stage0_done) else 0
                _lib.check(lib.agdiff_score_forward(ctypes.byref(pk.struct), ctypes.byref(topo.struct),
                                                    ctypes.byref(ws.struct), self.pos_p,
                                                    run_global | self.radius_flags | self._sampler_flag | cached, stream),
                           "agdiff_score_forward")
                self._stage0_done = self._stage0_done or bool(run_global)
                _lib.check(lib.agdiff_langevin_update(ctypes.byref(topo.struct), ctypes.byref(ws.struct),
                                                      ctypes.byref(a), stream), "agdiff_langevin_update")
                self.k += 1
                self._traj_ready = self.k
                if self.on_step is not None:
                    self.on_step(k, i, self.pos)
                if self.k % self.nan_every == 0 or self.k == len(self.steps):
                    self.check_nan()

    def check_nan(self):
        """dualenc.py:539-541.  ws.nan_flag[0] is set by the update kernel as soon as any position is NaN.  Polled every
        nan_check_every steps in BOTH modes, together with the split-fp16 range watch (model.range_report).  raise_on_nan
        (the reference's behaviour, dualenc.py:531-533): a NaN raises FloatingPointError and a range violation AgdiffRangeError
        for the whole call.  Otherwise (agdiff_amd.driver: per-molecule retry): the graphs that own out-of-range rows are
        taken out of the run like diverged ones -- their positions are set to NaN, which the next update quarantines and flags
        per graph (nan_graphs) -- and recorded in `range_graphs`, so that the driver re-samples exactly those molecules in
        split-bf16 (fp32's range) while the others run on."""
        self._traj_land()                    # (before the first synchronisation: the device still has the interval's steps queued)
        self._traj_send(self._traj_ready)    # (rows whose update is enqueued: with the fused front the last step's is still pending)
        if self.raise_on_nan and int(self.ws.nan_flag[0].item()) != 0:
            print("NaN detected. Please restart.")
            raise FloatingPointError()
        if self.raise_on_nan:
            self.model.check_range(self.ws, self.topo.batch64)
            return
        rep = self.model.range_report(self.ws, self.topo.batch64)
        if rep is not None:
            graphs = [g for g in rep[3] if g not in self.range_graphs]
            self.range_graphs.update(graphs)
            if graphs and self.k < len(self.steps):
                rows = torch.isin(self.topo.batch64, torch.as_tensor(graphs, device=self.topo.batch64.device))
                self.pos[rows] = float("nan")

    def _traj_land(self):
        """The chunk whose device -> pinned copy was issued at the last poll: pinned -> the host tensor (a host memcpy: called where
        the host is about to block on the device anyway, with the steps of the next interval already enqueued)."""
        if self._traj_pending is not None:
            ev, lo, hi, buf = self._traj_pending
            ev.synchronize()
            self._traj_host[lo:hi].copy_(buf[:hi - lo])
            self._traj_pending = None

    def _traj_send(self, upto):
        """Steps [_traj_sent, upto) of the device trajectory -> pinned staging on the side stream (see __init__)."""
        if self.traj is None or upto <= self._traj_sent or not self._traj_overlap:
            return
        if self._traj_host is None:
            self._traj_host = torch.empty((len(self.steps),) + tuple(self.traj.shape[1:]), dtype=torch.float32)
            chunk = max(1, min(len(self.steps), max(self.nan_every, 1)))
            self._traj_stage = [torch.empty((chunk,) + tuple(self.traj.shape[1:]), dtype=torch.float32, pin_memory=True) for _ in range(2)]
            self._traj_stream = torch.cuda.Stream(device=self.traj.device)
        chunk = self._traj_stage[0].shape[0]
        while self._traj_sent < upto:
            self._traj_land()
            lo, hi = self._traj_sent, min(upto, self._traj_sent + chunk)
            buf = self._traj_stage[(lo // chunk) & 1]
            ready = torch.cuda.Event()
            ready.record()                                  # the steps up to `hi` are enqueued on the compute stream
            with torch.cuda.stream(self._traj_stream):
                self._traj_stream.wait_event(ready)
                buf[:hi - lo].copy_(self.traj[lo:hi], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._traj_stream)
            self._traj_pending = (ev, lo, hi, buf)
            self._traj_sent = hi

    def nan_graphs(self):
        """Bool tensor [G] (host): graphs in which a position became NaN so far (ws.nan_flag[1 + g]) or that left the
        split-fp16 range (range_graphs: also when that showed at the very last poll, after the last update)."""
        bad = self.ws.nan_flag[1:1 + self.topo.G].cpu() != 0
        if self.range_graphs:
            bad[torch.as_tensor(sorted(self.range_graphs), dtype=torch.long)] = True
        return bad

    def finish(self):
        """(pos on device, pos_traj list of CPU tensors) as dualenc.py:547 returns them."""
        self.check_nan()
        if self.traj is not None and self._traj_overlap:
            self._traj_send(self.k)
            self._traj_land()
            pos_traj = list(self._traj_host[:self.k].unbind(0))
        else:
            pos_traj = list(self.traj[:self.k].cpu().unbind(0)) if self.traj is not None else []
        return self.pos, pos_traj


def get_model(config):
    """epsnet/__init__.py:7-11."""
    if config.network == "dualenc":
        return DualEncoderEpsNetwork(config)
    else:
        raise NotImplementedError("Unknown network: %s" % config.network)


# helpers with the reference's names (dualenc.py:550-589) that drivers import
def is_local_edge(edge_type):
    return edge_type > 0


def is_radius_edge(edge_type):
    return edge_type == 0
