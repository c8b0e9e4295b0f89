"""Shared test helpers (CPU side): golden loading, oracle state dicts, error metrics."""
import os

import numpy as np
import torch

from agdiff_amd.config import Config, drugs_model_config, qm9_model_config

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def t(x, dtype=None):
    r = torch.from_numpy(np.ascontiguousarray(x))
    return r if dtype is None else r.to(dtype)


def rel_err(a, b):
    """max |a-b| / max(|b|_inf, tiny): the 'relative fp32' figure used throughout the tests."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0 and b.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


FORWARD_CASES = {
    "g3_forward_qm9_small": lambda: qm9_model_config(),
    "g3_forward_smooth_sparse": lambda: drugs_model_config(),
    "g3_forward_drugs_capped": lambda: drugs_model_config(),
    "g3_forward_gaussian": lambda: qm9_model_config(edge_encoder="gaussian"),
    "g3_forward_gaussian_drugs": lambda: drugs_model_config(edge_encoder="gaussian"),
}
STAGE_CASES = ["g3_forward_qm9_small", "g3_forward_smooth_sparse", "g3_forward_gaussian"]
SAMPLER_CASES = ["g5_sampler_top", "g5_sampler_lowT_global", "g5_sampler_mixed_cliplocal", "g5_sampler_gaussian"]


def sampler_case_cfg(g, case=""):
    base = drugs_model_config if int(g["cfg_smooth"]) else qm9_model_config
    return base(num_diffusion_timesteps=int(g["cfg_T"]), beta_end=float(g["cfg_beta_end"]),
                edge_encoder="gaussian" if "gaussian" in case else "mlp")


def sampler_case_kwargs(g):
    kw = {}
    for k in g:
        if k.startswith("kw_"):
            v = float(g[k])
            kw[k[3:]] = None if np.isnan(v) else v
    return kw
