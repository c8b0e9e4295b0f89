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


def elem_err(a, b, floor_frac=1e-3):
    """Element-wise figure: max_i |a_i - b_i| / max(|b_i|, floor), floor = floor_frac * max|b| -- a value
    much smaller than the tensor's scale is measured against the floor instead of against itself."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0 and b.size == 0:
        return 0.0
    floor = max(floor_frac * np.abs(b).max(), 1e-30)
    return float((np.abs(a - b) / np.maximum(np.abs(b), floor)).max())


# Parity gates of the HIP path per arithmetic mode (tests/test_hip_parity.py, tests/test_hip_kernels.py).
# BASELINE.json's north_star asks for 1e-4 relative fp32; the two modes are gated a small factor above what they
# measure on MI355X (profiles/r02_parity_errors.json: exact-fp32 MFMA ~1e-6 normwise on the fixtures, split-bf16
# ~1.5e-5; element-wise with the 1e-3 floor 5e-4 / 1.2e-2), so that a regression of half an order of magnitude
# turns the suite red.  A call site that needs more room says why and passes `scale`.
# split-fp16 ("f16x3", the default mode since round 4): measured <= 5.7e-6 normwise on the reference fixtures, 1.1e-5 on
# default-initialised weights, 1.5e-5 after three sampler steps of a 300-atom molecule, element-wise <= 3.4e-3
# (profiles/r04_parity_errors.json)
TOL_NORM = {"f32": 5e-6, "bf16x3": 3e-5, "f16x3": 2e-5}
TOL_ELEM = {"f32": 2e-3, "bf16x3": 3e-2, "f16x3": 5e-3}
# (No environment switch loosens these gates: tests/conftest.py refuses to start a session in which
# AGDIFF_PARITY_GATE_SCALE -- the measuring knob of rounds 2-3 -- is set.)
_RECORDS = []


def check_close(name, got, ref, precision, scale=1.0):
    """Assert both parity figures for `got` vs `ref` in mode `precision`, and record them (dumped to
    gpurun_out/parity_errors.json at session end by conftest.py).  `scale` loosens both gates for quantities that
    accumulate over several steps / layers (stated at the call site)."""
    if hasattr(got, "detach"):
        got = got.detach().cpu().numpy()
    if hasattr(ref, "detach"):
        ref = ref.detach().cpu().numpy()
    rn, re_ = rel_err(got, ref), elem_err(got, ref)
    _RECORDS.append({"name": name, "precision": precision, "normwise": rn, "elementwise": re_,
                     "gate_norm": TOL_NORM[precision] * scale, "gate_elem": TOL_ELEM[precision] * scale})
    print("parity %-48s %-7s normwise %.2e  elementwise %.2e" % (name, precision, rn, re_))
    assert rn < TOL_NORM[precision] * scale, "%s (%s): normwise %.3e" % (name, precision, rn)
    assert re_ < TOL_ELEM[precision] * scale, "%s (%s): element-wise %.3e" % (name, precision, re_)
    return rn


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
