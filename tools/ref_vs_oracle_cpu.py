#!/usr/bin/env python3
"""SURVEY 8(d) cross-check (BUILD CONTAINER ONLY -- it imports /root/reference, which does not exist on the GPU box): seconds
per denoising step of the TRUE reference (its own model files on the stand-ins of tests/golden/ref_shims) against the oracle
(oracle/agdiff_oracle.py, bench.py's cpu_baseline "port") on this container's 8 cores, same weights, same batch, same step.
   python tools/ref_vs_oracle_cpu.py [--copies 100] [--steps 3]"""
import argparse, json, os, sys, time, types
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if not os.path.isdir("/root/reference/src"):
    raise SystemExit("needs /root/reference (build container only)")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden", "ref_shims"))
sys.path.insert(0, "/root/reference/src")
import numpy as np
import torch
ap = argparse.ArgumentParser()
ap.add_argument("--copies", type=int, default=100)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--threads", type=int, default=8)
args = ap.parse_args()
torch.set_num_threads(args.threads)
import agdiff, agdiff.utils
chem = types.ModuleType("agdiff.utils.chem")
chem.BOND_TYPES = {i: i for i in range(22)}
chem.BOND_NAMES = {i: str(i) for i in range(22)}
sys.modules["agdiff.utils.chem"] = chem
agdiff.utils.chem = chem
from agdiff.models.epsnet import get_model as ref_get_model
from agdiff.models.epsnet import dualenc as ref_dualenc
from agdiff_amd import synth
from agdiff_amd.config import drugs_model_config
from oracle import agdiff_oracle as O

cfg = drugs_model_config(beta_end=2e-5)
m = ref_get_model(cfg)
sd = synth.synth_state_dict(m.state_dict())
m.load_state_dict(sd)
m.eval()
b = synth.make_packed_batch("drugs", 1, args.copies, seed=2021)
T = lambda x: torch.from_numpy(np.ascontiguousarray(x))
at, bi, bt, ba = T(b["atom_type"]), T(b["bond_index"]), T(b["bond_type"]), T(b["batch"])
pos = torch.randn(at.shape[0], 3, generator=torch.Generator().manual_seed(2021))
kw = dict(extend_order=False, w_global=1.0, global_start_sigma=0.5, clip=1000.0)
ref_dualenc.tqdm = lambda it, **k: it


def timed(fn):
    fn()                                    # warm-up step
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = fn()
    return (time.perf_counter() - t0) / args.steps, out


t_ref, (p_ref, _) = timed(lambda: m.langevin_dynamics_sample_diffusion(at, pos, bi, bt, ba, b["num_graphs"], n_steps=1, **kw))
sdo = O.synth_state_dict_for(cfg)
t_or, (p_or, _) = timed(lambda: O.langevin_dynamics_sample_diffusion(sdo, cfg, at, pos, bi, bt, ba, b["num_graphs"], n_steps=1, **kw))
print(json.dumps({"atoms": int(at.shape[0]), "conformers": int(b["num_graphs"]), "threads": args.threads, "steps_timed": args.steps,
                  "reference_s_per_step": t_ref, "oracle_s_per_step": t_or, "oracle_over_reference": t_or / t_ref,
                  "note": "both draw their own noise (one step from the same start): outputs are not compared here -- "
                          "tests/test_oracle_golden.py pins the oracle to the reference's outputs"}))
