"""CPU: host logic of the sampling driver (scripts/test.py counterpart): test-set file format, batch
planning / packing, retry-on-NaN.  The model is a stub; nothing here computes on a GPU."""
import numpy as np
import torch

from agdiff_amd import driver, synth


def _mols(k=5):
    rng = np.random.default_rng(3)
    out = []
    for i in range(k):
        n = int(rng.integers(8, 20))
        at, r, c, t = synth.random_molecule(rng, n)
        out.append(dict(atom_type=at, edge_index=np.stack([r, c]), edge_type=t, num_refs=3 + i, name="m%d" % i))
    return out


def test_testset_roundtrip_and_num_confs(tmp_path):
    mols = _mols()
    p = str(tmp_path / "t.npz")
    driver.save_testset(p, mols)
    back = driver.load_testset(p)
    assert len(back) == len(mols)
    for a, b in zip(mols, back):
        assert np.array_equal(a["atom_type"], b["atom_type"]) and np.array_equal(a["edge_index"], b["edge_index"])
        assert a["num_refs"] == b["num_refs"] and a["name"] == b["name"]
    assert driver.num_confs("2x")(7) == 14 and driver.num_confs("50")(7) == 50


def test_plan_and_pack_batches():
    mols = _mols(6)
    confs = driver.num_confs("2x")
    batches = driver.plan_batches(mols, confs, max_atoms=150)
    assert sum(len(b) for b in batches) == 6 and all(len(b) >= 1 for b in batches)
    for b in batches:
        atoms = sum(m["atom_type"].shape[0] * confs(m["num_refs"]) for m in b)
        assert atoms <= 150 or len(b) == 1
    packed = driver.pack_batch(batches[0], confs)
    n_total = packed["atom_type"].shape[0]
    assert packed["batch"].shape[0] == n_total and packed["batch"][-1] == packed["num_graphs"] - 1
    assert np.all(np.diff(packed["batch"]) >= 0)
    bi = packed["bond_index"]
    assert bi.min() >= 0 and bi.max() < n_total and np.all(packed["batch"][bi[0]] == packed["batch"][bi[1]])
    off, n, g = packed["spans"][0]
    m0 = batches[0][0]
    assert np.array_equal(packed["atom_type"][off:off + n], m0["atom_type"]) and g == confs(m0["num_refs"])


class _StubModel:
    def __init__(self, fail_first):
        self.calls, self.fail_first = [], fail_first

    def langevin_dynamics_sample_diffusion(self, **kw):
        self.calls.append(kw["clip_local"])
        if self.fail_first and len(self.calls) == 1:
            raise FloatingPointError()
        n = kw["atom_type"].shape[0]
        return torch.zeros(n, 3), [torch.zeros(n, 3)] * 2


def test_retry_with_local_clipping():
    packed = driver.pack_batch(_mols(2), driver.num_confs("1"))
    m = _StubModel(fail_first=True)
    pos, traj = driver.sample_batch(m, packed, "cpu", dict(n_steps=2), save_traj=True, log=lambda s: None)
    assert m.calls == [None, 20] and pos.shape == (packed["atom_type"].shape[0], 3) and traj.shape[0] == 2
    m2 = _StubModel(fail_first=False)
    driver.sample_batch(m2, packed, "cpu", dict(n_steps=2))
    assert m2.calls == [None]
