"""CPU: host logic of the sampling driver (scripts/test.py counterpart): test-set file format, batch
planning / packing, retry-on-NaN.  The model is a stub; nothing here computes on a GPU."""
import numpy as np
import pytest
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


def test_retry_with_local_clipping_whole_batch_models():
    """A model without begin_sampling (the reference's own module would be one) gets test.py:143-181 as written."""
    packed = driver.pack_batch(_mols(2), driver.num_confs("1"))
    m = _StubModel(fail_first=True)
    pos, traj, ok = driver.sample_batch(m, packed, "cpu", dict(n_steps=2), save_traj=True, log=lambda s: None)
    assert m.calls == [None, 20] and pos.shape == (packed["atom_type"].shape[0], 3) and traj.shape[0] == 2 and ok.all()
    m2 = _StubModel(fail_first=False)
    driver.sample_batch(m2, packed, "cpu", dict(n_steps=2))
    assert m2.calls == [None]


class _FakeRun:
    """Stands where epsnet.LangevinRun does: positions = 100 * (molecule fingerprint) + atom index; a graph is
    flagged NaN when its first atom type equals `nan_type` and no local clipping is requested."""

    def __init__(self, owner, at, pos_init, batch, G, clip_local, save_traj, n_steps, on_step=None):
        self.owner, self.at, self.p0, self.batch, self.G = owner, at, pos_init, batch, G
        self.clip_local, self.save_traj, self.n_steps, self.on_step = clip_local, save_traj, n_steps, on_step
        self.ws = type("W", (), {"nan_flag": torch.zeros(1 + G, dtype=torch.int32)})()

    def remaining(self):
        return self.n_steps

    def advance(self, m):
        self.pos = self.p0 * 0.5 + self.at.to(torch.float32)[:, None]
        for k in range(m):
            if self.on_step is not None:
                self.on_step(k, k, self.pos)

    def finish(self):
        return self.pos, ([self.pos.clone()] * self.n_steps if self.save_traj else [])

    def nan_graphs(self):
        first = torch.zeros(self.G, dtype=torch.long).scatter_reduce(0, self.batch, self.at, "amin", include_self=False)
        return (first == self.owner.nan_type) & torch.tensor(self.clip_local is None)


class _FakeSampler:
    def __init__(self, nan_type=-1):
        self.nan_type, self.calls = nan_type, []

    def begin_sampling(self, at, pos_init, bi, bt, batch, G, extend_order, clip_local=None, save_traj=True,
                       raise_on_nan=True, noise=None, n_steps=2, **kw):
        assert raise_on_nan is False and extend_order is False
        self.calls.append((int(G), clip_local))
        return _FakeRun(self, at, pos_init, batch, G, clip_local, save_traj, n_steps)


def test_only_the_diverging_molecule_is_resampled():
    mols = _mols(4)
    mols[2]["atom_type"] = mols[2]["atom_type"].copy()
    mols[2]["atom_type"][:] = 9          # fingerprint of the molecule that "diverges"
    packed = driver.pack_batch(mols, driver.num_confs("2"))
    m = _FakeSampler(nan_type=9)
    p0 = torch.randn(packed["atom_type"].shape[0], 3)
    pos, traj, ok = driver.sample_batch(m, packed, "cpu", dict(n_steps=2), save_traj=True, log=lambda s: None, pos_init=p0)
    # first attempt: all 8 graphs, no clipping; second: only molecule 2's two conformers, clip_local=20
    assert m.calls == [(8, None), (2, 20)] and ok.all()
    expect = p0 * 0.5 + torch.from_numpy(packed["atom_type"]).float()[:, None]
    off, n, g = packed["spans"][2]
    keep = torch.ones(pos.shape[0], dtype=torch.bool)
    keep[off:off + n * g] = False
    assert torch.equal(pos[keep], expect[keep])                 # the healthy molecules keep their FIRST result
    assert torch.isfinite(pos).all() and not torch.equal(pos[~keep], expect[~keep])     # re-drawn pos_init
    assert traj.shape == (2,) + tuple(pos.shape)
    # a molecule that fails twice is dropped, the others stay
    m3 = _FakeSampler(nan_type=9)
    m3_run = m3.begin_sampling
    m3.begin_sampling = lambda *a, **k: m3_run(*a, **dict(k, clip_local=None))      # clipping does not help
    pos, _, ok = driver.sample_batch(m3, packed, "cpu", dict(n_steps=2), log=lambda s: None)
    assert ok.tolist() == [True, True, False, True] and torch.isnan(pos[off:off + n * g]).all()


class _RangeSampler(_FakeSampler):
    """A sampler whose runs report the graphs of molecule fingerprint `nan_type` as out of the split-fp16 range (and failed) while
    the model is not in split-bf16: what epsnet.LangevinRun.check_nan does with raise_on_nan=False."""

    def __init__(self, nan_type):
        super().__init__(nan_type)
        self.precision, self.precision_local, self.modes = "f16x3", None, []

    def arithmetic(self, precision=None, precision_local=None):
        import contextlib

        @contextlib.contextmanager
        def cm():
            old = (self.precision, self.precision_local)
            self.precision, self.precision_local = precision, precision_local
            try:
                yield self
            finally:
                self.precision, self.precision_local = old
        return cm()

    def begin_sampling(self, *a, **kw):
        run = super().begin_sampling(*a, **kw)
        self.modes.append((self.precision, self.precision_local))
        wide = self.precision == "bf16x3"
        bad = run.nan_graphs() if not wide else torch.zeros(run.G, dtype=torch.bool)
        run.range_graphs = set(torch.nonzero(bad).view(-1).tolist())
        run.nan_graphs = lambda: bad
        return run


def test_molecules_that_leave_the_split_fp16_range_are_resampled_in_split_bf16():
    """a range trip must not abort the job -- the affected molecules (and only they) are sampled again with the
    model in split-bf16 for that attempt; the healthy ones keep their first result; the mode is restored afterwards."""
    mols = _mols(4)
    mols[1]["atom_type"] = mols[1]["atom_type"].copy()
    mols[1]["atom_type"][:] = 9
    packed = driver.pack_batch(mols, driver.num_confs("2"))
    m = _RangeSampler(nan_type=9)
    before = dict(driver.SAMPLE_STATS)
    logs = []
    pos, _, ok = driver.sample_batch(m, packed, "cpu", dict(n_steps=2), log=logs.append)
    assert ok.all() and torch.isfinite(pos).all()
    # (a molecule whose only fault was the range keeps the sampler settings -- no local clipping -- and its extra pass
    # is not one of the max_retry attempts: test.py:143-181 clips only after a FloatingPointError)
    assert m.calls == [(8, None), (2, None)] and m.modes == [("f16x3", None), ("bf16x3", "bf16x3")]
    assert (m.precision, m.precision_local) == ("f16x3", None)
    assert driver.SAMPLE_STATS["range_trips"] - before["range_trips"] == 2
    assert driver.SAMPLE_STATS["bf16x3_retries"] - before["bf16x3_retries"] == 1
    assert any("split-bf16" in l for l in logs)


def test_a_range_trip_does_not_use_up_a_nan_attempt():
    """molecule 1 leaves the split-fp16 range in the first pass, and in its split-bf16 pass a NaN appears: it still
    gets its clipped retry (max_retry = 2 counts NaN attempts only), and a molecule that fails every attempt is counted as dropped."""
    mols = _mols(3)
    mols[1]["atom_type"] = mols[1]["atom_type"].copy()
    mols[1]["atom_type"][:] = 9
    packed = driver.pack_batch(mols, driver.num_confs("2"))

    class _RangeThenNan(_RangeSampler):
        def begin_sampling(self, *a, **kw):
            run = _FakeSampler.begin_sampling(self, *a, **kw)
            self.modes.append((self.precision, kw.get("clip_local")))
            bad = run.nan_graphs()                                   # (the fake run: NaN while clip_local is None)
            run.range_graphs = set(torch.nonzero(bad).view(-1).tolist()) if self.precision != "bf16x3" else set()
            run.nan_graphs = lambda: bad
            return run
    m = _RangeThenNan(nan_type=9)
    before = dict(driver.SAMPLE_STATS)
    pos, _, ok = driver.sample_batch(m, packed, "cpu", dict(n_steps=2), log=lambda s: None)
    assert m.modes == [("f16x3", None), ("bf16x3", None), ("bf16x3", 20)] and ok.all() and torch.isfinite(pos).all()
    assert driver.SAMPLE_STATS["dropped"] == before["dropped"]
    m2 = _RangeThenNan(nan_type=9)
    run2 = m2.begin_sampling
    m2.begin_sampling = lambda *a, **k: run2(*a, **dict(k, clip_local=None))        # clipping does not help either
    pos, _, ok = driver.sample_batch(m2, packed, "cpu", dict(n_steps=2), log=lambda s: None)
    assert ok.tolist() == [True, False, True] and driver.SAMPLE_STATS["dropped"] - before["dropped"] == 1


def test_subset_batch_rebases():
    mols = _mols(5)
    confs = driver.num_confs("2x")
    packed = driver.pack_batch(mols, confs)
    sub = driver.subset_batch(packed, [1, 3])
    ref = driver.pack_batch([mols[1], mols[3]], confs)
    for k in ("atom_type", "bond_index", "bond_type", "batch"):
        assert np.array_equal(sub[k], ref[k]), k
    assert sub["num_graphs"] == ref["num_graphs"] and sub["spans"] == ref["spans"]


def test_resume_keeps_every_finished_molecule(tmp_path):
    """a resumed run plans batches over the remaining molecules; its files must not replace files of the
    interrupted run that hold other molecules."""
    mols = _mols(7)
    for i, m in enumerate(mols):
        m["index"] = i
    confs, out = driver.num_confs("2"), str(tmp_path / "out")
    model = _FakeSampler()
    # "interrupted" run: only molecules 0..3 get sampled
    first = driver.run_job(model, mols[:4], out, confs, 70, dict(n_steps=1), "cpu", log=lambda s: None)
    assert sorted(k for k in first if k.startswith("pos_gen_")) == ["pos_gen_%d" % i for i in range(4)]
    before = {k: first[k].copy() for k in first if k.startswith("pos_gen_")}
    merged = driver.run_job(model, mols, out, confs, 70, dict(n_steps=1), "cpu", resume=True, log=lambda s: None)
    assert sorted(int(k.split("_")[-1]) for k in merged if k.startswith("pos_gen_")) == list(range(7))
    for k, v in before.items():
        assert np.array_equal(merged[k], v), k                    # nothing of the first run was overwritten
    z = np.load(out + "/samples_all.npz")
    assert all(("pos_gen_%d" % i) in z.files for i in range(7))
    # a third run with --resume has nothing left to do and changes nothing
    again = driver.run_job(model, mols, out, confs, 70, dict(n_steps=1), "cpu", resume=True, log=lambda s: None)
    assert all(np.array_equal(again[k], merged[k]) for k in merged)


def test_reference_checkpoint_with_easydict_config(tmp_path):
    """scripts/train.py:219-231 pickles `config` as easydict.EasyDict; the package is not installed here, so the
    file is written against a stand-in module that exists only while saving."""
    import sys
    import types
    from agdiff_amd import Config, compat, get_model, qm9_model_config

    class EasyDict(dict):                    # behaviour of the PyPI package: attribute <-> item, nested dicts wrapped
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                setattr(self, k, v)

        def __setattr__(self, name, value):
            if isinstance(value, dict) and not isinstance(value, EasyDict):
                value = EasyDict(value)
            super().__setattr__(name, value)
            super().__setitem__(name, value)
        __setitem__ = __setattr__
    EasyDict.__module__, EasyDict.__qualname__ = "easydict", "EasyDict"
    fake = types.ModuleType("easydict")
    fake.EasyDict = EasyDict
    cfg = qm9_model_config()
    m = get_model(cfg)
    path = str(tmp_path / "ckpt.pt")
    had = sys.modules.get("easydict")
    sys.modules["easydict"] = fake
    try:
        torch.save({"config": EasyDict({"model": dict(cfg), "train": {"seed": 2021, "optimizer": {"lr": 1e-3}}}),
                    "model": m.state_dict(), "optimizer_global": {"state": {}, "param_groups": [{"lr": 1e-3}]},
                    "iteration": 7, "avg_val_loss": 0.5}, path)
    finally:
        if had is None:
            del sys.modules["easydict"]
        else:
            sys.modules["easydict"] = had
    ckpt = compat.load_checkpoint(path)
    assert type(ckpt["config"]) is Config and type(ckpt["config"].train.optimizer) is Config
    mc = compat.model_config(ckpt)
    assert mc.hidden_dim == 128 and mc.edge_encoder == "mlp" and mc["cutoff"] == 10.0 and ckpt["iteration"] == 7
    m2 = get_model(mc)
    m2.load_state_dict(ckpt["model"], strict=True)
    assert compat.model_config({"config": {"model": dict(cfg)}}).num_convs == 6        # plain-dict configs too
    # nothing but tensors / containers / EasyDict is admitted
    import pickle
    bad = str(tmp_path / "bad.pt")
    torch.save({"config": Config(model=cfg), "model": {}, "x": __import__("pathlib").PurePosixPath("a")}, bad)
    with pytest.raises(pickle.UnpicklingError):
        compat.load_checkpoint(bad)
    assert compat.load_checkpoint(bad, trust=True)["x"].name == "a"
    # ... and nothing by module PREFIX: a global reached through an admitted module (torch.serialization imports os),
    # a dotted name, builtins.getattr and a real optimizer's numpy-scalar state are each decided by exact (module, name)
    import io
    for mod, name in (("torch.serialization", "os.getcwd"), ("torch.serialization", "os"), ("torch", "serialization.os.getcwd"),
                      ("builtins", "getattr"), ("builtins", "eval"), ("numpy", "load"), ("collections", "abc"),
                      ("torch.hub", "load"), ("torch", "load")):
        with pytest.raises(pickle.UnpicklingError):
            compat._Unpickler(io.BytesIO(b"")).find_class(mod, name)
    payload = (b"\x80\x04" + b"\x8c\x13torch.serialization" + b"\x8c\x09os.getcwd" + b"\x93" + b")R.")   # STACK_GLOBAL; call
    with pytest.raises(pickle.UnpicklingError):
        compat._Unpickler(io.BytesIO(payload)).load()
    for mod, name in (("torch._utils", "_rebuild_tensor_v2"), ("collections", "OrderedDict"), ("torch", "FloatStorage"),
                      ("torch", "float32"), ("torch", "Size"), ("numpy", "dtype")):
        assert compat._Unpickler(io.BytesIO(b"")).find_class(mod, name) is not None
    full = str(tmp_path / "full.pt")          # what scripts/train.py:219-231 really saves: live optimizer / scheduler state
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    sch = torch.optim.lr_scheduler.ReduceLROnPlateau(opt)
    torch.save({"config": Config(model=cfg), "model": m.state_dict(), "optimizer_global": opt.state_dict(),
                "scheduler_global": sch.state_dict(), "iteration": 3, "avg_val_loss": float(np.float32(0.25))}, full)
    assert compat.load_checkpoint(full)["scheduler_global"]["patience"] == 10


def test_compat_install_aliases_the_reference_import_path():
    import importlib
    import sys
    from agdiff_amd import compat, epsnet
    saved = {k: sys.modules.get(k) for k in ("agdiff", "agdiff.models", "agdiff.models.epsnet",
                                             "agdiff.models.epsnet.dualenc", "easydict")}
    try:
        compat.install()
        mod = importlib.import_module("agdiff.models.epsnet")             # scripts/test.py:21
        assert mod.get_model is epsnet.get_model
        from agdiff.models.epsnet import get_model as gm                  # noqa: F401
        from agdiff.models.epsnet.dualenc import DualEncoderEpsNetwork as D
        assert D is epsnet.DualEncoderEpsNetwork
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_prepare_batch_leaves_a_refused_topology_to_the_sampler():
    """driver.prepare_batch runs ahead of the sampler in a background thread: a batch whose topology cannot be built (a molecule
    beyond the atom limit, ...) must not raise there -- on one rank of a sharded job that would leave the other ranks inside
    their collectives -- but come back without a topology, so that the sampler builds it again and fails where failures are
    kept in step (dist.sample_batch_sharded)."""
    from agdiff_amd import driver

    class Refuses:
        def prepare_topology(self, *a, **k):
            raise NotImplementedError("graphs with more than 512 atoms are not supported")

    class Builds:
        def prepare_topology(self, atom_type, *a, **k):
            return ("topology of", int(np.asarray(atom_type).shape[0]))

    rng = np.random.default_rng(0)
    mols = []
    for i in range(2):
        n = 5 + i
        mols.append(dict(atom_type=rng.integers(1, 9, n), edge_index=np.stack([np.arange(n - 1), np.arange(1, n)]),
                         edge_type=np.ones(n - 1, dtype=np.int64), num_refs=1, name="m%d" % i, index=i))
    confs = driver.num_confs("2")
    packed, topo = driver.prepare_batch(Refuses(), mols, confs)
    assert topo is None and packed["num_graphs"] == 4
    packed, topo = driver.prepare_batch(Builds(), mols, confs)
    assert topo == ("topology of", packed["atom_type"].shape[0])
    packed, topo = driver.prepare_batch(object(), mols, confs)            # (test stubs without prepare_topology)
    assert topo is None


def test_preparation_worker_process_round_trip_and_fallback(tmp_path):
    """run_job prepares the next batch in a worker PROCESS (agdiff_amd/prep_worker.py; its own module, not a
    re-import of the caller's __main__).  The worker's packed batch equals pack_batch's, its BatchTopology arrives pickled with
    the struct re-pointed at its own tensors; a worker that went away is survived (the batch is prepared on the main thread)."""
    from agdiff_amd import topology
    from agdiff_amd.prep_worker import Client
    mols = _mols(3)
    confs = driver.num_confs("2x")
    counts = [confs(m["num_refs"]) for m in mols]
    opts = dict(order=3, group_targets=None, radius_column=True)
    w = Client()
    try:
        w.submit(mols, counts, 0, 1, opts)
        packed, topo = w.result()
        ref = driver.pack_batch(mols, confs)
        assert all(np.array_equal(packed[k], ref[k]) for k in ("atom_type", "bond_index", "bond_type", "batch")) and packed["spans"] == ref["spans"]
        here = topology.BatchTopology(ref["atom_type"], ref["bond_index"], ref["bond_type"], ref["batch"], ref["num_graphs"], device="cpu")
        assert topo.fingerprint == here.fingerprint and topo.struct.num_quads == here.struct.num_quads
        assert torch.equal(topo.quad_tgt, here.quad_tgt) and topo.struct.quad_tgt == topo.quad_tgt.data_ptr() != here.quad_tgt.data_ptr()
        w.proc.kill()
        w.proc.wait()
        with pytest.raises(Exception):
            w.submit(mols, counts, 0, 1, opts)
            w.result()
    finally:
        w.close()
