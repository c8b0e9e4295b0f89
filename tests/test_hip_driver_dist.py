"""GPU (MI355X): the sampling driver (scripts/test.py counterpart) against the oracle, the per-molecule NaN retry,
the RCCL all-gather path on one GPU (world_size 1, backend nccl), and size-independent properties at the
BASELINE.json config shapes that have no fixture (QM9-shaped test batch, 200-atom molecules)."""
import os
import socket

import numpy as np
import pytest
import torch

from helpers import check_close, t

pytestmark = pytest.mark.gpu


def _gpu_model(cfg, head_scale=1e-3, precision="f16x3"):
    from agdiff_amd import get_model
    from oracle import agdiff_oracle as O
    sd = O.synth_state_dict_for(cfg, head_scale=head_scale)
    m = get_model(cfg)
    m.precision = precision
    m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    return m.to("cuda:0").eval(), sd


def _three_molecules(seed=5):
    from agdiff_amd import synth
    rng = np.random.default_rng(seed)
    mols = []
    for i in range(3):
        at, r, c, ty = synth.random_molecule(rng, int(rng.integers(9, 22)))
        mols.append(dict(atom_type=at, edge_index=np.stack([r, c]), edge_type=ty, num_refs=1 + i, name="mol%d" % i, index=i))
    return mols


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x3"])
def test_driver_sample_batch_matches_oracle_per_molecule(precision):
    """every molecule's pos_gen out of driver.sample_batch (packed batch, injected pos_init / noise)
    against the oracle's langevin_dynamics_sample_diffusion with scripts/test.py's arguments."""
    from agdiff_amd import driver, qm9_model_config
    from oracle import agdiff_oracle as O
    cfg = qm9_model_config(num_diffusion_timesteps=14, beta_end=2e-3)
    m, sd = _gpu_model(cfg, precision=precision)
    mols = _three_molecules()
    packed = driver.pack_batch(mols, driver.num_confs("2x"))
    N, n_steps = packed["atom_type"].shape[0], 14
    gen = torch.Generator().manual_seed(11)
    pos_init, noise = torch.randn(N, 3, generator=gen), torch.randn(n_steps, N, 3, generator=gen)
    kw = dict(n_steps=n_steps, step_lr=1e-6, w_global=1.0, global_start_sigma=0.5, clip=1000.0)
    pos, traj, ok = driver.sample_batch(m, packed, "cuda:0", kw, save_traj=True, pos_init=pos_init, noise=noise)
    assert ok.all() and traj.shape == (n_steps, N, 3)
    ref, ref_traj = O.langevin_dynamics_sample_diffusion(
        sd, cfg, t(packed["atom_type"]), pos_init, t(packed["bond_index"]), t(packed["bond_type"]), t(packed["batch"]),
        packed["num_graphs"], False, noise=noise, **kw)
    for mol, (off, n, g) in zip(mols, packed["spans"]):
        check_close("driver pos_gen %s" % mol["name"], pos[off:off + n * g], ref[off:off + n * g], precision)
    check_close("driver traj", traj, torch.stack(ref_traj), precision)


def test_driver_resamples_only_the_diverging_molecule():
    """One molecule's initial positions hold a NaN: its conformers are flagged per graph (agdiff_ws_t.nan_flag) and
    re-sampled with clip_local=20; the molecules packed with it keep, bit for bit, what a clean run gives them."""
    from agdiff_amd import driver, qm9_model_config
    cfg = qm9_model_config(num_diffusion_timesteps=10)
    m, _ = _gpu_model(cfg)
    mols = _three_molecules(seed=6)
    packed = driver.pack_batch(mols, driver.num_confs("2"))
    N = packed["atom_type"].shape[0]
    gen = torch.Generator().manual_seed(3)
    pos_init, noise = torch.randn(N, 3, generator=gen), torch.randn(10, N, 3, generator=gen)
    kw = dict(n_steps=10, step_lr=1e-6, w_global=1.0, global_start_sigma=0.5, clip=1000.0)
    clean, _, ok0 = driver.sample_batch(m, packed, "cuda:0", kw, pos_init=pos_init, noise=noise)
    assert ok0.all()
    off, n, g = packed["spans"][1]
    bad = pos_init.clone()
    bad[off + n + 2, 0] = float("nan")                      # second conformer of molecule 1
    logs = []
    pos, _, ok = driver.sample_batch(m, packed, "cuda:0", kw, pos_init=bad, noise=noise, log=logs.append)
    assert ok.all() and torch.isfinite(pos).all() and len(logs) == 1 and "1 of 3" in logs[0]
    keep = torch.ones(N, dtype=torch.bool)
    keep[off:off + n * g] = False
    assert torch.equal(pos[keep], clean[keep])
    assert not torch.equal(pos[~keep], clean[~keep])
    # the module API keeps the reference's contract: the whole call raises (dualenc.py:539-541)
    with pytest.raises(FloatingPointError):
        m.langevin_dynamics_sample_diffusion(t(packed["atom_type"]).cuda(), bad.cuda(), t(packed["bond_index"]).cuda(),
                                             t(packed["bond_type"]).cuda(), t(packed["batch"]).cuda(),
                                             packed["num_graphs"], extend_order=False, n_steps=3)


def test_topology_prepared_ahead_on_the_host_gives_the_same_samples(tmp_path, monkeypatch):
    """driver.prepare_batch builds a batch's BatchTopology on the CPU (no GPU call), sample_batch moves it over instead of
    building its own: the same bits.  run_job prepares the NEXT batch from a background thread while the current one samples."""
    import threading
    from agdiff_amd import driver, qm9_model_config
    cfg = qm9_model_config(num_diffusion_timesteps=10)
    m, _ = _gpu_model(cfg)
    mols = _three_molecules(seed=6)
    confs = driver.num_confs("2")
    packed = driver.pack_batch(mols, confs)
    N = packed["atom_type"].shape[0]
    gen = torch.Generator().manual_seed(3)
    pos_init, noise = torch.randn(N, 3, generator=gen), torch.randn(10, N, 3, generator=gen)
    kw = dict(n_steps=10, step_lr=1e-6, w_global=1.0, global_start_sigma=0.5, clip=1000.0)
    plain, _, ok0 = driver.sample_batch(m, packed, "cuda:0", kw, pos_init=pos_init, noise=noise)
    packed2, topo = driver.prepare_batch(m, mols, confs)
    assert topo is not None and torch.device(topo.device).type == "cpu" and topo.N == N and topo.quad_tgt.device.type == "cpu"
    ahead, _, ok1 = driver.sample_batch(m, packed2, "cuda:0", kw, pos_init=pos_init, noise=noise, topology=topo)
    assert ok0.all() and ok1.all() and torch.equal(plain, ahead)
    assert m._batch_cache[1] is topo and topo.quad_tgt.device.type == "cuda"        # moved, not rebuilt
    # a topology prepared for another batch is refused
    _, other = driver.prepare_batch(m, mols[:2], confs)
    with pytest.raises(ValueError):
        driver.sample_batch(m, packed, "cuda:0", kw, topology=other)
    # the job loop: three batches of one molecule each.  AGDIFF_PREPARE=thread: every preparation off the main thread
    calls, orig = [], driver.prepare_batch
    monkeypatch.setattr(driver, "prepare_batch",
                        lambda *a, **k: (calls.append(threading.current_thread() is threading.main_thread()), orig(*a, **k))[1])
    smallest = max(len(x["atom_type"]) * confs(x["num_refs"]) for x in mols)
    monkeypatch.setenv("AGDIFF_PREPARE", "thread")
    res = driver.run_job(m, mols, str(tmp_path / "thread"), confs, smallest, kw, "cuda:0", log=lambda *_: None)
    assert calls == [False] * len(calls) and len(calls) >= 2
    for x in mols:
        g = res["pos_gen_%d" % x["index"]]
        assert g.shape == (confs(x["num_refs"]), len(x["atom_type"]), 3) and np.isfinite(g).all()
    # AGDIFF_PREPARE=process: a worker PROCESS prepares every batch but the first (which the main thread prepares while the worker
    # boots); the topologies arrive pickled, are moved to the GPU and sampled: every molecule there, nothing fell back
    calls.clear()
    monkeypatch.setenv("AGDIFF_PREPARE", "process")
    logs = []
    res2 = driver.run_job(m, mols, str(tmp_path / "process"), confs, smallest, kw, "cuda:0", log=logs.append)
    assert calls == [True] and not [l for l in logs if "did not arrive" in l or "no worker" in l], logs
    for x in mols:
        g = res2["pos_gen_%d" % x["index"]]
        assert g.shape == (confs(x["num_refs"]), len(x["atom_type"]), 3) and np.isfinite(g).all()


def test_a_molecule_that_leaves_the_split_fp16_range_mid_job_is_resampled_in_split_bf16():
    """VERDICT r4 item 2 / ADVICE: the split-fp16 range watch (|hl| <= 255, ...) used to be polled only at the end of a driver
    run and raised AgdiffRangeError for the whole packed batch.  Now it is polled every nan_check_every steps in the driver's
    mode too, NaN-masked; the graphs that own the offending rows are taken out of the run like diverged ones and their
    molecule alone is sampled again with the model in split-bf16 (fp32's range).  Here one molecule carries an atom type whose
    GIN embedding row is 300: its conformers trip the watch at the first poll; its neighbours keep, bit for bit, what they get
    in a batch without that atom type; the job ends with finite pos_gen for every molecule."""
    from agdiff_amd import driver, get_model, qm9_model_config, synth
    from oracle import agdiff_oracle as O
    cfg = qm9_model_config(num_diffusion_timesteps=12)
    sd = O.synth_state_dict_for(cfg)
    for k in list(sd):
        if synth.canonical_key(k) == "encoder_local.node_emb.weight":
            sd[k] = sd[k].clone()
            sd[k][17] = 300.0
    m = get_model(cfg)
    m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    m = m.to("cuda:0").eval()
    mols = _three_molecules(seed=8)
    for mol in mols:
        assert not (mol["atom_type"] == 17).any()
    clean = driver.pack_batch(mols, driver.num_confs("2"))
    mols[1]["atom_type"] = mols[1]["atom_type"].copy()
    mols[1]["atom_type"][0] = 17
    packed = driver.pack_batch(mols, driver.num_confs("2"))
    N = packed["atom_type"].shape[0]
    gen = torch.Generator().manual_seed(4)
    pos_init, noise = torch.randn(N, 3, generator=gen), torch.randn(12, N, 3, generator=gen)
    kw = dict(n_steps=12, step_lr=1e-6, w_global=1.0, global_start_sigma=0.5, clip=1000.0, nan_check_every=4)
    ref, _, ok0 = driver.sample_batch(m, clean, "cuda:0", kw, pos_init=pos_init, noise=noise)
    assert ok0.all()
    before, logs = dict(driver.SAMPLE_STATS), []
    pos, traj, ok = driver.sample_batch(m, packed, "cuda:0", kw, pos_init=pos_init, noise=noise, log=logs.append, save_traj=True)
    off, n, g = packed["spans"][1]
    assert ok.all() and torch.isfinite(pos).all() and torch.isfinite(traj).all()
    assert driver.SAMPLE_STATS["range_trips"] - before["range_trips"] == g
    assert driver.SAMPLE_STATS["bf16x3_retries"] - before["bf16x3_retries"] == 1
    assert len(logs) == 1 and "split-bf16" in logs[0] and "1 of 3" in logs[0]
    assert (m.precision, m.precision_local) == ("f16x3", None)               # restored after the retry
    keep = torch.ones(N, dtype=torch.bool)
    keep[off:off + n * g] = False
    assert torch.equal(pos[keep], ref[keep])
    # the module API keeps raising for the whole call (reference contract: one call, one batch)
    from agdiff_amd import _lib
    with pytest.raises(_lib.AgdiffRangeError) as e:
        m.langevin_dynamics_sample_diffusion(t(packed["atom_type"]).cuda(), pos_init.cuda(), t(packed["bond_index"]).cuda(),
                                             t(packed["bond_type"]).cuda(), t(packed["batch"]).cuda(),
                                             packed["num_graphs"], extend_order=False, n_steps=12, nan_check_every=4)
    assert e.value.tensor == "hl" and sorted(e.value.graphs) == [2, 3]


@pytest.mark.parametrize("mode", ["auto", "radius", "off"])
def test_a_graph_that_goes_nan_mid_run_never_touches_its_neighbours(mode):
    """with raise_on_nan=False a diverged graph stays in the batch for the rest of the job.  One conformer's noise
    turns NaN at step 3 of 150: from that update on the graph is quarantined (finite placeholder positions, sticky flag,
    NaN in its trajectory rows), and every other graph ends bit for bit where a run without the fault puts it -- on the
    polynomial path (`auto`) and with the local edges through the MLP kernel (`radius`); with every edge through the MLP
    kernels (`off`: 16-edge tiles there span graph boundaries and sum with 0 / 1 masks, which a NaN would cross) within
    rounding."""
    from agdiff_amd import driver, get_model, qm9_model_config
    from oracle import agdiff_oracle as O
    cfg = qm9_model_config(num_diffusion_timesteps=150, beta_end=2e-5)
    sd = O.synth_state_dict_for(cfg)
    m = get_model(cfg)
    m.radius_poly = mode
    m.load_state_dict({k: v.clone() for k, v in sd.items()})
    m = m.to("cuda:0").eval()
    packed = driver.pack_batch(_three_molecules(seed=9), driver.num_confs("3"))
    N, G, n_steps = packed["atom_type"].shape[0], packed["num_graphs"], 150
    gen = torch.Generator().manual_seed(21)
    pos_init, noise = torch.randn(N, 3, generator=gen), torch.randn(n_steps, N, 3, generator=gen)
    a = [t(packed[k]).cuda() for k in ("atom_type", "bond_index", "bond_type", "batch")]
    kw = dict(extend_order=False, n_steps=n_steps, w_global=1.0, global_start_sigma=0.5, clip=1000.0, raise_on_nan=False)

    def run(nz):
        r = m.begin_sampling(a[0], pos_init.cuda(), a[1], a[2], a[3], G, noise=nz.cuda(), **kw)
        r.advance(r.remaining())
        torch.cuda.synchronize()
        return r.pos.cpu(), r.traj.cpu(), r.nan_graphs()
    clean, ctraj, cbad = run(noise)
    assert not cbad.any() and torch.isfinite(clean).all()
    ba = t(packed["batch"])
    victim = 4                                            # a graph in the middle of the batch
    rows = (ba == victim).nonzero().flatten()
    bad_noise = noise.clone()
    bad_noise[3, rows[1], 2] = float("nan")
    pos, traj, bad = run(bad_noise)
    assert bad.tolist() == [g == victim for g in range(G)]
    keep = ba != victim
    if mode == "off":
        # the one-list MLP CFConv sums a target's list over 16-edge tiles of the FULL edge list: when the victim's edge count
        # changes, its neighbours' lists sit at other tile offsets and associate their fp32 sums differently (last-bit noise,
        # as between any two batch compositions) -- no NaN, nothing beyond rounding
        assert float((pos[keep] - clean[keep]).abs().max()) < 1e-6 * float(clean.abs().max())
    else:
        assert torch.equal(pos[keep], clean[keep]) and torch.equal(traj[:, keep], ctraj[:, keep])
    assert torch.isfinite(pos).all()                      # the quarantined graph holds its placeholder, not NaN
    assert torch.equal(traj[:3, ~keep], ctraj[:3, ~keep]) and torch.isnan(traj[3:, ~keep]).all()
    # a NaN in the INITIAL positions is quarantined before the first forward
    bad_init = pos_init.clone()
    bad_init[rows[0], 0] = float("inf")
    r = m.begin_sampling(a[0], bad_init.cuda(), a[1], a[2], a[3], G, noise=noise.cuda(), **kw)
    r.advance(r.remaining())
    assert r.nan_graphs().tolist() == [g == victim for g in range(G)]
    assert torch.equal(r.pos.cpu()[keep], clean[keep]) or mode == "off"


def test_non_finite_node_features_of_one_molecule_never_reach_another_through_dead_quad_rows():
    """k_cfconv_quad runs the rows between a target's radius count and its quad's tile end with scale 0 but
    still gathers x[src]; rows the front kernel never wrote used to hold the workspace's zero init, i.e. NODE 0 -- an atom of
    molecule 0 -- and 0 x NaN through such a dead row poisons another molecule's aggregate.  Here node 0 carries an atom type
    whose SchNet embedding row is NaN (a degenerate checkpoint): its h and xs are NaN in every forward, before and after the
    update quarantines its molecule.  Every row of a target now names the target itself until it is written
    (Workspace.rad_src): the bystanders end bit for bit where a batch with a tame molecule 0 puts them.  The same run with the
    old zero init restored shows the hazard (so that this test cannot pass vacuously on a batch without dead rows)."""
    from agdiff_amd import _lib, driver, get_model, qm9_model_config, synth
    from oracle import agdiff_oracle as O
    cfg = qm9_model_config(num_diffusion_timesteps=20, beta_end=2e-5)
    sd = O.synth_state_dict_for(cfg)
    for k in list(sd):
        if synth.canonical_key(k) == "encoder_global.embedding.weight":
            sd[k] = sd[k].clone()
            sd[k][17] = float("nan")
    m = get_model(cfg)
    m.group_targets = 4                                     # quads (and with them k_cfconv_quad) on a small batch
    m.load_state_dict({k: v.clone() for k, v in sd.items()})
    m = m.to("cuda:0").eval()
    mols = _three_molecules(seed=9)
    assert not any((mol["atom_type"] == 17).any() for mol in mols)
    packed = driver.pack_batch(mols, driver.num_confs("3"))
    N, G, n_steps = packed["atom_type"].shape[0], packed["num_graphs"], 20
    gen = torch.Generator().manual_seed(5)
    pos_init, noise = torch.randn(N, 3, generator=gen), torch.randn(n_steps, N, 3, generator=gen)
    a = [t(packed[k]).cuda() for k in ("atom_type", "bond_index", "bond_type", "batch")]
    kw = dict(extend_order=False, n_steps=n_steps, w_global=1.0, global_start_sigma=0.5, clip=1000.0, raise_on_nan=False)

    def run(at, old_init=False):
        r = m.begin_sampling(at, pos_init.cuda(), a[1], a[2], a[3], G, noise=noise.cuda(), **kw)
        if old_init:
            r.ws.rad_src.zero_()
        r.advance(r.remaining())
        torch.cuda.synchronize()
        assert int(r.ws.variant_log.item()) & _lib.DEFINES["AGDIFF_VAR_CFCONV_NODE_QUAD"]
        return r.pos.cpu(), r.nan_graphs()
    clean, cbad = run(a[0])
    assert not cbad.any() and torch.isfinite(clean).all()
    ba = t(packed["batch"])
    off, n, g = packed["spans"][0]                          # molecule 0 = graphs 0 .. g - 1, node 0 included
    at_bad = a[0].clone()
    at_bad[off:off + n * g:n] = 17                          # the first atom of each of its conformers
    pos, bad = run(at_bad)
    assert bad[:g].all() and not bad[g:].any()
    keep = ba >= g
    assert torch.equal(pos[keep], clean[keep]) and torch.isfinite(pos).all()
    _, bad_old = run(at_bad, old_init=True)
    assert bad_old[g:].any(), "the fixture has no dead quad row that pointed at node 0: the check above proves nothing"


def test_all_gather_path_on_one_gpu_nccl_world1():
    """SURVEY §8e on hardware: process group 'nccl' (= RCCL) with one rank; StepAllGather as the sampler's on_step.
    Every step's gathered shard equals the positions of that step (the snapshot is taken on the compute stream before
    the next step overwrites them), the NaN flag travels with it, and the sharded driver path gives the unsharded
    result."""
    import torch.distributed as dist
    from agdiff_amd import driver, qm9_model_config, synth
    from agdiff_amd.dist import StepAllGather, sample_batch_sharded
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        cfg = qm9_model_config(num_diffusion_timesteps=8)
        m, _ = _gpu_model(cfg)
        b = synth.make_packed_batch("qm9", 3, 2, seed=9)
        at, bi, bt, ba = [t(b[k]).cuda() for k in ("atom_type", "bond_index", "bond_type", "batch")]
        N = at.shape[0]
        gen = torch.Generator().manual_seed(2)
        pos_init, noise = torch.randn(N, 3, generator=gen).cuda(), torch.randn(6, N, 3, generator=gen).cuda()
        gather = StepAllGather(N, dev)
        seen = []

        def on_step(k, i, pos):
            gather(k, i, pos, run.ws.nan_flag)
            parts, any_nan = gather.result()               # waits for the side stream; test only
            seen.append((parts[0].clone(), any_nan))
        run = m.begin_sampling(at, pos_init, bi, bt, ba, b["num_graphs"], False, n_steps=6, w_global=1.0,
                               global_start_sigma=0.5, noise=noise, on_step=on_step)
        run.advance(6)
        pos, traj = run.finish()
        assert gather.calls == 6 and len(seen) == 6
        for k in range(6):
            assert torch.equal(seen[k][0].cpu(), traj[k]) and not seen[k][1]
        assert torch.equal(seen[-1][0], pos)
        # NaN flag propagation
        bad = pos_init.clone()
        bad[1, 1] = float("nan")
        g2 = StepAllGather(N, dev)
        run2 = m.begin_sampling(at, bad, bi, bt, ba, b["num_graphs"], False, n_steps=2, raise_on_nan=False)
        run2.on_step = lambda k, i, p: g2(k, i, p, run2.ws.nan_flag)
        run2.advance(2)
        assert g2.result()[1] and bool(run2.nan_graphs()[0]) and not bool(run2.nan_graphs()[1:].any())
        # the sharded driver path (one shard = the whole batch here) equals the plain one
        mols = _three_molecules(seed=8)
        packed = driver.pack_batch(mols, driver.num_confs("2"))
        n2 = packed["atom_type"].shape[0]
        p0, nz = torch.randn(n2, 3, generator=gen), torch.randn(5, n2, 3, generator=gen)
        kw = dict(n_steps=5, step_lr=1e-6, w_global=1.0, global_start_sigma=0.5, clip=1000.0)
        a, _, oka = driver.sample_batch(m, packed, dev, kw, pos_init=p0, noise=nz)
        bb, _, okb = sample_batch_sharded(m, packed, dev, kw, pos_init=p0, noise=nz)
        assert oka.all() and okb.all() and torch.equal(a, bb)
    finally:
        dist.destroy_process_group()


def test_bench_two_ranks_strong_scaling_rehearsal_on_one_gpu():
    """The command the driver's SCALE run launches (python -m torch.distributed.run ... bench.py --gpus N), with two ranks,
    as child processes: both compute on this box's one GPU and meet over gloo (--rehearse-on-one-gpu; RCCL refuses two ranks
    on one device).  The sharded job -- global batches cut into per-rank graph ranges, one all-gather per step and batch --
    runs through the real kernels and rank 0 prints one strong-scaling line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-extra", "--no-cpu-baseline", "--rehearse-on-one-gpu"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and d["config"]["conformers_total"] == 108874
    nb = len(d["config"]["batches"])
    assert d["config"]["all_gather_calls_rank0"] == 3 * nb          # one per step (1 warm-up + 2 timed) and global batch
    assert "rehearsal" in d["config"]


@pytest.mark.parametrize("kind,mols,copies,min_mean_deg", [("qm9", 40, 30, 14.0), ("large", 256, 1, 30.0)])
def test_full_size_properties_other_configs(kind, mols, copies, min_mean_deg):
    """BASELINE.json configs[1] (QM9-shaped, uncapped) and configs[4]'s per-GPU share (200-atom molecules x 256,
    32-cap active) at sizes the oracle cannot check: bitwise run-to-run determinism, CSR invariants, mirror-pair
    structure of the canonical list, centring, finiteness."""
    from agdiff_amd import drugs_model_config, qm9_model_config, synth
    cfg = (qm9_model_config if kind == "qm9" else drugs_model_config)(num_diffusion_timesteps=50, beta_end=2e-5)
    m, _ = _gpu_model(cfg)
    b = synth.make_packed_batch(kind, mols, copies, seed=91)
    at, bi, bt, ba = [t(b[k]).cuda() for k in ("atom_type", "bond_index", "bond_type", "batch")]
    gen = torch.Generator().manual_seed(4)
    pos_init = torch.randn(at.shape[0], 3, generator=gen).cuda()
    noise = torch.randn(3, at.shape[0], 3, generator=gen).cuda()
    kw = dict(extend_order=False, n_steps=3, w_global=1.0, global_start_sigma=0.5, clip=1000.0, noise=noise)
    p1, tr1 = m.langevin_dynamics_sample_diffusion(at, pos_init, bi, bt, ba, b["num_graphs"], **kw)
    p2, tr2 = m.langevin_dynamics_sample_diffusion(at, pos_init, bi, bt, ba, b["num_graphs"], **kw)
    assert torch.equal(p1, p2) and torch.equal(torch.stack(tr1), torch.stack(tr2))
    wsf, topo = m._batch_cache[2], m._batch_cache[1]
    # what the fused front (agdiff_sampler_front) left for the last forward: radius rows and the segmented canonical list
    cnt = wsf.rad_cnt.cpu().numpy()
    ccnt = wsf.canon_counter.cpu().numpy()           # [parity of the last graph build] = live length, the other 0
    assert cnt.max() <= 33 and ccnt.min() == 0 < ccnt.max()
    live = np.arange(int(ccnt.max()))
    cs, cd, cp, cm = [x.cpu().numpy()[live] for x in (wsf.c_src, wsf.c_dst, wsf.c_pos, wsf.c_mir)]
    RS = 48
    assert np.array_equal(cp // RS, cd) and np.all(cp % RS < cnt[cd])                 # an entry's row belongs to its target
    mk = cm >= 0
    assert np.array_equal(cm[mk] // RS, cs[mk]) and np.all(cs[mk] < cd[mk])            # its mirror's row to its source
    rsrc = wsf.rad_src.cpu().numpy()
    assert np.array_equal(rsrc[cp], cs) and np.array_equal(rsrc[cm[mk]], cd[mk])
    cover = np.bincount(np.concatenate([cp, cm[mk]]), minlength=topo.N * RS).reshape(topo.N, RS)
    assert all(np.array_equal(cover[i, :cnt[i]], np.ones(cnt[i], int)) and cover[i, cnt[i]:].sum() == 0 for i in range(0, topo.N, 7))
    C_front, Ec_front = int(ccnt.max()), int(cnt.sum())
    # the full destination-sorted list of the same (final) positions, as forward() builds it
    m(at, p1, bi, bt, ba, None, extend_order=False)
    ws = m._batch_cache[2]
    E, C = int(ws.num_edges.item()), C_front
    ip = ws.in_ptr.cpu().numpy()
    indeg = np.diff(ip)
    dst, src = ws.e_dst[:E].cpu().numpy(), ws.e_src[:E].cpu().numpy()
    assert E <= topo.max_edges and ip[-1] == E and indeg.max() <= topo.max_in_degree
    assert np.all(np.diff(dst) >= 0) and np.array_equal(np.bincount(dst, minlength=topo.N), indeg)
    assert np.all(np.diff(src)[np.diff(dst) == 0] > 0)
    assert indeg.mean() > min_mean_deg
    batch = b["batch"]
    assert np.array_equal(batch[src], batch[dst])                    # no edge crosses a molecule
    # the denoising loop with filter polynomials keeps only RADIUS edges in the canonical list (agdiff_graph_build_ex)
    Ec = Ec_front if m.packed().poly_kt > 0 else E
    if kind == "qm9":
        assert 2 * C == Ec                                           # uncapped: every edge has its mirror
    else:
        assert Ec // 2 < C < Ec
    cen = torch.zeros(b["num_graphs"], 3, device="cuda").index_add_(0, ba, p1)
    assert float(cen.abs().max()) < 2e-3 and torch.isfinite(p1).all()
