"""CPU, world_size 2, gloo: graph sharding and the per-step all-gather of position shards."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from agdiff_amd import synth
from agdiff_amd.dist import StepAllGather, shard_graphs, take_graph_range


def test_shard_graphs_balanced_and_contiguous():
    rng = np.random.default_rng(0)
    sizes = rng.integers(10, 90, size=57)
    loc = sizes * 8
    for world in (1, 2, 4, 8):
        parts = shard_graphs(sizes, loc, world)
        assert parts[0][0] == 0 and parts[-1][1] == len(sizes)
        assert all(parts[r][1] == parts[r + 1][0] for r in range(world - 1))
        w = sizes * np.minimum(sizes - 1, 33) + loc
        loads = [w[a:b].sum() for a, b in parts]
        assert max(loads) <= w.sum() / world + w.max()


def test_take_graph_range_rebases():
    b = synth.make_packed_batch("qm9", 4, 2, seed=3)
    at, bi, bt, ba = b["atom_type"], b["bond_index"], b["bond_type"], b["batch"]
    a2, i2, t2, b2, lo, hi = take_graph_range(at, bi, bt, ba, 3, 6)
    assert b2.min() == 0 and b2.max() == 2 and a2.shape[0] == hi - lo
    assert i2.min() >= 0 and i2.max() < a2.shape[0]
    assert np.array_equal(at[lo:hi], a2)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_local = 5 + 3 * rank
        ag = StepAllGather(n_local, "cpu")
        for step in range(3):
            pos = torch.full((n_local, 3), float(10 * rank + step))
            flag = torch.tensor([1 if (rank == 1 and step == 2) else 0], dtype=torch.int32)
            ag(step, step, pos, flag)
        parts, any_nan = ag.result()
        ok = all(parts[r].shape == (5 + 3 * r, 3) and bool((parts[r] == 10 * r + 2).all()) for r in range(world))
        out[rank] = int(ok and any_nan and ag.calls == 3)
    finally:
        dist.destroy_process_group()


def test_step_all_gather_world2_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert out[0] == 1 and out[1] == 1


def test_shards_reassemble_to_the_unsharded_topology():
    """shard_graphs + take_graph_range on a real packed batch: the shards' static topologies (graph offsets, local
    edge lists in reference order, their types, capacity bounds) concatenate to the unsharded batch's topology."""
    from agdiff_amd.dist import shard_of
    from agdiff_amd.topology import BatchTopology
    b = synth.make_packed_batch("drugs", 5, lambda rng: int(rng.integers(1, 4)), seed=17)
    full = BatchTopology(b["atom_type"], b["bond_index"], b["bond_type"], b["batch"], device="cpu")
    for world in (2, 3, 8, 16):               # 16 > number of graphs for some seeds: empty shards are allowed
        gp, src, dst, typ, at, me = [np.zeros(1, dtype=np.int64)], [], [], [], [], 0
        covered = 0
        for rank in range(world):
            mine, (g0, g1), (lo, hi) = shard_of(b, rank, world)
            assert g0 == covered
            covered = g1
            if mine is None:
                assert g1 == g0
                continue
            tp = BatchTopology(mine["atom_type"], mine["bond_index"], mine["bond_type"], mine["batch"], device="cpu")
            assert tp.G == g1 - g0 and tp.N == hi - lo
            gp.append(tp.graph_ptr.numpy()[1:].astype(np.int64) + lo)
            src.append(tp.loc_src.numpy().astype(np.int64) + lo); dst.append(tp.loc_dst.numpy().astype(np.int64) + lo)
            typ.append(tp.loc_type.numpy()); at.append(tp.atom_type.numpy())
            me += tp.max_edges
        assert covered == b["num_graphs"]
        assert np.array_equal(np.concatenate(gp), full.graph_ptr.numpy())
        assert np.array_equal(np.concatenate(src), full.loc_src.numpy()) and np.array_equal(np.concatenate(dst), full.loc_dst.numpy())
        assert np.array_equal(np.concatenate(typ), full.loc_type.numpy()) and np.array_equal(np.concatenate(at), full.atom_type.numpy())
        assert me == full.max_edges


def _shard_worker(rank, world, port, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_driver_cpu import _FakeSampler, _mols
    from agdiff_amd import driver
    from agdiff_amd.dist import sample_batch_sharded
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mols = _mols(5)
        mols[3]["atom_type"] = mols[3]["atom_type"].copy()
        mols[3]["atom_type"][:] = 9
        packed = driver.pack_batch(mols, driver.num_confs("3"))
        N = packed["atom_type"].shape[0]
        g = torch.Generator().manual_seed(5)
        p0 = torch.randn(N, 3, generator=g)
        model = _FakeSampler(nan_type=9)
        pos, traj, ok = sample_batch_sharded(model, packed, "cpu", dict(n_steps=3), save_traj=True, log=lambda s: None,
                                             pos_init=p0)
        ref_model = _FakeSampler(nan_type=-1)                       # unsharded, nothing diverges
        ref, _, _ = driver.sample_batch(ref_model, packed, "cpu", dict(n_steps=3), pos_init=p0, log=lambda s: None)
        off, n, gg = packed["spans"][3]
        keep = torch.ones(N, dtype=torch.bool)
        keep[off:off + n * gg] = False
        good = bool(ok.all()) and torch.equal(pos[keep], ref[keep]) and bool(torch.isfinite(pos).all())
        if rank == 0:                       # the trajectories go to rank 0 only (it writes them)
            good = good and traj.shape == (3, N, 3) and torch.equal(traj[-1][keep], ref[keep])
        else:
            good = good and traj is None
        # first attempt: this rank's share of the 15 graphs; second: its share of molecule 3's three conformers (with more
        # ranks than conformers a rank has no share: it only takes part in the collectives)
        c = model.calls
        good = good and len(c) in (1, 2) and c[0][1] is None and (len(c) == 1 or c[1][1] == 20) and (len(c) == 2 or world > 3)
        out[rank] = (int(good), c[0][0] if c else 0, c[1][0] if len(c) > 1 else 0)
    finally:
        dist.destroy_process_group()


def test_sharded_sampling_matches_unsharded_world2_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_shard_worker, args=(2, port, out), nprocs=2, join=True)
    assert out[0][0] == 1 and out[1][0] == 1
    assert out[0][1] + out[1][1] == 15 and out[0][2] + out[1][2] == 3      # graphs of both attempts, split over the ranks


def _shard_fault_worker(rank, world, port, out):
    """(a) more ranks than graphs + step_indices shorter than n_steps: the idle rank issues len(step_indices)
    collectives, not n_steps; (b) a rank that raises inside begin_sampling must not leave the other in all_gather."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_driver_cpu import _FakeSampler, _mols
    from agdiff_amd import driver
    from agdiff_amd.dist import sample_batch_sharded
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=60))
    try:
        one = driver.pack_batch(_mols(1), driver.num_confs("1"))          # ONE graph over two ranks: rank 1 idles
        model = _FakeSampler()
        run0 = model.begin_sampling
        model.begin_sampling = lambda *a, step_indices=None, **k: run0(*a, **dict(k, n_steps=len(step_indices)))
        pos, _, ok = sample_batch_sharded(model, one, "cpu", dict(n_steps=5000, step_indices=[9, 5, 1]), log=lambda s: None)
        a_ok = bool(ok.all()) and bool(torch.isfinite(pos).all())
        packed = driver.pack_batch(_mols(4), driver.num_confs("2"))
        bad = _FakeSampler()
        if rank == 1:
            def boom(*a, **k):
                raise MemoryError("rank 1 ran out of memory")
            bad.begin_sampling = boom
        msg = ""
        try:
            sample_batch_sharded(bad, packed, "cpu", dict(n_steps=3), log=lambda s: None)
        except Exception as e:
            msg = "%s: %s" % (type(e).__name__, e)
        out[rank] = (int(a_ok), msg)
    finally:
        dist.destroy_process_group()


def test_sharded_sampling_idle_rank_and_failing_rank_world2_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_shard_fault_worker, args=(2, port, out), nprocs=2, join=True)
    assert out[0][0] == 1 and out[1][0] == 1
    assert out[1][1].startswith("MemoryError") and "rank(s) [1] failed" in out[0][1] and "MemoryError" in out[0][1]


def test_sharded_sampling_world8_gloo():
    """the sharded driver path with EIGHT ranks (the node bench.py --gpus 8 runs on): 15 graphs over 8
    ranks, then the diverged molecule's 3 conformers over 8 ranks (five ranks without a graph take part in the collectives
    only); one graph over 8 ranks with a step_indices list shorter than n_steps; a rank that raises while the others sample."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_shard_worker, args=(8, port, out), nprocs=8, join=True)
    assert all(out[r][0] == 1 for r in range(8)), dict(out)
    assert sum(out[r][1] for r in range(8)) == 15 and sum(out[r][2] for r in range(8)) == 3
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = mgr.dict()
    mp.spawn(_shard_fault_worker, args=(8, port, out), nprocs=8, join=True)
    assert all(out[r][0] == 1 for r in range(8)), dict(out)
    assert out[1][1].startswith("MemoryError")
    assert all("rank(s) [1] failed" in out[r][1] and "MemoryError" in out[r][1] for r in range(8) if r != 1)


def test_bench_launcher_spawns_ranks_and_fails_cleanly():
    """`python bench.py --gpus N` (no torchrun) starts N fresh rank processes itself -- the parent never
    initialises a GPU -- relays rank 0's JSON line, and on a node with fewer GPUs than ranks it stops with a device-count
    message instead of a launcher hint.  The wiring (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*) is exercised over gloo."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--launcher-selftest"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec == {"selftest": True, "rccl_ranks": 3, "sum_of_ranks": 6, "local_rank_env": "0"}
    # a rank that dies before its first collective would leave the others waiting for ever: the launcher stops them and fails
    import time
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--launcher-selftest"],
                         capture_output=True, text=True, timeout=200, env=dict(env, AGDIFF_SELFTEST_FAIL_RANK="1"))
    assert out.returncode == 1 and "rank 1 exited with code 7" in out.stderr and out.stdout.strip() == "", (out.returncode, out.stderr[-800:])
    assert time.time() - t0 < 150
    if torch.cuda.device_count() < 2:
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                             capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode == 2 and "GPU(s) visible" in out.stderr and "nothing was measured" in out.stderr
        assert out.stdout.strip() == ""


# ---------------------------------------------------------------------------------------------------- bench.py's job loop
class _StubRun:
    """What bench.run_job_steps reads of a LangevinRun."""

    class _Topo:
        def __init__(self, n):
            self.N = n

    def __init__(self, pos):
        self.pos, self.topo = pos, _StubRun._Topo(pos.shape[0])

    def check_nan(self):
        pass


def _stub_timed(model, dev, b, cfg, W, K, schedule, skip, save_traj, seed, rank, use_dist, profile=False):
    """Stand-in for bench.timed_run on CPU: W + K "denoising steps" that move the positions deterministically and call the
    sampler's on_step hook exactly as LangevinRun does -- with the REAL StepAllGather behind it."""
    n = int(np.asarray(b["atom_type"]).shape[0])
    pos = torch.full((n, 3), float(seed))
    gather = StepAllGather(n, "cpu") if use_dist else None
    for k in range(W + K):
        pos = pos + 1.0 + rank
        if gather is not None:
            gather(k, k, pos, torch.zeros(1, dtype=torch.int32))
    return 1e-3 * K, _StubRun(pos), 1.0, gather, None


def _bench_job_worker(rank, world, port, out, strong):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import bench
    from agdiff_amd import driver
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)
        mols = []
        for i in range(6):
            at_, r_, c_, t_ = synth.random_molecule(rng, synth.sample_n_atoms(rng, "qm9"))
            mols.append(dict(atom_type=at_, edge_index=np.stack([r_, c_]), edge_type=t_, num_refs=2 + i, name="m%d" % i, index=i))
        confs_of = driver.num_confs("2x")
        batches = driver.plan_batches(mols, confs_of, 400)
        assert len(batches) >= 2
        gcalls = [0]
        W, K = 2, 5
        tot, G_local, recs, gf, last, prof = bench.run_job_steps(
            None, None, batches, confs_of, "saturated", True, W, K, False, dev="cpu", rank=rank, world=world, seed=11,
            save_traj=False, use_dist=True, strong=strong, gcalls=gcalls, timed=_stub_timed, edges_of=lambda r: 0,
            tiles_of=lambda r: None)
        G_job = sum(confs_of(m["num_refs"]) for m in mols)
        ok = gcalls[0] == len(batches) * (W + K) and len(recs) == len(batches)       # ONE gather per step per batch per rank
        ok = ok and (G_local == G_job if not strong else 0 <= G_local < G_job)      # (a batch may hold fewer graphs than ranks)
        out[rank] = (int(ok), G_local, gcalls[0])
    finally:
        dist.destroy_process_group()


def test_bench_job_loop_gathers_every_step_world2_gloo():
    """bench.py's own per-batch loop (run_job_steps) over gloo with two ranks, weak AND strong scaling: every rank issues
    exactly one all-gather per denoising step per batch (north_star) and the gathered shard of each
    rank equals its positions (asserted inside run_job_steps)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    for strong in (False, True):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(_bench_job_worker, args=(2, port, out, strong), nprocs=2, join=True)
        assert out[0][0] == 1 and out[1][0] == 1, (strong, dict(out))
        assert out[0][2] == out[1][2]
        if strong:
            G = sum(2 * (2 + i) for i in range(6))
            assert out[0][1] + out[1][1] == G


def test_bench_job_loop_world8_gloo():
    """... and with eight ranks, strong scaling (bench.py's default with more than one rank): some batches hold fewer graphs than
    ranks -- those ranks idle through the batch but issue its W + K collectives --, every rank's count agrees, the shards add up
    to the job."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_bench_job_worker, args=(8, port, out, True), nprocs=8, join=True)
    assert all(out[r][0] == 1 for r in range(8)), dict(out)
    assert len(set(out[r][2] for r in range(8))) == 1
    assert sum(out[r][1] for r in range(8)) == sum(2 * (2 + i) for i in range(6))


def _run_job_shard_worker(rank, world, port, out, tmp):
    """driver.run_job(shard=True): every rank prepares ITS graph range of the next batch in the background
    (driver.prepare_batch(rank, world)) and hands the topology to the sampler of that range."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_driver_cpu import _FakeSampler, _mols
    from agdiff_amd import driver
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=120))
    try:
        class Sampler(_FakeSampler):
            def __init__(self):
                super().__init__()
                self.prepared, self.given = [], []

            def prepare_topology(self, atom_type, bond_index, bond_type, batch, num_graphs, extend_order=False, device="cpu"):
                import threading
                tp = ("topology", int(np.asarray(atom_type).shape[0]), int(num_graphs), threading.current_thread() is threading.main_thread())
                self.prepared.append(tp)
                return tp

            def begin_sampling(self, at, pos_init, bi, bt, batch, G, extend_order, topology=None, **kw):
                self.given.append((topology, int(at.shape[0]), int(G)))
                return super().begin_sampling(at, pos_init, bi, bt, batch, G, extend_order, **kw)

        mols = _mols(5)
        for i, m in enumerate(mols):
            m["index"] = i
        confs = driver.num_confs("2")
        biggest = max(len(m["atom_type"]) * confs(m["num_refs"]) for m in mols)
        model = Sampler()
        res = driver.run_job(model, mols, os.path.join(tmp, "out"), confs, biggest, dict(n_steps=2), "cpu", rank=rank, world=world,
                             shard=True, log=lambda *_: None)
        good = len(model.prepared) >= 2 and not any(p[3] for p in model.prepared)          # several batches, none prepared on the main thread
        # every sampler call got the topology prepared for exactly its range (atoms and graphs agree)
        good = good and len(model.given) == len(model.prepared)
        for (tp, n, g) in model.given:
            good = good and tp is not None and tp[1] == n and tp[2] == g
        if rank == 0:
            good = good and all(("pos_gen_%d" % m["index"]) in res for m in mols)
        else:
            good = good and res is None
        out[rank] = int(good)
    finally:
        dist.destroy_process_group()


def test_run_job_sharded_prepares_each_ranks_range_in_the_background_world2_gloo(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_run_job_shard_worker, args=(2, port, out, str(tmp_path)), nprocs=2, join=True)
    assert out[0] == 1 and out[1] == 1
