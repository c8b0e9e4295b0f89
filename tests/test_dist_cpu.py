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
