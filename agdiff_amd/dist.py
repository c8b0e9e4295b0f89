"""Data-parallel sampling: one process per GPU, molecules (graphs) sharded across ranks.

Graphs are independent everywhere on the path (radius graph per `batch` id, eval-mode BatchNorm,
per-graph centring, i.i.d. noise: SURVEY.md §8e), so the only exchange is the one BASELINE.json's
north_star asks for: an all-gather of the shard positions (+ NaN flag) at the end of each
denoising step.  It is issued on a side stream from a snapshot taken on the compute stream, because no
rank needs remote positions for its next step: compute never waits for it (only, in principle, for the
previous step's collective to release the snapshot buffer).  `backend="nccl"` is RCCL on ROCm;
message sizes are tens of KB to ~0.6 MB per rank -> latency-bound, one collective per step.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_graphs(graph_sizes, graph_local_edges, world_size):
    """Contiguous graph ranges per rank, balanced by the edge-count proxy
    sum n_atoms * min(n_atoms - 1, 33) + local edges.  Returns [(g_begin, g_end)] * world_size."""
    n = np.asarray(graph_sizes, dtype=np.int64)
    w = n * np.minimum(n - 1, 33) + np.asarray(graph_local_edges, dtype=np.int64)
    cum = np.concatenate([[0], np.cumsum(w)])
    total = cum[-1]
    bounds = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        g = int(np.searchsorted(cum, target, side="left"))
        g = min(max(g, bounds[-1]), len(n))
        bounds.append(g)
    bounds.append(len(n))
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


def take_graph_range(atom_type, bond_index, bond_type, batch, g0, g1):
    """Slice graphs [g0, g1) out of a packed batch (numpy int64 arrays), re-basing node and graph ids."""
    node_sel = np.nonzero((batch >= g0) & (batch < g1))[0]
    lo = int(node_sel[0]) if node_sel.size else 0
    hi = int(node_sel[-1]) + 1 if node_sel.size else 0
    esel = (bond_index[0] >= lo) & (bond_index[0] < hi)
    return (atom_type[lo:hi], bond_index[:, esel] - lo, bond_type[esel], batch[lo:hi] - g0, lo, hi)


class StepAllGather:
    """Per-step all-gather of [positions | nan flag] shards, padded to the largest shard."""

    def __init__(self, n_local, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        sizes = torch.zeros(self.world, dtype=torch.int64, device=device)
        sizes[self.rank] = n_local
        dist.all_reduce(sizes, group=group)
        self.sizes = [int(x) for x in sizes.cpu()]
        self.pad = max(self.sizes) * 3 + 1
        self.stage = torch.zeros(self.pad, dtype=torch.float32, device=device)
        self.gathered = torch.zeros(self.world * self.pad, dtype=torch.float32, device=device)
        self.n_local = n_local
        self.is_cuda = torch.device(device).type == "cuda"
        self.side = torch.cuda.Stream(device=device) if self.is_cuda else None
        self.calls = 0
        self.prev_done = None

    def __call__(self, k, i, pos, nan_flag=None):
        """on_step hook of LangevinRun: snapshot `pos` on the compute stream (the next step updates it in place),
        then launch the collective from the snapshot on the side stream."""
        if self.is_cuda:
            cur = torch.cuda.current_stream()
            if self.prev_done is not None:
                cur.wait_event(self.prev_done)        # the previous collective has finished reading `stage`
            self._snapshot(pos, nan_flag)
            ev = torch.cuda.Event()
            ev.record(cur)
            with torch.cuda.stream(self.side):
                self.side.wait_event(ev)
                dist.all_gather_into_tensor(self.gathered, self.stage, group=self.group)
                self.prev_done = torch.cuda.Event()
                self.prev_done.record(self.side)
        else:
            self._snapshot(pos, nan_flag)
            dist.all_gather_into_tensor(self.gathered, self.stage, group=self.group)
        self.calls += 1

    def _snapshot(self, pos, nan_flag):
        self.stage[: self.n_local * 3].copy_(pos.reshape(-1), non_blocking=True)
        if nan_flag is not None:       # agdiff_ws_t.nan_flag: element 0 is the "any graph" flag
            self.stage[-1:].copy_(nan_flag.reshape(-1)[:1].to(torch.float32), non_blocking=True)

    def wait(self):
        if self.is_cuda:
            torch.cuda.current_stream().wait_stream(self.side)

    def result(self):
        """(list of per-rank [n_r, 3] position tensors, any_nan) from the last gather."""
        self.wait()
        g = self.gathered.view(self.world, self.pad)
        return [g[r, : self.sizes[r] * 3].view(-1, 3) for r in range(self.world)], bool((g[:, -1] != 0).any())


def graph_weights(packed):
    """(graph_sizes [G], local edges per graph [G]) of a packed batch: the inputs of shard_graphs."""
    ba = np.asarray(packed["batch"], dtype=np.int64)
    G = int(packed["num_graphs"])
    sizes = np.bincount(ba, minlength=G)
    loc = np.bincount(ba[np.asarray(packed["bond_index"])[0]], minlength=G) if np.asarray(packed["bond_type"]).size else np.zeros(G, dtype=np.int64)
    return sizes, loc


def shard_of(packed, rank, world):
    """This rank's contiguous graph range of one packed batch: (dict like `packed` for the range or None when the
    range is empty, (g0, g1), (node_lo, node_hi))."""
    sizes, loc = graph_weights(packed)
    g0, g1 = shard_graphs(sizes, loc, world)[rank]
    if g1 <= g0:
        return None, (g0, g1), (0, 0)
    at, bi, bt, ba, lo, hi = take_graph_range(np.asarray(packed["atom_type"]), np.asarray(packed["bond_index"]),
                                              np.asarray(packed["bond_type"]), np.asarray(packed["batch"]), g0, g1)
    return dict(atom_type=at, bond_index=bi, bond_type=bt, batch=ba, num_graphs=g1 - g0), (g0, g1), (lo, hi)


def sample_batch_sharded(model, packed, device, sampler_kwargs, save_traj=False, max_retry=2, log=print, group=None,
                         pos_init=None, noise=None, topology=None):
    """driver.sample_batch for one packed batch sharded over the ranks of `group` by contiguous graph ranges
    (SURVEY §8e): every rank samples its range, the shards' positions (+ NaN flag) are all-gathered after every
    denoising step (StepAllGather), and the last gather is the job's result on every rank.  Molecules in which a NaN
    appeared (on any rank) are re-sampled, sharded again, with clip_local=20 (test.py:143-181).
    `pos_init` [N,3] / `noise` [steps,N,3] for the WHOLE batch replace the first attempt's draws (tests).
    Returns (pos [N,3] cpu, traj or None, ok [num molecules]) like driver.sample_batch; pos and ok are identical on all
    ranks, the trajectories [steps, N, 3] are gathered to rank 0 only (None elsewhere).  A rank that raises while the
    others sample keeps issuing its collectives, then every rank raises."""
    from .driver import SAMPLE_STATS, _arithmetic, _sort_results, subset_batch
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(device)
    spans = packed["spans"]
    n_mol, N = len(spans), packed["atom_type"].shape[0]
    pos_out = torch.full((N, 3), float("nan"))
    traj_out = None
    ok = np.zeros(n_mol, dtype=bool)
    # collectives per pass = steps LangevinRun will take (epsnet.py: len(step_indices) if given, else n_steps): a rank
    # without graphs must issue exactly as many gathers as the ranks that sample
    si = sampler_kwargs.get("step_indices")
    n_steps = len(si) if si is not None else int(sampler_kwargs.get("n_steps", 5000))
    # passes still to run, as in driver.sample_batch: (molecule slots, clip_local, split-bf16?, attempts counted); every rank
    # takes the same decisions (the flags they rest on are gathered)
    passes = [(list(range(n_mol)), None, False, 0)]
    first = True
    while passes:
        todo, clip_local, wide, tries = passes.pop(0)
        sub = packed if len(todo) == n_mol else subset_batch(packed, todo)
        mine, (g0, g1), (lo, hi) = shard_of(sub, rank, world)
        gather = StepAllGather(hi - lo, device, group)
        empty, zero = torch.zeros(0, 3, device=device), torch.zeros(1, dtype=torch.int32, device=device)
        err, traj, bad_local, range_local = None, None, np.zeros(0, dtype=bool), []
        if mine is not None:
            try:
                p0 = pos_init[lo:hi].to(device) if (first and pos_init is not None) else torch.randn(hi - lo, 3).to(device)
                with _arithmetic(model, wide):
                    # (`topology`: this rank's range of the whole batch, prepared ahead by driver.prepare_batch: first attempt only)
                    extra = {"topology": topology} if (topology is not None and sub is packed) else {}
                    topology = None
                    run = model.begin_sampling(T(mine["atom_type"]), p0, T(mine["bond_index"]), T(mine["bond_type"]),
                                               T(mine["batch"]), mine["num_graphs"], False, clip_local=clip_local,
                                               save_traj=save_traj, raise_on_nan=False,
                                               noise=(noise[:, lo:hi] if (first and noise is not None) else None),
                                               **extra, **sampler_kwargs)
                    run.on_step = lambda k, i, pos: gather(k, i, pos, run.ws.nan_flag)
                    run.advance(run.remaining())
                    _, traj = run.finish()
                bad_local = run.nan_graphs().numpy()
                range_local = [g0 + int(g) for g in getattr(run, "range_graphs", ())]      # graph ids of `sub`
            except Exception as e:            # (AgdiffLimitError, out of memory, ...): the other ranks are inside the
                err = e                       # per-step collectives -- keep this rank's count whole, fail together below
        # a rank without graphs (more ranks than graphs) or one that failed takes part in the collectives only
        filler = empty if (mine is None or err is None) else torch.zeros(hi - lo, 3, device=device)
        for k in range(gather.calls, n_steps):
            gather(k, k, filler if mine is not None else empty, zero)
        errs = [None] * world
        dist.all_gather_object(errs, None if err is None else "%s: %s" % (type(err).__name__, err), group=group)
        if any(e is not None for e in errs):
            if err is not None:
                raise err
            raise RuntimeError("sample_batch_sharded: rank(s) %s failed: %s"
                               % ([r for r, e in enumerate(errs) if e is not None], [e for e in errs if e is not None]))
        parts, _ = gather.result()
        pos = torch.cat([p.cpu() for p in parts], dim=0)            # rank order == graph order
        bad_all = [None] * world
        dist.all_gather_object(bad_all, (bad_local.tolist(), range_local), group=group)
        bad_graph = np.array([b for part, _ in bad_all for b in part], dtype=bool)
        out_of_range = sorted(g for _, r in bad_all for g in r)   # conformers that left the split-fp16 range, on any rank
        SAMPLE_STATS["range_trips"] += len(out_of_range)
        first = False
        if save_traj:                          # [steps, N_r, 3] per rank: to rank 0 only (it writes them)
            trajs = [None] * world if rank == 0 else None
            dist.gather_object(None if traj is None else torch.stack(traj).numpy(), trajs, dst=0, group=group)
            if rank == 0:
                traj = torch.from_numpy(np.concatenate([x for x in trajs if x is not None], axis=1))
                if traj_out is None:
                    traj_out = torch.full((traj.shape[0], N, 3), float("nan"))
        nan_failed, range_failed = _sort_results(todo, sub["spans"], spans, bad_graph, out_of_range, wide, ok, pos_out, pos,
                                                 traj_out if (save_traj and rank == 0) else None, traj if (save_traj and rank == 0) else None)
        if range_failed:
            SAMPLE_STATS["bf16x3_retries"] += 1
            if rank == 0:
                log("%d conformers left the split-fp16 range: sampling their molecules (%d of %d) again in split-bf16."
                    % (len(out_of_range), len(range_failed), len(sub["spans"])))
            passes.append((range_failed, clip_local, True, tries))
        if nan_failed:
            if tries + 1 < max_retry:
                if rank == 0:
                    log("NaN in %d of %d molecules: retrying those with local clipping." % (len(nan_failed), len(sub["spans"])))
                passes.append((nan_failed, 20, wide, tries + 1))
            else:
                SAMPLE_STATS["dropped"] += len(nan_failed)
    return pos_out, traj_out, ok
