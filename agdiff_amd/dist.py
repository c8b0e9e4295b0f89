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
        if nan_flag is not None:
            self.stage[-1:].copy_(nan_flag.to(torch.float32), non_blocking=True)

    def wait(self):
        if self.is_cuda:
            torch.cuda.current_stream().wait_stream(self.side)

    def result(self):
        """(list of per-rank [n_r, 3] position tensors, any_nan) from the last gather."""
        self.wait()
        g = self.gathered.view(self.world, self.pad)
        return [g[r, : self.sizes[r] * 3].view(-1, 3) for r in range(self.world)], bool((g[:, -1] != 0).any())
