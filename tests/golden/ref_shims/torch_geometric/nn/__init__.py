"""radius_graph: restates torch_cluster's CUDA kernel semantics (the reference's default
device is cuda, scripts/test.py:45): for each target i scan candidates j of the same graph
in ASCENDING index order, keep those with ||x_i-x_j||^2 < r^2 (fp32, accumulated x,y,z in
that order) until max_num_neighbors+1 = 33 are found (self included), then drop self.
Edges are emitted as row=j (source), col=i (target), grouped by target."""
import torch
from .conv import MessagePassing


def radius_graph(x, r, batch=None, loop=False, max_num_neighbors=32, flow="source_to_target"):
    assert flow == "source_to_target" and not loop
    n = x.size(0)
    if batch is None:
        batch = torch.zeros(n, dtype=torch.long)
    x = x.float()
    rows, cols = [], []
    r2 = torch.tensor(r, dtype=torch.float32) * torch.tensor(r, dtype=torch.float32)
    counts = torch.bincount(batch)
    ptr = torch.cat([counts.new_zeros(1), counts.cumsum(0)])
    limit = max_num_neighbors + 1
    for g in range(counts.numel()):
        s, e = int(ptr[g]), int(ptr[g + 1])
        p = x[s:e]
        d = p[:, None, :] - p[None, :, :]            # d[i, j] = x_i - x_j
        d2 = d[..., 0] * d[..., 0]
        d2 = d2 + d[..., 1] * d[..., 1]
        d2 = d2 + d[..., 2] * d[..., 2]
        within = d2 < r2                              # [i, j]
        rank = within.long().cumsum(1)                # 1-based rank of j among in-radius of i
        keep = within & (rank <= limit)
        keep.fill_diagonal_(False)
        ti, sj = keep.nonzero(as_tuple=True)          # ordered by (i, j)
        rows.append(sj + s)
        cols.append(ti + s)
    return torch.stack([torch.cat(rows), torch.cat(cols)], dim=0)


def radius(*a, **k):
    raise NotImplementedError
