"""Stand-in for torch_scatter (rusty1s/pytorch_scatter): scatter_add / scatter_mean /
scatter_max / scatter along `dim` with optional `dim_size`; mean divides by max(count,1)."""
import torch


def _prep(src, index, dim, dim_size):
    dim = dim if dim >= 0 else src.dim() + dim
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    shape = list(src.shape)
    shape[dim] = dim_size
    if index.dim() != src.dim():
        view = [1] * src.dim()
        view[dim] = -1
        index = index.view(view).expand_as(src)
    return dim, shape, index


def scatter_add(src, index, dim=-1, out=None, dim_size=None):
    dim, shape, index = _prep(src, index, dim, dim_size)
    res = torch.zeros(shape, dtype=src.dtype, device=src.device)
    return res.scatter_add_(dim, index, src)


def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    dim, shape, idx = _prep(src, index, dim, dim_size)
    s = torch.zeros(shape, dtype=src.dtype, device=src.device).scatter_add_(dim, idx, src)
    cnt = torch.zeros(shape, dtype=src.dtype, device=src.device).scatter_add_(
        dim, idx, torch.ones_like(src))
    return s / cnt.clamp(min=1)


def scatter_max(src, index, dim=-1, out=None, dim_size=None):
    dim, shape, idx = _prep(src, index, dim, dim_size)
    res = torch.full(shape, torch.finfo(src.dtype).min if src.is_floating_point()
                     else torch.iinfo(src.dtype).min, dtype=src.dtype, device=src.device)
    res = res.scatter_reduce(dim, idx, src, reduce="amax", include_self=True)
    return res, None


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    if reduce in ("sum", "add"):
        return scatter_add(src, index, dim, out, dim_size)
    if reduce == "mean":
        return scatter_mean(src, index, dim, out, dim_size)
    raise NotImplementedError(reduce)
