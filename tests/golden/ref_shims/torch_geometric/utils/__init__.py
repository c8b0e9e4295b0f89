"""to_dense_adj: sum-scatter into [1,N,N]; dense_to_sparse: row-major non-zeros."""
import torch


def to_dense_adj(edge_index, batch=None, edge_attr=None, max_num_nodes=None):
    n = int(edge_index.max()) + 1 if max_num_nodes is None else max_num_nodes
    if edge_attr is None:
        edge_attr = torch.ones(edge_index.size(1), dtype=torch.long, device=edge_index.device)
    adj = torch.zeros(n * n, dtype=edge_attr.dtype, device=edge_index.device)
    adj.index_add_(0, edge_index[0] * n + edge_index[1], edge_attr)
    return adj.view(1, n, n)


def dense_to_sparse(adj):
    if adj.dim() == 3:
        adj = adj.squeeze(0)
    idx = adj.nonzero(as_tuple=False).t().contiguous()
    return idx, adj[idx[0], idx[1]]
