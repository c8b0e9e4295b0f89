"""MessagePassing: propagate(edge_index, **kw) gathers `<name>_j` = kw[name][edge_index[0]]
(flow source_to_target), calls self.message with the arguments it names, and sum-aggregates
at edge_index[1] (aggr='add')."""
import inspect
import torch


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", flow="source_to_target", node_dim=0, **kwargs):
        super().__init__()
        assert aggr == "add" and flow == "source_to_target"
        self.aggr = aggr

    def propagate(self, edge_index, size=None, **kwargs):
        params = list(inspect.signature(self.message).parameters)
        n = None
        args = {}
        for p in params:
            if p.endswith("_j") or p.endswith("_i"):
                src = kwargs[p[:-2]]
                if isinstance(src, (tuple, list)):
                    src = src[0] if p.endswith("_j") else src[1]
                n = src.size(0)
                args[p] = src[edge_index[0] if p.endswith("_j") else edge_index[1]]
            else:
                args[p] = kwargs[p]
        msg = self.message(**args)
        if n is None:
            n = int(edge_index.max()) + 1
        out = torch.zeros((n,) + tuple(msg.shape[1:]), dtype=msg.dtype, device=msg.device)
        out.index_add_(0, edge_index[1], msg)
        return out

    def message(self, x_j):
        return x_j
