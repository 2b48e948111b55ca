"""Minimal pure-torch stand-in for the third-party packages the reference imports.

TEST INFRASTRUCTURE ONLY.  The reference (tgp 1.0.1, /root/reference) is pure
Python on top of ``torch_geometric`` / ``torch_scatter`` (pyproject.toml:35-60),
neither of which is installed in the build container.  ``install()`` registers
build-authored re-statements of the ~25 leaf functions the SRC hot path calls
 under those module names, so that the reference's *own*
orchestration code (tgp/src.py, tgp/reduce/base_reduce.py, tgp/connect/*.py,
tgp/utils/ops.py, tgp/poolers/{topk,graclus,ndp,diffpool,mincut}.py) can be run
here to emit golden vectors (tests/golden/make_golden.py).

The leaf semantics are written from the published behaviour of PyG 2.6 /
torch_scatter 2.1.2 (the versions the reference's CI pins,
pre-requirements.txt:1-6); they are NOT the reference's code and contain none
of it.  Anything else under ``torch_geometric.*`` resolves to an inert
placeholder so that unrelated reference modules still import.

This file never travels into the product path and is never used on the GPU box.
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import math
import sys
import types
from typing import List, Optional

import torch
from torch import Tensor


# ----------------------------------------------------------------------------
# inert placeholders for names we do not restate
# ----------------------------------------------------------------------------
class _PlaceholderMeta(type):
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Placeholder

    def __getitem__(cls, item):
        return cls

    def __or__(cls, other):
        return cls

    def __ror__(cls, other):
        return cls


class _Placeholder(metaclass=_PlaceholderMeta):
    """Stands in for any un-restated PyG symbol (class, alias or function)."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        raise NotImplementedError("pyg_shim placeholder called")


class _LazyModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Placeholder


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    ROOTS = ("torch_geometric", "torch_scatter")

    def find_spec(self, fullname, path=None, target=None):
        root = fullname.split(".")[0]
        if root in self.ROOTS and fullname not in sys.modules:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _LazyModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


# ----------------------------------------------------------------------------
# torch_geometric.utils
# ----------------------------------------------------------------------------
def maybe_num_nodes(edge_index, num_nodes=None):
    if num_nodes is not None:
        return num_nodes
    if isinstance(edge_index, Tensor):
        if edge_index.is_sparse:
            return max(edge_index.size(0), edge_index.size(1))
        return int(edge_index.max()) + 1 if edge_index.numel() > 0 else 0
    raise NotImplementedError


def _broadcast(index: Tensor, ref: Tensor, dim: int) -> Tensor:
    dim = ref.dim() + dim if dim < 0 else dim
    size = [1] * ref.dim()
    size[dim] = -1
    return index.view(size).expand_as(ref)


def scatter(src: Tensor, index: Tensor, dim: int = 0,
            dim_size: Optional[int] = None, reduce: str = "sum") -> Tensor:
    if isinstance(index, Tensor) and index.dim() != 1:
        raise ValueError("index must be one-dimensional")
    dim = src.dim() + dim if dim < 0 else dim
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    size = list(src.size())
    size[dim] = dim_size
    if reduce in ("sum", "add", "any"):
        idx = _broadcast(index, src, dim)
        return src.new_zeros(size).scatter_add_(dim, idx, src)
    if reduce == "mean":
        count = src.new_zeros(dim_size)
        count.scatter_add_(0, index, src.new_ones(src.size(dim)))
        count = count.clamp(min=1)
        idx = _broadcast(index, src, dim)
        out = src.new_zeros(size).scatter_add_(dim, idx, src)
        return out / _broadcast(count, out, dim)
    if reduce in ("min", "max", "amin", "amax"):
        idx = _broadcast(index, src, dim)
        return src.new_zeros(size).scatter_reduce_(
            dim, idx, src, reduce=f"a{reduce[-3:]}", include_self=False)
    if reduce in ("mul", "prod"):
        idx = _broadcast(index, src, dim)
        return src.new_ones(size).scatter_reduce_(
            dim, idx, src, reduce="prod", include_self=True)
    raise ValueError(f"unknown reduce {reduce}")


def degree(index: Tensor, num_nodes: Optional[int] = None, dtype=None) -> Tensor:
    n = maybe_num_nodes(index, num_nodes)
    out = torch.zeros((n,), dtype=dtype, device=index.device)
    one = torch.ones((index.size(0),), dtype=out.dtype, device=out.device)
    return out.scatter_add_(0, index, one)


def cumsum(x: Tensor, dim: int = 0) -> Tensor:
    size = x.size()[:dim] + (x.size(dim) + 1,) + x.size()[dim + 1:]
    out = x.new_empty(size)
    out.narrow(dim, 0, 1).zero_()
    torch.cumsum(x, dim=dim, out=out.narrow(dim, 1, x.size(dim)))
    return out


def index_sort(inputs: Tensor, max_value: Optional[int] = None, stable: bool = False):
    return inputs.sort(stable=True)


def remove_self_loops(edge_index: Tensor, edge_attr: Optional[Tensor] = None):
    mask = edge_index[0] != edge_index[1]
    edge_index = edge_index[:, mask]
    if edge_attr is None:
        return edge_index, None
    return edge_index, edge_attr[mask]


def add_self_loops(edge_index, edge_attr=None, fill_value=None, num_nodes=None):
    n = maybe_num_nodes(edge_index, num_nodes)
    loop = torch.arange(0, n, device=edge_index.device).view(1, -1).repeat(2, 1)
    if edge_attr is not None:
        fv = 1.0 if fill_value is None else fill_value
        loop_attr = edge_attr.new_full((n,) + edge_attr.size()[1:], fv)
        edge_attr = torch.cat([edge_attr, loop_attr], dim=0)
    return torch.cat([edge_index, loop], dim=1), edge_attr


def add_remaining_self_loops(edge_index, edge_attr=None, fill_value=None, num_nodes=None):
    n = maybe_num_nodes(edge_index, num_nodes)
    mask = edge_index[0] != edge_index[1]
    loop = torch.arange(0, n, device=edge_index.device).view(1, -1).repeat(2, 1)
    if edge_attr is not None:
        fv = 1.0 if fill_value is None else fill_value
        loop_attr = edge_attr.new_full((n,) + edge_attr.size()[1:], fv)
        inv = ~mask
        loop_attr[edge_index[0][inv]] = edge_attr[inv]
        edge_attr = torch.cat([edge_attr[mask], loop_attr], dim=0)
    return torch.cat([edge_index[:, mask], loop], dim=1), edge_attr


_MISSING = "???"


def coalesce(edge_index: Tensor, edge_attr=_MISSING, num_nodes: Optional[int] = None,
             reduce: str = "sum", is_sorted: bool = False, sort_by_row: bool = True):
    nnz = edge_index.size(1)
    n = maybe_num_nodes(edge_index, num_nodes)
    idx = edge_index[0].new_empty(nnz + 1)
    idx[0] = -1
    idx[1:] = edge_index[1 - int(sort_by_row)]
    idx[1:].mul_(n).add_(edge_index[int(sort_by_row)])
    if not is_sorted:
        idx[1:], perm = index_sort(idx[1:], max_value=n * n)
        edge_index = edge_index[:, perm]
        if isinstance(edge_attr, Tensor):
            edge_attr = edge_attr[perm]
    mask = idx[1:] > idx[:-1]
    if mask.all():
        if edge_attr is None or isinstance(edge_attr, Tensor):
            return edge_index, edge_attr
        return edge_index
    edge_index = edge_index[:, mask]
    if edge_attr is None:
        return edge_index, None
    if isinstance(edge_attr, Tensor):
        dim_size = edge_index.size(1)
        seg = torch.arange(0, nnz, device=edge_index.device)
        seg.sub_(mask.logical_not_().cumsum(dim=0))
        return edge_index, scatter(edge_attr, seg, 0, dim_size, reduce)
    return edge_index


def to_undirected(edge_index, edge_attr=_MISSING, num_nodes=None, reduce="add"):
    row, col = edge_index[0], edge_index[1]
    row, col = torch.cat([row, col], dim=0), torch.cat([col, row], dim=0)
    edge_index = torch.stack([row, col], dim=0)
    if isinstance(edge_attr, Tensor):
        edge_attr = torch.cat([edge_attr, edge_attr], dim=0)
    if isinstance(edge_attr, str):
        return coalesce(edge_index, None, num_nodes, reduce)[0]
    return coalesce(edge_index, edge_attr, num_nodes, reduce)


def sort_edge_index(edge_index, edge_attr=_MISSING, num_nodes=None, sort_by_row=True):
    n = maybe_num_nodes(edge_index, num_nodes)
    idx = edge_index[1 - int(sort_by_row)] * n + edge_index[int(sort_by_row)]
    _, perm = index_sort(idx, max_value=n * n)
    edge_index = edge_index[:, perm]
    if isinstance(edge_attr, str):
        return edge_index
    if edge_attr is None:
        return edge_index, None
    return edge_index, edge_attr[perm]


def is_undirected(edge_index, edge_attr=None, num_nodes=None) -> bool:
    n = maybe_num_nodes(edge_index, num_nodes)
    ei1, ea1 = sort_edge_index(edge_index, edge_attr, n, True)
    ei2, ea2 = sort_edge_index(edge_index, edge_attr, n, False)
    if not torch.equal(ei1[0], ei2[1]) or not torch.equal(ei1[1], ei2[0]):
        return False
    if ea1 is not None and not torch.equal(ea1, ea2):
        return False
    return True


def index_to_mask(index: Tensor, size: Optional[int] = None) -> Tensor:
    size = int(index.max()) + 1 if size is None else size
    mask = index.new_zeros(size, dtype=torch.bool)
    mask[index] = True
    return mask


def subgraph(subset, edge_index, edge_attr=None, relabel_nodes=False,
             num_nodes=None, *, return_edge_mask=False):
    if isinstance(subset, (list, tuple)):
        subset = torch.tensor(subset, dtype=torch.long, device=edge_index.device)
    if subset.dtype != torch.bool:
        n = maybe_num_nodes(edge_index, num_nodes)
        node_mask = index_to_mask(subset, size=n)
    else:
        n = subset.size(0)
        node_mask = subset
        subset = node_mask.nonzero().view(-1)
    edge_mask = node_mask[edge_index[0]] & node_mask[edge_index[1]]
    edge_index = edge_index[:, edge_mask]
    edge_attr = edge_attr[edge_mask] if edge_attr is not None else None
    if relabel_nodes:
        # position of every endpoint inside `subset`
        pos = edge_index.new_full((n,), -1)
        pos[subset] = torch.arange(subset.numel(), device=subset.device)
        edge_index = pos[edge_index.view(-1)].view(2, -1)
    if return_edge_mask:
        return edge_index, edge_attr, edge_mask
    return edge_index, edge_attr


def softmax(src, index=None, ptr=None, num_nodes=None, dim: int = 0):
    n = maybe_num_nodes(index, num_nodes)
    src_max = scatter(src.detach(), index, dim, dim_size=n, reduce="max")
    out = src - src_max.index_select(dim, index)
    out = out.exp()
    out_sum = scatter(out, index, dim, dim_size=n, reduce="sum") + 1e-16
    return out / out_sum.index_select(dim, index)


def to_dense_batch(x, batch=None, fill_value=0.0, max_num_nodes=None, batch_size=None):
    if batch is None and max_num_nodes is None:
        mask = torch.ones(1, x.size(0), dtype=torch.bool, device=x.device)
        return x.unsqueeze(0), mask
    if batch is None:
        batch = x.new_zeros(x.size(0), dtype=torch.long)
    if batch_size is None:
        batch_size = int(batch.max()) + 1
    num_nodes = scatter(batch.new_ones(x.size(0)), batch, dim=0,
                        dim_size=batch_size, reduce="sum")
    cum_nodes = cumsum(num_nodes)
    filter_nodes = False
    if max_num_nodes is None:
        max_num_nodes = int(num_nodes.max())
    elif num_nodes.max() > max_num_nodes:
        filter_nodes = True
    tmp = torch.arange(batch.size(0), device=x.device) - cum_nodes[batch]
    idx = tmp + (batch * max_num_nodes)
    if filter_nodes:
        mask = tmp < max_num_nodes
        x, idx = x[mask], idx[mask]
    size = [batch_size * max_num_nodes] + list(x.size())[1:]
    out = torch.as_tensor(fill_value, device=x.device).to(x.dtype).repeat(size)
    out[idx] = x
    out = out.view([batch_size, max_num_nodes] + list(x.size())[1:])
    mask = torch.zeros(batch_size * max_num_nodes, dtype=torch.bool, device=x.device)
    mask[idx] = 1
    return out, mask.view(batch_size, max_num_nodes)


def to_dense_adj(edge_index, batch=None, edge_attr=None, max_num_nodes=None,
                 batch_size=None):
    if batch is None:
        max_index = int(edge_index.max()) + 1 if edge_index.numel() > 0 else 0
        batch = edge_index.new_zeros(max_index)
    if batch_size is None:
        batch_size = int(batch.max()) + 1 if batch.numel() > 0 else 1
    one = batch.new_ones(batch.size(0))
    num_nodes = scatter(one, batch, dim=0, dim_size=batch_size, reduce="sum")
    cum_nodes = cumsum(num_nodes)
    idx0 = batch[edge_index[0]]
    idx1 = edge_index[0] - cum_nodes[batch][edge_index[0]]
    idx2 = edge_index[1] - cum_nodes[batch][edge_index[1]]
    if max_num_nodes is None:
        max_num_nodes = int(num_nodes.max())
    elif ((idx1.numel() > 0 and idx1.max() >= max_num_nodes)
          or (idx2.numel() > 0 and idx2.max() >= max_num_nodes)):
        mask = (idx1 < max_num_nodes) & (idx2 < max_num_nodes)
        idx0, idx1, idx2 = idx0[mask], idx1[mask], idx2[mask]
        edge_attr = None if edge_attr is None else edge_attr[mask]
    if edge_attr is None:
        edge_attr = torch.ones(idx0.numel(), device=edge_index.device)
    size = [batch_size, max_num_nodes, max_num_nodes] + list(edge_attr.size())[1:]
    flattened_size = batch_size * max_num_nodes * max_num_nodes
    idx = idx0 * max_num_nodes * max_num_nodes + idx1 * max_num_nodes + idx2
    adj = scatter(edge_attr, idx, dim=0, dim_size=flattened_size, reduce="sum")
    return adj.view(size)


def unbatch(src: Tensor, batch: Tensor, dim: int = 0, batch_size=None) -> List[Tensor]:
    sizes = degree(batch, batch_size, dtype=torch.long).tolist()
    return list(src.split(sizes, dim))


def unbatch_edge_index(edge_index: Tensor, batch: Tensor, batch_size=None):
    deg = degree(batch, batch_size, dtype=torch.long)
    ptr = cumsum(deg)
    edge_batch = batch[edge_index[0]]
    edge_index = edge_index - ptr[edge_batch]
    sizes = degree(edge_batch, batch_size, dtype=torch.long).cpu().tolist()
    return list(edge_index.split(sizes, dim=1))


def get_laplacian(edge_index, edge_weight=None, normalization=None, dtype=None,
                  num_nodes=None):
    edge_index, edge_weight = remove_self_loops(edge_index, edge_weight)
    if edge_weight is None:
        edge_weight = torch.ones(edge_index.size(1), dtype=dtype, device=edge_index.device)
    n = maybe_num_nodes(edge_index, num_nodes)
    row, col = edge_index[0], edge_index[1]
    deg = scatter(edge_weight, row, 0, dim_size=n, reduce="sum")
    if normalization is None:
        edge_index, _ = add_self_loops(edge_index, num_nodes=n)
        edge_weight = torch.cat([-edge_weight, deg], dim=0)
    elif normalization == "sym":
        dis = deg.pow_(-0.5)
        dis.masked_fill_(dis == float("inf"), 0)
        edge_weight = dis[row] * edge_weight * dis[col]
        edge_index, edge_weight = add_self_loops(edge_index, -edge_weight,
                                                 fill_value=1.0, num_nodes=n)
    else:  # 'rw'
        di = 1.0 / deg
        di.masked_fill_(di == float("inf"), 0)
        edge_weight = di[row] * edge_weight
        edge_index, edge_weight = add_self_loops(edge_index, -edge_weight,
                                                 fill_value=1.0, num_nodes=n)
    return edge_index, edge_weight


def to_scipy_sparse_matrix(edge_index, edge_attr=None, num_nodes=None):
    import scipy.sparse
    row, col = edge_index.cpu()
    if edge_attr is None:
        edge_attr = torch.ones(row.size(0))
    else:
        edge_attr = edge_attr.view(-1).cpu()
    n = maybe_num_nodes(edge_index, num_nodes)
    return scipy.sparse.coo_matrix(
        (edge_attr.numpy(), (row.numpy(), col.numpy())), (n, n))


def from_scipy_sparse_matrix(A):
    A = A.tocoo()
    row = torch.from_numpy(A.row).to(torch.long)
    col = torch.from_numpy(A.col).to(torch.long)
    return torch.stack([row, col], dim=0), torch.from_numpy(A.data)


def is_torch_sparse_tensor(src) -> bool:
    return isinstance(src, Tensor) and src.layout in (
        torch.sparse_coo, torch.sparse_csr, torch.sparse_csc)


def is_sparse(src) -> bool:
    return is_torch_sparse_tensor(src)


# ----------------------------------------------------------------------------
# torch_geometric.nn pieces
# ----------------------------------------------------------------------------
def uniform(size: int, value) -> None:
    if isinstance(value, Tensor):
        bound = 1.0 / math.sqrt(size)
        value.data.uniform_(-bound, bound)


def activation_resolver(query="relu", *args, **kwargs):
    if not isinstance(query, str):
        return query
    table = {"tanh": torch.nn.Tanh, "relu": torch.nn.ReLU, "sigmoid": torch.nn.Sigmoid,
             "elu": torch.nn.ELU, "leakyrelu": torch.nn.LeakyReLU,
             "softplus": torch.nn.Softplus, "gelu": torch.nn.GELU,
             "identity": torch.nn.Identity}
    return table[query.lower().replace("_", "")](*args, **kwargs)


def topk(x: Tensor, ratio, batch: Tensor, min_score=None, tol: float = 1e-7) -> Tensor:
    if min_score is not None:
        scores_max = scatter(x, batch, reduce="max")[batch] - tol
        scores_min = scores_max.clamp(max=min_score)
        return (x > scores_min).nonzero().view(-1)
    num_nodes = scatter(batch.new_ones(x.size(0)), batch, reduce="sum")
    if ratio >= 1:
        k = num_nodes.new_full((num_nodes.size(0),), int(ratio))
        k = torch.min(k, num_nodes)
    else:
        k = (float(ratio) * num_nodes.to(x.dtype)).ceil().to(torch.long)
    x, x_perm = torch.sort(x.view(-1), descending=True)
    batch = batch[x_perm]
    batch, batch_perm = torch.sort(batch, descending=False, stable=True)
    arange = torch.arange(x.size(0), dtype=torch.long, device=x.device)
    ptr = cumsum(num_nodes)
    batched_arange = arange - ptr[batch]
    mask = batched_arange < k[batch]
    return x_perm[batch_perm[mask]]


class MLP(torch.nn.Module):
    """Plain-last MLP: Linear -> act -> dropout -> ... -> Linear (norm=None only)."""

    def __init__(self, channel_list, act=None, norm=None, dropout=0.0, **kwargs):
        super().__init__()
        assert norm is None
        self.channel_list = list(channel_list)
        self.lins = torch.nn.ModuleList(
            torch.nn.Linear(a, b, bias=True)
            for a, b in zip(self.channel_list[:-1], self.channel_list[1:]))
        self.act = activation_resolver(act)
        self.dropout = float(dropout)

    def reset_parameters(self):
        for lin in self.lins:
            lin.reset_parameters()

    def forward(self, x: Tensor) -> Tensor:
        last = len(self.lins) - 1
        for i, lin in enumerate(self.lins):
            x = lin(x)
            if i != last:
                if self.act is not None:
                    x = self.act(x)
                x = torch.nn.functional.dropout(x, p=self.dropout, training=self.training)
        return x


class Data:
    def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, pos=None, **kwargs):
        self.x, self.edge_index, self.edge_attr, self.y, self.pos = x, edge_index, edge_attr, y, pos
        for k, v in kwargs.items():
            setattr(self, k, v)


class BaseTransform:
    def __call__(self, data):
        return self.forward(data)

    def forward(self, data):
        raise NotImplementedError


# ----------------------------------------------------------------------------
def _mod(name: str, **attrs) -> types.ModuleType:
    m = _LazyModule(name)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install() -> None:
    """Register the shim modules.  Call before importing the reference's ``tgp``."""
    if "torch_geometric" in sys.modules and getattr(
            sys.modules["torch_geometric"], "_IS_TGP_SHIM", False):
        return
    sys.meta_path.insert(0, _Finder())
    utils = dict(
        scatter=scatter, coalesce=coalesce, subgraph=subgraph, softmax=softmax,
        to_dense_adj=to_dense_adj, to_dense_batch=to_dense_batch, unbatch=unbatch,
        unbatch_edge_index=unbatch_edge_index, remove_self_loops=remove_self_loops,
        add_remaining_self_loops=add_remaining_self_loops, add_self_loops=add_self_loops,
        cumsum=cumsum, degree=degree, index_sort=index_sort, to_undirected=to_undirected,
        is_undirected=is_undirected, get_laplacian=get_laplacian,
        to_scipy_sparse_matrix=to_scipy_sparse_matrix,
        from_scipy_sparse_matrix=from_scipy_sparse_matrix, sort_edge_index=sort_edge_index,
        is_sparse=is_sparse, is_torch_sparse_tensor=is_torch_sparse_tensor,
        index_to_mask=index_to_mask)
    root = _mod("torch_geometric", _IS_TGP_SHIM=True, __version__="2.6.1-shim")
    root.utils = _mod("torch_geometric.utils", **utils)
    _mod("torch_geometric.utils.num_nodes", maybe_num_nodes=maybe_num_nodes)
    root.typing = _mod("torch_geometric.typing", Adj=Tensor, OptTensor=Optional[Tensor],
                       PairTensor=Tensor, Tensor=Tensor, SparseTensor=_Placeholder,
                       WITH_INDEX_SORT=False)
    root.data = _mod("torch_geometric.data", Data=Data, Batch=Data)
    root.transforms = _mod("torch_geometric.transforms", BaseTransform=BaseTransform)
    nn = _mod("torch_geometric.nn")
    root.nn = nn
    nn.inits = _mod("torch_geometric.nn.inits", uniform=uniform)
    nn.resolver = _mod("torch_geometric.nn.resolver", activation_resolver=activation_resolver)
    _mod("torch_geometric.nn.models")
    _mod("torch_geometric.nn.models.mlp", MLP=MLP)
    _mod("torch_geometric.nn.pool")
    _mod("torch_geometric.nn.pool.select")
    _mod("torch_geometric.nn.pool.select.topk", topk=topk)
    aggr = _mod("torch_geometric.nn.aggr")
    aggr.Aggregation = None  # reference: reduce/get_aggr.py:12-20 tolerates a missing class

    def _ts_scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
        return scatter(src, index, dim=dim, dim_size=dim_size, reduce=reduce)

    _mod("torch_scatter", scatter=_ts_scatter,
         scatter_add=lambda s, i, dim=-1, out=None, dim_size=None: scatter(s, i, dim, dim_size, "sum"),
         scatter_mul=lambda s, i, dim=-1, out=None, dim_size=None: scatter(s, i, dim, dim_size, "mul"))
