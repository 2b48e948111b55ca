"""Select-Reduce-Connect composition: ``PoolingOutput``, ``SRCPooling``, ``DenseSRCPooling`` and the
pre-coarsening mix-ins (public surface of reference tgp/src.py)."""
from __future__ import annotations

import os as _os
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Dict, Iterator, List, Optional, Tuple, Union

import torch
from torch import Tensor

from . import functions as Fn
from . import kernels as K
from .connect import Connect, SparseConnect, _normalize_pooled_edges
from .imports import HAS_PYG
from .lift import Lift
from .reduce import BaseReduce, Reduce, _SparseReduceFn
from .select import Select, SelectOutput
from .utils import Signature, connectivity_to_edge_index, foo_signature
from .utils.ops import (batch_info, build_pooled_batch, graph_ptr, is_dense_adj, like_input_dtype, max_graph_size,
                        num_graphs_of)


# A/B switch (read once): 0 keeps the staged sparse operators whenever a gradient is required
_FOLD_TRAINING = _os.environ.get("TGP_FOLD_TRAINING", "1") != "0"


@dataclass
class PoolingOutput:
    """Result record of a pooling layer (reference src.py:19-116)."""

    x: Optional[Tensor] = None
    edge_index: Optional[Tensor] = None
    edge_weight: Optional[Tensor] = None
    batch: Optional[Tensor] = None
    so: Optional[SelectOutput] = None
    loss: Optional[Dict] = None

    @property
    def mask(self) -> Optional[Tensor]:
        return self.so.out_mask if self.so is not None else None

    @staticmethod
    def _shape(t) -> Optional[list]:
        return [*t.shape] if t is not None else None

    def __repr__(self) -> str:
        so = [self.so.num_nodes, self.so.num_supernodes] if self.so is not None else None
        loss = list(self.loss.keys()) if self.loss is not None else None
        return (f"PoolingOutput(so={so}, x={self._shape(self.x)}, edge_index={self._shape(self.edge_index)}, "
                f"edge_weight={self._shape(self.edge_weight)}, batch={self._shape(self.batch)}, "
                f"mask={self._shape(self.mask)}, loss={loss})")

    def __iter__(self) -> Iterator:
        return iter((self.x, self.edge_index, self.edge_weight, self.batch, self.mask, self.so, self.loss))

    @property
    def has_loss(self) -> bool:
        return bool(isinstance(self.loss, dict) and len(self.loss) > 0)

    def get_loss_value(self, name: str = None) -> Union[float, List[float]]:
        if not self.has_loss:
            return 0
        return list(self.loss.values()) if name is None else self.loss[name]

    def as_data(self):
        """A ``torch_geometric.data.Data`` when PyG is installed, else an attribute namespace with the
        same fields (reference src.py:94-116)."""
        if self.batch is not None:
            num_nodes = self.batch.numel()
        elif self.x is not None:
            num_nodes = self.x.size(-2)
        else:
            num_nodes = self.so.num_supernodes if self.so is not None else None
        fields = dict(x=self.x, edge_index=self.edge_index, edge_weight=self.edge_weight, batch=self.batch,
                      mask=self.mask, so=self.so, num_nodes=num_nodes)
        if HAS_PYG:  # pragma: no cover - PyG is absent from the MI355X image
            from torch_geometric.data import Data
            return Data(**fields)
        return SimpleNamespace(**fields)


class SRCPooling(torch.nn.Module):
    """select / reduce / connect / lift dispatch with optional caching of the select and connect results
    (reference src.py:119-307)."""

    def __init__(self, selector: Select = None, reducer: Reduce = None, lifter: Lift = None,
                 connector: Connect = None, cached: bool = False):
        super().__init__()
        self.selector, self.reducer, self.lifter, self.connector = selector, reducer, lifter, connector
        self.cached = cached
        self._so_cached = None
        self._pooled_edge_index = None
        self._pooled_edge_weight = None

    def reset_parameters(self):
        for op in (self.selector, self.reducer, self.lifter, self.connector):
            op.reset_parameters()

    def select(self, **kwargs) -> SelectOutput:
        if self.selector is None:
            raise NotImplementedError
        if self._so_cached is not None:
            return self._so_cached
        so = self.selector(**kwargs)
        if self.cached:
            self._so_cached = so
        return so

    def reduce(self, **kwargs):
        if self.reducer is None:
            raise NotImplementedError
        return self.reducer(**kwargs)

    def lift(self, **kwargs):
        if self.lifter is None:
            raise NotImplementedError
        return self.lifter(**kwargs)

    def connect(self, **kwargs):
        if self.connector is None:
            raise NotImplementedError
        if self._pooled_edge_index is not None:
            return self._pooled_edge_index, self._pooled_edge_weight
        ei, ew = self.connector(**kwargs)
        if self.cached:
            self._pooled_edge_index, self._pooled_edge_weight = ei, ew
        return ei, ew

    def reduce_connect(self, x: Tensor, edge_index, edge_weight: Optional[Tensor], so: SelectOutput,
                       batch: Optional[Tensor]):
        """Sparse Reduce + Connect of a batch of SMALL graphs as ONE native launch (fused A1 + A2 +
        A4/A5 + A6; ``tgp_sparse_pool_small_f32``): ``(x_pool, batch_pool, edge_index_pool, edge_weight_pool)`` with the
        values ``self.reduce`` + ``self.connect`` return, or None when the call is not that case -- host tensors, edge
        weights that need a gradient (the staged Connect is the differentiable one; x and the assignment weights get
        theirs from the sparse Reduce's backward attached to this call's x'), no or an unsorted batch vector, a graph
        of more than 64 nodes, non-tensor connectivity, caching -- or when the kernel's on-device checks refuse the
        input (an edge between two graphs, unsorted rows, ...).  The pooled ``edge_index`` is a new contiguous [2, E']
        tensor, values and order those of the staged operators (inside ``with tgp.kernels.output_views():`` it is a
        view of the kernel's capacity buffer instead: both rows contiguous, row stride E, no copy)."""
        c = self.connector
        if (self.cached or type(c) is not SparseConnect or type(self.reducer) is not BaseReduce or batch is None
                or not isinstance(x, Tensor) or not x.is_cuda or x.dtype != torch.float32
                or not isinstance(edge_index, Tensor) or edge_index.is_sparse or edge_index.dtype != torch.int64):
            return None
        xs, es, s = x.shape, edge_index.shape, so.s  # (this sits in front of a ~10 us kernel: every call here is counted)
        if (len(xs) != 2 or xs[0] == 0 or xs[1] == 0 or x.stride(1) != 1 or len(es) != 2 or es[0] != 2 or es[1] == 0
                or not (isinstance(s, Tensor) and s.is_sparse) or batch.dtype != torch.int64 or batch.numel() != xs[0]
                or s.size(-2) != xs[0]):
            return None
        ew = edge_weight
        if ew is not None:
            if ew.dtype != torch.float32 or ew.numel() != es[1]:
                return None
            if ew.dim() != 1:
                ew = ew.reshape(-1)
        weight = so.weight
        # training (r5, late): the same launch with the sparse Reduce's backward attached to x' (x and the assignment
        # weights -- TopK's scores -- get their gradients from it); edge weights that need a gradient keep the staged
        # operators, whose Connect is differentiable
        train = torch.is_grad_enabled() and (x.requires_grad or (weight is not None and weight.requires_grad))
        if torch.is_grad_enabled() and ew is not None and ew.requires_grad:
            return None
        if train and not _FOLD_TRAINING:
            return None
        if torch.cuda.is_current_stream_capturing():
            return None  # the call waits for its edge count on the host (the staged operators raise for the same reason)
        info = batch_info(batch)
        if (not info.is_sorted or info.num_graphs < 2 or info.max_nodes > K.sparse_pool_small_max_graph_nodes()
                or K.sparse_pool_small_declined(edge_index)):
            return None
        index = s.indices()  # [2, nnz]: (node_index, cluster_index)
        nnz, n = index.size(1), xs[0]
        if nnz < n:
            mode = 0  # sparse_connect's first branch: kept-node selection (base_conn.py:79-82)
        elif nnz == n:
            mode = 1  # one-over-K clustering (base_conn.py:83-89)
        else:
            return None
        if weight is not None and weight.dtype != torch.float32:
            return None
        num_supernodes = s.size(-1)
        out = K.sparse_pool_small(x.detach() if train else x, info.ptr, edge_index, ew, index,
                                  weight.detach() if (train and weight is not None) else weight, num_supernodes, mode,
                                  reduce_op=c.reduce_op, remove_self_loops=c.remove_self_loops,
                                  assign_ptr=so.__dict__.get("_assign_ptr") if mode == 0 else None, checked=True)
        if out is None:
            return None
        x_pool, batch_pool, ei, w_pool = out
        if train:
            x_pool = _SparseReduceFn.apply(x, weight, so, [x_pool])
        if c.degree_norm or c.edge_weight_norm:
            ei, w_pool = _normalize_pooled_edges(ei, w_pool, num_supernodes, c.degree_norm, c.edge_weight_norm, batch_pool)
        return x_pool, batch_pool, ei, w_pool

    def preprocessing(self, x: Tensor, edge_index, **kwargs):
        return x, edge_index, None

    @property
    def is_dense(self) -> bool:
        if self.selector is None:
            raise NotImplementedError
        return self.selector.is_dense

    @property
    def is_sparse(self) -> bool:
        return not self.is_dense

    @property
    def has_loss(self) -> bool:
        return self.compute_loss.__qualname__.split(".")[0] != "SRCPooling"

    @property
    def is_trainable(self) -> bool:
        return any(p.requires_grad for p in self.parameters())

    def compute_loss(self, *args, **kwargs) -> Optional[dict]:
        return None

    def clear_cache(self):
        self._so_cached = None
        self._pooled_edge_index = None
        self._pooled_edge_weight = None

    @property
    def is_precoarsenable(self) -> bool:
        return isinstance(self, Precoarsenable) and not self.is_trainable

    @classmethod
    def get_signature(cls) -> Signature:
        return foo_signature(cls)

    @classmethod
    def get_forward_signature(cls) -> Signature:
        return foo_signature(cls.forward)

    @staticmethod
    def data_transforms():
        return None

    def extra_repr_args(self) -> dict:
        return {}

    def __repr__(self) -> str:
        lines = [f"{self.__class__.__name__}(", f"\tselect={self.selector}", f"\treduce={self.reducer}",
                 f"\tlift={self.lifter}", f"\tconnect={self.connector}"]
        lines += [f"\t{k}={v}" for k, v in self.extra_repr_args().items()]
        return "\n".join(lines + [")"])


# --------------------------------------------------------------------------- sparse -> padded dense
def to_dense_batch(x: Tensor, batch: Optional[Tensor] = None, max_num_nodes: Optional[int] = None,
                   batch_size: Optional[int] = None) -> Tuple[Tensor, Tensor]:
    """[N,F] + batch -> ([B,Nmax,F], mask [B,Nmax])  (the algorithm of PyG ``to_dense_batch``)."""
    if batch is None and max_num_nodes is None:
        return x.unsqueeze(0), torch.ones(1, x.size(0), dtype=torch.bool, device=x.device)
    if batch is None:
        batch = x.new_zeros(x.size(0), dtype=torch.long)
    if batch_size is None:
        batch_size = num_graphs_of(batch)
    sizes, ptr = graph_ptr(batch, batch_size)
    if max_num_nodes is None:
        max_num_nodes = max_graph_size(batch)
    if x.is_cuda and x.dtype == torch.float32:
        return Fn.to_dense_batch(x, batch, ptr, batch_size, max_num_nodes)  # one HIP kernel (+ gather backward)
    local = torch.arange(batch.numel(), device=x.device) - ptr[batch]
    keep = local < max_num_nodes
    slot = (local + batch * max_num_nodes)[keep]
    out = x.new_zeros((batch_size * max_num_nodes,) + tuple(x.shape[1:]))
    out[slot] = x[keep]
    mask = torch.zeros(batch_size * max_num_nodes, dtype=torch.bool, device=x.device)
    mask[slot] = True
    return out.view((batch_size, max_num_nodes) + tuple(x.shape[1:])), mask.view(batch_size, max_num_nodes)


def to_dense_adj(edge_index: Tensor, batch: Optional[Tensor] = None, edge_attr: Optional[Tensor] = None,
                 max_num_nodes: Optional[int] = None, batch_size: Optional[int] = None,
                 transposed: bool = False) -> Tensor:
    """Edge list -> [B,Nmax,Nmax], duplicates summed (the algorithm of PyG ``to_dense_adj``; with
    ``batch=None`` the node count is inferred from ``edge_index.max()+1`` exactly as PyG does)."""
    if batch is None:
        n = int(edge_index.max()) + 1 if edge_index.numel() > 0 else 0
        batch = edge_index.new_zeros(n)
    if batch_size is None:
        batch_size = num_graphs_of(batch)
    sizes, ptr = graph_ptr(batch, batch_size)
    if edge_index.is_cuda and (edge_attr is None or (edge_attr.dim() == 1 and edge_attr.dtype == torch.float32)):
        nmax = max_num_nodes if max_num_nodes is not None else max_graph_size(batch)
        # one HIP kernel; differentiable w.r.t. the edge weights through the matching gather kernel
        return Fn.to_dense_adj(edge_index, edge_attr, batch, ptr, batch_size, nmax, transposed)
    if (edge_index.is_cuda and edge_attr is not None and edge_attr.dim() >= 2 and edge_attr.dtype == torch.float32
            and edge_attr.size(0) == edge_index.size(1) and edge_attr[0].numel() > 0 and batch.numel() > 0
            and not (torch.is_grad_enabled() and edge_attr.requires_grad)):
        # multi-channel edge attributes (r5): one native scatter-add, [B,Nmax,Nmax,C]; `transposed` swaps the two node
        # axes in the kernel (the torch form below returns the transposed view of the same values)
        from . import kernels as K
        nmax = max_num_nodes if max_num_nodes is not None else max_graph_size(batch)
        return K.to_dense_adj_channels(edge_index, edge_attr, batch, ptr, batch_size, nmax, transposed)
    # what is left for the torch form below: host tensors (CPU-side data preparation), float64 / differentiable
    # multi-channel edge attributes
    g = batch[edge_index[0]]
    r = edge_index[0] - ptr[g]
    c = edge_index[1] - ptr[batch[edge_index[1]]]
    if max_num_nodes is None:
        max_num_nodes = max_graph_size(batch)
    else:
        ok = (r < max_num_nodes) & (c < max_num_nodes)
        g, r, c = g[ok], r[ok], c[ok]
        edge_attr = None if edge_attr is None else edge_attr[ok]
    w = torch.ones(g.numel(), device=edge_index.device) if edge_attr is None else edge_attr
    flat = w.new_zeros((batch_size * max_num_nodes * max_num_nodes,) + tuple(w.shape[1:]))
    flat.index_add_(0, (g * max_num_nodes + r) * max_num_nodes + c, w)
    adj = flat.view((batch_size, max_num_nodes, max_num_nodes) + tuple(w.shape[1:]))
    return adj.transpose(1, 2) if transposed else adj


class DenseSRCPooling(SRCPooling):
    """Dense poolers: sparse -> padded-dense preprocessing, batched / unbatched modes, optional
    block-diagonal sparse output (reference src.py:310-557)."""

    def __init__(self, selector: Select = None, reducer: Reduce = None, lifter: Lift = None,
                 connector: Connect = None, cached: bool = False, adj_transpose: bool = False,
                 batched: bool = True, sparse_output: bool = False, cache_preprocessing: bool = False):
        super().__init__(selector=selector, reducer=reducer, lifter=lifter, connector=connector, cached=cached)
        self.batched = batched
        self.sparse_output = sparse_output
        self.adj_transpose = adj_transpose
        self.cache_preprocessing = cache_preprocessing
        self.preprocessing_cache = None

    def preprocessing(self, x: Tensor, edge_index, edge_weight: Optional[Tensor] = None,
                      batch: Optional[Tensor] = None, max_num_nodes: Optional[int] = None,
                      batch_size: Optional[int] = None, use_cache: bool = False, **kwargs):
        if use_cache and self.preprocessing_cache is not None:
            adj = self.preprocessing_cache
        else:
            ei, ew = connectivity_to_edge_index(edge_index, edge_weight)
            fused = self._densify_together(x, ei, ew, batch, max_num_nodes, batch_size)
            if fused is not None:  # x, mask and the zero-filled adjacency in one launch, the edge scatter behind it
                x, adj, mask = fused
                if use_cache:
                    self.preprocessing_cache = adj
                return x, adj, mask
            # adj_transpose (src.py:442-443): the HIP kernel writes A^T directly; the torch path (host
            # tensors / autograd through edge weights) returns the transposed view the reference builds
            adj = to_dense_adj(ei, batch, ew, max_num_nodes, batch_size, transposed=self.adj_transpose)
            if use_cache:
                self.preprocessing_cache = adj
        x, mask = to_dense_batch(x, batch, max_num_nodes, batch_size)
        return x, adj, mask

    def _densify_together(self, x, ei, ew, batch, max_num_nodes, batch_size):
        """Device batches with a sorted batch vector: to_dense_batch's launch also zero-fills the [B,Nmax,Nmax] buffer
        that to_dense_adj's edge scatter then adds into (three launches -> two in front of every dense pooler call;
        a dependent launch costs ~5 us here whatever it does).  None: the general functions above take the case."""
        if not (isinstance(x, Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and batch is not None
                and batch.numel() == x.size(0) and x.size(0) > 0 and x.size(1) > 0 and isinstance(ei, Tensor)
                and ei.is_cuda and (ew is None or (ew.dim() == 1 and ew.dtype == torch.float32))):
            return None
        info = batch_info(batch)
        if not info.is_sorted:
            return None
        if batch_size is None:
            batch_size = num_graphs_of(batch)
        _, ptr = graph_ptr(batch, batch_size)
        nmax = max_num_nodes if max_num_nodes is not None else max_graph_size(batch)
        adj0 = torch.empty(batch_size, nmax, nmax, dtype=torch.float32, device=x.device)
        xd, mask = Fn.to_dense_batch(x, batch, ptr, batch_size, nmax, also_zero=adj0)
        adj = Fn.to_dense_adj(ei, ew, batch, ptr, batch_size, nmax, self.adj_transpose, zeroed_out=adj0)
        return xd, adj, mask

    def _ensure_batched_inputs(self, x, edge_index, edge_weight, batch, mask, use_cache: Optional[bool] = None):
        if edge_index is None:
            raise ValueError("edge_index cannot be None when batched=True.")
        if use_cache is None:
            use_cache = self.cache_preprocessing
        if use_cache and batch is not None and batch.numel() > 0:
            lo, hi = torch.aminmax(batch)
            use_cache = int(lo) == int(hi)  # never cache a multi-graph batch
        if is_dense_adj(edge_index):
            x = x.unsqueeze(0) if x.dim() == 2 else x
            if mask is None:
                mask = x.new_ones(x.size(0), x.size(1), dtype=torch.bool)
            if use_cache:
                self.preprocessing_cache = edge_index
            return x, edge_index, mask
        return self.preprocessing(x=x, edge_index=edge_index, edge_weight=edge_weight, batch=batch,
                                  use_cache=use_cache)

    def clear_cache(self):
        super().clear_cache()
        self.preprocessing_cache = None

    def reduce_connect(self, x: Tensor, adj: Tensor, so: SelectOutput, want_raw: bool = False,
                       out_x: Optional[Tensor] = None, out_adj: Optional[Tensor] = None,
                       want_mincut_terms: bool = False, want_diff_losses=None):
        """Reduce + Connect of a padded dense batch as ONE native call (fused A3 + A7 + A8):
        ``(x_pool [B,K,F], raw S^T A S or None, adj_pool [B,K,K])``.  ``U = A S`` is formed once and
        ``S^T [U | X]`` runs as a single grid (one wave per graph when the graphs fit in LDS), so S is read
        once for both products.  Used by the dense poolers' forward whenever no gradient is required; under
        autograd the separately differentiable ``reduce`` / ``connect`` operators are used instead and this
        returns None."""
        from .connect import DenseConnect
        from .reduce import BaseReduce
        from . import kernels as K
        c = self.connector
        s = so.s
        if not (type(c) is DenseConnect and type(self.reducer) is BaseReduce and isinstance(s, Tensor)
                and not s.is_sparse and s.dim() == 3 and adj.dim() == 3 and x.dim() == 3 and s.is_cuda):
            return None
        if s.size(0) != adj.size(0):
            raise ValueError("Assignment and adjacency batch sizes do not match: "
                             f"got s.size(0)={s.size(0)} and adj.size(0)={adj.size(0)}.")
        flags = K.dense_flags(c.remove_self_loops, c.degree_norm, c.adj_transpose, c.edge_weight_norm)
        if torch.is_grad_enabled() and (s.requires_grad or adj.requires_grad or x.requires_grad):
            # training: batches of small graphs keep the fused kernel and get its one-launch backward (the adjacency
            # gets no gradient there, edge_weight_norm is not differentiated there: those keep the operator path)
            if (adj.requires_grad or c.edge_weight_norm or out_x is not None or out_adj is not None
                    or s.dtype != torch.float32 or x.dtype != torch.float32 or adj.dtype != torch.float32
                    or not K.dense_pool_is_small(s.size(0), s.size(1), s.size(2), x.size(2))):
                return None
            # want_diff_losses = (link_scale, ent_scale): DiffPool's two losses ride along as a last value [2]
            # (the auxiliary losses come as functions.LossPair: two 0-dim outputs of the fused Function)
            x_pool, raw, adj_pool, terms, diff = Fn.dense_pool_small(
                s, adj, x, flags, want_raw, want_mincut_terms, want_diff_losses, getattr(so, "_graph_sizes", None),
                loss_scalars=True)
            res = (x_pool, raw if want_raw else None, adj_pool)
            if want_mincut_terms:
                res = res + (terms,)
            return res + (diff,) if want_diff_losses is not None else res
        # out_x / out_adj (optional, float32 [B,K,F] / [B,K,K]): the kernels write the pooled outputs straight into
        # caller memory, e.g. the next slot of distributed.PackedGather's send buffer (no pack copy before the RCCL call)
        diff = None
        if (want_diff_losses is not None and not want_mincut_terms and out_x is None and out_adj is None
                and s.dtype == torch.float32 and x.dtype == torch.float32 and adj.dtype == torch.float32):
            # DiffPool, inference, a batch of small graphs (r6): both losses from the pooling launch's per-graph records
            # (stats is None when the batch takes another kernel: the caller's own tail computes them then)
            out = K.dense_pool(s, adj, x, flags, want_raw=want_raw, want_post=True,
                               graph_sizes=getattr(so, "_graph_sizes", None), diff_stats=True)
            if out[3] is not None:
                diff = K.diffpool_stats_tail(out[3], want_diff_losses[0], want_diff_losses[1])
        else:
            out = K.dense_pool(s, adj, x, flags, want_raw=want_raw, want_post=True,
                               graph_sizes=getattr(so, "_graph_sizes", None), out_x=out_x, out_adj=out_adj,
                               mincut_terms=want_mincut_terms)
        x_pool, raw, adj_pool = out[:3]
        # fp32 arithmetic (fp64 when an operand is float64); results carry the dtypes the reference's ATen ops would return
        res = (like_input_dtype(x_pool, x), like_input_dtype(raw, s), like_input_dtype(adj_pool, s))
        # want_mincut_terms: a fourth value, the [2,B] loss tails from inside the pooling kernel (None when the batch
        # does not take the one-wave-per-graph kernel); want_diff_losses: a last value, None on this (no-grad) path --
        # the caller's own inference tail computes the two losses
        if want_mincut_terms:
            res = res + (out[3],)
        return res + (diff,) if want_diff_losses is not None else res

    def _finalize_sparse_output(self, x_pool: Tensor, adj_pool: Tensor, batch: Optional[Tensor],
                                batch_pooled: Optional[Tensor], so: SelectOutput):
        """[B,K,F] / [B,K,K] -> compact block-diagonal representation restricted to the supernodes that
        own at least one node (reference src.py:500-557); the edge extraction + renumbering is one
        count->fill kernel pair."""
        B, Kc = adj_pool.size(0), adj_pool.size(1)
        x_flat = x_pool.reshape(-1, x_pool.size(-1))
        out_mask = so.out_mask
        if batch_pooled is None and batch is not None:
            batch_pooled = self.reducer.reduce_batch(so, batch)
        if batch_pooled is None and B > 1:
            batch_pooled = build_pooled_batch(B, Kc, x_pool.device)
        if batch_pooled is None and out_mask is not None:
            batch_pooled = torch.zeros(B * Kc, dtype=torch.long, device=x_pool.device)
        if out_mask is None:
            ei, ew = Fn.block_diag_edges(adj_pool)
            return x_flat, ei, ew, batch_pooled
        valid = out_mask.reshape(-1)
        idx = valid.nonzero(as_tuple=True)[0]
        relabel = torch.full((B * Kc,), -1, dtype=torch.long, device=x_pool.device)
        relabel[idx] = torch.arange(idx.numel(), device=x_pool.device)
        ei, ew = Fn.block_diag_edges(adj_pool, relabel, inverse=idx)
        return x_flat[idx], ei, ew, batch_pooled[valid]


class Precoarsenable:
    def precoarsening(self, **kwargs) -> PoolingOutput:
        raise NotImplementedError("Precoarsening is not supported by this pooler.")

    def multi_level_precoarsening(self, levels: int, edge_index=None, edge_weight: Optional[Tensor] = None, *,
                                  batch: Optional[Tensor] = None, num_nodes: Optional[int] = None,
                                  **kwargs) -> List[PoolingOutput]:
        """Greedy roll-out of ``levels`` coarsening steps (reference src.py:570-622)."""
        if levels < 1:
            raise ValueError(f"'levels' must be >= 1, got {levels}.")
        clear = getattr(self, "clear_cache", None)
        out = []
        for _ in range(levels):
            if callable(clear):
                clear()  # a cached SelectOutput of the previous level has the wrong size
            pooled = self.precoarsening(edge_index=edge_index, edge_weight=edge_weight, batch=batch,
                                        num_nodes=num_nodes, **kwargs)
            out.append(pooled)
            nxt = pooled.as_data()
            edge_index, edge_weight, batch, num_nodes = nxt.edge_index, nxt.edge_weight, nxt.batch, nxt.num_nodes
        if callable(clear):
            clear()
        return out


class BasePrecoarseningMixin(Precoarsenable):
    """select once + reduce_batch + connect, no features (reference src.py:625-692)."""

    def _precoarsening_from_select_output(self, so: SelectOutput, edge_index, edge_weight: Optional[Tensor] = None,
                                          *, batch: Optional[Tensor] = None, **kwargs) -> PoolingOutput:
        if batch is None:
            batch = getattr(so, "batch", None)
            if batch is None:
                batch = torch.zeros(so.num_nodes, dtype=torch.long, device=so.s.device)
            so.batch = batch
        batch_pooled = self.reducer.reduce_batch(select_output=so, batch=batch)
        connector = getattr(self, "preconnector", self.connector)
        ei, ew = connector(so=so, edge_index=edge_index, edge_weight=edge_weight, batch=batch,
                           batch_pooled=batch_pooled, **kwargs)
        return PoolingOutput(edge_index=ei, edge_weight=ew, batch=batch_pooled, so=so)

    def precoarsening(self, edge_index=None, edge_weight: Optional[Tensor] = None, *,
                      batch: Optional[Tensor] = None, num_nodes: Optional[int] = None, **kwargs) -> PoolingOutput:
        if edge_index is None:
            raise ValueError("edge_index cannot be None for precoarsening.")
        so = self.select(edge_index=edge_index, edge_weight=edge_weight, batch=batch, num_nodes=num_nodes,
                         **kwargs)
        return self._precoarsening_from_select_output(so=so, edge_index=edge_index, edge_weight=edge_weight,
                                                      batch=batch, **kwargs)
