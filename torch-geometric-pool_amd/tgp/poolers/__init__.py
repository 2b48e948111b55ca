"""Pooling layers of the north-star path and the ``get_pooler`` factory
(reference tgp/poolers/{__init__,topk,graclus,ndp,diffpool,mincut}.py).

Only the five poolers named by the hot path are built: ``topk``, ``graclus``, ``ndp``, ``diff``,
``mincut`` (+ ``diff_u`` / ``mincut_u``).  Every other alias of the reference raises the same
``ValueError("Unknown pooler_name=...")`` an unknown name does.
"""
from __future__ import annotations

import inspect
import weakref
from typing import Callable, List, Optional, Union

import torch
from torch import Tensor

from .. import functions as Fn
from .. import kernels as K
from ..connect import DenseConnect, KronConnect, SparseConnect
from ..lift import BaseLift
from ..reduce import BaseReduce
from ..select import GraclusSelect, MLPSelect, NDPSelect, SelectOutput, TopkSelect
from ..src import BasePrecoarseningMixin, DenseSRCPooling, PoolingOutput, SRCPooling
from ..utils.ops import batch_info, is_dense_adj
from ..utils.losses import (
    _MinCutTermsFn,
    entropy_loss,
    link_pred_loss,
    mincut_loss,
    mincut_loss_terms,
    orthogonality_loss,
    sparse_link_pred_loss,
    sparse_mincut_loss,
    unbatched_entropy_loss,
    unbatched_orthogonality_loss,
)
from ..utils.ops import connectivity_to_edge_index, postprocess_adj_pool_dense
from ..functions import LossPair

import os as _os

# r6: batched dense poolers on a SPARSE input whose graphs are large and sparse take the un-padded rows route (no [B,N,N]
# adjacency): TGP_ROWS_ROUTE=0 keeps the densifying route; the density bound is entries / (N x longest graph)
_ROWS_ROUTE = _os.environ.get("TGP_ROWS_ROUTE", "1") != "0"
# (a number: THE bound; None: the measured default 0.03 + 8 / K, at most 0.3 -- the rows route costs E x K gathered floats,
#  the densifying route N x Nmax x K flops on the matrix cores plus two passes over the dense adjacency, so the break-even
#  density falls with K: K = 256 ties at 5.8 %, K <= 128 still wins at 11-29 %, profiles/r06_rows_route_crossover.txt)
_ROWS_ROUTE_DENSITY = float(_os.environ["TGP_ROWS_ROUTE_DENSITY"]) if "TGP_ROWS_ROUTE_DENSITY" in _os.environ else None


def _rows_route_density(k: int) -> float:
    if _ROWS_ROUTE_DENSITY is not None:
        return float(_ROWS_ROUTE_DENSITY)
    return min(0.3, 0.03 + 8.0 / max(int(k), 1))
# A/B switch (read once): 0 keeps the selector and the pooling as two autograd nodes in training
_FOLD_TRAINING = _os.environ.get("TGP_FOLD_TRAINING", "1") != "0"
# ... 0 densifies sparse inputs (to_dense_batch + to_dense_adj) in front of the fused inference call as before
_FOLD_SPARSE_INPUTS = _os.environ.get("TGP_FOLD_SPARSE_INPUTS", "1") != "0"


# =============================================================================== sparse poolers
class TopkPooling(SRCPooling):
    r"""Top-k pooling: keep the best-scoring ``ceil(ratio * n)`` nodes of every graph, gate their
    features by the score, keep the induced subgraph (reference poolers/topk.py:14-195)."""

    def __init__(self, in_channels: int, ratio: Union[int, float] = 0.5, min_score: Optional[float] = None,
                 multiplier: float = 1.0, nonlinearity: Union[str, Callable] = "tanh",
                 lift: str = "precomputed", s_inv_op: str = "transpose", connect_red_op: str = "sum",
                 lift_red_op: str = "sum", remove_self_loops: bool = True, degree_norm: bool = False,
                 edge_weight_norm: bool = False):
        super().__init__(
            selector=TopkSelect(in_channels=in_channels, ratio=ratio, min_score=min_score, act=nonlinearity,
                                s_inv_op=s_inv_op),
            reducer=BaseReduce(),
            lifter=BaseLift(matrix_op=lift, reduce_op=lift_red_op),
            connector=SparseConnect(reduce_op=connect_red_op, degree_norm=degree_norm,
                                    edge_weight_norm=edge_weight_norm, remove_self_loops=remove_self_loops))
        self.multiplier = multiplier

    def forward(self, x: Tensor, adj=None, edge_weight: Optional[Tensor] = None,
                so: Optional[SelectOutput] = None, batch: Optional[Tensor] = None,
                attn: Optional[Tensor] = None, lifting: bool = False, **kwargs):
        if lifting:
            return self.lift(x_pool=x, so=so)
        if self._one_node_training(x, edge_weight, attn):
            out = self._forward_one_node(x, adj, edge_weight, batch)
            if out is not None:
                return out
        so = self.select(x=x if attn is None else attn, batch=batch)
        fused = self.reduce_connect(x, adj, edge_weight, so, batch)  # batches of small graphs, inference: ONE launch
        if fused is not None:
            x_pool, batch_pool, ei, ew = fused
            if self.multiplier != 1:
                x_pool = self.multiplier * x_pool
            return PoolingOutput(x=x_pool, edge_index=ei, edge_weight=ew, batch=batch_pool, so=so)
        x_pool, batch_pool = self.reduce(x=x, so=so, batch=batch)
        if self.multiplier != 1:
            x_pool = self.multiplier * x_pool
        ei, ew = self.connect(so=so, edge_index=adj, edge_weight=edge_weight, batch_pooled=batch_pool)
        return PoolingOutput(x=x_pool, edge_index=ei, edge_weight=ew, batch=batch_pool, so=so)

    def _one_node_training(self, x, edge_weight, attn) -> bool:
        """Training on device tensors in the selector's fused-score mode: the forward is the inference call and ONE
        autograd node carries the gradient to x and the projection (Fn.topk_pool_train)."""
        sel = self.selector
        return bool(_FOLD_TRAINING and attn is None and not self.cached and torch.is_grad_enabled()
                    and isinstance(x, Tensor) and x.is_cuda and x.dim() == 2 and x.size(0) > 0
                    and type(sel) is TopkSelect and sel.weight is not None and sel.min_score is None
                    and sel.ratio is not None and sel._fused_act is not None and type(self.reducer) is BaseReduce
                    and (x.requires_grad or sel.weight.requires_grad)
                    and not (edge_weight is not None and edge_weight.requires_grad)
                    and not torch.cuda.is_current_stream_capturing()
                    and K.topk_pool_bwd_fits(x, sel.weight))

    def _forward_one_node(self, x, adj, edge_weight, batch):
        sel = self.selector
        with torch.no_grad():
            so = self.select(x=x, batch=batch)
            if not so.is_sparse or so.num_supernodes == 0 or so.s._nnz() != so.num_supernodes:
                return None
            fused = self.reduce_connect(x, adj, edge_weight, so, batch)
            if fused is not None:
                x_pool, batch_pool, ei, ew = fused
            else:
                x_pool, batch_pool = self.reduce(x=x, so=so, batch=batch)
        index = so.s.indices()
        x_pool, values = Fn.topk_pool_train(x, sel.weight, x_pool, so.weight, index[0], index[1],
                                            sel._fused_act == "tanh")
        # S with the tracked values: what reads so.s downstream (Lift, a user's loss) reaches the projection through them
        so.s = torch.sparse_coo_tensor(index, values, so.s.size(), is_coalesced=True)
        so._hold_values(values)
        if self.multiplier != 1:
            x_pool = self.multiplier * x_pool
        if fused is None:
            ei, ew = self.connect(so=so, edge_index=adj, edge_weight=edge_weight, batch_pooled=batch_pool)
        return PoolingOutput(x=x_pool, edge_index=ei, edge_weight=ew, batch=batch_pool, so=so)

    def extra_repr_args(self) -> dict:
        return {"multiplier": self.multiplier}


class GraclusPooling(BasePrecoarseningMixin, SRCPooling):
    r"""Graclus pooling: greedy pairwise matching, features summed per pair, edges coalesced
    (reference poolers/graclus.py:14-159)."""

    def __init__(self, lift: str = "precomputed", s_inv_op: str = "transpose", connect_red_op: str = "sum",
                 lift_red_op: str = "sum", cached: bool = False, remove_self_loops: bool = True,
                 degree_norm: bool = False, edge_weight_norm: bool = False):
        super().__init__(
            selector=GraclusSelect(s_inv_op=s_inv_op),
            reducer=BaseReduce(),
            lifter=BaseLift(matrix_op=lift, reduce_op=lift_red_op),
            connector=SparseConnect(reduce_op=connect_red_op, remove_self_loops=remove_self_loops,
                                    degree_norm=degree_norm, edge_weight_norm=edge_weight_norm),
            cached=cached)
        self.cached = cached

    def forward(self, x: Tensor, adj=None, edge_weight: Optional[Tensor] = None,
                so: Optional[SelectOutput] = None, batch: Optional[Tensor] = None, lifting: bool = False,
                **kwargs):
        if lifting:
            return self.lift(x_pool=x, so=so)
        so = self.select(edge_index=adj, edge_weight=edge_weight, num_nodes=x.size(0), batch=batch)
        fused = self.reduce_connect(x, adj, edge_weight, so, batch)  # batches of small graphs, inference: ONE launch
        if fused is not None:
            x_pool, batch_pool, ei, ew = fused
            return PoolingOutput(x=x_pool, edge_index=ei, edge_weight=ew, batch=batch_pool, so=so)
        x_pool, batch_pool = self.reduce(x=x, so=so, batch=batch)
        ei, ew = self.connect(edge_index=adj, so=so, edge_weight=edge_weight, batch_pooled=batch_pool)
        return PoolingOutput(x=x_pool, edge_index=ei, edge_weight=ew, batch=batch_pool, so=so)

    def extra_repr_args(self) -> dict:
        return {"cached": self.cached}


class NDPPooling(BasePrecoarseningMixin, SRCPooling):
    r"""Node Decimation Pooling: spectral +-1 partition, keep one side, Kron-reduce the Laplacian
    (reference poolers/ndp.py:14-142)."""

    def __init__(self, lift: str = "precomputed", s_inv_op: str = "transpose", lift_red_op: str = "sum",
                 cached: bool = False):
        super().__init__(selector=NDPSelect(s_inv_op=s_inv_op), reducer=BaseReduce(),
                         lifter=BaseLift(matrix_op=lift, reduce_op=lift_red_op), connector=KronConnect(),
                         cached=cached)
        self.cached = cached

    def forward(self, x: Tensor, adj=None, edge_weight: Optional[Tensor] = None,
                so: Optional[SelectOutput] = None, batch: Optional[Tensor] = None, lifting: bool = False,
                **kwargs):
        if lifting:
            return self.lift(x_pool=x, so=so)
        so = self.select(edge_index=adj, edge_weight=edge_weight, batch=batch, num_nodes=x.size(0))
        x_pool, batch_pool = self.reduce(x=x, so=so, batch=batch)
        # batch: the Kron reduction is taken per graph (block-batched kernel); the reference's connector ignores it
        ei, ew = self.connect(edge_index=adj, so=so, edge_weight=edge_weight, batch=batch)
        return PoolingOutput(x=x_pool, edge_index=ei, edge_weight=ew, batch=batch_pool, so=so)

    def extra_repr_args(self) -> dict:
        return {"cached": self.cached}


# =============================================================================== dense poolers
class _DenseMLPPooling(DenseSRCPooling):
    """What DiffPool and MinCut share: MLPSelect + BaseReduce + DenseConnect wiring, the batched /
    unbatched control flow, and the optional block-diagonal sparse output."""

    def __init__(self, in_channels, k, act, dropout, remove_self_loops, degree_norm, edge_weight_norm,
                 adj_transpose, lift, s_inv_op, batched, sparse_output, cache_preprocessing):
        super().__init__(
            selector=MLPSelect(in_channels=in_channels, k=k, batched_representation=batched, act=act,
                               dropout=dropout, s_inv_op=s_inv_op),
            reducer=BaseReduce(),
            lifter=BaseLift(matrix_op=lift),
            connector=DenseConnect(remove_self_loops=remove_self_loops, degree_norm=degree_norm,
                                   adj_transpose=adj_transpose, edge_weight_norm=edge_weight_norm,
                                   sparse_output=sparse_output),
            adj_transpose=adj_transpose, cache_preprocessing=cache_preprocessing, batched=batched,
            sparse_output=sparse_output)

    # hooks filled in by the two poolers -------------------------------------------------
    def _batched_connect_and_loss(self, x, adj, so, mask, edge_weight, batch, batch_pooled):
        raise NotImplementedError

    def compute_sparse_loss(self, edge_index, edge_weight, S, batch) -> dict:
        raise NotImplementedError

    _loss_needs_raw = False  # MinCut's cut loss reads the raw S^T A S

    def _fused_diff_scales(self, adj, mask, adj_numel=None):
        """(link_scale, ent_scale) when the pooler's two losses can ride on the fused training call (DiffPool).
        ``adj_numel``: B * N * N of the padded adjacency when ``adj`` is not that tensor (sparse inputs)."""
        return None

    def _loss_from_fused(self, adj, so, mask, raw, terms=None, diff=None) -> dict:
        raise NotImplementedError

    def _lift(self, x, so, batch, batch_pooled):
        return self.lift(x_pool=x, so=so, batch=batch, batch_pooled=batch_pooled)

    def _select_reduce_connect(self, x, adj, mask, want_batch=False):
        """Inference on a batch of small graphs with a single-Linear selector: Select + Reduce + Connect as ONE launch
        (tgp_dense_pool_select_f32: S is formed in the pooling kernel's registers and written once).  Returns
        ``(SelectOutput, (x_pool, raw, adj_pool[, terms]), pooled batch vector or None)`` or None when the case is not
        that one."""
        from .. import kernels as K
        sel, c = self.selector, self.connector
        lins = getattr(getattr(sel, "mlp", None), "lins", None)
        if (type(sel) is not MLPSelect or lins is None or len(lins) != 1 or type(c) is not DenseConnect
                or type(self.reducer) is not BaseReduce or not (isinstance(x, Tensor) and isinstance(adj, Tensor))
                or x.dim() != 3 or adj.dim() != 3 or not x.is_cuda or x.dtype != torch.float32
                or adj.dtype != torch.float32 or (mask is not None and mask.dtype != torch.bool)):
            return None
        last = lins[0]
        if torch.is_grad_enabled() and (x.requires_grad or adj.requires_grad or last.weight.requires_grad
                                        or (last.bias is not None and last.bias.requires_grad)):
            return None
        if (last.weight.dtype != torch.float32 or adj.shape != (x.size(0), x.size(1), x.size(1))
                or not K.dense_pool_is_small(x.size(0), x.size(1), last.weight.size(0), x.size(2))):
            return None
        flags = K.dense_flags(c.remove_self_loops, c.degree_norm, c.adj_transpose, c.edge_weight_norm)
        diff_scales = self._fused_diff_scales(adj, mask)
        out = K.dense_pool_select(
            x, adj, last.weight.detach(), None if last.bias is None else last.bias.detach(), mask, flags,
            want_raw=self._loss_needs_raw, mincut_terms=self._loss_needs_raw, want_batch=want_batch,
            diff_stats=diff_scales is not None)
        s, x_pool, raw, adj_pool, terms = out[:5]
        so = SelectOutput(s=s, s_inv_op=sel.s_inv_op, in_mask=mask)
        fused = (x_pool, raw, adj_pool) + ((terms,) if self._loss_needs_raw else ())
        if diff_scales is not None:  # DiffPool (r6): both losses from the launch's per-graph records, one tail launch
            fused = fused + (K.diffpool_stats_tail(terms, diff_scales[0], diff_scales[1]),)
        return so, fused, (out[5] if want_batch else None)

    def _select_reduce_connect_sparse(self, x, edge_index, edge_weight, batch):
        """A sorted batch of small graphs that arrives as PyG hands it over (x [N,F], a row-sorted
        ``edge_index``): Select + Reduce + Connect + loss tails straight from the un-padded batch in ONE launch
        (tgp_dense_pool_select_sparse_f32: every graph's adjacency tile is built in LDS from its edges) -- neither
        ``to_dense_batch`` nor ``to_dense_adj`` runs and no [B,N,N] tensor exists.  Training takes the same launch as one
        autograd node (functions._SelectPoolSparseFn: the padded tensors the backward kernels read are side outputs).  Only
        for poolers whose losses come out of the kernel (MinCut); returns ``(SelectOutput, fused, pooled batch vector)``
        or None."""
        from .. import kernels as K
        sel, c = self.selector, self.connector
        lins = getattr(getattr(sel, "mlp", None), "lins", None)
        if (not _FOLD_SPARSE_INPUTS or self.cache_preprocessing
                or type(sel) is not MLPSelect or lins is None or len(lins) != 1 or type(c) is not DenseConnect
                or type(self.reducer) is not BaseReduce or not (isinstance(x, Tensor) and isinstance(edge_index, Tensor))
                or x.dim() != 2 or not x.is_cuda or x.dtype != torch.float32 or edge_index.dim() != 2
                or edge_index.size(0) != 2 or edge_index.dtype != torch.long or not edge_index.is_cuda
                or batch is None or batch.dtype != torch.long or batch.numel() != x.size(0) or x.size(0) == 0
                or (edge_weight is not None and (edge_weight.dim() != 1 or edge_weight.dtype != torch.float32))):
            return None
        last = lins[0]
        if torch.is_grad_enabled() and edge_weight is not None and edge_weight.requires_grad:
            return None  # (the edge weights get no gradient from the fused backward)
        training = torch.is_grad_enabled() and (x.requires_grad or last.weight.requires_grad
                                                or (last.bias is not None and last.bias.requires_grad))
        # (a pooler whose losses need the dense adjacency -- DiffPool's link loss -- gets it as a side output of the launch)
        if training and not (_FOLD_TRAINING and not c.edge_weight_norm
                             and K.mlp_select_bwd_fits(last.weight.size(0), x.size(1))):
            return None
        # the one-launch kernel walks every graph's edge range: the list must be grouped by ascending source node (what
        # PyG's loaders produce).  Known per tensor object once it has been looked at; a NEW list gets its ranges and
        # its verdict from one facts launch enqueued here, in front of the wait for the batch facts: the pooling kernel
        # is launched on the ranges at once and the verdict is read behind it (a list that fails it is remembered, the
        # outputs are dropped and the densified path below takes the call)
        known = K._rows_sorted_memo(edge_index) if edge_index.size(1) > 1 else True
        if known is False:
            return None
        pending = None
        if known is None:
            from ..utils.ops import prefetch_batch_info
            prefetch_batch_info(batch)
            pending = K.edge_facts_launch(edge_index, batch)
            if pending is None:
                return None
        info = batch_info(batch)  # (memoised per batch vector)
        if (not info.is_sorted or last.weight.dtype != torch.float32
                or not K.dense_pool_is_small(info.num_graphs, info.max_nodes, last.weight.size(0), x.size(1))):
            if pending is not None:
                K.edge_facts_finish(pending, edge_index, info.ptr)  # (the launch is out: keep what it found)
            return None
        edge_ptr = pending[2] if pending is not None else K.graph_edge_ptr(edge_index, info.ptr)
        flags = K.dense_flags(c.remove_self_loops, c.degree_norm, c.adj_transpose, c.edge_weight_norm)
        ad = None
        if training:  # one autograd node; the padded tensors the backward reads are side outputs of the same launch
            from .. import functions as Fn
            diff_scales = None
            if not self._loss_needs_raw:
                self._known_nodes = x.size(0)
                diff_scales = self._fused_diff_scales(None, None, info.num_graphs * info.max_nodes * info.max_nodes)
                if diff_scales is None:
                    if pending is not None:
                        K.edge_facts_finish(pending, edge_index, info.ptr)
                    return None
            s, mask, x_pool, raw, adj_pool, terms, bp = Fn.select_pool_sparse(
                x, last.weight, last.bias, edge_index, edge_weight, batch, info.ptr, edge_ptr, info.num_graphs,
                info.max_nodes, flags, self.adj_transpose, self._loss_needs_raw, diff_scales, info.sizes)
        elif self._loss_needs_raw:
            s, mask, x_pool, raw, adj_pool, terms, bp = K.dense_pool_select_sparse(
                x, edge_index, edge_weight, batch, info.ptr, edge_ptr, info.num_graphs, info.max_nodes,
                last.weight.detach(), None if last.bias is None else last.bias.detach(), flags, self.adj_transpose,
                want_raw=True, mincut_terms=True)
        else:  # DiffPool, inference (r6): both losses from per-graph records of the same launch -- no dense adjacency
            s, mask, x_pool, raw, adj_pool, terms, bp, dstats = K.dense_pool_select_sparse(
                x, edge_index, edge_weight, batch, info.ptr, edge_ptr, info.num_graphs, info.max_nodes,
                last.weight.detach(), None if last.bias is None else last.bias.detach(), flags, self.adj_transpose,
                want_raw=False, mincut_terms=False, diff_stats=True)
            numel = info.num_graphs * info.max_nodes * info.max_nodes  # adj.numel() of the padded batch (losses.py:651)
            link_scale = self.link_loss_coeff / numel if self.normalize_loss is True else self.link_loss_coeff
            terms = K.diffpool_stats_tail(dstats, float(link_scale), float(self.ent_loss_coeff) / x.size(0))
        if pending is not None and not K.edge_facts_finish(pending, edge_index, info.ptr):
            return None  # rows not sorted: what the kernel computed on clamped ranges is dropped
        so = SelectOutput(s=s, s_inv_op=sel.s_inv_op, in_mask=mask)
        so._graph_sizes = info.sizes
        if not self._loss_needs_raw:
            # (both losses came with the fused call: a LossPair -- training -- or a [2] tensor in the slot of `diff`)
            return so, (x_pool, None, adj_pool, None, terms), bp, None
        return so, (x_pool, raw, adj_pool, terms, None), bp, None

    def _select_reduce_connect_train(self, x, adj, mask, graph_sizes, want_batch=False):
        """Training on a batch of small graphs with a single-Linear selector: Select + Reduce + Connect + loss tails as
        ONE autograd node (functions._SelectPoolSmallFn: one forward launch, two backward launches; the selector and the
        pooling were two nodes with eight + one backward launches and an accumulation of the two gradients of X).
        Returns ``(SelectOutput, fused)`` like :meth:`_select_reduce_connect`, the losses as a ``LossPair``."""
        from .. import functions as Fn, kernels as K
        sel, c = self.selector, self.connector
        lins = getattr(getattr(sel, "mlp", None), "lins", None)
        if (type(sel) is not MLPSelect or lins is None or len(lins) != 1 or type(c) is not DenseConnect
                or type(self.reducer) is not BaseReduce or not (isinstance(x, Tensor) and isinstance(adj, Tensor))
                or x.dim() != 3 or adj.dim() != 3 or not x.is_cuda or x.dtype != torch.float32
                or adj.dtype != torch.float32 or (mask is not None and mask.dtype != torch.bool)
                or not torch.is_grad_enabled() or adj.requires_grad or c.edge_weight_norm):
            return None
        last = lins[0]
        if (last.weight.dtype != torch.float32 or adj.shape != (x.size(0), x.size(1), x.size(1))
                or not K.dense_pool_is_small(x.size(0), x.size(1), last.weight.size(0), x.size(2))
                or not K.mlp_select_bwd_fits(last.weight.size(0), x.size(2))):
            return None
        flags = K.dense_flags(c.remove_self_loops, c.degree_norm, c.adj_transpose, c.edge_weight_norm)
        diff_scales = self._fused_diff_scales(adj, mask)
        s, x_pool, raw, adj_pool, pair, bp = Fn.select_pool_small(
            x, adj, last.weight, last.bias, mask, flags, self._loss_needs_raw, self._loss_needs_raw, diff_scales,
            graph_sizes, want_batch)
        so = SelectOutput(s=s, s_inv_op=sel.s_inv_op, in_mask=mask)
        fused = (x_pool, raw if self._loss_needs_raw else None, adj_pool)
        if self._loss_needs_raw:
            fused = fused + (pair,)
        if diff_scales is not None:
            fused = fused + (pair,)
        return so, fused, bp

    def _select_reduce_connect_large(self, x, adj, mask, graph_sizes, so=None, symmetry=None):
        """Training on a padded batch whose graphs are beyond the one-wave kernels (C2-sized), r6: Select (single-Linear
        selector; otherwise ``so`` holds S) + Reduce + Connect + post-processing + the pooler's two losses as ONE
        autograd node (functions._PoolLargeFn: ~10 forward and ~13 backward launches where the operator-by-operator graph
        ran 65-71).  Returns ``(SelectOutput, fused, None)`` like :meth:`_select_reduce_connect_train`, or None."""
        from .. import functions as Fn, kernels as K
        sel, c = self.selector, self.connector
        if (not _FOLD_TRAINING or type(c) is not DenseConnect or type(self.reducer) is not BaseReduce
                or not (isinstance(x, Tensor) and isinstance(adj, Tensor)) or x.dim() != 3 or adj.dim() != 3
                or not x.is_cuda or x.dtype != torch.float32 or adj.dtype != torch.float32
                or (mask is not None and mask.dtype != torch.bool) or not torch.is_grad_enabled()
                or adj.requires_grad or c.edge_weight_norm or adj.shape != (x.size(0), x.size(1), x.size(1))
                or x.size(0) == 0 or x.size(1) == 0 or x.size(2) == 0):
            return None
        lins = getattr(getattr(sel, "mlp", None), "lins", None)
        weight = bias = s = None
        if so is None:
            if type(sel) is not MLPSelect or lins is None or len(lins) != 1 or lins[0].weight.dtype != torch.float32:
                return None
            weight, bias = lins[0].weight, lins[0].bias
            k = weight.size(0)
            if not (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)):
                return None
        else:
            s = so.s
            if not (isinstance(s, Tensor) and s.dim() == 3 and s.dtype == torch.float32 and s.is_cuda
                    and s.shape[:2] == x.shape[:2] and (s.requires_grad or x.requires_grad)):
                return None
            k = s.size(2)
        if k == 0 or k > 4096 or K.dense_pool_is_small(x.size(0), x.size(1), k, x.size(2)):
            return None
        flags = K.dense_flags(c.remove_self_loops, c.degree_norm, c.adj_transpose, c.edge_weight_norm)
        mode, scales = (1, (0.0, 0.0)) if self._loss_needs_raw else (0, (0.0, 0.0))
        if not self._loss_needs_raw:
            diff_scales = self._fused_diff_scales(adj, mask)
            if diff_scales is None:
                return None
            mode, scales = 2, diff_scales
        s_out, x_pool, raw, adj_pool, pair = Fn.pool_large(x, adj, weight, bias, mask, s, flags, mode, scales, graph_sizes,
                                                           symmetry() if callable(symmetry) else symmetry)
        if so is None:
            so = SelectOutput(s=s_out, s_inv_op=sel.s_inv_op, in_mask=mask)
        fused = (x_pool, raw if self._loss_needs_raw else None, adj_pool, pair)
        return so, fused, None

    @staticmethod
    def _adj_symmetry(edge_index, edge_weight, dense_adj, batch):
        """kernels.AdjSymmetry for the adjacency forward() densified from this edge list, or None."""
        from .. import kernels as K
        if not (isinstance(edge_index, Tensor) and edge_index.dim() == 2 and edge_index.size(0) == 2
                and edge_index.dtype == torch.long and edge_index.is_cuda and batch is not None
                and (edge_weight is None or isinstance(edge_weight, Tensor))):
            return None
        info = batch_info(batch)
        if not info.is_sorted or info.num_graphs != dense_adj.size(0):
            return None
        return K.AdjSymmetry(edge_index, edge_weight, dense_adj, batch, info.ptr)

    def _unbatched_fused(self, x, edge_index, edge_weight, batch, batched_out: bool = False):
        """Reduce, Connect and both auxiliary losses on the UN-padded rows, r6: ONE S^T [A S | X | S] product
        (tgp_segment_gemm_tn3_f32) behind the CSR SpMM -- the mincut numerator is trace(S_g^T (A S)_g), the link
        residual sum_e w_e^2 - 2 sum_g trace(raw_g) + sum_g |S_g^T S_g|^2 -- instead of per-edge dot products and
        index_add scatters (reference utils/losses.py:73-127, 204-240, 661-708; dense_conn.py:140-208;
        base_reduce.py:170-182): 40-45 launches -> ~10 per forward in the unbatched mode.  Under autograd the same
        forward is ONE autograd node (functions._PoolUnbatchedFn).

        ``batched_out`` (r6, late): the BATCHED poolers take the same route for a sparse input whose graphs are too
        large for the one-launch kernels and sparse enough (E <= d x N x longest graph, d = TGP_ROWS_ROUTE_DENSITY or the
        measured default 0.03 + 8 / K, at most 0.3):
        no [B,N,N] adjacency is ever built (reference src.py:374-452 densifies first; at the C2 shape that is 134 MB
        written and read three times per training step for 0.33 M entries).  The results are those of the batched mode:
        S^T A^T S when adj_transpose (= the transpose of S^T (A S); MinCut's degrees are then in-degrees, sum_i (A q)_i),
        S handed out padded [B,Nmax,K] with its mask, losses normalised as the batched forms do.

        Returns (SelectOutput, x_pool [B,K,F], adj_pool [B,K,K], pooled batch vector, losses); ``(SelectOutput,)`` when
        only Select could be done here (unbatched mode); None when the case is not this one."""
        from .. import kernels as K
        from .. import functions as Fn
        c, sel = self.connector, self.selector
        if (type(c) is not DenseConnect or type(self.reducer) is not BaseReduce
                or (self.sparse_output and not batched_out)
                or not (isinstance(x, Tensor) and isinstance(edge_index, Tensor))
                or x.dim() != 2 or not x.is_cuda or x.dtype != torch.float32
                or edge_index.dim() != 2 or edge_index.size(0) != 2 or edge_index.dtype != torch.long
                or not edge_index.is_cuda or edge_index.size(1) == 0 or x.size(0) == 0 or x.size(1) == 0
                or (edge_weight is not None and (not isinstance(edge_weight, Tensor) or edge_weight.dtype != torch.float32
                                                 or edge_weight.numel() != edge_index.size(1)))
                or (batch is not None and (batch.dtype != torch.long or batch.numel() != x.size(0)))):
            return None
        grad = torch.is_grad_enabled()
        if grad and edge_weight is not None and edge_weight.requires_grad:
            return None
        n = x.size(0)
        info = None
        if batch is not None:
            info = batch_info(batch)
            if not info.is_sorted:
                return None
            ptr, max_nodes, nb = info.ptr, info.max_nodes, info.num_graphs
        else:
            ptr, max_nodes, nb = Fn._whole_range(n, x.device), n, 1
        lins = getattr(getattr(sel, "mlp", None), "lins", None)
        single = type(sel) is MLPSelect and lins is not None and len(lins) == 1 and lins[0].weight.dtype == torch.float32
        if batched_out:  # every check comes BEFORE Select here: a bail-out must leave the batched flow untouched
            if (not _ROWS_ROUTE or self.cache_preprocessing or type(sel) is not MLPSelect or lins is None
                    or lins[-1].weight.dtype != torch.float32
                    or edge_index.size(1) > _rows_route_density(lins[-1].weight.size(0)) * float(n) * float(max_nodes)
                    or K.dense_pool_is_small(nb, max_nodes, lins[-1].weight.size(0), x.size(1))):
                return None
            k_out = lins[-1].weight.size(0)
            will_train = grad and (x.requires_grad or any(p.requires_grad for p in sel.parameters()))
            if will_train and (not _FOLD_TRAINING or c.edge_weight_norm or k_out > 4096):
                return None
        # the selector: folded into the training node when it is a single Linear, else run in front (its S handed over)
        so = weight = bias = None
        # inference with a single-Linear selector (r6, late): the selector rides in the ONE native call of the forward
        # (kernels.pool_rows_forward); the SelectOutput is built from the S it leaves
        fold_inference = (not grad and single and K._POOL_ROWS_ONE_CALL and x.dtype == torch.float32 and x.is_contiguous()
                          and lins[0].weight.size(0) <= 256)
        if fold_inference:
            weight, bias = lins[0].weight.detach(), None if lins[0].bias is None else lins[0].bias.detach()
            s, training = None, False
        elif not (grad and single and _FOLD_TRAINING):
            so = self.select(x=x, batch=batch) if not batched_out else self.select(x=x)
            s = so.s
            if batched_out and isinstance(s, Tensor) and s.dim() == 3 and s.size(0) == 1:
                s = s[0]  # (MLPSelect in its batched representation treats [N,F] as one padded graph)
            if not (isinstance(s, Tensor) and s.dim() == 2 and s.dtype == torch.float32 and s.size(0) == n):
                return None if batched_out else (so,)  # (the caller goes on with this SelectOutput on the operator path)
            training = grad and (s.requires_grad or x.requires_grad)
        else:
            weight, bias = lins[0].weight, lins[0].bias
            s = None
            training = x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)
            if not training:
                so = self.select(x=x, batch=batch) if not batched_out else self.select(x=x)
                s = so.s[0] if (batched_out and so.s.dim() == 3) else so.s
        k = s.size(1) if s is not None else weight.size(0)
        if training and (not _FOLD_TRAINING or c.edge_weight_norm or k > 4096):
            if batched_out:
                return None
            if so is None:
                so = self.select(x=x, batch=batch)
            return (so,)
        w_in = None if edge_weight is None else edge_weight.reshape(-1)
        ones = w_in is None
        # sorted + duplicate-summed A (what the reference's per-graph `.coalesce()` does), T = A S, then one product grid
        if ones and K.coalesced_memo(edge_index, n):  # (known to be coalesced: no vector of ones is made for the check)
            ei, w = edge_index, None
        else:
            ei, w = Fn.coalesce_sum(edge_index, torch.ones(edge_index.size(1), device=x.device) if ones else w_in.detach(), n)
        unit = ones and ei is edge_index  # (nothing merged: the weights are still all one)
        w_used = None if unit else w
        row_ptr = K.csr_offsets(ei, n)
        transposed = bool(batched_out and self.adj_transpose)
        flags = K.dense_flags(c.remove_self_loops, c.degree_norm, c.adj_transpose if batched_out else False,
                              c.edge_weight_norm)
        mincut = self._loss_needs_raw
        sw2, scales = 0.0, (0.0, 0.0)
        if not mincut:  # DiffPool: sum_e w_e^2 runs over the list as given (duplicates not merged, losses.py:680-690)
            # (batched form, losses.py:644-652: the dense A has the duplicates summed -- the same list after coalescing)
            if batched_out and not unit:
                sw2 = torch.dot(w, w)
            else:
                sw2 = float(edge_index.size(1)) if ones else torch.dot(w_in.detach(), w_in.detach())
            link_scale = float(self.link_loss_coeff)
            if self.normalize_loss is True:
                if batched_out:  # adj.numel() of the padded batch (losses.py:651)
                    denom = nb * max_nodes * max_nodes
                else:
                    denom = sum(v * v for v in info.sizes_host) if info is not None else n * n
                link_scale = link_scale / max(denom, 1)
            scales = (link_scale, float(self.ent_loss_coeff) / n)
        if training:
            sym = K.AdjSymmetry.of_edge_list(edge_index, edge_weight, ei, w_used, row_ptr, n)
            s_out, x_pool, raw, adj_pool, pair = Fn.pool_unbatched(
                x, weight, bias, s, ei, w_used, row_ptr, ptr, batch, max_nodes, flags, 1 if mincut else 2, scales, sw2, sym,
                transposed)
            s_flat = s_out
            both = pair
        else:
            one = K.pool_rows_forward(x, weight if s is None else None, bias if s is None else None, s, row_ptr, ei, w_used,
                                      ptr, max_nodes, transposed, flags, 1 if mincut else 2, scales, sw2)
            if one is None and s is None:  # (not the one-call entry's case after all: the selector runs on its own)
                so = self.select(x=x, batch=batch) if not batched_out else self.select(x=x)
                s = so.s[0] if (batched_out and so.s.dim() == 3) else so.s
        if not training and one is not None:
            s, raw, x_pool, adj_pool = one["s"], one["raw"], one["x_pool"], one["adj_pool"]
            both = one["both"] if mincut else one["lossv"]
            s_flat = s
        elif not training:
            if mincut:  # out-degrees and |S_i|^2 ride along with T = A S
                t, deg, q = K.spmm_csr(row_ptr, ei, w_used, n, s, want_stats=True)
            else:       # DiffPool: the entropy sum over S rides along
                t, ent_part = K.spmm_csr(row_ptr, ei, w_used, n, s, want_stats="entropy")
            raw, x_pool, gram, adj_pool = K.segment_gemm_tn3(s, [t, x, s], ptr, max_nodes, transpose0=transposed,
                                                             post_flags=flags)
            if mincut:
                if transposed:  # in-degrees: sum_j indeg_j q_j = sum_e w_e q[col_e]
                    both = K.mincut_terms_fused(raw, gram, None, q, ptr=ptr, want_means=True,
                                                edges=(row_ptr, ei, w_used))[3]
                else:
                    both = K.mincut_terms_fused(raw, gram, deg, q, ptr=ptr, want_means=True)[3]
            else:
                both = K.diffpool_unbatched_tail(raw, gram, s, sw2, scales[0], scales[1], ent_partials=ent_part)
            s_flat = s
        if batched_out:  # S as the batched mode hands it out: padded [B,Nmax,K] + the node mask (differentiable view of S)
            if n == nb * max_nodes:  # graphs of one size: the padded form is a view, the mask a constant
                s_pad = s_flat.view(nb, max_nodes, k)
                mask = torch.ones(nb, max_nodes, dtype=torch.bool, device=x.device)  # (a tensor of the caller's own)
            else:
                from ..src import to_dense_batch
                s_pad, mask = to_dense_batch(s_flat, batch, max_nodes, nb)
            so = SelectOutput(s=s_pad, s_inv_op=sel.s_inv_op, in_mask=mask)
            if info is not None:
                so._graph_sizes = info.sizes
        elif so is None:
            so = SelectOutput(s=s_flat, s_inv_op=sel.s_inv_op, batch=batch)
        if mincut:
            loss = {"cut_loss": both[0] if self.cut_loss_coeff == 1 else both[0] * self.cut_loss_coeff,
                    "ortho_loss": both[1] if self.ortho_loss_coeff == 1 else both[1] * self.ortho_loss_coeff}
        else:
            loss = {"link_loss": both[0], "entropy_loss": both[1]}
        batch_pool = self.reducer.reduce_batch(so, batch if batch is not None else so.batch)
        return so, x_pool, adj_pool, batch_pool, loss

    def _sizes_for(self, adj):
        """Real nodes per graph of the zero-padded batch ``adj`` belongs to, if forward() densified it itself."""
        hint = getattr(self, "_sizes_hint", None)
        return hint[1] if hint is not None and hint[0]() is adj else None

    def _real_nodes(self, mask):
        """Valid nodes of the padded batch: the host-side count when forward() had one, else a 0-dim device tensor
        (no host round trip either way)."""
        known = getattr(self, "_known_nodes", None)
        return known if known is not None else mask.sum()

    def forward(self, x: Tensor, adj=None, edge_weight: Optional[Tensor] = None,
                so: Optional[SelectOutput] = None, mask: Optional[Tensor] = None,
                batch: Optional[Tensor] = None, batch_pooled: Optional[Tensor] = None, lifting: bool = False,
                **kwargs):
        if lifting:
            return self._lift(x, so, batch, batch_pooled)
        if self.batched:
            if so is None and mask is None and not is_dense_adj(adj):
                self._sizes_hint = None
                sparse = self._select_reduce_connect_sparse(x, adj, edge_weight, batch)
                if sparse is not None:  # straight from the un-padded batch: no densification launches
                    self._known_nodes = x.size(0)
                    so, fused, batch_pool, dense_adj = sparse
                    x_pool, raw, adj_pool, terms, diff = fused
                    loss = self._loss_from_fused(dense_adj if dense_adj is not None else adj, so, None, raw, terms, diff)
                    if self.sparse_output:
                        x_pool, ei, ew, batch_pool = self._finalize_sparse_output(
                            x_pool=x_pool, adj_pool=adj_pool, batch=batch, batch_pooled=batch_pool, so=so)
                        return PoolingOutput(x=x_pool, edge_index=ei, edge_weight=ew, batch=batch_pool, so=so,
                                             loss=loss)
                    return PoolingOutput(x=x_pool, edge_index=adj_pool, so=so, loss=loss)
                rows = self._unbatched_fused(x, adj, edge_weight, batch, batched_out=True)
                if rows is not None:  # large sparse graphs: the un-padded rows route, no dense adjacency (r6)
                    so, x_pool, adj_pool, batch_pool, loss = rows
                    if self.sparse_output:
                        x_pool, ei, ew, batch_pool = self._finalize_sparse_output(
                            x_pool=x_pool, adj_pool=adj_pool, batch=batch, batch_pooled=batch_pool, so=so)
                        return PoolingOutput(x=x_pool, edge_index=ei, edge_weight=ew, batch=batch_pool, so=so,
                                             loss=loss)
                    return PoolingOutput(x=x_pool, edge_index=adj_pool, so=so, loss=loss)
            # number of real nodes behind the padded batch, when the host already knows it (a reduction over the mask
            # costs ~25 us on the device for any mask size)
            self._known_nodes = None
            graph_sizes = None
            sparse_in = False
            if not is_dense_adj(adj) and isinstance(x, Tensor) and x.dim() == 2:
                sparse_in = True
                self._known_nodes = x.size(0)
                if batch is not None and batch.numel() > 0 and x.is_cuda:
                    graph_sizes = batch_info(batch).sizes  # memoised: the densification below asks for it anyway
            elif mask is None and isinstance(x, Tensor) and x.dim() == 3:
                self._known_nodes = x.size(0) * x.size(1)
            sparse_edges = (adj, edge_weight) if sparse_in else None
            x, adj, mask = self._ensure_batched_inputs(x=x, edge_index=adj, edge_weight=edge_weight, batch=batch,
                                                       mask=mask)
            # (the folded kernels also write the pooled batch vector: arange(B).repeat_interleave(K) of the B graphs)
            want_bp = sparse_in and batch is not None and batch.dtype == torch.long and batch.numel() > 0
            folded = self._select_reduce_connect(x, adj, mask, want_bp)
            if folded is None and _FOLD_TRAINING:
                folded = self._select_reduce_connect_train(x, adj, mask, graph_sizes, want_bp)
            symmetry = None
            from .. import kernels as K_
            if sparse_in and _FOLD_TRAINING and torch.is_grad_enabled() and isinstance(adj, Tensor) and adj.dim() == 3:
                # (asked only when the one-node path takes the call: a launch over the entries, answered in its backward)
                dense_adj, ei_in, ew_in = adj, sparse_edges[0], sparse_edges[1]
                symmetry = lambda: self._adj_symmetry(ei_in, ew_in, dense_adj, batch)  # noqa: E731
            elif (not sparse_in and _FOLD_TRAINING and torch.is_grad_enabled() and isinstance(adj, Tensor)
                  and adj.dim() == 3 and adj.is_contiguous()):
                # a dense adjacency the caller holds (e.g. one fixed graph pooled every epoch): one pass over it, once
                # per tensor object, tells whether A = A^T (the backward then forms one N^2 K product less)
                held = adj
                symmetry = lambda: K_.AdjSymmetry.of_dense(held)  # noqa: E731
            if folded is None and _FOLD_TRAINING:  # graphs beyond the one-wave kernels: one autograd node as well (r6)
                folded = self._select_reduce_connect_large(x, adj, mask, graph_sizes, symmetry=symmetry)
            so = folded[0] if folded is not None else self.select(x=x, mask=mask)
            self._sizes_hint = None
            if graph_sizes is not None and graph_sizes.numel() == x.size(0):
                so._graph_sizes = graph_sizes
                self._sizes_hint = (weakref.ref(adj), graph_sizes)  # valid for exactly this adjacency tensor
            diff_scales = self._fused_diff_scales(adj, mask)
            fused = folded[1] if folded is not None else self.reduce_connect(
                x, adj, so, want_raw=self._loss_needs_raw, want_mincut_terms=self._loss_needs_raw,
                want_diff_losses=diff_scales)
            if fused is None and _FOLD_TRAINING:  # a selector with hidden layers made S: the pooling step is one node still
                big = self._select_reduce_connect_large(x, adj, mask, graph_sizes, so=so, symmetry=symmetry)
                if big is not None:
                    fused = big[1]
            if fused is not None:  # Reduce + Connect in one native call (training: batches of small graphs only)
                x_pool, raw, adj_pool = fused[:3]
                if folded is not None and folded[2] is not None:
                    batch_pool = folded[2]
                else:
                    batch_pool = self.reducer.reduce_batch(so, batch if batch is not None else so.batch)
                terms = fused[3] if self._loss_needs_raw else None
                diff = fused[-1] if diff_scales is not None else None
                loss = self._loss_from_fused(adj, so, mask, raw, terms, diff)
            else:
                x_pool, batch_pool = self.reduce(x=x, so=so, batch=batch)
                adj_pool, loss = self._batched_connect_and_loss(x, adj, so, mask, edge_weight, batch, batch_pool)
            if self.sparse_output:
                x_pool, ei, ew, batch_pool = self._finalize_sparse_output(
                    x_pool=x_pool, adj_pool=adj_pool, batch=batch, batch_pooled=batch_pool, so=so)
                return PoolingOutput(x=x_pool, edge_index=ei, edge_weight=ew, batch=batch_pool, so=so, loss=loss)
            return PoolingOutput(x=x_pool, edge_index=adj_pool, so=so, loss=loss)
        # unbatched: S [N,K], sparse A, sparse losses
        fused = self._unbatched_fused(x, adj, edge_weight, batch)
        if fused is not None and len(fused) == 5:  # the losses from the products the Connect forms anyway (r6)
            so, x_pool, adj_pool, batch_pool, loss = fused
            return PoolingOutput(x=x_pool, edge_index=adj_pool, edge_weight=None, batch=batch_pool, so=so, loss=loss)
        so = fused[0] if fused is not None else self.select(x=x, batch=batch)
        loss = self.compute_sparse_loss(adj, edge_weight, so.s, batch)
        x_pool, batch_pool = self.reduce(x=x, so=so, batch=batch, return_batched=not self.sparse_output)
        ei, ew = self.connect(edge_index=adj, so=so, edge_weight=edge_weight, batch=batch,
                              batch_pooled=batch_pool)
        return PoolingOutput(x=x_pool, edge_index=ei, edge_weight=ew, batch=batch_pool, so=so, loss=loss)


class DiffPool(_DenseMLPPooling):
    r"""DiffPool: S = softmax(MLP(X)), X' = S^T X, A' = S^T A S, link-prediction + entropy auxiliary
    losses (reference poolers/diffpool.py:22-331)."""

    def __init__(self, in_channels: Union[int, List[int]], k: int, act: str = None, dropout: float = 0.0,
                 link_loss_coeff: float = 1.0, ent_loss_coeff: float = 1.0, normalize_loss: bool = False,
                 remove_self_loops: bool = True, degree_norm: bool = True, edge_weight_norm: bool = False,
                 adj_transpose: bool = True, lift: str = "precomputed", s_inv_op: str = "transpose",
                 batched: bool = True, sparse_output: bool = False, cache_preprocessing: bool = False):
        super().__init__(in_channels, k, act, dropout, remove_self_loops, degree_norm, edge_weight_norm,
                         adj_transpose, lift, s_inv_op, batched, sparse_output, cache_preprocessing)
        self.link_loss_coeff = link_loss_coeff
        self.ent_loss_coeff = ent_loss_coeff
        self.normalize_loss = normalize_loss

    def _batched_connect_and_loss(self, x, adj, so, mask, edge_weight, batch, batch_pooled):
        adj_pool, _ = self.connect(edge_index=adj, so=so, edge_weight=edge_weight, batch=batch,
                                   batch_pooled=batch_pooled)
        loss = self.compute_loss(adj=adj, S=so.s, num_nodes=self._real_nodes(mask))
        return adj_pool, loss

    def _fused_diff_scales(self, adj, mask, adj_numel=None):
        num_nodes = self._real_nodes(mask)
        if not (isinstance(num_nodes, int) and num_nodes > 0):
            return None
        numel = adj.numel() if adj_numel is None else adj_numel
        link_scale = self.link_loss_coeff / numel if self.normalize_loss is True else self.link_loss_coeff
        return (float(link_scale), float(self.ent_loss_coeff) / num_nodes)

    def _loss_from_fused(self, adj, so, mask, raw, terms=None, diff=None) -> dict:
        if diff is not None:  # both losses came with the fused training call (and are differentiated by its backward)
            return {"link_loss": diff[0], "entropy_loss": diff[1]}
        return self.compute_loss(adj=adj, S=so.s, num_nodes=self._real_nodes(mask))

    def compute_loss(self, adj: Tensor, S: Tensor, num_nodes: int) -> dict:
        if (S.is_cuda and S.dim() == 3 and S.dtype == torch.float32 and adj.dim() == 3 and adj.is_contiguous()
                and isinstance(num_nodes, int) and num_nodes > 0
                and not (torch.is_grad_enabled() and (S.requires_grad or adj.requires_grad))):
            # inference: both losses from their native partial reductions and one tail launch (as torch ops behind
            # the kernels: sum, sqrt, a final-sum kernel, a division and two multiplications)
            from .. import kernels as K
            link_scale = self.link_loss_coeff / adj.numel() if self.normalize_loss is True else self.link_loss_coeff
            both = K.diffpool_loss_tail(S, adj, self._sizes_for(adj), link_scale, self.ent_loss_coeff / num_nodes)
            return {"link_loss": both[0], "entropy_loss": both[1]}
        return {"link_loss": link_pred_loss(S, adj, normalize_loss=self.normalize_loss,
                                            graph_sizes=self._sizes_for(adj)) * self.link_loss_coeff,
                "entropy_loss": entropy_loss(S, num_nodes) * self.ent_loss_coeff}

    def compute_sparse_loss(self, edge_index, edge_weight, S, batch) -> dict:
        ei, ew = connectivity_to_edge_index(edge_index, edge_weight)
        return {"link_loss": sparse_link_pred_loss(S, ei, ew, batch, normalize_loss=self.normalize_loss)
                * self.link_loss_coeff,
                "entropy_loss": unbatched_entropy_loss(S) * self.ent_loss_coeff}

    def extra_repr_args(self) -> dict:
        return {"batched": self.batched, "link_loss_coeff": self.link_loss_coeff,
                "ent_loss_coeff": self.ent_loss_coeff, "normalize_loss": self.normalize_loss}


class MinCutPooling(_DenseMLPPooling):
    r"""MinCut pooling: as DiffPool, with the normalised-cut and orthogonality losses computed on the
    *raw* S^T A S before it is post-processed (reference poolers/mincut.py:22-354)."""

    def __init__(self, in_channels: Union[int, List[int]], k: int, act: str = None, dropout: float = 0.0,
                 cut_loss_coeff: float = 1.0, ortho_loss_coeff: float = 1.0, remove_self_loops: bool = True,
                 degree_norm: bool = True, edge_weight_norm: bool = False, adj_transpose: bool = True,
                 lift: str = "precomputed", s_inv_op: str = "transpose", batched: bool = True,
                 sparse_output: bool = False, cache_preprocessing: bool = False):
        super().__init__(in_channels, k, act, dropout, remove_self_loops, degree_norm, edge_weight_norm,
                         adj_transpose, lift, s_inv_op, batched, sparse_output, cache_preprocessing)
        self.cut_loss_coeff = cut_loss_coeff
        self.ortho_loss_coeff = ortho_loss_coeff

    def _lift(self, x, so, batch, batch_pooled):
        return self.lift(x_pool=x, so=so, batch=batch if batch is not None else so.batch,
                         batch_pooled=batch_pooled)

    def _batched_connect_and_loss(self, x, adj, so, mask, edge_weight, batch, batch_pooled):
        c = self.connector
        raw = c.dense_connect(adj=adj, s=so.s)
        loss = self.compute_loss(adj, so.s, raw)
        adj_pool = postprocess_adj_pool_dense(raw, remove_self_loops=c.remove_self_loops,
                                              degree_norm=c.degree_norm, adj_transpose=c.adj_transpose,
                                              edge_weight_norm=c.edge_weight_norm)
        return adj_pool, loss

    _loss_needs_raw = True

    def _loss_from_fused(self, adj, so, mask, raw, terms=None, diff=None) -> dict:
        if terms is not None and so.s.dtype == torch.float32:
            # both per-graph loss tails came out of the pooling kernel itself (batches of small graphs); under autograd
            # their batch means arrive as two 0-dim outputs of the fused Function (functions.LossPair)
            both = terms if isinstance(terms, LossPair) else terms.mean(dim=1)
            return {"cut_loss": both[0] if self.cut_loss_coeff == 1 else both[0] * self.cut_loss_coeff,
                    "ortho_loss": both[1] if self.ortho_loss_coeff == 1 else both[1] * self.ortho_loss_coeff}
        return self.compute_loss(adj, so.s, raw)

    def compute_loss(self, adj: Tensor, S: Tensor, adj_pooled: Tensor) -> dict:
        if (S.is_cuda and S.dim() == 3 and S.dtype == torch.float32
                and not (torch.is_grad_enabled() and (S.requires_grad or adj.requires_grad
                                                       or adj_pooled.requires_grad))):
            # inference: both losses' per-graph tails in one launch (as torch ops: ~14 launches of a few hundred bytes)
            both = mincut_loss_terms(adj, S, adj_pooled, graph_sizes=self._sizes_for(adj)).mean(dim=1)
            return {"cut_loss": both[0] if self.cut_loss_coeff == 1 else both[0] * self.cut_loss_coeff,
                    "ortho_loss": both[1] if self.ortho_loss_coeff == 1 else both[1] * self.ortho_loss_coeff}
        if (S.is_cuda and S.dim() == 3 and S.dtype == torch.float32 and adj.dim() == 3 and not adj.requires_grad
                and adj_pooled.dtype == torch.float32 and adj_pooled.dim() == 3):
            # training: both losses as one Function with a native backward tail (the adjacency gets no gradient there)
            both = _MinCutTermsFn.apply(adj, S, adj_pooled, self._sizes_for(adj)).mean(dim=1)
            return {"cut_loss": both[0] if self.cut_loss_coeff == 1 else both[0] * self.cut_loss_coeff,
                    "ortho_loss": both[1] if self.ortho_loss_coeff == 1 else both[1] * self.ortho_loss_coeff}
        return {"cut_loss": mincut_loss(adj, S, adj_pooled, batch_reduction="mean",
                                        graph_sizes=self._sizes_for(adj)) * self.cut_loss_coeff,
                "ortho_loss": orthogonality_loss(S, batch_reduction="mean",
                                                 graph_sizes=self._sizes_for(adj)) * self.ortho_loss_coeff}

    def compute_sparse_loss(self, edge_index, edge_weight, S, batch) -> dict:
        ei, ew = connectivity_to_edge_index(edge_index, edge_weight)
        return {"cut_loss": sparse_mincut_loss(ei, S, ew, batch, batch_reduction="mean") * self.cut_loss_coeff,
                "ortho_loss": unbatched_orthogonality_loss(S, batch, batch_reduction="mean")
                * self.ortho_loss_coeff}

    def extra_repr_args(self) -> dict:
        return {"batched": self.batched, "cut_loss_coeff": self.cut_loss_coeff,
                "ortho_loss_coeff": self.ortho_loss_coeff}


# =============================================================================== factory
pooler_classes = ["DiffPool", "GraclusPooling", "MinCutPooling", "NDPPooling", "TopkPooling"]

pooler_map = {
    "diff": DiffPool,
    "graclus": GraclusPooling,
    "mincut": MinCutPooling,
    "ndp": NDPPooling,
    "topk": TopkPooling,
}


def _missing_required(cls, given: dict) -> List[str]:
    out = []
    for name, p in inspect.signature(cls.__init__).parameters.items():
        if name == "self" or p.kind in (p.VAR_POSITIONAL, p.VAR_KEYWORD):
            continue
        if p.default is p.empty and name not in given:
            out.append(name)
    return out


def get_pooler(pooler_name: str, **kwargs):
    """Build a pooler from its alias (reference poolers/__init__.py:91-147): case-insensitive, a ``_u``
    suffix selects the unbatched mode, kwargs the constructor does not take are dropped silently, a
    missing required argument raises ``TypeError``."""
    alias = pooler_name.lower()
    if alias.endswith("_u") and alias[:-2] in pooler_map:
        alias = alias[:-2]
        kwargs.setdefault("batched", False)
    if alias not in pooler_map:
        raise ValueError(f"Unknown pooler_name='{pooler_name.lower()}'. "
                         f"Available poolers: {list(pooler_map.keys())}")
    cls = pooler_map[alias]
    sig = cls.get_signature()
    init_kwargs = kwargs if sig.has_kwargs else {k: v for k, v in kwargs.items() if k in sig.args}
    missing = _missing_required(cls, init_kwargs)
    if missing:
        raise TypeError(f"Missing required argument(s) for pooler '{alias}' ({cls.__name__}): "
                        f"{', '.join(missing)}")
    return cls(**init_kwargs)


__all__ = ["get_pooler", "pooler_map", "pooler_classes"] + pooler_classes
