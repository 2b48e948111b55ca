"""Auxiliary losses DiffPool / MinCut compute between Reduce and Connect
(reference: tgp/utils/losses.py:39-123, 476-483, 644-708).

The batched dense losses run on native kernels: the link-prediction residual is
reduced inside the GEMM epilogue so S S^T [B,N,N] is never materialised, the entropy and the
trace(S^T D S) terms are single-pass reductions, S^T S runs on the matrix cores; each has a closed-form
backward.  The unbatched (sparse-adjacency) variants are differentiable torch ops over the edge list.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import Tensor

from .. import eps
from .. import functions as Fn
from .. import kernels as K
from .ops import check_and_filter_edge_weights, graph_ptr, max_graph_size, num_graphs_of


def _reduce(loss: Tensor, how: str) -> Tensor:
    if how == "mean":
        return loss.mean(dim=0)
    if how == "sum":
        return loss.sum(dim=0)
    raise ValueError(f"Batch reduction {how} not allowed, must be one of ['mean', 'sum'].")


def _seg_sum(src: Tensor, index: Tensor, size: int) -> Tensor:
    return src.new_zeros((size,) + tuple(src.shape[1:])).index_add_(0, index, src)


class _CutDenFn(torch.autograd.Function):
    """den[b] = trace(S^T D S) = sum_i deg_i ||S_i||^2 with deg = row sums of adj, one pass over adj."""

    @staticmethod
    def forward(ctx, adj, S, graph_sizes=None):
        deg, q, den = K.cut_terms(adj, S, graph_sizes)
        ctx.save_for_backward(S, deg, q)
        return den

    @staticmethod
    def backward(ctx, g):
        S, deg, q = ctx.saved_tensors
        g = g.view(-1, 1, 1)
        g_adj = g_s = None
        if ctx.needs_input_grad[0]:
            g_adj = (g * q.unsqueeze(-1)).expand(-1, -1, q.size(1)).contiguous()
        if ctx.needs_input_grad[1]:
            g_s = 2.0 * g * deg.unsqueeze(-1) * S
        return g_adj, g_s, None


class _GramFn(torch.autograd.Function):
    """G = S^T S on the matrix cores (split over the node dimension); dS = S (g + g^T)."""

    @staticmethod
    def forward(ctx, S, graph_sizes=None):
        ctx.save_for_backward(S)
        return K.dense_pool(S, None, S, graph_sizes=graph_sizes)[0]

    @staticmethod
    def backward(ctx, g):
        (S,) = ctx.saved_tensors
        return K.bmm(S, (g + g.transpose(-1, -2)).contiguous()), None


class _LinkNormFn(torch.autograd.Function):
    """||adj - S S^T||_F over the whole batch; S S^T only ever exists tile by tile in the MFMA
    accumulators (forward) and the backward uses d/dS = (4 S S^T S - 2 (A + A^T) S) / (2 norm)."""

    @staticmethod
    def forward(ctx, S, adj, graph_sizes=None):
        norm = torch.sqrt(K.link_loss_sq(S, adj, graph_sizes).sum())
        ctx.save_for_backward(S, adj, norm)
        ctx.products = Fn.shared_products(S, adj)  # A S, A^T S: shared with DenseConnect's backward
        return norm

    @staticmethod
    def backward(ctx, g):
        S, adj, norm = ctx.saved_tensors
        coef = torch.where(norm > 0, g / norm, torch.zeros_like(norm))  # torch.norm's subgradient at 0
        g_s = g_adj = None
        if ctx.needs_input_grad[0]:
            gram = K.dense_pool(S, None, S)[0]
            g_s = (2.0 * K.bmm(S, gram) - ctx.products.get_u(S, adj) - ctx.products.get_v(S, adj)) * coef
        if ctx.needs_input_grad[1]:
            g_adj = (adj - torch.matmul(S, S.transpose(1, 2))) * coef
        return g_s, g_adj, None


class _EntropySumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, S):
        ctx.save_for_backward(S)
        return K.entropy_sum(S)

    @staticmethod
    def backward(ctx, g):
        (S,) = ctx.saved_tensors
        if S.is_cuda and S.dtype == torch.float32 and g.numel() == 1:
            return K.entropy_bwd(S, g)  # one elementwise launch
        return -(torch.log(S + eps) + S / (S + eps)) * g


def mincut_loss(adj: Tensor, S: Tensor, adj_pooled: Tensor, batch_reduction: str = "mean",
                graph_sizes: Optional[Tensor] = None) -> Tensor:
    """``graph_sizes`` (this build only): real nodes per graph of a zero-padded batch, lets the pass over adj skip the
    padding."""
    num = torch.diagonal(adj_pooled, dim1=-2, dim2=-1).sum(-1)
    den = _CutDenFn.apply(adj, S, graph_sizes)  # trace(S^T D S) without forming D
    return _reduce(-(num / (den + eps)), batch_reduction)


class _MinCutTermsFn(torch.autograd.Function):
    """[2,B] per-graph values of :func:`mincut_loss` and :func:`orthogonality_loss` under autograd with a native backward
    tail: forward = the three kernels of :func:`mincut_loss_terms`; backward = one launch for the K x K parts
    (tgp_mincut_loss_terms_bwd_f32), one product S (W + W^T) and one fused multiply-add for 2 c1 D S.  The adjacency gets
    no gradient here (callers whose adj requires one keep the composed Functions above)."""

    @staticmethod
    def forward(ctx, adj, S, raw, graph_sizes):
        deg, _, den = K.cut_terms(adj, S, graph_sizes)
        gram = K.dense_pool(S, None, S, graph_sizes=graph_sizes)[0]
        ctx.save_for_backward(S, raw, deg, den, gram)
        return K.mincut_loss_terms(raw, den, gram)

    @staticmethod
    def backward(ctx, g):
        S, raw, deg, den, gram = ctx.saved_tensors
        g_raw, c1, W = K.mincut_loss_terms_bwd(raw, den, gram, g)
        g_s = K.bmm(S, W + W.transpose(1, 2))
        g_s.addcmul_((2.0 * c1).view(-1, 1, 1) * deg.unsqueeze(-1), S)
        return None, g_s.to(S.dtype), g_raw.to(raw.dtype), None


def mincut_loss_terms(adj: Tensor, S: Tensor, adj_pooled: Tensor, graph_sizes: Optional[Tensor] = None) -> Tensor:
    """[2,B] per-graph values of :func:`mincut_loss` and :func:`orthogonality_loss` (before the batch reduction) for
    a device batch outside autograd: den and S^T S from their kernels, both tails in ONE launch."""
    _, _, den = K.cut_terms(adj, S, graph_sizes)
    gram = K.dense_pool(S, None, S, graph_sizes=graph_sizes)[0]
    return K.mincut_loss_terms(adj_pooled, den, gram)


class _OrthoFromGramFn(torch.autograd.Function):
    """l = || G / ||G||_F - I / sqrt(K) ||_F per graph (losses.py:59-70) with its closed-form gradient
    dl/dG = (Y - <Y, G> G / n^2) / (n l),  Y = G / n - I / sqrt(K),  n = ||G||_F."""

    @staticmethod
    def forward(ctx, gram):
        k = gram.size(-1)
        n = torch.linalg.matrix_norm(gram, keepdim=True)
        y = gram / n
        y.diagonal(dim1=-2, dim2=-1).sub_(1.0 / math.sqrt(k))
        l = torch.linalg.matrix_norm(y)
        ctx.save_for_backward(gram, n, y, l)
        return l

    @staticmethod
    def backward(ctx, g):
        gram, n, y, l = ctx.saved_tensors
        yg = (y * gram).sum(dim=(-2, -1), keepdim=True)
        coef = (g / l).view(-1, 1, 1) / n
        return coef * (y - yg * gram / (n * n))


def orthogonality_loss(S: Tensor, batch_reduction: str = "mean", graph_sizes: Optional[Tensor] = None) -> Tensor:
    if S.dim() == 2:  # a single graph [N,K]: the reference's transpose(-2,-1) / norm(dim=(-2,-1)) accept it (losses.py:59-70)
        return orthogonality_loss(S.unsqueeze(0), batch_reduction, None).reshape(())
    sts = _GramFn.apply(S, graph_sizes if S.dim() == 3 else None)
    if sts.is_cuda and torch.is_grad_enabled() and sts.requires_grad:
        return _reduce(_OrthoFromGramFn.apply(sts), batch_reduction)
    sts = sts / torch.norm(sts, dim=(-2, -1), keepdim=True)
    k = S.size(-1)
    target = torch.eye(k, device=S.device, dtype=S.dtype) / math.sqrt(k)
    return _reduce(torch.norm(sts - target, dim=(-2, -1)), batch_reduction)


def link_pred_loss(S: Tensor, adj: Tensor, normalize_loss: bool = True,
                   graph_sizes: Optional[Tensor] = None) -> Tensor:
    loss = _LinkNormFn.apply(S, adj, graph_sizes)
    return loss / adj.numel() if normalize_loss is True else loss


def unbatched_entropy_loss(S: Tensor, num_nodes: Optional[int] = None) -> Tensor:
    if num_nodes is None:
        num_nodes = S.size(0)
    if S.is_cuda and S.dtype == torch.float32 and S.numel() > 0:
        return _EntropySumFn.apply(S) / num_nodes  # the same sum as the batched form: one launch (+ one in backward)
    return (-(S * torch.log(S + eps)).sum(dim=-1)).sum() / num_nodes


def entropy_loss(S: Tensor, num_nodes: int) -> Tensor:
    return _EntropySumFn.apply(S) / num_nodes


def _edge_weights(edge_index: Tensor, edge_weight: Optional[Tensor], like: Tensor) -> Tensor:
    if edge_weight is None:
        return torch.ones(edge_index.size(1), device=like.device, dtype=like.dtype)
    return check_and_filter_edge_weights(edge_weight).view(-1).to(like.dtype)


def _batch_or_zeros(batch: Optional[Tensor], n: int, device) -> Tensor:
    return torch.zeros(n, dtype=torch.long, device=device) if batch is None else batch


def sparse_mincut_loss(edge_index: Tensor, S: Tensor, edge_weight: Optional[Tensor] = None,
                       batch: Optional[Tensor] = None, batch_reduction: str = "mean") -> Tensor:
    n = S.size(0)
    w = _edge_weights(edge_index, edge_weight, S)
    nb = num_graphs_of(batch)
    batch = _batch_or_zeros(batch, n, S.device)
    deg = _seg_sum(w, edge_index[0], n)
    den = _seg_sum(deg * (S * S).sum(-1), batch, nb)
    contrib = w * Fn.edge_dot(S, edge_index)
    num = _seg_sum(contrib, batch[edge_index[0]], nb)
    return _reduce(-(num / (den + eps)), batch_reduction)


def _per_graph_gram(S: Tensor, batch: Optional[Tensor], nb: int) -> Tensor:
    """[B,K,K] stack of S_g^T S_g (batch is sorted, as everywhere in PyG-style batching): one launch."""
    if batch is None or nb == 1:
        return Fn.bmm(S, S, trans_a=True).unsqueeze(0)
    _, ptr = graph_ptr(batch, nb)
    return Fn.segment_gemm_tn(S, S, ptr, max_graph_size(batch))


def unbatched_orthogonality_loss(S: Tensor, batch: Optional[Tensor] = None,
                                 batch_reduction: str = "mean") -> Tensor:
    n, k = S.shape
    nb = num_graphs_of(batch)
    gram = _per_graph_gram(S, batch, nb)
    gram = gram / torch.norm(gram, dim=(-2, -1), keepdim=True)
    target = torch.eye(k, device=S.device, dtype=S.dtype) / math.sqrt(k)
    return _reduce(torch.norm(gram - target, dim=(-2, -1)), batch_reduction)


def sparse_link_pred_loss(S: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor] = None,
                          batch: Optional[Tensor] = None, normalize_loss: bool = True) -> Tensor:
    n = S.size(0)
    w = _edge_weights(edge_index, edge_weight, S)
    nb = num_graphs_of(batch)
    ss = Fn.edge_dot(S, edge_index)
    gram = _per_graph_gram(S, batch, nb)
    # ||A - S S^T||_F^2 = sum_E (w - ss)^2 + sum_g ||S_g^T S_g||_F^2 - sum_E ss^2
    sq = ((w - ss) ** 2).sum() + (gram * gram).sum() - (ss ** 2).sum()
    loss = torch.sqrt(torch.clamp(sq, min=0.0))
    if not normalize_loss:
        return loss
    if batch is None:
        return loss / (n * n) if n > 0 else loss
    sizes, _ = graph_ptr(batch, nb)
    return loss / (sizes * sizes).sum().clamp(min=1)  # stays on the device: no host round trip
