"""Auxiliary losses DiffPool / MinCut compute between Reduce and Connect
(reference: tgp/utils/losses.py:39-123, 476-483, 644-708).

These are training objectives, not part of the timed Reduce+Connect path; they are written with
differentiable torch ops on the device the inputs live on (SURVEY.md 8(f) N3 lists fusing them
into the GEMM epilogue as a later step).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import Tensor

from .. import eps
from .ops import check_and_filter_edge_weights


def _reduce(loss: Tensor, how: str) -> Tensor:
    if how == "mean":
        return loss.mean(dim=0)
    if how == "sum":
        return loss.sum(dim=0)
    raise ValueError(f"Batch reduction {how} not allowed, must be one of ['mean', 'sum'].")


def _seg_sum(src: Tensor, index: Tensor, size: int) -> Tensor:
    return src.new_zeros((size,) + tuple(src.shape[1:])).index_add_(0, index, src)


def mincut_loss(adj: Tensor, S: Tensor, adj_pooled: Tensor, batch_reduction: str = "mean") -> Tensor:
    num = torch.diagonal(adj_pooled, dim1=-2, dim2=-1).sum(-1)
    deg = adj.sum(-1)  # [B,N]
    den = (deg.unsqueeze(-1) * S * S).sum(dim=(-2, -1))  # trace(S^T D S) without forming D
    return _reduce(-(num / (den + eps)), batch_reduction)


def orthogonality_loss(S: Tensor, batch_reduction: str = "mean") -> Tensor:
    sts = torch.matmul(S.transpose(-2, -1), S)
    sts = sts / torch.norm(sts, dim=(-2, -1), keepdim=True)
    k = S.size(-1)
    target = torch.eye(k, device=S.device, dtype=S.dtype) / math.sqrt(k)
    return _reduce(torch.norm(sts - target, dim=(-2, -1)), batch_reduction)


def link_pred_loss(S: Tensor, adj: Tensor, normalize_loss: bool = True) -> Tensor:
    loss = torch.norm(adj - torch.matmul(S, S.transpose(1, 2)), p=2)
    return loss / adj.numel() if normalize_loss is True else loss


def unbatched_entropy_loss(S: Tensor, num_nodes: Optional[int] = None) -> Tensor:
    if num_nodes is None:
        num_nodes = S.size(0)
    return (-(S * torch.log(S + eps)).sum(dim=-1)).sum() / num_nodes


def entropy_loss(S: Tensor, num_nodes: int) -> Tensor:
    return unbatched_entropy_loss(S.reshape(-1, S.size(-1)), num_nodes)


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
    batch = _batch_or_zeros(batch, n, S.device)
    nb = int(batch.max()) + 1
    deg = _seg_sum(w, edge_index[0], n)
    den = _seg_sum(deg * (S * S).sum(-1), batch, nb)
    contrib = w * (S[edge_index[0]] * S[edge_index[1]]).sum(-1)
    num = _seg_sum(contrib, batch[edge_index[0]], nb)
    return _reduce(-(num / (den + eps)), batch_reduction)


def _per_graph_gram(S: Tensor, batch: Tensor, nb: int) -> Tensor:
    """[B,K,K] stack of S_g^T S_g (batch is sorted, as everywhere in PyG-style batching)."""
    sizes = torch.bincount(batch, minlength=nb).tolist()
    return torch.stack([p.t().matmul(p) for p in S.split(sizes)])


def unbatched_orthogonality_loss(S: Tensor, batch: Optional[Tensor] = None,
                                 batch_reduction: str = "mean") -> Tensor:
    n, k = S.shape
    batch = _batch_or_zeros(batch, n, S.device)
    nb = int(batch.max()) + 1
    gram = _per_graph_gram(S, batch, nb)
    gram = gram / torch.norm(gram, dim=(-2, -1), keepdim=True)
    target = torch.eye(k, device=S.device, dtype=S.dtype) / math.sqrt(k)
    return _reduce(torch.norm(gram - target, dim=(-2, -1)), batch_reduction)


def sparse_link_pred_loss(S: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor] = None,
                          batch: Optional[Tensor] = None, normalize_loss: bool = True) -> Tensor:
    n = S.size(0)
    w = _edge_weights(edge_index, edge_weight, S)
    batch = _batch_or_zeros(batch, n, S.device)
    nb = int(batch.max()) + 1
    ss = (S[edge_index[0]] * S[edge_index[1]]).sum(-1)
    gram = _per_graph_gram(S, batch, nb)
    # ||A - S S^T||_F^2 = sum_E (w - ss)^2 + sum_g ||S_g^T S_g||_F^2 - sum_E ss^2
    sq = ((w - ss) ** 2).sum() + (gram * gram).sum() - (ss ** 2).sum()
    loss = torch.sqrt(torch.clamp(sq, min=0.0))
    sizes = torch.bincount(batch, minlength=nb)
    numel = int((sizes * sizes).sum())
    return loss / numel if normalize_loss and numel > 0 else loss
