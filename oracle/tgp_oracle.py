"""CPU oracle for the SRC pooling hot path (Select -> Reduce -> Connect).

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import this file;
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` do, and there only as the checker / the reported CPU baseline.

It is a plain-torch (CPU, fp32 / int64) re-statement of the reference algorithm
(tgp 1.0.1), one function per row of SURVEY.md section 8(a); every function cites
the reference file:line it follows.  The third-party leaves the reference
delegates to (torch_geometric 2.6 ``scatter/coalesce/subgraph/topk/softmax/
to_dense_adj/to_dense_batch/unbatch``, torch_scatter 2.1.2 ``scatter``) are absent
from /root/reference and from this image; their published algorithms are
re-stated inline.

Parity pin: the oracle is checked in ``tests/test_oracle_golden.py`` against
(i) every literal known-answer the reference's own tests hold for this path
 and (ii) ``tests/golden/golden_v1.pt`` —
outputs of the reference's own orchestration code run in the build container over
a build-authored PyG stand-in (tests/golden/make_golden.py).  The ordering of TopK
``x_pool`` rows vs relabelled ``edge_index``, ``coalesce``/``subgraph`` output order
and the Kron values are unpinned by the reference's tests; for those the
golden dump is the only authority.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

EPS = 1e-8  # tgp/__init__.py:6


# =============================================================================
# generic leaves (PyG / torch_scatter published semantics)
# =============================================================================
def seg_sum(src: Tensor, index: Tensor, size: int) -> Tensor:
    """scatter(..., reduce='sum') == zeros.index_add_ (sequential on CPU)."""
    out = src.new_zeros((size,) + tuple(src.shape[1:]))
    return out.index_add_(0, index, src)


def seg_reduce(src: Tensor, index: Tensor, size: int, op: str) -> Tensor:
    if op in ("sum", "add"):
        return seg_sum(src, index, size)
    if op == "mean":
        cnt = seg_sum(torch.ones_like(src), index, size).clamp(min=1)
        return seg_sum(src, index, size) / cnt
    if op in ("max", "min"):
        out = src.new_zeros(size)
        return out.scatter_reduce_(0, index, src, reduce="a" + op, include_self=False)
    if op == "mul":
        out = src.new_ones(size)
        return out.scatter_reduce_(0, index, src, reduce="prod", include_self=True)
    raise ValueError(op)


def counts_to_ptr(counts: Tensor) -> Tensor:
    return torch.cat([counts.new_zeros(1), counts.cumsum(0)])


def graph_sizes(batch: Tensor, num_graphs: Optional[int] = None) -> Tensor:
    if num_graphs is None:
        num_graphs = int(batch.max()) + 1 if batch.numel() else 0
    return torch.bincount(batch, minlength=num_graphs)


# =============================================================================
# A0  SelectOutput construction  (select/base_select.py:19-71)
# =============================================================================
def sort_assignment(node_index: Tensor, cluster_index: Tensor,
                    weight: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor]:
    """base_select.py:58-65 — assignments are stored sorted by node id; a missing
    weight becomes ones."""
    node_sorted, perm = torch.sort(node_index)
    w = weight[perm] if weight is not None else torch.ones(node_index.numel())
    return node_sorted, cluster_index[perm], w


# =============================================================================
# A12  TopkSelect  (select/topk_select.py:163-203 + PyG topk)
# =============================================================================
def topk_perm(score: Tensor, ratio, batch: Tensor, min_score=None, tol=1e-7) -> Tensor:
    nb = int(batch.max()) + 1 if batch.numel() else 0
    if min_score is not None:
        smax = seg_reduce(score, batch, nb, "max")[batch] - tol
        smin = smax.clamp(max=min_score)
        return (score > smin).nonzero().view(-1)
    n_per = graph_sizes(batch, nb)
    if ratio >= 1:
        k = torch.minimum(torch.full_like(n_per, int(ratio)), n_per)
    else:
        k = (float(ratio) * n_per.to(score.dtype)).ceil().to(torch.long)
    _, order = torch.sort(score.view(-1), descending=True)
    b_sorted, b_perm = torch.sort(batch[order], stable=True)
    ptr = counts_to_ptr(n_per)
    rank_in_graph = torch.arange(score.numel()) - ptr[b_sorted]
    keep = rank_in_graph < k[b_sorted]
    return order[b_perm[keep]]


def topk_select(x: Tensor, p: Optional[Tensor], ratio, batch: Optional[Tensor],
                min_score=None, act: str = "tanh"):
    """Returns (node_index, cluster_index, weight) already in stored (node-sorted) order."""
    if batch is None:
        batch = torch.zeros(x.size(0), dtype=torch.long)
    if p is None:
        score = x.view(-1)
    else:
        xx = x.view(-1, 1) if x.dim() == 1 else x
        score = (xx * p).sum(-1)
        if min_score is None:
            score = score / p.norm(p=2, dim=-1)
    if min_score is None:
        score = torch.tanh(score) if act == "tanh" else score
    else:
        nb = int(batch.max()) + 1
        smax = seg_reduce(score, batch, nb, "max")
        e = (score - smax[batch]).exp()
        score = e / (seg_sum(e, batch, nb) + 1e-16)[batch]
    sel = topk_perm(score, ratio, batch, min_score)
    return sort_assignment(sel, torch.arange(sel.numel()), score[sel])


# =============================================================================
# A13  MLPSelect  (select/mlp_select.py:105-147)
# =============================================================================
def mlp_select(x: Tensor, weights: Sequence[Tensor], biases: Sequence[Tensor],
               mask: Optional[Tensor] = None, act: Optional[str] = None) -> Tensor:
    h = x
    for i, (w, b) in enumerate(zip(weights, biases)):
        h = torch.nn.functional.linear(h, w, b)
        if i != len(weights) - 1 and act is not None:
            h = getattr(torch, act)(h)
    s = torch.softmax(h, dim=-1)
    if mask is not None:
        s = s * mask.unsqueeze(-1)
    return s


# =============================================================================
# A1 / A2  sparse Reduce  (reduce/base_reduce.py:14-53, 141-155)
# =============================================================================
def reduce_sparse(x: Tensor, node_index: Tensor, cluster_index: Tensor,
                  weight: Tensor, num_supernodes: int) -> Tensor:
    src = x[node_index] * weight.view(-1, 1)
    return seg_sum(src, cluster_index, num_supernodes)


def reduce_batch_sparse(batch: Optional[Tensor], node_index: Tensor,
                        cluster_index: Tensor, num_supernodes: int) -> Optional[Tensor]:
    if batch is None:
        return None
    out = torch.arange(num_supernodes)
    return out.scatter_(0, cluster_index, batch[node_index])


def reduce_batch_dense(batch: Optional[Tensor], num_supernodes: int) -> Optional[Tensor]:
    """base_reduce.py:43-53 + utils/ops.py:152-169."""
    if batch is None:
        return None
    if batch.numel() == 0:
        return batch.new_empty((0,))
    nb = int(batch.max()) + 1
    return torch.arange(nb, dtype=batch.dtype).repeat_interleave(num_supernodes)


# =============================================================================
# A3 / A3'  dense Reduce  (reduce/base_reduce.py:158-190)
# =============================================================================
def is_multi_graph(batch: Optional[Tensor]) -> bool:
    return batch is not None and batch.numel() > 0 and int(batch.min()) != int(batch.max())


def reduce_dense(s: Tensor, x: Tensor, batch: Optional[Tensor] = None,
                 return_batched: bool = False) -> Tensor:
    if s.dim() == 3:
        return s.transpose(-2, -1).matmul(x)
    if is_multi_graph(batch):
        sizes = graph_sizes(batch).tolist()
        parts = [si.t().matmul(xi) for si, xi in zip(s.split(sizes), x.split(sizes))]
        return torch.stack(parts) if return_batched else torch.cat(parts)
    out = s.t().matmul(x)
    return out.unsqueeze(0) if return_batched else out


# =============================================================================
# A5  TopK Connect: induced subgraph + relabel  (connect/base_conn.py:79-82)
# =============================================================================
def subgraph_connect(edge_index: Tensor, edge_weight: Optional[Tensor],
                     node_index: Tensor, num_nodes: int):
    keep_node = torch.zeros(num_nodes, dtype=torch.bool)
    keep_node[node_index] = True
    keep = keep_node[edge_index[0]] & keep_node[edge_index[1]]
    pos = torch.full((num_nodes,), -1, dtype=torch.long)
    pos[node_index] = torch.arange(node_index.numel())
    ei = pos[edge_index[:, keep]]
    ew = edge_weight[keep] if edge_weight is not None else None
    return ei, ew


# =============================================================================
# A4  one-over-K Connect: relabel + coalesce  (connect/base_conn.py:83-89)
# =============================================================================
def coalesce_connect(edge_index: Tensor, edge_weight: Optional[Tensor],
                     cluster_index: Tensor, num_supernodes: int, reduce_op: str = "sum"):
    ei = cluster_index[edge_index]
    key = ei[0] * num_supernodes + ei[1]
    key_sorted, perm = torch.sort(key, stable=True)
    ei = ei[:, perm]
    first = torch.ones(key.numel(), dtype=torch.bool)
    first[1:] = key_sorted[1:] > key_sorted[:-1]
    if edge_weight is None:
        return ei[:, first], None
    ew = edge_weight[perm]
    if bool(first.all()):
        return ei, ew
    seg = first.cumsum(0) - 1
    return ei[:, first], seg_reduce(ew, seg, int(first.sum()), reduce_op)


# =============================================================================
# A6  sparse post-processing  (utils/ops.py:338-419)
# =============================================================================
def postprocess_sparse(edge_index: Tensor, edge_weight: Optional[Tensor], num_nodes: int,
                       remove_self_loops: bool = False, degree_norm: bool = False,
                       edge_weight_norm: bool = False, batch_pooled: Optional[Tensor] = None):
    if remove_self_loops:
        keep = edge_index[0] != edge_index[1]
        edge_index = edge_index[:, keep]
        edge_weight = edge_weight[keep] if edge_weight is not None else None
    if edge_weight is not None:
        edge_weight = edge_weight.view(-1)
        if edge_weight.numel() > 0:
            keep = edge_weight.abs() > EPS
            edge_index, edge_weight = edge_index[:, keep], edge_weight[keep]
    if degree_norm:
        if edge_weight is None:
            edge_weight = torch.ones(edge_index.size(1))
        deg = seg_sum(edge_weight, edge_index[0], num_nodes).clamp(min=EPS)
        dis = deg.pow(-0.5)
        edge_weight = edge_weight * dis[edge_index[0]] * dis[edge_index[1]]
    if edge_weight_norm and edge_weight is not None:
        eb = batch_pooled[edge_index[0]]
        nb = int(eb.max()) + 1 if eb.numel() else 0
        mx = seg_reduce(edge_weight.abs(), eb, nb, "max")
        mx = torch.where(mx == 0, torch.ones_like(mx), mx)
        edge_weight = edge_weight / mx[eb]
    return edge_index, edge_weight


def sparse_connect(edge_index: Tensor, edge_weight: Optional[Tensor], node_index: Tensor,
                   cluster_index: Tensor, num_nodes: int, num_supernodes: int,
                   remove_self_loops: bool = True, reduce_op: str = "sum",
                   edge_weight_norm: bool = False, batch_pooled: Optional[Tensor] = None,
                   degree_norm: bool = False):
    """connect/base_conn.py:57-112 on a [2,E] edge list."""
    if node_index is not None and node_index.numel() < num_nodes:
        ei, ew = subgraph_connect(edge_index, edge_weight, node_index, num_nodes)
    elif cluster_index is not None and cluster_index.numel() == num_nodes:
        ei, ew = coalesce_connect(edge_index, edge_weight, cluster_index, num_supernodes, reduce_op)
    else:
        raise RuntimeError
    return postprocess_sparse(ei, ew, num_supernodes, remove_self_loops, degree_norm,
                              edge_weight_norm, batch_pooled)


# =============================================================================
# A7 / A7' / A8  dense Connect  (connect/dense_conn.py:111-208, utils/ops.py:282-335)
# =============================================================================
def dense_connect(s: Tensor, adj: Tensor) -> Tensor:
    return torch.matmul(torch.matmul(s.transpose(-2, -1), adj), s)


def postprocess_dense(adj_pool: Tensor, remove_self_loops: bool = False,
                      degree_norm: bool = False, adj_transpose: bool = False,
                      edge_weight_norm: bool = False) -> Tensor:
    adj_pool = adj_pool.clone()
    if remove_self_loops:
        torch.diagonal(adj_pool, dim1=-2, dim2=-1)[:] = 0
    if degree_norm:
        d = adj_pool.sum(-2 if adj_transpose else -1, keepdim=True)
        # adj_transpose only selects the summed axis; nothing is transposed here
        d = torch.sqrt(d.clamp(min=EPS))
        adj_pool = (adj_pool / d) / d.transpose(-2, -1)
    if edge_weight_norm:
        b = adj_pool.size(0)
        mx = adj_pool.reshape(b, -1).abs().max(dim=1, keepdim=True)[0].unsqueeze(-1)
        mx = torch.where(mx == 0, torch.ones_like(mx), mx)
        adj_pool = adj_pool / mx
    return adj_pool


def dense_connect_unbatched(edge_index: Tensor, edge_weight: Optional[Tensor],
                            batch: Optional[Tensor], s: Tensor) -> Tensor:
    """dense_conn.py:140-208 — sparse A, dense S [N,K] -> [B,K,K]."""
    n, k = s.shape
    nb = 1 if batch is None else int(batch.max()) + 1
    e = edge_index.size(1)
    w = torch.ones(e) if edge_weight is None else edge_weight.view(-1)
    if nb == 1:
        if e == 0:
            return s.new_zeros((1, k, k))
        a = torch.sparse_coo_tensor(edge_index, w, (n, n)).coalesce()
        return s.t().matmul(torch.sparse.mm(a, s)).unsqueeze(0)
    sizes = graph_sizes(batch, nb)
    ptr = counts_to_ptr(sizes)
    s_parts = s.split(sizes.tolist())
    if e == 0:
        return torch.stack([sp.new_zeros((k, k)) for sp in s_parts])
    eb = batch[edge_index[0]]
    local = edge_index - ptr[eb]
    e_sizes = torch.bincount(eb, minlength=nb).tolist()
    outs = []
    for ei_g, w_g, s_g in zip(local.split(e_sizes, dim=1), w.split(e_sizes), s_parts):
        n_g = s_g.size(0)
        a = torch.sparse_coo_tensor(ei_g, w_g, (n_g, n_g)).coalesce()
        outs.append(s_g.t().matmul(torch.sparse.mm(a, s_g)))
    return torch.stack(outs)


# =============================================================================
# A10  block-diagonal sparse output  (utils/ops.py:53-82, src.py:500-557)
# =============================================================================
def dense_to_block_diag(adj_pool: Tensor):
    if adj_pool.dim() == 2:
        adj_pool = adj_pool.unsqueeze(0)
    k = adj_pool.size(1)
    b, r, c = (adj_pool.abs() > EPS).nonzero(as_tuple=True)
    return torch.stack([r + b * k, c + b * k]), adj_pool[b, r, c]


def out_mask_dense(s: Tensor, batch: Optional[Tensor] = None) -> Tensor:
    """utils/ops.py:85-132."""
    if s.dim() == 3:
        return s.sum(-2) > 0
    if batch is None:
        return (s.sum(-2) > 0).unsqueeze(0)
    nb = int(batch.max()) + 1
    m = torch.zeros(nb, s.size(-1), dtype=torch.bool)
    for g in range(nb):
        sel = batch == g
        if sel.any():
            m[g] = s[sel].sum(0) > 0
    return m


def finalize_sparse_output(x_pool: Tensor, adj_pool: Tensor, batch: Optional[Tensor],
                           batch_pooled: Optional[Tensor], s: Tensor, so_batch=None):
    """src.py:500-557 (out_mask is never None for dense S)."""
    b, k = adj_pool.size(0), adj_pool.size(1)
    x_flat = x_pool.reshape(-1, x_pool.size(-1))
    om = out_mask_dense(s, so_batch)
    if batch_pooled is None and batch is not None:
        batch_pooled = reduce_batch_dense(batch, k)
    if batch_pooled is None and b > 1:
        batch_pooled = torch.arange(b).repeat_interleave(k)
    if batch_pooled is None:
        batch_pooled = torch.zeros(b * k, dtype=torch.long)
    valid = om.reshape(-1)
    vidx = valid.nonzero().view(-1)
    fm = om.to(adj_pool.dtype)
    ei, ew = dense_to_block_diag(adj_pool * fm.unsqueeze(-1) * fm.unsqueeze(-2))
    remap = torch.full((b * k,), -1, dtype=torch.long)
    remap[vidx] = torch.arange(vidx.numel())
    keep = (remap[ei[0]] >= 0) & (remap[ei[1]] >= 0)
    ei = torch.stack([remap[ei[0][keep]], remap[ei[1][keep]]])
    return x_flat[vidx], ei, ew[keep], batch_pooled[valid]


# =============================================================================
# A11  sparse -> padded dense  (src.py:374-452 + PyG to_dense_adj/to_dense_batch)
# =============================================================================
def to_dense_batch(x: Tensor, batch: Optional[Tensor]):
    if batch is None:
        return x.unsqueeze(0), torch.ones(1, x.size(0), dtype=torch.bool)
    nb = int(batch.max()) + 1
    sizes = graph_sizes(batch, nb)
    ptr = counts_to_ptr(sizes)
    nmax = int(sizes.max())
    slot = torch.arange(batch.numel()) - ptr[batch] + batch * nmax
    out = x.new_zeros((nb * nmax,) + tuple(x.shape[1:]))
    out[slot] = x
    mask = torch.zeros(nb * nmax, dtype=torch.bool)
    mask[slot] = True
    return out.view((nb, nmax) + tuple(x.shape[1:])), mask.view(nb, nmax)


def to_dense_adj(edge_index: Tensor, edge_weight: Optional[Tensor], batch: Optional[Tensor]):
    if batch is None:
        n = int(edge_index.max()) + 1 if edge_index.numel() else 0
        batch = torch.zeros(n, dtype=torch.long)
    nb = int(batch.max()) + 1 if batch.numel() else 1
    sizes = graph_sizes(batch, nb)
    ptr = counts_to_ptr(sizes)
    nmax = int(sizes.max())
    g = batch[edge_index[0]]
    r = edge_index[0] - ptr[batch][edge_index[0]]
    c = edge_index[1] - ptr[batch][edge_index[1]]
    w = torch.ones(g.numel()) if edge_weight is None else edge_weight
    flat = seg_sum(w, g * nmax * nmax + r * nmax + c, nb * nmax * nmax)
    return flat.view(nb, nmax, nmax)


def dense_preprocessing(x, edge_index, edge_weight, batch, adj_transpose: bool):
    adj = to_dense_adj(edge_index, edge_weight, batch)
    if adj_transpose:
        adj = adj.transpose(-1, -2)
    xd, mask = to_dense_batch(x, batch)
    return xd, adj, mask


# =============================================================================
# A9  Kron reduction  (connect/kron_conn.py:46-165) — host scipy, like the reference
# =============================================================================
def laplacian_scipy(edge_index: Tensor, edge_weight: Optional[Tensor], num_nodes: int):
    import scipy.sparse as sp
    keep = edge_index[0] != edge_index[1]
    ei = edge_index[:, keep]
    w = torch.ones(ei.size(1)) if edge_weight is None else edge_weight[keep]
    deg = seg_sum(w, ei[0], num_nodes)
    loops = torch.arange(num_nodes)
    rows = torch.cat([ei[0], loops]).numpy()
    cols = torch.cat([ei[1], loops]).numpy()
    vals = torch.cat([-w, deg]).numpy()
    return sp.coo_matrix((vals, (rows, cols)), (num_nodes, num_nodes)).tocsr()


def kron_connect(L, idx_pos: Tensor, sparse_threshold: float = 1e-2):
    import scipy.sparse as sp
    import scipy.sparse.linalg  # noqa: F401
    n = L.shape[0]
    all_nodes = torch.arange(n)
    idx_neg = all_nodes[~torch.isin(all_nodes, idx_pos)]
    if idx_pos.numel() <= 1:
        lnew = sp.csc_matrix(-np.ones((1, 1)))
    else:
        ip, ineg = idx_pos.numpy(), idx_neg.numpy()
        l_red = L[np.ix_(ip, ip)]
        l_io = L[np.ix_(ip, ineg)]
        l_oi = L[np.ix_(ineg, ip)].tocsc()
        l_c = L[np.ix_(ineg, ineg)].tocsc()
        try:
            lnew = l_red - l_io.dot(sp.linalg.spsolve(l_c, l_oi))
        except RuntimeError:
            damp = sp.csc_matrix(sp.eye(l_c.shape[0]) * 1e-6)
            lnew = l_red - l_io.dot(sp.linalg.spsolve(damp + l_c, l_oi))
        if np.abs(lnew - lnew.T).sum() < np.spacing(1) * np.abs(lnew).sum():
            lnew = (lnew + lnew.T) / 2.0
    a = -lnew
    if sparse_threshold > 0:
        a = a.multiply(np.abs(a) > sparse_threshold)
    a = sp.lil_matrix(a) if not sp.issparse(a) else a.tolil()
    a.setdiag(0)
    a = a.tocsr()
    a.eliminate_zeros()
    a = a.astype(np.float32).tocoo()
    ei = torch.stack([torch.from_numpy(a.row).long(), torch.from_numpy(a.col).long()])
    return ei, torch.from_numpy(a.data)


# =============================================================================
# auxiliary losses computed between Reduce and Connect  (utils/losses.py)
# =============================================================================
def mincut_loss(adj, s, adj_pooled):  # losses.py:39-56
    num = torch.einsum("ijj->i", adj_pooled)
    d = adj.sum(-1)
    den = torch.einsum("bnk,bn,bnk->b", s, d, s)
    return (-(num / (den + EPS))).mean()


def orthogonality_loss(s):  # losses.py:59-70
    sts = torch.matmul(s.transpose(-2, -1), s)
    sts = sts / torch.norm(sts, dim=(-2, -1), keepdim=True)
    k = s.size(-1)
    eye = torch.eye(k) / math.sqrt(k)
    return torch.norm(sts - eye, dim=(-2, -1)).mean()


def link_pred_loss(s, adj, normalize_loss=False):  # losses.py:644-652
    loss = torch.norm(adj - torch.matmul(s, s.transpose(1, 2)), p=2)
    return loss / adj.numel() if normalize_loss else loss


def entropy_loss(s, num_nodes):  # losses.py:476-483, 655-658
    s2 = s.reshape(-1, s.size(-1))
    return (-(s2 * torch.log(s2 + EPS)).sum(-1)).sum() / num_nodes


def sparse_mincut_loss(edge_index, s, edge_weight, batch):  # losses.py:73-127
    n = s.size(0)
    w = torch.ones(edge_index.size(1)) if edge_weight is None else edge_weight.view(-1)
    if batch is None:
        batch = torch.zeros(n, dtype=torch.long)
    nb = int(batch.max()) + 1
    deg = seg_sum(w, edge_index[0], n)
    den = seg_sum(deg * (s * s).sum(-1), batch, nb)
    contrib = w * (s[edge_index[0]] * s[edge_index[1]]).sum(-1)
    num = seg_sum(contrib, batch[edge_index[0]], nb)
    return (-(num / (den + EPS))).mean()


def unbatched_orthogonality_loss(s, batch):  # losses.py:204-240
    n, k = s.shape
    if batch is None:
        batch = torch.zeros(n, dtype=torch.long)
    eye = torch.eye(k) / math.sqrt(k)
    vals = []
    for g in range(int(batch.max()) + 1):
        sg = s[batch == g]
        sts = sg.t().matmul(sg)
        vals.append(torch.norm(sts / torch.norm(sts) - eye))
    return torch.stack(vals).mean()


def sparse_link_pred_loss(s, edge_index, edge_weight, batch, normalize_loss=False):  # losses.py:661-708
    n = s.size(0)
    w = torch.ones(edge_index.size(1)) if edge_weight is None else edge_weight.view(-1)
    if batch is None:
        batch = torch.zeros(n, dtype=torch.long)
    ss = (s[edge_index[0]] * s[edge_index[1]]).sum(-1)
    resid = ((w - ss) ** 2).sum()
    ss_sq = (ss ** 2).sum()
    total = torch.tensor(0.0)
    numel = 0
    for g in range(int(batch.max()) + 1):
        sg = s[batch == g]
        sts = sg.t().matmul(sg)
        total = total + (sts * sts).sum()
        numel += sg.size(0) ** 2
    loss = torch.sqrt(torch.clamp(resid + total - ss_sq, min=0.0))
    return loss / numel if normalize_loss and numel > 0 else loss


# =============================================================================
# A16  whole-pooler forwards (poolers/{topk,graclus,ndp,diffpool,mincut}.py)
# =============================================================================
def topk_pool(x, edge_index, edge_weight, batch, p, ratio=0.5, min_score=None,
              multiplier=1.0, act="tanh", remove_self_loops=True, degree_norm=False,
              edge_weight_norm=False, reduce_op="sum"):
    """poolers/topk.py:120-190."""
    ni, ci, w = topk_select(x, p, ratio, batch, min_score, act)
    k = ni.numel()
    xp = reduce_sparse(x, ni, ci, w, k)
    bp = reduce_batch_sparse(batch, ni, ci, k)
    if multiplier != 1:
        xp = multiplier * xp
    ei, ew = sparse_connect(edge_index, edge_weight, ni, ci, x.size(0), k, remove_self_loops,
                            reduce_op, edge_weight_norm, bp, degree_norm)
    return dict(x=xp, edge_index=ei, edge_weight=ew, batch=bp,
                node_index=ni, cluster_index=ci, weight=w)


def cluster_pool(x, edge_index, edge_weight, batch, cluster_index, num_supernodes,
                 weight=None, reduce_op="sum", remove_self_loops=True, degree_norm=False,
                 edge_weight_norm=False):
    """poolers/graclus.py:91-156 given the cluster assignment (selector output)."""
    n = cluster_index.numel()
    ni, ci, w = sort_assignment(torch.arange(n), cluster_index, weight)
    xp = reduce_sparse(x, ni, ci, w, num_supernodes) if x is not None else None
    bp = reduce_batch_sparse(batch, ni, ci, num_supernodes)
    ei, ew = sparse_connect(edge_index, edge_weight, ni, ci, n, num_supernodes, remove_self_loops,
                            reduce_op, edge_weight_norm, bp, degree_norm)
    return dict(x=xp, edge_index=ei, edge_weight=ew, batch=bp)


def dense_pool(alias, x, adj, edge_weight, batch, weights, biases, act=None, mask=None,
               remove_self_loops=True, degree_norm=True, edge_weight_norm=False,
               adj_transpose=True, batched=True, sparse_output=False, normalize_loss=False):
    """poolers/diffpool.py:145-260 and poolers/mincut.py:150-289 (eval mode, coeffs = 1)."""
    dense_in = adj.dim() == 3 or (adj.dim() == 2 and adj.size(0) == adj.size(1)
                                  and adj.is_floating_point())
    if batched:
        if dense_in:
            xd = x.unsqueeze(0) if x.dim() == 2 else x
            a = adj
            if mask is None:
                mask = torch.ones(xd.size(0), xd.size(1), dtype=torch.bool)
        else:
            xd, a, mask = dense_preprocessing(x, adj, edge_weight, batch, adj_transpose)
        s = mlp_select(xd, weights, biases, mask, act)
        xp = reduce_dense(s, xd)
        bp = reduce_batch_dense(batch, s.size(-1))
        raw = dense_connect(s, a)
        ap = postprocess_dense(raw, remove_self_loops, degree_norm, adj_transpose, edge_weight_norm)
        if alias == "diff":
            loss = dict(link_loss=link_pred_loss(s, a, normalize_loss),
                        entropy_loss=entropy_loss(s, int(mask.sum())))
        else:
            loss = dict(cut_loss=mincut_loss(a, s, raw), ortho_loss=orthogonality_loss(s))
        out = dict(s=s, loss=loss, mask=out_mask_dense(s))
        if sparse_output:
            xo, ei, ew, bo = finalize_sparse_output(xp, ap, batch, bp, s)
            out.update(x=xo, edge_index=ei, edge_weight=ew, batch=bo)
        else:
            out.update(x=xp, edge_index=ap, edge_weight=None, batch=None)
        return out
    # unbatched: S [N,K], sparse A
    s = mlp_select(x, weights, biases, None, act)
    if alias == "diff":
        loss = dict(link_loss=sparse_link_pred_loss(s, adj, edge_weight, batch, normalize_loss),
                    entropy_loss=entropy_loss(s, s.size(0)))
    else:
        loss = dict(cut_loss=sparse_mincut_loss(adj, s, edge_weight, batch),
                    ortho_loss=unbatched_orthogonality_loss(s, batch))
    xp = reduce_dense(s, x, batch, return_batched=not sparse_output)
    bp = reduce_batch_dense(batch, s.size(-1))
    raw = dense_connect_unbatched(adj, edge_weight, batch, s)
    out = dict(s=s, loss=loss, mask=out_mask_dense(s, batch), x=xp, batch=bp)
    if not sparse_output:
        out.update(edge_index=postprocess_dense(raw, remove_self_loops, degree_norm, False,
                                                edge_weight_norm), edge_weight=None)
    else:
        ei, ew = dense_to_block_diag(raw)
        ei, ew = postprocess_sparse(ei, ew, raw.size(0) * raw.size(1), remove_self_loops,
                                    degree_norm, edge_weight_norm, bp)
        out.update(edge_index=ei, edge_weight=ew)
    return out


# =============================================================================
# deterministic greedy matching used as the Graclus stand-in (input generator only)
# =============================================================================
def greedy_matching(edge_index: Tensor, edge_weight: Optional[Tensor], num_nodes: int) -> Tensor:
    """Nodes in index order; each unmatched node pairs with its heaviest unmatched
    neighbour (first on ties); label = min of the pair; then ``unique`` relabel
    (select/graclus_select.py:66-70)."""
    row, col = edge_index[0].tolist(), edge_index[1].tolist()
    w = [1.0] * len(row) if edge_weight is None else edge_weight.tolist()
    nbrs: List[List[Tuple[int, float]]] = [[] for _ in range(num_nodes)]
    for r, c, ww in zip(row, col, w):
        if r != c:
            nbrs[r].append((c, ww))
    lab = [-1] * num_nodes
    for u in range(num_nodes):
        if lab[u] >= 0:
            continue
        best, bw = -1, -float("inf")
        for v, ww in nbrs[u]:
            if lab[v] < 0 and ww > bw:
                best, bw = v, ww
        lab[u] = u
        if best >= 0:
            lab[best] = u
    _, inv = torch.unique(torch.tensor(lab), sorted=True, return_inverse=True)
    return inv


# ------------------------------------------------------------------ A0: SelectOutput.assign_all_nodes
def get_assignments(kept, edge_index, max_iter, num_nodes):  # utils/ops.py:1222-1440 (deterministic part)
    """Plain-loop restatement: kept nodes label themselves 1..K; in every round each unassigned node takes the
    label most of its already-labelled in-neighbours carry (smallest label on ties), all updates of a round
    being computed from the labels of the previous round.  Raises if nodes are left for the reference's random
    fallback (unpinned).  Returns [2, N]: node ids, consecutive supernode ids (rank of the kept node's id)."""
    kept = [int(k) for k in kept]
    label = [0] * num_nodes
    for i, k in enumerate(kept):
        label[k] = i + 1
    src, dst = edge_index[0].tolist(), edge_index[1].tolist()
    for _ in range(max_iter):
        if all(label):
            break
        votes = {}
        for s_, d_ in zip(src, dst):
            if label[s_] > 0 and label[d_] == 0:
                votes.setdefault(d_, {}).setdefault(label[s_], 0)
                votes[d_][label[s_]] += 1
        for d_, v in votes.items():
            best = max(v.values())
            label[d_] = min(c for c, n in v.items() if n == best)
    if not all(label):
        raise ValueError("nodes left for the random fallback: parity unpinned")
    target = [kept[c - 1] for c in label]            # node id of the supernode's kept node
    rank = {k: i for i, k in enumerate(sorted(set(target)))}
    return torch.tensor([list(range(num_nodes)), [rank[v] for v in target]], dtype=torch.long)
