"""Host-side helpers of the SRC hot path (public names follow reference tgp/utils/ops.py).

Shape / type dispatch lives here in Python; every pass over edge or adjacency data is a HIP
kernel (``tgp.kernels``).  Error types and message fragments follow the reference so its
tests' ``pytest.raises(..., match=...)`` expectations hold.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from .. import eps
from .. import kernels as K
from ..imports import HAS_TORCH_SPARSE, is_sparsetensor


def rank3_trace(x: Tensor) -> Tensor:
    return torch.diagonal(x, dim1=-2, dim2=-1).sum(-1)


def rank3_diag(x: Tensor) -> Tensor:
    return torch.diag_embed(x)


# ----------------------------------------------------------------------------- small predicates
def is_dense_adj(edge_index) -> bool:
    """[B,N,N] float tensor, or square 2-D float tensor (reference ops.py:267-279)."""
    if not isinstance(edge_index, Tensor) or edge_index.is_sparse:
        return False
    if edge_index.dim() == 3:
        return True
    return edge_index.dim() == 2 and edge_index.size(0) == edge_index.size(1) and edge_index.is_floating_point()


def as_compute_dtype(t):
    """fp32 and fp64 are the arithmetic types of the kernels (r5: float64 tensors run on the fp64 forms, dense GEMM path
    included, like the reference's ATen ops do for a ``model.double()`` run).  bf16 / half tensors are converted to
    fp32 on the way in -- differentiably -- and :func:`like_input_dtype` converts results back, so the dtype a caller
    sees is the one the reference would return."""
    if (isinstance(t, Tensor) and not t.is_sparse and t.is_floating_point()
            and t.dtype not in (torch.float32, torch.float64)):
        return t.float()
    return t


def like_input_dtype(out, like):
    if (isinstance(out, Tensor) and isinstance(like, Tensor) and like.is_floating_point()
            and out.is_floating_point() and out.dtype != like.dtype):
        return out.to(like.dtype)
    return out


class BatchInfo:
    """Host-side facts about a batch vector (number of graphs, graph sizes, CSR offsets, longest graph).
    Reading them costs two device round trips; one pooler call asks for them in half a dozen places (dense
    preprocessing, Reduce, Connect, Lift, the losses), so they are memoised per tensor OBJECT: the entry is
    valid only while the weak reference still resolves to the very same tensor and its version counter has
    not moved (an in-place write bumps it), which rules out stale hits from recycled memory."""

    __slots__ = ("ref", "version", "num_graphs", "sizes", "_sizes_host", "ptr", "max_nodes", "distinct", "is_sorted",
                 "memo")  # memo: facts derived from the sizes by a caller (e.g. TopkSelect's per-graph k for a ratio)

    @property
    def sizes_host(self):
        """The graph sizes as a Python list: read back on first use (the device route of batch_info needs only four
        numbers on the host; KronConnect / NDPSelect ask for the list when they size per-graph work)."""
        if self._sizes_host is None:
            self._sizes_host = self.sizes.tolist()
        return self._sizes_host


_BATCH_INFO: dict = {}
_BATCH_FACTS_ONE_LAUNCH = __import__("os").environ.get("TGP_BATCH_FACTS_ONE_LAUNCH", "1") != "0"  # A/B switch


def _batch_facts_device(batch: Tensor, info: "BatchInfo", topk_ratio: float = 0.0) -> bool:
    """Graph sizes, count, longest graph, sortedness from csrc/densify.hip's tgp_batch_facts_i64 (one read-back instead
    of bincount's and a second one).  False: an id outside [0, N] (more graph ids than nodes, or a negative id) -- the
    caller takes the torch route, which raises / sizes its output as the reference's ops do."""
    from .. import _native as N
    n = batch.numel()
    sizes = torch.empty(n + 1, dtype=torch.long, device=batch.device)
    facts = torch.empty(5, dtype=torch.long, device=batch.device)
    N.check(N.lib().tgp_batch_facts_i64(N.ptr(batch), n, N.ptr(sizes), N.ptr(facts), float(topk_ratio or 0.0),
                                        N.stream_ptr(batch.device)), "tgp_batch_facts_i64")
    max_id, flags, longest, distinct, keep = facts.tolist()
    if flags & 2:
        return False
    if topk_ratio and topk_ratio > 0:
        info.memo[("topk_total", float(topk_ratio))] = keep  # (TopkSelect's sum of k_g: no read-back of its own)
    info.num_graphs = max_id + 1
    info.sizes = sizes[:info.num_graphs]
    info.is_sorted = (flags & 1) == 0
    info.max_nodes, info.distinct = longest, distinct
    return True


_FACTS_PENDING: dict = {}  # id(batch) -> (weakref, version, ratio, launched call) of prefetch_batch_info


def prefetch_batch_info(batch: Optional[Tensor], topk_ratio: float = 0.0) -> None:
    """Launch the one-launch batch facts NOW and read them later: a caller that has device work to enqueue first (TopK's
    score pass over x) puts that work between this launch and the ``batch_info`` call that polls for the result, so the
    host wait covers the facts kernel only and the rest runs while the host goes on.  A no-op when the facts are
    memoised, the vector is not eligible, or a prefetch is already pending."""
    import weakref
    if (batch is None or not _BATCH_FACTS_ONE_LAUNCH or not batch.is_cuda or batch.dtype != torch.long
            or not batch.is_contiguous() or batch.numel() == 0 or batch.numel() > (1 << 24)):
        return
    hit = _BATCH_INFO.get(id(batch))
    if hit is not None and hit.ref() is batch and hit.version == batch._version:
        return
    if id(batch) in _FACTS_PENDING or torch.cuda.is_current_stream_capturing():
        return
    if len(_FACTS_PENDING) >= 8:
        _FACTS_PENDING.clear()  # (abandoned prefetches: their launches complete on their own)
    _FACTS_PENDING[id(batch)] = (weakref.ref(batch), batch._version, float(topk_ratio or 0.0),
                                 _facts_sorted_launch(batch, topk_ratio))


def _facts_sorted_launch(batch: Tensor, topk_ratio: float):
    """Enqueue tgp_batch_facts_sorted_i64; returns what ``_facts_sorted_finish`` needs to read its result."""
    from .. import _native as N
    dev = batch.device
    n = batch.numel()
    st = N.stream_ptr(dev)
    state = K._sps_state(dev, st, 0)
    want_plan = bool(topk_ratio and topk_ratio > 0)
    # one allocation: ptr [n + 2] | sizes [n + 1] (| k [n + 1] | koff [n + 2])
    buf = torch.empty((2 * n + 3) + ((2 * n + 3) if want_plan else 0), dtype=torch.long, device=dev)
    base = buf.data_ptr()
    o_sizes, o_k, o_koff = n + 2, 2 * n + 3, 3 * n + 4
    tag = state.next_facts_tag()
    N.check(N.lib().tgp_batch_facts_sorted_i64(batch.data_ptr(), n, base, base + 8 * o_sizes, float(topk_ratio or 0.0),
                                               (base + 8 * o_k) if want_plan else None,
                                               (base + 8 * o_koff) if want_plan else None,
                                               state.ticket.data_ptr(), state.facts_slot(tag), tag, st),
            "tgp_batch_facts_sorted_i64")
    return state, tag, buf, n, want_plan


def _batch_facts_sorted(batch: Tensor, info: "BatchInfo", topk_ratio: float = 0.0) -> bool:
    """r5: every fact of a SORTED batch vector -- CSR offsets, sizes, graph count, longest graph, non-empty graphs and,
    for a TopK selector, its per-graph keep counts with their prefix sums -- from ONE launch
    (csrc/densify.hip batch_facts_sorted_kernel: no memset, no counting atomics, no cumsum), handed over in a pinned host
    word this thread polls (no device-to-host copy, no stream synchronise).  What a forward on a NEW batch vector paid
    before: two memsets, two kernels, a copy back, a fill and a two-kernel cumsum (~60 us of device time on 2048 small
    graphs, `profiles/r05_fresh_profile_before.txt`).  False: the vector is not sorted (or holds ids outside [0, N], or
    long runs of ids without nodes) -- the general route below takes it."""
    if torch.cuda.is_current_stream_capturing():
        return False
    pend = _FACTS_PENDING.pop(id(batch), None)
    launched = None
    if pend is not None and pend[0]() is batch and pend[1] == batch._version:
        # a prefetch whose pinned slot (eight rotate: tag & 7) has been handed to a later launch is not waited for -- its
        # words are gone and the wait would spin into its bound: the facts are launched again
        fresh = pend[3][0].facts_tag - pend[3][1] < 8
        if fresh and (pend[2] == float(topk_ratio or 0.0) or pend[2] > 0):
            launched = pend[3]
            if pend[2] > 0 and not (topk_ratio and topk_ratio > 0):
                topk_ratio = pend[2]  # (the prefetch also made a TopK plan: kept)
    if launched is None:
        launched = _facts_sorted_launch(batch, topk_ratio)
    state, tag, buf, n, want_plan = launched
    o_sizes, o_k, o_koff = n + 2, 2 * n + 3, 3 * n + 4
    max_id, flags, longest, distinct, keep = state.wait_facts(tag)
    if flags:
        return False
    nb = max_id + 1
    info.num_graphs = nb
    info.ptr = buf[: nb + 1]
    info.sizes = buf[o_sizes: o_sizes + nb]
    info.is_sorted = True
    info.max_nodes, info.distinct = longest, distinct
    if want_plan:  # TopkSelect's plan (total, k, koff) came with the same launch
        info.memo[("topk", float(topk_ratio))] = (keep, buf[o_k: o_k + nb], buf[o_koff: o_koff + nb + 1])
        info.memo[("topk_total", float(topk_ratio))] = keep
    return True


def batch_info(batch: Tensor, topk_ratio: float = 0.0) -> BatchInfo:
    """``topk_ratio``: a TopK selector asking first also gets its total number of kept nodes from the same read-back."""
    import weakref
    hit = _BATCH_INFO.get(id(batch))
    if hit is not None and hit.ref() is batch and hit.version == batch._version:
        return hit
    info = BatchInfo()
    info.ref, info.version, info.memo = weakref.ref(batch), batch._version, {}
    info._sizes_host = None
    info.ptr = None
    if batch.numel() == 0:
        info.sizes = torch.zeros(0, dtype=torch.long, device=batch.device)
        info.num_graphs, info.max_nodes, info.distinct, info._sizes_host, info.is_sorted = 0, 0, 0, [], True
    elif (batch.is_cuda and batch.dtype == torch.long and batch.is_contiguous() and batch.numel() <= (1 << 24)
          and _BATCH_FACTS_ONE_LAUNCH and _batch_facts_sorted(batch, info, topk_ratio)):
        pass  # ONE launch, one pinned-word poll; the CSR offsets came with it
    elif (batch.is_cuda and batch.dtype == torch.long and batch.is_contiguous() and batch.numel() <= (1 << 24)
          and _batch_facts_device(batch, info, topk_ratio)):  # (its size table has one slot per NODE: 128 MB at the cap)
        pass  # (unsorted vectors) two launches, ONE host read of four numbers
    else:
        info.sizes = torch.bincount(batch)  # sync 1: the output length is max(batch) + 1
        info.num_graphs = info.sizes.numel()
        unsorted = (batch[1:] < batch[:-1]).any().to(info.sizes.dtype).view(1)
        host = torch.cat([info.sizes, unsorted]).tolist()  # sync 2 (B + 1 integers)
        info._sizes_host, info.is_sorted = host[:-1], host[-1] == 0
        info.max_nodes = max(info._sizes_host)
        info.distinct = info.num_graphs - info._sizes_host.count(0)  # (list.count: a generator over 2048 sizes was 60 us)
    if getattr(info, "ptr", None) is None:
        info.ptr = torch.zeros(info.num_graphs + 1, dtype=torch.long, device=batch.device)
        if info.num_graphs:
            torch.cumsum(info.sizes, 0, out=info.ptr[1:])
    if len(_BATCH_INFO) >= 16:  # a handful of live batch vectors at most; drop dead and old entries
        for key in [k for k, v in _BATCH_INFO.items() if v.ref() is None]:
            del _BATCH_INFO[key]
        while len(_BATCH_INFO) >= 16:
            del _BATCH_INFO[next(iter(_BATCH_INFO))]
    _BATCH_INFO[id(batch)] = info
    return info


def is_multi_graph_batch(batch: Optional[Tensor]) -> bool:
    if batch is None or batch.numel() == 0:
        return False
    return batch_info(batch).distinct > 1


def num_graphs_of(batch: Optional[Tensor]) -> int:
    return 1 if batch is None or batch.numel() == 0 else batch_info(batch).num_graphs


def max_graph_size(batch: Tensor) -> int:
    return batch_info(batch).max_nodes


_pooled_batch_memo: dict = {}


def build_pooled_batch(batch_size: int, num_supernodes: int, device, dtype: torch.dtype = torch.long) -> Tensor:
    """arange(B).repeat_interleave(K) (reference utils/ops.py:152-169).  A pure function of (B, K): the last few
    results are kept and handed out as a fresh copy (one launch instead of three per pooler call)."""
    key = (int(batch_size), int(num_supernodes), torch.device(device), dtype)
    hit = _pooled_batch_memo.get(key)
    if hit is None:
        if len(_pooled_batch_memo) >= 8:
            _pooled_batch_memo.clear()
        hit = torch.arange(batch_size, dtype=dtype, device=device).repeat_interleave(num_supernodes)
        _pooled_batch_memo[key] = hit
    return hit.clone()


def graph_ptr(batch: Tensor, batch_size: Optional[int] = None) -> Tuple[Tensor, Tensor]:
    """(sizes [B], ptr [B+1]) of a sorted batch vector (``batch_size`` may exceed max(batch)+1: trailing
    empty graphs)."""
    info = batch_info(batch)
    if batch_size is None or batch_size == info.num_graphs:
        return info.sizes, info.ptr
    sizes = torch.bincount(batch, minlength=batch_size)[:batch_size]
    ptr = torch.zeros(batch_size + 1, dtype=torch.long, device=batch.device)
    torch.cumsum(sizes, 0, out=ptr[1:])
    return sizes, ptr


def check_and_filter_edge_weights(edge_weight: Optional[Tensor]) -> Optional[Tensor]:
    if edge_weight is not None and edge_weight.ndim > 1:
        if edge_weight.ndim == 2 and edge_weight.size(-1) == 1:
            return edge_weight.flatten()
        raise RuntimeError(f"Edge weights must be of shape [E] or [E, 1], but got {edge_weight.shape}.")
    return edge_weight


def maybe_num_nodes(edge_index, num_nodes: Optional[int] = None) -> int:
    if num_nodes is not None:
        return num_nodes
    if isinstance(edge_index, Tensor):
        if edge_index.is_sparse:
            return max(edge_index.size(0), edge_index.size(1))
        return int(edge_index.max()) + 1 if edge_index.numel() > 0 else 0
    if is_sparsetensor(edge_index):
        return max(edge_index.size(0), edge_index.size(1))
    raise NotImplementedError


# ----------------------------------------------------------------------------- connectivity formats
def _reject_dense(edge_index: Tensor, fn: str) -> None:
    if edge_index.dim() == 3 or (edge_index.dim() == 2 and edge_index.size(0) != 2):
        raise ValueError(
            f"Dense adjacency matrices are not supported by {fn}(). Expected a sparse connectivity "
            "representation (edge_index with shape [2, E], a torch COO sparse tensor, or a "
            "torch_sparse.SparseTensor).")
    if edge_index.dim() != 2:
        raise ValueError(f"{fn}() expected edge_index with shape [2, E] when given a dense Tensor, "
                         f"got a Tensor with {edge_index.dim()} dimensions.")
    if edge_index.dtype != torch.int64:
        raise ValueError(f"{fn}() expected edge_index indices to be an integer tensor (dtype torch.long), "
                         f"got dtype={edge_index.dtype}.")


def connectivity_to_edge_index(edge_index, edge_weight: Optional[Tensor] = None):
    if isinstance(edge_index, Tensor):
        if edge_index.is_sparse:
            return edge_index.indices().clone(), edge_index.values().clone()
        _reject_dense(edge_index, "connectivity_to_edge_index")
        return edge_index, check_and_filter_edge_weights(edge_weight)
    if is_sparsetensor(edge_index):
        row, col, value = edge_index.coo()
        return torch.stack([row, col], dim=0), value
    raise NotImplementedError()


def connectivity_to_torch_coo(edge_index, edge_weight: Optional[Tensor] = None,
                              num_nodes: Optional[int] = None) -> Tensor:
    if not isinstance(edge_index, Tensor) and not is_sparsetensor(edge_index):
        raise ValueError(f"Edge index must be of type Tensor or SparseTensor, got {type(edge_index)}")
    if isinstance(edge_index, Tensor) and edge_index.is_sparse:
        return edge_index
    if isinstance(edge_index, Tensor):
        _reject_dense(edge_index, "connectivity_to_torch_coo")
        n = maybe_num_nodes(edge_index, num_nodes)
        w = check_and_filter_edge_weights(edge_weight)
        if w is None:
            w = torch.ones(edge_index.size(1), device=edge_index.device)
        return torch.sparse_coo_tensor(edge_index, w, (n, n)).coalesce()
    if not is_sparsetensor(edge_index):  # defensive (reference ops.py: final else of the type dispatch)
        raise ValueError("Edge index must be a Tensor or SparseTensor.")
    row, col, value = edge_index.coo()
    n = maybe_num_nodes(edge_index, num_nodes)
    if value is None:
        value = torch.ones(row.size(0), device=row.device)
    return torch.sparse_coo_tensor(torch.stack([row, col]), value, (n, n)).coalesce()


def connectivity_to_sparsetensor(edge_index, edge_weight: Optional[Tensor] = None,
                                 num_nodes: Optional[int] = None):
    if isinstance(edge_index, Tensor) and not edge_index.is_sparse:
        _reject_dense(edge_index, "connectivity_to_sparsetensor")
    if not HAS_TORCH_SPARSE:
        raise ImportError("Cannot convert connectivity to sparse tensor: torch_sparse is not installed.")
    from torch_sparse import SparseTensor  # pragma: no cover

    if isinstance(edge_index, SparseTensor):  # pragma: no cover
        return edge_index
    n = maybe_num_nodes(edge_index, num_nodes)  # pragma: no cover
    if edge_index.is_sparse:  # pragma: no cover
        edge_index, edge_weight = edge_index.indices().clone(), edge_index.values().clone()
    return SparseTensor.from_edge_index(edge_index, check_and_filter_edge_weights(edge_weight), (n, n))  # pragma: no cover


# ----------------------------------------------------------------------------- dense S helpers
def get_mask_from_dense_s(s: Tensor, batch: Optional[Tensor] = None) -> Tensor:
    """Pooled-supernode validity mask [B,K]: supernode k of graph b has an assigned node
    (reference ops.py:85-132; the 2-D + batch case is one segment sum instead of a Python loop)."""
    assert not s.is_sparse, "s must be a dense tensor"
    if s.dim() not in (2, 3):
        raise ValueError(f"s must have shape [N, K] or [B, N, K], got ndim={s.dim()}")
    if s.dim() == 3:
        return s.sum(dim=-2) > 0
    if batch is None:
        return (s.sum(dim=-2) > 0).unsqueeze(0)
    nb = num_graphs_of(batch)
    acc = s.new_zeros(nb, s.size(-1)).index_add_(0, batch, s)
    return acc > 0


def apply_dense_node_mask(x: Tensor, mask: Tensor) -> Tuple[Tensor, Tensor]:
    if x.dim() != 3:
        raise ValueError(f"apply_dense_node_mask expects x to be 3D [B, N, F], got ndim={x.dim()}")
    if mask.dim() != 2 or tuple(mask.shape) != tuple(x.shape[:2]):
        raise ValueError(f"apply_dense_node_mask expects mask shape [B, N]={tuple(x.shape[:2])}, "
                         f"got {tuple(mask.shape)}")
    B, Nn, F = x.shape
    valid = mask.reshape(-1).nonzero(as_tuple=True)[0]
    return x.reshape(B * Nn, F)[valid], (valid // Nn)


def expand_compacted_rows(x_compact: Tensor, valid_mask: Optional[Tensor], expected_rows: int) -> Tensor:
    if x_compact.dim() == 0:
        raise ValueError("x_compact must be at least 1D with a row dimension.")
    if valid_mask is None or valid_mask.numel() != expected_rows:
        got = None if valid_mask is None else int(valid_mask.numel())
        raise ValueError("Cannot expand compact rows: valid_mask must contain exactly "
                         f"{expected_rows} entries (got {got}).")
    idx = valid_mask.reshape(-1).nonzero(as_tuple=True)[0]
    if idx.size(0) != x_compact.size(0):
        raise ValueError(f"Cannot expand compact rows: x_compact has {x_compact.size(0)} rows but "
                         f"valid_mask marks {idx.size(0)} valid rows.")
    out = x_compact.new_zeros((expected_rows, *x_compact.shape[1:]))
    out[idx] = x_compact
    return out


def pseudo_inverse(edge_index: Tensor) -> Tensor:
    if not isinstance(edge_index, Tensor):
        raise NotImplementedError()
    sparse_in = edge_index.is_sparse
    dense = edge_index.to_dense() if sparse_in else edge_index
    inv = torch.linalg.pinv(dense.float())
    if sparse_in:
        inv = torch.where(inv.abs() < 1e-5, torch.zeros_like(inv), inv).to_sparse_coo()
    return inv


# ----------------------------------------------------------------------------- post-processing
def postprocess_adj_pool_dense(adj_pool: Tensor, remove_self_loops: bool = False, degree_norm: bool = False,
                               adj_transpose: bool = False, edge_weight_norm: bool = False) -> Tensor:
    """A8 (reference ops.py:282-335): diag <- 0, D^-1/2 A D^-1/2, / max|A| per graph; returns a new tensor."""
    if not (remove_self_loops or degree_norm or edge_weight_norm):
        return adj_pool
    squeeze = adj_pool.dim() == 2
    a = adj_pool.unsqueeze(0) if squeeze else adj_pool
    if adj_pool.requires_grad and torch.is_grad_enabled():
        if a.is_cuda and a.dtype == torch.float32 and not edge_weight_norm:
            out = _PostprocessDenseFn.apply(a, remove_self_loops, degree_norm, adj_transpose)
        else:
            out = _postprocess_dense_autograd(a, remove_self_loops, degree_norm, adj_transpose, edge_weight_norm)
        return out.squeeze(0) if squeeze else out
    flags = K.dense_flags(remove_self_loops, degree_norm, adj_transpose, edge_weight_norm)
    # NB the reference also clears the diagonal of its *input* in place (ops.py:308); every caller passes a
    # freshly computed S^T A S, so that side effect is not reproduced (it would cost an extra pass).
    out = K.postprocess_dense(a, flags)
    return out.squeeze(0) if squeeze else out


class _PostprocessDenseFn(torch.autograd.Function):
    """A8 under autograd: the native kernel forward, a closed-form backward (a dozen [B,K,K] ops where autograd
    over the elementwise form records ~45 tiny kernels per step).
        R1 = R (1 - I);  c = sum_axis(R1);  d = sqrt(max(c, eps));  out_ij = R1_ij / (d_i d_j)
        dR1_ij = G_ij / (d_i d_j) + dc_[axis index],   dc_k = -[c_k > eps] (rowsum_k(G out) + colsum_k(G out)) / (2 d_k^2)
    """

    @staticmethod
    def forward(ctx, a, remove_self_loops, degree_norm, adj_transpose):
        flags = K.dense_flags(remove_self_loops, degree_norm, adj_transpose, False)
        out = K.postprocess_dense(a, flags)
        ctx.cfg = (remove_self_loops, degree_norm, adj_transpose)
        ctx.save_for_backward(a, out)
        return out

    @staticmethod
    def backward(ctx, g):
        a, out = ctx.saved_tensors
        rsl, dn, at = ctx.cfg
        native = K.postprocess_dense_bwd(a, g, K.dense_flags(rsl, dn, at, False))  # one launch (K <= 4096)
        if native is not None:
            return native, None, None, None
        k = a.size(-1)
        keep = None
        if rsl:
            keep = 1.0 - torch.eye(k, device=a.device, dtype=a.dtype)
        ga = g
        if dn:
            r1 = a * keep if rsl else a
            c = r1.sum(-2 if at else -1)                        # [B,K]
            d = torch.sqrt(c.clamp(min=eps))
            inv = 1.0 / (d.unsqueeze(-1) * d.unsqueeze(-2))     # 1 / (d_i d_j)
            go = g * out
            dc = -(go.sum(-1) + go.sum(-2)) / (2.0 * d * d)
            dc = torch.where(c >= eps, dc, torch.zeros_like(dc))  # clamp passes the gradient where c >= eps
            ga = g * inv + (dc.unsqueeze(-2) if at else dc.unsqueeze(-1))
        if rsl:
            ga = ga * keep
        return ga, None, None, None


def _postprocess_dense_autograd(a, remove_self_loops, degree_norm, adj_transpose, edge_weight_norm):
    """Differentiable form (training): elementwise torch ops on the [B,K,K] output of the GEMMs."""
    if remove_self_loops:
        a = a * (1.0 - torch.eye(a.size(-1), device=a.device, dtype=a.dtype))
    if degree_norm:
        d = a.sum(-2 if adj_transpose else -1, keepdim=True)
        d = torch.sqrt(d.clamp(min=eps))
        a = (a / d) / d.transpose(-2, -1)
    if edge_weight_norm:
        m = a.reshape(a.size(0), -1).abs().max(dim=1, keepdim=True)[0].unsqueeze(-1)
        a = a / torch.where(m == 0, torch.ones_like(m), m)
    return a


def postprocess_adj_pool_sparse(edge_index: Tensor, edge_weight: Optional[Tensor], num_nodes: int,
                                remove_self_loops: bool = False, degree_norm: bool = False,
                                edge_weight_norm: bool = False,
                                batch_pooled: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
    """A6 (reference ops.py:338-419) on an arbitrary pooled edge list."""
    if edge_weight is not None:
        edge_weight = edge_weight.view(-1)
    if remove_self_loops or edge_weight is not None:
        from .. import functions as Fn  # (imports this module)
        edge_index, edge_weight = Fn.filter_edges(edge_index, edge_weight, None, num_nodes, remove_self_loops)
    return _normalize_pooled_edges(edge_index, edge_weight, num_nodes, degree_norm, edge_weight_norm, batch_pooled)


def _normalize_pooled_edges(edge_index, edge_weight, num_nodes, degree_norm, edge_weight_norm, batch_pooled):
    if degree_norm and edge_weight is None:
        edge_weight = torch.ones(edge_index.size(1), device=edge_index.device)  # (default dtype, as ops.py:385)
    do_ewn = edge_weight_norm and edge_weight is not None
    if (degree_norm or do_ewn) and edge_index.size(1) > 0 and edge_weight.requires_grad and torch.is_grad_enabled():
        # training through the pooled weights: differentiable torch form of the same arithmetic
        w = edge_weight
        if degree_norm:
            deg = torch.zeros(num_nodes, dtype=w.dtype, device=w.device).index_add(0, edge_index[0], w)
            dis = deg.clamp(min=eps).pow(-0.5)
            w = w * dis[edge_index[0]] * dis[edge_index[1]]
        if do_ewn:
            eb = batch_pooled[edge_index[0]]
            ng = int(eb.max()) + 1
            mx = torch.zeros(ng, dtype=w.dtype, device=w.device).scatter_reduce(0, eb, w.abs(), "amax")
            mx = torch.where(mx == 0, torch.ones_like(mx), mx)
            w = w / mx[eb]
        return edge_index, w
    if (degree_norm or do_ewn) and edge_index.size(1) > 0:
        if edge_weight.dtype not in (torch.float32, torch.float64):
            edge_weight = edge_weight.to(torch.float32)
        if not edge_weight.is_contiguous():
            edge_weight = edge_weight.contiguous()
        ng = 0
        if do_ewn:
            ng = int(batch_pooled.max()) + 1 if batch_pooled.numel() else 0
        K.normalize_edges_(edge_index, edge_weight, num_nodes, degree_norm, do_ewn, batch_pooled, ng)
    return edge_index, edge_weight


def dense_to_block_diag(adj_pool: Tensor) -> Tuple[Tensor, Tensor]:
    """A10 (reference ops.py:53-82): entries with |a| > eps as a block-diagonal edge list."""
    if adj_pool.dim() == 2:
        adj_pool = adj_pool.unsqueeze(0)
    if adj_pool.dim() != 3:
        raise ValueError("adj_pool must have shape [B, K, K] or [K, K].")
    return K.block_diag_edges(adj_pool)


# --------------------------------------------------------------------------------------------------
# SelectOutput.assign_all_nodes support (reference utils/ops.py:1177-1440).  Host-side selector logic
# (device torch ops, outside the timed Reduce + Connect path): every node that a sparse selection left
# out joins the supernode most of its already-assigned in-neighbours belong to; leftovers are dealt out
# at random inside their own graph.
# --------------------------------------------------------------------------------------------------
def get_random_map_mask(kept_nodes: Tensor, mask: Tensor, batch: Optional[Tensor] = None) -> Tensor:
    """[2, #unassigned]: unassigned node -> randomly drawn kept node (same graph when ``batch`` is given;
    kept nodes must then be grouped by graph, reference ops.py:1177-1219)."""
    todo = (~mask).nonzero().view(-1)
    pick = torch.randint(0, kept_nodes.size(0), (todo.size(0),), device=kept_nodes.device)
    if batch is not None:
        per_graph = torch.bincount(batch[kept_nodes])
        first = torch.cumsum(per_graph, 0) - per_graph  # offset of each graph's kept nodes
        g = batch[todo]
        pick = kept_nodes[pick % per_graph[g] + first[g]]
    return torch.stack([todo, pick])


def propagate_assignments_sparse(assignments: Tensor, edge_index: Tensor, kept_node_tensor: Tensor, mask: Tensor,
                                 num_clusters: int):
    """One round of label propagation over the COO edge list (reference ops.py:1222-1314): an unassigned
    destination takes the cluster (1..K, 0 = none) that most of its assigned sources carry, the smallest
    cluster id on ties.  Returns (assignments, [2, #new] node -> kept-node map, mask)."""
    src, dst = edge_index[0], edge_index[1]
    label = assignments[src]
    live = (label > 0) & ~mask[dst]
    none = torch.empty((2, 0), device=assignments.device, dtype=torch.long)
    if not bool(live.any()):
        return assignments, none, mask
    # (dst, label) pairs with multiplicities; `unique` returns them sorted by dst, then label
    pair, votes = torch.unique(dst[live] * (num_clusters + 1) + label[live], return_counts=True)
    p_dst, p_label = pair // (num_clusters + 1), pair % (num_clusters + 1)
    # stable sort by (dst, votes descending): the first entry of every dst run is its winner, and among equal
    # vote counts the smaller label stays in front
    top = int(votes.max()) + 1
    order = torch.sort(p_dst * top + (top - 1 - votes), stable=True)[1]
    p_dst, p_label = p_dst[order], p_label[order]
    first = torch.ones_like(p_dst, dtype=torch.bool)
    first[1:] = p_dst[1:] != p_dst[:-1]
    new_nodes, new_labels = p_dst[first], p_label[first]
    real = new_labels > 0  # labels beyond num_clusters alias to 0 in the packed key: nothing to assign
    if not bool(real.any()):
        return assignments, none, mask
    new_nodes, new_labels = new_nodes[real], new_labels[real]
    assignments = assignments.clone()
    assignments[new_nodes] = new_labels
    mask = mask.clone()
    mask[new_nodes] = True
    return assignments, torch.stack([new_nodes, kept_node_tensor[new_labels - 1]]), mask


def get_assignments(kept_node_indices, edge_index: Optional[Tensor] = None, max_iter: int = 5,
                    batch: Optional[Tensor] = None, num_nodes: Optional[int] = None) -> Tensor:
    """[2, N] map node -> consecutive supernode id: kept nodes are their own supernode, the others join through
    up to ``max_iter`` propagation rounds, the rest at random (reference ops.py:1317-1440)."""
    if isinstance(kept_node_indices, Tensor):
        kept = torch.squeeze(kept_node_indices).to(torch.long)
    else:
        kept = torch.tensor(kept_node_indices, dtype=torch.long)
    if num_nodes is None:
        if batch is not None:
            num_nodes = batch.size(0)
        elif edge_index is not None:
            num_nodes = int(edge_index.max()) + 1
        else:
            raise ValueError("Either num_nodes, batch, or edge_index must be provided to determine the number "
                             "of nodes")
    device = edge_index.device if edge_index is not None else (batch.device if batch is not None else kept.device)
    kept = kept.to(device)
    mask = torch.zeros(num_nodes, device=device, dtype=torch.bool)
    mask[kept] = True
    maps = [torch.stack([kept, kept])]
    if max_iter > 0:
        if edge_index is None:
            raise ValueError("edge_index must be provided when max_iter > 0")
        ei = edge_index.coalesce().indices() if (isinstance(edge_index, Tensor) and edge_index.is_sparse) \
            else edge_index
        k = kept.size(0)
        labels = torch.zeros(num_nodes, device=device, dtype=torch.long)
        labels[kept] = torch.arange(1, k + 1, device=device)
        for _ in range(max_iter):
            if bool(mask.all()):
                break
            labels, new_map, mask = propagate_assignments_sparse(labels, ei, kept, mask, k)
            if new_map.size(1) > 0:
                maps.append(new_map)
    if not bool(mask.all()):
        maps.append(get_random_map_mask(kept, mask, batch))
    out = torch.cat(maps, dim=1)
    out = out[:, out[0].argsort()]
    out[1] = torch.unique(out[1], return_inverse=True)[1]
    return out
