"""Differentiable wrappers of the native products that the operators compose.

The reference gets its gradients from ATen autograd over ``matmul`` / ``scatter`` / sparse ``mm``; the native
kernels are opaque to autograd, so each product used on a trainable path has an explicit backward here, itself
made of the same native kernels.  Every wrapper calls the kernel directly when nothing requires grad.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from . import kernels as K


def _needs_grad(*ts) -> bool:
    return torch.is_grad_enabled() and any(isinstance(t, Tensor) and t.requires_grad for t in ts)


# --------------------------------------------------------------------------------------- batched GEMM
class _BmmFn(torch.autograd.Function):
    """C = op(A) B.  trans_a False: dA = dC B^T, dB = A^T dC.  trans_a True (A stored [Kd,M]): dA = B dC^T, dB = A dC."""

    @staticmethod
    def forward(ctx, a, b, trans_a):
        ctx.save_for_backward(a, b)
        ctx.trans_a = trans_a
        return K.bmm(a, b, trans_a)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        ga = gb = None
        if ctx.needs_input_grad[0]:
            if ctx.trans_a:
                ga = K.bmm(b, g.transpose(-1, -2).contiguous())
            else:
                ga = K.bmm(g, b.transpose(-1, -2).contiguous())
            ga = _unbroadcast(ga, a)
        if ctx.needs_input_grad[1]:
            gb = K.bmm(a, g) if ctx.trans_a else K.bmm(a, g, trans_a=True)
            gb = _unbroadcast(gb, b)
        return ga, gb, None


def _unbroadcast(grad: Tensor, like: Tensor) -> Tensor:
    """Undo K.bmm's batch broadcasting (a 2-D or batch-1 operand against a batched one)."""
    if grad.dim() == 3 and like.dim() == 2:
        return grad.sum(0)
    if grad.dim() == 3 and like.dim() == 3 and like.size(0) == 1 and grad.size(0) != 1:
        return grad.sum(0, keepdim=True)
    return grad


def bmm(a: Tensor, b: Tensor, trans_a: bool = False) -> Tensor:
    return _BmmFn.apply(a, b, trans_a) if _needs_grad(a, b) else K.bmm(a, b, trans_a)


# ------------------------------------------------------------------------- products on an un-padded batch
class _SegmentGemmTnFn(torch.autograd.Function):
    """C[b] = S_b^T Y_b (reduce/base_reduce.py:170-182, connect/dense_conn.py:195-206).
    dS_b = Y_b dC_b^T, dY_b = S_b dC_b: two row-side segment products."""

    @staticmethod
    def forward(ctx, s, y, ptr, max_nodes):
        ctx.save_for_backward(s, y, ptr)
        ctx.max_nodes = max_nodes
        return K.segment_gemm_tn(s, y, ptr, max_nodes)

    @staticmethod
    def backward(ctx, g):
        s, y, ptr = ctx.saved_tensors
        g = g.contiguous()
        gs = gy = None
        if ctx.needs_input_grad[0]:
            gs = K.segment_gemm_nn(y, g.transpose(1, 2).contiguous(), ptr, ctx.max_nodes)
        if ctx.needs_input_grad[1]:
            gy = K.segment_gemm_nn(s, g, ptr, ctx.max_nodes)
        return gs, gy, None, None


def segment_gemm_tn(s: Tensor, y: Tensor, ptr: Tensor, max_nodes: int) -> Tensor:
    if _needs_grad(s, y):
        return _SegmentGemmTnFn.apply(s, y, ptr, max_nodes)
    return K.segment_gemm_tn(s, y, ptr, max_nodes)


class _SegmentGemmNnFn(torch.autograd.Function):
    """C[rows of b] = A[rows of b] M_b (lift/base_lift.py:138-247).  dA = dC M_b^T (row side again),
    dM_b = A_b^T dC_b (the reduction-side product)."""

    @staticmethod
    def forward(ctx, a, m, ptr, max_nodes):
        ctx.save_for_backward(a, m, ptr)
        ctx.max_nodes = max_nodes
        return K.segment_gemm_nn(a, m, ptr, max_nodes)

    @staticmethod
    def backward(ctx, g):
        a, m, ptr = ctx.saved_tensors
        g = g.contiguous()
        ga = gm = None
        if ctx.needs_input_grad[0]:
            ga = K.segment_gemm_nn(g, m.transpose(1, 2).contiguous(), ptr, ctx.max_nodes)
        if ctx.needs_input_grad[1]:
            gm = K.segment_gemm_tn(a, g, ptr, ctx.max_nodes)
        return ga, gm, None, None


def segment_gemm_nn(a: Tensor, m: Tensor, ptr: Tensor, max_nodes: int) -> Tensor:
    if _needs_grad(a, m):
        return _SegmentGemmNnFn.apply(a, m, ptr, max_nodes)
    return K.segment_gemm_nn(a, m, ptr, max_nodes)


# --------------------------------------------------------------------------------------------- CSR SpMM
class _SpmmFn(torch.autograd.Function):
    """T = A S for a coalesced, row-sorted edge list (connect/dense_conn.py:165,204).
    dS = A^T dT (the same kernel on the column-sorted list), dw_e = <dT[row_e], S[col_e]>."""

    @staticmethod
    def forward(ctx, edge_index, edge_weight, num_rows, s):
        ctx.save_for_backward(edge_index, edge_weight, s)
        ctx.num_rows = num_rows
        return K.spmm_sorted(edge_index, edge_weight, num_rows, s)

    @staticmethod
    def backward(ctx, g):
        edge_index, edge_weight, s = ctx.saved_tensors
        g = g.contiguous()
        gw = gs = None
        if ctx.needs_input_grad[1]:
            gw = (g[edge_index[0]] * s[edge_index[1]]).sum(-1)
        if ctx.needs_input_grad[3]:
            n = max(ctx.num_rows, s.size(0))
            ident = torch.arange(n, device=s.device)
            w = edge_weight if edge_weight is not None else torch.ones(edge_index.size(1), device=s.device)
            # transpose = re-sort by column; the list is coalesced, so nothing merges
            ei_t, w_t = K.coalesce_edges(edge_index.flip(0), w, ident, n, "sum", remove_self_loops=False,
                                         eps_filter=False)
            gs = K.spmm_sorted(ei_t, w_t, s.size(0), g)
        return None, gw, None, gs


def spmm_sorted(edge_index: Tensor, edge_weight: Optional[Tensor], num_rows: int, s: Tensor) -> Tensor:
    if _needs_grad(edge_weight, s):
        return _SpmmFn.apply(edge_index, edge_weight, num_rows, s)
    return K.spmm_sorted(edge_index, edge_weight, num_rows, s)


# ------------------------------------------------------------------------ duplicate-summing coalesce
class _CoalescedWeightsFn(torch.autograd.Function):
    """Weights of a sum-coalesced list as a function of the input weights: out[j] = sum of the inputs that
    landed on entry j, so dw_e = dout[slot(e)].  The forward values come from the native kernel."""

    @staticmethod
    def forward(ctx, edge_weight, slot, out_values):
        ctx.save_for_backward(slot)
        return out_values.clone()

    @staticmethod
    def backward(ctx, g):
        (slot,) = ctx.saved_tensors
        return g[slot], None, None


def coalesce_sum(edge_index: Tensor, edge_weight: Tensor, num_nodes: int):
    """Row-sorted, duplicate-summed copy of the list (what ``.coalesce()`` does, connect/dense_conn.py:163,202)."""
    if edge_index.size(1) > 0:
        # PyG hands out coalesced lists; one fused comparison + the same single host round trip the count -> fill
        # pair would cost tells us the sort can be skipped altogether (remembered per tensor object: five launches and
        # the round trip once, not once per forward)
        if K.coalesced_memo(edge_index, num_nodes):
            return edge_index, edge_weight
        key = edge_index[0] * num_nodes + edge_index[1]
        if bool((key[1:] > key[:-1]).all()):
            K.remember_coalesced(edge_index, num_nodes)
            return edge_index, edge_weight
    ident = torch.arange(num_nodes, device=edge_index.device)
    ei, ew = K.coalesce_edges(edge_index, edge_weight, ident, num_nodes, "sum", remove_self_loops=False,
                              eps_filter=False)
    if _needs_grad(edge_weight) and ei.size(1) > 0:
        slot = torch.searchsorted(ei[0] * num_nodes + ei[1], edge_index[0] * num_nodes + edge_index[1])
        ew = _CoalescedWeightsFn.apply(edge_weight, slot, ew)
    return ei, ew


# ------------------------------------------------------------------- edge weights through sparse Connect
class _FilteredWeightsFn(torch.autograd.Function):
    """Weights of the edges that survive the subgraph / self-loop / eps filters (connect/base_conn.py:79-82,
    utils/ops.py:370-380): a pass-through of the kept inputs, so dw_e = dout[position of e] for kept edges and 0
    for dropped ones.  ``edge_id`` (input position of every kept edge) comes from the native fill kernel."""

    @staticmethod
    def forward(ctx, edge_weight, edge_id, holder):
        ctx.save_for_backward(edge_id)
        ctx.shape = edge_weight.shape
        return holder[0]  # the values the kernel compacted (handed over in a tuple: not an input tensor)

    @staticmethod
    def backward(ctx, g):
        (edge_id,) = ctx.saved_tensors
        n = 1
        for d in ctx.shape:
            n *= d
        gi = torch.zeros(n, dtype=g.dtype, device=g.device)
        gi[edge_id] = g  # kept edges are distinct inputs: a plain scatter
        return gi.view(ctx.shape), None, None


def filter_edges(edge_index: Tensor, edge_weight: Optional[Tensor], node_index: Optional[Tensor], num_nodes: int,
                 remove_self_loops: bool, member_directory: Optional[Tensor] = None):
    """K.filter_edges whose pooled weights stay differentiable w.r.t. ``edge_weight``."""
    if not _needs_grad(edge_weight):
        return K.filter_edges(edge_index, edge_weight, node_index, num_nodes, remove_self_loops,
                              member_directory=member_directory)
    ei, ew, eid = K.filter_edges(edge_index, edge_weight.detach(), node_index, num_nodes, remove_self_loops,
                                 want_edge_id=True, member_directory=member_directory)
    w32 = edge_weight if edge_weight.dtype in (torch.float32, torch.float64) else edge_weight.float()
    return ei, _FilteredWeightsFn.apply(w32.reshape(-1), eid, (ew,))


class _CoalescedEdgeWeightsFn(torch.autograd.Function):
    """Weights of ``coalesce(cluster_index[edge_index], w, reduce=op)`` (connect/base_conn.py:86-89) as a function of
    the input weights; the values come from the native kernel, the backward is the one of PyG's ``scatter``:
    sum: dw_e = g[slot e]; mean: / group size; min / max: split evenly among the group's extremal entries
    (torch ``scatter_reduce`` amax / amin rule); mul: g out / w_e.  Entries whose merged edge was filtered out (self
    loop, |w| <= eps) get no gradient -- the reference's masks are not differentiable either."""

    @staticmethod
    def forward(ctx, edge_weight, slot, valid, op, holder):
        out = holder[0]
        ctx.save_for_backward(edge_weight, slot, valid, out)
        ctx.op = op
        return out

    @staticmethod
    def backward(ctx, g):
        w, slot, valid, out = ctx.saved_tensors
        op = ctx.op
        gs = g[slot]
        if op in ("sum", "add"):
            gi = gs
        elif op == "mean":
            cnt = torch.bincount(slot[valid], minlength=out.numel()).to(g.dtype)
            gi = gs / cnt[slot].clamp(min=1)
        elif op in ("min", "max"):
            sel = valid & (w == out[slot])
            cnt = torch.bincount(slot[sel], minlength=out.numel()).to(g.dtype)
            gi = torch.where(sel, gs / cnt[slot].clamp(min=1), torch.zeros_like(gs))
        else:  # mul: surviving groups have |prod| > eps, so no factor is zero
            gi = gs * out[slot] / w
        return torch.where(valid, gi, torch.zeros_like(gi)), None, None, None, None


def coalesce_edges(edge_index: Tensor, edge_weight: Optional[Tensor], cluster_index: Tensor, num_supernodes: int,
                   reduce_op: str, remove_self_loops: bool, assign_index=None, csr=None):
    """K.coalesce_edges whose pooled weights stay differentiable w.r.t. ``edge_weight``."""
    if not _needs_grad(edge_weight):
        return K.coalesce_edges(edge_index, edge_weight, cluster_index, num_supernodes, reduce_op, remove_self_loops,
                                assign_index=assign_index, csr=csr)
    w = edge_weight.reshape(-1)
    w32 = w if w.dtype in (torch.float32, torch.float64) else w.float()
    ei, ew = K.coalesce_edges(edge_index, w32.detach(), cluster_index, num_supernodes, reduce_op, remove_self_loops,
                              assign_index=assign_index, csr=csr)
    if ei.size(1) == 0:
        return ei, ew
    key_in = cluster_index[edge_index[0]] * num_supernodes + cluster_index[edge_index[1]]
    key_out = ei[0] * num_supernodes + ei[1]  # row-major sorted and unique
    slot = torch.searchsorted(key_out, key_in).clamp_(max=key_out.numel() - 1)
    valid = key_out[slot] == key_in
    return ei, _CoalescedEdgeWeightsFn.apply(w32, slot, valid, reduce_op, (ew,))


# ------------------------------------------------------------------------------------ block-diagonal export
class _BlockDiagWeightsFn(torch.autograd.Function):
    """Values of dense_to_block_diag (utils/ops.py:53-82) as a function of the dense tensor: a gather at the
    positions the native compaction reported; backward scatters the edge gradients back."""

    @staticmethod
    def forward(ctx, adj, b, r, c):
        ctx.save_for_backward(b, r, c)
        ctx.shape = adj.shape
        return adj[b, r, c]

    @staticmethod
    def backward(ctx, g):
        b, r, c = ctx.saved_tensors
        ga = torch.zeros(ctx.shape, dtype=g.dtype, device=g.device)
        ga[b, r, c] = g
        return ga, None, None, None


def block_diag_edges(adj_pool: Tensor, relabel: Optional[Tensor] = None, remove_self_loops: bool = False,
                     inverse: Optional[Tensor] = None):
    """``inverse``: new id -> flat supernode id b*K + k (needed with ``relabel`` to find the dense positions)."""
    ei, ew = K.block_diag_edges(adj_pool, relabel, remove_self_loops)
    if _needs_grad(adj_pool) and ei.size(1) > 0 and (relabel is None or inverse is not None):
        a = adj_pool if adj_pool.dim() == 3 else adj_pool.unsqueeze(0)
        k = a.size(1)
        r, c = (ei[0], ei[1]) if relabel is None else (inverse[ei[0]], inverse[ei[1]])
        b = torch.div(r, k, rounding_mode="floor")
        ew = _BlockDiagWeightsFn.apply(a, b, r - b * k, c - b * k)
    return ei, ew


# ------------------------------------------------------------------------------------------ sparse Lift
class _SparseLiftFn(torch.autograd.Function):
    """X_lift[row] += v * X_pool[col] (lift/base_lift.py:102-111) = the sparse Reduce kernel with the roles of
    node and supernode swapped; dX_pool is the Reduce itself, dv a row-dot."""

    @staticmethod
    def forward(ctx, x_pool, values, row, col, index, back_index_fn):
        ctx.save_for_backward(x_pool, values, row, col)
        ctx.back_index_fn = back_index_fn
        return K.reduce_sparse(x_pool, col, values, index)

    @staticmethod
    def backward(ctx, g):
        x_pool, values, row, col = ctx.saved_tensors
        g = g.contiguous()
        gx = gv = None
        if ctx.needs_input_grad[0]:
            gx = K.reduce_sparse(g, row, values, ctx.back_index_fn())
        if ctx.needs_input_grad[1]:
            gv = K.pair_dot(g, row, x_pool, col) if g.dim() == 2 else (g[row] * x_pool[col]).sum(-1)
        return gx, gv, None, None, None, None


def sparse_lift(x_pool: Tensor, values: Tensor, row: Tensor, col: Tensor, index, back_index_fn) -> Tensor:
    """``index``: inverted index over ``row`` (output nodes); ``back_index_fn()``: the one over ``col``."""
    if _needs_grad(x_pool, values):
        return _SparseLiftFn.apply(x_pool, values, row, col, index, back_index_fn)
    return K.reduce_sparse(x_pool, col, values, index)


# ------------------------------------------------------------------------------- sparse -> padded dense
class _ToDenseBatchFn(torch.autograd.Function):
    """to_dense_batch (src.py:448-450) with the native scatter forward; the backward is the matching gather
    (torch's own backward of the indexed assignment sorts the indices first)."""

    @staticmethod
    def forward(ctx, x, batch, ptr, num_graphs, max_nodes, also_zero=None):
        ctx.set_materialize_grads(False)  # (the mask's "gradient" would be a zero fill of [B,N] bytes per step)
        out, mask = K.to_dense_batch(x, batch, ptr, num_graphs, max_nodes, also_zero)
        ctx.save_for_backward(batch, ptr)
        ctx.max_nodes = max_nodes
        ctx.mark_non_differentiable(mask)
        return out, mask

    @staticmethod
    def backward(ctx, g, _gmask):
        batch, ptr = ctx.saved_tensors
        if g is None:
            return None, None, None, None, None, None
        # one gather kernel; nodes beyond a caller-imposed max_num_nodes were dropped in the forward: zero gradient
        return K.from_dense_batch(g, batch, ptr, ctx.max_nodes), None, None, None, None, None


def to_dense_batch(x: Tensor, batch: Tensor, ptr: Tensor, num_graphs: int, max_nodes: int,
                   also_zero: Optional[Tensor] = None):
    if _needs_grad(x):
        return _ToDenseBatchFn.apply(x, batch, ptr, num_graphs, max_nodes, also_zero)
    return K.to_dense_batch(x, batch, ptr, num_graphs, max_nodes, also_zero)


class _ToDenseAdjFn(torch.autograd.Function):
    """to_dense_adj (src.py:434-443) with the native scatter-add forward; the gradient of the edge weights is the
    matching gather (the reference gets it from ATen autograd over PyG's scatter)."""

    @staticmethod
    def forward(ctx, edge_weight, edge_index, batch, ptr, num_graphs, max_nodes, transposed, zeroed_out=None):
        ctx.save_for_backward(edge_index, batch, ptr)
        ctx.max_nodes, ctx.transposed, ctx.shape = max_nodes, transposed, edge_weight.shape
        return K.to_dense_adj(edge_index, edge_weight, batch, ptr, num_graphs, max_nodes, transposed, zeroed_out)

    @staticmethod
    def backward(ctx, g):
        edge_index, batch, ptr = ctx.saved_tensors
        gw = K.from_dense_adj(g.contiguous(), edge_index, batch, ptr, ctx.max_nodes, ctx.transposed)
        return gw.view(ctx.shape), None, None, None, None, None, None, None


def to_dense_adj(edge_index: Tensor, edge_weight: Optional[Tensor], batch: Tensor, ptr: Tensor, num_graphs: int,
                 max_nodes: int, transposed: bool, zeroed_out: Optional[Tensor] = None) -> Tensor:
    if _needs_grad(edge_weight):
        return _ToDenseAdjFn.apply(edge_weight, edge_index, batch, ptr, num_graphs, max_nodes, transposed, zeroed_out)
    return K.to_dense_adj(edge_index, edge_weight, batch, ptr, num_graphs, max_nodes, transposed, zeroed_out)


# ---------------------------------------------------------------------------------------- Linear layer
_WHOLE_RANGE: dict = {}


def _uniform_ptr(num_graphs: int, n: int, device) -> Tensor:
    """[0, n, 2n, ...]: a padded batch seen as an un-padded one (segment products with split node ranges)."""
    key = ("u", num_graphs, n, str(device))
    if key not in _WHOLE_RANGE:
        if len(_WHOLE_RANGE) > 64:
            _WHOLE_RANGE.clear()
        _WHOLE_RANGE[key] = torch.arange(num_graphs + 1, dtype=torch.long, device=device) * n
    return _WHOLE_RANGE[key]


def _whole_range(n: int, device) -> Tensor:
    key = (n, str(device))
    if key not in _WHOLE_RANGE:
        if len(_WHOLE_RANGE) > 64:
            _WHOLE_RANGE.clear()
        _WHOLE_RANGE[key] = torch.tensor([0, n], dtype=torch.long, device=device)
    return _WHOLE_RANGE[key]


class _LinearFn(torch.autograd.Function):
    """Y = X W^T + b of the selector MLP (select/mlp_select.py:67).  Forward and dX are plain library GEMMs;
    dW = dY^T X reduces over every node of the batch into a tiny [out,in] matrix, which the library runs as a
    handful of workgroups (136 us at 32768 x 64 -> 128) -- here it is the node-range-split segment product."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        gx = gw = gb = None
        g2 = g.reshape(-1, g.size(-1))
        if ctx.needs_input_grad[0]:
            gx = g.matmul(weight)
        if ctx.needs_input_grad[1]:
            x2 = x.reshape(-1, x.size(-1))
            gw = _tall_skinny_tn(g2, x2)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g2.sum(0)
        return gx, gw, gb


def _tall_skinny_tn(a: Tensor, b: Tensor) -> Tensor:
    """a^T b for a [n, p], b [n, q] with n >> p, q (a weight gradient: every node row contributes to a tiny matrix).
    Narrow operands: the rows are cut into slabs of 64, every slab is one batch element of the one-wave-per-strip
    product (both operands read in MFMA layout, no LDS) and the slab results are added -- thousands of waves where
    the tiled GEMM would run a single mostly-padded 64 x 128 tile per split (83 -> ~15 us at n = 122880, 20 x 32)."""
    n, p, q = a.size(0), a.size(1), b.size(1)
    slab = 64
    if p <= 512 and q <= 64 and n >= 512 * slab:
        full = (n // slab) * slab
        out = K.bmm(a[:full].view(-1, slab, p), b[:full].view(-1, slab, q), trans_a=True).sum(0)
        if full < n:
            out = out + K.bmm(a[full:].unsqueeze(0), b[full:].unsqueeze(0), trans_a=True)[0]
        return out
    return K.segment_gemm_tn(a, b, _whole_range(n, a.device), n)[0]


class _MlpSelectFn(torch.autograd.Function):
    """S = softmax(X W^T + b) * mask (select/mlp_select.py:139-145) in one native pass; backward: dY = S (dS - <dS,S>)
    (one kernel), dX = dY W (fp32-MFMA GEMM), dW = dY^T X (node-range-split product), db = column sums of dY."""

    @staticmethod
    def forward(ctx, x, weight, bias, mask):
        ctx.set_materialize_grads(False)
        s = K.mlp_select(x, weight, bias, mask)
        ctx.save_for_backward(x, weight, s)
        ctx.has_bias = bias is not None
        return s

    @staticmethod
    def backward(ctx, g):
        x, weight, s = ctx.saved_tensors
        if g is None:
            return None, None, None, None
        if K.mlp_select_bwd_fits(s.size(-1), x.size(-1)):  # r5: one launch for all three gradients
            need = ctx.needs_input_grad
            gx, gw, gb = K.mlp_select_bwd(s, g, x, weight, want_gx=need[0], want_gw=need[1],
                                          want_gb=ctx.has_bias and need[2])
            return (gx.view(x.shape) if gx is not None else None), gw, gb, None
        dy = K.softmax_bwd(s, g)
        dy2 = dy.reshape(-1, dy.size(-1))
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = K.bmm(dy2, weight).view(x.shape)
        if ctx.needs_input_grad[1]:
            gw = _tall_skinny_tn(dy2, x.reshape(-1, x.size(-1)))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = dy2.sum(0)
        return gx, gw, gb, None


def mlp_select(x: Tensor, weight: Tensor, bias: Optional[Tensor], mask: Optional[Tensor]) -> Tensor:
    """Last layer of MLPSelect: softmax(linear(x)) * mask, one kernel."""
    if _needs_grad(x, weight, bias):
        return _MlpSelectFn.apply(x, weight, bias, mask)
    return K.mlp_select(x, weight, bias, mask)


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor]) -> Tensor:
    if x.is_cuda and x.dtype == torch.float32 and _needs_grad(x, weight, bias):
        return _LinearFn.apply(x, weight, bias)
    return torch.nn.functional.linear(x, weight, bias)


# ------------------------------------------------------------------------------------ per-edge <S_i, S_j>
class _EdgeDotFn(torch.autograd.Function):
    """ss[e] = <S[row_e], S[col_e]>.  dS[i] = sum_{e: row_e = i} g_e S[col_e] + sum_{e: col_e = i} g_e S[row_e]:
    two runs of the segmented gather-sum kernel of the sparse Reduce over inverted indices of the edge list
    (deterministic; torch's backward of the two gathers is a pair of sort-based index_add's over [E,K])."""

    @staticmethod
    def forward(ctx, s, edge_index):
        ctx.save_for_backward(s, edge_index)
        return K.edge_dot(s, edge_index)

    @staticmethod
    def backward(ctx, g):
        s, edge_index = ctx.saved_tensors
        g = g.contiguous()
        n = s.size(0)
        row, col = edge_index[0], edge_index[1]
        gs = K.reduce_sparse(s, col, g, K.build_assign_index(row, n))
        gs = gs + K.reduce_sparse(s, row, g, K.build_assign_index(col, n))
        return gs, None


def edge_dot(s: Tensor, edge_index: Tensor) -> Tensor:
    return _EdgeDotFn.apply(s, edge_index) if _needs_grad(s) else K.edge_dot(s, edge_index)


# ------------------------------------------------------------------ TopkSelect, min_score mode
class _SegmentSoftmaxSelectFn(torch.autograd.Function):
    """(prob, node_index) of TopkSelect's min_score mode (select/topk_select.py:186-194) from the native kernels;
    d prob / d score is the per-graph softmax Jacobian: g_in = p * (g - sum_graph(g * p))."""

    @staticmethod
    def forward(ctx, score, ptr, batch, min_score):
        prob, node_index = K.topk_minscore(score, ptr, min_score)
        ctx.save_for_backward(prob, batch)
        ctx.num_graphs = ptr.numel() - 1
        ctx.mark_non_differentiable(node_index)
        return prob, node_index

    @staticmethod
    def backward(ctx, g, _unused):
        prob, batch = ctx.saved_tensors
        gp = g * prob
        seg = gp.new_zeros(ctx.num_graphs).index_add_(0, batch, gp)
        return gp - prob * seg[batch], None, None, None


def segment_softmax_select(score: Tensor, ptr: Tensor, batch: Tensor, min_score: float):
    if _needs_grad(score):
        return _SegmentSoftmaxSelectFn.apply(score, ptr, batch, min_score)
    return K.topk_minscore(score, ptr, min_score)


# ------------------------------------------------------------------ gather at unique positions
class _TakeUniqueFn(torch.autograd.Function):
    """x[index] for an index without repeats (the kept nodes of a top-k selection): the backward is a plain
    scatter into zeros, where torch's generic indexing backward sorts the indices to merge duplicates."""

    @staticmethod
    def forward(ctx, x, index):
        ctx.save_for_backward(index)
        ctx.n = x.size(0)
        return x[index]

    @staticmethod
    def backward(ctx, g):
        (index,) = ctx.saved_tensors
        out = torch.zeros((ctx.n,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        out[index] = g
        return out, None


def take_unique(x: Tensor, index: Tensor) -> Tensor:
    return _TakeUniqueFn.apply(x, index) if _needs_grad(x) else x[index]


# --------------------------------------------------------------------------------- TopK scoring x.w
class _RowDotFn(torch.autograd.Function):
    """score = x w (select/topk_select.py:176): one pass over x; dx = g w^T, dw = x^T g (one more pass)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return K.row_dot(x, w)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = g.unsqueeze(1) * w.reshape(1, -1)
        if ctx.needs_input_grad[1]:
            gw = K.weighted_colsum(x, g.contiguous()).view_as(w)
        return gx, gw


def row_dot(x: Tensor, w: Tensor) -> Tensor:
    return _RowDotFn.apply(x, w) if _needs_grad(x, w) else K.row_dot(x, w)


class _TopkScoreFn(torch.autograd.Function):
    """TopkSelect's ratio-mode score act(x w / ||w||_2), act = tanh or identity (select/topk_select.py:176-184), as one
    node of the graph: forward = the one pass over x (+ tanh), backward from t = x w / ||w||:
    dx = (g_t / ||w||) w^T,  dw = x^T (g_t / ||w||) - (sum_i g_t[i] t[i]) w / ||w||^2,  g_t = g (1 - s^2) for tanh."""

    @staticmethod
    def forward(ctx, x, w, use_tanh):
        t = K.topk_score(x, w, False)
        s = torch.tanh(t) if use_tanh else t
        ctx.save_for_backward(x, w, t, s)
        ctx.use_tanh = use_tanh
        return s

    @staticmethod
    def backward(ctx, g):
        x, w, t, s = ctx.saved_tensors
        gt = torch.ops.aten.tanh_backward(g, s) if ctx.use_tanh else g
        inv = w.norm(p=2).reciprocal()
        gtn = (gt * inv).contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = gtn.unsqueeze(1) * w.reshape(1, -1)
        if ctx.needs_input_grad[1]:
            gw = K.weighted_colsum(x, gtn).view_as(w) - ((gtn * t).sum() * inv) * w
        return gx, gw, None


def topk_score(x: Tensor, w: Tensor, use_tanh: bool) -> Tensor:
    return _TopkScoreFn.apply(x, w, use_tanh) if _needs_grad(x, w) else K.topk_score(x, w, use_tanh)


class _TopkPoolTrainFn(torch.autograd.Function):
    """The ONE autograd node of TopK pooling's trained path (r5): x' = s_a x[i_a] and the kept scores s_a = act(x w /
    ||w||)[i_a] (the values of S) as functions of x and the projection w.  The forward values come from the caller --
    the inference kernels, run without a graph: fused score, selection, the one-launch Reduce + Connect where it
    applies -- this node supplies the backward, one pass over the kept rows (kernels.topk_pool_bwd), where the
    operator-by-operator graph (score, indexing, sparse Reduce) ran ~18 launches."""

    @staticmethod
    def forward(ctx, x, w, computed, node_index, cluster_index, use_tanh):
        x_pool, values = computed
        ctx.save_for_backward(x, w, node_index, cluster_index, values)
        ctx.use_tanh = use_tanh
        ctx.set_materialize_grads(False)
        return x_pool, values

    @staticmethod
    def backward(ctx, g_xpool, g_values):
        x, w, node_index, cluster_index, values = ctx.saved_tensors
        want_gx, want_gw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if g_xpool is None and g_values is None:
            return None, None, None, None, None, None
        gx, gw = K.topk_pool_bwd(x, node_index, cluster_index, values, g_xpool, g_values, w, ctx.use_tanh, want_gx,
                                 want_gw)
        return gx, (gw.view_as(w) if gw is not None else None), None, None, None, None


def topk_pool_train(x: Tensor, w: Tensor, x_pool: Tensor, values: Tensor, node_index: Tensor, cluster_index: Tensor,
                    use_tanh: bool):
    """``(x_pool, values)`` attached to the graph of ``x`` and ``w`` (see :class:`_TopkPoolTrainFn`)."""
    return _TopkPoolTrainFn.apply(x, w, [x_pool, values], node_index, cluster_index, use_tanh)


# ------------------------------------------------- A S and A^T S shared between Connect and the link loss
class LossPair(tuple):
    """Two scalar auxiliary losses that left a fused Function as separate 0-dim outputs (MinCut: the batch means of the
    cut / orthogonality terms; DiffPool: link / entropy).  Selecting them from a [2] tensor instead would put two
    select-backwards, two zero fills and an add into every training step."""
    __slots__ = ()


def _pool_small_forward(sd, ad, xd, flags, want_raw, want_terms, diff_scales, graph_sizes, scalars, pre=None):
    """Shared forward of the two fused Functions below: (x_pool, raw, adj_pool, terms, diff, loss_a, loss_b)."""
    from . import kernels as K
    stats = None
    if pre is None and diff_scales is not None and not want_terms:
        # DiffPool (r6): the kernel leaves per-graph records instead of MinCut's terms; both losses are one tail launch
        x_pool, raw, adj_pool, stats = K.dense_pool(sd, ad, xd, flags, want_raw=want_raw, want_post=True, diff_stats=True)
        terms = sd.new_empty(0) if stats is not None else None
    elif pre is None:
        x_pool, raw, adj_pool, terms = K.dense_pool(sd, ad, xd, flags, want_raw=want_raw, want_post=True,
                                                    mincut_terms=True)
    else:
        x_pool, raw, adj_pool, terms = pre[:4]
        stats = pre[4] if len(pre) > 4 else None
    if terms is None:
        raise RuntimeError("the batch does not take the one-wave-per-graph kernel")
    empty = sd.new_empty(0)
    diff = empty
    if diff_scales is not None and stats is not None:
        diff = K.diffpool_stats_tail(stats, diff_scales[0], diff_scales[1])
    elif diff_scales is not None:
        diff = K.diffpool_loss_tail(sd, ad, graph_sizes, diff_scales[0], diff_scales[1])
    la = lb = empty
    if scalars:
        if diff_scales is not None:
            la, lb = diff[0], diff[1]
        elif want_terms:
            both = terms.mean(dim=1)
            la, lb = both[0], both[1]
    return x_pool, (empty if raw is None else raw), adj_pool, terms, diff, la, lb


class _DensePoolSmallFn(torch.autograd.Function):
    """Reduce + Connect (+ MinCut's two loss tails, + DiffPool's two losses) of a batch of small graphs as ONE kernel in
    each direction (csrc/dense_graph_kernels.h: dense_pool_small_kernel / dense_pool_small_bwd_kernel).  Outputs:
    x_pool, raw S^T A S, post-processed adj_pool, terms [2,B], diff [2], and -- ``scalars`` -- the two losses as 0-dim
    tensors (MinCut: means of the terms over the batch; DiffPool: diff[0], diff[1]), which are then the differentiable
    form; raw / terms / diff are only differentiated when they were asked for.  Upstream gradients that are missing
    stay missing (no zero tensors are made for them): each is a NULL pointer of the C call."""

    @staticmethod
    def forward(ctx, s, adj, x, flags, want_raw, want_terms, diff_scales, graph_sizes, scalars=False):
        ctx.set_materialize_grads(False)
        sd, ad = s.detach(), adj.detach()
        x_pool, raw, adj_pool, terms, diff, la, lb = _pool_small_forward(
            sd, ad, x.detach(), flags, want_raw, want_terms, diff_scales, graph_sizes, scalars)
        ctx.save_for_backward(s, adj, x, diff)
        ctx.flags = flags
        ctx.want_gx = x.requires_grad
        ctx.diff_scales = diff_scales
        nd = [] if want_raw else [raw]
        if scalars or not want_terms:
            nd.append(terms)
        if scalars or diff_scales is None:
            nd.append(diff)
        if not scalars or (diff_scales is None and not want_terms):
            nd += [la, lb]
        ctx.mark_non_differentiable(*nd)
        return x_pool, raw, adj_pool, terms, diff, la, lb

    @staticmethod
    def backward(ctx, g_x, g_raw, g_adj, g_terms, g_diff, g_la, g_lb):
        from . import kernels as K
        s, adj, x, diff = ctx.saved_tensors
        gs, gx = _pool_small_backward(ctx, s, adj, x, diff, g_x, g_raw, g_adj, g_terms, g_diff, g_la, g_lb)
        return gs.to(s.dtype), None, (gx.to(x.dtype) if gx is not None else None), None, None, None, None, None, None


def _pool_small_backward(ctx, s, adj, x, diff, g_x, g_raw, g_adj, g_terms, g_diff, g_la, g_lb):
    from . import kernels as K
    if g_raw is not None and g_raw.numel() == 0:
        g_raw = None
    if g_terms is not None and g_terms.numel() == 0:
        g_terms = None
    ds = ctx.diff_scales
    mean_terms, diff_pair = (None, None), (None, None)
    if ds is not None:
        if g_diff is not None and g_diff.numel() == 0:
            g_diff = None
        diff_pair = (g_la, g_lb)
    else:
        g_diff, ds = None, (0.0, 0.0)
        mean_terms = (g_la, g_lb)
    need_losses = g_diff is not None or diff_pair[0] is not None or diff_pair[1] is not None
    return K.dense_pool_small_bwd(s, adj, x, ctx.flags, g_x, g_adj, g_raw, g_terms, want_gx=ctx.want_gx,
                                  g_diff=g_diff, diff_losses=diff if need_losses else None,
                                  link_scale=ds[0], ent_scale=ds[1], g_mean_terms=mean_terms, g_diff_pair=diff_pair)


def dense_pool_small(s: Tensor, adj: Tensor, x: Tensor, flags: int, want_raw: bool, want_terms: bool,
                     diff_scales=None, graph_sizes: Optional[Tensor] = None, loss_scalars: bool = False):
    """Differentiable fused Reduce + Connect for batches ``kernels.dense_pool_is_small`` accepts (adj gets no gradient:
    callers check ``adj.requires_grad`` first): (x_pool, raw, adj_pool, terms, diff).  ``diff_scales = (link_scale,
    ent_scale)``: also DiffPool's link and entropy losses [2], differentiated by the same backward launch.
    ``loss_scalars``: the auxiliary losses come as a :class:`LossPair` of 0-dim tensors instead -- in place of ``terms``
    the batch means of MinCut's two terms, in place of ``diff`` its two entries."""
    out = _DensePoolSmallFn.apply(s, adj, x, flags, want_raw, want_terms, diff_scales, graph_sizes, loss_scalars)
    if not loss_scalars:
        return out[:5]
    pair = LossPair((out[5], out[6]))
    if diff_scales is not None:
        return out[0], out[1], out[2], out[3], pair
    return out[0], out[1], out[2], (pair if want_terms else out[3]), out[4]


class _SelectPoolSmallFn(torch.autograd.Function):
    """MLPSelect's last Linear + softmax + mask, Reduce, Connect, post-processing and the loss tails of a batch of small
    graphs as ONE autograd node: forward = ``tgp_dense_pool_select_f32`` (one launch; DiffPool's two losses one more
    call), backward = ``tgp_dense_pool_small_bwd_f32`` + ``tgp_mlp_select_bwd_f32`` (two launches: the selector's
    backward adds its dX into the buffer the pooling backward wrote, so autograd has nothing to accumulate).  A MinCut
    training step on a PROTEINS-shaped batch was 34 device launches with the selector and the pooling as two nodes and
    the losses selected from small tensors (profiles/r05_e2e_train_steps.txt)."""

    @staticmethod
    def forward(ctx, x, adj, weight, bias, mask, flags, want_raw, want_terms, diff_scales, graph_sizes, want_batch):
        from . import kernels as K
        ctx.set_materialize_grads(False)
        xd, ad = x.detach(), adj.detach()
        dstats = diff_scales is not None and not want_terms  # DiffPool (r6): per-graph records instead of MinCut's terms
        s, x_pool, raw, adj_pool, terms, bp = K.dense_pool_select(
            xd, ad, weight.detach(), None if bias is None else bias.detach(), mask, flags, want_raw=want_raw,
            mincut_terms=not dstats, want_batch=True, diff_stats=dstats) if want_batch else K.dense_pool_select(
            xd, ad, weight.detach(), None if bias is None else bias.detach(), mask, flags, want_raw=want_raw,
            mincut_terms=not dstats, diff_stats=dstats) + (torch.empty(0, dtype=torch.long, device=x.device),)
        pre = (x_pool, raw, adj_pool, s.new_empty(0), terms) if dstats else (x_pool, raw, adj_pool, terms)
        x_pool, raw, adj_pool, terms, diff, la, lb = _pool_small_forward(
            s, ad, xd, flags, want_raw, want_terms, diff_scales, graph_sizes, True, pre=pre)
        ctx.save_for_backward(s, adj, x, diff, weight)
        ctx.flags = flags
        ctx.want_gx = x.requires_grad  # (node features that are data, not activations: neither backward forms dX)
        ctx.diff_scales = diff_scales
        ctx.has_bias = bias is not None
        nd = [terms, diff, bp] + ([] if want_raw else [raw])
        if diff_scales is None and not want_terms:
            nd += [la, lb]
        ctx.mark_non_differentiable(*nd)
        return s, x_pool, raw, adj_pool, terms, diff, la, lb, bp

    @staticmethod
    def backward(ctx, g_s, g_x, g_raw, g_adj, _g_terms, _g_diff, g_la, g_lb, _g_bp):
        from . import kernels as K
        s, adj, x, diff, weight = ctx.saved_tensors
        gs, gx = _pool_small_backward(ctx, s, adj, x, diff, g_x, g_raw, g_adj, None, None, g_la, g_lb)
        if g_s is not None:  # S was used outside the pooler as well
            gs = gs + g_s
        need = ctx.needs_input_grad
        gx, gw, gb = K.mlp_select_bwd(s, gs, x, weight, want_gx=need[0], want_gw=need[2],
                                      want_gb=ctx.has_bias and need[3], gx_accumulate=gx if need[0] else None)
        return ((gx.to(x.dtype) if need[0] else None), None, gw, (gb if ctx.has_bias else None), None, None, None, None,
                None, None, None)


class _SelectPoolSparseFn(torch.autograd.Function):
    """:class:`_SelectPoolSmallFn` on the batch as PyG hands it over (x [Ntot,F], a row-sorted edge list): the forward is
    ``tgp_dense_pool_select_sparse_f32`` -- the adjacency tiles are built in LDS from the edges, and the zero-padded x and
    the dense adjacency the BACKWARD kernels read leave the same launch as side outputs (no ``to_dense_batch`` /
    ``to_dense_adj`` launches, no zero fill) --, the backward the pooling backward, the selector's backward and the gather
    back to the un-padded rows (``tgp_from_dense_batch_f32``).  MinCut (the losses come out of the kernel); the edge
    weights get no gradient here."""

    @staticmethod
    def forward(ctx, x, weight, bias, edge_index, edge_weight, batch, node_ptr, edge_ptr, num_graphs, max_nodes, flags,
                adj_transpose, want_terms, diff_scales=None, graph_sizes=None):
        from . import kernels as K
        ctx.set_materialize_grads(False)
        got = K.dense_pool_select_sparse(
            x.detach(), edge_index, edge_weight, batch, node_ptr, edge_ptr, num_graphs, max_nodes, weight.detach(),
            None if bias is None else bias.detach(), flags, adj_transpose, want_raw=True,
            mincut_terms=diff_scales is None, want_dense=True, diff_stats=diff_scales is not None)
        s, mask, x_pool, raw, adj_pool, terms, bp, xd, ad = got[:9]
        empty = s.new_empty(0)
        la = lb = diff = empty
        if diff_scales is not None:
            # DiffPool's two losses from the per-graph records the launch left (utils/losses.py:644-658, 476-483):
            # |A - S S^T|^2 = sum A^2 - 2 trace(S^T A S) + |S^T S|^2 -- one tail launch (r6; four behind the adjacency)
            diff = K.diffpool_stats_tail(got[9], diff_scales[0], diff_scales[1])
            la, lb = diff[0], diff[1]
        elif want_terms:
            both = terms.mean(dim=1)
            la, lb = both[0], both[1]
        ctx.save_for_backward(s, ad, xd, weight, batch, node_ptr, diff)
        ctx.flags, ctx.max_nodes = flags, max_nodes
        ctx.want_gx = x.requires_grad
        ctx.diff_scales = diff_scales
        ctx.has_bias = bias is not None
        ctx.mark_non_differentiable(mask, bp, *([] if (want_terms or diff_scales is not None) else [la, lb]))
        return s, mask, x_pool, raw, adj_pool, la, lb, bp

    @staticmethod
    def backward(ctx, g_s, _g_mask, g_x, g_raw, g_adj, g_la, g_lb, _g_bp):
        from . import kernels as K
        s, ad, xd, weight, batch, node_ptr, diff = ctx.saved_tensors
        gs, gxd = _pool_small_backward(ctx, s, ad, xd, diff, g_x, g_raw, g_adj, None, None, g_la, g_lb)
        if g_s is not None:
            gs = gs + g_s
        need = ctx.needs_input_grad
        gxd, gw, gb = K.mlp_select_bwd(s, gs, xd, weight, want_gx=need[0], want_gw=need[1],
                                       want_gb=ctx.has_bias and need[2], gx_accumulate=gxd if need[0] else None)
        gx = K.from_dense_batch(gxd, batch, node_ptr, ctx.max_nodes) if need[0] else None
        return (gx, gw, (gb if ctx.has_bias else None)) + (None,) * 12


def select_pool_sparse(x: Tensor, weight: Tensor, bias: Optional[Tensor], edge_index: Tensor,
                       edge_weight: Optional[Tensor], batch: Tensor, node_ptr: Tensor, edge_ptr: Tensor, num_graphs: int,
                       max_nodes: int, flags: int, adj_transpose: bool, want_terms: bool, diff_scales=None,
                       graph_sizes: Optional[Tensor] = None):
    """(s, mask, x_pool, raw, adj_pool, LossPair or None, pooled batch vector): see :class:`_SelectPoolSparseFn`.
    ``diff_scales`` = (link_scale, ent_scale): the pair holds DiffPool's two losses instead of MinCut's."""
    out = _SelectPoolSparseFn.apply(x, weight, bias, edge_index, edge_weight, batch, node_ptr, edge_ptr, num_graphs,
                                    max_nodes, flags, adj_transpose, want_terms, diff_scales, graph_sizes)
    pair = LossPair((out[5], out[6])) if (want_terms or diff_scales is not None) else None
    return out[0], out[1], out[2], out[3], out[4], pair, out[7]


def select_pool_small(x: Tensor, adj: Tensor, weight: Tensor, bias: Optional[Tensor], mask: Optional[Tensor], flags: int,
                      want_raw: bool, want_terms: bool, diff_scales=None, graph_sizes: Optional[Tensor] = None,
                      want_batch: bool = False):
    """(s, x_pool, raw, adj_pool, LossPair or None, pooled batch vector or None): see :class:`_SelectPoolSmallFn`."""
    out = _SelectPoolSmallFn.apply(x, adj, weight, bias, mask, flags, want_raw, want_terms, diff_scales, graph_sizes,
                                   want_batch)
    pair = LossPair((out[6], out[7])) if (diff_scales is not None or want_terms) else None
    return out[0], out[1], out[2], out[3], pair, (out[8] if want_batch else None)


# ----------------------------------------- the dense poolers' training step beyond the one-wave kernels (r6)
POOL_LARGE_STATS = {"symmetric": 0, "general": 0}  # backward passes by route (diagnostics / tests)


class _PoolLargeFn(torch.autograd.Function):
    """Select (optional) + Reduce + Connect + post-processing + the pooler's two auxiliary losses of a padded batch whose
    graphs are too large for the one-wave / one-workgroup kernels (C2: 32 x 1024 nodes, K = 128), as ONE autograd node
    (reference poolers/mincut.py:220-237, diffpool.py:208-218 under ATen autograd; harness
    examples/time_and_mem_test.py:396-401).  The operator-by-operator graph ran 65-71 launches per step there.

    forward  (mode 1 = MinCut, 2 = DiffPool, 0 = no losses):
        [S = softmax(X W^T + b) mask]            tgp_mlp_select_f32            (selector form only)
        U = A S -> acat[:, :, :K]; S^T [U | X | S] -> raw, x_pool, gram; adj_pool     tgp_dense_pool_train_fwd_f32 (3 launches)
        MinCut: deg, q (one pass over A), den + both per-graph terms, batch means    3 launches
        DiffPool: link residual in the GEMM epilogue, entropy partials, tail         4 launches
    backward: with gR = the gradient of raw (post-processing backward + upstream + the loss' diagonal term)
        gS = [U | X | 1000 | S | V] [gR^T ; g_x'^T ; 0 ; RS ; gR]   ONE GEMM over the operand buffer acat [B,N,3K+F+4]
             (V = A^T S is written into its column block by the one N^2 K product of the backward -- not at all when A is
             known to be symmetric (kernels.AdjSymmetry: V = U, the first block's right-hand side becomes gR + gR^T); X, a
             [1 0 0 0] block and S are copied into theirs by one launch; RS = W + W^T (MinCut's orthogonality term) or 2 c G (DiffPool's link term), the
             -c I / -(g_cut / den) I terms sit on the diagonals of the first two blocks: tgp_dense_pool_train_rhs_f32)
        selector form: dY = softmax backward of gS + 2 c1 D S (MinCut) - g_ent (log S + ...) (DiffPool) in one launch,
             written over the V block; gX = [S | dY] [g_x' ; W] (one product over acat's last two blocks),
             [gW | gb] = dY^T [X | 1] (one product over acat's X block and the column of ones behind it).
    The adjacency gets no gradient (callers check); edge_weight_norm is not differentiated here (callers check)."""

    @staticmethod
    def forward(ctx, x, adj, weight, bias, mask, s_given, flags, mode, scales, graph_sizes, sym):
        from . import _native as N
        ctx.set_materialize_grads(False)
        xd = N.f32c(x.detach())
        B, Nn, F = xd.shape
        selector = s_given is None
        if selector:
            s = K.mlp_select(xd, weight.detach(), None if bias is None else bias.detach(), mask)
        else:
            s = N.f32c(s_given.detach())
        Kc = s.size(-1)
        ad = adj.detach()
        mem, tflag = K._dense_adj_layout(ad)
        acat = torch.empty(B, Nn, 3 * Kc + F + K.TRAIN_PAD, dtype=torch.float32, device=xd.device)
        x_pool, raw, adj_pool, gram = K.dense_pool_train_fwd(s, mem, xd, flags | tflag, acat, want_gram=mode != 0)
        empty = s.new_empty(0)
        la, lb = s.new_empty(0), s.new_empty(0)  # (distinct objects: both are outputs of this node)
        deg = den = lossv = stats = None
        if mode == 1:
            deg, q = K.cut_rows(ad, s, graph_sizes)
            den, terms, stats, both = K.mincut_terms_fused(raw, gram, deg, q, want_means=True)
            la, lb = both[0], both[1]
        elif mode == 2:
            # DiffPool's link loss from what the step already holds (r6, late): |A - S S^T|^2 = sum A^2 - 2 sum_b
            # trace(raw_b) + sum_b |S_b^T S_b|^2 (utils/losses.py:644-658) -- one streaming pass over A for sum A^2 instead
            # of the residual product S S^T against A (0.076 -> 0.025 ms at C2); rows of padded nodes of S are zero, so
            # the identity holds whatever the padding of A holds
            flat = mem.reshape(-1)
            lossv = K.diffpool_unbatched_tail(raw, gram, s, torch.dot(flat, flat), scales[0], scales[1])
            la, lb = lossv[0], lossv[1]
        keep = [t if t is not None else empty for t in (gram, deg, den, lossv, stats)]
        ctx.save_for_backward(s, mem, xd, empty if weight is None else weight, acat, raw, *keep)
        ctx.flags, ctx.tflag, ctx.mode, ctx.scales, ctx.selector = flags, tflag, mode, scales, selector
        ctx.has_bias = bias is not None
        ctx.sym = sym
        if mode == 0:
            ctx.mark_non_differentiable(la, lb)
        if selector:
            return s, x_pool, raw, adj_pool, la, lb
        no_s = s.new_empty(0)
        ctx.mark_non_differentiable(no_s)
        return no_s, x_pool, raw, adj_pool, la, lb

    @staticmethod
    def backward(ctx, g_s, g_xp, g_raw, g_adj, g_la, g_lb):
        from . import _native as N
        s, mem, xd, weight, acat, raw, gram, deg, den, lossv, stats = ctx.saved_tensors
        B, Nn, Kc = s.shape
        F = xd.size(2)
        mode, selector = ctx.mode, ctx.selector
        dev = s.device
        if mode == 0:
            g_la = g_lb = None
        want_gx = ctx.needs_input_grad[0]
        nothing = (None,) * 11
        if g_s is None and g_xp is None and g_raw is None and g_adj is None and g_la is None and g_lb is None:
            return nothing
        ga = None
        if g_adj is not None:
            ga = K.postprocess_dense_bwd(raw, g_adj, ctx.flags)
            if ga is None:
                raise RuntimeError("dense pooling backward: K > 4096 is not supported by the post-processing backward")
        gb = None if g_raw is None else N.f32c(g_raw)
        gx_t, gx_bc = K._bcast_or_dense(g_xp, (B, Kc, F))
        link_loss = lossv[0:1] if mode == 2 else None
        symmetric = ctx.sym is not None and ctx.sym.get()  # A = A^T: V = U, the second N^2 K product is not needed
        POOL_LARGE_STATS["symmetric" if symmetric else "general"] += 1
        fold_gx = selector and want_gx  # gX = [S | dY] [g_x ; W] as one product over acat's last two blocks
        rcat, c1, gwcat = K.dense_pool_train_rhs(
            ga, gb, mode, stats if mode == 1 else None, den if mode == 1 else None, gram if mode else None, g_la,
            g_lb if mode == 1 else None, 1.0 / B, link_loss, ctx.scales[0] if mode == 2 else 0.0, gx_t, gx_bc, symmetric,
            weight if fold_gx else None, B, Kc, F, dev)
        pad = K.TRAIN_PAD
        ld = 3 * Kc + F + pad
        c_x, c_one, c_s, c_v = Kc, Kc + F, Kc + F + pad, 2 * Kc + F + pad
        vblock = acat[:, :, c_v:]
        if not symmetric:  # V = A^T S into its column block (mem holds A, or A^T when tflag)
            K.bmm_into(mem, s, vblock, trans_a=not ctx.tflag)
        K.copy_cols2(xd.view(B * Nn, F), s.view(B * Nn, Kc), acat.view(B * Nn, ld), c_x, c_s, one_col=c_one)
        kd = c_v if symmetric else ld
        gs = torch.empty(B, Nn, Kc, dtype=torch.float32, device=dev)
        K.bmm_into(acat[:, :, :kd], rcat[:, :kd, :], gs)
        ent_g = g_lb if mode == 2 else None
        if not selector:
            gxd = None
            if want_gx and g_xp is not None:
                gxd = K.bmm(s, g_xp.contiguous() if gx_bc else gx_t)
            if mode == 1 and c1 is not None:
                gs.addcmul_((2.0 * c1).view(-1, 1, 1) * deg.unsqueeze(-1), s)
            if ent_g is not None:
                gs += K.entropy_bwd(s, ent_g, ctx.scales[1])
            return (gxd, None, None, None, None, gs, None, None, None, None, None)
        # the selector's backward: dY over the V block (no longer needed), then two products
        K.softmax_bwd_ex(s, gs, extra=g_s, c1=c1 if mode == 1 else None, deg=deg if mode == 1 else None,
                         ent_g=ent_g, ent_scale=ctx.scales[1] if mode == 2 else 0.0, out=vblock)
        gxd = gw = gbias = None
        if want_gx:
            gxd = torch.empty(B, Nn, F, dtype=torch.float32, device=dev)
            K.bmm_into(acat[:, :, c_s:], gwcat, gxd)
        want_gw, want_gb = ctx.needs_input_grad[2], ctx.has_bias and ctx.needs_input_grad[3]
        if want_gw or want_gb:
            # dY^T [X | 1 0 0 0]: weight and bias gradient from one product over the operand buffer's X block and the
            # column of ones behind it, as a batched product over row slabs + the sum of the slab results (20 us at
            # 32768 x 128 x 64; the slab-wise small product + a column sum of dY took 47 us)
            rows = B * Nn
            slabs = 64
            while slabs > 1 and (rows % slabs or rows // slabs < 256):
                slabs //= 2
            part = torch.empty(slabs, Kc, F + pad, dtype=torch.float32, device=dev)
            flat = acat.view(slabs, rows // slabs, ld)
            K.bmm_into(flat[:, :, c_v:], flat[:, :, c_x:c_x + F + pad], part, trans_a=True)
            gw, gbias = K.slab_sum_split(part, F, want_gw, want_gb)  # two contiguous tensors, slabs added in order
        return (gxd, None, gw, gbias, None, None, None, None, None, None, None)


def _slab_ptr(rows: int, device) -> Tensor:
    """Row ranges of ~equal length (at least 256 rows, at most 64 ranges): the slabs of a weight-gradient product."""
    slabs = max(1, min(64, rows // 256))
    key = ("slab", rows, slabs, str(device))
    if key not in _WHOLE_RANGE:
        if len(_WHOLE_RANGE) > 64:
            _WHOLE_RANGE.clear()
        _WHOLE_RANGE[key] = (torch.arange(slabs + 1, dtype=torch.long, device=device) * rows) // slabs
    return _WHOLE_RANGE[key]


class _PoolUnbatchedFn(torch.autograd.Function):
    """:class:`_PoolLargeFn` for the UNBATCHED mode (S [Ntot,K], sparse A, sorted batch vector; reference
    connect/dense_conn.py:140-208, reduce/base_reduce.py:170-182, utils/losses.py:73-127, 204-240, 661-708 under ATen
    autograd): one autograd node for Select (optional) + Reduce + Connect + post-processing + both losses, no dense
    adjacency anywhere.
    forward: T = A S (CSR SpMM), S^T [T | X | S] per graph (one product grid + a combine), the loss tail; the mincut
    numerator is trace(raw_g), the link residual sum_e w_e^2 - 2 sum_g trace(raw_g) + sum_g |G_g|^2.
    backward: the same K-sized right-hand sides as the padded form (tgp_dense_pool_train_rhs_f32), then
        gS = [T | X | 1000 | S | T'] [gR^T ; g_x'^T ; 0 ; RS ; gR]  as ONE segment product over the operand buffer
    (T' = A^T S: a second SpMM over the column-sorted list -- not at all when the list is symmetric, which one launch
    in the forward finds out: kernels.AdjSymmetry.of_edge_list), softmax backward with the elementwise loss terms folded
    in, gX = [S | dY] [g_x' ; W], [gW | gb] = dY^T [X | 1] over row slabs.  ``ei`` / ``ew``: the coalesced row-sorted
    list; the edge weights get no gradient here (callers check)."""

    @staticmethod
    def forward(ctx, x, weight, bias, s_given, ei, ew, row_ptr, ptr, batch, max_nodes, flags, mode, scales, sw2, sym,
                transposed=False):
        # transposed: the pooled adjacency is S^T A^T S (what the BATCHED poolers compute from a sparse input with
        # adj_transpose=True, src.py:442-443) = the transpose of S^T (A S); MinCut's degrees are then the in-degrees
        from . import _native as N
        ctx.set_materialize_grads(False)
        xd = N.f32c(x.detach())
        n, F = xd.shape
        selector = s_given is None
        one = K.pool_rows_forward(xd, None if weight is None else weight.detach(), None if bias is None else bias.detach(),
                                  None if selector else N.f32c(s_given.detach()), row_ptr, ei, ew, ptr, max_nodes,
                                  transposed, flags, mode, scales, sw2)
        if one is not None:  # the whole forward as ONE native call (r6, late: same launches, no host work between them)
            s, t, raw, x_pool, gram, adj_pool = (one[k] for k in ("s", "t", "raw", "x_pool", "gram", "adj_pool"))
            empty = s.new_empty(0)
            la, lb = s.new_empty(0), s.new_empty(0)
            deg = den = lossv = stats = None
            if mode == 1:
                deg, den, stats, both = one["deg"], one["den"], one["stats"], one["both"]
                la, lb = both[0], both[1]
            elif mode == 2:
                lossv = one["lossv"]
                la, lb = lossv[0], lossv[1]
            return _PoolUnbatchedFn._finish_forward(ctx, s, t, xd, weight, raw, ei, row_ptr, ptr, gram, deg, den, lossv,
                                                    stats, ew, batch, flags, mode, scales, selector, max_nodes, bias, sym,
                                                    transposed, x_pool, adj_pool, la, lb)
        s = K.mlp_select(xd, weight.detach(), None if bias is None else bias.detach(), None) if selector \
            else N.f32c(s_given.detach())
        Kc = s.size(1)
        B = ptr.numel() - 1
        deg = q = ent_part = None
        if mode == 1:  # MinCut: out-degrees and |S_i|^2 ride along with T = A S
            t, deg, q = K.spmm_csr(row_ptr, ei, ew, n, s, want_stats=True)
        elif mode == 2:  # DiffPool: the entropy sum over S rides along
            t, ent_part = K.spmm_csr(row_ptr, ei, ew, n, s, want_stats="entropy")
        else:
            t = K.spmm_csr(row_ptr, ei, ew, n, s)
        raw, x_pool, gram, adj_pool = K.segment_gemm_tn3(s, [t, xd, s], ptr, max_nodes, transpose0=transposed,
                                                         post_flags=flags)
        empty = s.new_empty(0)
        la, lb = s.new_empty(0), s.new_empty(0)
        den = lossv = stats = None
        if mode == 1:
            if transposed:  # den = sum_j indeg_j q_j = sum_e w_e q[col_e]: the tail walks the graph's entries itself
                den, terms, stats, both = K.mincut_terms_fused(raw, gram, None, q, ptr=ptr, want_means=True,
                                                               edges=(row_ptr, ei, ew))
            else:
                den, terms, stats, both = K.mincut_terms_fused(raw, gram, deg, q, ptr=ptr, want_means=True)
            la, lb = both[0], both[1]
        elif mode == 2:
            lossv = K.diffpool_unbatched_tail(raw, gram, s, sw2, scales[0], scales[1], ent_partials=ent_part)
            la, lb = lossv[0], lossv[1]
        return _PoolUnbatchedFn._finish_forward(ctx, s, t, xd, weight, raw, ei, row_ptr, ptr, gram, deg, den, lossv, stats,
                                                ew, batch, flags, mode, scales, selector, max_nodes, bias, sym, transposed,
                                                x_pool, adj_pool, la, lb)

    @staticmethod
    def _finish_forward(ctx, s, t, xd, weight, raw, ei, row_ptr, ptr, gram, deg, den, lossv, stats, ew, batch, flags, mode,
                        scales, selector, max_nodes, bias, sym, transposed, x_pool, adj_pool, la, lb):
        empty = s.new_empty(0)
        keep = [v if v is not None else empty for v in (gram, deg, den, lossv, stats, ew, batch)]
        ctx.save_for_backward(s, t, xd, empty if weight is None else weight, raw, ei, row_ptr, ptr, *keep)
        ctx.flags, ctx.mode, ctx.scales, ctx.selector, ctx.max_nodes = flags, mode, scales, selector, max_nodes
        ctx.has_bias, ctx.sym, ctx.has_w, ctx.has_batch = bias is not None, sym, ew is not None, batch is not None
        ctx.transposed = transposed
        if mode == 0:
            ctx.mark_non_differentiable(la, lb)
        if selector:
            return s, x_pool, raw, adj_pool, la, lb
        no_s = s.new_empty(0)
        ctx.mark_non_differentiable(no_s)
        return no_s, x_pool, raw, adj_pool, la, lb

    @staticmethod
    def backward(ctx, g_s, g_xp, g_raw, g_adj, g_la, g_lb):
        from . import _native as N
        s, t, xd, weight, raw, ei, row_ptr, ptr, gram, deg, den, lossv, stats, ew, batch = ctx.saved_tensors
        ew = ew if ctx.has_w else None
        batch = batch if ctx.has_batch else None
        n, Kc = s.shape
        F = xd.size(1)
        B = ptr.numel() - 1
        mode, selector = ctx.mode, ctx.selector
        dev = s.device
        if mode == 0:
            g_la = g_lb = None
        nothing = (None,) * 16
        if g_s is None and g_xp is None and g_raw is None and g_adj is None and g_la is None and g_lb is None:
            return nothing
        want_gx = ctx.needs_input_grad[0]
        symmetric = ctx.sym is not None and ctx.sym.get()
        POOL_LARGE_STATS["symmetric" if symmetric else "general"] += 1
        if selector and symmetric and weight.numel() and Kc <= 4096:
            # the common case as ONE native call (r6, late): the same launches as below, no host work between them
            want_gw, want_gb = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
            one = K.pool_rows_backward(s, t, xd, N.f32c(weight.detach()), raw, gram if mode else None,
                                       stats if mode == 1 else None, den if mode == 1 else None, deg if mode == 1 else None,
                                       lossv[0:1] if mode == 2 else None, ptr, batch, _slab_ptr(n, dev), ctx.max_nodes,
                                       ctx.flags, mode, ctx.transposed, ctx.scales, g_adj,
                                       None if g_raw is None else N.f32c(g_raw), g_xp, g_s, g_la, g_lb, want_gx, want_gw,
                                       want_gb)
            if one is not None:
                return (one[0], one[1], one[2]) + (None,) * 13
        ga = None
        if g_adj is not None:
            ga = K.postprocess_dense_bwd(raw, g_adj, ctx.flags)
            if ga is None:
                raise RuntimeError("dense pooling backward: K > 4096 is not supported by the post-processing backward")
        gb = None if g_raw is None else N.f32c(g_raw)
        gx_t, gx_bc = K._bcast_or_dense(g_xp, (B, Kc, F))
        fold_gx = selector and want_gx
        # the operand buffer's first block holds T = A S.  With raw = S^T A S that is "U" (right-hand side gR^T) and
        # T' = A^T S is "V" (gR); for the transposed form raw = S^T A^T S the two trade places (flag bit 1).
        sym_flags = 1 if symmetric else (2 if ctx.transposed else 0)
        rcat, c1, gwcat = K.dense_pool_train_rhs(
            ga, gb, mode, stats if mode == 1 else None, den if mode == 1 else None, gram if mode else None, g_la,
            g_lb if mode == 1 else None, 1.0 / B, lossv[0:1] if mode == 2 else None,
            ctx.scales[0] if mode == 2 else 0.0, gx_t, gx_bc, sym_flags, weight if fold_gx else None, B, Kc, F, dev)
        pad = K.TRAIN_PAD
        ld = 3 * Kc + F + pad
        c_x, c_one, c_s, c_v = Kc, Kc + F, Kc + F + pad, 2 * Kc + F + pad
        acat = torch.empty(n, ld, dtype=torch.float32, device=dev)
        K.copy_cols3(t, xd, s, acat, 0, c_x, c_s, one_col=c_one)  # [T | X | 1 0 0 0 | S | .]: one launch
        vblock = acat[:, c_v:]
        if not symmetric:  # T' = A^T S: the SpMM over the column-sorted list (the list is coalesced: nothing merges)
            ident = torch.arange(n, device=dev)
            w1 = ew if ew is not None else torch.ones(ei.size(1), device=dev)
            ei_t, w_t = K.coalesce_edges(ei.flip(0), w1, ident, n, "sum", remove_self_loops=False, eps_filter=False)
            rp_t = K.csr_offsets(ei_t, n)
            K.copy_cols2(K.spmm_csr(rp_t, ei_t, w_t, n, s), s.new_empty(n, 0), acat, c_v, c_v)
            if ctx.transposed and mode == 1:  # the in-degrees: row sums of the transposed list
                deg = K.edge_row_stats(rp_t, w_t, s)[0]
        kd = c_v if symmetric else ld
        gs = torch.empty(n, Kc, dtype=torch.float32, device=dev)
        K.segment_gemm_nn_into(acat[:, :kd], rcat[:, :kd, :], ptr, gs, ctx.max_nodes)
        ent_g = g_lb if mode == 2 else None
        if not selector:
            gxd = None
            if want_gx and g_xp is not None:
                gxd = K.segment_gemm_nn(s, g_xp.contiguous() if gx_bc else gx_t, ptr, ctx.max_nodes)
            if mode == 1 and c1 is not None:
                rowc = 2.0 * (c1[batch] if batch is not None else c1) * deg
                gs.addcmul_(rowc.unsqueeze(-1), s)
            if ent_g is not None:
                gs += K.entropy_bwd(s, ent_g, ctx.scales[1])
            return (gxd, None, None, gs) + (None,) * 12
        K.softmax_bwd_ex(s, gs, extra=g_s, c1=c1 if mode == 1 else None, deg=deg if mode == 1 else None, ent_g=ent_g,
                         ent_scale=ctx.scales[1] if mode == 2 else 0.0, out=vblock, batch=batch)
        gxd = gw = gbias = None
        if want_gx:
            gxd = torch.empty(n, F, dtype=torch.float32, device=dev)
            K.segment_gemm_nn_into(acat[:, c_s:], gwcat, ptr, gxd, ctx.max_nodes)
        want_gw, want_gb = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        if want_gw or want_gb:
            part = K.segment_gemm_tn_into(acat[:, c_v:], acat[:, c_x:c_x + F + pad], _slab_ptr(n, dev))
            gw, gbias = K.slab_sum_split(part, F, want_gw, want_gb)
        return (gxd, gw, gbias) + (None,) * 13


def pool_unbatched(x: Tensor, weight: Optional[Tensor], bias: Optional[Tensor], s: Optional[Tensor], ei: Tensor,
                   ew: Optional[Tensor], row_ptr: Tensor, ptr: Tensor, batch: Optional[Tensor], max_nodes: int, flags: int,
                   mode: int, scales=(0.0, 0.0), sw2=0.0, symmetry=None, transposed: bool = False):
    """(s, x_pool [B,K,F], raw, adj_pool, LossPair or None): see :class:`_PoolUnbatchedFn`."""
    out = _PoolUnbatchedFn.apply(x, weight, bias, s, ei, ew, row_ptr, ptr, batch, max_nodes, flags, mode, tuple(scales), sw2,
                                 symmetry, bool(transposed))
    pair = LossPair((out[4], out[5])) if mode else None
    return (out[0] if s is None else s), out[1], out[2], out[3], pair


def pool_large(x: Tensor, adj: Tensor, weight: Optional[Tensor], bias: Optional[Tensor], mask: Optional[Tensor],
               s: Optional[Tensor], flags: int, mode: int, scales=(0.0, 0.0), graph_sizes: Optional[Tensor] = None,
               symmetry=None):
    """(s, x_pool, raw, adj_pool, LossPair or None): see :class:`_PoolLargeFn`.  Give either the selector's single Linear
    (``weight`` [K,F], ``bias``, ``mask``: S is formed inside and returned) or ``s`` itself.  ``symmetry``: a
    :class:`kernels.AdjSymmetry` (or anything with ``get() -> bool``) when the caller can tell whether A = A^T."""
    out = _PoolLargeFn.apply(x, adj, weight, bias, mask, s, flags, mode, tuple(scales), graph_sizes, symmetry)
    pair = LossPair((out[4], out[5])) if mode else None
    return (out[0] if s is None else s), out[1], out[2], out[3], pair


class ASProducts:
    """U = A S and V = A^T S of one (S, A) pair, computed at most once.  DiffPool's training step needs U in the
    Connect forward, U and V in its backward and both again in the link-prediction loss' backward
    (poolers/diffpool.py:208-218): five N^2 K products where two suffice.  The object is found again in the forward
    by the identity of the two live tensors and carried into the backward by the autograd contexts."""

    __slots__ = ("s_ref", "adj_ref", "versions", "u", "v", "__weakref__")

    def __init__(self, s: Tensor, adj: Tensor):
        import weakref
        self.s_ref, self.adj_ref = weakref.ref(s), weakref.ref(adj)
        self.versions = (s._version, adj._version)
        self.u = self.v = None

    def matches(self, s: Tensor, adj: Tensor) -> bool:
        return self.s_ref() is s and self.adj_ref() is adj and self.versions == (s._version, adj._version)

    @staticmethod
    def _product(s: Tensor, adj: Tensor, transposed: bool) -> Tensor:
        # adj may be the transposed view of contiguous memory (src.py:442-443); float64 operands stay float64
        mem, tflag = K._dense_adj_layout(adj, torch.float64 if K._any_f64(s, adj) else torch.float32)
        return K.bmm(mem, s, trans_a=bool(tflag) != transposed)

    def get_u(self, s: Tensor, adj: Tensor) -> Tensor:
        if self.u is None:
            self.u = self._product(s.detach(), adj.detach(), False)
        return self.u

    def get_v(self, s: Tensor, adj: Tensor) -> Tensor:
        if self.v is None:
            self.v = self._product(s.detach(), adj.detach(), True)
        return self.v


import weakref as _weakref

_PRODUCTS: "_weakref.WeakValueDictionary" = _weakref.WeakValueDictionary()


def shared_products(s: Tensor, adj: Tensor) -> ASProducts:
    key = (id(s), id(adj))
    hit = _PRODUCTS.get(key)
    if hit is not None and hit.matches(s, adj):
        return hit
    fresh = ASProducts(s, adj)
    _PRODUCTS[key] = fresh
    return fresh
