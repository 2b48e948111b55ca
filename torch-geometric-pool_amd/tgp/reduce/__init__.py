"""``reduce`` operators: pooled features X' = S^T X  (reference tgp/reduce/base_reduce.py)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor, nn

from .. import functions as Fn
from .. import kernels as K
from ..select import SelectOutput
from ..utils.ops import (as_compute_dtype, build_pooled_batch, graph_ptr, is_multi_graph_batch, like_input_dtype,
                         max_graph_size, num_graphs_of)


class Reduce(nn.Module):
    """Template of the reduce operator."""

    @staticmethod
    def reduce_batch(select_output: SelectOutput, batch: Optional[Tensor]) -> Optional[Tensor]:
        """Batch vector of the pooled graph (reference base_reduce.py:14-53)."""
        if batch is None:
            return None
        if select_output.s.is_sparse:
            return K.reduce_batch_sparse(batch, select_output.node_index, select_output.cluster_index,
                                         select_output.num_supernodes,
                                         every_cluster_has_a_node=bool(select_output.__dict__.get("_no_empty_cluster")))
        if batch.numel() == 0:
            return batch.new_empty((0,), dtype=batch.dtype)
        return build_pooled_batch(num_graphs_of(batch), select_output.num_supernodes, batch.device,
                                  dtype=batch.dtype)

    def reset_parameters(self):
        pass

    def forward(self, x: Tensor, so: SelectOutput, *, batch: Optional[Tensor] = None,
                **kwargs) -> Tuple[Tensor, Optional[Tensor]]:
        raise NotImplementedError

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}()"


class BaseReduce(Reduce):
    r"""X' = S^T X.  Sparse S: segmented gather-sum kernel; dense [B,N,K]: fp32-MFMA batched GEMM;
    dense [N,K] with a batch vector: one segment-GEMM launch instead of the reference's Python loop
    over graphs (reference base_reduce.py:108-190)."""

    def forward(self, x: Tensor, so: SelectOutput, *, batch: Optional[Tensor] = None,
                return_batched: bool = False, **kwargs) -> Tuple[Tensor, Optional[Tensor]]:
        if so.s.is_sparse and (x.dtype == torch.float64 or so.s.dtype == torch.float64) and x.is_cuda:
            # float64 features / assignment weights: the sparse Reduce runs in fp64 like the reference's scatter
            # (base_reduce.py:146-153; r4) -- the result has the promoted dtype, no fp32 narrowing, no warning
            if return_batched:
                raise ValueError("return_batched=True is only supported for dense assignment matrices.")
            if batch is None and so.batch is not None:
                batch = so.batch
            return _SparseReduceFn.apply(x, so.weight, so), self.reduce_batch(so, batch)
        # fp32 arithmetic (the GEMM paths; sparse fp32 / half inputs); the result carries the dtype of x like the
        # reference's ATen ops would
        x_pool, batch_pool = self._forward_f32(as_compute_dtype(x), so, batch=batch, return_batched=return_batched)
        return like_input_dtype(x_pool, x), batch_pool

    def _forward_f32(self, x: Tensor, so: SelectOutput, *, batch: Optional[Tensor] = None,
                     return_batched: bool = False) -> Tuple[Tensor, Optional[Tensor]]:
        if batch is None and so.batch is not None:
            batch = so.batch
        if so.s.is_sparse:
            if return_batched:
                raise ValueError("return_batched=True is only supported for dense assignment matrices.")
            x_pool = _SparseReduceFn.apply(x, as_compute_dtype(so.weight), so)
            return x_pool, self.reduce_batch(so, batch)
        s = as_compute_dtype(so.s)
        if s.dim() == 3:
            return _DenseReduceFn.apply(s, x, getattr(so, "_graph_sizes", None)), self.reduce_batch(so, batch)
        if s.dim() != 2:
            raise ValueError(f"Dense SelectOutput.s must be 2D [N, K] or 3D [B, N, K], got ndim={s.dim()}.")
        if is_multi_graph_batch(batch):
            sizes, ptr = graph_ptr(batch)
            x_pool = Fn.segment_gemm_tn(s, x, ptr, max_graph_size(batch))  # [B,K,F]
            if not return_batched:
                x_pool = x_pool.reshape(-1, x_pool.size(-1))
            return x_pool, self.reduce_batch(so, batch)
        x_pool = Fn.bmm(s, x, trans_a=True)  # [K,F]
        if return_batched:
            x_pool = x_pool.unsqueeze(0)
        return x_pool, self.reduce_batch(so, batch)


class _SparseReduceFn(torch.autograd.Function):
    """x_pool = S^T X for sparse S; backward = the same kernel with node/cluster roles swapped
    (dX = S dX') plus a row-dot for the assignment weights (TopK trains its scores through them)."""

    @staticmethod
    def forward(ctx, x, weight, so, computed=None):
        # ``computed``: [x_pool] when the caller already has it -- SRCPooling.reduce_connect's one launch for Reduce +
        # Connect, which computes the same sums in the same order; this node then only supplies the backward
        if computed is not None:
            out = computed[0]
        else:
            # a clustering whose S has row index 0..N-1 and unit values (GraclusSelect; a cluster vector): known, not probed
            out = K.reduce_sparse(x, so.node_index, weight, so.assign_index(),
                                  identity_source=bool(so.__dict__.get("_identity_nodes", False)),
                                  unit_weight=bool(so.__dict__.get("_unit_values", False)))
        ctx.so = so
        ctx.save_for_backward(x, weight)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, weight = ctx.saved_tensors
        so = ctx.so
        gx = gw = None
        grad_out = grad_out.contiguous()
        if ctx.needs_input_grad[0]:
            from ..lift import lift_index_of
            gx = K.reduce_sparse(grad_out, so.cluster_index, weight, lift_index_of(so))
        if ctx.needs_input_grad[1]:
            if x.dim() == 2 and x.is_cuda and x.dtype == torch.float32 and grad_out.dtype == torch.float32:
                gw = K.pair_dot(x, so.node_index, grad_out, so.cluster_index)
            else:
                gw = (x[so.node_index] * grad_out[so.cluster_index]).reshape(so.node_index.numel(), -1).sum(-1)
        return gx, gw, None, None


class _DenseReduceFn(torch.autograd.Function):
    """X' = S^T X on the matrix cores; dS = X dX'^T, dX = S dX'."""

    @staticmethod
    def forward(ctx, s, x, graph_sizes=None):
        ctx.save_for_backward(s, x)
        return K.dense_pool(s, None, x, graph_sizes=graph_sizes)[0]

    @staticmethod
    def backward(ctx, g):
        s, x = ctx.saved_tensors
        gs = gx = None
        g = g.contiguous()
        if ctx.needs_input_grad[0]:
            gs = K.bmm(x, g.transpose(-1, -2).contiguous())
        if ctx.needs_input_grad[1]:
            gx = K.bmm(s, g)
        return gs, gx, None


__all__ = ["Reduce", "BaseReduce"]
