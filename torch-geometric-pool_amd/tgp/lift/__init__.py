"""``lift`` operators: un-pooling X_lift = S_inv^T X_pool (reference tgp/lift/base_lift.py)."""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor, nn

from .. import functions as Fn
from .. import kernels as K
from ..select import SelectOutput
from ..utils.ops import (as_compute_dtype, like_input_dtype, build_pooled_batch, expand_compacted_rows, graph_ptr, is_multi_graph_batch, max_graph_size,
                         num_graphs_of, pseudo_inverse)


def lift_index_of(so: SelectOutput):
    """node -> assignments inverted index (the transpose of SelectOutput.assign_index), cached."""
    if so._lift_index is None:
        if so.__dict__.get("_identity_nodes") and so.s.is_cuda and so.s._nnz() == so.num_nodes:
            # node i owns assignment i (one-over-K selectors whose node_index is 0..N-1): no table to build
            so._lift_index = K.AssignIndex(None, None, so.num_nodes, so.num_nodes, device=so.s.device)
        else:
            so._lift_index = K.build_assign_index(so.node_index, so.num_nodes)
    return so._lift_index


class Lift(nn.Module):
    def reset_parameters(self):
        pass

    def forward(self, x_pool: Tensor, so: SelectOutput, **kwargs) -> Tensor:
        raise NotImplementedError

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}()"


class BaseLift(Lift):
    r"""X_lift = M X_pool with M = S (``transpose``), S_inv^T (``precomputed``) or pinv(S)^T
    (``inverse``)  (reference lift/base_lift.py:39-254).  Sparse M: the segmented gather-sum kernel of
    the sparse Reduce with node / supernode roles swapped; dense M: fp32-MFMA GEMMs."""

    def __init__(self, matrix_op: str = "precomputed", reduce_op: str = "sum"):
        super().__init__()
        self.matrix_op = matrix_op
        self.reduce_op = reduce_op

    def _get_lift_matrix(self, so: SelectOutput) -> Tensor:
        if self.matrix_op == "transpose":
            return so.s
        if self.matrix_op == "precomputed":
            if getattr(so, "s_inv_is_transpose_of_s", False):
                return so.s  # S_inv is S^T: its transpose is S itself (already coalesced, node-sorted)
            matrix = so.s_inv
        elif self.matrix_op == "inverse":
            matrix = pseudo_inverse(so.s)
        else:
            raise RuntimeError("'matrix_op' must be one of ['transpose', 'inverse', 'precomputed'] "
                               f"({self.matrix_op} given)")
        matrix = matrix.transpose(-2, -1)
        return matrix.coalesce() if matrix.is_sparse else matrix

    def _lift_sparse(self, lift_matrix: Tensor, x_pool: Tensor, so: SelectOutput) -> Tensor:
        if self.reduce_op not in ("sum", "add"):
            raise NotImplementedError(f"BaseLift(reduce_op='{self.reduce_op}') is not available in the MI355X "
                                      "build; only 'sum' is (the default of every pooler in scope).")
        row, col = lift_matrix.indices()
        same = lift_matrix.size() == so.s.size() and row.numel() == so.node_index.numel() and (
            row.data_ptr() == so.node_index.data_ptr() or torch.equal(row, so.node_index))
        index = lift_index_of(so) if same else K.build_assign_index(row, lift_matrix.size(0))
        back = so.assign_index if same else (lambda: K.build_assign_index(col, lift_matrix.size(1)))
        return Fn.sparse_lift(x_pool, as_compute_dtype(lift_matrix.values()), row, col, index, back)

    @staticmethod
    def _lift_dense_multi_graph(lift_matrix, x_pool_flat, batch, batch_pooled) -> Tensor:
        """Per-graph M_b X_pool_b for an un-padded batch: rows of graph b use the K pooled rows of b."""
        nb = num_graphs_of(batch)
        k = lift_matrix.size(-1)
        counts = torch.bincount(batch_pooled, minlength=nb)
        if counts.numel() != nb or not bool((counts == k).all()):
            raise ValueError("Inconsistent per-graph blocks while lifting dense [N, K] assignments: "
                             f"got {nb} assignment blocks and pooled blocks of sizes {counts.tolist()}.")
        xp = x_pool_flat.view(nb, k, -1)
        sizes, ptr = graph_ptr(batch, nb)
        # one launch over all graphs (the reference loops over them, base_lift.py:205-215)
        return Fn.segment_gemm_nn(lift_matrix, xp, ptr, max_graph_size(batch))

    def forward(self, x_pool: Tensor, so: SelectOutput = None, batch: Optional[Tensor] = None,
                batch_pooled: Optional[Tensor] = None, **kwargs) -> Tensor:
        # fp32 arithmetic; the result carries the dtype of x_pool like the reference's ATen ops would
        out = self._forward_f32(as_compute_dtype(x_pool), so, batch, batch_pooled)
        return like_input_dtype(out, x_pool)

    def _forward_f32(self, x_pool: Tensor, so: SelectOutput, batch: Optional[Tensor],
                     batch_pooled: Optional[Tensor]) -> Tensor:
        if batch is None and so.batch is not None:
            batch = so.batch
        m = self._get_lift_matrix(so)
        if m.is_sparse:
            return self._lift_sparse(m, x_pool, so)
        m = as_compute_dtype(m)
        k = m.size(-1)
        multi = is_multi_graph_batch(batch)
        if m.dim() == 2 and x_pool.dim() == 2 and multi:
            nb = num_graphs_of(batch)
            if x_pool.size(0) == k:
                return Fn.bmm(m, x_pool)
            if x_pool.size(0) != nb * k:
                raise ValueError("Unexpected pooled feature shape for dense [N, K] lifting with a multi-graph "
                                 f"batch: got x_pool.size(0)={x_pool.size(0)}, expected {k} or {nb * k}.")
            if batch_pooled is None:
                batch_pooled = build_pooled_batch(nb, k, x_pool.device)
            elif batch_pooled.size(0) != x_pool.size(0):
                raise ValueError("batch_pooled has an unexpected length for dense [N, K] lifting "
                                 f"(got {batch_pooled.size(0)}, expected {x_pool.size(0)}).")
            return self._lift_dense_multi_graph(m, x_pool, batch, batch_pooled)
        if m.dim() == 2 and x_pool.dim() == 3:
            if not multi:
                return Fn.bmm(m, x_pool.squeeze(0))
            nb = x_pool.size(0)
            flat = x_pool.reshape(nb * k, x_pool.size(-1))
            if batch_pooled is None:
                batch_pooled = build_pooled_batch(nb, k, x_pool.device)
            elif batch_pooled.size(0) != nb * k:
                raise ValueError("batch_pooled has an unexpected length for dense [N, K] lifting "
                                 f"(got {batch_pooled.size(0)}, expected {nb * k}).")
            return self._lift_dense_multi_graph(m, flat, batch, batch_pooled)
        if m.dim() == 3 and x_pool.dim() == 2:
            nb = m.size(0)
            if x_pool.size(0) != nb * k:
                x_pool = expand_compacted_rows(x_pool, so.out_mask, nb * k)
            x_pool = x_pool.view(nb, k, x_pool.size(-1))
        return Fn.bmm(m, x_pool)

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(matrix_op={self.matrix_op}, reduce_op={self.reduce_op})"


__all__ = ["Lift", "BaseLift"]
