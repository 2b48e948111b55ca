"""``connect`` operators: pooled connectivity A' (reference tgp/connect/{base,dense,kron}_conn.py)."""
from __future__ import annotations

import warnings
from typing import Optional, Tuple

import torch
from torch import Tensor

from .. import functions as Fn
from .. import kernels as K
from ..imports import is_sparsetensor
from ..select import SelectOutput
from ..utils.ops import (
    as_compute_dtype, like_input_dtype,
    _normalize_pooled_edges,
    connectivity_to_edge_index,
    connectivity_to_sparsetensor,
    connectivity_to_torch_coo,
    batch_info,
    graph_ptr,
    max_graph_size,
    num_graphs_of,
    is_dense_adj,
    maybe_num_nodes,
    postprocess_adj_pool_dense,
    postprocess_adj_pool_sparse,
)


class Connect(torch.nn.Module):
    """Template of the connect operator."""

    def reset_parameters(self):
        pass

    def forward(self, edge_index, so: SelectOutput, *, edge_weight: Optional[Tensor] = None, **kwargs):
        raise NotImplementedError

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}()"


def _restore_format(template, edge_index, edge_weight, num_supernodes):
    """Output adjacency format mirrors the input format (reference base_conn.py:71-76,103-110)."""
    if is_sparsetensor(template):
        return connectivity_to_sparsetensor(edge_index, edge_weight, num_supernodes), None
    if isinstance(template, Tensor) and template.is_sparse:
        return connectivity_to_torch_coo(edge_index, edge_weight, num_supernodes), None
    return edge_index, edge_weight


def sparse_connect(edge_index, edge_weight: Optional[Tensor] = None, node_index: Tensor = None,
                   cluster_index: Optional[Tensor] = None, num_nodes: int = None, num_supernodes: int = None,
                   remove_self_loops: bool = True, reduce_op: str = "sum", edge_weight_norm: bool = False,
                   batch_pooled: Optional[Tensor] = None, degree_norm: bool = False, assign_index=None,
                   edge_csr: Optional[Tensor] = None, member_directory: Optional[Tensor] = None):
    r"""Coarsen an edge list (reference connect/base_conn.py:57-112).

    * kept-node selection (TopK): induced subgraph, endpoints relabelled to their position in the
      ascending ``node_index``; input edge order is kept.
    * one-over-K clustering (Graclus, ...): endpoints mapped through ``cluster_index``, duplicates
      merged with ``reduce_op``; output is row-major sorted and unique.

    Both run as one count->fill kernel pair with ``remove_self_loops`` and the ``|w| > eps`` filter of
    ``postprocess_adj_pool_sparse`` fused into the compaction; the degree / max-abs normalisations
    follow as in-place kernels on the (much smaller) pooled list.
    """
    template = edge_index
    edge_index, edge_weight = connectivity_to_edge_index(edge_index, edge_weight)
    if edge_weight is not None:
        edge_weight = edge_weight.view(-1)
    num_nodes = maybe_num_nodes(edge_index, num_nodes)
    # Fn.*: the native kernels, with the pooled weights kept differentiable w.r.t. edge_weight when it needs a gradient
    if node_index is not None and len(node_index) < num_nodes:
        ei, ew = Fn.filter_edges(edge_index, edge_weight, node_index, num_nodes, remove_self_loops,
                                 member_directory=member_directory)
    elif cluster_index is not None and len(cluster_index) == num_nodes:
        ei, ew = Fn.coalesce_edges(edge_index, edge_weight, cluster_index, num_supernodes, reduce_op,
                                   remove_self_loops, assign_index=assign_index,
                                   csr=None if edge_csr is None else (edge_csr, None))
    else:
        raise RuntimeError
    ei, ew = _normalize_pooled_edges(ei, ew, num_supernodes, degree_norm, edge_weight_norm, batch_pooled)
    return _restore_format(template, ei, ew, num_supernodes)


class SparseConnect(Connect):
    r"""Connect for sparse poolers where a node belongs to at most one supernode
    (reference connect/base_conn.py:115-224)."""

    def __init__(self, reduce_op: str = "sum", remove_self_loops: bool = True, edge_weight_norm: bool = False,
                 degree_norm: bool = False):
        super().__init__()
        self.reduce_op = reduce_op
        self.remove_self_loops = remove_self_loops
        self.edge_weight_norm = edge_weight_norm
        self.degree_norm = degree_norm

    def forward(self, edge_index, so: SelectOutput, *, edge_weight: Optional[Tensor] = None,
                batch_pooled: Optional[Tensor] = None, **kwargs):
        if self.edge_weight_norm and batch_pooled is None:
            raise AssertionError("edge_weight_norm=True but batch_pooled=None. batch_pooled parameter is "
                                 "required for per-graph normalization in SparseConnect.")
        # every node assigned (one-over-K poolers): hand over the supernode->member index (cached on `so`,
        # shared with Reduce) so the sort-free row-local coalesce can be used
        ni = so.node_index  # None for a dense assignment: sparse_connect then raises the reference's RuntimeError
        all_assigned = ni is not None and ni.numel() == so.num_nodes and ni.is_cuda
        # fp32 arithmetic for fp32 / half weights (pooled weights carry the dtype of the input weights); float64 weights
        # stay float64 through the edge-list kernels (r4: the reference's coalesce / scatter keep them in fp64)
        w32 = edge_weight if (isinstance(edge_weight, Tensor) and edge_weight.dtype == torch.float64
                              and edge_weight.is_cuda) else as_compute_dtype(edge_weight)
        adj_pool, w_pool = sparse_connect(edge_index, w32, node_index=so.node_index, cluster_index=so.cluster_index,
                                          num_nodes=so.num_nodes, num_supernodes=so.num_supernodes,
                                          remove_self_loops=self.remove_self_loops, reduce_op=self.reduce_op,
                                          edge_weight_norm=self.edge_weight_norm, batch_pooled=batch_pooled,
                                          degree_norm=self.degree_norm,
                                          assign_index=so.assign_index() if all_assigned else None,
                                          edge_csr=so.edge_csr_for(edge_index) if all_assigned else None,
                                          # (TopkSelect on large graphs: the kept-node bitmap + rank directory its
                                          #  compaction pass wrote -- the subgraph Connect is then ONE launch)
                                          member_directory=K.member_directory_for(so.__dict__.get("_assign_index"),
                                                                                  so.node_index))
        return adj_pool, like_input_dtype(w_pool, edge_weight)

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}(reduce_op={self.reduce_op}, "
                f"remove_self_loops={self.remove_self_loops}, edge_weight_norm={self.edge_weight_norm}, "
                f"degree_norm={self.degree_norm})")


class _DenseConnectFn(torch.autograd.Function):
    """R = S^T A S on the matrix cores.  dS = A S dR^T + A^T S dR, dA = S dR S^T.  For graphs too large for the
    one-workgroup-per-graph kernels U = A S is a tensor of its own here, shared with the backward and with the
    link-prediction loss (functions.ASProducts)."""

    @staticmethod
    def forward(ctx, s, adj, graph_sizes=None):
        ctx.save_for_backward(s, adj)
        ctx.products = Fn.shared_products(s, adj)
        B, n, k = s.shape
        if n <= 512 and k <= 64:  # small / medium graphs: the fused kernel; U is cheap to redo in the backward
            return K.dense_pool(s, adj, None, 0, want_raw=True, want_post=False, graph_sizes=graph_sizes)[1]
        u = ctx.products.get_u(s, adj)
        ptr = Fn._uniform_ptr(B, n, s.device)
        return K.segment_gemm_tn(s.detach().reshape(B * n, k), u.reshape(B * n, k), ptr, n)

    @staticmethod
    def backward(ctx, g):
        s, adj = ctx.saved_tensors
        g = g.contiguous()
        gs = ga = None
        if ctx.needs_input_grad[0]:
            u = ctx.products.get_u(s, adj)   # A S
            v = ctx.products.get_v(s, adj)   # A^T S
            gs = K.bmm(u, g.transpose(-1, -2).contiguous())
            gs = K.bmm(v, g, accumulate_into=gs)  # second term in the GEMM epilogue (no elementwise add launch)
        if ctx.needs_input_grad[1]:
            ga = K.bmm(K.bmm(s, g), s.transpose(-1, -2).contiguous())
        return gs, ga, None


class DenseConnect(Connect):
    r"""A' = S^T A S for dense assignments (reference connect/dense_conn.py:22-364).

    Batched dense inputs ([B,N,N], [B,N,K]) run as fp32-MFMA GEMMs with the post-processing fused into
    the split-K combine.  Unbatched inputs (sparse A, S [N,K]) run as one CSR SpMM + one segment GEMM
    over all graphs - no densification and no per-graph Python loop.
    """

    def __init__(self, remove_self_loops: bool = True, degree_norm: bool = True, adj_transpose: bool = True,
                 edge_weight_norm: bool = False, sparse_output: bool = False):
        super().__init__()
        if not isinstance(sparse_output, bool):
            raise TypeError("sparse_output must be a bool.")
        self.remove_self_loops = remove_self_loops
        self.degree_norm = degree_norm
        self.adj_transpose = adj_transpose
        self.edge_weight_norm = edge_weight_norm
        self.sparse_output = sparse_output

    # -- validation helpers (same contracts as the reference) -----------------------------
    @staticmethod
    def _prepare_batched_dense_inputs(s: Tensor, adj: Tensor) -> Tuple[Tensor, Tensor]:
        s = s.unsqueeze(0) if s.dim() == 2 else s
        adj = adj.unsqueeze(0) if adj.dim() == 2 else adj
        if s.dim() != 3 or adj.dim() != 3:
            raise ValueError("Expected batched dense inputs with 3 dimensions.")
        if s.size(0) != adj.size(0):
            raise ValueError("Assignment and adjacency batch sizes do not match: "
                             f"got s.size(0)={s.size(0)} and adj.size(0)={adj.size(0)}.")
        return s, adj

    @staticmethod
    def _validate_select_output(so: SelectOutput) -> Tensor:
        if so is None:
            raise ValueError("SelectOutput is required for DenseConnect.")
        if not isinstance(so.s, Tensor):
            raise TypeError("SelectOutput.s must be a torch.Tensor.")
        if so.s.is_sparse:
            raise ValueError("DenseConnect expects a dense assignment matrix.")
        return so.s

    @staticmethod
    def _dense_connect(s: Tensor, adj: Tensor, graph_sizes: Optional[Tensor] = None) -> Tensor:
        return _DenseConnectFn.apply(s, adj, graph_sizes)

    def dense_connect(self, adj: Tensor, s: Tensor) -> Tensor:
        """Raw S^T A S (MinCut needs it for its loss before post-processing, poolers/mincut.py:226)."""
        s32, adj32 = self._prepare_batched_dense_inputs(as_compute_dtype(s), as_compute_dtype(adj))
        return like_input_dtype(self._dense_connect(s32, adj32), s)

    @staticmethod
    def _dense_connect_unbatched(edge_index, edge_weight, batch, s, num_nodes, num_clusters, batch_size):
        """[B,K,K] = per-graph S_b^T A_b S_b from a sparse A (reference dense_conn.py:140-208)."""
        if batch_size == 1 and not isinstance(edge_index, Tensor) and not is_sparsetensor(edge_index):
            # the reference's single-graph path converts with connectivity_to_torch_coo (dense_conn.py:158-160)
            raise ValueError(f"Edge index must be of type Tensor or SparseTensor, got {type(edge_index)}")
        ei, ew = connectivity_to_edge_index(edge_index, edge_weight)
        if ei.size(1) == 0:
            return s.new_zeros((batch_size, num_clusters, num_clusters))
        # duplicates are SUMMED by the reference's per-graph `.coalesce()`, also for unweighted input
        ew = torch.ones(ei.size(1), device=ei.device) if ew is None else ew.view(-1)
        # sorted + duplicate-summed A (what `.coalesce()` does per graph), then T = A S, then S_b^T T_b
        ei, ew = Fn.coalesce_sum(ei, ew, num_nodes)
        t = Fn.spmm_sorted(ei, ew, num_nodes, s)
        if batch_size == 1:
            return Fn.bmm(s, t, trans_a=True).unsqueeze(0)
        sizes, ptr = graph_ptr(batch, batch_size)
        return Fn.segment_gemm_tn(s, t, ptr, max_graph_size(batch))

    def forward(self, edge_index, so: SelectOutput, *, edge_weight: Optional[Tensor] = None,
                batch: Optional[Tensor] = None, batch_pooled: Optional[Tensor] = None, **kwargs):
        s = self._validate_select_output(so)
        s32 = as_compute_dtype(s)  # fp32 arithmetic; the pooled adjacency carries the dtype of S
        if is_dense_adj(edge_index):
            adj_pool, w = self._forward_batched_inputs(as_compute_dtype(edge_index), s32, getattr(so, "_graph_sizes", None))
        else:
            adj_pool, w = self._forward_unbatched_inputs(edge_index, as_compute_dtype(edge_weight), batch, s32,
                                                         batch_pooled)
        if w is None:
            return like_input_dtype(adj_pool, s), None
        return adj_pool, like_input_dtype(w, s)

    def _forward_batched_inputs(self, adj: Tensor, s: Tensor, graph_sizes: Optional[Tensor] = None):
        s, adj = self._prepare_batched_dense_inputs(s, adj)
        if graph_sizes is not None and graph_sizes.numel() != s.size(0):
            graph_sizes = None
        if torch.is_grad_enabled() and (s.requires_grad or adj.requires_grad):
            raw = self._dense_connect(s, adj, graph_sizes)
            return postprocess_adj_pool_dense(raw, self.remove_self_loops, self.degree_norm, self.adj_transpose,
                                              self.edge_weight_norm), None
        flags = K.dense_flags(self.remove_self_loops, self.degree_norm, self.adj_transpose, self.edge_weight_norm)
        return K.dense_pool(s, adj, None, flags, graph_sizes=graph_sizes)[2], None

    def _forward_unbatched_inputs(self, edge_index, edge_weight, batch, s, batch_pooled):
        batch_size = num_graphs_of(batch)
        if s.dim() == 3 and s.size(0) == 1:
            s = s.squeeze(0)
        elif s.dim() != 2:
            raise ValueError("[DenseConnect - unbatched]: SelectOutput.s must have shape "
                             f"[N, K] or [1, N, K], but got {s.size()}.")
        num_nodes, num_clusters = s.size()
        adj_dense = self._dense_connect_unbatched(edge_index, edge_weight, batch, s, num_nodes, num_clusters,
                                                  batch_size)
        if not self.sparse_output:
            return postprocess_adj_pool_dense(adj_dense, self.remove_self_loops, self.degree_norm, False,
                                              self.edge_weight_norm), None
        if self.edge_weight_norm and batch_pooled is None:
            raise AssertionError("edge_weight_norm=True but batch_pooled=None. batch_pooled parameter is "
                                 "required for per-graph normalization in DenseConnect.")
        num_supernodes = batch_size * num_clusters
        ei, ew = Fn.block_diag_edges(adj_dense, None, self.remove_self_loops)
        ei, ew = _normalize_pooled_edges(ei, ew, num_supernodes, self.degree_norm, self.edge_weight_norm,
                                         batch_pooled)
        return _restore_format(edge_index, ei, ew, num_supernodes)

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}(remove_self_loops={self.remove_self_loops}, "
                f"degree_norm={self.degree_norm}, adj_transpose={self.adj_transpose}, "
                f"edge_weight_norm={self.edge_weight_norm}, sparse_output={self.sparse_output})")


class KronConnect(Connect):
    r"""Kron reduction L' = L[+,+] - L[+,-] L[-,-]^{-1} L[-,+] of the graph Laplacian, used by NDP
    (reference connect/kron_conn.py:26-168).

    The reference builds one sparse Laplacian for the whole batch and calls scipy's sparse LU on the host.  The batch
    Laplacian is block diagonal, so on the GPU every graph's Schur complement is taken independently by
    ``tgp_kron_batched_{count,fill}``: one workgroup per graph, dense fp64 elimination of the dropped nodes in LDS
    (graphs up to 128 nodes) or panel by panel in a workspace slab (up to ``tgp_kron_batched_max_graph_nodes`` = 8192
    nodes since r5; 4096 before), threshold / zero-diagonal / fp32 cast fused,
    edges emitted in the row-major order the reference's CSR -> COO conversion gives.  Nothing but the selector's
    Laplacian (uploaded once per SelectOutput) crosses PCIe.  Graphs beyond that size are skipped by the kernel and
    reduced one by one with the dense fp64 library solve on the device (``torch.linalg.solve`` -> rocSOLVER, up to
    ``dense_solve_max_nodes`` nodes per graph), the edge lists merged; a batch without a usable graph partition takes
    that library solve as a whole (up to ``dense_solve_max_nodes`` nodes in total), else the reference's scipy route
    unchanged.  All routes give the same
    edge set; weights agree to solver round-off before the fp32 cast."""

    def __init__(self, sparse_threshold: float = 1e-2, dense_solve_max_nodes: int = 8192,
                 native_max_nodes: Optional[int] = None):
        super().__init__()
        self.sparse_threshold = sparse_threshold
        self.dense_solve_max_nodes = dense_solve_max_nodes
        # graphs beyond this many nodes leave the hand-written kernels for the library solve (None: the kernels' own
        # limit, tgp_kron_batched_max_graph_nodes = 8192; a smaller value is a tuning / testing knob)
        self.native_max_nodes = native_max_nodes

    # ---------------------------------------------------------------- native block-batched route
    @staticmethod
    def _laplacian_csr_on_device(so: SelectOutput, device):
        """(indptr int32, col int64, val fp64) of ``so.L`` on the device; uploaded once per SelectOutput."""
        import numpy as np
        import scipy.sparse as sp
        hit = getattr(so, "_kron_csr", None)
        if hit is not None and hit[0] is so.L and hit[1] == device:
            return hit[2]
        dev_csr = getattr(so, "_L_device_csr", None)  # NDPSelect's device-built Laplacian, when it has one
        if dev_csr is not None and dev_csr[0].device == device:
            return dev_csr
        L = sp.csr_matrix(so.L)
        L.sum_duplicates()
        csr = (torch.from_numpy(L.indptr.astype(np.int32)).to(device),
               torch.from_numpy(L.indices.astype(np.int64)).to(device),
               torch.from_numpy(L.data.astype(np.float64)).to(device))
        so._kron_csr = (so.L, device, csr)
        return csr

    def _kron_native(self, edge_index: Tensor, edge_weight: Optional[Tensor], so: SelectOutput, idx_pos: Tensor,
                     has_laplacian: bool, batch: Optional[Tensor]):
        n = so.num_nodes
        dev = edge_index.device
        if n == 0 or idx_pos.device != dev:
            return None
        if batch is None:
            batch = getattr(so, "_node_batch", None)  # NDPSelect remembers the batch vector it partitioned by
        if batch is not None and batch.numel() == n and batch.device == dev:
            info = batch_info(batch)
            if not info.is_sorted:
                return None
            ptr, max_nodes, sizes_host = info.ptr, info.max_nodes, info.sizes_host
        else:
            ptr, max_nodes, sizes_host = torch.tensor([0, n], dtype=torch.long, device=dev), n, [n]
        limit = K.kron_max_graph_nodes()
        if self.native_max_nodes is not None:
            limit = max(1, min(limit, int(self.native_max_nodes)))
        oversize = None
        if max_nodes > limit:
            # A few graphs beyond the kernel's size limit must not send the whole batch to the host (one 1500-node
            # graph among 2000 small ones did): the kernel skips them, each is reduced by the dense fp64 library solve
            # on ITS block, and the edge lists are merged.  Only when such a graph is itself beyond the library route
            # (or the batch is one single graph) does the call fall back as a whole.
            sizes = ptr[1:] - ptr[:-1]
            oversize = (sizes > limit).nonzero().view(-1).tolist()
            if ptr.numel() <= 2 or int(sizes.max()) > self.dense_solve_max_nodes:
                return None
        adj_csr = so.__dict__.get("_adj_device_csr")
        if has_laplacian and adj_csr is not None and adj_csr[0].device == dev and adj_csr[0].numel() == n + 1:
            # NDPSelect's device route left the symmetrised adjacency on the GPU: L = D - A is formed in the kernel
            indptr, col, val, perm, from_adj = adj_csr[0], adj_csr[1], adj_csr[2], None, True
        elif has_laplacian:
            indptr, col, val = self._laplacian_csr_on_device(so, dev)
            if indptr.numel() != n + 1:
                return None
            perm, from_adj = None, False
        else:
            row = edge_index[0]
            if row.numel() > 1 and K._rows_sorted(edge_index, row):
                indptr, perm = torch.empty(n + 1, dtype=torch.int32, device=dev), None
                K.rowptr_from_sorted(row, n, indptr)
            else:
                index = K.build_assign_index(row, n)
                indptr, perm = index.row_ptr, index.perm
            col, val, from_adj = edge_index[1], edge_weight, True
        # NDPSelect compacted the kept nodes itself and left their prefix counts: valid for exactly the node_index it built
        held = so.__dict__.get("_node_rank")
        node_rank = None
        if held is not None and held[0] is not None and (idx_pos.data_ptr(), idx_pos.numel()) == (held[1], held[2]):
            node_rank = held[0]
        out = K.kron_batched(indptr, col, val, perm, from_adj, n, ptr, min(max_nodes, limit), idx_pos,
                             self.sparse_threshold, skip_oversize=oversize is not None, graph_sizes_host=sizes_host,
                             node_rank=node_rank)
        if out is None or not oversize:
            return out
        eis, ews = [out[0]], [out[1]]
        for g in oversize:
            p0, p1 = int(ptr[g]), int(ptr[g + 1])
            lap = self._dense_laplacian_block(indptr, col, val, perm, from_adj, p0, p1)
            if lap is None:
                return None  # an entry couples this graph to another one: not a block-diagonal batch
            lo, hi = torch.searchsorted(idx_pos, torch.tensor([p0, p1], device=dev)).tolist()
            if hi == lo:
                continue
            ei_g, ew_g = self._kron_on_device(lap, idx_pos[lo:hi] - p0)
            eis.append(ei_g + lo)
            ews.append(ew_g)
        ei_all, ew_all = torch.cat(eis, 1), torch.cat(ews)
        order = torch.sort(ei_all[0], stable=True).indices  # every block is row-major already; rows of one graph only
        return ei_all[:, order].contiguous(), ew_all[order]

    @staticmethod
    def _dense_laplacian_block(indptr: Tensor, col: Tensor, val: Optional[Tensor], perm: Optional[Tensor],
                               from_adjacency: bool, p0: int, p1: int) -> Optional[Tensor]:
        """Dense fp64 Laplacian of the graph that owns nodes [p0, p1), from the same CSR the kernel reads
        (from_adjacency: entries are edge weights, self loops skipped, duplicates summed, L = D - A)."""
        n = p1 - p0
        e0, e1 = int(indptr[p0]), int(indptr[p1])
        counts = (indptr[p0 + 1: p1 + 1] - indptr[p0: p1]).long()
        rows = torch.repeat_interleave(torch.arange(n, device=indptr.device), counts)
        slots = torch.arange(e0, e1, device=indptr.device)
        entries = slots if perm is None else perm[slots].long()
        c = col[entries] - p0
        if c.numel() and (int(c.min()) < 0 or int(c.max()) >= n):
            return None
        v = torch.ones(entries.numel(), dtype=torch.float64, device=indptr.device) if val is None else \
            val[entries].to(torch.float64)
        m = torch.zeros(n, n, dtype=torch.float64, device=indptr.device)
        m.index_put_((rows, c), v, accumulate=True)
        if not from_adjacency:
            return m
        m.fill_diagonal_(0)
        return torch.diag(m.sum(1)) - m

    # ---------------------------------------------------------------- library route (one dense solve)
    def _kron_on_device(self, L: Tensor, idx_pos: Tensor) -> Tuple[Tensor, Tensor]:
        """Dense fp64 Kron reduction of the WHOLE batch Laplacian on the GPU; returns the pooled (edge_index, fp32
        weights), row-major sorted like the scipy route's CSR -> COO conversion."""
        n = L.size(0)
        if idx_pos.numel() <= 1:
            l_new = -torch.ones((1, 1), dtype=torch.float64, device=L.device)
        else:
            keep = torch.zeros(n, dtype=torch.bool, device=L.device)
            keep[idx_pos] = True
            idx_neg = (~keep).nonzero().view(-1)
            l_pp = L[idx_pos][:, idx_pos]
            if idx_neg.numel() == 0:
                l_new = l_pp
            else:
                l_pn, l_np, l_nn = L[idx_pos][:, idx_neg], L[idx_neg][:, idx_pos], L[idx_neg][:, idx_neg]
                try:
                    x = torch.linalg.solve(l_nn, l_np)
                except RuntimeError:  # exactly singular complement: Marquardt-Levenberg damping (kron_conn.py:131-135)
                    damp = 1e-6 * torch.eye(l_nn.size(0), dtype=torch.float64, device=L.device)
                    x = torch.linalg.solve(l_nn + damp, l_np)
                l_new = l_pp - l_pn.matmul(x)
            if float((l_new - l_new.t()).abs().sum()) < float(torch.finfo(torch.float64).eps) * float(l_new.abs().sum()):
                l_new = (l_new + l_new.t()) / 2.0
        a = -l_new
        if self.sparse_threshold > 0:
            a = a * (a.abs() > self.sparse_threshold)
        a.fill_diagonal_(0)
        nz = a.nonzero()
        return nz.t().contiguous(), a[nz[:, 0], nz[:, 1]].to(torch.float32)

    def forward(self, edge_index, so: SelectOutput, edge_weight: Optional[Tensor] = None, **kwargs):
        import numpy as np
        import scipy.sparse as sp
        import scipy.sparse.linalg as spla

        template = edge_index
        edge_index, edge_weight = connectivity_to_edge_index(edge_index, edge_weight)
        device = edge_index.device
        n = so.num_nodes
        has_laplacian = so._has_laplacian() if hasattr(so, "_has_laplacian") else hasattr(so, "L")
        if has_laplacian:
            idx_pos_t = so.node_index
        else:
            warnings.warn("Laplacian not provided. The SelectOutput is not computed with NDPSelect.")
            if len(so.node_index) == so.num_supernodes:
                idx_pos_t = so.node_index
            elif getattr(so, "mis", None) is not None:
                idx_pos_t = so.mis
                if bool((idx_pos_t >= n).any()):
                    raise ValueError(f"MIS indices out of range: max idx={int(idx_pos_t.max())}, but graph has only "
                                     f"{n} nodes.")
            else:
                raise ValueError("Inconsistent number of clusters and node indices.")
        if device.type == "cuda":
            w = None if edge_weight is None else edge_weight.detach().reshape(-1)
            out = self._kron_native(edge_index, w, so, idx_pos_t, has_laplacian, kwargs.get("batch"))
            if out is not None:
                return _restore_format(template, out[0], out[1], so.num_supernodes)
        idx_pos = idx_pos_t.cpu().numpy()
        if has_laplacian:
            L = sp.csr_matrix(so.L)
        else:
            ei = edge_index.cpu().numpy()
            w = np.ones(ei.shape[1], dtype=np.float32) if edge_weight is None else \
                edge_weight.detach().reshape(-1).cpu().numpy()
            off = ei[0] != ei[1]
            A = sp.coo_matrix((w[off], (ei[0][off], ei[1][off])), shape=(n, n)).tocsr()
            L = (sp.diags(np.asarray(A.sum(1)).reshape(-1).astype(w.dtype)) - A).tocsr()
        if device.type == "cuda" and 0 < L.shape[0] <= self.dense_solve_max_nodes:
            l_dev = torch.from_numpy(L.toarray().astype(np.float64)).to(device)
            ei_out, ew_out = self._kron_on_device(l_dev, torch.as_tensor(idx_pos, dtype=torch.long, device=device))
            return _restore_format(template, ei_out, ew_out, so.num_supernodes)
        keep = np.zeros(L.shape[0], dtype=bool)
        keep[idx_pos] = True
        idx_neg = np.nonzero(~keep)[0]
        if len(idx_pos) <= 1:
            l_new = sp.csc_matrix(-np.ones((1, 1)))
        else:
            l_pp = L[np.ix_(idx_pos, idx_pos)]
            l_pn = L[np.ix_(idx_pos, idx_neg)]
            l_np = L[np.ix_(idx_neg, idx_pos)].tocsc()
            l_nn = L[np.ix_(idx_neg, idx_neg)].tocsc()
            try:
                l_new = l_pp - l_pn.dot(spla.spsolve(l_nn, l_np))
            except RuntimeError:
                damp = sp.csc_matrix(sp.eye(l_nn.shape[0]) * 1e-6)  # Marquardt-Levenberg damping
                l_new = l_pp - l_pn.dot(spla.spsolve(damp + l_nn, l_np))
            if np.abs(l_new - l_new.T).sum() < np.spacing(1) * np.abs(l_new).sum():
                l_new = (l_new + l_new.T) / 2.0
        a_pool = -l_new
        if self.sparse_threshold > 0:
            a_pool = a_pool.multiply(np.abs(a_pool) > self.sparse_threshold)
        a_pool = sp.lil_matrix(a_pool)
        a_pool.setdiag(0)
        a_pool = a_pool.tocsr()
        a_pool.eliminate_zeros()
        a_pool = a_pool.astype(np.float32).tocoo()
        ei_out = torch.stack([torch.from_numpy(a_pool.row).long(), torch.from_numpy(a_pool.col).long()]).to(device)
        ew_out = torch.from_numpy(a_pool.data).to(device)
        return _restore_format(template, ei_out, ew_out, so.num_supernodes)

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(sparse_threshold={self.sparse_threshold})"


__all__ = ["Connect", "SparseConnect", "DenseConnect", "KronConnect", "sparse_connect"]
