"""Tensor-level entry points of the hot path: allocate outputs/workspaces with PyTorch,
hand raw device pointers + the current HIP stream to the C ABI (``include/tgp_hip.h``).

Each function names the reference lines whose arithmetic it replaces.  Nothing here runs
on the CPU and nothing falls back to ATen kernels.
"""
from __future__ import annotations

import os
from typing import Optional, Sequence, Tuple
from weakref import ref as _weakref

import torch
from torch import Tensor

from . import _native as N


_OPS_MODULE = None


def ops_eps() -> float:
    """``tgp.utils.ops.eps`` read at CALL time, as the reference's ops functions read their module global
    (utils/ops.py:21,72,318,377,395; reference tests monkeypatch it: tests/connect/test_dense_conn.py:444-463)."""
    global _OPS_MODULE
    if _OPS_MODULE is None:
        from .utils import ops
        _OPS_MODULE = ops
    return float(_OPS_MODULE.eps)


def losses_eps() -> float:
    from .utils import losses
    return float(losses.eps)


# ------------------------------------------------------------------------- A1 / A2
class AssignIndex:
    """Inverted index of a sparse assignment (supernode -> its assignments, in ascending
    assignment order).  A function of the SelectOutput only, so SelectOutput caches it."""

    __slots__ = ("_row_ptr", "perm", "nnz", "num_targets", "_device", "pack", "pack_key", "member_directory",
                 "member_directory_key")

    def __init__(self, row_ptr: Optional[Tensor], perm: Optional[Tensor], nnz: int, num_targets: int, device=None):
        # row_ptr None: exactly one assignment per target (TopK, NDP) -- the table is arange and the Reduce kernel
        # skips reading it; it is materialised only for a consumer that wants the general form.  perm None (with
        # row_ptr None): target t owns assignment t -- the transposed index of a one-over-K assignment whose
        # node_index is 0..N-1 (Graclus): nothing to build, the kernel reads no table at all
        if row_ptr is None and nnz != num_targets:
            raise ValueError("an AssignIndex without row_ptr must be one-to-one")
        if perm is None and device is None and row_ptr is None:
            raise ValueError("an AssignIndex without perm and row_ptr must name its device")
        self._row_ptr, self.perm, self.nnz, self.num_targets = row_ptr, perm, nnz, num_targets
        # one-to-one indices (TopK, NDP) may carry the PACKED form: pack[c] = {int32 source row, fp32 weight} of target
        # c's single assignment (int64 words), valid for exactly the (source_index, weight) storage named by pack_key =
        # (source_index.data_ptr(), weight.data_ptr() or 0).  It is a snapshot of the selector's own output: like the
        # inverted index itself it is not refreshed if a caller edits S's indices / values in place afterwards.
        self.pack, self.pack_key = None, None
        # r5 (TopkSelect on large graphs): int32 [5 * blocks] = the kept-node bitmap (4 words per 128-node block) followed
        # by the rank directory (kept nodes in front of every block), by-products of the selector's compaction pass that
        # the subgraph Connect of the same selection starts from (tgp_connect_subgraph_single's member_bits_in)
        # ... valid for exactly the node_index storage named by member_directory_key = (data_ptr, length) (a
        # SelectOutput whose S was replaced must not get a Connect relabelled by the stale selection)
        self.member_directory = None
        self.member_directory_key = None
        self._device = perm.device if perm is not None else (row_ptr.device if row_ptr is not None else
                                                              torch.device(device))

    @property
    def one_to_one(self) -> bool:
        return self._row_ptr is None

    @property
    def row_ptr(self) -> Tensor:
        if self._row_ptr is None:
            self._row_ptr = torch.arange(self.num_targets + 1, dtype=torch.int32, device=self._device)
        return self._row_ptr


def build_assign_index(target_index: Tensor, num_targets: int) -> AssignIndex:
    dev = N.require_device(target_index)
    target_index = N.i64c(target_index)
    nnz = target_index.numel()
    row_ptr = torch.empty(num_targets + 1, dtype=torch.int32, device=dev)
    perm = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
    L = N.lib()
    wsb = L.tgp_assign_index_workspace_bytes(nnz, num_targets)
    ws = N.workspace(wsb, dev)
    N.check(L.tgp_assign_index_build(N.ptr(target_index), nnz, num_targets, N.ptr(row_ptr), N.ptr(perm),
                                     N.ptr(ws), ws.numel(), N.stream_ptr(dev)), "tgp_assign_index_build")
    return AssignIndex(row_ptr, perm, nnz, num_targets)


def reduce_sparse(x: Tensor, source_index: Tensor, weight: Optional[Tensor], index: AssignIndex,
                  identity_source: bool = False, unit_weight: bool = False) -> Tensor:
    """out[t,:] = sum_{i: target[i]==t} weight[i] * x[source_index[i],:]
    (reduce/base_reduce.py:146-153; lift/base_lift.py:102-111 with the roles swapped).
    ``identity_source`` / ``unit_weight``: the caller KNOWS that source_index is 0..nnz-1 / that every weight is 1.0 (a
    clustering: GraclusSelect, a cluster vector) -- the fp32 kernel then skips those two loads per assignment; same bits."""
    dev = N.require_device(x, source_index, weight)
    squeeze = x.dim() == 1
    x2 = x.view(-1, 1) if squeeze else x
    if x2.dim() != 2:
        raise ValueError(f"sparse reduce expects x of shape [N, F], got {tuple(x.shape)}")
    if x2.dtype == torch.float64 or (weight is not None and weight.dtype == torch.float64):
        # float64 features or weights: the product promotes to fp64 in the reference (base_reduce.py:146-153) -- fp64
        # kernel, same summation order
        x2 = N.f64c(x2) if x2.stride(1) == 1 and x2.dtype == torch.float64 else x2.to(torch.float64).contiguous()
        src64 = N.i64c(source_index)
        w64 = None if weight is None else N.f64c(weight.reshape(-1))
        out = torch.empty(index.num_targets, x2.size(1), dtype=torch.float64, device=dev)
        N.check(N.lib().tgp_reduce_sparse_f64(N.ptr(x2), x2.size(0), x2.size(1), x2.stride(0), N.ptr(src64), N.ptr(w64),
                                              N.ptr(index._row_ptr), N.ptr(index.perm), index.nnz, index.num_targets,
                                              N.ptr(out), N.stream_ptr(dev)), "tgp_reduce_sparse_f64")
        return out.view(-1) if squeeze else out
    x2 = x2.to(torch.float32) if x2.dtype != torch.float32 else x2
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    source_index = N.i64c(source_index)
    w = None if weight is None else N.f32c(weight.reshape(-1))
    F = x2.size(1)
    out = torch.empty(index.num_targets, F, dtype=torch.float32, device=dev)
    if (index.pack is not None and index.pack_key == (source_index.data_ptr(), 0 if w is None else w.data_ptr())
            and F % 4 == 0 and F >= 32 and x2.stride(0) % 4 == 0 and x2.data_ptr() % 16 == 0):
        # one node per target (TopK, NDP): the packed index, one streamed 8-byte load per pooled row
        N.check(N.lib().tgp_reduce_one_to_one_f32(N.ptr(x2), x2.size(0), F, x2.stride(0), N.ptr(index.pack),
                                                  0 if w is None else 1, index.num_targets, N.ptr(out),
                                                  N.stream_ptr(dev)), "tgp_reduce_one_to_one_f32")
        return out.view(-1) if squeeze else out
    fast = index._row_ptr is not None and index.nnz <= x2.size(0)
    N.check(N.lib().tgp_reduce_sparse_f32(N.ptr(x2), x2.size(0), F, x2.stride(0),
                                          None if (identity_source and fast) else N.ptr(source_index),
                                          None if unit_weight else N.ptr(w),
                                          N.ptr(index._row_ptr), N.ptr(index.perm), index.nnz,
                                          index.num_targets, N.ptr(out), N.stream_ptr(dev)),
            "tgp_reduce_sparse_f32")
    return out.view(-1) if squeeze else out


def one_to_one_index(node_index: Tensor, cluster_index: Tensor, weight: Optional[Tensor]) -> AssignIndex:
    """Inverted index of an assignment whose supernodes own exactly one node each (cluster_index a permutation of
    0..k-1: TopK, NDP): perm = the inverse permutation, plus the packed {source row, weight} form the Reduce kernel
    streams -- one launch, no sort (select/base_select.py:58 keeps the assignment node-sorted)."""
    dev = N.require_device(node_index, cluster_index, weight)
    ni, ci = N.i64c(node_index), N.i64c(cluster_index)
    w = None if weight is None else N.f32c(weight.reshape(-1))
    k = ci.numel()
    perm = torch.empty(max(k, 1), dtype=torch.int32, device=dev)
    pack = torch.empty(max(k, 1), dtype=torch.int64, device=dev)
    N.check(N.lib().tgp_one_to_one_index_build(N.ptr(ni), N.ptr(ci), N.ptr(w), k, N.ptr(perm), N.ptr(pack),
                                               N.stream_ptr(dev)), "tgp_one_to_one_index_build")
    index = AssignIndex(None, perm, k, k)
    index.pack, index.pack_key = pack, (ni.data_ptr(), 0 if w is None else w.data_ptr())
    return index


def reduce_batch_sparse(batch: Tensor, node_index: Tensor, cluster_index: Tensor, num_supernodes: int,
                        every_cluster_has_a_node: bool = False) -> Tensor:
    """reduce/base_reduce.py:37-41.  ``every_cluster_has_a_node``: the caller knows that no supernode is empty (the
    ``arange`` underneath the scatter never shows): one launch instead of two."""
    dev = N.require_device(batch, node_index, cluster_index)
    batch, node_index, cluster_index = N.i64c(batch), N.i64c(node_index), N.i64c(cluster_index)
    out = torch.empty(num_supernodes, dtype=torch.int64, device=dev)
    N.check(N.lib().tgp_reduce_batch_i64(N.ptr(batch), N.ptr(node_index), N.ptr(cluster_index),
                                         node_index.numel(), num_supernodes, 1 if every_cluster_has_a_node else 0,
                                         N.ptr(out), N.stream_ptr(dev)),
            "tgp_reduce_batch_i64")
    return out


# ------------------------------------------------------------------------- A4 / A5 / A6
def _edge_rows(edge_index: Tensor) -> Tuple[Tensor, Tensor]:
    """The two rows of a [2, E] list as contiguous int64 vectors.  The C ABI takes them as two pointers, so a list whose
    ROWS are contiguous (e.g. the narrowed capacity buffers of ``sparse_pool_small``) needs no copy."""
    ei = edge_index
    if ei.dtype != torch.int64:
        ei = ei.to(torch.int64)
    if not (ei.dim() == 2 and ei.size(0) == 2 and (ei.stride(1) == 1 or ei.size(1) <= 1)):
        ei = ei.contiguous()
    return ei[0], ei[1]


# ------------------------------------------------------------------------- A1 + A2 + A4/A5 + A6, batches of small graphs
_SPS_STATE: dict = {}  # (device index, stream handle) -> _SpsState
_SPS_DECLINED: dict = {}
_SPS_WORDS: dict = {}  # (graphs, mode) -> look-back words of the one-launch kernel (a native call otherwise)
# r6: the exact-size outputs of the one-launch sparse pooling are carved out of ONE allocation of at most this many bytes
# (what a retained x' can pin); TGP_SPS_ARENA=0: four allocations of their own, sized once the count has arrived
_SPS_ARENA = os.environ.get("TGP_SPS_ARENA", "1") != "0"
_SPS_ARENA_BYTES = 16 << 20
_SPS_ONE_ALLOC_BYTES = 32 << 20  # outputs of the one-launch sparse pooling below this size share one allocation
_SPS_GIVE_PTRS = os.environ.get("TGP_SPS_GIVE_PTRS", "1") != "0"  # A/B switch: hand the per-graph offsets to the kernel
_GRACLUS_FUSED = os.environ.get("TGP_GRACLUS_FUSED", "1") != "0"  # A/B switch: one-launch GraclusSelect of small graphs
_PUBLISH_COUNTS = os.environ.get("TGP_PUBLISH_COUNTS", "1") != "0"  # A/B switch of _read_count (read once)
SPS_COMPACT_BYTES = 1 << 30  # capacity buffers above this are replaced by exact copies when mostly empty

# Output contract of the single-call operators (r5).  The reference hands out new tensors of exactly the pooled size
# (connect/base_conn.py:103-112; SURVEY 8(b) "Ownership"); the one-launch kernels write into CAPACITY-sized buffers
# because the count is only known afterwards.  Default (False): exact-size outputs -- ``edge_index`` a contiguous [2, E']
# tensor that owns 16 E' bytes -- filled from the capacity scratch by one native launch (tgp_edges_compact).  True: the
# r4 behaviour, ``edge_index`` a [2, E'] VIEW of the capacity buffer (row stride E, E-sized storage kept alive; no copy)
# for callers that consume the pooled graph at once, e.g. straight into SparseGather or the next pooling level.
_OUTPUT_VIEWS_DEFAULT = os.environ.get("TGP_OUTPUT_VIEWS", "0") == "1"
_OUTPUT_VIEWS_LOCAL = __import__("threading").local()  # per thread: a `with output_views()` in one thread must not
#                                                        change the layout other threads (autograd workers) hand out


def _output_views() -> bool:
    return getattr(_OUTPUT_VIEWS_LOCAL, "value", _OUTPUT_VIEWS_DEFAULT)


def set_output_views(enable: bool) -> bool:
    """Choose the layout of the pooled edge lists of the single-call operators for THIS thread (default: the process-wide
    ``TGP_OUTPUT_VIEWS`` setting); returns the previous setting."""
    prev = _output_views()
    _OUTPUT_VIEWS_LOCAL.value = bool(enable)
    return prev


class output_views:
    """``with tgp.kernels.output_views():`` -- capacity views inside the block (opt-in), restored afterwards."""

    def __init__(self, enable: bool = True):
        self.enable = enable

    def __enter__(self):
        self.prev = set_output_views(self.enable)
        return self

    def __exit__(self, *exc):
        set_output_views(self.prev)
        return False


def _compact_edges(L, st, dev, row_p: int, col_p: int, w_p, w_dtype, id_p, n_out: int):
    """Exact-size (edge_index [2,n], weight [n] or None, edge_id [n] or None) from the first n entries of capacity
    arrays: two or three allocations and ONE launch."""
    out_ei = torch.empty(2, n_out, dtype=torch.int64, device=dev)
    out_w = None if w_p is None else torch.empty(n_out, dtype=w_dtype, device=dev)
    out_id = None if id_p is None else torch.empty(n_out, dtype=torch.int64, device=dev)
    if n_out:
        o = out_ei.data_ptr()
        N.check(L.tgp_edges_compact(row_p, col_p, w_p, 0 if w_p is None else out_w.element_size(), id_p, n_out,
                                    o, o + 8 * n_out, N.ptr(out_w), N.ptr(out_id), st), "tgp_edges_compact")
    return out_ei, out_w, out_id


class _SpsState:
    """Per (device, stream): the look-back words of the one-launch kernels (``tgp_sparse_pool_small_f32``, the single-pass
    subgraph Connect, the coalesce Connect's scans, the one-launch GraclusSelect; device memory, never cleared: every
    word carries the epoch of the call that wrote it), the epoch counter, and ONE pinned host word the kernel's last
    workgroup stores {epoch, refused, total} into -- the host polls it instead of paying a device-to-host copy kernel
    and a stream synchronise for eight bytes (``tgp_count_publish`` gives every count -> fill pair the same read)."""

    __slots__ = ("status", "epoch", "pinned", "host", "ticket", "facts_pinned", "facts_host", "facts_tag")

    def __init__(self, dev, words):
        self.status = torch.zeros(max(int(words), 4096), dtype=torch.int64, device=dev)
        self.epoch = 0
        self.pinned = torch.zeros(8, dtype=torch.int64).pin_memory()
        self.host = self.pinned.numpy()  # the same memory
        # the one-launch batch facts (utils.ops._batch_facts_sorted): arrival ticket + flag word (zero between calls) and
        # the pinned words {tag, B - 1, flags, longest graph, non-empty graphs, sum of TopK keep counts}
        self.ticket = torch.zeros(8, dtype=torch.int32, device=dev)  # words 0-1 batch facts, 2-3 edge facts / symmetry,
        #                                                               4 the MinCut tail's batch means (r6)
        self.facts_pinned = torch.zeros(8 * 8, dtype=torch.int64).pin_memory()  # 8 slots of 8 words (prefetched calls)
        self.facts_host = self.facts_pinned.numpy()
        self.facts_tag = 0

    def next_facts_tag(self) -> int:
        self.facts_tag += 1
        return self.facts_tag

    def facts_slot(self, tag: int) -> int:
        """Address of the pinned result words of call ``tag`` (calls rotate through eight slots: a prefetched call's
        words are not overwritten by the calls enqueued behind it)."""
        return self.facts_pinned.data_ptr() + 64 * (tag & 7)

    def wait_facts(self, tag: int):
        host, spins, o = self.facts_host, 0, 8 * (tag & 7)
        while int(host[o]) != tag:
            spins += 1
            if spins > 4_000_000:
                torch.cuda.synchronize(self.status.device)
                if int(host[o]) != tag:
                    raise N.TgpNativeError("tgp_batch_facts_sorted_i64 finished without storing its result word")
        return int(host[o + 1]), int(host[o + 2]), int(host[o + 3]), int(host[o + 4]), int(host[o + 5])

    def peek_facts(self, tag: int):
        """The flag word of call ``tag``, or None when its slot has been reused by a later call (eight slots rotate)."""
        host, spins, o = self.facts_host, 0, 8 * (tag & 7)
        while True:
            have = int(host[o])
            if have == tag:
                return int(host[o + 2])
            if have > tag:
                return None
            spins += 1
            if spins > 4_000_000:
                torch.cuda.synchronize(self.status.device)
                if int(host[o]) < tag:
                    raise N.TgpNativeError("a facts launch finished without storing its result word")

    def next_epoch(self) -> int:
        self.epoch += 1
        if self.epoch >= (1 << 29) - 1:  # epochs of a buffer never repeat: start over on a cleared buffer
            torch.cuda.synchronize(self.status.device)
            self.status.zero_()
            self.host[0] = 0
            self.epoch = 1
        return self.epoch

    def wait(self, epoch: int) -> int:
        """The result word of call ``epoch`` (spins on the pinned word; a stuck device is turned into an error)."""
        host, spins = self.host, 0
        while True:
            word = int(host[0])
            if (word >> 34) == epoch:
                return word
            spins += 1
            if spins > 4_000_000:
                torch.cuda.synchronize(self.status.device)  # surfaces a device fault, if that is what happened
                word = int(host[0])
                if (word >> 34) == epoch:
                    return word
                raise N.TgpNativeError("a kernel that hands its count over in a pinned host word finished without storing it")


def _sps_state(dev: torch.device, stream: int, words: int) -> "_SpsState":
    key = (dev.index, stream)
    ent = _SPS_STATE.get(key)
    if ent is None or ent.status.numel() < words:
        ent = _SpsState(dev, words)
        _SPS_STATE[key] = ent
    return ent


def sparse_pool_small_declined(edge_index: Tensor) -> bool:
    """Did the one-launch kernel refuse this very edge list before (unsorted rows, an edge between two graphs, ...)?
    Remembered per tensor object + version, like the row-order memo, so that a refusal costs one launch once."""
    hit = _SPS_DECLINED.get(id(edge_index))
    return hit is not None and hit[0]() is edge_index and hit[1] == edge_index._version


def _sps_remember_declined(edge_index: Tensor) -> None:
    import weakref
    if len(_SPS_DECLINED) >= 16:
        for key in [k for k, v in _SPS_DECLINED.items() if v[0]() is None]:
            del _SPS_DECLINED[key]
        while len(_SPS_DECLINED) >= 16:
            del _SPS_DECLINED[next(iter(_SPS_DECLINED))]
    _SPS_DECLINED[id(edge_index)] = (weakref.ref(edge_index), edge_index._version)


def sparse_pool_small_max_graph_nodes() -> int:
    return int(N.lib().tgp_sparse_pool_small_max_graph_nodes())


_EDGE_PTR: dict = {}  # id(edge_index) -> (weakref, version, id(graph_ptr), weakref(graph_ptr), edge_ptr)


def _edge_ptr_memo(edge_index: Tensor, graph_ptr: Tensor) -> Optional[Tensor]:
    hit = _EDGE_PTR.get(id(edge_index))
    if (hit is not None and hit[0]() is edge_index and hit[1] == edge_index._version and hit[2] == id(graph_ptr)
            and hit[3]() is graph_ptr):
        return hit[4]
    return None


def _edge_ptr_remember(edge_index: Tensor, graph_ptr: Tensor, out: Tensor) -> None:
    key = id(edge_index)
    if key in _EDGE_PTR:
        del _EDGE_PTR[key]  # (re-inserted at the young end)
    elif len(_EDGE_PTR) >= 16:
        del _EDGE_PTR[next(iter(_EDGE_PTR))]  # the oldest entry (dicts keep insertion order): no scan on the hot path
    _EDGE_PTR[key] = (_weakref(edge_index), edge_index._version, id(graph_ptr), _weakref(graph_ptr), out)


def graph_edge_ptr(edge_index: Tensor, graph_ptr: Tensor) -> Tensor:
    """First edge of every graph of a sorted batch in a row-sorted list ([B+1] int64: lower bounds of ``graph_ptr`` in the
    row array), remembered per (edge list object + version, graph_ptr object): one tiny launch for a new pair, nothing
    for a pair that is pooled again.  The consumers re-check what they read through it."""
    hit = _edge_ptr_memo(edge_index, graph_ptr)
    if hit is not None:
        return hit
    dev = edge_index.device
    gp = N.i64c(graph_ptr)
    out = torch.empty(gp.numel(), dtype=torch.int64, device=dev)
    row = edge_index[0]
    N.check(N.lib().tgp_graph_lower_bounds_i64(N.ptr(row) if row.numel() else None, row.numel(), N.ptr(gp),
                                               gp.numel() - 1, N.ptr(out), N.stream_ptr(dev)),
            "tgp_graph_lower_bounds_i64")
    _edge_ptr_remember(edge_index, graph_ptr, out)
    return out


def sparse_pool_small(x: Tensor, graph_ptr: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor],
                      assign_index: Tensor, weight: Optional[Tensor], num_supernodes: int,
                      mode: int, reduce_op: str = "sum", remove_self_loops: bool = True, want_batch: bool = True,
                      assign_ptr: Optional[Tensor] = None, views: Optional[bool] = None, checked: bool = False):
    """Sparse Reduce + Connect of a sorted batch of graphs of at most 64 nodes in ONE launch
    (reduce/base_reduce.py:14-53,141-155; connect/base_conn.py:79-89; the filters of utils/ops.py:370-380):
    ``(x_pool [K,F], batch_pool [K] or None, edge_index' [2,E'], edge_weight' [E'] or None)``, bit-identical to
    ``reduce_sparse`` + ``reduce_batch_sparse`` + ``filter_edges`` (mode 0) / ``coalesce_edges`` (mode 1).  The pooled
    edges are written once at their final offsets of capacity-E scratch and -- by default -- moved into exact-size
    outputs by one more launch once the count is known (four tensors of their own, ``edge_index'`` contiguous);
    ``views=True`` (default: ``tgp.kernels.output_views``) hands out the r4 layout instead: all four outputs carved out
    of ONE allocation, ``edge_index'`` a view of the capacity buffer whose two rows are contiguous, no copy.
    ``assign_index`` [2, nnz] = (node_index, cluster_index), the indices of the sparse S.  None: a precondition checked
    on the device does not hold (the caller takes the staged operators)."""
    if views is None:
        views = _output_views()
    # (this wrapper sits in front of a ~10 us kernel: pointers by arithmetic instead of row views; `checked`: the caller
    #  -- SRCPooling.reduce_connect -- has validated x, edge_index and the weights' dtypes: no second pass over them)
    if checked:
        dev = x.device
        if not (graph_ptr.is_cuda and assign_index.is_cuda and (weight is None or weight.is_cuda)):
            dev = N.require_device(x, graph_ptr, edge_index, edge_weight, assign_index, weight)
    else:
        dev = N.require_device(x, graph_ptr, edge_index, edge_weight, assign_index, weight)
        if x.dim() != 2 or x.dtype != torch.float32 or x.stride(1) != 1:
            raise ValueError("sparse_pool_small expects float32 x [N, F] with unit feature stride")
    ei = edge_index
    es0, es1 = ei.stride() if ei.dim() == 2 else (0, 0)
    if not (ei.dtype == torch.int64 and ei.dim() == 2 and ei.size(0) == 2 and (es1 == 1 or ei.size(1) <= 1)):
        ei = N.i64c(ei)
        es0 = ei.stride(0)
    E = ei.size(1)
    row_p = ei.data_ptr()
    col_p = row_p + 8 * es0
    w = edge_weight
    if w is not None and not (w.dim() == 1 and w.dtype == torch.float32 and (w.stride(0) == 1 or E <= 1)):
        w = N.f32c(w.reshape(-1))
    ai = assign_index
    as0, as1 = ai.stride() if ai.dim() == 2 else (0, 0)
    if not (ai.dtype == torch.int64 and ai.dim() == 2 and ai.size(0) == 2 and (as1 == 1 or ai.size(1) <= 1)):
        ai = N.i64c(ai)
        as0 = ai.stride(0)
    nnz = ai.size(1)
    ni_p = ai.data_ptr()
    ci_p = ni_p + 8 * as0
    gp = graph_ptr if (graph_ptr.dtype == torch.int64 and graph_ptr.is_contiguous()) else N.i64c(graph_ptr)
    wt = weight
    if wt is not None and not (wt.dim() == 1 and wt.dtype == torch.float32 and (wt.stride(0) == 1 or nnz <= 1)):
        wt = N.f32c(wt.reshape(-1))
    (n, F), K, B = x.shape, int(num_supernodes), gp.numel() - 1
    ecap = max(E, 1)
    # ONE allocation for all four outputs when they are small (a batch of small graphs: a few MB): only addresses are
    # needed to launch, and the typed views are made WHILE the kernel runs, behind the launch -- this wrapper sits in
    # front of a ~10 us kernel and every allocation in front of the launch is ~1.2 us the GPU idles.  (Large inputs keep
    # separate buffers: a retained x_pool must not pin tens of MB of edge capacity.)
    #   views (tgp.kernels.output_views):  x' | batch' | rows cap | cols cap | w cap        edge_index' = [2, n] of row
    #                                      stride ecap, no copy
    #   default (r6: the arena):           x' | batch' | rows cap + room for n cols | w cap | col scratch
    #                                      the kernel's rows and weights ARE the outputs; one launch moves the n columns
    #                                      behind the n rows once the count is known: edge_index' contiguous [2, n]
    ob = (K * F * 4 + 15) & ~15
    oe = (ob + (K * 8 if want_batch else 0) + 15) & ~15
    ow = (oe + 2 * ecap * 8 + 15) & ~15
    oc = (ow + (ecap * 4 if w is not None else 0) + 15) & ~15
    arena = (not views) and _SPS_ARENA and E > 0 and oc + 8 * ecap <= _SPS_ARENA_BYTES
    one = views and oc <= _SPS_ONE_ALLOC_BYTES
    if one or arena:
        buf = torch.empty((oc + (8 * ecap if arena else 0)) >> 3, dtype=torch.int64, device=dev)
        base = buf.data_ptr()
        xp_p, bp_p, cap_p, cw_p = base, (base + ob) if want_batch else None, base + oe, (base + ow) if w is not None else None
        col_out = base + oc if arena else cap_p + 8 * ecap
    elif not views:
        # exact outputs: x' and batch' are sized before the launch; the edges go through one capacity scratch buffer
        x_pool = torch.empty(K, F, dtype=torch.float32, device=dev)
        batch_pool = torch.empty(K, dtype=torch.int64, device=dev) if want_batch else None
        scratch = torch.empty(2 * ecap * 8 + (ecap * 4 if w is not None else 0), dtype=torch.uint8, device=dev)
        cap_p = scratch.data_ptr()
        cw_p = cap_p + 16 * ecap if w is not None else None
        xp_p, bp_p = x_pool.data_ptr(), N.ptr(batch_pool)
        col_out = cap_p + 8 * ecap
    else:
        x_pool = torch.empty(K, F, dtype=torch.float32, device=dev)
        batch_pool = torch.empty(K, dtype=torch.int64, device=dev) if want_batch else None
        cap = torch.empty(2, ecap, dtype=torch.int64, device=dev)
        cap_w = None if w is None else torch.empty(ecap, dtype=torch.float32, device=dev)
        xp_p, bp_p, cap_p, cw_p = x_pool.data_ptr(), N.ptr(batch_pool), cap.data_ptr(), N.ptr(cap_w)
        col_out = cap_p + 8 * ecap
    L = N.lib()
    st = N.stream_ptr(dev)
    words = _SPS_WORDS.get((B, mode))
    if words is None:
        if len(_SPS_WORDS) > 256:
            _SPS_WORDS.clear()
        words = _SPS_WORDS[(B, mode)] = int(L.tgp_sparse_pool_small_status_words(B, mode))
    state = _sps_state(dev, st, words)
    epoch = state.next_epoch()
    flags = (N.REMOVE_SELF_LOOPS if remove_self_loops else 0) | (N.EPS_FILTER if w is not None else 0)
    # per-graph offsets the caller side already has: the kernel skips its searches (and re-checks what it reads).  r5: an
    # edge list that is NEW to this process (every mini-batch of a training loop) no longer pays a lower-bounds launch
    # for them: the kernel searches for its ranges as it did before r4 and LEAVES them in `eptr_out`, which is remembered
    # for the next call on the same tensors
    eptr = _edge_ptr_memo(edge_index, graph_ptr) if (E and _SPS_GIVE_PTRS) else None
    eptr_out = None
    if eptr is None and E and _SPS_GIVE_PTRS:
        eptr_out = torch.empty(B + 1, dtype=torch.int64, device=dev)
    aptr = None
    if eptr is not None and mode == 0:
        if assign_ptr is not None and assign_ptr.dtype == torch.int64 and assign_ptr.numel() == B + 1 and assign_ptr.is_cuda:
            aptr = assign_ptr if assign_ptr.is_contiguous() else assign_ptr.contiguous()
        else:
            eptr = None  # (mode 0 needs both tables)
    pinned_p = state.pinned.data_ptr()
    N.check(L.tgp_sparse_pool_small_f32(x.data_ptr(), n, F, x.stride(0), gp.data_ptr(), B, N.ptr(eptr), N.ptr(aptr),
                                        N.ptr(eptr_out), row_p if E else None,
                                        col_p if E else None, N.ptr(w), E, ni_p, ci_p, N.ptr(wt),
                                        nnz, K, mode, N.REDUCE_OPS[reduce_op], flags, ops_eps(),
                                        xp_p, bp_p, cap_p, col_out, cw_p,
                                        state.status.data_ptr(), state.status.numel(), pinned_p, epoch,
                                        st), "tgp_sparse_pool_small_f32")
    if one or arena:  # (behind the launch: the kernel is running)
        f32 = buf.view(torch.float32)
        x_pool = torch.as_strided(f32, (K, F), (F, 1), 0)
        batch_pool = torch.as_strided(buf, (K,), (1,), ob >> 3) if want_batch else None
    if eptr_out is not None:  # (forgotten again below if the kernel refuses the input)
        _edge_ptr_remember(edge_index, graph_ptr, eptr_out)
    if arena:
        # the call's one host wait (the reference's .item() syncs) and the column move, in one native call (r6)
        N.check(L.tgp_result_wait_pack_cols(pinned_p, epoch, col_out, cap_p, st), "tgp_result_wait_pack_cols")
        total = int(state.host[0])
        if (total >> 34) != epoch:
            raise N.TgpNativeError("tgp_result_wait_pack_cols returned before the result word of its call was stored")
    else:
        total = state.wait(epoch)
    if total & 0x80000000:
        if eptr_out is not None:
            _EDGE_PTR.pop(id(edge_index), None)
        _sps_remember_declined(edge_index)
        return None
    n_out = total & 0x7FFFFFFF
    if arena:
        ei = torch.as_strided(buf, (2, n_out), (max(n_out, 1), 1), oe >> 3)
        ew = torch.as_strided(f32, (n_out,), (1,), ow >> 2) if w is not None else None
        return x_pool, batch_pool, ei, ew
    if not views:
        ei, ew, _ = _compact_edges(L, st, dev, cap_p, col_out, cw_p, torch.float32, None, n_out)
        return x_pool, batch_pool, ei, ew
    if one:
        ei = torch.as_strided(buf, (2, n_out), (ecap, 1), oe >> 3)
        ew = torch.as_strided(f32, (n_out,), (1,), ow >> 2) if w is not None else None
    else:
        ei = cap[:, :n_out]
        ew = None if cap_w is None else cap_w[:n_out]
    if E * 16 > SPS_COMPACT_BYTES and 2 * n_out < E:
        ei, ew = ei.contiguous(), None if ew is None else ew.clone()
    return x_pool, batch_pool, ei, ew


def _decode_count(word: int) -> int:
    """The count of a published result word: bits 0..33, two's complement (decline codes are negative)."""
    n = word & ((1 << 34) - 1)
    return n - (1 << 34) if n >= 1 << 33 else n


def _read_count(d_count: Tensor) -> int:
    # the single host wait of a count -> fill pair (the reference pays `.item()` syncs too).  r4: a one-thread kernel
    # stores the count into a pinned host word this thread polls -- a few microseconds instead of the copy kernel +
    # stream synchronise of `.item()`; the wait stays in the MIDDLE of the call, so the fill still overlaps whatever
    # the caller launches next.  Under stream capture neither form is legal: the count routes are not capturable.
    if _PUBLISH_COUNTS and not torch.cuda.is_current_stream_capturing():
        dev = d_count.device
        st = N.stream_ptr(dev)
        state = _sps_state(dev, st, 0)
        epoch = state.next_epoch()
        N.check(N.lib().tgp_count_publish(N.ptr(d_count), state.pinned.data_ptr(), epoch, st), "tgp_count_publish")
        n = _decode_count(state.wait(epoch))
    else:
        n = int(d_count.item())
    if n == -2:  # refusal code of the count kernels: an endpoint (or cluster id) outside its table
        raise IndexError("edge_index holds node ids outside [0, num_nodes) (or cluster ids outside [0, num_supernodes)): "
                         "the reference's index ops raise for these inputs too")
    return n


def member_directory_for(assign: Optional["AssignIndex"], node_index: Optional[Tensor]) -> Optional[Tensor]:
    """The selector's kept-node bitmap + rank directory, if it still describes ``node_index`` (the same storage and length
    as when TopkSelect wrote it: a SelectOutput whose ``s`` was replaced does not get the old selection's directory);
    None otherwise."""
    if assign is None or node_index is None:
        return None
    md, key = getattr(assign, "member_directory", None), getattr(assign, "member_directory_key", None)
    if md is None or key is None or key != (node_index.data_ptr(), node_index.numel()):
        return None
    return md


def filter_edges(edge_index: Tensor, edge_weight: Optional[Tensor], node_index: Optional[Tensor],
                 num_nodes: int, remove_self_loops: bool, want_edge_id: bool = False, views: Optional[bool] = None,
                 member_directory: Optional[Tensor] = None):
    """Induced subgraph + relabel (connect/base_conn.py:79-82) fused with remove_self_loops and the
    |w| > eps filter (utils/ops.py:370-380).  node_index=None: filters only.  Keeps input order.
    ``want_edge_id``: also return the input position of every kept edge (what the backward of the weight
    pass-through scatters by).  ``views`` (default ``tgp.kernels.output_views``, i.e. False): hand out views of the
    single-pass kernel's capacity-E buffers instead of exact-size tensors.  ``member_directory``: the selector's own
    bitmap + rank directory of ``node_index`` (``AssignIndex.member_directory``; node_index must be ascending, as a
    SelectOutput's is): the call is then ONE launch -- no memset, no scatter of the kept nodes, no directory scan."""
    if views is None:
        views = _output_views()
    dev = N.require_device(edge_index, edge_weight, node_index)
    row, col = _edge_rows(edge_index)
    E = row.numel()
    f64 = edge_weight is not None and edge_weight.dtype == torch.float64  # fp64 weights pass through as they are
    w = None if edge_weight is None else (N.f64c if f64 else N.f32c)(edge_weight.reshape(-1))
    ni = None if node_index is None else N.i64c(node_index)
    flags = (N.REMOVE_SELF_LOOPS if remove_self_loops else 0) | (N.EPS_FILTER if w is not None else 0)
    if want_edge_id:
        flags |= N.WANT_EDGE_ID
    if ni is not None:
        flags |= N.NODE_FILTER
        if ni.numel() == 0:  # no node kept: no edge survives (an empty tensor has no device pointer to hand over)
            out = (torch.empty(2, 0, dtype=torch.int64, device=dev),
                   None if w is None else torch.empty(0, dtype=w.dtype, device=dev))
            return out + (torch.empty(0, dtype=torch.int64, device=dev),) if want_edge_id else out
    L = N.lib()
    st = N.stream_ptr(dev)
    eps = ops_eps()
    if E > 0 and not torch.cuda.is_current_stream_capturing():
        # ONE pass (r4): survivors written once at their final offsets of capacity-E buffers, which are then narrowed
        # (edge_index' is a view whose two rows are contiguous); the count arrives in a pinned host word
        md = member_directory if (ni is not None and member_directory is not None and member_directory.is_cuda
                                  and member_directory.dtype == torch.int32
                                  and member_directory.numel() == 5 * int(L.tgp_topk_select_directory_blocks(num_nodes))
                                  ) else None
        got = _filter_edges_single(L, dev, st, row, col, w, E, ni, num_nodes, flags, eps, want_edge_id, views, md)
        if got is None and f64:  # (a look-back spin bound on a shared device: once more; fp64 has no count -> fill pair)
            got = _filter_edges_single(L, dev, st, row, col, w, E, ni, num_nodes, flags, eps, want_edge_id, views, md)
        if got is not None:
            return got
    if f64 and E == 0:
        out = (torch.empty(2, 0, dtype=torch.int64, device=dev), torch.empty(0, dtype=torch.float64, device=dev))
        return out + (torch.empty(0, dtype=torch.int64, device=dev),) if want_edge_id else out
    if f64:
        # the single pass declined twice (a look-back spin bound: the device is shared) or a stream is being captured:
        # staged route for float64 weights -- the count -> fill pair on the indices alone (node filter + self loops) with
        # the kept edges' input positions, then the |w| > eps test on the gathered weights IN DOUBLE (this
        # used to raise)
        ei2, _, eid = filter_edges(edge_index, None, node_index, num_nodes, remove_self_loops, want_edge_id=True,
                                   views=views)
        w2 = w[eid]
        keep = w2.abs() > eps
        ei2, w2, eid = ei2[:, keep].contiguous(), w2[keep], eid[keep]
        return (ei2, w2, eid) if want_edge_id else (ei2, w2)
    ws = N.workspace(L.tgp_connect_subgraph_workspace_bytes(E, num_nodes), dev)
    d_count = torch.empty(1, dtype=torch.int64, device=dev)
    N.check(L.tgp_connect_subgraph_count(N.ptr(row), N.ptr(col), N.ptr(w), E, N.ptr(ni),
                                         0 if ni is None else ni.numel(), num_nodes, flags, eps, N.ptr(ws),
                                         ws.numel(), N.ptr(d_count), st), "tgp_connect_subgraph_count")
    n_out = _read_count(d_count)
    out_ei = torch.empty(2, n_out, dtype=torch.int64, device=dev)
    out_w = None if w is None else torch.empty(n_out, dtype=torch.float32, device=dev)
    out_id = torch.empty(n_out, dtype=torch.int64, device=dev) if want_edge_id else None
    N.check(L.tgp_connect_subgraph_fill(N.ptr(row), N.ptr(col), N.ptr(w), E, num_nodes, flags, eps, N.ptr(ws), n_out,
                                        N.ptr(out_ei[0]) if n_out else None,
                                        N.ptr(out_ei[1]) if n_out else None, N.ptr(out_w),
                                        N.ptr(out_id) if n_out else None, st),
            "tgp_connect_subgraph_fill")
    return (out_ei, out_w, out_id) if want_edge_id else (out_ei, out_w)


def _filter_edges_single(L, dev, st, row, col, w, E, ni, num_nodes, flags, eps, want_edge_id, views=False, md=None):
    """``tgp_connect_subgraph_single``; None when the kernel refused for a reason other than bad node ids (a look-back
    spin bound on a shared device): the caller takes the count -> fill pair."""
    ws = N.workspace(L.tgp_connect_subgraph_single_workspace_bytes(num_nodes), dev)
    cap = torch.empty(2, E, dtype=torch.int64, device=dev)
    cap_w = None if w is None else torch.empty(E, dtype=w.dtype, device=dev)
    cap_id = torch.empty(E, dtype=torch.int64, device=dev) if want_edge_id else None
    state = _sps_state(dev, st, L.tgp_connect_subgraph_single_status_words(E))
    epoch = state.next_epoch()
    cap_p = cap.data_ptr()
    entry = (L.tgp_connect_subgraph_single_f64 if (w is not None and w.dtype == torch.float64)
             else L.tgp_connect_subgraph_single)
    nblk = 0 if md is None else md.numel() // 5
    N.check(entry(row.data_ptr(), col.data_ptr(), N.ptr(w), E, N.ptr(ni),
                                          0 if ni is None else ni.numel(), num_nodes, flags, eps, ws.data_ptr(),
                                          ws.numel(), cap_p, cap_p + 8 * E, N.ptr(cap_w), N.ptr(cap_id),
                                          None if md is None else md.data_ptr(),
                                          None if md is None else md.data_ptr() + 16 * nblk,
                                          state.status.data_ptr(), state.status.numel(), state.pinned.data_ptr(), epoch,
                                          st), "tgp_connect_subgraph_single")
    total = state.wait(epoch)
    if total & 0x80000000:
        # refused: node ids outside [0, num_nodes) -- the reference's index ops raise for these inputs too; a chunk that
        # met one says so in word [1] of the status buffer, tagged with this call's epoch -- or a look-back spin bound
        word = int(state.status[1])
        if (word & 0xFFFFFFFF) == epoch:
            raise IndexError("edge_index holds node ids outside [0, num_nodes) (or cluster ids outside "
                             "[0, num_supernodes)): the reference's index ops raise for these inputs too")
        return None
    n_out = total & 0x7FFFFFFF
    if not views:  # exact-size outputs (the default): one launch moves the survivors out of the capacity buffers
        out_ei, out_w, out_id = _compact_edges(L, st, dev, cap_p, cap_p + 8 * E, N.ptr(cap_w),
                                               None if cap_w is None else cap_w.dtype, N.ptr(cap_id), n_out)
        return (out_ei, out_w, out_id) if want_edge_id else (out_ei, out_w)
    out_ei, out_w = cap[:, :n_out], None if cap_w is None else cap_w[:n_out]
    out_id = None if cap_id is None else cap_id[:n_out]
    if E * 16 > SPS_COMPACT_BYTES and 4 * n_out < E:  # mostly empty capacity buffers of a large list: exact copies
        out_ei = out_ei.contiguous()
        out_w = None if out_w is None else out_w.clone()
        out_id = None if out_id is None else out_id.clone()
    return (out_ei, out_w, out_id) if want_edge_id else (out_ei, out_w)


FUSED_MAX_SUPERNODES = 32 * 1024  # 1024 tiles of 32 supernode rows
# ... and while supernode rows are short on average: a row of more than 64 raw entries is sorted by the whole workgroup,
# one such row at a time, with its successors waiting in their look-back (a batch of denser mid-size graphs -- the
# reference's own timing harness, 27 entries per row on average -- took 1.45 ms in the fused kernel, 0.1 ms staged)
FUSED_MAX_AVG_ROW = 16


def coalesce_edges(edge_index: Tensor, edge_weight: Optional[Tensor], cluster_index: Tensor,
                   num_supernodes: int, reduce_op: str, remove_self_loops: bool,
                   eps_filter: bool = True, assign_index: Optional[AssignIndex] = None,
                   route: Optional[str] = None, csr: Optional[Tuple[Tensor, Tensor]] = None
                   ) -> Tuple[Tensor, Optional[Tensor]]:
    """cluster_index[edge_index] + PyG coalesce (connect/base_conn.py:86-89) fused with
    remove_self_loops and the |w| > eps filter (utils/ops.py:370-380).

    With the supernode->member index of the assignment at hand (``assign_index``) the sort-free row-local
    path is tried first; it declines (count = -1) for unsorted rows or very long supernode rows.
    The row-local path is first tried as ONE fused kernel (``tgp_connect_coalesce_fused_*``: survivors written at
    their final offsets through a decoupled look-back), then as the staged pipeline (any member count per supernode).
    ``csr`` = (row_ptr int32 [N+1], col int32 [E] or None) of exactly this row-sorted edge list (GraclusSelect builds
    the offsets): the pass over the row array is skipped; the fused kernel can also stream the 4-byte columns.
    ``route`` ("fused" / "staged" / "rows" / "grouped" / "general"; tests): take exactly that route, raise if it
    declines ("rows" = fused, then staged)."""
    if route not in (None, "fused", "staged", "rows", "grouped", "general"):
        raise ValueError(f"unknown route {route!r}")
    if reduce_op not in N.REDUCE_OPS:
        raise ValueError(f"unknown reduce_op '{reduce_op}', expected one of {sorted(N.REDUCE_OPS)}")
    dev = N.require_device(edge_index, edge_weight, cluster_index)
    row, col = _edge_rows(edge_index)
    E = row.numel()
    if edge_weight is not None and edge_weight.dtype == torch.float64:
        # fp64 weights are merged in fp64 (the reference's coalesce / scatter do).  r5: row-sorted input with the
        # assignment's member index at hand takes the row-local pipeline in double (no device-wide sort); anything else
        # (unsorted rows, a hub row, stream capture) the sort-based route, in double too
        if route not in (None, "general", "staged", "rows"):
            raise RuntimeError("float64 edge weights take the staged row-local or the general coalesce route")
        w = N.f64c(edge_weight.reshape(-1))
        cl = N.i64c(cluster_index)
        flags = (N.REMOVE_SELF_LOOPS if remove_self_loops else 0) | (N.EPS_FILTER if eps_filter else 0)
        L = N.lib()
        rowish = route in ("staged", "rows")
        if rowish and assign_index is None:
            assign_index = build_assign_index(cl, num_supernodes)
        if csr is not None and (csr[0].dtype != torch.int32 or csr[0].numel() != cl.numel() + 1
                                or not csr[0].is_contiguous()):
            raise ValueError("csr must be (int32 [N+1], int32 [E] or None) contiguous tensors of this edge list")
        rows_ok = (route != "general" and assign_index is not None and assign_index.nnz == cl.numel()
                   and assign_index.num_targets == num_supernodes and num_supernodes < (1 << 26) and E > 0
                   and not torch.cuda.is_current_stream_capturing()
                   and (rowish or csr is not None or _rows_sorted_memo(edge_index) is not False))
        if rows_ok:
            st = N.stream_ptr(dev)
            ws = N.workspace(L.tgp_connect_coalesce_rows_workspace_bytes_f64(E, cl.numel(), num_supernodes), dev)
            d_count = torch.empty(1, dtype=torch.int64, device=dev)
            state = _sps_state(dev, st, L.tgp_connect_coalesce_rows_count_status_words(num_supernodes, cl.numel()))
            epoch = state.next_epoch()
            N.check(L.tgp_connect_coalesce_rows_count_published_f64(
                N.ptr(row), N.ptr(col), None, N.ptr(w), E, N.ptr(cl), cl.numel(), num_supernodes,
                N.ptr(assign_index.row_ptr), N.ptr(assign_index.perm), N.ptr(csr[0]) if csr is not None else None,
                N.REDUCE_OPS[reduce_op], flags, ops_eps(), N.ptr(ws), ws.numel(), N.ptr(d_count),
                state.status.data_ptr(), state.status.numel(), state.pinned.data_ptr(), epoch, st),
                "tgp_connect_coalesce_rows_count_published_f64")
            n_out = _decode_count(state.wait(epoch))
            if n_out >= 0:
                out_ei = torch.empty(2, n_out, dtype=torch.int64, device=dev)
                out_w = torch.empty(n_out, dtype=torch.float64, device=dev)
                N.check(L.tgp_connect_coalesce_rows_fill_f64(N.ptr(ws), E, cl.numel(), num_supernodes, n_out,
                                                             N.ptr(out_ei[0]) if n_out else None,
                                                             N.ptr(out_ei[1]) if n_out else None,
                                                             N.ptr(out_w) if n_out else None, st),
                        "tgp_connect_coalesce_rows_fill_f64")
                return out_ei, out_w
            del ws
            if rowish:
                raise RuntimeError("row-local coalesce route declined (unsorted rows or a supernode row too long)")
            if n_out == -1 and csr is None and _rows_sorted_memo(edge_index) is None:
                _rows_sorted(edge_index, row)  # remember an unsorted list: later calls skip the attempt
        elif rowish:
            raise RuntimeError("row-local coalesce route not applicable")
        ws = N.workspace(L.tgp_connect_coalesce_workspace_bytes_f64(E, cl.numel(), num_supernodes), dev)
        d_count = torch.empty(1, dtype=torch.int64, device=dev)
        st = N.stream_ptr(dev)
        N.check(L.tgp_connect_coalesce_count_f64(N.ptr(row), N.ptr(col), N.ptr(w), E, N.ptr(cl), cl.numel(),
                                                 num_supernodes, N.REDUCE_OPS[reduce_op], flags, ops_eps(), N.ptr(ws),
                                                 ws.numel(), N.ptr(d_count), st), "tgp_connect_coalesce_count_f64")
        n_out = _read_count(d_count)
        out_ei = torch.empty(2, n_out, dtype=torch.int64, device=dev)
        out_w = torch.empty(n_out, dtype=torch.float64, device=dev)
        N.check(L.tgp_connect_coalesce_fill_f64(N.ptr(ws), E, cl.numel(), num_supernodes, 1, flags, n_out,
                                                N.ptr(out_ei[0]) if n_out else None,
                                                N.ptr(out_ei[1]) if n_out else None, N.ptr(out_w) if n_out else None, st),
                "tgp_connect_coalesce_fill_f64")
        return out_ei, out_w
    w = None if edge_weight is None else N.f32c(edge_weight.reshape(-1))
    cl = N.i64c(cluster_index)
    flags = (N.REMOVE_SELF_LOOPS if remove_self_loops else 0) | (N.EPS_FILTER if (w is not None and eps_filter) else 0)
    L = N.lib()
    eps = ops_eps()
    if csr is not None:
        cptr, ccol = csr
        if (cptr.dtype != torch.int32 or cptr.numel() != cl.numel() + 1 or not cptr.is_contiguous()
                or (ccol is not None and (ccol.dtype != torch.int32 or ccol.numel() != E or not ccol.is_contiguous()))):
            raise ValueError("csr must be (int32 [N+1], int32 [E] or None) contiguous tensors of this edge list")
    rowish = route in ("fused", "staged", "rows")
    if rowish and assign_index is None:
        assign_index = build_assign_index(cl, num_supernodes)
    rows_ok = (route is None or rowish) and assign_index is not None and assign_index.nnz == cl.numel() \
        and assign_index.num_targets == num_supernodes and num_supernodes < (1 << 26) \
        and (rowish or csr is not None or _rows_sorted_memo(edge_index) is not False)
    staged_ok = rows_ok
    # the fused kernel's workgroups wait for their predecessors: it is used while all tiles are resident at once
    # (<= 1024 of them: one launch wave), i.e. for batches of small graphs, where the staged pipeline's ten launches
    # dominate; large lists take the staged pipeline, whose kernels never wait for each other
    fused_ok = rows_ok and (route == "fused" or (num_supernodes <= FUSED_MAX_SUPERNODES
                                                  and E <= FUSED_MAX_AVG_ROW * num_supernodes))
    if fused_ok and route != "staged":
        ws = N.workspace(L.tgp_connect_coalesce_fused_workspace_bytes(E, cl.numel(), num_supernodes), dev)
        d_count = torch.empty(1, dtype=torch.int64, device=dev)
        cap_w = None if w is None else torch.empty(max(E, 1), dtype=torch.float32, device=dev)
        st = N.stream_ptr(dev)
        cptr = ccol = None
        if csr is not None:
            cptr, ccol = csr
        N.check(L.tgp_connect_coalesce_fused_count(N.ptr(row), N.ptr(col), N.ptr(cptr), N.ptr(ccol), N.ptr(w), E,
                                                   N.ptr(cl), cl.numel(), num_supernodes,
                                                   N.ptr(assign_index.row_ptr), N.ptr(assign_index.perm),
                                                   N.REDUCE_OPS[reduce_op], flags, eps, N.ptr(cap_w), N.ptr(ws),
                                                   ws.numel(), N.ptr(d_count), st), "tgp_connect_coalesce_fused_count")
        n_out = _read_count(d_count)
        if n_out >= 0:
            out_ei = torch.empty(2, n_out, dtype=torch.int64, device=dev)
            N.check(L.tgp_connect_coalesce_fused_fill(N.ptr(ws), E, cl.numel(), num_supernodes, n_out,
                                                      N.ptr(out_ei[0]) if n_out else None,
                                                      N.ptr(out_ei[1]) if n_out else None, st),
                    "tgp_connect_coalesce_fused_fill")
            out_w = None
            if w is not None:  # the kernel wrote the weights at their final offsets of the capacity-E buffer
                out_w = cap_w[:n_out] if 2 * n_out >= E else cap_w[:n_out].clone()
            return out_ei, out_w
        del ws, cap_w
        if route == "fused":
            raise RuntimeError(f"fused coalesce route declined (code {n_out})")
        # -3: only the fused kernel's tile limit; -1 on a list whose rows ARE sorted: a supernode row too long for the
        # fused kernel -- the staged pipeline takes both (it sorts hub rows device-wide)
        if n_out != -3 and E > 1 and csr is None and _rows_sorted_memo(edge_index) is None:
            _rows_sorted(edge_index, row)
        staged_ok = n_out == -3 or (n_out == -1 and (csr is not None or _rows_sorted_memo(edge_index) is True))
    if staged_ok:
        # a list known to hold hub rows (a supernode row beyond 1024 raw entries) asks for the huge-row kernels at once;
        # any other list finds out from the count (-5), once per edge_index object
        hub = _HUB_ROWS.get(id(edge_index))
        hub = hub is not None and hub[0]() is edge_index and hub[1] == edge_index._version
        published = _PUBLISH_COUNTS and not torch.cuda.is_current_stream_capturing()
        for attempt in range(2):
            fl = flags | (N.HUGE_ROWS if hub else 0)
            nbytes = (L.tgp_connect_coalesce_rows_huge_workspace_bytes if hub else
                      L.tgp_connect_coalesce_rows_workspace_bytes)(E, cl.numel(), num_supernodes)
            ws = N.workspace(nbytes, dev)
            st = N.stream_ptr(dev)
            d_count = torch.empty(1, dtype=torch.int64, device=dev)
            if published:
                # r4: the survivor scan is one look-back launch whose last workgroup stores the count into a pinned host
                # word: no scan pair, no copy kernel, no stream synchronise; int32 columns when Select's CSR holds them
                state = _sps_state(dev, st, L.tgp_connect_coalesce_rows_count_status_words(num_supernodes, cl.numel()))
                epoch = state.next_epoch()
                N.check(L.tgp_connect_coalesce_rows_count_published(
                    N.ptr(row), N.ptr(col), N.ptr(csr[1]) if csr is not None else None, N.ptr(w), E, N.ptr(cl),
                    cl.numel(), num_supernodes, N.ptr(assign_index.row_ptr), N.ptr(assign_index.perm),
                    N.ptr(csr[0]) if csr is not None else None, N.REDUCE_OPS[reduce_op], fl, eps, N.ptr(ws),
                    ws.numel(), N.ptr(d_count), state.status.data_ptr(), state.status.numel(),
                    state.pinned.data_ptr(), epoch, st), "tgp_connect_coalesce_rows_count_published")
                n_out = _decode_count(state.wait(epoch))
            else:
                N.check(L.tgp_connect_coalesce_rows_count(N.ptr(row), N.ptr(col), N.ptr(w), E, N.ptr(cl), cl.numel(),
                                                          num_supernodes, N.ptr(assign_index.row_ptr),
                                                          N.ptr(assign_index.perm),
                                                          N.ptr(csr[0]) if csr is not None else None,
                                                          N.REDUCE_OPS[reduce_op], fl, eps, N.ptr(ws),
                                                          ws.numel(), N.ptr(d_count), st), "tgp_connect_coalesce_rows_count")
                n_out = _read_count(d_count)
            if n_out != -5 or hub:
                break
            hub = True
            _remember_hub_rows(edge_index)
            del ws
        if n_out >= 0:
            out_ei = torch.empty(2, n_out, dtype=torch.int64, device=dev)
            out_w = None if w is None else torch.empty(n_out, dtype=torch.float32, device=dev)
            N.check(L.tgp_connect_coalesce_rows_fill(N.ptr(ws), E, cl.numel(), num_supernodes,
                                                     (0 if w is None else 1) | (2 if hub else 0),
                                                     n_out, N.ptr(out_ei[0]) if n_out else None,
                                                     N.ptr(out_ei[1]) if n_out else None, N.ptr(out_w), st),
                    "tgp_connect_coalesce_rows_fill")
            return out_ei, out_w
        del ws  # declined: fall through to the sort-based path
        if rowish:
            raise RuntimeError("row-local coalesce route declined (unsorted rows or a supernode row too long)")
        if E > 1 and _rows_sorted_memo(edge_index) is None:
            # remember WHY for this tensor object (one comparison pass, once): an unsorted list skips the row-local
            # attempt (~50 us + a host round trip) on every later call
            _rows_sorted(edge_index, row)
    if rowish:
        raise RuntimeError("row-local coalesce route declined or not applicable")
    if route == "grouped" or (route is None and 65536 < num_supernodes < (1 << 26)):
        # more than 32 bits of (row, col) key: sort by supernode row only (half the radix passes) and order the short
        # rows in LDS; declines (count = -1) when a supernode row is too long for that
        ws = N.workspace(L.tgp_connect_coalesce_grouped_workspace_bytes(E, cl.numel(), num_supernodes), dev)
        d_count = torch.empty(1, dtype=torch.int64, device=dev)
        st = N.stream_ptr(dev)
        N.check(L.tgp_connect_coalesce_grouped_count(N.ptr(row), N.ptr(col), N.ptr(w), E, N.ptr(cl), cl.numel(),
                                                     num_supernodes, N.REDUCE_OPS[reduce_op], flags, eps, N.ptr(ws),
                                                     ws.numel(), N.ptr(d_count), st), "tgp_connect_coalesce_grouped_count")
        n_out = _read_count(d_count)
        if n_out >= 0:
            out_ei = torch.empty(2, n_out, dtype=torch.int64, device=dev)
            out_w = None if w is None else torch.empty(n_out, dtype=torch.float32, device=dev)
            N.check(L.tgp_connect_coalesce_rows_fill(N.ptr(ws), E, cl.numel(), num_supernodes, 0 if w is None else 1,
                                                     n_out, N.ptr(out_ei[0]) if n_out else None,
                                                     N.ptr(out_ei[1]) if n_out else None, N.ptr(out_w), st),
                    "tgp_connect_coalesce_rows_fill")
            return out_ei, out_w
        del ws
        if route == "grouped":
            raise RuntimeError("grouped coalesce route declined (a supernode row too long for the in-LDS sort)")
    ws = N.workspace(L.tgp_connect_coalesce_workspace_bytes(E, cl.numel(), num_supernodes), dev)
    d_count = torch.empty(1, dtype=torch.int64, device=dev)
    st = N.stream_ptr(dev)
    N.check(L.tgp_connect_coalesce_count(N.ptr(row), N.ptr(col), N.ptr(w), E, N.ptr(cl), cl.numel(),
                                         num_supernodes, N.REDUCE_OPS[reduce_op], flags, eps, N.ptr(ws), ws.numel(),
                                         N.ptr(d_count), st), "tgp_connect_coalesce_count")
    n_out = _read_count(d_count)
    out_ei = torch.empty(2, n_out, dtype=torch.int64, device=dev)
    out_w = None if w is None else torch.empty(n_out, dtype=torch.float32, device=dev)
    N.check(L.tgp_connect_coalesce_fill(N.ptr(ws), E, cl.numel(), num_supernodes, 0 if w is None else 1, flags, n_out,
                                        N.ptr(out_ei[0]) if n_out else None,
                                        N.ptr(out_ei[1]) if n_out else None, N.ptr(out_w), st),
            "tgp_connect_coalesce_fill")
    return out_ei, out_w


def normalize_edges_(edge_index: Tensor, edge_weight: Tensor, num_nodes: int, degree_norm: bool,
                     edge_weight_norm: bool, batch_pooled: Optional[Tensor], num_graphs: int) -> Tensor:
    """In-place D^-1/2 A D^-1/2 and per-graph max-abs normalisation (utils/ops.py:383-417)."""
    dev = N.require_device(edge_index, edge_weight, batch_pooled)
    row, col = _edge_rows(edge_index)
    flags = (N.DEGREE_NORM if degree_norm else 0) | (N.EDGE_WEIGHT_NORM if edge_weight_norm else 0)
    if flags == 0 or row.numel() == 0:
        return edge_weight
    bp = None if batch_pooled is None else N.i64c(batch_pooled)
    L = N.lib()
    if edge_weight.dtype == torch.float64:
        ws = N.workspace(L.tgp_postprocess_sparse_workspace_bytes_f64(row.numel(), num_nodes, num_graphs), dev)
        N.check(L.tgp_postprocess_sparse_norm_f64(N.ptr(row), N.ptr(col), N.ptr(edge_weight), row.numel(), num_nodes,
                                                  flags, ops_eps(), N.ptr(bp), num_graphs, N.ptr(ws), ws.numel(),
                                                  N.stream_ptr(dev)), "tgp_postprocess_sparse_norm_f64")
        return edge_weight
    ws = N.workspace(L.tgp_postprocess_sparse_workspace_bytes(row.numel(), num_nodes, num_graphs), dev)
    N.check(L.tgp_postprocess_sparse_norm_f32(N.ptr(row), N.ptr(col), N.ptr(edge_weight), row.numel(), num_nodes,
                                              flags, ops_eps(), N.ptr(bp), num_graphs, N.ptr(ws), ws.numel(),
                                              N.stream_ptr(dev)), "tgp_postprocess_sparse_norm_f32")
    return edge_weight


# ------------------------------------------------------------------------- A3 / A7 / A8
def _any_f64(*ts) -> bool:
    """Does any operand carry float64?  Then the product runs on the fp64 matrix path (r5: the reference's
    torch.matmul computes model.double() inputs in fp64, base_reduce.py:158-161, dense_conn.py:111-122)."""
    return any(isinstance(t, Tensor) and t.dtype == torch.float64 for t in ts)


def dense_flags(remove_self_loops: bool, degree_norm: bool, adj_transpose: bool, edge_weight_norm: bool) -> int:
    return ((N.REMOVE_SELF_LOOPS if remove_self_loops else 0) | (N.DEGREE_NORM if degree_norm else 0)
            | (N.SUM_AXIS_ROWS if adj_transpose else 0) | (N.EDGE_WEIGHT_NORM if edge_weight_norm else 0))


def _dense_adj_layout(adj: Tensor, dtype: torch.dtype = torch.float32) -> Tuple[Tensor, int]:
    """Accept A [B,N,N] as contiguous or as the transposed view produced by
    DenseSRCPooling.preprocessing (src.py:442-443) without materialising the transpose."""
    if adj.dtype != dtype:
        adj = adj.to(dtype)
    if adj.is_contiguous():
        return adj, 0
    t = adj.transpose(-1, -2)
    if t.is_contiguous():
        return t, N.ADJ_TRANSPOSED
    return adj.contiguous(), 0


def _out_buffer(out: Optional[Tensor], shape, dev, dtype: torch.dtype = torch.float32) -> Tensor:
    if out is None:
        return torch.empty(shape, dtype=dtype, device=dev)
    if tuple(out.shape) != tuple(shape) or out.dtype != dtype or out.device != dev or not out.is_contiguous():
        raise ValueError(f"output buffer must be a contiguous {dtype} {tuple(shape)} tensor on {dev}, got "
                         f"{tuple(out.shape)} {out.dtype} on {out.device}")
    return out


def _dense_pool_f64(s, adj, x, flags, want_raw, want_post, out_x, out_adj):
    """float64 operands: the same fused A3 + A7 + A8 call on v_mfma_f64_16x16x4_f64 (tgp_dense_pool_f64)."""
    dev = N.require_device(s, adj, x)
    s = N.f64c(s)
    B, Nn, K = s.shape
    F = 0
    x_pool = adj_raw = adj_pool = None
    if x is not None:
        x = N.f64c(x)
        if x.shape[:2] != (B, Nn):
            raise ValueError(f"x {tuple(x.shape)} does not match s {tuple(s.shape)}")
        F = x.size(2)
        x_pool = _out_buffer(out_x, (B, K, F), dev, torch.float64)
    a = None
    if adj is not None:
        if adj.shape != (B, Nn, Nn):
            raise ValueError(f"adj {tuple(adj.shape)} does not match s {tuple(s.shape)}")
        a, tflag = _dense_adj_layout(adj, torch.float64)
        flags |= tflag
        if want_raw:
            adj_raw = torch.empty(B, K, K, dtype=torch.float64, device=dev)
        if want_post:
            adj_pool = _out_buffer(out_adj, (B, K, K), dev, torch.float64)
    L = N.lib()
    ws = N.workspace(L.tgp_dense_pool_workspace_bytes_f64(B, Nn, K, F), dev)
    N.check(L.tgp_dense_pool_f64(N.ptr(s), N.ptr(a), N.ptr(x), B, Nn, K, F, flags, ops_eps(), N.ptr(x_pool),
                                 N.ptr(adj_raw), N.ptr(adj_pool), N.ptr(ws), ws.numel(), N.stream_ptr(dev)),
            "tgp_dense_pool_f64")
    return x_pool, adj_raw, adj_pool


def dense_pool(s: Tensor, adj: Optional[Tensor], x: Optional[Tensor], flags: int = 0, want_raw: bool = False,
               want_post: bool = True, graph_sizes: Optional[Tensor] = None, out_x: Optional[Tensor] = None,
               out_adj: Optional[Tensor] = None, mincut_terms: bool = False, diff_stats: bool = False):
    """(x_pool, adj_raw, adj_pool) = (S^T X, S^T A S, postprocess(S^T A S)) for a padded batch
    (reduce/base_reduce.py:158-161, connect/dense_conn.py:111-122, utils/ops.py:282-335).  ``graph_sizes`` [B]
    (optional): real nodes per graph when they are the leading rows and the padding is zero (to_dense_batch layout).
    ``mincut_terms``: return a fourth value, the [2,B] per-graph tails of MinCut's losses taken inside the pooling
    kernel (batches of small graphs only; None when the batch takes another kernel).  float64 operands (any of the
    three) run the fp64 form of the same call; ``graph_sizes`` is a speed hint the fp64 form does not use."""
    if _any_f64(s, adj, x):
        out = _dense_pool_f64(s, adj, x, flags, want_raw, want_post, out_x, out_adj)
        return out + (None,) if mincut_terms else out
    dev = N.require_device(s, adj, x)
    s = N.f32c(s)
    B, Nn, K = s.shape
    F = 0
    x_pool = adj_raw = adj_pool = None
    if x is not None:
        x = N.f32c(x)
        if x.shape[:2] != (B, Nn):
            raise ValueError(f"x {tuple(x.shape)} does not match s {tuple(s.shape)}")
        F = x.size(2)
        x_pool = _out_buffer(out_x, (B, K, F), dev)  # caller-provided: e.g. a slot of an all-gather send buffer
    a = None
    if adj is not None:
        if adj.shape != (B, Nn, Nn):
            raise ValueError(f"adj {tuple(adj.shape)} does not match s {tuple(s.shape)}")
        a, tflag = _dense_adj_layout(adj)
        flags |= tflag
        if want_raw:
            adj_raw = torch.empty(B, K, K, dtype=torch.float32, device=dev)
        if want_post:
            adj_pool = _out_buffer(out_adj, (B, K, K), dev)
    L = N.lib()
    if diff_stats:
        # (r6) a fourth value: DiffPool's per-graph records [B,4] from the one-wave-per-graph kernel (:func:`diffpool_stats_tail`),
        # or None when the batch does not take that kernel (the caller computes the losses from the adjacency)
        if (a is None or x is None or out_x is not None or out_adj is not None or not (want_raw or want_post)
                or not L.tgp_dense_pool_is_small(B, Nn, K, F)):
            return dense_pool(s, adj, x, flags & ~N.ADJ_TRANSPOSED, want_raw, want_post, graph_sizes, out_x, out_adj) + (None,)
        stats = torch.empty(B, 4, dtype=torch.float32, device=dev)
        N.check(L.tgp_dense_pool_small_diff_f32(N.ptr(s), N.ptr(a), N.ptr(x), None, None, None, B, Nn, K, F, flags, ops_eps(),
                                                losses_eps(), None, N.ptr(x_pool), N.ptr(adj_raw), N.ptr(adj_pool),
                                                N.ptr(stats), None, N.stream_ptr(dev)), "tgp_dense_pool_small_diff_f32")
        return x_pool, adj_raw, adj_pool, stats
    ws = N.workspace(L.tgp_dense_pool_workspace_bytes(B, Nn, K, F), dev)
    if mincut_terms:
        terms = None
        if a is not None and L.tgp_dense_pool_is_small(B, Nn, K, F):
            terms = torch.empty(2, B, dtype=torch.float32, device=dev)
            N.check(L.tgp_dense_pool_mincut_f32(N.ptr(s), N.ptr(a), N.ptr(x), B, Nn, K, F, flags, ops_eps(),
                                                losses_eps(), N.ptr(x_pool), N.ptr(adj_raw), N.ptr(adj_pool),
                                                N.ptr(terms), N.ptr(ws), ws.numel(), N.stream_ptr(dev)),
                    "tgp_dense_pool_mincut_f32")
            return x_pool, adj_raw, adj_pool, terms
    gs = _sizes_arg(graph_sizes, B, dev)
    N.check(L.tgp_dense_pool_f32(N.ptr(s), N.ptr(a), N.ptr(x), B, Nn, K, F, flags, ops_eps(), N.ptr(gs), N.ptr(x_pool),
                                 N.ptr(adj_raw), N.ptr(adj_pool), N.ptr(ws), ws.numel(), N.stream_ptr(dev)),
            "tgp_dense_pool_f32")
    return (x_pool, adj_raw, adj_pool, None) if mincut_terms else (x_pool, adj_raw, adj_pool)


def dense_pool_select(x: Tensor, adj: Tensor, weight: Tensor, bias: Optional[Tensor], mask: Optional[Tensor], flags: int,
                      want_raw: bool = False, mincut_terms: bool = False, want_batch: bool = False,
                      diff_stats: bool = False):
    """(s, x_pool, adj_raw, adj_pool, terms): MLPSelect's last Linear + softmax + mask, Reduce, Connect and the
    post-processing of a batch of small graphs in ONE launch (select/mlp_select.py:105-147, base_reduce.py:158-161,
    dense_conn.py:111-122, utils/ops.py:282-335); callers check :func:`dense_pool_is_small` first.  ``want_batch``: a
    sixth value, the pooled batch vector ``arange(B).repeat_interleave(K)`` (utils/ops.py:152-169) written by the same
    launch."""
    dev = N.require_device(x, adj, weight, bias, mask)
    x, weight = N.f32c(x), N.f32c(weight)
    B, Nn, F = x.shape
    K = weight.size(0)
    if weight.shape != (K, F) or adj.shape != (B, Nn, Nn):
        raise ValueError(f"dense_pool_select: shapes x {tuple(x.shape)}, adj {tuple(adj.shape)}, weight {tuple(weight.shape)}")
    a, tflag = _dense_adj_layout(adj)
    b = None if bias is None else N.f32c(bias)
    m = None
    if mask is not None:
        m = mask.to(torch.uint8) if mask.dtype != torch.uint8 and mask.dtype != torch.bool else mask
        m = m.contiguous().view(torch.uint8) if m.dtype == torch.bool else m.contiguous()
        if tuple(m.shape) != (B, Nn):
            raise ValueError(f"dense_pool_select: mask {tuple(mask.shape)} does not match x {tuple(x.shape)}")
    s = torch.empty(B, Nn, K, dtype=torch.float32, device=dev)
    x_pool = torch.empty(B, K, F, dtype=torch.float32, device=dev)
    adj_pool = torch.empty(B, K, K, dtype=torch.float32, device=dev)
    adj_raw = torch.empty(B, K, K, dtype=torch.float32, device=dev) if want_raw else None
    terms = torch.empty(2, B, dtype=torch.float32, device=dev) if mincut_terms else None
    bp = torch.empty(B * K, dtype=torch.int64, device=dev) if want_batch else None
    if diff_stats:  # (r6) DiffPool: `terms` holds the per-graph records [B,4] of :func:`diffpool_stats_tail` instead
        terms = torch.empty(B, 4, dtype=torch.float32, device=dev)
        N.check(N.lib().tgp_dense_pool_small_diff_f32(None, N.ptr(a), N.ptr(x), N.ptr(weight), N.ptr(b), N.ptr(m), B, Nn, K,
                                                      F, flags | tflag, ops_eps(), losses_eps(), N.ptr(s), N.ptr(x_pool),
                                                      N.ptr(adj_raw), N.ptr(adj_pool), N.ptr(terms), N.ptr(bp),
                                                      N.stream_ptr(dev)), "tgp_dense_pool_small_diff_f32")
        return (s, x_pool, adj_raw, adj_pool, terms, bp) if want_batch else (s, x_pool, adj_raw, adj_pool, terms)
    N.check(N.lib().tgp_dense_pool_select_f32(N.ptr(x), N.ptr(a), N.ptr(weight), N.ptr(b), N.ptr(m), B, Nn, K, F,
                                              flags | tflag, ops_eps(), losses_eps(), N.ptr(s), N.ptr(x_pool),
                                              N.ptr(adj_raw), N.ptr(adj_pool), N.ptr(terms), N.ptr(bp),
                                              N.stream_ptr(dev)),
            "tgp_dense_pool_select_f32")
    if want_batch:
        return s, x_pool, adj_raw, adj_pool, terms, bp
    return s, x_pool, adj_raw, adj_pool, terms


def edge_facts_launch(edge_index: Tensor, batch: Tensor):
    """Enqueue ``tgp_edge_facts_sorted_i64`` for a NEW row-sorted edge list: the per-graph edge ranges land in a device
    buffer the consumer can be launched on at once, the verdict (rows sorted?) in a pinned host word that
    :func:`edge_facts_finish` reads afterwards.  None when there is nothing to look at (no entries) or a stream is being
    captured."""
    row, _ = _edge_rows(edge_index)
    E, n = row.numel(), batch.numel()
    if E == 0 or n == 0 or torch.cuda.is_current_stream_capturing():
        return None
    dev = edge_index.device
    st = N.stream_ptr(dev)
    state = _sps_state(dev, st, 0)
    buf = torch.empty(n + 2, dtype=torch.long, device=dev)
    tag = state.next_facts_tag()
    N.check(N.lib().tgp_edge_facts_sorted_i64(N.ptr(row), E, N.ptr(N.i64c(batch)), n, N.ptr(buf),
                                              state.ticket.data_ptr() + 8, state.facts_slot(tag), tag, st),
            "tgp_edge_facts_sorted_i64")
    return state, tag, buf


def edge_facts_finish(handle, edge_index: Tensor, graph_ptr: Tensor) -> bool:
    """The verdict of :func:`edge_facts_launch`: True = the rows are grouped by ascending source node and the ranges in the
    buffer are exact (both are remembered for this tensor object: later calls need no launch), False = not (remembered
    as well: later calls skip the attempt)."""
    state, tag, buf = handle
    _, flags, _, _, _ = state.wait_facts(tag)
    if flags:
        _remember_rows_sorted(edge_index, False)
        return False
    _remember_rows_sorted(edge_index, True)
    _edge_ptr_remember(edge_index, graph_ptr, buf[: graph_ptr.numel()])
    return True


_COALESCED: dict = {}  # id(edge_index) -> (weakref, version, num_nodes): the list is strictly row-major sorted (no duplicates)


def coalesced_memo(edge_index: Tensor, num_nodes: int) -> bool:
    hit = _COALESCED.get(id(edge_index))
    return bool(hit is not None and hit[0]() is edge_index and hit[1] == edge_index._version and hit[2] == num_nodes)


def remember_coalesced(edge_index: Tensor, num_nodes: int) -> None:
    import weakref
    if len(_COALESCED) >= 16:
        for key in [k for k, v in _COALESCED.items() if v[0]() is None]:
            del _COALESCED[key]
        while len(_COALESCED) >= 16:
            del _COALESCED[next(iter(_COALESCED))]
    _COALESCED[id(edge_index)] = (weakref.ref(edge_index), edge_index._version, num_nodes)


_ADJ_SYMMETRIC: dict = {}  # id(edge_index) -> (weakref, version, weakref of the weights or None, its version, flag)


def _adj_symmetric_memo(edge_index: Tensor, edge_weight: Optional[Tensor]) -> Optional[bool]:
    hit = _ADJ_SYMMETRIC.get(id(edge_index))
    if (hit is not None and hit[0]() is edge_index and hit[1] == edge_index._version
            and ((hit[2] is None and edge_weight is None)
                 or (hit[2] is not None and hit[2]() is edge_weight and hit[3] == edge_weight._version))):
        return hit[4]
    return None


def _remember_adj_symmetric(edge_index: Tensor, edge_weight: Optional[Tensor], flag: bool) -> None:
    import weakref
    if len(_ADJ_SYMMETRIC) >= 16:
        for key in [k for k, v in _ADJ_SYMMETRIC.items() if v[0]() is None]:
            del _ADJ_SYMMETRIC[key]
        while len(_ADJ_SYMMETRIC) >= 16:
            del _ADJ_SYMMETRIC[next(iter(_ADJ_SYMMETRIC))]
    _ADJ_SYMMETRIC[id(edge_index)] = (weakref.ref(edge_index), edge_index._version,
                                      None if edge_weight is None else weakref.ref(edge_weight),
                                      None if edge_weight is None else edge_weight._version, flag)


class AdjSymmetry:
    """Is the dense adjacency ``to_dense_adj`` built from this edge list symmetric?  Asked in the forward of the dense
    poolers' training step (one launch over the entries, tgp_adj_symmetry_f32, verdict into a pinned host word),
    answered in its backward -- hundreds of microseconds later, so the answer is simply there.  A symmetric A makes
    V = A^T S equal to U = A S: the backward then runs ONE N^2 K product less.  Remembered per (edge_index, edge_weight)
    tensor objects.  ``get()`` = True only when the answer is known to be yes."""

    __slots__ = ("state", "tag", "answer", "ei", "ew")

    def __init__(self, edge_index: Tensor, edge_weight: Optional[Tensor], adj: Tensor, batch: Tensor, ptr: Tensor):
        import weakref
        self.state = self.tag = None
        self.ei, self.ew = weakref.ref(edge_index), (None if edge_weight is None else weakref.ref(edge_weight))
        self.answer = _adj_symmetric_memo(edge_index, edge_weight)
        if self.answer is not None:
            return
        row, col = _edge_rows(edge_index)
        E = row.numel()
        if (E == 0 or adj.dim() != 3 or adj.size(1) != adj.size(2) or not adj.is_contiguous() or adj.dtype != torch.float32
                or torch.cuda.is_current_stream_capturing()):
            self.answer = E == 0 and not torch.cuda.is_current_stream_capturing()
            return
        dev = adj.device
        st = N.stream_ptr(dev)
        self.state = _sps_state(dev, st, 0)
        self.tag = self.state.next_facts_tag()
        N.check(N.lib().tgp_adj_symmetry_f32(N.ptr(row), N.ptr(col), E, N.ptr(N.i64c(batch)), N.ptr(N.i64c(ptr)),
                                             adj.size(1), N.ptr(adj), self.state.ticket.data_ptr() + 8,
                                             self.state.facts_slot(self.tag), self.tag, st), "tgp_adj_symmetry_f32")

    @classmethod
    def of_dense(cls, adj: Tensor) -> "AdjSymmetry":
        """The same question for a dense [B,N,N] float32 adjacency the caller holds (tgp_dense_symmetry_f32: one pass over
        the matrix); remembered for this tensor object + version, so a fixed dense graph pooled every epoch pays it once."""
        import weakref
        self = cls.__new__(cls)
        self.state = self.tag = None
        self.ei, self.ew = weakref.ref(adj), None
        self.answer = _adj_symmetric_memo(adj, None)
        if self.answer is not None:
            return self
        if (adj.dim() != 3 or adj.size(1) != adj.size(2) or adj.dtype != torch.float32 or not adj.is_cuda
                or not adj.is_contiguous() or adj.numel() == 0 or torch.cuda.is_current_stream_capturing()):
            self.answer = False
            return self
        dev = adj.device
        st = N.stream_ptr(dev)
        self.state = _sps_state(dev, st, 0)
        self.tag = self.state.next_facts_tag()
        N.check(N.lib().tgp_dense_symmetry_f32(N.ptr(adj), adj.size(0), adj.size(1), self.state.ticket.data_ptr() + 8,
                                               self.state.facts_slot(self.tag), self.tag, st), "tgp_dense_symmetry_f32")
        return self

    @classmethod
    def of_edge_list(cls, key_index: Tensor, key_weight: Optional[Tensor], edge_index: Tensor,
                     edge_weight: Optional[Tensor], row_ptr: Tensor, num_nodes: int) -> "AdjSymmetry":
        """The same question for a COALESCED row-sorted list with its CSR offsets (tgp_edge_symmetry_f32: every entry must
        have a mirror entry of equal weight); remembered for the (key_index, key_weight) objects the caller holds."""
        import weakref
        self = cls.__new__(cls)
        self.state = self.tag = None
        self.ei, self.ew = weakref.ref(key_index), (None if key_weight is None else weakref.ref(key_weight))
        self.answer = _adj_symmetric_memo(key_index, key_weight)
        if self.answer is not None:
            return self
        row, col = _edge_rows(edge_index)
        E = row.numel()
        if E == 0 or torch.cuda.is_current_stream_capturing():
            self.answer = E == 0 and not torch.cuda.is_current_stream_capturing()
            return self
        dev = edge_index.device
        st = N.stream_ptr(dev)
        self.state = _sps_state(dev, st, 0)
        self.tag = self.state.next_facts_tag()
        w = None if edge_weight is None else N.f32c(edge_weight.reshape(-1))
        N.check(N.lib().tgp_edge_symmetry_f32(N.ptr(row), N.ptr(col), N.ptr(w), E, N.ptr(row_ptr), num_nodes,
                                              self.state.ticket.data_ptr() + 8, self.state.facts_slot(self.tag), self.tag,
                                              st), "tgp_edge_symmetry_f32")
        return self

    def get(self) -> bool:
        if self.answer is None:
            flags = self.state.peek_facts(self.tag)
            self.answer = flags == 0 if flags is not None else False  # (slot reused by later calls: not known)
            if flags is not None:
                ei, ew = self.ei(), (None if self.ew is None else self.ew())
                if ei is not None and (self.ew is None or ew is not None):
                    _remember_adj_symmetric(ei, ew, self.answer)
        return bool(self.answer)


def dense_pool_select_sparse(x: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor], batch: Tensor,
                             node_ptr: Tensor, edge_ptr: Tensor, num_graphs: int, max_nodes: int, weight: Tensor,
                             bias: Optional[Tensor], flags: int, adj_transpose: bool, want_raw: bool = False,
                             mincut_terms: bool = False, want_dense: bool = False, diff_stats: bool = False):
    """(s [B,N,K], mask [B,N], x_pool, adj_raw, adj_pool, terms, batch_pool): :func:`dense_pool_select` straight from
    the un-padded batch -- x [Ntot,F], a ROW-SORTED ``edge_index`` with the per-graph ranges ``node_ptr`` / ``edge_ptr``
    [B+1] -- in ONE launch: the adjacency tiles are built in LDS from the edges, neither ``to_dense_batch`` nor
    ``to_dense_adj`` runs, no [B,N,N] tensor exists (src.py:434-450 + select/mlp_select.py:105-147 + the fused Reduce /
    Connect call).  The caller has checked the row order (``_rows_sorted``) and the ranges (``graph_edge_ptr``)."""
    dev = N.require_device(x, edge_index, edge_weight, batch, node_ptr, edge_ptr, weight, bias)
    x, weight = N.f32c(x), N.f32c(weight)
    F = x.size(1)
    K, B, Nn = weight.size(0), int(num_graphs), int(max_nodes)
    if weight.shape != (K, F) or batch.numel() != x.size(0) or node_ptr.numel() != B + 1 or edge_ptr.numel() < B + 1:
        raise ValueError("dense_pool_select_sparse: inconsistent shapes")
    row, col = _edge_rows(edge_index)
    E = row.numel()
    w = None if edge_weight is None else N.f32c(edge_weight.reshape(-1))
    b = None if bias is None else N.f32c(bias)
    s = torch.empty(B, Nn, K, dtype=torch.float32, device=dev)
    mask = torch.empty(B, Nn, dtype=torch.bool, device=dev)
    x_pool = torch.empty(B, K, F, dtype=torch.float32, device=dev)
    adj_pool = torch.empty(B, K, K, dtype=torch.float32, device=dev)
    adj_raw = torch.empty(B, K, K, dtype=torch.float32, device=dev) if want_raw else None
    terms = torch.empty(2, B, dtype=torch.float32, device=dev) if mincut_terms else None
    bp = torch.empty(B * K, dtype=torch.int64, device=dev)
    # want_dense (training): also the zero-padded x [B,N,F] and the adjacency [B,N,N] the backward kernels read
    # (want_dense="adj": the adjacency alone -- DiffPool's losses read it in inference)
    xd = torch.empty(B, Nn, F, dtype=torch.float32, device=dev) if want_dense is True else None
    ad = torch.empty(B, Nn, Nn, dtype=torch.float32, device=dev) if want_dense else None
    # diff_stats (r6): a LAST value [B,4] = per-graph (sum A^2, trace(S^T A S), |S^T S|_F^2, entropy sum): DiffPool's two
    # losses without the dense adjacency (:func:`diffpool_stats_tail`)
    dstats = torch.empty(B, 4, dtype=torch.float32, device=dev) if diff_stats else None
    N.check(N.lib().tgp_dense_pool_select_sparse_f32(
        N.ptr(x), x.size(0), N.ptr(row) if E else None, N.ptr(col) if E else None, N.ptr(w), E, N.ptr(N.i64c(batch)),
        N.ptr(N.i64c(node_ptr)), N.ptr(N.i64c(edge_ptr)), N.ptr(weight), N.ptr(b), B, Nn, K, F, flags,
        1 if adj_transpose else 0, ops_eps(), losses_eps(), N.ptr(s), mask.data_ptr(), N.ptr(x_pool), N.ptr(adj_raw),
        N.ptr(adj_pool), N.ptr(terms), N.ptr(bp), N.ptr(xd), N.ptr(ad), N.ptr(dstats), N.stream_ptr(dev)),
        "tgp_dense_pool_select_sparse_f32")
    tail = (dstats,) if diff_stats else ()
    if want_dense:
        return (s, mask, x_pool, adj_raw, adj_pool, terms, bp, xd, ad) + tail
    return (s, mask, x_pool, adj_raw, adj_pool, terms, bp) + tail


def diffpool_stats_tail(stats: Tensor, link_scale: float, ent_scale: float) -> Tensor:
    """[2]: DiffPool's (link loss * link_scale, entropy sum * ent_scale) from the [B,4] records of
    ``dense_pool_select_sparse(diff_stats=True)`` (utils/losses.py:644-658, 476-483), one launch."""
    dev = N.require_device(stats)
    stats = N.f32c(stats)
    out = torch.empty(2, dtype=torch.float32, device=dev)
    N.check(N.lib().tgp_diffpool_stats_tail_f32(N.ptr(stats), stats.size(0), float(link_scale), float(ent_scale),
                                                N.ptr(out), N.stream_ptr(dev)), "tgp_diffpool_stats_tail_f32")
    return out


def dense_pool_is_small(B: int, Nn: int, K: int, F: int) -> bool:
    """Does the one-wave-per-graph kernel take this padded batch (and so its fused backward)?"""
    return bool(N.lib().tgp_dense_pool_is_small(B, Nn, K, F))


def _bcast_or_dense(t: Optional[Tensor], shape) -> Tuple[Optional[Tensor], bool]:
    """An upstream gradient as the kernel takes it: (contiguous fp32 tensor, False), or (one-element tensor, True) when
    every stride is zero -- `loss = out.sum()` sends an expanded scalar, which is read as one value instead of being
    materialised as a [B,K,F] copy."""
    if t is None:
        return None, False
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f"upstream gradient {tuple(t.shape)} does not match {tuple(shape)}")
    if t.dtype == torch.float32 and t.numel() > 1 and not any(t.stride()):
        return t.as_strided((1,), (1,)), True
    return N.f32c(t), False


def dense_pool_small_bwd(s: Tensor, adj: Tensor, x: Optional[Tensor], flags: int, g_x_pool: Optional[Tensor],
                         g_adj_pool: Optional[Tensor], g_adj_raw: Optional[Tensor], g_terms: Optional[Tensor],
                         want_gx: bool = True, g_diff: Optional[Tensor] = None, diff_losses: Optional[Tensor] = None,
                         link_scale: float = 0.0, ent_scale: float = 0.0, g_mean_terms=(None, None),
                         g_diff_pair=(None, None)):
    """Gradients (gS, gX) of ``dense_pool`` (with its in-kernel MinCut terms) for a batch of small graphs, one launch:
    base_reduce.py:158-161, dense_conn.py:111-122, utils/ops.py:282-335, utils/losses.py:39-70 under autograd.
    ``g_mean_terms`` = upstream gradients (0-dim tensors or None) of the batch MEANS of the two per-graph terms;
    ``g_diff_pair`` = those of DiffPool's (link, entropy) losses as two 0-dim tensors (``g_diff`` [2] is the other form)."""
    dev = N.require_device(s, adj, x)
    s = N.f32c(s)
    B, Nn, K = s.shape
    a, tflag = _dense_adj_layout(adj)
    F = 0
    if x is not None:
        x = N.f32c(x)
        F = x.size(2)

    def grad(t, shape):
        if t is None:
            return None
        t = N.f32c(t)
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"upstream gradient {tuple(t.shape)} does not match {tuple(shape)}")
        return t

    g_x_pool, bx = _bcast_or_dense(g_x_pool, (B, K, F)) if x is not None else (None, False)
    g_adj_pool, ba = _bcast_or_dense(g_adj_pool, (B, K, K))
    g_adj_raw = grad(g_adj_raw, (B, K, K))
    g_terms = grad(g_terms, (2, B))
    g_link, g_ent = g_diff_pair
    if g_diff is not None:
        g_diff = grad(g_diff, (2,))
        g_link, g_ent = g_diff[0], g_diff[1]
    g_link, g_ent = grad(g_link, ()), grad(g_ent, ())
    g_cut, g_ortho = grad(g_mean_terms[0], ()), grad(g_mean_terms[1], ())
    diff_losses = grad(diff_losses, (2,)) if (g_link is not None or g_ent is not None) else None
    gs = torch.empty(B, Nn, K, dtype=torch.float32, device=dev)
    gx = torch.empty(B, Nn, F, dtype=torch.float32, device=dev) if (want_gx and x is not None) else None
    N.check(N.lib().tgp_dense_pool_small_bwd_f32(N.ptr(s), N.ptr(a), N.ptr(x), B, Nn, K, F, flags | tflag, ops_eps(),
                                                 losses_eps(), N.ptr(g_x_pool), N.ptr(g_adj_pool), N.ptr(g_adj_raw),
                                                 N.ptr(g_terms), N.ptr(g_cut), N.ptr(g_ortho), N.ptr(g_link),
                                                 N.ptr(g_ent), N.ptr(diff_losses), float(link_scale),
                                                 float(ent_scale), losses_eps(), (1 if bx else 0) | (2 if ba else 0),
                                                 N.ptr(gs), N.ptr(gx), N.stream_ptr(dev)),
            "tgp_dense_pool_small_bwd_f32")
    return gs, gx


def mlp_select_bwd_fits(K: int, F: int) -> bool:
    return bool(N.lib().tgp_mlp_select_bwd_fits(F, K))


def mlp_select_bwd(s: Tensor, g_s: Tensor, x: Tensor, weight: Tensor, want_gx: bool = True, want_gw: bool = True,
                   want_gb: bool = True, gx_accumulate: Optional[Tensor] = None):
    """(gx, gw, gb) of ``mlp_select`` in one pass over S, dS and X + a tiny combine of the workgroups' partial gw / gb
    (select/mlp_select.py:139-145 under autograd; K <= 32, F <= 64):
    dY = S (dS - <dS,S>), gx = dY W -- added IN PLACE to ``gx_accumulate`` when that is given (the gradient the pooling
    backward already holds for the same X) --, gw = dY^T X, gb = column sums of dY."""
    dev = N.require_device(s, g_s, x, weight)
    s, g_s, x, weight = N.f32c(s), N.f32c(g_s), N.f32c(x), N.f32c(weight)
    K = s.size(-1)
    F = x.size(-1)
    M = s.numel() // max(K, 1)
    if g_s.shape != s.shape or x.numel() != M * F or weight.shape != (K, F):
        raise ValueError(f"mlp_select_bwd: shapes s {tuple(s.shape)}, g_s {tuple(g_s.shape)}, x {tuple(x.shape)}, "
                         f"weight {tuple(weight.shape)}")
    gx = None
    if gx_accumulate is not None:
        if (gx_accumulate.dtype != torch.float32 or not gx_accumulate.is_contiguous()
                or gx_accumulate.numel() != M * F):
            raise ValueError("mlp_select_bwd: gx_accumulate must be a contiguous float32 tensor of x's size")
        gx = gx_accumulate
    elif want_gx:
        gx = torch.empty_like(x)
    gw = torch.empty(K, F, dtype=torch.float32, device=dev) if (want_gw or want_gb) else None
    gb = torch.empty(K, dtype=torch.float32, device=dev) if want_gb else None
    L = N.lib()
    st = N.stream_ptr(dev)
    ws = None
    if gw is not None:
        ws = N.workspace(L.tgp_mlp_select_bwd_workspace_bytes(M, F, K), dev)
    N.check(L.tgp_mlp_select_bwd_f32(N.ptr(s), N.ptr(g_s), N.ptr(x), N.ptr(weight), M, F, K, N.ptr(gx),
                                     1 if gx_accumulate is not None else 0, N.ptr(gw), N.ptr(gb), N.ptr(ws),
                                     0 if ws is None else ws.numel(), st), "tgp_mlp_select_bwd_f32")
    return gx, (gw if want_gw else None), gb


def postprocess_dense(adj_pool: Tensor, flags: int, inplace: bool = False) -> Tensor:
    """utils/ops.py:282-335 on a [B,K,K] tensor."""
    dev = N.require_device(adj_pool)
    if adj_pool.dtype == torch.float64:  # a float64 pooled adjacency is post-processed in fp64 (utils/ops.py:282-335)
        src = N.f64c(adj_pool)
        dst = src if inplace else torch.empty_like(src)
        B, K = src.size(0), src.size(1)
        L = N.lib()
        ws = N.workspace(L.tgp_postprocess_dense_workspace_bytes_f64(B, K), dev)
        N.check(L.tgp_postprocess_dense_f64(N.ptr(src), N.ptr(dst), B, K, flags, ops_eps(), N.ptr(ws), ws.numel(),
                                            N.stream_ptr(dev)), "tgp_postprocess_dense_f64")
        return dst
    src = N.f32c(adj_pool)
    dst = src if inplace else torch.empty_like(src)
    B, K = src.size(0), src.size(1)
    L = N.lib()
    ws = N.workspace(L.tgp_postprocess_dense_workspace_bytes(B, K), dev)
    N.check(L.tgp_postprocess_dense_f32(N.ptr(src), N.ptr(dst), B, K, flags, ops_eps(), N.ptr(ws), ws.numel(),
                                        N.stream_ptr(dev)), "tgp_postprocess_dense_f32")
    return dst


def postprocess_dense_bwd(raw: Tensor, g_post: Tensor, flags: int) -> Optional[Tensor]:
    """Gradient of :func:`postprocess_dense` with respect to its input (utils/ops.py:282-335 under autograd), one
    launch; None when the kernel does not take the case (K > 4096)."""
    dev = N.require_device(raw, g_post)
    raw = N.f32c(raw)
    B, K = raw.size(0), raw.size(1)
    if K > 4096 or tuple(g_post.shape) != tuple(raw.shape):
        return None
    # an expanded scalar (the gradient of a plain `.sum()`) is handed over as one value: no [B,K,K] copy of it (r6)
    g_post, one = _bcast_or_dense(g_post, raw.shape)
    out = torch.empty_like(raw)
    N.check(N.lib().tgp_postprocess_dense_bwd_f32(N.ptr(raw), N.ptr(g_post), B, K, flags | ((1 << 16) if one else 0),
                                                  ops_eps(), N.ptr(out), N.stream_ptr(dev)), "tgp_postprocess_dense_bwd_f32")
    return out


def entropy_bwd(s: Tensor, g: Tensor, scale: float = 1.0) -> Tensor:
    """-(log(s + eps) + s / (s + eps)) * g * scale, g a 0-d device tensor (utils/losses.py:476-483 under autograd)."""
    dev = N.require_device(s, g)
    s32 = N.f32c(s)
    out = torch.empty_like(s32)
    N.check(N.lib().tgp_entropy_bwd_f32(N.ptr(s32), s32.numel(), losses_eps(), N.ptr(N.f32c(g.reshape(1))), float(scale),
                                        N.ptr(out), N.stream_ptr(dev)), "tgp_entropy_bwd_f32")
    return out


def _sizes_arg(graph_sizes: Optional[Tensor], num_graphs: int, dev) -> Optional[Tensor]:
    if graph_sizes is None:
        return None
    gs = N.i64c(graph_sizes)
    if gs.numel() != num_graphs or gs.device != dev:
        raise ValueError(f"graph_sizes must hold one count per graph on {dev}, got {tuple(gs.shape)} on {gs.device}")
    return gs


def link_loss_sq(s: Tensor, adj: Tensor, graph_sizes: Optional[Tensor] = None) -> Tensor:
    """sq[b] = ||adj[b] - s[b] s[b]^T||_F^2 without materialising s s^T (utils/losses.py:644-708).  float64: s s^T on
    the fp64 matrix path and the residual as the reference writes it (losses.py:644-652)."""
    dev = N.require_device(s, adj)
    if _any_f64(s, adj):
        s64 = N.f64c(s)
        sst = bmm(s64, s64.transpose(1, 2).contiguous())
        return ((adj.to(torch.float64) - sst) ** 2).sum(dim=(1, 2))
    s, adj = N.f32c(s), N.f32c(adj)
    B, Nn, K = s.shape
    if adj.shape != (B, Nn, Nn):
        raise ValueError(f"adj {tuple(adj.shape)} does not match s {tuple(s.shape)}")
    sq = torch.empty(B, dtype=torch.float32, device=dev)
    L = N.lib()
    ws = N.workspace(L.tgp_link_loss_workspace_bytes(B, Nn, K), dev)
    N.check(L.tgp_link_loss_f32(N.ptr(s), N.ptr(adj), B, Nn, K, N.ptr(_sizes_arg(graph_sizes, B, dev)), N.ptr(sq),
                                N.ptr(ws), ws.numel(), N.stream_ptr(dev)), "tgp_link_loss_f32")
    return sq


def diffpool_loss_tail(s: Tensor, adj: Tensor, graph_sizes: Optional[Tensor], link_scale: float,
                       ent_scale: float) -> Tensor:
    """[2]: DiffPool's link-prediction loss sqrt(sum_b ||adj_b - s_b s_b^T||^2) * link_scale and entropy loss
    sum(-s log(s + eps)) * ent_scale (poolers/diffpool.py:262-284) -- the two native partial reductions and ONE tail
    launch (outside autograd)."""
    dev = N.require_device(s, adj)
    sq = link_loss_sq(s, adj, graph_sizes)
    s32 = N.f32c(s)
    L = N.lib()
    ws = N.workspace(L.tgp_entropy_sum_workspace_bytes(s32.numel()), dev)
    import ctypes as _ct
    n_partial = _ct.c_int(0)
    st = N.stream_ptr(dev)
    N.check(L.tgp_entropy_partials_f32(N.ptr(s32), s32.numel(), losses_eps(), N.ptr(ws), ws.numel(),
                                       _ct.addressof(n_partial), st), "tgp_entropy_partials_f32")
    out = torch.empty(2, dtype=torch.float32, device=dev)
    N.check(L.tgp_diffpool_loss_tail_f32(N.ptr(sq), sq.numel(), N.ptr(ws), n_partial.value, float(link_scale),
                                         float(ent_scale), N.ptr(out), st), "tgp_diffpool_loss_tail_f32")
    return out


def entropy_sum(s: Tensor) -> Tensor:
    """0-d tensor sum(-s log(s + eps)) over every element (utils/losses.py:476-483 before / num_nodes)."""
    dev = N.require_device(s)
    if s.dtype == torch.float64:  # one elementwise pass + a sum, in double like the reference's
        return (-(s * torch.log(s + losses_eps()))).sum()
    s = N.f32c(s)
    out = torch.empty((), dtype=torch.float32, device=dev)
    L = N.lib()
    ws = N.workspace(L.tgp_entropy_sum_workspace_bytes(s.numel()), dev)
    N.check(L.tgp_entropy_sum_f32(N.ptr(s), s.numel(), losses_eps(), N.ptr(out), N.ptr(ws), ws.numel(),
                                  N.stream_ptr(dev)),
            "tgp_entropy_sum_f32")
    return out


def cut_terms(adj: Tensor, s: Tensor, graph_sizes: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """(deg [B,N], q [B,N], den [B]): row sums of adj, squared row norms of s, trace(s^T D s)
    (utils/losses.py:39-81)."""
    dev = N.require_device(adj, s)
    if _any_f64(adj, s):
        a64, s64 = adj.to(torch.float64), s.to(torch.float64)
        deg, q = a64.sum(-1), (s64 * s64).sum(-1)
        return deg, q, (deg * q).sum(-1)
    adj, s = N.f32c(adj), N.f32c(s)
    B, Nn, K = s.shape
    if adj.shape != (B, Nn, Nn):
        raise ValueError(f"adj {tuple(adj.shape)} does not match s {tuple(s.shape)}")
    deg = torch.empty(B, Nn, dtype=torch.float32, device=dev)
    q = torch.empty(B, Nn, dtype=torch.float32, device=dev)
    den = torch.empty(B, dtype=torch.float32, device=dev)
    N.check(N.lib().tgp_cut_terms_f32(N.ptr(adj), N.ptr(s), B, Nn, K, N.ptr(_sizes_arg(graph_sizes, B, dev)), N.ptr(deg),
                                      N.ptr(q), N.ptr(den), N.stream_ptr(dev)), "tgp_cut_terms_f32")
    return deg, q, den


def cut_rows(adj: Tensor, s: Tensor, graph_sizes: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """(deg [B,N], q [B,N]) of :func:`cut_terms` without the per-graph dot product (one launch: the training step's fused
    loss tail forms den itself, :func:`mincut_terms_fused`)."""
    dev = N.require_device(adj, s)
    adj, s = N.f32c(adj), N.f32c(s)
    B, Nn, K = s.shape
    if adj.shape != (B, Nn, Nn):
        raise ValueError(f"adj {tuple(adj.shape)} does not match s {tuple(s.shape)}")
    deg = torch.empty(B, Nn, dtype=torch.float32, device=dev)
    q = torch.empty(B, Nn, dtype=torch.float32, device=dev)
    N.check(N.lib().tgp_cut_terms_f32(N.ptr(adj), N.ptr(s), B, Nn, K, N.ptr(_sizes_arg(graph_sizes, B, dev)), N.ptr(deg),
                                      N.ptr(q), None, N.stream_ptr(dev)), "tgp_cut_terms_f32")
    return deg, q


def mincut_terms_fused(raw: Tensor, gram: Tensor, deg: Optional[Tensor], q: Optional[Tensor],
                       ptr: Optional[Tensor] = None, want_means: bool = False, edges=None):
    """(den [B], terms [2,B], stats [B,4][, means [2]]): MinCut's per-graph loss tails with den = sum_i deg_i q_i formed
    in the same launch (utils/losses.py:39-70); stats = (trace(raw), |G|^2, trace(G), |Y|) per graph, the scalars the
    backward's right-hand sides need; ``want_means``: also the batch means of the two terms (what the pooler hands out)."""
    dev = N.require_device(raw, gram, deg, q)
    raw, gram = N.f32c(raw), N.f32c(gram)
    deg = None if deg is None else N.f32c(deg)
    q = None if q is None else N.f32c(q)  # (None: deg already carries the factor, den = sum of deg)
    # edges = (row_ptr int32 [Ntot+1], edge_index, weights or None) of an un-padded batch: den = sum_e w_e q[col_e], the
    # in-degree form (S^T A^T S of the batched poolers on the rows route); deg is then not read
    e_rp = e_col = e_w = None
    if edges is not None:
        if ptr is None or q is None:
            raise ValueError("mincut_terms_fused(edges=...) needs ptr and q")
        e_rp, e_ei, e_w = edges
        e_col = _edge_rows(e_ei)[1]
        e_w = None if e_w is None else N.f32c(e_w.reshape(-1))
    B, Kc, Nn = raw.size(0), raw.size(-1), (deg if deg is not None else q).size(-1)
    den = torch.empty(B, dtype=torch.float32, device=dev)
    out = torch.empty(2, B, dtype=torch.float32, device=dev)
    stats = torch.empty(B, 4, dtype=torch.float32, device=dev)
    st = N.stream_ptr(dev)
    means = ticket = None
    if want_means and B <= 256 and not torch.cuda.is_current_stream_capturing():
        # the batch means come with the same launch (its last workgroup adds the terms up in graph order); for thousands
        # of graphs the arrival tickets on ONE word cost more than the reduction launch they save (2048 graphs: +40 us)
        means = torch.empty(2, dtype=torch.float32, device=dev)
        ticket = _sps_state(dev, st, 0).ticket.data_ptr() + 16
    # ptr: deg / q belong to an un-padded batch (graph b owns entries ptr[b] .. ptr[b+1])
    N.check(N.lib().tgp_mincut_terms_fused_f32(N.ptr(raw), N.ptr(gram), N.ptr(deg), N.ptr(q), B, Nn, Kc, losses_eps(),
                                               N.ptr(den), N.ptr(out), N.ptr(stats),
                                               N.ptr(None if ptr is None else N.i64c(ptr)), ticket, N.ptr(means),
                                               N.ptr(e_rp), N.ptr(e_col), N.ptr(e_w), st),
            "tgp_mincut_terms_fused_f32")
    if want_means:
        return den, out, stats, (means if means is not None else out.mean(dim=1))
    return den, out, stats


def dense_pool_train_fwd(s: Tensor, adj_mem: Tensor, x: Tensor, flags: int, acat: Tensor, want_gram: bool,
                         want_post: bool = True):
    """(x_pool, raw, adj_pool, gram): forward of the dense poolers' training step beyond the one-wave kernels
    (base_reduce.py:158-161, dense_conn.py:111-122, utils/ops.py:282-335).  ``adj_mem`` is the contiguous adjacency
    memory (``flags`` carries ``ADJ_TRANSPOSED`` when it holds A^T), ``acat`` [B,N,3K+F+4] the operand buffer of the
    backward whose first K columns receive U = A S; ``gram`` = S^T S when asked for."""
    dev = N.require_device(s, adj_mem, x, acat)
    B, Nn, Kc = s.shape
    F = x.size(2)
    ld = acat.size(2)
    if (adj_mem.shape != (B, Nn, Nn) or x.shape[:2] != (B, Nn) or acat.shape[:2] != (B, Nn) or ld < Kc
            or not (s.is_contiguous() and adj_mem.is_contiguous() and x.is_contiguous() and acat.is_contiguous())):
        raise ValueError("dense_pool_train_fwd: inconsistent shapes or non-contiguous operands")
    x_pool = torch.empty(B, Kc, F, dtype=torch.float32, device=dev)
    raw = torch.empty(B, Kc, Kc, dtype=torch.float32, device=dev)
    adj_pool = torch.empty(B, Kc, Kc, dtype=torch.float32, device=dev) if want_post else None
    gram = torch.empty(B, Kc, Kc, dtype=torch.float32, device=dev) if want_gram else None
    L = N.lib()
    ws = N.workspace(L.tgp_dense_pool_train_workspace_bytes(B, Nn, Kc, F), dev)
    N.check(L.tgp_dense_pool_train_fwd_f32(N.ptr(s), N.ptr(adj_mem), N.ptr(x), B, Nn, Kc, F, flags, ops_eps(),
                                           N.ptr(acat), ld, N.ptr(x_pool), N.ptr(raw), N.ptr(adj_pool), N.ptr(gram),
                                           N.ptr(ws), ws.numel(), N.stream_ptr(dev)), "tgp_dense_pool_train_fwd_f32")
    return x_pool, raw, adj_pool, gram


def dense_pool_train_rhs(g_raw_a: Optional[Tensor], g_raw_b: Optional[Tensor], mode: int, stats: Optional[Tensor],
                         den: Optional[Tensor], gram: Optional[Tensor], g_la: Optional[Tensor], g_lb: Optional[Tensor],
                         scale: float, link_loss: Optional[Tensor], link_scale: float, g_x: Optional[Tensor],
                         gx_bcast: bool, symmetric: bool, weight: Optional[Tensor], B: int, Kc: int, F: int, dev):
    """(rcat [B,3K+F+4,K], c1 [B] or None, gw [B,2K,F] or None): the right-hand sides [RU ; RX ; 0 ; RS ; RV] of the
    training step's backward GEMM and, with the selector's ``weight``, [g_x ; W] for gX = [S | dY] [g_x ; W]
    (see tgp_dense_pool_train_rhs_f32 in include/tgp_hip.h).  ``symmetric``: bool, or the flag word (bit 0: A = A^T,
    bit 1: the RU / RV slots trade places)."""
    rcat = torch.empty(B, 3 * Kc + F + TRAIN_PAD, Kc, dtype=torch.float32, device=dev)
    c1 = torch.empty(B, dtype=torch.float32, device=dev) if mode == 1 else None
    gw = torch.empty(B, 2 * Kc, F, dtype=torch.float32, device=dev) if weight is not None else None
    w = None if weight is None else N.f32c(weight.detach())
    N.check(N.lib().tgp_dense_pool_train_rhs_f32(N.ptr(g_raw_a), N.ptr(g_raw_b), mode, N.ptr(stats), N.ptr(den),
                                                 N.ptr(gram), N.ptr(g_la), N.ptr(g_lb), float(scale), N.ptr(link_loss),
                                                 float(link_scale), losses_eps(), N.ptr(g_x), 1 if gx_bcast else 0,
                                                 int(symmetric), N.ptr(w), B, Kc, F, N.ptr(rcat), N.ptr(c1),
                                                 N.ptr(gw), N.stream_ptr(dev)), "tgp_dense_pool_train_rhs_f32")
    return rcat, c1, gw


def softmax_bwd_ex(s: Tensor, ds: Tensor, extra: Optional[Tensor] = None, c1: Optional[Tensor] = None,
                   deg: Optional[Tensor] = None, ent_g: Optional[Tensor] = None, ent_scale: float = 0.0,
                   out: Optional[Tensor] = None, batch: Optional[Tensor] = None) -> Tensor:
    """:func:`softmax_bwd` on dS + extra + 2 c1[graph] deg[row] S - ent_g ent_scale (log(S + eps) + S / (S + eps)):
    the elementwise parts of the pooling step's gradient folded into the selector's softmax backward.  s [B,N,K];
    ``out``: a [B,N,K] float32 view with unit last stride and uniform row stride (a column block of a wider buffer)."""
    dev = N.require_device(s, ds)
    if s.dim() == 2:  # un-padded batch [Ntot,K]: the graph of a row comes from ``batch`` (one graph without it)
        B, (Nn, Kc) = 1, s.shape
    else:
        B, Nn, Kc = s.shape
    s2, d2 = N.f32c(s), N.f32c(ds)
    ex = None if extra is None else N.f32c(extra)
    if d2.numel() != s2.numel() or (ex is not None and ex.numel() != s2.numel()):
        raise ValueError("softmax_bwd_ex: gradient shapes do not match s")
    if out is None:
        out = torch.empty_like(s2)
    elif (out.shape != s2.shape or out.dtype != torch.float32 or out.stride(-1) != 1
          or (out.dim() == 3 and out.stride(0) != Nn * out.stride(1))):
        raise ValueError("softmax_bwd_ex: out must be a float32 view of s's shape with a uniform row stride")
    N.check(N.lib().tgp_softmax_bwd_ex_f32(N.ptr(s2), N.ptr(d2), N.ptr(ex), N.ptr(None if c1 is None else N.f32c(c1)),
                                           N.ptr(None if deg is None else N.f32c(deg)), Nn,
                                           N.ptr(None if ent_g is None else N.f32c(ent_g.reshape(1))), float(ent_scale),
                                           losses_eps(), out.data_ptr(), out.stride(-2), B * Nn, Kc,
                                           N.ptr(None if batch is None else N.i64c(batch)), N.stream_ptr(dev)),
            "tgp_softmax_bwd_ex_f32")
    return out


TRAIN_PAD = 4  # width of the [1 0 0 0] column block of the training step's operand buffer (csrc/losses.hip)


def copy_cols2(a: Tensor, b: Tensor, dst: Tensor, col_a: int, col_b: int, one_col: int = -1) -> None:
    """dst[:, col_a:col_a+wa] = a, dst[:, col_b:col_b+wb] = b for row-major 2-D float32 tensors, one launch;
    ``one_col >= 0``: also dst[:, one_col:one_col+4] = [1 0 0 0]."""
    dev = N.require_device(a, b, dst)
    if not (a.is_contiguous() and b.is_contiguous() and dst.is_contiguous()) or a.size(0) != dst.size(0) \
            or b.size(0) != dst.size(0):
        raise ValueError("copy_cols2: operands must be contiguous with equal row counts")
    N.check(N.lib().tgp_copy_cols2_f32(N.ptr(a), a.size(1), N.ptr(b), b.size(1), dst.size(0), N.ptr(dst), dst.size(1),
                                       col_a, col_b, one_col, N.stream_ptr(dev)), "tgp_copy_cols2_f32")


def copy_cols3(a: Tensor, b: Tensor, c: Tensor, dst: Tensor, col_a: int, col_b: int, col_c: int, one_col: int = -1) -> None:
    """:func:`copy_cols2` with a third source block: dst[:, col_c:col_c+wc] = c."""
    dev = N.require_device(a, b, c, dst)
    if not all(t.is_contiguous() and t.size(0) == dst.size(0) for t in (a, b, c)) or not dst.is_contiguous():
        raise ValueError("copy_cols3: operands must be contiguous with equal row counts")
    N.check(N.lib().tgp_copy_cols3_f32(N.ptr(a), a.size(1), N.ptr(b), b.size(1), N.ptr(c), c.size(1), dst.size(0),
                                       N.ptr(dst), dst.size(1), col_a, col_b, col_c, one_col, N.stream_ptr(dev)),
            "tgp_copy_cols3_f32")


def slab_sum_split(part: Tensor, F: int, want_gw: bool = True, want_gb: bool = True):
    """(gw [K,F], gb [K]) from part [slabs,K,W]: columns 0..F-1 and column F of the slab sum as two contiguous tensors."""
    dev = N.require_device(part)
    slabs, Kc, W = part.shape
    gw = torch.empty(Kc, F, dtype=torch.float32, device=dev) if want_gw else None
    gb = torch.empty(Kc, dtype=torch.float32, device=dev) if want_gb else None
    N.check(N.lib().tgp_slab_sum_split_f32(N.ptr(part), slabs, Kc, F, W, N.ptr(gw), N.ptr(gb), N.stream_ptr(dev)),
            "tgp_slab_sum_split_f32")
    return gw, gb


def bmm_into(a: Tensor, b: Tensor, out: Tensor, trans_a: bool = False, accumulate: bool = False) -> Tensor:
    """out[g] (+)= op(a[g]) @ b[g] for float32 3-D VIEWS: any batch / row strides, unit stride along the last dimension
    (column blocks of a wider buffer: no copies in, no copy out)."""
    dev = N.require_device(a, b, out)
    for t in (a, b, out):
        if t.dim() != 3 or t.dtype != torch.float32 or t.stride(2) != 1:
            raise ValueError("bmm_into: float32 3-D views with unit last stride expected")
    G = out.size(0)
    Kd, M = (a.size(1), a.size(2)) if trans_a else (a.size(2), a.size(1))
    Nc = b.size(2)
    if b.size(1) != Kd or out.shape[1:] != (M, Nc) or a.size(0) not in (1, G) or b.size(0) not in (1, G):
        raise ValueError(f"bmm_into: shapes {tuple(a.shape)} x {tuple(b.shape)} -> {tuple(out.shape)}")
    sA = 0 if a.size(0) == 1 else a.stride(0)
    sB = 0 if b.size(0) == 1 else b.stride(0)
    fn = N.lib().tgp_bmm_accumulate_f32 if accumulate else N.lib().tgp_bmm_f32
    N.check(fn(a.data_ptr(), b.data_ptr(), out.data_ptr(), G, M, Nc, Kd, 1 if trans_a else 0, a.stride(1), b.stride(1),
               out.stride(1), sA, sB, out.stride(0), N.stream_ptr(dev)), "tgp_bmm_f32")
    return out


def mincut_loss_terms(raw: Tensor, den: Tensor, gram: Tensor) -> Tensor:
    """[2,B]: per-graph -trace(raw)/(den + eps) and ||G/||G|| - I/sqrt(K)||_F (utils/losses.py:39-70), one launch."""
    dev = N.require_device(raw, den, gram)
    raw, den, gram = N.f32c(raw), N.f32c(den), N.f32c(gram)
    B, Kc = raw.size(0), raw.size(-1)
    out = torch.empty(2, B, dtype=torch.float32, device=dev)
    N.check(N.lib().tgp_mincut_loss_terms_f32(N.ptr(raw), N.ptr(den), N.ptr(gram), B, Kc, losses_eps(), N.ptr(out),
                                              N.stream_ptr(dev)), "tgp_mincut_loss_terms_f32")
    return out


def mincut_loss_terms_bwd(raw: Tensor, den: Tensor, gram: Tensor, g_terms: Tensor):
    """(g_raw [B,K,K], c1 [B], W [B,K,K]): gradients of :func:`mincut_loss_terms` with respect to raw, den and gram from
    the upstream gradients [2,B] (utils/losses.py:39-70 under autograd), one launch."""
    dev = N.require_device(raw, den, gram, g_terms)
    raw, den, gram, g_terms = N.f32c(raw), N.f32c(den), N.f32c(gram), N.f32c(g_terms)
    B, Kc = raw.size(0), raw.size(-1)
    g_raw = torch.empty_like(raw)
    W = torch.empty_like(raw)
    c1 = torch.empty(B, dtype=torch.float32, device=dev)
    N.check(N.lib().tgp_mincut_loss_terms_bwd_f32(N.ptr(raw), N.ptr(den), N.ptr(gram), N.ptr(g_terms), B, Kc, losses_eps(),
                                                  N.ptr(g_raw), N.ptr(c1), N.ptr(W), N.stream_ptr(dev)),
            "tgp_mincut_loss_terms_bwd_f32")
    return g_raw, c1, W


def topk_minscore(score: Tensor, ptr: Tensor, min_score: float, tol: float = 1e-7) -> Tuple[Tensor, Tensor]:
    """(prob [N], node_index [k]): per-graph softmax of ``score`` and the nodes above the min_score threshold, ascending
    (select/topk_select.py:186-194 with PyG's softmax / topk); ``ptr`` = node offsets of the sorted batch."""
    dev = N.require_device(score, ptr)
    score, ptr = N.f32c(score.reshape(-1)), N.i64c(ptr)
    n, B = score.numel(), ptr.numel() - 1
    prob = torch.empty(n, dtype=torch.float32, device=dev)
    L = N.lib()
    ws = N.workspace(L.tgp_topk_minscore_workspace_bytes(n, B), dev)
    d_count = torch.empty(1, dtype=torch.int64, device=dev)
    st = N.stream_ptr(dev)
    N.check(L.tgp_topk_minscore_count(N.ptr(score), N.ptr(ptr), n, B, float(min_score), float(tol), N.ptr(prob),
                                      N.ptr(ws), ws.numel(), N.ptr(d_count), st), "tgp_topk_minscore_count")
    k = _read_count(d_count)
    node_index = torch.empty(k, dtype=torch.int64, device=dev)
    N.check(L.tgp_topk_minscore_fill(N.ptr(ws), N.ptr(ptr), n, B, k, N.ptr(node_index) if k else None, st),
            "tgp_topk_minscore_fill")
    return prob, node_index


def topk_plan(sizes: Tensor, ratio: float) -> Tuple[Tensor, Tensor]:
    """(k [B], koff [B+1]): nodes kept per graph (PyG topk: ceil(ratio * n) in fp32, or min(ratio, n)) and their
    exclusive prefix sums, in one launch."""
    dev = N.require_device(sizes)
    sizes = N.i64c(sizes)
    B = sizes.numel()
    k = torch.empty(B, dtype=torch.int64, device=dev)
    koff = torch.empty(B + 1, dtype=torch.int64, device=dev)
    N.check(N.lib().tgp_topk_plan(N.ptr(sizes), B, float(ratio), N.ptr(k), N.ptr(koff), N.stream_ptr(dev)),
            "tgp_topk_plan")
    return k, koff


def topk_select(score: Tensor, batch: Optional[Tensor], num_graphs: int, ptr: Tensor, k: Tensor, koff: Tensor,
                k_total: int, segments_max_nodes: int = 0, with_values: bool = False, with_lift: bool = False):
    """Per-graph top-k (select/topk_select.py:194 -> PyG ``topk``) fused with the row sort of SelectOutput
    (select/base_select.py:58): (index [2, k_total] = node_index ascending over cluster_index, supernode -> assignment
    index[, values = score[node_index] when ``with_values``: no gradient][, the node -> assignment index when
    ``with_lift``: what Reduce's backward and Lift walk, from the same compaction])."""
    dev = N.require_device(score, batch, ptr, k, koff)
    score = N.f32c(score.reshape(-1))
    n = score.numel()
    kk = max(k_total, 1)
    one = with_values and 32 * kk <= _SPS_ONE_ALLOC_BYTES  # (index 16 B + values 4 + perm 4 + pack 8 per kept node)
    if one:
        # the selector's outputs of a batch of small graphs are a few hundred KB: ONE allocation, typed views behind the
        # launch (every torch.empty in front of a ~5 us kernel is ~1.2 us of idle GPU)
        o_val = (16 * k_total + 15) & ~15
        o_perm = (o_val + 4 * k_total + 15) & ~15
        o_pack = (o_perm + 4 * kk + 15) & ~15
        buf = torch.empty((o_pack + 8 * kk + 15) & ~15, dtype=torch.uint8, device=dev)
        base = buf.data_ptr()
        idx_p, val_p, perm_p, pack_p = base, base + o_val, base + o_perm, base + o_pack
    else:
        index = torch.empty(2, k_total, dtype=torch.int64, device=dev)  # the indices of the sparse S, written in place
        values = torch.empty(k_total, dtype=torch.float32, device=dev) if with_values else None
        perm = torch.empty(kk, dtype=torch.int32, device=dev)
        # the packed one-to-one index {node, score} the sparse Reduce streams (when the scores ARE the weights of S)
        pack = torch.empty(kk, dtype=torch.int64, device=dev) if with_values else None
        idx_p, val_p, perm_p, pack_p = index.data_ptr(), N.ptr(values), perm.data_ptr(), N.ptr(pack)
    lift_ptr = torch.empty(n + 1, dtype=torch.int32, device=dev) if with_lift and n > 0 else None
    with_lift = lift_ptr is not None
    L = N.lib()
    ws = N.workspace(L.tgp_topk_select_workspace_bytes(n), dev)
    # r5: graphs beyond the per-graph sort routes (the device-wide route) also get the kept-node bitmap and its rank
    # directory from the compaction pass -- what the subgraph Connect of this very selection starts from
    directory = None
    import ctypes as _ct
    wrote = _ct.c_int(0)
    if n > 0 and k_total > 0 and not (0 < segments_max_nodes <= 8192):
        nblk = int(L.tgp_topk_select_directory_blocks(n))
        directory = torch.empty(5 * nblk, dtype=torch.int32, device=dev)  # [4 nblk] bitmap words | [nblk] rank128
    N.check(L.tgp_topk_select(N.ptr(score), N.ptr(None if batch is None else N.i64c(batch)), n, num_graphs,
                              N.ptr(N.i64c(ptr)), N.ptr(N.i64c(k)), N.ptr(N.i64c(koff)), segments_max_nodes, N.ptr(ws),
                              ws.numel(), idx_p, idx_p + 8 * k_total, perm_p, val_p,
                              N.ptr(lift_ptr), pack_p,
                              None if directory is None else directory.data_ptr(),
                              None if directory is None else directory.data_ptr() + 16 * nblk,
                              _ct.addressof(wrote) if directory is not None else None,
                              N.stream_ptr(dev)), "tgp_topk_select")
    if not wrote.value:
        directory = None
    if one:
        i64, i32, f32 = buf.view(torch.int64), buf.view(torch.int32), buf.view(torch.float32)
        index = torch.as_strided(i64, (2, k_total), (k_total, 1), 0)
        values = torch.as_strided(f32, (k_total,), (1,), o_val >> 2)
        perm = torch.as_strided(i32, (kk,), (1,), o_perm >> 2)
        pack = torch.as_strided(i64, (kk,), (1,), o_pack >> 3)
    assign = AssignIndex(None, perm, k_total, k_total)
    assign.member_directory = directory  # (bitmap | rank128) of the kept nodes, or None
    assign.member_directory_key = (index.data_ptr(), k_total) if directory is not None else None
    if pack is not None and k_total > 0 and n < (1 << 31):
        assign.pack, assign.pack_key = pack, (index.data_ptr(), values.data_ptr())
    out = (index, assign) + ((values,) if with_values else ())
    return out + ((AssignIndex(lift_ptr, None, k_total, n),) if with_lift else ())


_ROWS_SORTED: dict = {}
_HUB_ROWS: dict = {}  # edge lists on which the row-local coalesce met a supernode row beyond its LDS sort


def _remember_hub_rows(edge_index: Tensor) -> None:
    import weakref
    if len(_HUB_ROWS) >= 16:
        for key in [k for k, v in _HUB_ROWS.items() if v[0]() is None]:
            del _HUB_ROWS[key]
        while len(_HUB_ROWS) >= 16:
            del _HUB_ROWS[next(iter(_HUB_ROWS))]
    _HUB_ROWS[id(edge_index)] = (weakref.ref(edge_index), edge_index._version)


def _rows_sorted_memo(edge_index: Tensor) -> Optional[bool]:
    """What is already known about this tensor object: True / False, or None when it has not been looked at."""
    hit = _ROWS_SORTED.get(id(edge_index))
    if hit is not None and hit[0]() is edge_index and hit[1] == edge_index._version:
        return hit[2]
    return None


def _rows_sorted(edge_index: Tensor, row: Tensor) -> bool:
    """Is the list grouped by ascending source node?  Memoised per tensor object (weak reference + version counter,
    like utils.ops.batch_info), so an unchanged edge_index costs one host round trip in total, not one per call."""
    import weakref
    hit = _ROWS_SORTED.get(id(edge_index))
    if hit is not None and hit[0]() is edge_index and hit[1] == edge_index._version:
        return hit[2]
    flag = bool((row[1:] >= row[:-1]).all())
    _remember_rows_sorted(edge_index, flag)
    return flag


def _remember_rows_sorted(edge_index: Tensor, flag: bool) -> None:
    import weakref
    if len(_ROWS_SORTED) >= 16:
        for key in [k for k, v in _ROWS_SORTED.items() if v[0]() is None]:
            del _ROWS_SORTED[key]
        while len(_ROWS_SORTED) >= 16:
            del _ROWS_SORTED[next(iter(_ROWS_SORTED))]
    _ROWS_SORTED[id(edge_index)] = (weakref.ref(edge_index), edge_index._version, flag)


def graclus_match(edge_index: Tensor, edge_weight: Optional[Tensor], num_nodes: int,
                  max_rounds: Optional[int] = None, return_row_ptr: bool = False,
                  graph_ptr: Optional[Tensor] = None, max_graph_nodes: Optional[int] = None,
                  relabel: bool = False):
    """label[i] = min(i, partner) of a heavy-edge maximal matching (select/graclus_select.py:66 ->
    torch_cluster.graclus_cluster): handshake rounds on the device until a round matches nothing, however many that
    takes (a path with monotone weights matches one pair per round: n/2 rounds; torch_cluster iterates until done
    too).  ``max_rounds`` bounds the loop for callers that accept a non-maximal matching; if it is hit a
    RuntimeWarning says so.  ``return_row_ptr``: also return the int32 CSR offsets [num_nodes + 1] of the list when
    it is row-sorted (else None) -- the matcher builds them anyway and SparseConnect can reuse them.
    ``graph_ptr`` [B+1] / ``max_graph_nodes``: node offsets of the graphs of a sorted batch and its longest graph; when
    every graph fits one workgroup all rounds of all graphs run as ONE launch (same matching).
    ``relabel``: return ``(index [2, N], K, assign_index, ones [N])`` instead of the labels -- row 0 = 0..N-1, row 1 = the consecutive cluster id
    of every node (``torch.unique(label, return_inverse=True)[1]``, graclus_select.py:68) from two more launches, its
    count read back together with the matcher's status word."""
    dev = N.require_device(edge_index, edge_weight)
    row, col = _edge_rows(edge_index)
    E = row.numel()
    w = None if edge_weight is None else N.f32c(edge_weight.reshape(-1))
    L = N.lib()
    st = N.stream_ptr(dev)
    if (relabel and max_rounds is None and graph_ptr is not None and max_graph_nodes is not None and E > 0
            and num_nodes > 0 and graph_ptr.numel() >= 2 and max_graph_nodes <= 64 and _GRACLUS_FUSED
            and _rows_sorted_memo(edge_index) is not False and not torch.cuda.is_current_stream_capturing()):
        # r4: matching + relabelling + members index of a batch of small graphs in ONE launch, straight from the edge list
        # (no CSR); the kernel validates row order itself, so a list nobody has looked at yet needs no check pass
        gp = N.i64c(graph_ptr)
        B = gp.numel() - 1
        state = _sps_state(dev, st, L.tgp_graclus_match_graphs_fused_status_words(B))
        epoch = state.next_epoch()
        # one allocation for the four outputs (a batch of small graphs: a few hundred KB); the typed views are made
        # behind the launch, while the kernel runs
        o_idx = 0
        o_ptr = (o_idx + 16 * num_nodes + 15) & ~15
        o_perm = (o_ptr + 4 * (num_nodes + 1) + 15) & ~15
        o_ones = (o_perm + 4 * num_nodes + 15) & ~15
        buf = torch.empty((o_ones + 4 * num_nodes + 15) & ~15, dtype=torch.uint8, device=dev)
        base = buf.data_ptr()
        # (a list pooled before brings its per-graph offsets along; a new one lets the kernel search: r5, no launch for them)
        eptr = _edge_ptr_memo(edge_index, graph_ptr) if _SPS_GIVE_PTRS else None
        N.check(L.tgp_graclus_match_graphs_fused(N.ptr(row), N.ptr(col), N.ptr(w), num_nodes, E, N.ptr(gp), B,
                                                 N.ptr(eptr), None,
                                                 base + o_idx, base + o_ptr, base + o_perm, base + o_ones,
                                                 state.status.data_ptr(), state.status.numel(),
                                                 state.pinned.data_ptr(), epoch, st), "tgp_graclus_match_graphs_fused")
        i64, i32, f32 = buf.view(torch.int64), buf.view(torch.int32), buf.view(torch.float32)
        index = torch.as_strided(i64, (2, num_nodes), (num_nodes, 1), o_idx >> 3)
        a_ptr = torch.as_strided(i32, (num_nodes + 1,), (1,), o_ptr >> 2)
        a_perm = torch.as_strided(i32, (num_nodes,), (1,), o_perm >> 2)
        ones = torch.as_strided(f32, (num_nodes,), (1,), o_ones >> 2)
        word = state.wait(epoch)
        if not word & 0x80000000:
            k = word & 0x7FFFFFFF
            if E > 1 and _rows_sorted_memo(edge_index) is None:
                _remember_rows_sorted(edge_index, True)  # (the kernel refuses lists whose rows are not ascending)
            out = (index, k, AssignIndex(a_ptr[:k + 1], a_perm, num_nodes, k), ones)
            return (out, None) if return_row_ptr else out
        del index, a_ptr, a_perm, ones  # refused: the staged route below decides why
    label = torch.empty(num_nodes, dtype=torch.int64, device=dev)
    sorted_ptr = None
    # PyG lists are sorted by source: one comparison pass + round trip decides whether the CSR needs a sort at all
    # (remembered per edge_index object: full-batch training pools the same graph every epoch)
    words = torch.empty(4, dtype=torch.int32, device=dev)  # [status, rows-not-sorted flag, K (int64)]
    per_graph = (max_rounds is None and graph_ptr is not None and max_graph_nodes is not None and E > 0
                 and num_nodes > 0 and max_graph_nodes <= L.tgp_graclus_match_max_graph_nodes()
                 and graph_ptr.numel() >= 2)
    # a list nobody has looked at yet, on the one-launch route: go on as if it were sorted, the check rides on the
    # offsets kernel and its flag is read with the status word (no round trip of its own)
    optimistic = per_graph and E > 1 and _rows_sorted_memo(edge_index) is None
    if optimistic or (E > 1 and _rows_sorted(edge_index, row)):
        row_ptr, perm = torch.empty(num_nodes + 1, dtype=torch.int32, device=dev), None
        if optimistic:
            N.check(L.tgp_rowptr_from_sorted_flag_i64(N.ptr(row), E, num_nodes, N.ptr(row_ptr), N.ptr(words[1:]), st),
                    "tgp_rowptr_from_sorted_flag_i64")
        else:
            N.check(L.tgp_rowptr_from_sorted_i64(N.ptr(row), E, num_nodes, N.ptr(row_ptr), st),
                    "tgp_rowptr_from_sorted_i64")
        sorted_ptr = row_ptr
    else:
        index = build_assign_index(row, num_nodes)
        row_ptr, perm = index.row_ptr, index.perm
    ws = N.workspace(L.tgp_graclus_match_workspace_bytes(num_nodes, E), dev)
    finished = num_nodes == 0 or E == 0

    pairs = {}

    def relabelled():
        index = torch.empty(2, num_nodes, dtype=torch.int64, device=dev)
        rws = N.workspace(L.tgp_graclus_relabel_workspace_bytes(num_nodes), dev)
        # the supernode -> members index comes out of the same kernels (a matching: a representative and at most one
        # partner per cluster), so Reduce / Connect do not build it from the ids
        pairs["row_ptr"] = torch.empty(num_nodes + 1, dtype=torch.int32, device=dev)
        pairs["perm"] = torch.empty(max(num_nodes, 1), dtype=torch.int32, device=dev)
        pairs["ones"] = torch.empty(num_nodes, dtype=torch.float32, device=dev)  # the values of S, same launch
        N.check(L.tgp_graclus_relabel_i64(N.ptr(label), num_nodes, N.ptr(rws), rws.numel(), N.ptr(index),
                                          N.ptr(words[2:]), N.ptr(pairs["row_ptr"]), N.ptr(pairs["perm"]),
                                          N.ptr(pairs["ones"]), st), "tgp_graclus_relabel_i64")
        return index

    def finish(index=None, k=None):
        out = label
        if relabel:
            if index is None:
                index = relabelled()
                k = words.tolist()[2]
            k = int(k)
            out = (index, k, AssignIndex(pairs["row_ptr"][:k + 1], pairs["perm"], num_nodes, k), pairs["ones"])
        return (out, sorted_ptr) if return_row_ptr else out

    def start(init_state=1):
        N.check(L.tgp_graclus_match_start(N.ptr(row), N.ptr(col), N.ptr(w), N.ptr(row_ptr), N.ptr(perm), num_nodes, E,
                                          N.ptr(ws), ws.numel(), N.ptr(label), init_state, st),
                "tgp_graclus_match_start")
    start(0 if (per_graph and not finished) else 1)  # (the one-launch route sets the labels itself)
    if not finished and per_graph:
        gp = N.i64c(graph_ptr)
        N.check(L.tgp_graclus_match_graphs(N.ptr(row_ptr), num_nodes, E, N.ptr(ws), N.ptr(gp), gp.numel() - 1,
                                           int(max_graph_nodes), N.ptr(label), N.ptr(words), st),
                "tgp_graclus_match_graphs")
        index = relabelled() if relabel else None  # (optimistic: one round trip for the status word and the count)
        got = words.tolist()
        if optimistic:
            _remember_rows_sorted(edge_index, got[1] == 0)
            if got[1] != 0:  # not sorted after all: everything above walked meaningless offsets -- again, knowing it
                return graclus_match(edge_index, edge_weight, num_nodes, max_rounds, return_row_ptr, graph_ptr,
                                     max_graph_nodes, relabel)
        if got[0] == 0:
            return finish(index, got[2])
        start()  # an entry that leaves its graph (or a longer graph than declared): the device-wide rounds, from scratch
    done, step = 0, 6  # late rounds are cheap (free nodes are packed before scanning): prefer fewer round trips
    while not finished and (max_rounds is None or done < max_rounds):
        if max_rounds is not None:
            step = min(step, max_rounds - done)
        matched = torch.empty(step + 1, dtype=torch.int32, device=dev)  # [round flags ..., tail status]
        N.check(L.tgp_graclus_match_rounds(N.ptr(row_ptr), num_nodes, E, N.ptr(ws), step, N.ptr(matched),
                                           N.ptr(label), st), "tgp_graclus_match_rounds")
        done += step
        if max_rounds is None:
            # the few thousand nodes that are still free after these rounds: all their remaining rounds in one workgroup
            # (declines when there are more than that), read back with the round flags
            N.check(L.tgp_graclus_match_tail(N.ptr(row_ptr), num_nodes, E, N.ptr(ws), N.ptr(label),
                                             N.ptr(matched[step:]), st), "tgp_graclus_match_tail")
            last, tail = matched[step - 1:].tolist()
            finished = last == 0 or tail == 1
        else:
            finished = int(matched[step - 1].item()) == 0  # one round trip per batch of rounds
        # random-like graphs finish in a handful of rounds; a long tail means chain-like structure: batch more
        # rounds per round trip (4, then doubling up to 256) so that n/2 rounds cost n/512 host synchronisations
        step = 4 if done <= 6 else min(2 * step, 256)
    if not finished:
        import warnings
        warnings.warn(f"graclus_match stopped after max_rounds={max_rounds} rounds: the matching is not maximal",
                      RuntimeWarning)
    return finish()


def _rows_f32(x: Tensor) -> Tensor:
    x = x.to(torch.float32) if x.dtype != torch.float32 else x
    return x if x.stride(1) == 1 else x.contiguous()


def row_dot(x: Tensor, w: Tensor) -> Tensor:
    """out[i] = <x[i,:], w> (select/topk_select.py:176) in one pass over x."""
    dev = N.require_device(x, w)
    x, w = _rows_f32(x), N.f32c(w.reshape(-1))
    out = torch.empty(x.size(0), dtype=torch.float32, device=dev)
    N.check(N.lib().tgp_row_dot_f32(N.ptr(x), x.size(0), x.size(1), x.stride(0), N.ptr(w), N.ptr(out),
                                    N.stream_ptr(dev)), "tgp_row_dot_f32")
    return out


def topk_score(x: Tensor, w: Tensor, tanh: bool) -> Tensor:
    """act(x.w / ||w||_2), act = tanh or identity: TopkSelect's ratio-mode score (select/topk_select.py:176-184) in
    the one pass over x (no gradient: inference and detached inputs)."""
    dev = N.require_device(x, w)
    x, w = _rows_f32(x), N.f32c(w.reshape(-1))
    out = torch.empty(x.size(0), dtype=torch.float32, device=dev)
    N.check(N.lib().tgp_topk_score_f32(N.ptr(x), x.size(0), x.size(1), x.stride(0), N.ptr(w), 1 if tanh else 0,
                                       N.ptr(out), N.stream_ptr(dev)), "tgp_topk_score_f32")
    return out


def weighted_colsum(x: Tensor, g: Tensor) -> Tensor:
    """out[f] = sum_i g[i] x[i,f]: the weight gradient of :func:`row_dot`."""
    dev = N.require_device(x, g)
    x, g = _rows_f32(x), N.f32c(g.reshape(-1))
    out = torch.empty(x.size(1), dtype=torch.float32, device=dev)
    L = N.lib()
    ws = N.workspace(L.tgp_weighted_colsum_workspace_bytes(x.size(1)), dev)
    N.check(L.tgp_weighted_colsum_f32(N.ptr(x), x.size(0), x.size(1), x.stride(0), N.ptr(g), N.ptr(out), N.ptr(ws),
                                      ws.numel(), N.stream_ptr(dev)), "tgp_weighted_colsum_f32")
    return out


def topk_pool_bwd_fits(x: Tensor, w: Tensor) -> bool:
    """Shapes / layouts :func:`topk_pool_bwd` takes (the others keep the operator-by-operator backward)."""
    return bool(x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and w.dtype == torch.float32
                and w.numel() == x.size(1) and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0
                and N.lib().tgp_topk_pool_bwd_fits(x.size(1)))


def topk_pool_bwd(x: Tensor, node_index: Tensor, cluster_index: Optional[Tensor], values: Tensor,
                  g_xpool: Optional[Tensor], g_values: Optional[Tensor], w: Tensor, use_tanh: bool,
                  want_gx: bool, want_gw: bool):
    """``(gx [N,F] or None, gw [F] or None)``: backward of TopK pooling's trained path -- score, kept values of S, gated
    Reduce -- in one pass over the kept rows (``tgp_topk_pool_bwd_f32``; poolers/topk.py:150-190)."""
    dev = N.require_device(x, node_index, cluster_index, values, g_xpool, g_values, w)
    n, F = x.size(0), x.size(1)
    k = node_index.numel()
    ni = N.i64c(node_index)
    ci = None if cluster_index is None else N.i64c(cluster_index)
    vals = N.f32c(values.reshape(-1))
    gp = None if g_xpool is None else N.f32c(g_xpool.reshape(k, F))
    gv = None if g_values is None else N.f32c(g_values.reshape(-1))
    wc = N.f32c(w.reshape(-1))
    # 16-byte loads: an upstream gradient or a parameter that is a view at an odd offset of a larger buffer (a flattened
    # parameter bucket, a slice of a concatenation) is copied once
    if gp is not None and gp.data_ptr() % 16:
        gp = gp.clone()
    if wc.data_ptr() % 16:
        wc = wc.clone()
    gx = torch.empty(n, F, dtype=torch.float32, device=dev) if want_gx else None
    gw = torch.empty(F, dtype=torch.float32, device=dev) if want_gw else None
    L = N.lib()
    ws = N.workspace(L.tgp_topk_pool_bwd_workspace_bytes(F), dev) if want_gw else None
    N.check(L.tgp_topk_pool_bwd_f32(x.data_ptr(), n, F, x.stride(0), N.ptr(ni), N.ptr(ci), N.ptr(vals), k, N.ptr(gp),
                                    N.ptr(gv), N.ptr(wc), 1 if use_tanh else 0, N.ptr(gx), N.ptr(gw), N.ptr(ws),
                                    ws.numel() if ws is not None else 0, N.stream_ptr(dev)), "tgp_topk_pool_bwd_f32")
    return gx, gw


def pair_dot(a: Tensor, ia: Tensor, b: Tensor, ib: Tensor) -> Tensor:
    """out[e] = <a[ia_e,:], b[ib_e,:]>: the assignment-weight gradient of the sparse Reduce / Lift."""
    dev = N.require_device(a, ia, b, ib)
    a, b, ia, ib = N.f32c(a), N.f32c(b), N.i64c(ia), N.i64c(ib)
    if a.dim() != 2 or b.dim() != 2 or a.size(1) != b.size(1) or ia.numel() != ib.numel():
        raise ValueError(f"pair_dot: a {tuple(a.shape)}, b {tuple(b.shape)}, {ia.numel()} / {ib.numel()} indices")
    out = torch.empty(ia.numel(), dtype=torch.float32, device=dev)
    N.check(N.lib().tgp_pair_dot_f32(N.ptr(ia), N.ptr(ib), ia.numel(), N.ptr(a), N.ptr(b), a.size(1), N.ptr(out),
                                     N.stream_ptr(dev)), "tgp_pair_dot_f32")
    return out


def edge_dot(s: Tensor, edge_index: Tensor) -> Tensor:
    """ss[e] = <S[row_e], S[col_e]> (utils/losses.py:73-127, 661-708: ``(S[src] * S[dst]).sum(-1)``) in one pass."""
    dev = N.require_device(s, edge_index)
    row, col = _edge_rows(edge_index)
    if s.dtype == torch.float64:
        return (s[row] * s[col]).sum(-1)
    s = N.f32c(s)
    out = torch.empty(row.numel(), dtype=torch.float32, device=dev)
    N.check(N.lib().tgp_edge_dot_f32(N.ptr(row), N.ptr(col), row.numel(), N.ptr(s), s.size(0), s.size(1), N.ptr(out),
                                     N.stream_ptr(dev)), "tgp_edge_dot_f32")
    return out


def bmm(a: Tensor, b: Tensor, trans_a: bool = False, accumulate_into: Optional[Tensor] = None) -> Tensor:
    """C[g] = op(A[g]) @ B[g] on the fp32 matrix cores.  a: [G,M,Kd] (or [G,Kd,M] when trans_a),
    b: [G,Kd,Nc]; 2-D operands are treated as G = 1.  ``accumulate_into`` (contiguous fp32 [G,M,Nc]): C += product in
    the GEMM epilogue, returned as that tensor.  A float64 operand sends the product to the fp64 matrix path
    (tgp_bmm_f64); the result is float64."""
    dev = N.require_device(a, b)
    f64 = _any_f64(a, b) or (accumulate_into is not None and accumulate_into.dtype == torch.float64)
    conv, dt = (N.f64c, torch.float64) if f64 else (N.f32c, torch.float32)
    a3 = conv(a if a.dim() == 3 else a.unsqueeze(0))
    b3 = conv(b if b.dim() == 3 else b.unsqueeze(0))
    G = max(a3.size(0), b3.size(0))
    if trans_a:
        Kd, M = a3.size(1), a3.size(2)
    else:
        M, Kd = a3.size(1), a3.size(2)
    if b3.size(1) != Kd:
        raise ValueError(f"bmm inner dimensions differ: {tuple(a.shape)} x {tuple(b.shape)}")
    Nc = b3.size(2)
    sA = 0 if a3.size(0) == 1 else a3.stride(0)
    sB = 0 if b3.size(0) == 1 else b3.stride(0)
    if accumulate_into is not None:
        out = accumulate_into
        if (out.dtype != dt or not out.is_contiguous() or out.numel() != G * M * Nc or out.device != dev):
            raise ValueError(f"bmm: accumulate_into must be a contiguous {dt} tensor of the product's shape")
        if f64:
            N.check(N.lib().tgp_bmm_f64(N.ptr(a3), N.ptr(b3), N.ptr(out), G, M, Nc, Kd, 1 if trans_a else 0,
                                        a3.stride(1), b3.stride(1), Nc, sA, sB, M * Nc, 1, N.stream_ptr(dev)),
                    "tgp_bmm_f64")
            return out
        N.check(N.lib().tgp_bmm_accumulate_f32(N.ptr(a3), N.ptr(b3), N.ptr(out), G, M, Nc, Kd, 1 if trans_a else 0,
                                               a3.stride(1), b3.stride(1), Nc, sA, sB, M * Nc, N.stream_ptr(dev)),
                "tgp_bmm_accumulate_f32")
        return out
    out = torch.empty(G, M, Nc, dtype=dt, device=dev)
    if f64:
        N.check(N.lib().tgp_bmm_f64(N.ptr(a3), N.ptr(b3), N.ptr(out), G, M, Nc, Kd, 1 if trans_a else 0, a3.stride(1),
                                    b3.stride(1), Nc, sA, sB, M * Nc, 0, N.stream_ptr(dev)), "tgp_bmm_f64")
    else:
        N.check(N.lib().tgp_bmm_f32(N.ptr(a3), N.ptr(b3), N.ptr(out), G, M, Nc, Kd, 1 if trans_a else 0, a3.stride(1),
                                    b3.stride(1), Nc, sA, sB, M * Nc, N.stream_ptr(dev)), "tgp_bmm_f32")
    return out if (a.dim() == 3 or b.dim() == 3) else out[0]


# ------------------------------------------------------------------------- A13
def mlp_select(x: Tensor, weight: Tensor, bias: Optional[Tensor], mask: Optional[Tensor]) -> Tensor:
    """S = softmax(x W^T + b) * mask[..., None] (select/mlp_select.py:139-145, single Linear: :67) in one pass over
    x.  x [..., F], weight [K, F], bias [K] or None, mask broadcastable to x.shape[:-1] (bool) or None."""
    dev = N.require_device(x, weight)
    lead = x.shape[:-1]
    x2 = N.f32c(x).reshape(-1, x.size(-1))
    w, b = N.f32c(weight), (None if bias is None else N.f32c(bias))
    M, F, Kc = x2.size(0), x2.size(1), w.size(0)
    if w.size(1) != F:
        raise ValueError(f"mlp_select: weight {tuple(weight.shape)} does not match features {F}")
    m8 = None
    if mask is not None:
        m8 = mask.reshape(-1)
        if m8.numel() != M:
            raise ValueError(f"mlp_select: mask {tuple(mask.shape)} does not match x {tuple(x.shape)}")
        if m8.dtype != torch.bool and m8.dtype != torch.uint8:
            m8 = m8 != 0
        m8 = m8.contiguous().view(torch.uint8)
    L = N.lib()
    st = N.stream_ptr(dev)
    if Kc <= L.tgp_mlp_select_max_fused_k():
        out = torch.empty(M, Kc, dtype=torch.float32, device=dev)
        N.check(L.tgp_mlp_select_f32(N.ptr(x2), N.ptr(w), N.ptr(b), N.ptr(m8), M, F, Kc, N.ptr(out), st),
                "tgp_mlp_select_f32")
    else:  # wider than eight accumulator tiles: tiled GEMM for the logits, then bias + softmax + mask in place
        out = bmm(x2, w.t().contiguous())
        N.check(L.tgp_softmax_rows_f32(N.ptr(out), N.ptr(b), N.ptr(m8), M, Kc, st), "tgp_softmax_rows_f32")
    return out.view(*lead, Kc)


def softmax_bwd(s: Tensor, ds: Tensor) -> Tensor:
    """dY = S * (dS - <dS, S>_row): gradient of S = softmax(Y) * mask w.r.t. Y (masked rows have S = 0)."""
    dev = N.require_device(s, ds)
    s2, d2 = N.f32c(s).reshape(-1, s.size(-1)), N.f32c(ds).reshape(-1, s.size(-1))
    out = torch.empty_like(s2)
    N.check(N.lib().tgp_softmax_bwd_f32(N.ptr(s2), N.ptr(d2), N.ptr(out), s2.size(0), s2.size(1),
                                        N.stream_ptr(dev)), "tgp_softmax_bwd_f32")
    return out.view(s.shape)


def segment_gemm_tn(s: Tensor, y: Tensor, ptr: Tensor, max_nodes: int) -> Tensor:
    """C[b] = S_b^T Y_b over node rows ptr[b]..ptr[b+1] (reduce/base_reduce.py:170-182,
    connect/dense_conn.py:195-206) in one launch."""
    dev = N.require_device(s, y, ptr)
    if _any_f64(s, y):
        s, y, ptr = N.f64c(s), N.f64c(y), N.i64c(ptr)
        B = ptr.numel() - 1
        K, F = s.size(1), y.size(1)
        out = torch.empty(B, K, F, dtype=torch.float64, device=dev)
        L = N.lib()
        ws = N.workspace(L.tgp_segment_gemm_tn_workspace_bytes_f64(B, K, F, max_nodes), dev)
        N.check(L.tgp_segment_gemm_tn_f64(N.ptr(s), N.ptr(y), N.ptr(ptr), N.ptr(out), B, s.size(0), K, F, max_nodes,
                                          N.ptr(ws), ws.numel(), N.stream_ptr(dev)), "tgp_segment_gemm_tn_f64")
        return out
    s, y, ptr = N.f32c(s), N.f32c(y), N.i64c(ptr)
    B = ptr.numel() - 1
    K, F = s.size(1), y.size(1)
    out = torch.empty(B, K, F, dtype=torch.float32, device=dev)
    L = N.lib()
    ws = N.workspace(L.tgp_segment_gemm_tn_workspace_bytes(B, K, F, max_nodes), dev)
    N.check(L.tgp_segment_gemm_tn_f32(N.ptr(s), N.ptr(y), N.ptr(ptr), N.ptr(out), B, s.size(0), K, F,
                                      max_nodes, N.ptr(ws), ws.numel(), N.stream_ptr(dev)), "tgp_segment_gemm_tn_f32")
    return out


def segment_gemm_nn(a: Tensor, bm: Tensor, ptr: Tensor, max_nodes: int) -> Tensor:
    """C[rows of graph b] = A[rows of graph b] @ Bm[b] (lift/base_lift.py:138-247 on an un-padded batch; the
    backward of :func:`segment_gemm_tn`) in one launch.  a: [Ntot,Kd], bm: [B,Kd,Nc]."""
    dev = N.require_device(a, bm, ptr)
    f64 = _any_f64(a, bm)
    conv = N.f64c if f64 else N.f32c
    a, bm, ptr = conv(a), conv(bm), N.i64c(ptr)
    B = ptr.numel() - 1
    if bm.dim() != 3 or bm.size(0) != B or bm.size(1) != a.size(1):
        raise ValueError(f"segment_gemm_nn: a {tuple(a.shape)} x bm {tuple(bm.shape)} with {B} graphs")
    if f64:
        out = torch.empty(a.size(0), bm.size(2), dtype=torch.float64, device=dev)
        N.check(N.lib().tgp_segment_gemm_nn_f64(N.ptr(a), N.ptr(bm), N.ptr(ptr), N.ptr(out), B, a.size(0), a.size(1),
                                                bm.size(2), max_nodes, N.stream_ptr(dev)), "tgp_segment_gemm_nn_f64")
        return out
    out = torch.empty(a.size(0), bm.size(2), dtype=torch.float32, device=dev)
    N.check(N.lib().tgp_segment_gemm_nn_f32(N.ptr(a), N.ptr(bm), N.ptr(ptr), N.ptr(out), B, a.size(0), a.size(1),
                                            bm.size(2), max_nodes, N.stream_ptr(dev)), "tgp_segment_gemm_nn_f32")
    return out


def rowptr_from_sorted(row: Tensor, num_rows: int, out: Tensor) -> Tensor:
    """CSR offsets (int32 [num_rows+1]) of an ascending int64 row vector."""
    dev = N.require_device(row, out)
    row = N.i64c(row)
    N.check(N.lib().tgp_rowptr_from_sorted_i64(N.ptr(row), row.numel(), num_rows, N.ptr(out), N.stream_ptr(dev)),
            "tgp_rowptr_from_sorted_i64")
    return out


def spmm_sorted(edge_index: Tensor, edge_weight: Optional[Tensor], num_rows: int, s: Tensor) -> Tensor:
    """T = A S for a row-sorted (coalesced) edge list (connect/dense_conn.py:165,204)."""
    dev = N.require_device(edge_index, edge_weight, s)
    row, col = _edge_rows(edge_index)
    f64 = _any_f64(s, edge_weight)
    conv = N.f64c if f64 else N.f32c
    s = conv(s)
    w = None if edge_weight is None else conv(edge_weight)
    L = N.lib()
    st = N.stream_ptr(dev)
    row_ptr = torch.empty(num_rows + 1, dtype=torch.int32, device=dev)
    N.check(L.tgp_rowptr_from_sorted_i64(N.ptr(row), row.numel(), num_rows, N.ptr(row_ptr), st),
            "tgp_rowptr_from_sorted_i64")
    if f64:
        out = torch.empty(num_rows, s.size(1), dtype=torch.float64, device=dev)
        N.check(L.tgp_spmm_csr_f64(N.ptr(row_ptr), N.ptr(col), N.ptr(w), num_rows, row.numel(), N.ptr(s), s.size(1),
                                   N.ptr(out), st), "tgp_spmm_csr_f64")
        return out
    out = torch.empty(num_rows, s.size(1), dtype=torch.float32, device=dev)
    N.check(L.tgp_spmm_csr_f32(N.ptr(row_ptr), N.ptr(col), N.ptr(w), num_rows, row.numel(), N.ptr(s), s.size(1),
                               N.ptr(out), st), "tgp_spmm_csr_f32")
    return out


def spmm_sorted_csr(edge_index: Tensor, edge_weight: Optional[Tensor], num_rows: int, s: Tensor):
    """(T, row_ptr int32 [num_rows+1]): :func:`spmm_sorted` for float32 operands, handing the CSR offsets it built back."""
    dev = N.require_device(edge_index, edge_weight, s)
    row, col = _edge_rows(edge_index)
    s = N.f32c(s)
    w = None if edge_weight is None else N.f32c(edge_weight)
    L = N.lib()
    st = N.stream_ptr(dev)
    row_ptr = torch.empty(num_rows + 1, dtype=torch.int32, device=dev)
    N.check(L.tgp_rowptr_from_sorted_i64(N.ptr(row), row.numel(), num_rows, N.ptr(row_ptr), st),
            "tgp_rowptr_from_sorted_i64")
    out = torch.empty(num_rows, s.size(1), dtype=torch.float32, device=dev)
    N.check(L.tgp_spmm_csr_f32(N.ptr(row_ptr), N.ptr(col), N.ptr(w), num_rows, row.numel(), N.ptr(s), s.size(1),
                               N.ptr(out), st), "tgp_spmm_csr_f32")
    return out, row_ptr


_CSR_OFFSETS: dict = {}  # id(edge_index) -> (weakref, version, num_rows, row_ptr)


def csr_offsets(edge_index: Tensor, num_rows: int) -> Tensor:
    """int32 [num_rows+1] CSR offsets of a row-sorted list: one launch for a new list, remembered per tensor object +
    version (full-batch training pools the same ``edge_index`` every step), like the per-graph edge ranges."""
    key = id(edge_index)
    hit = _CSR_OFFSETS.get(key)
    if hit is not None and hit[0]() is edge_index and hit[1] == edge_index._version and hit[2] == num_rows:
        return hit[3]
    row, _ = _edge_rows(edge_index)
    out = torch.empty(num_rows + 1, dtype=torch.int32, device=edge_index.device)
    rowptr_from_sorted(row, num_rows, out)
    if not torch.cuda.is_current_stream_capturing():
        if key in _CSR_OFFSETS:
            del _CSR_OFFSETS[key]
        elif len(_CSR_OFFSETS) >= 16:
            del _CSR_OFFSETS[next(iter(_CSR_OFFSETS))]
        _CSR_OFFSETS[key] = (_weakref(edge_index), edge_index._version, num_rows, out)
    return out


def spmm_csr(row_ptr: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor], num_rows: int, s: Tensor,
             want_stats: bool = False):
    """T = A S for a row-sorted coalesced float32 list whose CSR offsets the caller holds (one launch).
    ``want_stats``: ``(T, deg, q)`` -- the row sums of the weights and |S_i|^2 from the same launch
    (:func:`edge_row_stats` without its own pass; needs num_rows == S rows)."""
    dev = N.require_device(edge_index, edge_weight, s, row_ptr)
    _, col = _edge_rows(edge_index)
    s = N.f32c(s)
    w = None if edge_weight is None else N.f32c(edge_weight.reshape(-1))
    out = torch.empty(num_rows, s.size(1), dtype=torch.float32, device=dev)
    if want_stats == "entropy":
        # (T, partials or None): DiffPool's entropy sum over S rides along as per-workgroup shares (None: this shape does
        # not take the row kernel; diffpool_unbatched_tail then runs its own pass over S)
        if s.size(0) != num_rows:
            raise ValueError("spmm_csr(want_stats='entropy') needs a square A: one row of S per row of A")
        import ctypes as _ct
        part = torch.empty(max(num_rows, 1), dtype=torch.float32, device=dev)
        n_part = _ct.c_int(-1)
        N.check(N.lib().tgp_spmm_csr_entropy_f32(N.ptr(row_ptr), N.ptr(col), N.ptr(w), num_rows, col.numel(), N.ptr(s),
                                                  s.size(1), N.ptr(out), losses_eps(), N.ptr(part),
                                                  _ct.addressof(n_part), N.stream_ptr(dev)), "tgp_spmm_csr_entropy_f32")
        return out, (part[: n_part.value] if n_part.value >= 0 else None)
    if want_stats:
        if s.size(0) != num_rows:
            raise ValueError("spmm_csr(want_stats=True) needs a square A: one row of S per row of A")
        dq = torch.empty(2, num_rows, dtype=torch.float32, device=dev)
        N.check(N.lib().tgp_spmm_csr_stats_f32(N.ptr(row_ptr), N.ptr(col), N.ptr(w), num_rows, col.numel(), N.ptr(s),
                                               s.size(1), N.ptr(out), dq.data_ptr(), dq.data_ptr() + 4 * num_rows,
                                               N.stream_ptr(dev)), "tgp_spmm_csr_stats_f32")
        return out, dq[0], dq[1]
    N.check(N.lib().tgp_spmm_csr_f32(N.ptr(row_ptr), N.ptr(col), N.ptr(w), num_rows, col.numel(), N.ptr(s), s.size(1),
                                     N.ptr(out), N.stream_ptr(dev)), "tgp_spmm_csr_f32")
    return out


def edge_row_stats(row_ptr: Tensor, edge_weight: Optional[Tensor], s: Tensor) -> Tuple[Tensor, Tensor]:
    """(deg [N], q [N]): row sums of the weights of a CSR edge list (entry counts without weights) and |S_i|^2, one
    launch (utils/losses.py:73-127: the degree term of sparse_mincut_loss without an index_add)."""
    dev = N.require_device(row_ptr, edge_weight, s)
    s = N.f32c(s)
    n = s.size(0)
    w = None if edge_weight is None else N.f32c(edge_weight.reshape(-1))
    deg = torch.empty(n, dtype=torch.float32, device=dev)
    q = torch.empty(n, dtype=torch.float32, device=dev)
    N.check(N.lib().tgp_edge_row_stats_f32(N.ptr(row_ptr), N.ptr(w), N.ptr(s), n, s.size(1), N.ptr(deg), N.ptr(q),
                                           N.stream_ptr(dev)), "tgp_edge_row_stats_f32")
    return deg, q


def segment_gemm_tn3(s: Tensor, ys, ptr: Tensor, max_nodes: int, transpose0: bool = False,
                     post_flags: Optional[int] = None):
    """[S_b^T Y_j,b for j] for up to three float32 right-hand sides [Ntot,F_j] over node rows ptr[b]..ptr[b+1]: ONE
    product grid + one combine launch (the unbatched dense poolers' S^T [A S | X | S]).
    ``post_flags`` (r6; ys[0] must be [Ntot,K]): a last value ``adj_pool`` = the post-processing of the first product
    (utils/ops.py:282-335) from the same native call -- for 64 < K <= 176 its launch sums the slabs of all the products
    itself, no combine launch."""
    dev = N.require_device(s, ptr, *ys)
    s, ptr = N.f32c(s), N.i64c(ptr)
    ys = [N.f32c(y) for y in ys]
    if not 1 <= len(ys) <= 3 or any(y.dim() != 2 or y.size(0) != s.size(0) for y in ys):
        raise ValueError("segment_gemm_tn3: one to three [Ntot,F] right-hand sides expected")
    B, Kc = ptr.numel() - 1, s.size(1)
    fs = [y.size(1) for y in ys] + [0] * (3 - len(ys))
    outs = [torch.empty(B, Kc, y.size(1), dtype=torch.float32, device=dev) for y in ys]
    yp = [N.ptr(y) for y in ys] + [None] * (3 - len(ys))
    op = [N.ptr(o) for o in outs] + [None] * (3 - len(ys))
    L = N.lib()
    if post_flags is not None:
        if fs[0] != Kc:
            raise ValueError("segment_gemm_tn3(post_flags=...): the first right-hand side must be [Ntot,K]")
        adj_pool = torch.empty(B, Kc, Kc, dtype=torch.float32, device=dev)
        ws = N.workspace(L.tgp_segment_gemm_tn3_post_workspace_bytes(B, Kc, fs[1], fs[2], max_nodes), dev)
        N.check(L.tgp_segment_gemm_tn3_post_f32(N.ptr(s), yp[0], yp[1], fs[1], yp[2], fs[2], N.ptr(ptr), op[0], op[1],
                                                op[2], N.ptr(adj_pool), B, s.size(0), Kc, max_nodes,
                                                1 if transpose0 else 0, int(post_flags), ops_eps(), N.ptr(ws),
                                                ws.numel(), N.stream_ptr(dev)), "tgp_segment_gemm_tn3_post_f32")
        return outs + [adj_pool]
    ws = N.workspace(L.tgp_segment_gemm_tn3_workspace_bytes(B, Kc, fs[0], fs[1], fs[2], max_nodes), dev)
    # transpose0: the first result (K x K) leaves the combine launch transposed
    N.check(L.tgp_segment_gemm_tn3_f32(N.ptr(s), yp[0], fs[0], yp[1], fs[1], yp[2], fs[2], N.ptr(ptr), op[0], op[1], op[2],
                                       B, s.size(0), Kc, max_nodes, 1 if transpose0 else 0, N.ptr(ws), ws.numel(),
                                       N.stream_ptr(dev)), "tgp_segment_gemm_tn3_f32")
    return outs


_POOL_ROWS_ONE_CALL = os.environ.get("TGP_POOL_ROWS_ONE_CALL", "1") != "0"  # A/B switch (read once)


def pool_rows_forward(x: Tensor, weight: Optional[Tensor], bias: Optional[Tensor], s_given: Optional[Tensor],
                      row_ptr: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor], ptr: Tensor, max_nodes: int,
                      transposed: bool, flags: int, mode: int, scales=(0.0, 0.0), sw2=0.0):
    """The forward of the dense poolers' un-padded rows route as ONE native call (``tgp_pool_rows_fwd_f32``: selector,
    T = A S with the losses' row statistics, S^T [T | X | S] + post-processing, loss tail) -- the launches of
    :func:`mlp_select`, :func:`spmm_csr`, :func:`segment_gemm_tn3` and :func:`mincut_terms_fused` /
    :func:`diffpool_unbatched_tail`, without the host work between them.  ``mode`` 0: no losses, 1: MinCut, 2: DiffPool.
    Returns a dict: s, t, raw, x_pool, gram, adj_pool and -- mode 1 -- deg, q, den, terms, stats, both; -- mode 2 --
    lossv.  None when the case is not this entry's (the caller composes the operators): float32 contiguous operands, a
    selector of at most 256 clusters."""
    if not _POOL_ROWS_ONE_CALL or x.dtype != torch.float32 or x.dim() != 2 or not x.is_contiguous():
        return None
    selector = s_given is None
    n, F = x.shape
    if selector:
        if (weight is None or weight.dtype != torch.float32 or weight.dim() != 2 or weight.size(1) != F
                or not weight.is_contiguous() or weight.size(0) > 256
                or (bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous()))):
            return None
        Kc = weight.size(0)
    else:
        if s_given.dtype != torch.float32 or s_given.dim() != 2 or s_given.size(0) != n or not s_given.is_contiguous():
            return None
        Kc = s_given.size(1)
    if (n == 0 or Kc == 0 or row_ptr.dtype != torch.int32 or row_ptr.numel() != n + 1 or ptr.dtype != torch.int64
            or not ptr.is_contiguous()):
        return None
    w = edge_weight
    if w is not None:
        if w.dtype != torch.float32:
            return None
        w = w.reshape(-1)
        if not w.is_contiguous():
            w = w.contiguous()
    dev = N.require_device(x, weight, bias, s_given, row_ptr, edge_index, w, ptr)
    _, col = _edge_rows(edge_index)
    B = ptr.numel() - 1
    f32 = dict(dtype=torch.float32, device=dev)
    s = torch.empty(n, Kc, **f32) if selector else s_given
    t = torch.empty(n, Kc, **f32)
    raw = torch.empty(B, Kc, Kc, **f32)
    x_pool = torch.empty(B, Kc, F, **f32)
    adj_pool = torch.empty(B, Kc, Kc, **f32)
    gram = torch.empty(B, Kc, Kc, **f32) if mode else None
    rowstat = den = terms = stats = means = dstats = out2 = None
    ticket = None
    L = N.lib()
    st = N.stream_ptr(dev)
    sw2_dev = None
    if mode == 1:
        rowstat = torch.empty(2, n, **f32)
        den, terms, stats = torch.empty(B, **f32), torch.empty(2, B, **f32), torch.empty(B, 4, **f32)
        if B <= 256 and not torch.cuda.is_current_stream_capturing():
            means = torch.empty(2, **f32)
            ticket = _sps_state(dev, st, 0).ticket.data_ptr() + 16
    elif mode == 2:
        rowstat = torch.empty(n, **f32)
        dstats, out2 = torch.empty(B, 2, **f32), torch.empty(2, **f32)
        sw2_dev = N.f32c(sw2.reshape(1)) if isinstance(sw2, Tensor) else None
    ws = N.workspace(L.tgp_pool_rows_fwd_workspace_bytes(B, Kc, F, max_nodes, n), dev)
    N.check(L.tgp_pool_rows_fwd_f32(
        x.data_ptr(), n, F, N.ptr(weight) if selector else None, N.ptr(bias) if selector else None, s.data_ptr(),
        row_ptr.data_ptr(), N.ptr(col), N.ptr(w), col.numel(), ptr.data_ptr(), B, Kc, max_nodes, 1 if transposed else 0,
        int(flags), ops_eps(), losses_eps(), int(mode), N.ptr(sw2_dev),
        0.0 if (sw2_dev is not None or mode != 2) else float(sw2), float(scales[0]), float(scales[1]), t.data_ptr(),
        raw.data_ptr(), x_pool.data_ptr(), N.ptr(gram), adj_pool.data_ptr(), N.ptr(rowstat), N.ptr(den), N.ptr(terms),
        N.ptr(stats), N.ptr(means), ticket, N.ptr(dstats), N.ptr(out2), ws.data_ptr(), ws.numel(), st),
        "tgp_pool_rows_fwd_f32")
    out = dict(s=s, t=t, raw=raw, x_pool=x_pool, gram=gram, adj_pool=adj_pool)
    if mode == 1:
        out.update(deg=rowstat[0], q=rowstat[1], den=den, terms=terms, stats=stats,
                   both=means if means is not None else terms.mean(dim=1))
    elif mode == 2:
        out.update(lossv=out2)
    return out


def pool_rows_backward(s: Tensor, t: Tensor, x: Tensor, weight: Tensor, raw: Tensor, gram, stats, den, deg, lossv,
                       ptr: Tensor, batch, slab_ptr: Tensor, max_nodes: int, flags: int, mode: int, transposed: bool,
                       scales, g_adj, g_raw, g_xp, g_s, g_la, g_lb, want_gx: bool, want_gw: bool, want_gb: bool):
    """(gX, gW, gbias) of the rows route's training step as ONE native call (``tgp_pool_rows_bwd_f32``) for the common
    case -- the selector folded in, A = A^T --, or None when an operand is not float32 / contiguous (the caller then makes
    the same launches one by one).  See functions._PoolUnbatchedFn.backward for the arithmetic."""
    if not _POOL_ROWS_ONE_CALL:
        return None
    n, Kc = s.shape
    F = x.size(1)
    B = ptr.numel() - 1
    dev = s.device
    f32 = torch.float32

    def ok(v, shape=None):
        return v is None or (v.dtype == f32 and v.is_contiguous() and (shape is None or tuple(v.shape) == tuple(shape)))

    if not (ok(s) and ok(t, (n, Kc)) and ok(x, (n, F)) and ok(weight, (Kc, F)) and ok(raw, (B, Kc, Kc)) and ok(gram)
            and ok(stats) and ok(den) and ok(deg) and ok(lossv) and ptr.dtype == torch.int64 and ptr.is_contiguous()
            and (batch is None or (batch.dtype == torch.int64 and batch.is_contiguous())) and ok(g_raw, (B, Kc, Kc))
            and ok(g_s, (n, Kc)) and (g_la is None or g_la.dtype == f32) and (g_lb is None or g_lb.dtype == f32)):
        return None
    ga_bc = gx_bc = False
    if g_adj is not None:
        if tuple(g_adj.shape) != (B, Kc, Kc) or g_adj.dtype != f32:
            return None
        g_adj, ga_bc = _bcast_or_dense(g_adj, (B, Kc, Kc))
    if g_xp is not None:
        if tuple(g_xp.shape) != (B, Kc, F) or g_xp.dtype != f32:
            return None
        g_xp, gx_bc = _bcast_or_dense(g_xp, (B, Kc, F))
    pad = TRAIN_PAD
    ld = 3 * Kc + F + pad
    e = dict(dtype=f32, device=dev)
    ga = torch.empty(B, Kc, Kc, **e) if g_adj is not None else None
    rcat = torch.empty(B, ld, Kc, **e)
    c1 = torch.empty(B, **e) if mode == 1 else None
    gwcat = torch.empty(B, 2 * Kc, F, **e) if want_gx else None
    acat = torch.empty(n, ld, **e)
    gs = torch.empty(n, Kc, **e)
    gx = torch.empty(n, F, **e) if want_gx else None
    slabs = slab_ptr.numel() - 1
    part = torch.empty(slabs, Kc, F + pad, **e) if (want_gw or want_gb) else None
    gw = torch.empty(Kc, F, **e) if want_gw else None
    gb = torch.empty(Kc, **e) if want_gb else None
    N.check(N.lib().tgp_pool_rows_bwd_f32(
        s.data_ptr(), t.data_ptr(), x.data_ptr(), weight.data_ptr(), raw.data_ptr(), N.ptr(gram), N.ptr(stats), N.ptr(den),
        N.ptr(deg), N.ptr(lossv), ptr.data_ptr(), N.ptr(batch), slab_ptr.data_ptr(), slabs, n, B, Kc, F, max_nodes,
        int(flags), ops_eps(), losses_eps(), int(mode), 1 if transposed else 0, 1.0 / B,
        float(scales[0]) if mode == 2 else 0.0, float(scales[1]) if mode == 2 else 0.0, N.ptr(g_adj), 1 if ga_bc else 0,
        N.ptr(g_raw), N.ptr(g_xp), 1 if gx_bc else 0, N.ptr(g_s), N.ptr(g_la), N.ptr(g_lb), N.ptr(ga), rcat.data_ptr(),
        N.ptr(c1), N.ptr(gwcat), acat.data_ptr(), gs.data_ptr(), N.ptr(gx), N.ptr(part), N.ptr(gw), N.ptr(gb),
        N.stream_ptr(dev)), "tgp_pool_rows_bwd_f32")
    return gx, gw, gb


def segment_gemm_nn_into(a: Tensor, bm: Tensor, ptr: Tensor, out: Tensor, max_nodes: int) -> Tensor:
    """out[rows of graph b] = a[rows of graph b] @ bm[b] for float32 VIEWS with unit last stride: ``a`` [Ntot,Kd] and
    ``out`` [Ntot,Nc] with any row stride (column blocks of a wider buffer), ``bm`` [B,Kd,Nc] with any row / batch stride."""
    dev = N.require_device(a, bm, ptr, out)
    if (a.dim() != 2 or bm.dim() != 3 or out.dim() != 2 or a.stride(1) != 1 or bm.stride(2) != 1 or out.stride(1) != 1
            or bm.size(1) != a.size(1) or out.size(1) != bm.size(2) or out.size(0) != a.size(0)
            or any(t.dtype != torch.float32 for t in (a, bm, out))):
        raise ValueError(f"segment_gemm_nn_into: {tuple(a.shape)} x {tuple(bm.shape)} -> {tuple(out.shape)}")
    ptr = N.i64c(ptr)
    N.check(N.lib().tgp_segment_gemm_nn_ld_f32(a.data_ptr(), a.stride(0), bm.data_ptr(), bm.stride(1), bm.stride(0),
                                               N.ptr(ptr), out.data_ptr(), out.stride(0), ptr.numel() - 1, a.size(0),
                                               a.size(1), bm.size(2), max_nodes, N.stream_ptr(dev)),
            "tgp_segment_gemm_nn_ld_f32")
    return out


def segment_gemm_tn_into(a: Tensor, y: Tensor, ptr: Tensor) -> Tensor:
    """[a[rows of b]^T @ y[rows of b] for b] -> [B,M,Nc] for float32 2-D VIEWS with unit last stride and any row stride
    (no node-range split: meant for short row ranges, e.g. the row slabs of a weight gradient)."""
    dev = N.require_device(a, y, ptr)
    if (a.dim() != 2 or y.dim() != 2 or a.stride(1) != 1 or y.stride(1) != 1 or a.size(0) != y.size(0)
            or a.dtype != torch.float32 or y.dtype != torch.float32):
        raise ValueError(f"segment_gemm_tn_into: {tuple(a.shape)}^T x {tuple(y.shape)}")
    ptr = N.i64c(ptr)
    B = ptr.numel() - 1
    out = torch.empty(B, a.size(1), y.size(1), dtype=torch.float32, device=dev)
    N.check(N.lib().tgp_segment_gemm_tn_ld_f32(a.data_ptr(), a.stride(0), y.data_ptr(), y.stride(0), N.ptr(ptr),
                                               N.ptr(out), B, a.size(0), a.size(1), y.size(1), N.stream_ptr(dev)),
            "tgp_segment_gemm_tn_ld_f32")
    return out


def diffpool_unbatched_tail(raw: Tensor, gram: Tensor, s: Tensor, sw2, link_scale: float, ent_scale: float,
                            ent_partials: Optional[Tensor] = None) -> Tensor:
    """[2]: DiffPool's unbatched link-prediction loss sqrt(max(sum_e w_e^2 - 2 sum_b trace(raw_b) + sum_b |gram_b|^2, 0))
    * link_scale and entropy loss sum(-s log(s + eps)) * ent_scale (utils/losses.py:661-708, 476-483); ``sw2``: a
    0-dim device tensor or a Python number.  ``ent_partials``: shares of the entropy sum that are already there
    (``spmm_csr(want_stats="entropy")``): no pass over S here."""
    dev = N.require_device(raw, gram, s, ent_partials)
    raw, gram = N.f32c(raw), N.f32c(gram)
    B, Kc = raw.size(0), raw.size(-1)
    L = N.lib()
    st = N.stream_ptr(dev)
    import ctypes as _ct
    n_partial = _ct.c_int(0)
    if ent_partials is not None:
        ws = N.f32c(ent_partials.reshape(-1))
        n_partial.value = ws.numel()
    else:
        s32 = N.f32c(s)
        ws = N.workspace(L.tgp_entropy_sum_workspace_bytes(s32.numel()), dev)
        N.check(L.tgp_entropy_partials_f32(N.ptr(s32), s32.numel(), losses_eps(), N.ptr(ws), ws.numel(),
                                           _ct.addressof(n_partial), st), "tgp_entropy_partials_f32")
    stats = torch.empty(B, 2, dtype=torch.float32, device=dev)
    out = torch.empty(2, dtype=torch.float32, device=dev)
    sw2_dev = N.f32c(sw2.reshape(1)) if isinstance(sw2, Tensor) else None
    N.check(L.tgp_diffpool_unbatched_tail_f32(N.ptr(raw), N.ptr(gram), B, Kc, N.ptr(sw2_dev),
                                              0.0 if sw2_dev is not None else float(sw2), N.ptr(ws), n_partial.value,
                                              float(link_scale), float(ent_scale), N.ptr(stats), N.ptr(out), st),
            "tgp_diffpool_unbatched_tail_f32")
    return out


# ------------------------------------------------------------------------- A14 (NDP)
def ndp_max_graph_nodes() -> int:
    return int(N.lib().tgp_ndp_max_graph_nodes())


def ndp_symmetric_max(edge_index: Tensor, weight: Optional[Tensor], num_nodes: int, indptr: Tensor):
    """(w_max [E], flag [1] int32): for a list that already is row-major sorted, duplicate-free, loop-free and
    pattern-symmetric, the weights NDPSelect's symmetrisation would produce (max with the reverse entry) and flag = 0;
    flag = 1 when the list is anything else (ndp_select.py:198-202 then needs the general route)."""
    dev = N.require_device(edge_index, weight, indptr)
    row, col = _edge_rows(edge_index)
    w = None if weight is None else N.f32c(weight.reshape(-1))
    out = torch.empty(row.numel(), dtype=torch.float32, device=dev)
    flag = torch.empty(1, dtype=torch.int32, device=dev)
    N.check(N.lib().tgp_ndp_symmetric_max_f32(N.ptr(row), N.ptr(col), N.ptr(w), row.numel(), num_nodes, N.ptr(indptr),
                                              N.ptr(out), N.ptr(flag), N.stream_ptr(dev)), "tgp_ndp_symmetric_max_f32")
    return out, flag


def ndp_partition(indptr: Tensor, col: Tensor, weight: Optional[Tensor], num_nodes: int, graph_ptr: Tensor,
                  max_graph_nodes: int, seed: int, max_iter: int = 500, tol: float = 1e-6, raw_keep: bool = False):
    """(keep [N] bool, info [B] int32, status int): NDPSelect's per-graph spectral +-1 partition
    (select/ndp_select.py:187-256) on a symmetric, self-loop-free CSR adjacency.  info[g] = LOBPCG steps used, -1 =
    the reference's random fallback (cut < 0.5); status != 0: declined, see include/tgp_hip.h.  ``tol``: relative
    eigen-residual |Ls x - lambda x| <= tol * lambda at which the iteration stops."""
    dev = N.require_device(indptr, col, weight, graph_ptr)
    if indptr.dtype != torch.int32:
        raise ValueError("ndp_partition: indptr must be int32")
    col, graph_ptr = N.i64c(col), N.i64c(graph_ptr)
    w = None if weight is None else N.f32c(weight.reshape(-1))
    B = graph_ptr.numel() - 1
    # status | info | keep in one allocation, laid out as tgp_ndp_partition clears them with a single memset (r6)
    b_pad, n_keep = (max(B, 1) + 3) // 4 * 4, max(num_nodes, 1)
    buf = torch.empty(16 + 4 * b_pad + n_keep, dtype=torch.uint8, device=dev)
    status = buf[:4].view(torch.int32)
    info = buf[16:16 + 4 * b_pad].view(torch.int32)
    keep = buf[16 + 4 * b_pad:]
    N.check(N.lib().tgp_ndp_partition(N.ptr(indptr.contiguous()), N.ptr(col), N.ptr(w), num_nodes, col.numel(),
                                      N.ptr(graph_ptr), B, max_graph_nodes, int(seed) & ((1 << 64) - 1), max_iter,
                                      float(tol), N.ptr(keep), N.ptr(info), N.ptr(status), N.stream_ptr(dev)),
            "tgp_ndp_partition")
    return (keep[:num_nodes] if raw_keep else keep[:num_nodes].bool()), info[:B], status


_MASK_INDEX_SCRATCH: dict = {}


def mask_index(mask: Tensor, declined: Optional[Tensor] = None, want_rank: bool = False, want_ones: bool = False,
               want_assign: bool = False, want_node_rank: bool = False):
    """The sorted positions of the non-zero bytes of ``mask`` [n] uint8 -- `mask.nonzero().view(-1)` in two launches with
    one pinned-word wait between them (torch's own: seven launches and a synchronising copy) -- as row 0 of an int64
    ``[2, k]`` array whose row 1 is ``arange(k)`` (``want_rank``; the indices of NDPSelect's S), plus ``ones [k]`` fp32
    (``want_ones``; its values).  ``declined`` (int32 [1] on the device): when non-zero the call returns None (the flag
    of the kernel that made the mask; read with the count, no copy of its own).  Returns (index [2,k] or [1,k], ones or
    None) -- and with ``want_assign`` (needs both) a third item: the one-to-one :class:`AssignIndex` (perm + packed
    {row, weight}) of the assignment these arrays describe, which :func:`one_to_one_index` would build in a launch of its
    own.  ``want_node_rank``: one more item, int32 ``[n + 1]`` = set positions in front of every position (total at
    ``[n]``): what :func:`kron_batched` takes as ``node_rank``.  reference: select/ndp_select.py:257-262."""
    dev = N.require_device(mask, declined)
    if mask.dtype != torch.uint8 or not mask.is_contiguous():
        raise ValueError("mask_index: a contiguous uint8 mask is required")
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("mask_index reads its count on the host between two launches: not capturable")
    n = mask.numel()
    L = N.lib()
    st = N.stream_ptr(dev)
    words = int(L.tgp_mask_index_scratch_words(n))
    key = (dev.index, st)
    scratch = _MASK_INDEX_SCRATCH.get(key)
    if scratch is None or scratch.numel() < words:
        scratch = torch.zeros(max(words, 1024), dtype=torch.int32, device=dev)
        _MASK_INDEX_SCRATCH[key] = scratch
    state = _sps_state(dev, st, 0)
    epoch = state.next_epoch()
    N.check(L.tgp_mask_index_count(N.ptr(mask), n, N.ptr(declined), N.ptr(scratch), state.pinned.data_ptr(), epoch, st),
            "tgp_mask_index_count")
    k = _decode_count(state.wait(epoch))
    if k < 0:
        return None
    rows = 2 if want_rank else 1
    index = torch.empty((rows, k), dtype=torch.long, device=dev)
    ones = torch.empty(k, dtype=torch.float32, device=dev) if want_ones else None
    assign = want_assign and want_rank and want_ones and 0 < k and n < (1 << 31)
    perm = torch.empty(k, dtype=torch.int32, device=dev) if assign else None
    pack = torch.empty(k, dtype=torch.int64, device=dev) if assign else None
    node_rank = torch.empty(n + 1, dtype=torch.int32, device=dev) if want_node_rank and n else None
    if k or node_rank is not None:
        N.check(L.tgp_mask_index_fill(N.ptr(mask), n, N.ptr(scratch), k, N.ptr(index) if k else None,
                                      index.data_ptr() + 8 * k if want_rank and k else None, N.ptr(ones) if k else None,
                                      N.ptr(perm), N.ptr(pack), N.ptr(node_rank), st), "tgp_mask_index_fill")
    out = (index, ones)
    if want_assign:
        a_index = None
        if assign:
            a_index = AssignIndex(None, perm, k, k)
            a_index.pack, a_index.pack_key = pack, (index.data_ptr(), ones.data_ptr())
        out = out + (a_index,)
    if want_node_rank:
        out = out + (node_rank,)
    return out


def ndp_partition_large(indptr: Tensor, col: Tensor, weight: Optional[Tensor], p0: int, p1: int, seed: int,
                        keep: Tensor, status: Tensor, max_iter: int = 2000, tol: float = 1e-6,
                        steps_per_batch: int = 16, want_state: bool = False):
    """NDPSelect's spectral +-1 partition (select/ndp_select.py:187-256) of ONE large graph -- the nodes [p0, p1) of a
    symmetric, self-loop-free CSR -- on the whole chip: the LOBPCG iteration of :func:`ndp_partition` with grid-wide
    vector kernels (tgp_ndp_large_*).  Writes keep[p0:p1] (uint8 view of the caller's mask); returns the steps used
    (-1: the reference's random fallback).  One host read per ``steps_per_batch`` steps (the done flag)."""
    dev = N.require_device(indptr, col, weight, keep, status)
    if indptr.dtype != torch.int32 or keep.dtype != torch.uint8 or status.dtype != torch.int32:
        raise ValueError("ndp_partition_large: indptr / status must be int32, keep uint8")
    col = N.i64c(col)
    w = None if weight is None else N.f32c(weight.reshape(-1))
    L = N.lib()
    n = p1 - p0
    ws = N.workspace(L.tgp_ndp_large_workspace_bytes(n), dev)
    st = N.stream_ptr(dev)
    N.check(L.tgp_ndp_large_start(N.ptr(indptr), N.ptr(col), N.ptr(w), p0, p1, max_iter, N.ptr(ws), ws.numel(),
                                  N.ptr(status), st), "tgp_ndp_large_start")
    progress = torch.zeros(2, dtype=torch.int32, device=dev)
    done = False
    while not done:
        N.check(L.tgp_ndp_large_steps(N.ptr(indptr), N.ptr(col), N.ptr(w), p0, p1, steps_per_batch, float(tol),
                                      N.ptr(ws), ws.numel(), N.ptr(progress), N.ptr(status), st),
                "tgp_ndp_large_steps")
        flag, it = progress.tolist()  # one round trip per batch of steps
        done = flag != 0 or it >= max_iter
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    N.check(L.tgp_ndp_large_finish(N.ptr(indptr), N.ptr(col), N.ptr(w), p0, p1, int(seed) & ((1 << 64) - 1),
                                   N.ptr(ws), ws.numel(), N.ptr(keep), N.ptr(info), st), "tgp_ndp_large_finish")
    if want_state:
        import ctypes as _ct
        out = (_ct.c_double * 5)()
        N.check(L.tgp_ndp_large_state(N.ptr(ws), n, _ct.addressof(out), st), "tgp_ndp_large_state")
        return info, {"lambda": out[0], "residual_sq": out[1], "steps": int(out[2]), "random": bool(out[3]),
                      "cut": out[4]}
    return info


# ------------------------------------------------------------------------- A9
def kron_max_graph_nodes() -> int:
    return int(N.lib().tgp_kron_batched_max_graph_nodes())


_KRON_CAPS: dict = {}


def _kron_caps(graph_sizes_host, limit: Optional[int] = None) -> Tuple[int, int, int]:
    """Workspace figures of the Kron kernels from the graphs' node counts: (sum of n^2 over the graphs inside the size
    limit of THIS call -- the kernels' own, or the smaller ``limit`` a caller declared: graphs beyond it are skipped by
    the kernel and get no slab --, sum of n (n | 1) over those beyond the LDS kernel's 128 nodes, how many those are).
    Vectorised and remembered per sizes list (r5: three Python passes over 2048 graphs were ~120 us of every NDP pooler
    call); the memo holds the list object and re-checks its length, its sum and the limit, so a list edited in place
    does not hand back stale figures."""
    import numpy as np
    lim = kron_max_graph_nodes() if limit is None else max(1, min(int(limit), kron_max_graph_nodes()))
    key = id(graph_sizes_host)
    hit = _KRON_CAPS.get(key)
    v = None
    if hit is not None and hit[0] is graph_sizes_host and hit[1] == len(graph_sizes_host) and hit[3] == lim:
        v = np.asarray(graph_sizes_host, dtype=np.int64)
        if int(v.sum()) == hit[4]:
            return hit[2]
    if v is None:
        v = np.asarray(graph_sizes_host, dtype=np.int64)
    total = int(v.sum())
    v = v[v <= lim]
    big = v[v > 128]
    caps = (int((v * v).sum()), int((big * (big | 1)).sum()), int(big.size))
    if len(_KRON_CAPS) > 16:
        _KRON_CAPS.clear()
    _KRON_CAPS[key] = (graph_sizes_host, len(graph_sizes_host), caps, lim, total)  # (holds the list: the id stays its own)
    return caps


def kron_batched(indptr: Tensor, col: Tensor, val: Optional[Tensor], perm: Optional[Tensor], from_adjacency: bool,
                 num_nodes: int, graph_ptr: Tensor, max_graph_nodes: int, node_index: Tensor,
                 threshold: float, skip_oversize: bool = False,
                 graph_sizes_host: Optional[Sequence[int]] = None,
                 node_rank: Optional[Tensor] = None) -> Optional[Tuple[Tensor, Tensor]]:
    """Block-batched Kron reduction (connect/kron_conn.py:117-165): one workgroup per graph, fp64 elimination of the
    dropped nodes, thresholded fp32 edge list in row-major order.  ``indptr`` int32 [N+1] / ``col`` int64 / ``val``
    fp32 or fp64 (None = ones) / ``perm`` int32 (None = identity) describe the Laplacian entries, or the edge weights
    when ``from_adjacency``.  ``graph_sizes_host``: the graphs' node counts on the host (the caller's memoised
    batch info): the workspace is then sized from the real graphs instead of num_nodes x longest graph.
    ``node_rank`` (int32 [num_nodes + 1], optional): kept nodes in front of every node, from the selector that compacted
    ``node_index`` itself (:func:`mask_index`): the call skips its flag scatter and scan.
    Returns None when the library declines (see include/tgp_hip.h)."""
    dev = N.require_device(indptr, col, val, perm, graph_ptr, node_index)
    cap_dense = cap_big = num_big = -1
    if graph_sizes_host is not None:
        cap_dense, cap_big, num_big = _kron_caps(graph_sizes_host, max_graph_nodes)
    if indptr.dtype != torch.int32 or (perm is not None and perm.dtype != torch.int32):
        raise ValueError("kron_batched: indptr / perm must be int32")
    col, graph_ptr, node_index = N.i64c(col), N.i64c(graph_ptr), N.i64c(node_index)
    v32 = v64 = None
    if val is not None:
        if val.dtype == torch.float64:
            v64 = val.contiguous()
        else:
            v32 = N.f32c(val.reshape(-1))
    B = graph_ptr.numel() - 1
    L = N.lib()
    if node_rank is not None and not (node_rank.dtype == torch.int32 and node_rank.numel() == num_nodes + 1
                                      and node_rank.is_contiguous() and node_rank.device == dev):
        node_rank = None  # (not the table of THIS batch: the call builds its own)
    ws = N.workspace(L.tgp_kron_batched_workspace_bytes(num_nodes, B, max_graph_nodes, cap_dense, cap_big), dev)
    d_count = torch.empty(1, dtype=torch.int64, device=dev)
    st = N.stream_ptr(dev)
    N.check(L.tgp_kron_batched_count(N.ptr(indptr.contiguous()), N.ptr(col), N.ptr(v32), N.ptr(v64),
                                     N.ptr(None if perm is None else perm.contiguous()),
                                     (1 if from_adjacency else 0) | (2 if skip_oversize else 0),
                                     num_nodes, col.numel(), N.ptr(graph_ptr), B, max_graph_nodes, cap_dense, cap_big,
                                     num_big, N.ptr(node_index),
                                     node_index.numel(), float(threshold), N.ptr(node_rank), N.ptr(ws), ws.numel(),
                                     N.ptr(d_count), st),
            "tgp_kron_batched_count")
    n_out = _read_count(d_count)
    if n_out < 0:
        return None
    ei = torch.empty(2, n_out, dtype=torch.int64, device=dev)
    ew = torch.empty(n_out, dtype=torch.float32, device=dev)
    N.check(L.tgp_kron_batched_fill(N.ptr(ws), num_nodes, B, max_graph_nodes, cap_dense, cap_big, num_big,
                                    N.ptr(graph_ptr), n_out,
                                    N.ptr(ei[0]) if n_out else None, N.ptr(ei[1]) if n_out else None,
                                    N.ptr(ew) if n_out else None, N.ptr(node_rank), st), "tgp_kron_batched_fill")
    return ei, ew


# ------------------------------------------------------------------------- A10
def block_diag_edges(adj_pool: Tensor, relabel: Optional[Tensor] = None,
                     remove_self_loops: bool = False) -> Tuple[Tensor, Tensor]:
    """dense_to_block_diag (utils/ops.py:53-82) with the optional valid-supernode renumbering of
    DenseSRCPooling._finalize_sparse_output (src.py:526-552)."""
    dev = N.require_device(adj_pool, relabel)
    f64 = adj_pool.dtype == torch.float64
    a = (N.f64c if f64 else N.f32c)(adj_pool if adj_pool.dim() == 3 else adj_pool.unsqueeze(0))
    B, K = a.size(0), a.size(1)
    rl = None if relabel is None else N.i64c(relabel)
    flags = N.REMOVE_SELF_LOOPS if remove_self_loops else 0
    L = N.lib()
    ws = N.workspace(L.tgp_block_diag_workspace_bytes(B, K), dev)
    d_count = torch.empty(1, dtype=torch.int64, device=dev)
    st = N.stream_ptr(dev)
    eps = ops_eps()
    count, fill = ((L.tgp_block_diag_count_f64, L.tgp_block_diag_fill_f64) if f64 else
                   (L.tgp_block_diag_count, L.tgp_block_diag_fill))
    N.check(count(N.ptr(a), B, K, N.ptr(rl), flags, eps, N.ptr(ws), ws.numel(), N.ptr(d_count), st),
            "tgp_block_diag_count")
    n_out = _read_count(d_count)
    ei = torch.empty(2, n_out, dtype=torch.int64, device=dev)
    ew = torch.empty(n_out, dtype=a.dtype, device=dev)
    N.check(fill(N.ptr(a), B, K, N.ptr(rl), flags, eps, N.ptr(ws), n_out,
                                  N.ptr(ei[0]) if n_out else None, N.ptr(ei[1]) if n_out else None,
                                  N.ptr(ew) if n_out else None, st), "tgp_block_diag_fill")
    return ei, ew


# ------------------------------------------------------------------------- A11
def to_dense_adj(edge_index: Tensor, edge_weight: Optional[Tensor], batch: Tensor, ptr: Tensor, num_graphs: int,
                 max_nodes: int, transposed: bool, zeroed_out: Optional[Tensor] = None) -> Tensor:
    """PyG to_dense_adj (src.py:434-443): [B,Nmax,Nmax], duplicates summed; optionally written transposed.
    ``zeroed_out``: a zero-filled [B,Nmax,Nmax] buffer to add into (:func:`to_dense_batch` can zero it in its launch)."""
    dev = N.require_device(edge_index, edge_weight, batch, ptr)
    row, col = _edge_rows(edge_index)
    w = None if edge_weight is None else N.f32c(edge_weight.reshape(-1))
    batch, ptr = N.i64c(batch), N.i64c(ptr)
    adj = zeroed_out if zeroed_out is not None else torch.empty(num_graphs, max_nodes, max_nodes, dtype=torch.float32,
                                                                device=dev)
    if tuple(adj.shape) != (num_graphs, max_nodes, max_nodes) or adj.dtype != torch.float32 or not adj.is_contiguous():
        raise ValueError("to_dense_adj: zeroed_out must be a contiguous float32 [B,Nmax,Nmax] tensor")
    N.check(N.lib().tgp_to_dense_adj_f32(N.ptr(row), N.ptr(col), N.ptr(w), row.numel(), N.ptr(batch), N.ptr(ptr),
                                         num_graphs, max_nodes, 1 if transposed else 0, 1 if zeroed_out is not None else 0,
                                         N.ptr(adj), N.stream_ptr(dev)), "tgp_to_dense_adj_f32")
    return adj


def to_dense_adj_channels(edge_index: Tensor, edge_attr: Tensor, batch: Tensor, ptr: Tensor, num_graphs: int,
                          max_nodes: int, transposed: bool) -> Tensor:
    """PyG to_dense_adj with multi-channel edge attributes [E, C] (src.py:434): [B,Nmax,Nmax,C], duplicates summed."""
    dev = N.require_device(edge_index, edge_attr, batch, ptr)
    row, col = _edge_rows(edge_index)
    E = row.numel()
    a = N.f32c(edge_attr.reshape(E, -1))
    C = a.size(1)
    batch, ptr = N.i64c(batch), N.i64c(ptr)
    adj = torch.empty(num_graphs, max_nodes, max_nodes, C, dtype=torch.float32, device=dev)
    N.check(N.lib().tgp_to_dense_adj_channels_f32(N.ptr(row), N.ptr(col), N.ptr(a), E, C, N.ptr(batch), N.ptr(ptr),
                                                  num_graphs, max_nodes, 1 if transposed else 0, N.ptr(adj),
                                                  N.stream_ptr(dev)), "tgp_to_dense_adj_channels_f32")
    return adj.view((num_graphs, max_nodes, max_nodes) + tuple(edge_attr.shape[1:]))


def from_dense_adj(grad_adj: Tensor, edge_index: Tensor, batch: Tensor, ptr: Tensor, max_nodes: int,
                   transposed: bool) -> Tensor:
    """grad of :func:`to_dense_adj` w.r.t. the edge weights: one gather kernel."""
    dev = N.require_device(grad_adj, edge_index, batch, ptr)
    row, col = _edge_rows(edge_index)
    g = N.f32c(grad_adj)
    batch, ptr = N.i64c(batch), N.i64c(ptr)
    out = torch.empty(row.numel(), dtype=torch.float32, device=dev)
    N.check(N.lib().tgp_from_dense_adj_f32(N.ptr(g), N.ptr(row), N.ptr(col), row.numel(), N.ptr(batch), N.ptr(ptr),
                                           g.size(0), max_nodes, 1 if transposed else 0, N.ptr(out),
                                           N.stream_ptr(dev)), "tgp_from_dense_adj_f32")
    return out


def from_dense_batch(dense: Tensor, batch: Tensor, ptr: Tensor, max_nodes: int) -> Tensor:
    """[B,Nmax,F] -> [N,F]: the rows that to_dense_batch scattered, gathered back (its backward)."""
    dev = N.require_device(dense, batch, ptr)
    d = N.f32c(dense.reshape(dense.size(0), dense.size(1), -1))
    batch, ptr = N.i64c(batch), N.i64c(ptr)
    n, F = batch.numel(), d.size(2)
    out = torch.empty(n, F, dtype=torch.float32, device=dev)
    N.check(N.lib().tgp_from_dense_batch_f32(N.ptr(d), n, F, N.ptr(batch), N.ptr(ptr), d.size(0), max_nodes, N.ptr(out),
                                             N.stream_ptr(dev)), "tgp_from_dense_batch_f32")
    return out.view((n,) + tuple(dense.shape[2:]))


def to_dense_batch(x: Tensor, batch: Tensor, ptr: Tensor, num_graphs: int, max_nodes: int,
                   also_zero: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """PyG to_dense_batch (src.py:448-450): ([B,Nmax,F], mask [B,Nmax]).  ``also_zero``: a contiguous float32 buffer to
    zero-fill on the way (in the same launch when the batch vector is sorted)."""
    dev = N.require_device(x, batch, ptr)
    x2 = N.f32c(x if x.dim() == 2 else x.reshape(x.size(0), -1))
    batch, ptr = N.i64c(batch), N.i64c(ptr)
    F = x2.size(1)
    out = torch.empty(num_graphs, max_nodes, F, dtype=torch.float32, device=dev)
    mask = torch.empty(num_graphs, max_nodes, dtype=torch.bool, device=dev)
    from .utils.ops import batch_info
    if F > 0 and ptr.numel() == num_graphs + 1 and batch_info(batch).is_sorted:
        # sorted batch vector (memoised fact): one output-parallel kernel writes rows, padding and mask -- no memsets
        N.check(N.lib().tgp_to_dense_batch_sorted_f32(N.ptr(x2), x2.size(0), F, N.ptr(ptr), num_graphs, max_nodes,
                                                      N.ptr(out), N.ptr(mask), N.ptr(also_zero),
                                                      also_zero.numel() if also_zero is not None else 0,
                                                      N.stream_ptr(dev)),
                "tgp_to_dense_batch_sorted_f32")
        return out.view((num_graphs, max_nodes) + tuple(x.shape[1:])), mask
    if also_zero is not None:
        also_zero.zero_()
    N.check(N.lib().tgp_to_dense_batch_f32(N.ptr(x2), x2.size(0), F, N.ptr(batch), N.ptr(ptr), num_graphs, max_nodes,
                                           N.ptr(out), N.ptr(mask), N.stream_ptr(dev)), "tgp_to_dense_batch_f32")
    return out.view((num_graphs, max_nodes) + tuple(x.shape[1:])), mask
