"""Multi-GPU data parallelism for the pooling hot path: one process per GPU, graphs sharded by id,
pooled outputs all-gathered with ``torch.distributed`` (backend ``nccl`` == RCCL over xGMI on ROCm).

The reference has no distributed code at all; graphs of a batch are independent in
Select, Reduce, Connect and their per-graph normalisations, so the path shards without any data-path
collective.  The only exchange is the final gather of the (small) pooled outputs, whose merge rule is
the one the reference uses when it collates pooled graphs on the CPU (tgp/data/collate.py:144-153):
node ids and graph ids of later shards are shifted by the totals of the earlier ones.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch import Tensor


import ctypes as _ctypes

_COPY_SRC, _COPY_DST, _COPY_BYTES = (_ctypes.c_void_p * 8)(), (_ctypes.c_void_p * 8)(), (_ctypes.c_int64 * 8)()


def shard_bounds(num_graphs: int, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced graph-id ranges (``batch`` stays sorted inside every shard)."""
    base, extra = divmod(num_graphs, world_size)
    out, lo = [], 0
    for r in range(world_size):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def shard_sparse_batch(x: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor], batch: Tensor,
                       rank: int, world_size: int, num_graphs: Optional[int] = None):
    """Slice a PyG-style batch (sorted ``batch``) down to this rank's graphs; node ids and graph ids are
    made local (start at 0)."""
    if num_graphs is None:
        num_graphs = int(batch.max()) + 1 if batch.numel() else 0
    lo, hi = shard_bounds(num_graphs, world_size)[rank]
    node_sel = (batch >= lo) & (batch < hi)
    nodes = node_sel.nonzero().view(-1)
    n0 = int(nodes[0]) if nodes.numel() else 0
    edge_sel = node_sel[edge_index[0]]
    ei = edge_index[:, edge_sel] - n0
    ew = None if edge_weight is None else edge_weight[edge_sel]
    return x[nodes], ei, ew, batch[nodes] - lo


def shard_dense_batch(rank: int, world_size: int, *tensors: Tensor):
    """Slice padded dense tensors [B, ...] along the graph dimension."""
    lo, hi = shard_bounds(tensors[0].size(0), world_size)[rank]
    return tuple(t[lo:hi] for t in tensors)


def _world(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


class PackedGather:
    """RCCL all-gather of fixed-shape pooled outputs, bucketed, without pack copies.

    The per-graph tensors of a step ([B,K,F], [B,K,K], ...) live in one flat send buffer, every tensor a contiguous
    slice of its step's region; the regions of ``bucket_steps`` consecutive steps go out as ONE
    ``all_gather_into_tensor`` (fewer, larger collectives: at 8 GPUs a 3 MB-per-rank gather is latency-bound on xGMI,
    four steps' worth is not).  ``slots(shapes)`` hands out the next step's slices so that the pooling kernels write
    their outputs straight into the send buffer (``DenseSRCPooling.reduce_connect(out_x=, out_adj=)``); ``start``
    then only books the step (tensors that are not those slices are copied in).  The collective is asynchronous on
    RCCL's own stream, so it overlaps the next steps' kernels; at most one is in flight and the send buffers are
    double-buffered, so a step never waits for it.  Every rank must hold the same number of graphs.

    ``start(tensors)`` once per step; ``take_ready()`` -> finished results without waiting; ``wait()`` -> also
    waits for the collective in flight; ``flush()`` -> sends a partly filled bucket and returns everything not
    yet handed out.  A result is the list of gathered tensors of one step (rank-major along the graph dimension, like
    ``torch.cat`` of the shards); with ``bucket_steps == 1`` ``wait()`` returns that list directly."""

    def __init__(self, group=None, bucket_steps: int = 1, force_collective: bool = False):
        if bucket_steps < 1:
            raise ValueError("bucket_steps must be >= 1")
        self.group = group
        self.world = _world(group)
        # a one-rank process group still goes through the collective when asked to (single-GPU test of the RCCL
        # path: ``force_collective=True``, which bench.py passes under TGP_BENCH_FORCE_DIST=1, or TGP_FORCE_COLLECTIVE)
        self._collective = self.world > 1 or (dist.is_available() and dist.is_initialized()
                                              and (force_collective or bool(os.environ.get("TGP_FORCE_COLLECTIVE"))))
        self.bucket = bucket_steps
        self._send = [None, None]   # double-buffered flat [bucket * per_step]
        self._cur = 0
        self._fill = 0
        self._shapes = None         # full shapes [B, ...] of one step's tensors
        self._per_step = 0
        self._pending = None        # (work, out, nsteps, shapes, per_step)
        self._ready: List[List[Tensor]] = []

    # ---- layout ---------------------------------------------------------------------------------
    def _ensure(self, shapes, dtype, device) -> None:
        shapes = [tuple(int(d) for d in shp) for shp in shapes]
        per_step = 0
        for shp in shapes:
            n = 1
            for d in shp:
                n *= d
            per_step += n
        buf = self._send[self._cur]
        if (buf is None or self._shapes != shapes or buf.dtype != dtype or buf.device != device):
            if self._fill:
                raise ValueError("PackedGather: shapes changed inside an open bucket")
            self._collect()  # nothing of the old layout may still be in flight
            self._send = [torch.empty(self.bucket * per_step, dtype=dtype, device=device) for _ in range(2)]
            self._shapes, self._per_step = shapes, per_step

    def _views(self, flat: Tensor, step: int, lead=None) -> List[Tensor]:
        out, off = [], step * self._per_step
        for shp in self._shapes:
            n = 1
            for d in shp:
                n *= d
            out.append(flat[off: off + n].view(shp))
            off += n
        return out

    def slots(self, shapes, dtype=torch.float32, device=None) -> List[Tensor]:
        """Contiguous views of the NEXT step's region of the send buffer, one per tensor shape [B, ...]: write the
        step's outputs there, then call ``start`` with exactly these tensors."""
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else "cpu"
        self._ensure(shapes, dtype, torch.device(device))
        return self._views(self._send[self._cur], self._fill)

    # ---- collective -----------------------------------------------------------------------------
    def _unpack(self, out: Tensor, nsteps: int, shapes, per_step: int) -> List[List[Tensor]]:
        # out: [world, nsteps * per_step]
        res = []
        for j in range(nsteps):
            tensors, off = [], j * per_step
            for shp in shapes:
                n = 1
                for d in shp:
                    n *= d
                tensors.append(out[:, off: off + n].reshape((out.size(0) * shp[0],) + tuple(shp[1:])))
                off += n
            res.append(tensors)
        return res

    def _collect(self) -> None:
        if self._pending is None:
            return
        work, out, nsteps, shapes, per_step = self._pending
        self._pending = None
        if work is not None:
            work.wait()
        self._ready.extend(self._unpack(out, nsteps, shapes, per_step))

    def _launch(self) -> None:
        self._collect()  # at most one collective in flight; its send buffer becomes free here
        n = self._fill
        send = self._send[self._cur][: n * self._per_step]
        out = torch.empty((self.world, send.numel()), dtype=send.dtype, device=send.device)
        work = None
        if self._collective:
            work = dist.all_gather_into_tensor(out.view(-1), send, group=self.group, async_op=True)
        else:
            out[0].copy_(send)
        self._pending = (work, out, n, self._shapes, self._per_step)
        self._cur ^= 1
        self._fill = 0

    def start(self, tensors: Sequence[Tensor]) -> None:
        shapes = [tuple(int(d) for d in t.shape) for t in tensors]
        buf = self._send[self._cur]
        # a layout that slots() set up keeps its dtype (the kernels' fp32): outputs cast back to a caller's half /
        # double dtype are converted by the pack copy instead of re-allocating the send buffers every step
        dtype = buf.dtype if (buf is not None and self._shapes == shapes and buf.device == tensors[0].device) \
            else tensors[0].dtype
        self._ensure(shapes, dtype, tensors[0].device)
        for dst, t in zip(self._views(self._send[self._cur], self._fill), tensors):
            if t.data_ptr() != dst.data_ptr() or not t.is_contiguous():
                dst.copy_(t)  # not produced in place: pack
        self._fill += 1
        if self._fill == self.bucket:
            self._launch()

    def wait(self):
        self._collect()
        if not self._ready:
            return None
        if self.bucket == 1:
            return self._ready.pop(0)
        res, self._ready = self._ready, []
        return res

    def take_ready(self) -> List[List[Tensor]]:
        """Results of collectives that have already been collected (never blocks, never waits on a stream)."""
        res, self._ready = self._ready, []
        return res

    def flush(self) -> List[List[Tensor]]:
        if self._fill:
            self._launch()
        self._collect()
        res, self._ready = self._ready, []
        return res


def all_gather_dense(tensors: Sequence[Tensor], group=None) -> List[Tensor]:
    """Gather fixed-shape per-graph outputs ([B_local, K, F], [B_local, K, K], ...) from every rank and
    concatenate along the graph dimension.  Ranks may hold different B_local (padded to the max)."""
    world = _world(group)
    if world == 1:
        return list(tensors)
    dev = tensors[0].device
    b_local = torch.tensor([tensors[0].size(0)], dtype=torch.long, device=dev)
    counts = [torch.zeros_like(b_local) for _ in range(world)]
    dist.all_gather(counts, b_local, group=group)
    counts = [int(c) for c in counts]
    b_max = max(counts)
    out = []
    for t in tensors:
        pad = t
        if t.size(0) < b_max:
            pad = torch.cat([t, t.new_zeros((b_max - t.size(0),) + tuple(t.shape[1:]))])
        buf = torch.empty((world * b_max,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
        dist.all_gather_into_tensor(buf, pad.contiguous(), group=group)
        parts = [buf[r * b_max: r * b_max + counts[r]] for r in range(world)]
        out.append(torch.cat(parts) if any(c != b_max for c in counts) else buf)
    return out


_GP_MAGIC = 0x7467705F67617468
_GP_HEADER = 16  # int64 words


def _gp_align(v: int) -> int:
    return (v + 15) & ~15


def _gp_layout(K: int, E: int, Fw: int, w_words: int):
    """Byte offsets of a packed step (csrc/gather_pack.hip gp_layout): ``Fw`` = 4-byte words per feature row,
    ``w_words`` = words per edge weight (0 = no weights)."""
    x = _GP_HEADER * 8
    batch = _gp_align(x + K * Fw * 4)
    row = _gp_align(batch + K * 8)
    col = _gp_align(row + E * 8)
    w = _gp_align(col + E * 8)
    end = _gp_align(w + E * 4 * w_words)
    return x, batch, row, col, w, end


def _wire(t: Tensor) -> Tuple[Tensor, int]:
    """A value tensor in the dtype it travels in, and its 4-byte words per element.  4- and 8-byte element types
    (fp32, fp64, int32, int64) are copied bit for bit -- float64 features / weights used to be narrowed to fp32 on the
    way through the gather; smaller types widen losslessly (bf16 / half -> fp32, small ints -> int32) and
    are narrowed back on the receiving side."""
    if t.element_size() in (4, 8):
        return t, t.element_size() // 4
    return (t.to(torch.float32) if t.is_floating_point() else t.to(torch.int32)), 1


class SparseGather:
    """All-gather of VARIABLE-SIZE pooled sparse outputs (x [K,F], edge_index [2,E], edge_weight [E] or None, batch [K])
    as ONE payload collective per bucket of steps, asynchronous like :class:`PackedGather`.

    ``start`` packs the rank's outputs behind a 128-byte header {K, E, B, F, w words, needed bytes, x words} into the
    next slot of a byte buffer (one native launch on a device, ``tgp_gather_pack_f32``); ``bucket_steps`` slots go out
    as ONE ``all_gather_into_tensor`` on the collective's own stream.  There is no count exchange in front of it: the
    slot capacity is agreed FROM the headers -- every rank sees every header, so when some rank needs more room than
    the current capacity all ranks reach the same verdict, grow to the same size and repeat the steps concerned (a
    payload that does not fit travels as a header only).  On a device nothing of this blocks the host: behind the
    collective (a stream dependency) one native launch per bucket unpacks into capacity-sized outputs, shifting node ids
    / graph ids of rank r by the totals of the ranks before it (``tgp_gather_unpack_f32``; the merge rule of
    tgp/data/collate.py:144-153), and leaves the totals in a pinned host word that ``take_ready`` polls.  At most
    ``depth`` buckets are in flight, so a bucket's gather overlaps the next steps' kernels.  Every rank must call
    ``start`` (and ``flush``) the same number of times.

    Values keep their dtype: x and edge_weight travel as 4-byte words, float64 / int64 as two words per element (r5; r4
    narrowed everything to fp32), and every rank must pack the same feature width and value types -- the unpack launch
    checks the headers against what this rank packed and the host raises when they disagree.

    A result is ``(x, edge_index, edge_weight, batch)`` of ALL ranks' graphs, rank-major, as a single process would
    have produced them for the concatenated batch: new contiguous tensors of exactly the merged size, like the
    reference's collate.  ``views=True`` hands out views of the bucket's capacity buffer instead (``edge_index`` then
    has row stride e_cap; no copies -- what ``bench.py`` times).

    A gather is consumed EITHER by polling (``take_ready`` ... ``flush``) OR by blocking (``wait`` ... ``flush``), not
    both: a capacity change re-issues collectives, so it may only be decided at points every rank reaches in the same
    order.  ``take_ready`` never decides one; ``wait`` does, and which bucket it looks at depends on what earlier
    ``take_ready`` calls happened to find finished on THIS rank -- mixing the two lets ranks disagree about the next
    collective.  The first call fixes the mode; the other one then raises."""

    INITIAL_CAPACITY = 64 * 1024  # bytes per step; the SAME on every rank (a collective needs equal buffer sizes): it
                                  # only grows, and only by the rule below, which every rank applies to the same headers

    def __init__(self, group=None, force_collective: bool = False, depth: int = 2, capacity: Optional[int] = None,
                 bucket_steps: int = 1, views: bool = False):
        self.group = group
        self.world = _world(group)
        self._collective = self.world > 1 or (dist.is_available() and dist.is_initialized()
                                              and (force_collective or bool(os.environ.get("TGP_FORCE_COLLECTIVE"))))
        self.depth = max(int(depth), 1)
        self.bucket = min(max(int(bucket_steps), 1), 8)  # (tgp_gather_max_bucket_steps)
        # ``capacity``: a caller that knows its payloads may start larger (the same value on every rank!)
        self.capacity = max(int(capacity if capacity is not None else self.INITIAL_CAPACITY), _GP_HEADER * 8)
        self.views = bool(views)
        self._send: Optional[Tensor] = None
        self._open: List[tuple] = []       # inputs of the steps packed into the open bucket
        self._inflight: List[dict] = []    # buckets whose collective has been issued
        self._ready: List[tuple] = []
        self._pin = None                   # pinned result words (device path), 8 per step slot
        self._host = None
        self._tick = 0
        self._mode: Optional[str] = None   # "poll" (take_ready) or "block" (wait): see the class docstring

    # ---- packing --------------------------------------------------------------------------------------------------
    @staticmethod
    def _dims(x, edge_index):
        K, F = (x.size(0), x.size(1)) if x.dim() == 2 else (x.size(0), 1)
        return K, F, edge_index.size(1)

    @staticmethod
    def _wire_step(inputs):
        """(x2 [K,F] in its wire dtype with unit inner stride, xw, ei int64 rows, w wire / None, ww, batch int64 / None)"""
        x, edge_index, edge_weight, batch, _ = inputs
        K, F = (x.size(0), x.size(1)) if x.dim() == 2 else (x.size(0), 1)
        E = edge_index.size(1)
        x2, xw = _wire(x if x.dim() == 2 else x.reshape(K, F))
        if not (F <= 1 or x2.stride(1) == 1):
            x2 = x2.contiguous()
        ei = edge_index
        if not (ei.dtype == torch.int64 and (E <= 1 or ei.stride(1) == 1)):
            ei = ei.to(torch.int64).contiguous()
        w, ww = None, 0
        if edge_weight is not None:
            w, ww = _wire(edge_weight.reshape(-1))
            if not (E <= 1 or w.stride(0) == 1):
                w = w.contiguous()
        b = batch
        if b is not None and not (b.dtype == torch.int64 and b.dim() == 1 and (K <= 1 or b.stride(0) == 1)):
            b = b.to(torch.int64).contiguous()
        return x2, xw, ei, w, ww, b

    def _pack(self, inputs, dst: Tensor) -> None:
        """Host tensors (the gloo tests of the N > 1 logic): the layout of csrc/gather_pack.hip with torch ops."""
        num_graphs = inputs[4]
        x2, xw, ei, w, ww, b = self._wire_step(inputs)
        K, F, E = x2.size(0), x2.size(1), ei.size(1)
        cap = dst.numel()
        ox, ob, orow, ocol, ow, end = _gp_layout(K, E, F * xw, ww)
        head = torch.zeros(_GP_HEADER, dtype=torch.int64)
        head[:8] = torch.tensor([_GP_MAGIC, K, E, int(num_graphs), F * xw, ww, end, xw])
        dst[: _GP_HEADER * 8] = head.view(torch.uint8)
        if end <= cap:
            dst[ox: ox + K * F * xw * 4] = x2.reshape(-1).contiguous().view(torch.uint8)
            bb = torch.zeros(K, dtype=torch.int64) if b is None else b
            dst[ob: ob + K * 8] = bb.contiguous().view(torch.uint8)
            dst[orow: orow + E * 8] = ei[0].contiguous().view(torch.uint8)
            dst[ocol: ocol + E * 8] = ei[1].contiguous().view(torch.uint8)
            if w is not None:
                dst[ow: ow + E * ww * 4] = w.contiguous().view(torch.uint8)

    def _pack_bucket(self, steps, cap: int) -> Tensor:
        """Device path: every step of the bucket into its slot of a fresh send buffer, ONE launch."""
        import ctypes
        from . import _native as N
        n = len(steps)
        dev = steps[0][0].device
        send = torch.empty(n * cap, dtype=torch.uint8, device=dev)
        ptrs = (ctypes.c_void_p * (5 * n))()
        dims = (ctypes.c_int64 * (7 * n))()
        keep = []  # converted copies must outlive the launch (stream order: the allocator recycles them after it)
        for j, inputs in enumerate(steps):
            x2, xw, ei, w, ww, b = self._wire_step(inputs)
            K, F, E = x2.size(0), x2.size(1), ei.size(1)
            keep.append((x2, ei, w, b))
            o = 5 * j
            ptrs[o] = x2.data_ptr() if K * F else None
            ptrs[o + 1] = None if b is None else b.data_ptr()
            ptrs[o + 2] = ei.data_ptr() if E else None
            ptrs[o + 3] = ei.data_ptr() + 8 * ei.stride(0) if E else None
            ptrs[o + 4] = None if w is None else w.data_ptr()
            q = 7 * j
            dims[q], dims[q + 1], dims[q + 2], dims[q + 3] = (x2.stride(0) if K else F) * xw, K, E, inputs[4]
            dims[q + 4], dims[q + 5], dims[q + 6] = F * xw, ww, xw
        N.check(N.lib().tgp_gather_pack_bucket_f32(ptrs, dims, n, cap, send.data_ptr(), N.stream_ptr(dev)),
                "tgp_gather_pack_bucket_f32")
        return send

    @staticmethod
    def _grown(need: int) -> int:
        return ((need + need // 4 + 4095) // 4096) * 4096

    # ---- public ---------------------------------------------------------------------------------------------------
    def start(self, x: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor], batch: Optional[Tensor],
              num_graphs_local: int) -> None:
        inputs = (x, edge_index, edge_weight, batch, int(num_graphs_local))
        if x.is_cuda:
            # device path: the whole bucket is packed by ONE launch when it goes out (`_launch`); until then the tensors
            # handed over must not be written to (the pooled outputs of a step are fresh tensors)
            self._open.append(inputs)
        else:
            if self._send is None:
                self._send = torch.empty(self.bucket * self.capacity, dtype=torch.uint8, device=x.device)
            j = len(self._open)
            self._pack(inputs, self._send[j * self.capacity: (j + 1) * self.capacity])
            self._open.append(inputs)
        if len(self._open) == self.bucket:
            self._launch()

    def _launch(self, partial: bool = False) -> None:
        # The sequence of collectives must be the same on every rank, so a capacity change (which re-issues steps) may
        # only be decided at points every rank reaches at the same place of that sequence: here, before bucket
        # b + depth goes out (bucket b is finalised, blocking), and in wait / flush.  take_ready hands out finished
        # results early but never re-issues anything.
        while len(self._inflight) >= self.depth:
            self._finalise_oldest(block=True, allow_redo=True)
        if not self._open or (len(self._open) < self.bucket and not partial):
            return
        n, cap = len(self._open), self.capacity
        if self._open[0][0].is_cuda:
            send = self._pack_bucket(self._open, cap)
        else:
            send = self._send[: n * cap]
        dev = send.device
        if self._collective:
            gathered = torch.empty(self.world * n * cap, dtype=torch.uint8, device=dev)
            work = dist.all_gather_into_tensor(gathered, send, group=self.group, async_op=True)
        else:
            gathered, work = send, None
        bucket = dict(work=work, gathered=gathered, send=send, n=n, cap=cap, inputs=self._open, steps=None)
        self._send, self._open = None, []
        self._inflight.append(bucket)
        if dev.type == "cuda":
            self._enqueue_unpack(bucket)

    def _enqueue_unpack(self, bucket) -> None:
        """Device path: everything behind the collective is enqueued NOW (stream dependency, no host wait): one unpack
        launch per bucket into capacity-sized outputs, totals into pinned words."""
        from . import _native as N
        if bucket["work"] is not None:
            bucket["work"].wait()  # the current stream waits for the collective; the host does not
        n, cap, gathered = bucket["n"], bucket["cap"], bucket["gathered"]
        dev = gathered.device
        nslots = (self.depth + 2) * self.bucket
        if self._pin is None:
            self._pin = torch.zeros(nslots * 8, dtype=torch.int64).pin_memory()
            self._host = self._pin.numpy()
        L, st = N.lib(), N.stream_ptr(dev)
        steps = []
        # ONE allocation per bucket for all capacity outputs (the merged tensors are carved out of it when they are
        # handed out; at enqueue time only addresses are needed): per step just the unpack launch
        plan, total = [], 0
        for inputs in bucket["inputs"]:
            x, edge_index, edge_weight, batch, _ = inputs
            K, F, E = self._dims(x, edge_index)
            xw = x.element_size() // 4 if x.element_size() in (4, 8) else 1
            ww = 0 if edge_weight is None else (edge_weight.element_size() // 4
                                               if edge_weight.element_size() in (4, 8) else 1)
            Fw = F * xw
            k_cap = self.world * (cap // (4 * Fw + 8) + 1)
            e_cap = self.world * (cap // (16 + 4 * ww) + 1)
            ox = total
            ob = _gp_align(ox + k_cap * Fw * 4)
            oe = _gp_align(ob + k_cap * 8)
            ow = _gp_align(oe + 2 * e_cap * 8)
            total = _gp_align(ow + e_cap * 4 * ww)
            plan.append((F, xw, ww, k_cap, e_cap, ox, ob, oe, ow))
        import ctypes
        out8 = torch.empty(total, dtype=torch.uint8, device=dev)
        # two typed views of the bucket's output buffer, made once: every merged tensor is then ONE as_strided away
        out = (out8.view(torch.float32), out8.view(torch.int64), out8)
        base, gbase = out8.data_ptr(), gathered.data_ptr()
        max_words = cap // 4
        ptrs = (ctypes.c_void_p * (6 * n))()
        dims = (ctypes.c_int64 * (6 * n))()
        pin = self._pin.data_ptr()
        for j, (F, xw, ww, k_cap, e_cap, ox, ob, oe, ow) in enumerate(plan):
            self._tick += 1
            slot, tag = self._tick % nslots, self._tick
            o = 6 * j
            ptrs[o], ptrs[o + 1], ptrs[o + 2], ptrs[o + 3] = base + ox, base + ob, base + oe, base + oe + 8 * e_cap
            ptrs[o + 4] = base + ow if ww else None
            ptrs[o + 5] = pin + slot * 64
            dims[o], dims[o + 1], dims[o + 2], dims[o + 3], dims[o + 4], dims[o + 5] = k_cap, e_cap, tag, F * xw, ww, xw
            steps.append((slot, tag, out, plan[j]))
        # ONE launch for the whole bucket (grid z = step)
        N.check(L.tgp_gather_unpack_bucket_f32(gbase, cap, n * cap, self.world, max_words, n, ptrs, dims, st),
                "tgp_gather_unpack_bucket_f32")
        bucket["steps"] = steps

    def _poll(self, slot: int, tag: int, block: bool) -> bool:
        host, spins = self._host, 0
        while int(host[slot * 8]) != tag:
            if not block:
                return False
            spins += 1
            if spins > 4_000_000:
                torch.cuda.synchronize()
                if int(host[slot * 8]) != tag:
                    raise RuntimeError("SparseGather: the unpack launch never reported (device fault?)")
        return True

    def _redo_with(self, need: int, first_bucket: dict) -> None:
        """A payload did not fit: every rank sees the same headers, grows alike and repeats, in order, the steps of this
        bucket, of the buckets issued behind it, and of the open one."""
        redo = list(first_bucket["inputs"])
        for bk in self._inflight:
            if bk["work"] is not None:
                bk["work"].wait()
            redo.extend(bk["inputs"])
        redo.extend(self._open)
        if first_bucket["gathered"].is_cuda:
            torch.cuda.synchronize(first_bucket["gathered"].device)  # nothing of the old layout may still be running
        self._inflight, self._open, self._send = [], [], None
        self.capacity = max(self.capacity, self._grown(need))
        for inp in redo:
            self.start(*inp)

    @staticmethod
    def _restore(t: Tensor, like: Tensor) -> Tensor:
        """A merged value tensor back in the dtype its input had (bf16 / half / small ints travelled widened)."""
        return t if t.dtype == like.dtype else t.to(like.dtype)

    def _hand_out(self, x_in, w_in, b_in, xo, eo, wo, bo) -> None:
        if x_in.dim() == 1:
            xo = xo.reshape(-1)
        xo = self._restore(xo, x_in)
        if wo is not None:
            wo = self._restore(wo, w_in)
        if not self.views:  # the reference's collate hands out fresh tensors of the merged size: so does the default
            xo, eo, wo, bo = self._own(xo, eo, wo, bo)
        self._ready.append((xo, eo, wo, bo))

    @staticmethod
    def _own(*tensors):
        """Exact-size copies of the slices of a receive buffer: device tensors in ONE launch (tgp_copy_arrays; a tensor
        the dtype restore already made fresh is kept), host tensors with clone()."""
        out = list(tensors)
        todo = [i for i, t in enumerate(out)
                if t is not None and not (t.is_contiguous() and t.untyped_storage().nbytes() == t.numel() * t.element_size())]
        if not todo:
            return out
        first = out[todo[0]]
        if not first.is_cuda or any((out[i].numel() * out[i].element_size()) & 3 for i in todo):
            for i in todo:
                out[i] = out[i].clone(memory_format=torch.contiguous_format)
            return out
        from . import _native as N
        src, dst, nbytes = _COPY_SRC, _COPY_DST, _COPY_BYTES  # (reused ctypes arrays: the call reads them at once)
        n = 0
        for i in todo:
            t = out[i]
            es = t.element_size()
            new = torch.empty(t.shape, dtype=t.dtype, device=t.device)
            if t.dim() == 2 and not t.is_contiguous():  # edge_index: a [2, E'] view whose rows are contiguous
                rows, cols = t.shape
                sp, dp, step = t.data_ptr(), new.data_ptr(), t.stride(0) * es
                for r in range(rows):
                    src[n], dst[n], nbytes[n] = sp + r * step, dp + r * cols * es, cols * es
                    n += 1
            else:
                src[n], dst[n], nbytes[n] = t.data_ptr(), new.data_ptr(), t.numel() * es
                n += 1
            out[i] = new
        N.check(N.lib().tgp_copy_arrays(src, dst, nbytes, n, N.stream_ptr(first.device)), "tgp_copy_arrays")
        return out

    def _finalise_oldest(self, block: bool, allow_redo: bool = False) -> bool:
        if not self._inflight:
            return False
        bucket = self._inflight[0]
        if bucket["steps"] is not None:  # device path: totals arrive in the pinned words
            for (slot, tag, *_rest) in bucket["steps"]:
                if not self._poll(slot, tag, block):
                    return False
            results, need, status = [], 0, 7
            for inputs, (slot, tag, out, plan) in zip(bucket["inputs"], bucket["steps"]):
                h = self._host[slot * 8: slot * 8 + 8]
                kt, et, nd, st = int(h[1]), int(h[2]), int(h[3]), int(h[4])
                need = max(need, nd)
                status &= st | (0 if nd <= bucket["cap"] else 4)  # (room is only meaningful for a payload that fitted)
                results.append((inputs, kt, et, out, plan))
            if not status & 1:
                raise RuntimeError("SparseGather: a gathered buffer does not start with a pack header")
            if not status & 2:
                raise RuntimeError("SparseGather: the ranks packed different feature widths / value dtypes / edge-weight "
                                   "presence for the same step -- every rank must gather tensors of the same layout")
            if need > bucket["cap"]:
                if not allow_redo:
                    return False
                self._inflight.pop(0)
                self._redo_with(need, bucket)
                return self._finalise_oldest(block, allow_redo)
            if not status & 4:
                raise RuntimeError("SparseGather: merged outputs exceed the capacity buffers (internal sizing error)")
            self._inflight.pop(0)
            for inputs, kt, et, out, (F, xw, ww, k_cap, e_cap, ox, ob, oe, ow) in results:
                x_in, _, w_in, b_in, _ = inputs
                f32, i64, raw = out
                x_wire = x_in.dtype if x_in.element_size() in (4, 8) else (
                    torch.float32 if x_in.is_floating_point() else torch.int32)
                if x_wire == torch.float32:
                    xo = torch.as_strided(f32, (kt, F), (F, 1), ox >> 2)
                else:
                    xo = raw[ox: ox + kt * F * xw * 4].view(x_wire).view(kt, F)
                eo = torch.as_strided(i64, (2, et), (e_cap, 1), oe >> 3)
                wo = None
                if ww:
                    w_wire = w_in.dtype if w_in.element_size() in (4, 8) else (
                        torch.float32 if w_in.is_floating_point() else torch.int32)
                    if w_wire == torch.float32:
                        wo = torch.as_strided(f32, (et,), (1,), ow >> 2)
                    else:
                        wo = raw[ow: ow + et * ww * 4].view(w_wire)
                bo = torch.as_strided(i64, (kt,), (1,), ob >> 3) if b_in is not None else None
                self._hand_out(x_in, w_in, b_in, xo, eo, wo, bo)
            return True
        # host tensors (gloo): synchronous, with torch ops
        if bucket["work"] is not None:
            bucket["work"].wait()
        n, cap = bucket["n"], bucket["cap"]
        g3 = bucket["gathered"].view(self.world if self._collective else 1, n, cap)
        world = g3.size(0)
        heads = g3[:, :, : _GP_HEADER * 8].contiguous().view(torch.int64).view(world, n, _GP_HEADER)
        if not bool((heads[..., 0] == _GP_MAGIC).all()):
            raise RuntimeError("SparseGather: a gathered buffer does not start with a pack header")
        need = int(heads[..., 6].max())
        for j, inputs in enumerate(bucket["inputs"]):  # the same agreement check the unpack kernel makes
            x2, xw, _, w, ww, _ = self._wire_step(inputs)
            mine = torch.tensor([x2.size(1) * xw, ww, xw])
            if not bool((heads[:, j][:, [4, 5, 7]] == mine).all()):
                raise RuntimeError("SparseGather: the ranks packed different feature widths / value dtypes / edge-weight "
                                   "presence for the same step -- every rank must gather tensors of the same layout")
        if need > cap:
            if not allow_redo:
                return False
            self._inflight.pop(0)
            self._redo_with(need, bucket)
            return self._finalise_oldest(block, allow_redo)
        self._inflight.pop(0)
        for j, inputs in enumerate(bucket["inputs"]):
            x_in, _, w_in, b_in, _ = inputs
            x2, xw, _, w2, ww, _ = self._wire_step(inputs)
            hj = heads[:, j]
            F = x2.size(1)
            Fw = F * xw
            Kt, Et = int(hj[:, 1].sum()), int(hj[:, 2].sum())
            x_out = torch.empty(Kt * Fw * 4, dtype=torch.uint8)
            b_out = torch.empty(Kt, dtype=torch.int64)
            ei_out = torch.empty(2, Et, dtype=torch.int64)
            w_out = torch.empty(Et * ww * 4, dtype=torch.uint8) if ww else None
            koff = eoff = goff = 0
            for r in range(world):
                K, E, B = int(hj[r, 1]), int(hj[r, 2]), int(hj[r, 3])
                ox, ob, orow, ocol, ow, _ = _gp_layout(K, E, Fw, ww)
                buf = g3[r, j]
                x_out[koff * Fw * 4: (koff + K) * Fw * 4] = buf[ox: ox + K * Fw * 4]
                b_out[koff: koff + K] = buf[ob: ob + K * 8].contiguous().view(torch.int64) + goff
                ei_out[0, eoff: eoff + E] = buf[orow: orow + E * 8].contiguous().view(torch.int64) + koff
                ei_out[1, eoff: eoff + E] = buf[ocol: ocol + E * 8].contiguous().view(torch.int64) + koff
                if ww:
                    w_out[eoff * ww * 4: (eoff + E) * ww * 4] = buf[ow: ow + E * ww * 4]
                koff, eoff, goff = koff + K, eoff + E, goff + B
            xo = x_out.view(x2.dtype).view(Kt, F)
            wo = w_out.view(w2.dtype) if ww else None
            self._hand_out(x_in, w_in, b_in, xo, ei_out, wo, b_out if b_in is not None else None)
        return True

    def _enter(self, mode: str) -> None:
        if self._mode is None:
            self._mode = mode
        elif self._mode != mode:
            raise RuntimeError("SparseGather: take_ready() and wait() must not be mixed on one gather (a capacity change "
                               "re-issues collectives and may only be decided where every rank decides it: see the class "
                               "docstring); use take_ready ... flush OR wait ... flush")

    def take_ready(self) -> List[tuple]:
        """Results of buckets whose unpack has already reported (never waits for a collective still in flight)."""
        self._enter("poll")
        while self._inflight:
            bk = self._inflight[0]
            if bk["steps"] is None and bk["work"] is not None and not bk["work"].is_completed():
                break
            if not self._finalise_oldest(block=False, allow_redo=False):
                break
        res, self._ready = self._ready, []
        return res

    def wait(self):
        """The oldest result not yet handed out (sends a partly filled bucket and waits if it has to), or None.  Like
        ``start`` / ``flush``, every rank must call it at the same place."""
        self._enter("block")
        if not self._ready:
            if not self._inflight:
                self._launch(partial=True)
            self._finalise_oldest(block=True, allow_redo=True)
        return self._ready.pop(0) if self._ready else None

    def flush(self) -> List[tuple]:
        while self._inflight:  # (first what is in flight: a capacity change re-issues the open steps with it)
            self._finalise_oldest(block=True, allow_redo=True)
        self._launch(partial=True)
        while self._inflight:
            self._finalise_oldest(block=True, allow_redo=True)
        res, self._ready = self._ready, []
        return res


_SYNC_GATHERS: dict = {}


def all_gather_sparse(x: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor], batch: Tensor,
                      num_graphs_local: int, group=None, force_collective: bool = False):
    """Gather variable-size pooled graphs from every rank, synchronously: one :class:`SparseGather` step (one payload
    collective; node ids / graph ids of rank r shifted by the totals of ranks < r).  ``force_collective`` (or
    ``TGP_FORCE_COLLECTIVE``): a one-rank process group still runs the collective.  Returns new contiguous tensors in
    the dtypes of the inputs.  The gather object -- and with it the slot capacity the ranks have agreed on so far -- is
    kept per process group: a payload beyond the initial 64 KiB pays the header-only round and the regrowth once, not on
    every call (growth is decided from the same headers on every rank, so the kept value is the same everywhere)."""
    world = _world(group)
    forced = (dist.is_available() and dist.is_initialized()
              and (force_collective or bool(os.environ.get("TGP_FORCE_COLLECTIVE"))))
    if world == 1 and not forced:
        return x, edge_index, edge_weight, batch
    key = (id(group) if group is not None else 0, bool(forced), world)
    g = _SYNC_GATHERS.get(key)
    if g is None or g.group is not group or g._inflight or g._open or g._ready:
        # A gather that an exception left with open steps is replaced -- with the capacity it had reached: capacity only
        # ever grows, by a rule every rank applies to the same headers, so carrying it over keeps the ranks' buffer sizes
        # equal even when only ONE rank went through the exception.  Entries are never dropped by a per-rank
        # count either (ranks in different numbers of groups would then disagree): a dead group's entry is replaced when
        # its id is reused (`g.group is not group`).
        keep = g.capacity if (g is not None and g.group is group) else None
        g = _SYNC_GATHERS[key] = SparseGather(group=group, force_collective=force_collective, depth=1, bucket_steps=1,
                                              capacity=keep)
    g.start(x, edge_index, edge_weight, batch, num_graphs_local)
    return g.wait()
