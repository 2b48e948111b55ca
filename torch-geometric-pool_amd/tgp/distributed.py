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


def all_gather_sparse(x: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor], batch: Tensor,
                      num_graphs_local: int, group=None, force_collective: bool = False):
    """Gather variable-size pooled graphs: counts first, then max-padded payloads, then shift the
    pooled node ids / graph ids of rank r by the totals of ranks < r.  ``force_collective`` (or
    ``TGP_FORCE_COLLECTIVE``): a one-rank process group still runs the collectives (single-GPU exercise of the
    variable-size RCCL path)."""
    world = _world(group)
    forced = (dist.is_available() and dist.is_initialized()
              and (force_collective or bool(os.environ.get("TGP_FORCE_COLLECTIVE"))))
    if world == 1 and not forced:
        return x, edge_index, edge_weight, batch
    dev = x.device
    mine = torch.tensor([x.size(0), edge_index.size(1), num_graphs_local], dtype=torch.long, device=dev)
    allc = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allc, mine, group=group)
    allc = torch.stack(allc).cpu()
    k_max, e_max = int(allc[:, 0].max()), int(allc[:, 1].max())

    def gather_rows(t: Tensor, n_max: int) -> Tensor:
        pad = t.new_zeros((n_max,) + tuple(t.shape[1:]))
        pad[: t.size(0)] = t
        buf = torch.empty((world * n_max,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
        dist.all_gather_into_tensor(buf, pad, group=group)
        return buf.view((world, n_max) + tuple(t.shape[1:]))

    gx = gather_rows(x, k_max)
    gb = gather_rows(batch, k_max)
    ge = gather_rows(edge_index.t().contiguous(), e_max)
    gw = None if edge_weight is None else gather_rows(edge_weight, e_max)
    xs, bs, es, ws = [], [], [], []
    node_off = graph_off = 0
    for r in range(world):
        k, e, g = (int(v) for v in allc[r])
        xs.append(gx[r, :k])
        bs.append(gb[r, :k] + graph_off)
        es.append(ge[r, :e] + node_off)
        if gw is not None:
            ws.append(gw[r, :e])
        node_off += k
        graph_off += g
    return (torch.cat(xs), torch.cat(es).t().contiguous(), None if gw is None else torch.cat(ws), torch.cat(bs))
