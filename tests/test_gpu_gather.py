"""Gathering pooled outputs across ranks (tgp/distributed.py: PackedGather, SparseGather, all_gather_sparse; csrc/gather_pack.hip) on one GPU: one-rank RCCL groups and simulated ranks.

Regrouped by operator in round 6 from the per-round files test_gpu_round2..5.py; the test bodies are unchanged."""
import pytest
import torch
import os
import socket
import sys
import warnings

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture()
def one_rank_rccl(dev):
    """A one-rank RCCL process group (created here unless the process already has one)."""
    import torch.distributed as dist
    created = False
    if not dist.is_initialized():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        created = True
    yield dist
    if created:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------ one-launch sparse pooling of small graphs
def _small_batch(num_graphs, lo, hi, f, seed, dev, deg=4, dup=False):
    """PyG-style batch: sorted batch vector, row-major sorted undirected edge list (optionally with duplicate entries)."""
    g = torch.Generator().manual_seed(seed)
    sizes = torch.randint(lo, hi + 1, (num_graphs,), generator=g)
    n = int(sizes.sum())
    batch = torch.repeat_interleave(torch.arange(num_graphs), sizes)
    start = torch.cumsum(sizes, 0) - sizes
    src = torch.arange(n).repeat_interleave(max(deg // 2, 1))
    dst = start[batch[src]] + (torch.rand(src.numel(), generator=g) * sizes[batch[src]]).long()
    keep = src != dst
    src, dst = src[keep], dst[keep]
    key = torch.cat([src * n + dst, dst * n + src])
    key = torch.sort(key)[0] if dup else torch.unique(key)
    ei = torch.stack([key // n, key % n])
    x = torch.randn(n, f, generator=g)
    ew = torch.rand(ei.size(1), generator=g) + 0.25
    ew[torch.rand(ei.size(1), generator=g) < 0.05] = 0.0  # some weights the eps filter drops
    return x.to(dev), ei.to(dev), ew.to(dev), batch.to(dev), sizes


def test_all_gather_sparse_over_rccl_one_rank_group(dev):
    """The variable-size gather of pooled sparse outputs (merge rule tgp/data/collate.py:144-153) through
    REAL RCCL collectives on the device: a one-rank process group with ``force_collective`` runs the count exchange, the
    padded payload gathers and the offset merge; the merged result must equal the local one bit for bit."""
    import os
    import socket
    import torch.distributed as dist
    from tgp.connect import SparseConnect
    from tgp.distributed import all_gather_sparse
    from tgp.reduce import BaseReduce
    from tgp.select import TopkSelect
    created = False
    if not dist.is_initialized():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        created = True
    try:
        g = torch.Generator().manual_seed(0)
        sizes = torch.randint(20, 61, (300,), generator=g)
        n = int(sizes.sum())
        batch = torch.repeat_interleave(torch.arange(300), sizes).to(dev)
        start = (torch.cumsum(sizes, 0) - sizes).to(dev)
        src = torch.arange(n, device=dev).repeat_interleave(2)
        dst = start[batch[src]] + (torch.rand(src.numel(), device=dev) * sizes.to(dev)[batch[src]]).long()
        key = torch.unique(torch.cat([src * n + dst, dst * n + src]))
        ei = torch.stack([key // n, key % n])
        x = torch.randn(n, 16, device=dev)
        ew = torch.rand(ei.size(1), device=dev) + 0.5
        with torch.no_grad():
            so = TopkSelect(in_channels=16, ratio=0.5).to(dev)(x=x, batch=batch)
            xp, bp = BaseReduce()(x, so, batch=batch)
            pe, pw = SparseConnect()(ei, so, edge_weight=ew, batch_pooled=bp)
        mx, me, mw, mb = all_gather_sparse(xp, pe, pw, bp, 300, force_collective=True)
        assert mx.data_ptr() != xp.data_ptr()  # went through the gather buffers, not the early return
        assert torch.equal(mx, xp) and torch.equal(me, pe) and torch.equal(mw, pw) and torch.equal(mb, bp)
    finally:
        if created:
            dist.destroy_process_group()


def test_packed_gather_in_place_over_rccl_one_rank_group(dev, one_rank_rccl):
    """SURVEY 8(e), dense outputs: ``PackedGather.slots()`` hands the pooling call slices of the all-gather send buffer
    (``reduce_connect(out_x=, out_adj=)``: the kernels write INTO it, no pack copy) and the bucket goes out as one REAL
    RCCL collective on a one-rank group.  Gathered == a plain local call, bit for bit, for every step of two buckets
    (one full, one flushed partly filled), and the outputs really are the send buffer's memory."""
    from tgp.connect import DenseConnect
    from tgp.distributed import PackedGather
    from tgp.reduce import BaseReduce
    from tgp.select import SelectOutput
    from tgp.src import DenseSRCPooling
    g = torch.Generator(device=dev).manual_seed(3)
    B, N, K, F = 6, 256, 32, 24
    pool = DenseSRCPooling(reducer=BaseReduce(), connector=DenseConnect(), adj_transpose=True)
    pg = PackedGather(bucket_steps=2, force_collective=True)
    want, got = [], []
    for step in range(3):
        A = (torch.rand(B, N, N, device=dev, generator=g) < 0.05).float()
        A = torch.maximum(A, A.transpose(1, 2)).contiguous()
        X = torch.randn(B, N, F, device=dev, generator=g)
        S = torch.softmax(torch.randn(B, N, K, device=dev, generator=g), -1)
        so = SelectOutput(s=S)
        with torch.no_grad():
            x_ref, _, a_ref = pool.reduce_connect(X, A, so)
            ox, oa = pg.slots([(B, K, F), (B, K, K)], device=dev)
            x_pool, _, adj_pool = pool.reduce_connect(X, A, so, out_x=ox, out_adj=oa)
        assert x_pool.data_ptr() == ox.data_ptr() and adj_pool.data_ptr() == oa.data_ptr()
        want.append((x_ref.clone(), a_ref.clone()))
        pg.start([x_pool, adj_pool])
        got.extend(pg.take_ready())
    got.extend(pg.flush())
    assert len(got) == 3
    for (xw, aw), (xg, ag) in zip(want, got):
        assert xg.data_ptr() != xw.data_ptr()
        assert torch.equal(xg, xw) and torch.equal(ag, aw)


def test_sparse_gather_async_buckets_over_rccl_one_rank_group(dev, one_rank_rccl):
    """SURVEY 8(e), sparse outputs: SparseGather on a one-rank RCCL group with REAL collectives -- five steps, two per
    payload collective, two buckets in flight, a first capacity that is too small (grown from the headers, the steps
    repeated), totals through pinned host words.  Every merged result equals the local one bit for bit (one rank: no
    offsets), in step order; the pooled edge_index of the one-launch kernel (a strided view) travels unchanged."""
    from tgp.distributed import SparseGather
    from tgp.poolers import get_pooler
    x, ei, ew, batch, sizes = _small_batch(300, 5, 60, 16, 21, dev)
    pooler = get_pooler("topk", in_channels=16, ratio=0.5).to(dev).eval()
    import tgp
    with torch.no_grad(), tgp.output_views():
        out = pooler(x=x, adj=ei, edge_weight=ew, batch=batch)
    assert not out.edge_index.is_contiguous()  # opt-in: the capacity-buffer view of tgp_sparse_pool_small_f32
    sg = SparseGather(force_collective=True, depth=2, bucket_steps=2, capacity=1024)
    got = []
    for j in range(5):
        sg.start(out.x * (j + 1), out.edge_index, out.edge_weight * (j + 1), out.batch, 300)
        got.extend(sg.take_ready())
    got.extend(sg.flush())
    assert len(got) == 5 and sg.capacity > 1024
    for j, (gx, gei, gw, gb) in enumerate(got):
        assert torch.equal(gx, out.x * (j + 1)) and torch.equal(gei, out.edge_index)
        assert torch.equal(gw, out.edge_weight * (j + 1)) and torch.equal(gb, out.batch)
    # unweighted, no batch vector
    sg2 = SparseGather(force_collective=True)
    sg2.start(out.x, out.edge_index, None, None, 300)
    gx, gei, gw, gb = sg2.wait()
    assert gw is None and gb is None and torch.equal(gx, out.x) and torch.equal(gei, out.edge_index)


def test_gather_unpack_kernels_with_three_simulated_ranks(dev):
    """the device unpack of the variable-size gather for world > 1, which no 1-GPU box can reach through RCCL:
    the gathered buffer of THREE ranks is laid out by hand (every rank's pooled graphs packed into its slot by
    tgp_gather_pack_f32), then tgp_gather_unpack_f32 (one step) and tgp_gather_unpack_bucket_f32 (two steps per bucket,
    rank stride = 2 slots) must give the merge rule of tgp/data/collate.py:144-153 -- node ids of rank r shifted by the
    supernodes, graph ids by the graphs of the ranks before it -- bit for bit."""
    import ctypes
    from tgp import _native as N
    L = N.lib()
    st = N.stream_ptr(dev)
    world, nsteps, F = 3, 2, 5
    g = torch.Generator().manual_seed(123)

    def part(seed_k):
        K = int(torch.randint(3, 40, (1,), generator=g))
        E = int(torch.randint(0, 90, (1,), generator=g))
        B = int(torch.randint(1, 5, (1,), generator=g))
        x = torch.randn(K, F, generator=g)
        ei = torch.randint(0, K, (2, E), generator=g)
        ew = torch.rand(E, generator=g)
        b = torch.sort(torch.randint(0, B, (K,), generator=g))[0]
        return x.to(dev), ei.to(dev), ew.to(dev), b.to(dev), B

    parts = [[part(0) for _ in range(world)] for _ in range(nsteps)]          # parts[step][rank]
    need = max(int(L.tgp_gather_pack_bytes(p[0].size(0), p[1].size(1), F, 1)) for s in parts for p in s)
    cap = ((need + 4095) // 4096) * 4096
    gathered = torch.zeros(world * nsteps * cap, dtype=torch.uint8, device=dev)   # [rank][step][cap]
    for j in range(nsteps):
        for r in range(world):
            x, ei, ew, b, B = parts[j][r]
            dst = gathered[(r * nsteps + j) * cap:]
            N.check(L.tgp_gather_pack_f32(x.data_ptr(), x.stride(0), b.data_ptr(), ei[0].contiguous().data_ptr() if ei.size(1) else None,
                                          ei[1].contiguous().data_ptr() if ei.size(1) else None, ew.data_ptr() if ei.size(1) else None,
                                          x.size(0), ei.size(1), B, F, 1, 1, cap, dst.data_ptr(), st), "pack")

    def expected(j):
        xs, eis, ews, bs, koff, goff = [], [], [], [], 0, 0
        for r in range(world):
            x, ei, ew, b, B = parts[j][r]
            xs.append(x); eis.append(ei + koff); ews.append(ew); bs.append(b + goff)
            koff += x.size(0); goff += B
        return torch.cat(xs), torch.cat(eis, 1), torch.cat(ews), torch.cat(bs)

    k_cap = sum(max(p[0].size(0) for p in s) for s in parts) * world + 8
    e_cap = sum(max(p[1].size(1) for p in s) for s in parts) * world + 8
    pin = torch.zeros(8 * nsteps, dtype=torch.int64).pin_memory()
    host = pin.numpy()
    # (a) one step per launch: the step's slots sit nsteps * cap apart
    for j in range(nsteps):
        xo = torch.empty(k_cap, F, device=dev); bo = torch.empty(k_cap, dtype=torch.int64, device=dev)
        eo = torch.empty(2, e_cap, dtype=torch.int64, device=dev); wo = torch.empty(e_cap, device=dev)
        N.check(L.tgp_gather_unpack_f32(gathered.data_ptr() + j * cap, cap, nsteps * cap, world, cap // 4, k_cap, e_cap,
                                        F, 1, 1, xo.data_ptr(), bo.data_ptr(), eo[0].data_ptr(), eo[1].data_ptr(),
                                        wo.data_ptr(), pin.data_ptr() + 64 * j, 1000 + j, st), "unpack")
        torch.cuda.synchronize()
        assert int(host[8 * j]) == 1000 + j and int(host[8 * j + 4]) == 7  # headers valid | layouts agree | fits
        kt, et = int(host[8 * j + 1]), int(host[8 * j + 2])
        ex, eei, eew, eb = expected(j)
        assert kt == ex.size(0) and et == eei.size(1)
        assert torch.equal(xo[:kt], ex) and torch.equal(bo[:kt], eb)
        assert torch.equal(eo[:, :et], eei) and torch.equal(wo[:et], eew)
    # (b) the whole bucket in one launch
    outs, ptrs, dims = [], (ctypes.c_void_p * (6 * nsteps))(), (ctypes.c_int64 * (6 * nsteps))()
    pin.zero_()
    for j in range(nsteps):
        xo = torch.empty(k_cap, F, device=dev); bo = torch.empty(k_cap, dtype=torch.int64, device=dev)
        eo = torch.empty(2, e_cap, dtype=torch.int64, device=dev); wo = torch.empty(e_cap, device=dev)
        outs.append((xo, bo, eo, wo))
        o = 6 * j
        ptrs[o], ptrs[o + 1], ptrs[o + 2], ptrs[o + 3] = xo.data_ptr(), bo.data_ptr(), eo[0].data_ptr(), eo[1].data_ptr()
        ptrs[o + 4], ptrs[o + 5] = wo.data_ptr(), pin.data_ptr() + 64 * j
        dims[6 * j], dims[6 * j + 1], dims[6 * j + 2] = k_cap, e_cap, 2000 + j
        dims[6 * j + 3], dims[6 * j + 4], dims[6 * j + 5] = F, 1, 1
    N.check(L.tgp_gather_unpack_bucket_f32(gathered.data_ptr(), cap, nsteps * cap, world, cap // 4, nsteps, ptrs, dims, st),
            "unpack_bucket")
    torch.cuda.synchronize()
    for j in range(nsteps):
        assert int(host[8 * j]) == 2000 + j and int(host[8 * j + 4]) == 7
        kt, et = int(host[8 * j + 1]), int(host[8 * j + 2])
        ex, eei, eew, eb = expected(j)
        xo, bo, eo, wo = outs[j]
        assert kt == ex.size(0) and et == eei.size(1)
        assert torch.equal(xo[:kt], ex) and torch.equal(bo[:kt], eb)
        assert torch.equal(eo[:, :et], eei) and torch.equal(wo[:et], eew)
    # (c) r5: a caller that expects another layout than the ranks packed (here: fp64 weights) is told so in the status
    # word (bit 2 clear) and nothing is written
    pin.zero_()
    xo = torch.full((k_cap, F), -7.0, device=dev)
    N.check(L.tgp_gather_unpack_f32(gathered.data_ptr(), cap, nsteps * cap, world, cap // 4, k_cap, e_cap, F, 2, 1,
                                    xo.data_ptr(), None, eo[0].data_ptr(), eo[1].data_ptr(), None, pin.data_ptr(), 3000,
                                    st), "unpack")
    torch.cuda.synchronize()
    assert int(host[0]) == 3000 and int(host[4]) & 1 and not int(host[4]) & 2 and bool((xo == -7.0).all())


def test_sparse_gather_device_path_with_a_simulated_second_rank(dev, monkeypatch):
    """SparseGather's DEVICE path for world = 2 on one GPU: the collective is replaced by a stand-in that delivers the
    local bucket twice (as if the second rank held the same graphs), everything else -- bucket packing, capacity growth from
    the headers, slot arithmetic, the unpack launch per bucket, the offsets of rank 1 -- is the code the 8-GPU run takes.
    Expected merge (tgp/data/collate.py:144-153): the local result followed by a copy shifted by K supernodes / G graphs."""
    import torch.distributed as dist
    from tgp import distributed as D
    from tgp.poolers import get_pooler

    class _Done:
        def wait(self):
            return True

    def fake_all_gather(out, inp, group=None, async_op=False):
        n = inp.numel()
        out[:n].copy_(inp)
        out[n: 2 * n].copy_(inp)
        return _Done()
    monkeypatch.setattr(dist, "all_gather_into_tensor", fake_all_gather)
    x, ei, ew, batch, sizes = _small_batch(120, 5, 40, 8, 55, dev)
    pooler = get_pooler("topk", in_channels=8, ratio=0.5).to(dev).eval()
    G = int(batch.max()) + 1
    sg = D.SparseGather(depth=2, bucket_steps=3, capacity=4096)   # too small at first: grown from the headers
    sg.world, sg._collective = 2, True
    steps = []
    with torch.no_grad():
        for s in range(7):                                         # 2 full buckets + a partial one
            out = pooler(x=x * (s + 1), adj=ei, edge_weight=ew, batch=batch)
            steps.append(out)
            sg.start(out.x, out.edge_index, out.edge_weight, out.batch, G)
            sg.take_ready()
    got = []
    sg2 = D.SparseGather(depth=2, bucket_steps=3, capacity=4096)
    sg2.world, sg2._collective = 2, True
    with torch.no_grad():
        for out in steps:
            sg2.start(out.x, out.edge_index, out.edge_weight, out.batch, G)
            got.extend(sg2.take_ready())
    got.extend(sg2.flush())
    assert len(got) == len(steps)
    for out, (gx, gei, gew, gb) in zip(steps, got):
        K = out.x.size(0)
        assert torch.equal(gx, torch.cat([out.x, out.x]))
        assert torch.equal(gei, torch.cat([out.edge_index, out.edge_index + K], 1))
        assert torch.equal(gew, torch.cat([out.edge_weight, out.edge_weight]))
        assert torch.equal(gb, torch.cat([out.batch, out.batch + G]))


def test_sparse_gather_device_path_keeps_float64_values(dev, monkeypatch):
    """SparseGather sent x and edge_weight as float32.  Device path, world = 2 simulated (the stand-in
    collective delivers the local bucket twice): float64 features / weights come back float64 and bit-identical, integer
    features as integers, and the default results are contiguous exact-size tensors (views=True: views of the bucket)."""
    import torch.distributed as dist
    from tgp import distributed as D

    class _Done:
        def wait(self):
            return True

    def fake_all_gather(out, inp, group=None, async_op=False):
        n = inp.numel()
        out[:n].copy_(inp)
        out[n: 2 * n].copy_(inp)
        return _Done()
    monkeypatch.setattr(dist, "all_gather_into_tensor", fake_all_gather)
    g = torch.Generator().manual_seed(4)
    K, F, E, B = 37, 5, 90, 4
    x = (torch.randn(K, F, generator=g, dtype=torch.float64) * (1 + 2.0 ** -40)).to(dev)
    ei = torch.randint(0, K, (2, E), generator=g).to(dev)
    w = (torch.rand(E, generator=g, dtype=torch.float64) + 2.0 ** -45).to(dev)
    b = torch.sort(torch.randint(0, B, (K,), generator=g))[0].to(dev)
    for views in (False, True):
        sg = D.SparseGather(depth=2, bucket_steps=2, capacity=1024, views=views)
        sg.world, sg._collective = 2, True
        got = []
        for j in range(3):
            sg.start(x * (j + 1), ei, w, b, B)
            got.extend(sg.take_ready())
        got.extend(sg.flush())
        assert len(got) == 3 and sg.capacity > 1024
        for j, (gx, gei, gw, gb) in enumerate(got):
            assert gx.dtype == torch.float64 and gw.dtype == torch.float64
            assert torch.equal(gx, torch.cat([x * (j + 1)] * 2)) and torch.equal(gw, torch.cat([w, w]))
            assert torch.equal(gei, torch.cat([ei, ei + K], 1)) and torch.equal(gb, torch.cat([b, b + B]))
            assert gei.is_contiguous() == (not views)
            if not views:
                assert gei.untyped_storage().nbytes() == gei.numel() * 8
    # integer features (bit copy) and bf16 weights (widened on the wire, narrowed back)
    sg = D.SparseGather(depth=1)
    sg.world, sg._collective = 2, True
    xi = torch.randint(-(1 << 40), 1 << 40, (K, F), generator=g).to(dev)
    sg.start(xi, ei, w.bfloat16(), None, B)
    gx, gei, gw, gb = sg.wait()
    assert gx.dtype == torch.int64 and torch.equal(gx, torch.cat([xi, xi]))
    assert gw.dtype == torch.bfloat16 and torch.equal(gw, torch.cat([w.bfloat16()] * 2)) and gb is None
    # ranks that disagree on the layout: rank "1" (the copy) is made to look different by a wrong local expectation
    sg = D.SparseGather(depth=1)
    sg.world, sg._collective = 2, True
    sg.start(x, ei, w, b, B)
    sg._inflight or sg._launch(partial=True)
    bucket = sg._inflight[0]
    import numpy as np
    torch.cuda.synchronize()
    slot = bucket["steps"][0][0]
    assert int(sg._host[slot * 8 + 4]) == 7
    sg._host[slot * 8 + 4] = 5  # what the unpack launch reports when a rank packed another layout
    with pytest.raises(RuntimeError, match="different feature widths"):
        sg.flush()


@pytest.mark.gpu
def test_copy_arrays_entry_point(dev):
    """tgp_copy_arrays: up to eight unrelated arrays in one launch (16-, 8- and 4-byte paths, empty arrays, odd offsets)."""
    import ctypes
    from tgp import _native as N
    g = torch.Generator().manual_seed(3)
    srcs = [torch.randn(n, generator=g).to(dev) for n in (1000, 7, 0, 4096, 33, 1, 12345, 64)]
    srcs[4] = srcs[4][1:]  # 4-byte aligned only
    dsts = [torch.full_like(s, -1.0) for s in srcs]
    n = len(srcs)
    N.check(N.lib().tgp_copy_arrays((ctypes.c_void_p * n)(*[s.data_ptr() for s in srcs]),
                                    (ctypes.c_void_p * n)(*[d.data_ptr() for d in dsts]),
                                    (ctypes.c_int64 * n)(*[s.numel() * 4 for s in srcs]), n, N.stream_ptr(dev)),
            "tgp_copy_arrays")
    for s, d in zip(srcs, dsts):
        assert torch.equal(s, d)
