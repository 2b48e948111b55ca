"""world_size-2 CPU (gloo) test of the N>1 path: graph-id sharding, the all-gather of pooled outputs and
the id-offset merge rule.  The per-shard pooling itself is done by the CPU oracle here (tests may use it);
on GPUs the same functions receive the HIP path's outputs (bench.py --gpus N)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_batch():
    g = torch.Generator().manual_seed(0)
    sizes = [9, 14, 6, 11, 8]
    xs, eis, ews, bs, off = [], [], [], [], 0
    for gi, n in enumerate(sizes):
        a = torch.triu(torch.rand(n, n, generator=g) < 0.4, 1)
        a = a | a.t()
        ei = a.nonzero().t() + off
        eis.append(ei)
        ews.append(torch.rand(ei.size(1), generator=g) + 0.1)
        xs.append(torch.randn(n, 4, generator=g))
        bs.append(torch.full((n,), gi))
        off += n
    return torch.cat(xs), torch.cat(eis, 1), torch.cat(ews), torch.cat(bs), len(sizes)


def _worker(rank, world, port, out_dir):
    for p in (os.path.join(ROOT, "torch-geometric-pool_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import tgp_oracle as O
    from tgp import distributed as D
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x, ei, ew, batch, nb = _make_batch()
        p = torch.linspace(-1, 1, 4).view(1, 4)
        # ---- sparse pooler (TopK) on this rank's graphs
        xl, eil, ewl, bl = D.shard_sparse_batch(x, ei, ew, batch, rank, world, nb)
        lo, hi = D.shard_bounds(nb, world)[rank]
        loc = O.topk_pool(xl, eil, ewl, bl, p, ratio=0.5)
        gx, gei, gew, gb = D.all_gather_sparse(loc["x"], loc["edge_index"], loc["edge_weight"], loc["batch"], hi - lo)
        # ---- dense pooler outputs (fixed shape per graph)
        xd, ad, mask = O.dense_preprocessing(x, ei, ew, batch, True)
        s = torch.softmax(xd @ torch.ones(4, 3), -1) * mask.unsqueeze(-1)
        sl, al, xdl = D.shard_dense_batch(rank, world, s, ad, xd)
        xp = O.reduce_dense(sl, xdl)
        ap = O.postprocess_dense(O.dense_connect(sl, al), True, True, True, False)
        gxp, gap = D.all_gather_dense([xp, ap])
        # packed single-collective variant (equal graph counts per rank): graphs 0..3 only
        pg = D.PackedGather()
        pg.start([xp[:2], ap[:2]])
        pxp, pap = pg.wait()
        # bucketed variant: 5 steps, 3 per collective -> one full bucket + one flushed partial bucket; step j
        # sends (j + 1) * its tensors so that steps cannot be confused
        pb = D.PackedGather(bucket_steps=3)
        handed = []
        for j in range(5):
            pb.start([xp[:2] * (j + 1), ap[:2] * (j + 1)])
            r = pb.wait() if j == 3 else None  # the full bucket is in flight from step 2 on
            if r:
                handed.extend(r)
        handed.extend(pb.flush())
        bucket_ok = len(handed) == 5 and all(
            torch.allclose(h[0], pxp * (j + 1)) and torch.allclose(h[1], pap * (j + 1)) for j, h in enumerate(handed))
        assert pb.flush() == [] and pb.wait() is None
        # in-place mode: the producer writes into the slots the gather hands out (what bench.py does with the
        # kernels' out_x / out_adj); no pack copy, same results
        pi = D.PackedGather(bucket_steps=2)
        inplace = []
        for j in range(3):
            sx, sa = pi.slots([xp[:2].shape, ap[:2].shape], dtype=xp.dtype, device=xp.device)
            sx.copy_(xp[:2] * (j + 1))
            sa.copy_(ap[:2] * (j + 1))
            ptrs = (sx.data_ptr(), sa.data_ptr())
            pi.start([sx, sa])
            assert (sx.data_ptr(), sa.data_ptr()) == ptrs
        inplace.extend(pi.flush())
        bucket_ok = bucket_ok and len(inplace) == 3 and all(
            torch.allclose(h[0], pxp * (j + 1)) and torch.allclose(h[1], pap * (j + 1)) for j, h in enumerate(inplace))
        # asynchronous variable-size gather: three steps in flight two deep, a capacity that is too small at first (the
        # ranks grow it from the headers, alike, and repeat the step) -- every step's merged result must be the
        # synchronous one scaled by its step number
        sg = D.SparseGather(depth=2, capacity=256, bucket_steps=2)
        outs = []
        for j in range(3):
            sg.start(loc["x"] * (j + 1), loc["edge_index"], loc["edge_weight"] * (j + 1), loc["batch"], hi - lo)
            outs.extend(sg.take_ready())
        outs.extend(sg.flush())
        sparse_ok = len(outs) == 3 and sg.capacity > 256 and all(
            torch.allclose(o[0], gx * (j + 1)) and torch.equal(o[1], gei) and torch.allclose(o[2], gew * (j + 1))
            and torch.equal(o[3], gb) for j, o in enumerate(outs))
        # unweighted lists and 1-D features travel too
        sg2 = D.SparseGather(depth=1)
        sg2.start(loc["x"][:, 0].contiguous(), loc["edge_index"], None, loc["batch"], hi - lo)
        o2 = sg2.wait()
        sparse_ok = sparse_ok and o2[2] is None and torch.equal(o2[1], gei) and torch.allclose(o2[0], gx[:, 0])
        if rank == 0:
            torch.save(dict(gx=gx, gei=gei, gew=gew, gb=gb, gxp=gxp, gap=gap, pxp=pxp, pap=pap,
                            bucket_ok=torch.tensor(bucket_ok), sparse_ok=torch.tensor(sparse_ok)),
                       os.path.join(out_dir, "gathered.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_shard_bounds():
    sys.path.insert(0, os.path.join(ROOT, "torch-geometric-pool_amd"))
    from tgp.distributed import shard_bounds
    assert shard_bounds(5, 2) == [(0, 3), (3, 5)]
    assert shard_bounds(8, 8) == [(i, i + 1) for i in range(8)]
    assert shard_bounds(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)]


@pytest.mark.timeout(180)
def test_two_rank_gather_matches_single_process(tmp_path):
    import tgp_oracle as O
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = torch.load(os.path.join(tmp_path, "gathered.pt"), weights_only=True)
    x, ei, ew, batch, nb = _make_batch()
    p = torch.linspace(-1, 1, 4).view(1, 4)
    full = O.topk_pool(x, ei, ew, batch, p, ratio=0.5)
    # Graph-id shards are contiguous and TopK numbers its supernodes graph-major, so the merged outputs must be the
    # single-process ones EXACTLY: batch vector and edge_index bit for bit (north_star: indices bit-exact), x rows in
    # the same order, and -- per graph, after taking the graph's first supernode id off -- the same local edge lists.
    assert torch.equal(got["gb"], full["batch"])
    assert torch.equal(got["gei"], full["edge_index"])
    torch.testing.assert_close(got["gx"], full["x"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(got["gew"], full["edge_weight"], rtol=1e-6, atol=1e-6)
    for gi in range(nb):
        first_g = int((got["gb"] == gi).nonzero()[0])
        first_f = int((full["batch"] == gi).nonzero()[0])
        eg = got["gei"][:, got["gb"][got["gei"][0]] == gi] - first_g
        ef = full["edge_index"][:, full["batch"][full["edge_index"][0]] == gi] - first_f
        assert torch.equal(eg, ef), gi
    # edges stay inside their graph after the offsets are applied
    assert torch.equal(got["gb"][got["gei"][0]], got["gb"][got["gei"][1]])
    # dense outputs: simple concatenation over graphs
    xd, ad, mask = O.dense_preprocessing(x, ei, ew, batch, True)
    s = torch.softmax(xd @ torch.ones(4, 3), -1) * mask.unsqueeze(-1)
    torch.testing.assert_close(got["gxp"], O.reduce_dense(s, xd), rtol=1e-6, atol=1e-6)
    ap_full = O.postprocess_dense(O.dense_connect(s, ad), True, True, True, False)
    torch.testing.assert_close(got["gap"], ap_full, rtol=1e-6, atol=1e-6)
    # packed gather took graphs [0,1] of rank 0 (global 0,1) and [0,1] of rank 1 (global 3,4)
    torch.testing.assert_close(got["pxp"], O.reduce_dense(s, xd)[[0, 1, 3, 4]], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(got["pap"], ap_full[[0, 1, 3, 4]], rtol=1e-6, atol=1e-6)
    assert bool(got["bucket_ok"])  # bucketed gather: five steps, three per collective, order and values kept
    assert bool(got["sparse_ok"])  # SparseGather: async steps, capacity growth agreed from the headers, no weights, 1-D x
