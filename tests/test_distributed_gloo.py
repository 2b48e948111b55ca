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
        # r5: values keep their dtype across ranks -- float64 features and weights bit for bit (they were
        # narrowed to fp32 on the way), int64 features as integers, bf16 back as bf16; results are fresh contiguous
        # tensors of exactly the merged size; the synchronous helper keeps its agreed capacity between calls
        x64 = loc["x"].double() * (1.0 + 2.0 ** -40)        # not representable in fp32
        w64 = loc["edge_weight"].double() + 2.0 ** -45
        o64 = D.all_gather_sparse(x64, loc["edge_index"], w64, loc["batch"], hi - lo)
        parts_x, parts_w = [None] * world, [None] * world
        dist.all_gather_object(parts_x, x64)
        dist.all_gather_object(parts_w, w64)
        dtype_ok = (o64[0].dtype == torch.float64 and o64[2].dtype == torch.float64
                    and torch.equal(o64[0], torch.cat(parts_x)) and torch.equal(o64[2], torch.cat(parts_w))
                    and torch.equal(o64[1], gei) and o64[1].is_contiguous()
                    and o64[1].untyped_storage().nbytes() == o64[1].numel() * 8)
        xi = (loc["x"] * 1000).long() + (1 << 40)
        oi = D.all_gather_sparse(xi, loc["edge_index"], loc["edge_weight"].bfloat16(), loc["batch"], hi - lo)
        dist.all_gather_object(parts_x, xi)
        dtype_ok = (dtype_ok and oi[0].dtype == torch.int64 and torch.equal(oi[0], torch.cat(parts_x))
                    and oi[2].dtype == torch.bfloat16)
        from tgp.distributed import _SYNC_GATHERS
        dtype_ok = dtype_ok and len(_SYNC_GATHERS) == 1  # one cached gather per process group, capacity kept
        # polling and blocking consumption must not be mixed on one gather
        sg3 = D.SparseGather(depth=1)
        sg3.start(loc["x"], loc["edge_index"], None, loc["batch"], hi - lo)
        sg3.take_ready()
        try:
            sg3.wait()
            mode_ok = False
        except RuntimeError as e:
            mode_ok = "must not be mixed" in str(e)
        sg3.flush()
        # ranks that pack different layouts for the same step are told so (rank 1 sends no weights here) instead of being
        # handed views of uninitialised memory
        sg4 = D.SparseGather(depth=1)
        sg4.start(loc["x"], loc["edge_index"], loc["edge_weight"] if rank == 0 else None, loc["batch"], hi - lo)
        try:
            sg4.flush()
            agree_ok = False
        except RuntimeError as e:
            agree_ok = "different feature widths" in str(e)
        if rank == 0:
            torch.save(dict(gx=gx, gei=gei, gew=gew, gb=gb, gxp=gxp, gap=gap, pxp=pxp, pap=pap,
                            bucket_ok=torch.tensor(bucket_ok), sparse_ok=torch.tensor(sparse_ok),
                            dtype_ok=torch.tensor(dtype_ok), mode_ok=torch.tensor(mode_ok),
                            agree_ok=torch.tensor(agree_ok)),
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
    assert bool(got["dtype_ok"])   # float64 / int64 / bf16 values keep their dtype and bits; contiguous exact-size results
    assert bool(got["mode_ok"])    # take_ready + wait on one gather raises
    assert bool(got["agree_ok"])   # ranks that disagree on the packed layout are told so


def _sized_batch(sizes, seed, f=4):
    g = torch.Generator().manual_seed(seed)
    xs, eis, ews, bs, off = [], [], [], [], 0
    for gi, n in enumerate(sizes):
        a = torch.triu(torch.rand(n, n, generator=g) < 0.4, 1)
        a = a | a.t()
        ei = a.nonzero().t() + off
        eis.append(ei)
        ews.append(torch.rand(ei.size(1), generator=g) + 0.1)
        xs.append(torch.randn(n, f, generator=g))
        bs.append(torch.full((n,), gi))
        off += n
    return torch.cat(xs), torch.cat(eis, 1), torch.cat(ews), torch.cat(bs), len(sizes)


_STEP_SIZES = [[5, 6, 4], [7, 3, 5], [5, 60, 4], [4, 4, 4], [6, 5, 7]]  # step 2: rank 1's graph outgrows the slot


def _worker4(rank, world, port, out_dir):
    """world_size 4 over 3 graphs: rank 3 holds NO graph in any step; rank 1's payload overflows the slot capacity at step
    2 only, so the capacity change must be decided at the same point of the collective sequence on all four ranks although
    every rank polls (take_ready) at its own cadence.  (tgp/data/collate.py:144-153: the merge rule under test.)"""
    for p in (os.path.join(ROOT, "torch-geometric-pool_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    from tgp import distributed as D
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sg = D.SparseGather(depth=2, capacity=4096, bucket_steps=2)
        outs = []
        for j, sizes in enumerate(_STEP_SIZES):
            x, ei, ew, batch, nb = _sized_batch(sizes, seed=100 + j)
            lo, hi = D.shard_bounds(nb, world)[rank]
            xl, eil, ewl, bl = D.shard_sparse_batch(x, ei, ew, batch, rank, world, nb)
            if rank == 3:
                assert xl.size(0) == 0 and eil.size(1) == 0 and hi == lo
            # the "pooled outputs" of this test are the shard itself (an identity pooler): the gather must rebuild the batch
            sg.start(xl, eil, ewl, bl, hi - lo)
            if j % (rank + 1) == 0:          # every rank polls at its own cadence
                outs.extend(sg.take_ready())
        outs.extend(sg.flush())
        caps = [None] * world
        dist.all_gather_object(caps, sg.capacity)
        ok = len(outs) == len(_STEP_SIZES) and len(set(caps)) == 1 and caps[0] > 4096
        for j, (sizes, o) in enumerate(zip(_STEP_SIZES, outs)):
            x, ei, ew, batch, nb = _sized_batch(sizes, seed=100 + j)
            ok = ok and (torch.equal(o[0], x) and torch.equal(o[1], ei) and torch.equal(o[2], ew)
                         and torch.equal(o[3], batch) and o[1].is_contiguous())
        # the synchronous helper with an empty rank, twice (the second call reuses the gather and its capacity)
        for j in (2, 0):
            x, ei, ew, batch, nb = _sized_batch(_STEP_SIZES[j], seed=100 + j)
            lo, hi = D.shard_bounds(nb, world)[rank]
            xl, eil, ewl, bl = D.shard_sparse_batch(x, ei, ew, batch, rank, world, nb)
            mx, me, mw, mb = D.all_gather_sparse(xl, eil, ewl, bl, hi - lo)
            ok = ok and torch.equal(mx, x) and torch.equal(me, ei) and torch.equal(mw, ew) and torch.equal(mb, batch)
        oks = [None] * world
        dist.all_gather_object(oks, bool(ok))
        if rank == 0:
            torch.save(dict(ok=torch.tensor(all(oks)), caps=torch.tensor(caps)), os.path.join(out_dir, "w4.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_four_rank_sparse_gather_with_an_empty_rank_and_a_late_overflow(tmp_path):
    world = 4
    mp.spawn(_worker4, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = torch.load(os.path.join(tmp_path, "w4.pt"), weights_only=True)
    assert bool(got["ok"]), got["caps"]
