"""Round-4 GPU parity tests (all through the C ABI): the dense in-place RCCL gather, the one-launch sparse pooling of
batches of small graphs, long rows in the coalesce Connect, float64 value types of the HBM-bound operators."""
import os
import socket

import pytest
import torch

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



def _oracle_topk_given_selection(O, x, ei, ew, batch, so, **kw):
    """The oracle's TopK Reduce + Connect (poolers/topk.py:150-190) evaluated on the selection the GPU made: two nodes whose
    scores differ in the last bit may be ranked differently by the CPU's summation order, which is Select's business
    (test_native_topk_select_vs_oracle pins it on exactly representable scores), not Reduce's or Connect's."""
    ni, ci, w = so.node_index.cpu(), so.cluster_index.cpu(), so.weight.detach().cpu()
    k = int(so.num_supernodes)
    xp = O.reduce_sparse(x.cpu(), ni, ci, w, k)
    bp = O.reduce_batch_sparse(batch.cpu(), ni, ci, k)
    rei, rew = O.sparse_connect(ei.cpu(), None if ew is None else ew.cpu(), ni, ci, x.size(0), k,
                                kw.get("remove_self_loops", True), "sum", kw.get("edge_weight_norm", False), bp,
                                kw.get("degree_norm", False))
    return dict(x=xp, edge_index=rei, edge_weight=rew, batch=bp)


def _staged_reduce_connect(pooler, x, ei, ew, so, batch):
    xp, bp = pooler.reducer(x, so, batch=batch)
    pe, pw = pooler.connector(ei, so, edge_weight=ew, batch_pooled=bp)
    return xp, bp, pe, pw


def _same(a, b):
    if a is None or b is None:
        return a is None and b is None
    return a.shape == b.shape and torch.equal(a, b)


@pytest.mark.parametrize("weighted", [True, False])
@pytest.mark.parametrize("f", [32, 7, 128])
@pytest.mark.parametrize("kw", [dict(), dict(remove_self_loops=False), dict(degree_norm=True, edge_weight_norm=True)])
def test_sparse_pool_small_topk_equals_staged_operators(dev, weighted, f, kw):
    """TopK: the one-launch Reduce + Connect of a batch of small graphs (tgp_sparse_pool_small_f32, mode 0) against
    BaseReduce + SparseConnect (reduce/base_reduce.py:141-155, connect/base_conn.py:79-82, utils/ops.py:338-419):
    x_pool, batch, edge_index and weights bit for bit, and against the oracle end to end."""
    import tgp_oracle as O
    from tgp.poolers import get_pooler
    x, ei, ew, batch, sizes = _small_batch(300, 1, 64, f, 5, dev)
    if not weighted:
        ew = None
    pooler = get_pooler("topk", in_channels=f, ratio=0.5, **kw).to(dev).eval()
    with torch.no_grad():
        so = pooler.selector(x=x, batch=batch)
        fused = pooler.reduce_connect(x, ei, ew, so, batch)
        assert fused is not None, "the one-launch path declined a sorted batch of small graphs"
        staged = _staged_reduce_connect(pooler, x, ei, ew, so, batch)
        out = pooler(x=x, adj=ei, edge_weight=ew, batch=batch)
    for a, b in zip(fused, staged):
        assert _same(a, b)
    assert _same(out.edge_index, staged[2]) and _same(out.x, staged[0])
    ref = _oracle_topk_given_selection(O, x, ei, ew, batch, out.so, **kw)
    assert torch.equal(out.edge_index.cpu(), ref["edge_index"]) and torch.equal(out.batch.cpu(), ref["batch"])
    torch.testing.assert_close(out.x.cpu(), ref["x"], rtol=1e-5, atol=1e-5)
    if ref["edge_weight"] is not None:
        torch.testing.assert_close(out.edge_weight.cpu(), ref["edge_weight"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("op", ["sum", "mean", "min", "max", "mul"])
@pytest.mark.parametrize("weighted", [True, False])
def test_sparse_pool_small_cluster_equals_staged_operators(dev, op, weighted):
    """Graclus-style clusterings (mode 1): relabel + coalesce with every reduce op, duplicate input entries included,
    against BaseReduce + SparseConnect's coalesce routes (connect/base_conn.py:83-89), bit for bit; and the oracle."""
    import tgp_oracle as O
    from tgp.poolers import get_pooler
    x, ei, ew, batch, sizes = _small_batch(257, 1, 64, 20, 11, dev, dup=True)
    if not weighted:
        ew = None
    pooler = get_pooler("graclus", connect_red_op=op).to(dev).eval()
    with torch.no_grad():
        so = pooler.selector(edge_index=ei, edge_weight=ew, num_nodes=x.size(0), batch=batch)
        fused = pooler.reduce_connect(x, ei, ew, so, batch)
        assert fused is not None, "the one-launch path declined a sorted batch of small graphs"
        staged = _staged_reduce_connect(pooler, x, ei, ew, so, batch)
    for a, b in zip(fused, staged):
        assert _same(a, b)
    ref = O.cluster_pool(x.cpu(), ei.cpu(), None if ew is None else ew.cpu(), batch.cpu(), so.cluster_index.cpu(),
                         so.num_supernodes, reduce_op=op)
    assert torch.equal(fused[2].cpu(), ref["edge_index"]) and torch.equal(fused[1].cpu(), ref["batch"])
    torch.testing.assert_close(fused[0].cpu(), ref["x"], rtol=1e-5, atol=1e-5)
    if ref["edge_weight"] is not None:
        torch.testing.assert_close(fused[3].cpu(), ref["edge_weight"], rtol=1e-5, atol=1e-5)


def test_sparse_pool_small_refuses_what_it_cannot_check_off(dev):
    """The kernel's on-device checks: an edge between two graphs, unsorted rows, a cluster spanning two graphs -> it
    declines (None) and the pooler's staged operators give the reference's result; a graph beyond 64 nodes is not
    offered to it at all."""
    import tgp_oracle as O
    from tgp import kernels as K
    from tgp.poolers import get_pooler
    from tgp.select import SelectOutput
    x, ei, ew, batch, sizes = _small_batch(50, 5, 40, 8, 3, dev)
    pooler = get_pooler("topk", in_channels=8, ratio=0.5).to(dev).eval()
    with torch.no_grad():
        so = pooler.selector(x=x, batch=batch)
        # (a) an edge that leaves its graph (legal for the reference: both endpoints kept or not decides)
        cross = torch.tensor([[0], [x.size(0) - 1]], device=dev)
        ei_a = torch.cat([cross, ei], 1)
        ew_a = torch.cat([torch.ones(1, device=dev), ew])
        assert pooler.reduce_connect(x, ei_a, ew_a, so, batch) is None
        assert K.sparse_pool_small_declined(ei_a)
        out = pooler(x=x, adj=ei_a, edge_weight=ew_a, batch=batch)
        ref = _oracle_topk_given_selection(O, x, ei_a, ew_a, batch, out.so)
        assert torch.equal(out.edge_index.cpu(), ref["edge_index"])
        # (b) rows in random order
        perm = torch.randperm(ei.size(1), device=dev)
        ei_b, ew_b = ei[:, perm].contiguous(), ew[perm]
        assert pooler.reduce_connect(x, ei_b, ew_b, so, batch) is None
        out = pooler(x=x, adj=ei_b, edge_weight=ew_b, batch=batch)
        ref = _oracle_topk_given_selection(O, x, ei_b, ew_b, batch, out.so)
        assert torch.equal(out.edge_index.cpu(), ref["edge_index"])
        torch.testing.assert_close(out.edge_weight.cpu(), ref["edge_weight"], rtol=1e-5, atol=1e-5)
        # (c) a clustering whose ids are not contiguous per graph (random labels over the whole batch)
        gp = get_pooler("graclus").to(dev).eval()
        g = torch.Generator().manual_seed(0)
        cl = torch.randint(0, 40, (x.size(0),), generator=g).to(dev)
        so_c = SelectOutput(cluster_index=cl, num_nodes=x.size(0), num_supernodes=40)
        assert gp.reduce_connect(x, ei, ew, so_c, batch) is None
        # (d) one graph of 65 nodes in the batch: never offered
        x2, ei2, ew2, batch2, _ = _small_batch(20, 65, 65, 8, 4, dev)
        so2 = pooler.selector(x=x2, batch=batch2)
        assert pooler.reduce_connect(x2, ei2, ew2, so2, batch2) is None


def test_sparse_pool_small_c3_sized_batch_and_empty_graphs(dev):
    """2048 PROTEINS-shaped graphs (the bench's topk_batch workload) and a batch with graphs that own no edge and no
    kept node: fused == staged, bit for bit; repeated calls reuse the epoch-tagged status buffer."""
    from tgp.poolers import get_pooler
    for num_graphs, lo, hi in ((2048, 20, 60), (700, 1, 3)):
        x, ei, ew, batch, sizes = _small_batch(num_graphs, lo, hi, 32, 9, dev, deg=2 if hi <= 3 else 4)
        for alias in ("topk", "graclus"):
            pooler = (get_pooler("topk", in_channels=32, ratio=0.5) if alias == "topk" else get_pooler("graclus")).to(dev).eval()
            with torch.no_grad():
                so = (pooler.selector(x=x, batch=batch) if alias == "topk" else
                      pooler.selector(edge_index=ei, edge_weight=ew, num_nodes=x.size(0), batch=batch))
                staged = _staged_reduce_connect(pooler, x, ei, ew, so, batch)
                for _ in range(3):
                    fused = pooler.reduce_connect(x, ei, ew, so, batch)
                    assert fused is not None
                    for a, b in zip(fused, staged):
                        assert _same(a, b)


# ------------------------------------------------------------------------------ hub rows in the row-local coalesce
def _hub_graph(n, pairs, hubs, hub_deg, seed):
    """Undirected, row-major sorted, duplicate-free edge list with `hubs` nodes of ~hub_deg neighbours each."""
    g = torch.Generator().manual_seed(seed)
    a = torch.randint(0, n, (pairs,), generator=g)
    b = torch.randint(0, n, (pairs,), generator=g)
    hub_ids = torch.randperm(n, generator=g)[:hubs]
    ha = hub_ids.repeat_interleave(hub_deg)
    hb = torch.randint(0, n, (hubs * hub_deg,), generator=g)
    a, b = torch.cat([a, ha]), torch.cat([b, hb])
    keep = a != b
    a, b = a[keep], b[keep]
    key = torch.unique(torch.cat([a * n + b, b * n + a]))
    return torch.stack([key // n, key % n]), hub_ids


@pytest.mark.parametrize("weighted", [True, False])
@pytest.mark.parametrize("op", ["sum", "mean", "min", "max", "mul"])
def test_rowlocal_coalesce_hub_rows_equal_the_radix_route(dev, weighted, op):
    """Supernode rows of 100 000 raw entries (hubs) inside a row-sorted list: the row-local route sorts those rows
    device-wide (TGP_HUGE_ROWS) and keeps every other row in LDS -- same edges, same weights (bit for bit: duplicates are
    folded in input order on both routes) as the general radix route and the oracle (connect/base_conn.py:83-89)."""
    import tgp_oracle as O
    from tgp import kernels
    n = 120_000
    ei, hub_ids = _hub_graph(n, 200_000, 5, 100_000, 41)
    g = torch.Generator().manual_seed(42)
    # pairs (Graclus-shaped) + a few clusters of many nodes; hubs are paired with each other -> rows of ~2 x 100 000
    perm = torch.randperm(n, generator=g)
    cluster = torch.empty(n, dtype=torch.long)
    cluster[perm] = torch.arange(n) // 2
    cluster[perm[:3000]] = 7          # one cluster of 3000 ordinary nodes: a long row made of many short members
    cluster[hub_ids[:2]] = 11         # two hubs in one supernode
    k = int(cluster.max()) + 1
    ew = (torch.rand(ei.size(1), generator=g) - 0.3) if weighted else None
    ei_d, cl_d = ei.to(dev), cluster.to(dev)
    ew_d = None if ew is None else ew.to(dev)
    idx = kernels.build_assign_index(cl_d, k)
    got_ei, got_ew = kernels.coalesce_edges(ei_d, ew_d, cl_d, k, op, True, assign_index=idx, route="staged")
    ref_ei, ref_ew = kernels.coalesce_edges(ei_d, ew_d, cl_d, k, op, True, route="general")
    assert torch.equal(got_ei, ref_ei)
    assert (got_ew is None) == (ref_ew is None)
    if got_ew is not None:
        assert torch.equal(got_ew, ref_ew)
    # a second call on the same edge_index object asks for the hub kernels at once (no -5 round trip) and agrees
    again_ei, again_ew = kernels.coalesce_edges(ei_d, ew_d, cl_d, k, op, True, assign_index=idx)
    assert torch.equal(again_ei, ref_ei) and (again_ew is None or torch.equal(again_ew, ref_ew))
    if op == "sum":
        o_ei, o_ew = O.sparse_connect(ei, ew, torch.arange(n), cluster, n, k, reduce_op=op)
        assert torch.equal(got_ei.cpu(), o_ei)
        if o_ew is not None:
            torch.testing.assert_close(got_ew.cpu(), o_ew, rtol=1e-5, atol=1e-5)


def test_graclus_pooler_on_a_hub_graph_stays_on_the_rowlocal_route(dev):
    """Whole `graclus` forward on a graph with hub nodes: the Connect no longer falls to the radix route for one long
    supernode row, and the result equals the oracle's Reduce + Connect given the same clustering."""
    import tgp_oracle as O
    from tgp import kernels
    from tgp.poolers import get_pooler
    n = 60_000
    ei, _ = _hub_graph(n, 150_000, 4, 30_000, 7)
    x = torch.randn(n, 16, generator=torch.Generator().manual_seed(1))
    pooler = get_pooler("graclus").to(dev).eval()
    ei_d = ei.to(dev)
    with torch.no_grad():
        out = pooler(x=x.to(dev), adj=ei_d)
    hub = kernels._HUB_ROWS.get(id(ei_d))
    assert hub is not None and hub[0]() is ei_d  # the row-local route met the hub rows and took them itself
    ref = O.cluster_pool(x, ei, None, None, out.so.cluster_index.cpu(), out.so.num_supernodes)
    assert torch.equal(out.edge_index.cpu(), ref["edge_index"])
    torch.testing.assert_close(out.x.cpu(), ref["x"], rtol=1e-5, atol=1e-5)


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


# ------------------------------------------------------------------------------ float64 value types (r4)
def _f64_graph(n, pairs, seed):
    g = torch.Generator().manual_seed(seed)
    a = torch.randint(0, n, (pairs,), generator=g)
    b = torch.randint(0, n, (pairs,), generator=g)
    keep = a != b
    a, b = a[keep], b[keep]
    key = torch.sort(torch.cat([a * n + b, b * n + a]))[0]       # duplicates stay: coalesce has something to merge
    ei = torch.stack([key // n, key % n])
    ew = torch.rand(ei.size(1), generator=g, dtype=torch.float64) - 0.2
    ew[torch.rand(ei.size(1), generator=g) < 0.05] = 1e-9        # below eps in fp64, and in fp32
    return ei, ew, g


def _oracle64(fn, *a, **k):
    """The oracle evaluated in float64 (its `torch.ones` defaults follow the default dtype)."""
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        return fn(*a, **k)
    finally:
        torch.set_default_dtype(old)


def test_float64_sparse_reduce_and_connect_run_in_fp64(dev):
    """model.double() inputs: sparse Reduce (base_reduce.py:141-155) and SparseConnect (base_conn.py:79-89 +
    utils/ops.py:338-419) compute in fp64 like the reference's ATen ops -- against the oracle evaluated in fp64 at
    1e-12 (fp32 arithmetic would miss by 1e-7), indices bit-exact, float64 outputs, and NO fp32-narrowing warning."""
    import warnings
    import tgp_oracle as O
    from tgp.connect import SparseConnect
    from tgp.reduce import BaseReduce
    from tgp.select import SelectOutput
    import tgp.utils.ops as ops
    n, f = 3000, 24
    ei, ew, g = _f64_graph(n, 12_000, 51)
    x = torch.randn(n, f, generator=g, dtype=torch.float64)
    batch = torch.sort(torch.randint(0, 7, (n,), generator=g))[0]
    ops._WARNED_F64 = False
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # the fp32-narrowing UserWarning must not fire on these paths
        # (a) kept-node selection with fp64 scores (TopK-shaped)
        kept = torch.sort(torch.randperm(n, generator=g)[: n // 2])[0]
        ci = torch.randperm(kept.numel(), generator=g)
        sw = torch.rand(kept.numel(), generator=g, dtype=torch.float64)
        so = SelectOutput(node_index=kept.to(dev), num_nodes=n, cluster_index=ci.to(dev), num_supernodes=kept.numel(),
                          weight=sw.to(dev))
        xp, bp = BaseReduce()(x.to(dev), so, batch=batch.to(dev))
        assert xp.dtype == torch.float64
        nis, cis, ws = O.sort_assignment(kept, ci, sw)
        ref = _oracle64(O.reduce_sparse, x, nis, cis, ws, kept.numel())
        torch.testing.assert_close(xp.cpu(), ref, rtol=1e-12, atol=1e-12)
        for kw in (dict(), dict(degree_norm=True, edge_weight_norm=True), dict(remove_self_loops=False)):
            pe, pw = SparseConnect(**kw)(ei.to(dev), so, edge_weight=ew.to(dev), batch_pooled=bp)
            r_ei, r_ew = _oracle64(O.sparse_connect, ei, ew, nis, cis, n, kept.numel(), batch_pooled=bp.cpu(), **kw)
            assert pw.dtype == torch.float64 and torch.equal(pe.cpu(), r_ei)
            torch.testing.assert_close(pw.cpu(), r_ew, rtol=1e-12, atol=1e-12)
        # (b) clustering (Graclus-shaped), every reduce op
        cl = torch.randint(0, n // 3, (n,), generator=g)
        cl[: n // 3] = torch.arange(n // 3)
        so2 = SelectOutput(cluster_index=cl.to(dev), num_nodes=n, num_supernodes=n // 3)
        xp2, _ = BaseReduce()(x.to(dev), so2)
        ref2 = _oracle64(O.reduce_sparse, x, torch.arange(n), cl, torch.ones(n, dtype=torch.float64), n // 3)
        assert xp2.dtype == torch.float64
        torch.testing.assert_close(xp2.cpu(), ref2, rtol=1e-12, atol=1e-12)
        for op in ("sum", "mean", "min", "max", "mul"):
            pe, pw = SparseConnect(reduce_op=op, degree_norm=(op == "sum"))(ei.to(dev), so2, edge_weight=ew.to(dev))
            r_ei, r_ew = _oracle64(O.sparse_connect, ei, ew, torch.arange(n), cl, n, n // 3, reduce_op=op,
                                   degree_norm=(op == "sum"))
            assert pw.dtype == torch.float64 and torch.equal(pe.cpu(), r_ei)
            torch.testing.assert_close(pw.cpu(), r_ew, rtol=1e-12, atol=1e-12)
        # fp32 features with fp64 assignment weights promote, as the reference's x[node_index] * weight does
        xp3, _ = BaseReduce()(x.float().to(dev), so)
        assert xp3.dtype == torch.float64
        torch.testing.assert_close(xp3.cpu(), _oracle64(O.reduce_sparse, x.float().double(), nis, cis, ws, kept.numel()),
                                   rtol=1e-12, atol=1e-12)


def test_float64_dense_postprocessing_and_block_diag_run_in_fp64(dev):
    """postprocess_adj_pool_dense on a float64 [B,K,K] tensor (utils/ops.py:282-335), all 16 flag combinations, and
    dense_to_block_diag (utils/ops.py:53-82) in fp64: 1e-12 against the oracle in fp64."""
    import tgp_oracle as O
    from tgp import functions as Fn
    from tgp.utils.ops import postprocess_adj_pool_dense
    g = torch.Generator().manual_seed(77)
    for (B, K) in ((5, 17), (2, 130)):
        a = torch.rand(B, K, K, generator=g, dtype=torch.float64) * (torch.rand(B, K, K, generator=g) < 0.4)
        a[0, :, 3] = 0.0  # an empty column: the clamp(min=eps) branch
        for rsl in (True, False):
            for dn in (True, False):
                for at in (True, False):
                    for ewn in (True, False):
                        got = postprocess_adj_pool_dense(a.to(dev), rsl, dn, at, ewn)
                        ref = O.postprocess_dense(a, rsl, dn, at, ewn)
                        assert got.dtype == torch.float64
                        torch.testing.assert_close(got.cpu(), ref, rtol=1e-12, atol=1e-12)
        ei, w = Fn.block_diag_edges(a.to(dev))
        r_ei, r_w = O.dense_to_block_diag(a)
        assert w.dtype == torch.float64 and torch.equal(ei.cpu(), r_ei) and torch.equal(w.cpu(), r_w)


def test_float64_dense_gemm_path_runs_in_fp64_without_a_warning(dev):
    """r5: the dense GEMM path has an fp64 form (v_mfma_f64_16x16x4_f64): a float64 DiffPool input is computed in double
    and nothing is announced (replaces r4's test that asserted the fp32 narrowing; the value checks live in
    tests/test_gpu_round5.py)."""
    import warnings
    from tgp.poolers import get_pooler
    x, ei, ew, batch, _ = _small_batch(8, 20, 40, 8, 1, dev)
    pooler = get_pooler("diff", in_channels=8, k=4).to(dev).double().eval()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        with torch.no_grad():
            out = pooler(x=x.double(), adj=ei, edge_weight=ew.double(), batch=batch)
    assert out.x.dtype == torch.float64 and out.edge_index.dtype == torch.float64


def test_bmm_accumulate_in_the_epilogue(dev):
    """C += op(A) B in the GEMM epilogue (tgp_bmm_accumulate_f32): the bits of the separate product + add, for both
    operand layouts, ragged shapes and the big-tile path; DenseConnect's two-term dS uses it."""
    from tgp import kernels as K
    g = torch.Generator(device=dev).manual_seed(0)
    for (G, M, Kd, Nc, ta) in ((3, 70, 33, 20, False), (2, 1024, 1024, 128, False), (4, 100, 257, 65, True)):
        a = torch.randn((G, Kd, M) if ta else (G, M, Kd), device=dev, generator=g)
        b = torch.randn(G, Kd, Nc, device=dev, generator=g)
        c0 = torch.randn(G, M, Nc, device=dev, generator=g)
        want = c0 + K.bmm(a, b, trans_a=ta)
        got = K.bmm(a, b, trans_a=ta, accumulate_into=c0.clone())
        assert torch.equal(got, want)


def test_graclus_optimistic_route_with_unsorted_rows(dev):
    """A batch of small graphs whose edge list nobody has looked at yet takes the one-launch matching optimistically
    (the row-order check rides on the offsets kernel).  With UNSORTED rows the offsets are no CSR: the per-graph kernel
    must refuse before reading through them (status 4) and the call must still return a valid maximal matching."""
    from tgp import kernels as K
    x, ei, ew, batch, sizes = _small_batch(200, 10, 60, 4, 33, dev)
    g = torch.Generator(device=dev).manual_seed(1)
    perm = torch.randperm(ei.size(1), device=dev, generator=g)
    ei_u, ew_u = ei[:, perm].contiguous(), ew[perm].abs() + 0.1  # a fresh tensor object: no row-order memo yet
    n = x.size(0)
    ptr = torch.zeros(sizes.numel() + 1, dtype=torch.long)
    ptr[1:] = torch.cumsum(sizes, 0)
    assert K._rows_sorted_memo(ei_u) is None
    label = K.graclus_match(ei_u, ew_u, n, graph_ptr=ptr.to(dev), max_graph_nodes=int(sizes.max()))
    assert K._rows_sorted_memo(ei_u) is False  # found out on the way, remembered
    lab = label.cpu()
    cnt = torch.bincount(lab, minlength=n)
    assert int(cnt.max()) <= 2 and bool((lab <= torch.arange(n)).all())
    r, c = ei_u.cpu()
    paired = cnt[lab] == 2
    partner_ok = torch.zeros(n, dtype=torch.bool)
    same = lab[r] == lab[c]
    partner_ok[r[same & (r != c)]] = True
    assert bool((partner_ok | ~paired).all())            # every pair is an edge
    free = ~paired
    assert not bool((free[r] & free[c] & (r != c)).any())  # maximal: no edge between two single nodes


@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("n,deg", [(5000, 6), (300_000, 10)])
def test_published_count_and_int32_columns_equal_the_plain_count_route(dev, weighted, n, deg, monkeypatch):
    """Row-sorted coalesce Connect (connect/base_conn.py:83-89): the survivor scan as one look-back launch that hands
    the count over in a pinned host word, with and without the int32 column copy of the list's CSR, against the
    count -> `.item()` -> fill pair and the general radix route -- same edges, same weights, bit for bit."""
    from tgp import kernels
    g = torch.Generator().manual_seed(n + deg)
    src = torch.randint(0, n, (n * deg,), generator=g)
    dst = torch.randint(0, n, (n * deg,), generator=g)
    ei = torch.stack([torch.cat([src, dst]), torch.cat([dst, src])])
    key = torch.unique(ei[0] * n + ei[1])
    ei = torch.stack([key // n, key % n])  # row-major sorted, no duplicates (PyG convention)
    perm = torch.randperm(n, generator=g)
    cluster = torch.empty(n, dtype=torch.long)
    cluster[perm] = torch.arange(n) // 2
    k = int(cluster.max()) + 1
    ew = (torch.rand(ei.size(1), generator=g) - 0.3) if weighted else None
    ei_d, cl_d = ei.to(dev), cluster.to(dev)
    ew_d = None if ew is None else ew.to(dev)
    idx = kernels.build_assign_index(cl_d, k)
    ptr = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    ptr[1:] = torch.bincount(ei_d[0], minlength=n).cumsum(0).to(torch.int32)
    csr = (ptr, ei_d[1].to(torch.int32).contiguous())
    ref_ei, ref_ew = kernels.coalesce_edges(ei_d, ew_d, cl_d, k, "sum", True, route="general")
    monkeypatch.setattr(kernels, "_PUBLISH_COUNTS", False)
    plain_ei, plain_ew = kernels.coalesce_edges(ei_d, ew_d, cl_d, k, "sum", True, assign_index=idx, route="staged")
    monkeypatch.setattr(kernels, "_PUBLISH_COUNTS", True)
    for kw in ({}, {"csr": csr}, {"csr": (ptr, None)}):
        for _ in range(3):  # the status words of the previous calls are stale, never cleared
            got_ei, got_ew = kernels.coalesce_edges(ei_d, ew_d, cl_d, k, "sum", True, assign_index=idx, route="staged",
                                                    **kw)
            assert torch.equal(got_ei, ref_ei) and torch.equal(got_ei, plain_ei)
            if weighted:
                assert torch.equal(got_ew, plain_ew) and torch.equal(got_ew, ref_ew)
            else:
                assert got_ew is None


@pytest.mark.parametrize("n,weighted", [(50_000, True), (50_000, False), (200_000, True)])
def test_ndp_large_partition_with_hub_rows_vs_scipy(dev, n, weighted):
    """NDPSelect's chip-wide LOBPCG (select/ndp_select.py:187-256) on a graph with hub nodes: rows beyond 1024 entries
    are listed and reduced by whole workgroups in the start, mat-vec and cut kernels (n <= 131072: the two-launch step
    with round B folded into the mat-vec; beyond: the three-launch step).  Same contract as the hub-free test: the
    Rayleigh quotient equals scipy's largest eigenvalue of Ls within 1e-6, the residual meets the tolerance, the cut
    of the returned partition is the one the kernels report."""
    import numpy as np
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    from tgp import kernels as K
    g = torch.Generator().manual_seed(n + int(weighted))
    a = torch.randint(0, n, (4 * n,), generator=g)
    b = torch.randint(0, n, (4 * n,), generator=g)
    hubs = [0, 1, 2, n // 2, n - 1]                      # neighbouring hubs + two elsewhere
    degs = [30_000, 5_000, 1_500, 12_000, 2_000]         # one barely beyond the threshold
    ha = torch.cat([torch.full((d,), h) for h, d in zip(hubs, degs)])
    hb = torch.cat([torch.randint(0, n, (d,), generator=g) for d in degs])
    a, b = torch.cat([a, ha]), torch.cat([b, hb])
    keep = a != b
    a, b = a[keep], b[keep]
    key = torch.unique(torch.cat([a * n + b, b * n + a]))
    ei = torch.stack([key // n, key % n])
    if weighted:
        k2 = torch.minimum(ei[0], ei[1]) * n + torch.maximum(ei[0], ei[1])
        w = ((k2 * 2654435761) % 1000).float() / 1000 + 0.5
    else:
        w = None
    vals = np.ones(ei.size(1)) if w is None else w.double().numpy()
    A = sp.coo_matrix((vals, (ei[0].numpy(), ei[1].numpy())), shape=(n, n)).tocsr()
    deg = np.asarray(A.sum(1)).reshape(-1)
    assert int((np.diff(A.indptr) > 1024).sum()) == len(hubs)
    dis = np.where(deg > 0, 1.0 / np.sqrt(np.maximum(deg, 1e-300)), 0.0)
    Ls = sp.eye(n) - sp.diags(dis) @ A @ sp.diags(dis)
    lam_ref = float(spla.eigsh(Ls.tocsc(), k=1, which="LA", tol=1e-10, return_eigenvectors=False)[0])
    eid = ei.to(dev)
    wd = None if w is None else w.to(dev)
    indptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    K.rowptr_from_sorted(eid[0], n, indptr)
    keepv = torch.zeros(n, dtype=torch.uint8, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    info, state = K.ndp_partition_large(indptr, eid[1], wd, 0, n, 7, keepv, status, want_state=True)
    assert int(status.item()) == 0
    assert abs(state["lambda"] - lam_ref) <= 1e-6 * lam_ref, (state, lam_ref)
    assert state["residual_sq"] <= (1e-6 * state["lambda"]) ** 2 * 1.0001
    kb = keepv.bool().cpu()
    assert 0 < int(kb.sum()) < n
    z = np.where(kb.numpy(), 1.0, -1.0)
    L = sp.diags(deg) - A
    cut = float(z @ (L @ z)) / (2.0 * A.sum())
    if int(info.item()) >= 0:
        assert cut >= 0.5 and abs(cut - state["cut"]) < 1e-9
    else:
        assert bool(kb[0]) and not bool(kb[1])
    # the same call again gives the same partition bit for bit (the hub list is sorted: fixed summation order)
    keep2 = torch.zeros(n, dtype=torch.uint8, device=dev)
    info2, state2 = K.ndp_partition_large(indptr, eid[1], wd, 0, n, 7, keep2, status, want_state=True)
    assert torch.equal(keep2, keepv) and state2["lambda"] == state["lambda"] and state2["steps"] == state["steps"]


def _graclus_select_outputs(ei, ew, n, gptr, gmax):
    from tgp import kernels
    (index, k, assign, ones), _ = kernels.graclus_match(ei, ew, n, return_row_ptr=True, graph_ptr=gptr,
                                                        max_graph_nodes=gmax, relabel=True)
    return index, k, assign.row_ptr[:k + 1].clone(), assign.perm[:n].clone(), ones


@pytest.mark.parametrize("weighted", [True, False])
@pytest.mark.parametrize("shape", [(300, 5, 64, 4, False), (2048, 20, 60, 4, False), (257, 1, 30, 6, True),
                                   (64, 40, 64, 16, False)])
def test_one_launch_graclus_select_equals_the_staged_route(dev, weighted, shape, monkeypatch):
    """GraclusSelect of a sorted batch of small graphs (select/graclus_select.py:62-81) as ONE launch -- matching, consecutive
    cluster ids, supernode -> members index -- against the staged route (offsets, CSR gather, symmetrise, one-launch rounds,
    relabel kernels): the same pairs (both use the key on global ids), hence the same index / K / members index, bit for
    bit.  Ragged sizes incl. one-node graphs, isolated nodes, duplicate entries, weight ties (unweighted)."""
    from tgp import kernels
    from tgp.utils.ops import batch_info
    B, lo, hi, deg, dup = shape
    x, ei, ew, batch, sizes = _small_batch(B, lo, hi, 8, 100 + B, dev, deg=deg, dup=dup)
    ew = ew if weighted else None
    n = x.size(0)
    info = batch_info(batch)
    monkeypatch.setattr(kernels, "_GRACLUS_FUSED", False)
    ref = _graclus_select_outputs(ei, ew, n, info.ptr, info.max_nodes)
    monkeypatch.setattr(kernels, "_GRACLUS_FUSED", True)
    for _ in range(3):  # stale status words of earlier calls must read as "not ready"
        got = _graclus_select_outputs(ei, ew, n, info.ptr, info.max_nodes)
        assert got[1] == ref[1]
        for a, b in zip((got[0], got[2], got[3], got[4]), (ref[0], ref[2], ref[3], ref[4])):
            assert torch.equal(a, b)
    # the cluster ids describe a maximal matching: every cluster has one or two members, two members share an edge
    idx, k = got[0], got[1]
    sizes_c = torch.bincount(idx[1], minlength=k)
    assert int(sizes_c.min()) >= 1 and int(sizes_c.max()) <= 2


def test_one_launch_graclus_select_refusals_fall_back(dev, monkeypatch):
    """What the one-launch kernel refuses still gives the staged route's answer through the same call: a directed list
    (entries without a reverse are dropped by both), an UNSORTED list (refused: rows not ascending), a batch with a graph
    beyond 64 nodes (not attempted), an edge that leaves its graph (refused by both per-graph kernels: device-wide rounds)."""
    from tgp import kernels
    from tgp.utils.ops import batch_info
    x, ei, ew, batch, sizes = _small_batch(200, 10, 50, 8, 7, dev)
    n = x.size(0)
    info = batch_info(batch)

    def both(e, w, gptr, gmax):
        monkeypatch.setattr(kernels, "_GRACLUS_FUSED", False)
        ref = _graclus_select_outputs(e.clone(), w, n, gptr, gmax)
        monkeypatch.setattr(kernels, "_GRACLUS_FUSED", True)
        got = _graclus_select_outputs(e.clone(), w, n, gptr, gmax)
        assert got[1] == ref[1]
        for a, b in zip((got[0], got[2], got[3], got[4]), (ref[0], ref[2], ref[3], ref[4])):
            assert torch.equal(a, b)
        return got

    keep = torch.rand(ei.size(1), generator=torch.Generator().manual_seed(3)).to(dev) < 0.8   # directed: reverses missing
    both(ei[:, keep].contiguous(), ew[keep].contiguous(), info.ptr, info.max_nodes)
    perm = torch.randperm(ei.size(1), generator=torch.Generator().manual_seed(4)).to(dev)       # unsorted rows
    both(ei[:, perm].contiguous(), ew[perm].contiguous(), info.ptr, info.max_nodes)
    leak = ei.clone()
    leak[1, 0] = n - 1                                                                         # leaves its graph
    both(leak, ew, info.ptr, info.max_nodes)
    x2, ei2, ew2, batch2, _ = _small_batch(40, 30, 100, 8, 9, dev)                             # graphs beyond 64 nodes
    info2 = batch_info(batch2)
    n2 = x2.size(0)
    monkeypatch.setattr(kernels, "_GRACLUS_FUSED", True)
    (index, k, assign, ones), _ = kernels.graclus_match(ei2, ew2, n2, return_row_ptr=True, graph_ptr=info2.ptr,
                                                        max_graph_nodes=info2.max_nodes, relabel=True)
    assert index.shape == (2, n2) and 0 < k <= n2


def test_ndp_select_with_an_unsorted_batch_vector_stays_on_device(dev, monkeypatch):
    """NDPSelect (select/ndp_select.py:187-256) on a batch whose nodes are NOT grouped by graph: r3 handed such batches to
    the host (scipy eigsh per graph); now the nodes are renumbered graph by graph on the device, partitioned by the same
    kernels and the kept set mapped back -- equal to the sorted batch's selection under the node permutation."""
    import scipy.sparse.linalg as spla
    from tgp.select import NDPSelect

    def boom(*a, **k):
        raise AssertionError("host eigen-solver called")
    monkeypatch.setattr(spla, "eigsh", boom)
    x, ei, ew, batch, sizes = _small_batch(60, 8, 50, 4, 33, dev)
    n = x.size(0)
    sel = NDPSelect()
    torch.manual_seed(5)                          # (the random-fallback seed is drawn from torch's generator)
    so_sorted = sel(ei, ew, batch=batch, num_nodes=n)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(6)).to(dev)   # new id of node i: perm[i]
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(n, device=dev)
    ei_u = perm[ei]
    order = torch.argsort(ei_u[0] * n + ei_u[1])
    ei_u, ew_u = ei_u[:, order].contiguous(), ew[order].contiguous()
    batch_u = batch[inv].contiguous()
    assert not bool((batch_u[1:] >= batch_u[:-1]).all())
    torch.manual_seed(5)
    so_u = sel(ei_u, ew_u, batch=batch_u, num_nodes=n)
    ni = so_u.node_index
    assert ni.is_cuda and bool((ni[1:] > ni[:-1]).all())
    assert torch.equal(so_u.cluster_index, torch.arange(ni.numel(), device=dev))
    # graph by graph the same nodes are kept.  The stable renumbering keeps the order of a graph's nodes as they appear
    # in the unsorted numbering, which differs from the sorted batch's order: the spectral partition is the same SET up
    # to the eigenvector's sign, the random fallback (cut < 0.5) is not comparable -- compare the graphs that kept it
    kept_sorted = torch.zeros(n, dtype=torch.bool, device=dev)
    kept_sorted[so_sorted.node_index] = True
    kept_u = torch.zeros(n, dtype=torch.bool, device=dev)
    kept_u[ni] = True
    kept_u_in_sorted_ids = kept_u[perm]
    spectral = (so_sorted._partition_info >= 0) & (so_u._partition_info >= 0)
    assert int(spectral.sum()) > 0
    same = 0
    for g in spectral.nonzero().view(-1).tolist():
        m = batch == g
        a, b = kept_sorted[m], kept_u_in_sorted_ids[m]
        assert torch.equal(a, b) or torch.equal(a, ~b)   # the sign of an eigenvector is a convention
        same += 1
    assert same == int(spectral.sum())
    # the reference's so.L in the caller's numbering (built lazily on the host)
    L = so_u.L
    assert L.shape == (n, n) and abs(L.sum()) < 1e-3


@pytest.mark.parametrize("alias", ["topk", "graclus"])
def test_one_launch_pooling_rechecks_the_offsets_it_is_handed(dev, alias):
    """The per-graph edge offsets the one-launch kernels are handed (r5: left behind by the first call on an edge list --
    `edge_ptr_out` -- and remembered per edge-list object; r4: a lower-bounds launch) are NOT trusted: a table that is off
    by one edge somewhere makes an edge fall outside its graph's node range -> refusal -> the staged operators give the
    same result as with a correct table."""
    from tgp import kernels
    from tgp.poolers import get_pooler
    if not kernels._SPS_GIVE_PTRS:
        pytest.skip("TGP_SPS_GIVE_PTRS=0: the kernels search for their ranges themselves")
    x, ei, ew, batch, sizes = _small_batch(200, 10, 60, 16, 77, dev)
    kw = dict(in_channels=16, ratio=0.5) if alias == "topk" else {}
    pooler = get_pooler(alias, **kw).to(dev).eval()
    with torch.no_grad():
        good = pooler(x=x, adj=ei, edge_weight=ew, batch=batch)
        ei2 = ei.clone()                 # (a new object: no "declined before" memo, no cached selection)
        first = pooler(x=x, adj=ei2, edge_weight=ew, batch=batch)   # searches for its ranges, leaves them in the memo
    assert not kernels.sparse_pool_small_declined(ei2)
    assert torch.equal(first.edge_index, good.edge_index) and torch.equal(first.x, good.x)
    hit = kernels._EDGE_PTR.get(id(ei2))
    assert hit is not None and hit[0]() is ei2, "the first call must have remembered the ranges it searched for"
    table = hit[4]
    gp = hit[3]()                        # the batch vector's CSR offsets the table belongs to
    assert torch.equal(table, torch.searchsorted(ei2[0].contiguous(), gp))  # = lower bounds of graph_ptr in the row array
    table[table.numel() // 2] += 1       # still ascending, still 0 .. E: one edge now sits in the wrong graph's range
    with torch.no_grad():
        bad = pooler(x=x, adj=ei2, edge_weight=ew, batch=batch)
    assert kernels.sparse_pool_small_declined(ei2)
    assert torch.equal(bad.edge_index, good.edge_index) and torch.equal(bad.x, good.x)
    assert torch.equal(bad.edge_weight, good.edge_weight) and torch.equal(bad.batch, good.batch)


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
