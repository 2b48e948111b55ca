"""Sparse Connect (csrc/sparse_connect.hip, coalesce_rows.hip, sparse_f64.hip; reference connect/base_conn.py:57-112, utils/ops.py:338-419): coalesce routes, subgraph route, filters, gradients of the edge weights, float64.

Regrouped by operator in round 6 from the per-round files test_gpu_round2..5.py; the test bodies are unchanged."""
import math
import warnings
import pytest
import torch
import os
import socket
import sys

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


# --------------------------------------------------------------------------- SparseConnect weight gradients
def _dense_ref_coalesce(ei, w, cl, k, op, remove_self_loops, eps=1e-8):
    """Differentiable torch restatement of cluster -> coalesce(reduce=op) -> filters, returning the pooled weights in
    row-major order (what PyG's scatter-based coalesce + postprocess_adj_pool_sparse compute)."""
    key = cl[ei[0]] * k + cl[ei[1]]
    uniq, inv = torch.unique(key, return_inverse=True)
    if op in ("sum", "mean"):
        out = torch.zeros(uniq.numel(), dtype=w.dtype).index_add(0, inv, w)
        if op == "mean":
            out = out / torch.bincount(inv, minlength=uniq.numel()).to(w.dtype)
    elif op == "mul":
        out = torch.ones(uniq.numel(), dtype=w.dtype).scatter_reduce(0, inv, w, "prod", include_self=True)
    else:
        out = torch.zeros(uniq.numel(), dtype=w.dtype).scatter_reduce(0, inv, w, "amax" if op == "max" else "amin",
                                                                     include_self=False)
    r, c = uniq // k, uniq % k
    keep = out.abs() > eps
    if remove_self_loops:
        keep = keep & (r != c)
    return torch.stack([r[keep], c[keep]]), out[keep]


# ------------------------------------------------------------------------------ fused coalesce route (r3)
def _coalesce_case(case):
    import random
    rng = random.Random(case)
    g = torch.Generator().manual_seed(case)
    n = rng.choice([3, 50, 700, 5_000, 40_000, 200_000])
    e = rng.choice([0, 1, 7, n, 4 * n, 12 * n, 40 * n if n <= 5_000 else 6 * n])
    shape = rng.choice(["pairs", "pairs", "random", "few_big", "many_empty", "triples"])
    if shape == "pairs":
        k = max(1, n // 2)
        cluster = (torch.randperm(n, generator=g) // 2).clamp(max=k - 1)
    elif shape == "triples":
        k = max(1, n // 3)
        cluster = (torch.randperm(n, generator=g) // 3).clamp(max=k - 1)
    elif shape == "random":
        k = max(1, rng.choice([n // 3, n // 10, n]))
        cluster = torch.randint(0, k, (n,), generator=g)
    elif shape == "few_big":
        k = max(1, min(n, rng.choice([2, 5, 40, 300])))
        cluster = torch.randint(0, k, (n,), generator=g)
    else:
        k = 2 * n + 5
        cluster = torch.randint(0, max(1, n // 4), (n,), generator=g) * 3
    ei = torch.randint(0, n, (2, e), generator=g)
    if e > 0 and rng.random() < 0.2:  # a hub node: one long supernode row
        ei[0, : e // 3] = int(torch.randint(0, n, (1,), generator=g))
    sorted_rows = rng.random() < 0.8 and e > 0
    if sorted_rows:
        ei = ei[:, torch.argsort(ei[0], stable=True)]
    ew = (torch.rand(e, generator=g) - 0.3) if rng.random() < 0.7 else None
    if ew is not None and e:
        ew[torch.rand(e, generator=g) < 0.1] = 0.0
    op = rng.choice(["sum", "sum", "mean", "min", "max", "mul"])
    return n, k, cluster, ei, ew, op, rng.random() < 0.5, sorted_rows


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


def test_eps_reaches_every_kernel_that_uses_it(dev, monkeypatch):
    """|w| > eps filters of the sparse Connect (ops.py:377), the degree clamp of both post-processings (ops.py:318,
    395) and the entropy term (losses.py:498) follow the patched module globals, against the oracle with the same eps."""
    import tgp_oracle as O
    from tgp.connect import SparseConnect
    from tgp.select import SelectOutput
    from tgp.utils import losses as losses_module
    from tgp.utils import ops as ops_module
    from tgp.utils.ops import postprocess_adj_pool_dense, postprocess_adj_pool_sparse
    g = torch.Generator().manual_seed(0)
    n, e = 60, 400
    ei = torch.randint(0, n, (2, e), generator=g)
    ew = torch.rand(e, generator=g)
    cl = torch.randint(0, 12, (n,), generator=g)
    so = SelectOutput(cluster_index=cl.to(dev), num_supernodes=12)
    monkeypatch.setattr(ops_module, "eps", 2.5)
    monkeypatch.setattr(O, "EPS", 2.5)
    out_ei, out_w = SparseConnect(degree_norm=True)(ei.to(dev), so, edge_weight=ew.to(dev))
    ref_ei, ref_w = O.sparse_connect(ei, ew, torch.arange(n), cl, n, 12, degree_norm=True)
    assert 0 < ref_ei.size(1) < 12 * 12 - 12  # the patched eps really dropped merged edges
    assert torch.equal(out_ei.cpu(), ref_ei)
    torch.testing.assert_close(out_w.cpu(), ref_w, rtol=1e-5, atol=1e-6)
    # plain filter pass on an arbitrary list
    f_ei, f_w = postprocess_adj_pool_sparse(ei.to(dev), (ew * 5).to(dev), n, remove_self_loops=True)
    keep = (ei[0] != ei[1]) & ((ew * 5).abs() > 2.5)
    assert torch.equal(f_ei.cpu(), ei[:, keep])
    # dense degree clamp: rows with tiny degree are clamped at eps
    for K in (8, 40, 100, 200):  # tiny / small / LDS / multi-kernel post-processing
        a = torch.rand(3, K, K, generator=g) * 0.05
        got = postprocess_adj_pool_dense(a.to(dev).clone(), True, True, True, False)
        ref = O.postprocess_dense(a.clone(), True, True, True, False)
        torch.testing.assert_close(got.cpu(), ref, rtol=1e-5, atol=1e-6)
    monkeypatch.setattr(losses_module, "eps", 0.5)
    s = torch.softmax(torch.randn(4, 50, 6, generator=g), -1)
    got = losses_module.entropy_loss(s.to(dev), num_nodes=200)
    ref = (-(s * torch.log(s + 0.5)).sum(-1)).sum() / 200
    torch.testing.assert_close(got.cpu(), ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("op", ["sum", "mean", "max", "min", "mul"])
def test_sparse_connect_coalesce_edge_weight_gradient(dev, op):
    from tgp.connect import SparseConnect
    from tgp.select import SelectOutput
    g = torch.Generator().manual_seed(5)
    n, e, k = 40, 300, 9
    ei = torch.randint(0, n, (2, e), generator=g)
    w = (torch.rand(e, generator=g) + 0.5)
    if op in ("max", "min"):  # ties share the gradient: make a few exact ties
        w[10:20] = w[0]
    cl = torch.randint(0, k, (n,), generator=g)
    so = SelectOutput(cluster_index=cl.to(dev), num_supernodes=k)
    wd = w.to(dev).requires_grad_(True)
    out_ei, out_w = SparseConnect(reduce_op=op)(ei.to(dev), so, edge_weight=wd)
    assert out_w.requires_grad
    wr = w.clone().requires_grad_(True)
    ref_ei, ref_w = _dense_ref_coalesce(ei, wr, cl, k, op, True)
    assert torch.equal(out_ei.cpu(), ref_ei)
    torch.testing.assert_close(out_w.detach().cpu(), ref_w.detach(), rtol=1e-5, atol=1e-6)
    coef = torch.randn(ref_w.numel(), generator=g)
    (out_w * coef.to(dev)).sum().backward()
    (ref_w * coef).sum().backward()
    torch.testing.assert_close(wd.grad.cpu(), wr.grad, rtol=1e-5, atol=1e-6)


def test_out_of_range_node_ids_are_refused_not_dereferenced(dev):
    """node ids in edge_index beyond num_nodes (and cluster ids beyond num_supernodes) used to index
    the kernels' tables unchecked.  Now every Connect path guards them and the call raises, as the reference's
    `cluster_index[edge_index]` / `subgraph` would."""
    from tgp import kernels as K
    from tgp.connect import SparseConnect
    from tgp.select import SelectOutput
    n = 50
    g = torch.Generator().manual_seed(0)
    ei = torch.randint(0, n, (2, 300), generator=g)
    bad = ei.clone()
    bad[1, 17] = n + 3
    neg = ei.clone()
    neg[0, 5] = -1
    w = torch.rand(300, generator=g)
    kept = torch.arange(0, n, 2)
    so_topk = SelectOutput(node_index=kept.to(dev), cluster_index=torch.arange(kept.numel(), device=dev), num_nodes=n,
                           num_supernodes=kept.numel())
    so_cl = SelectOutput(cluster_index=(torch.arange(n) // 2).to(dev), num_supernodes=n // 2)
    for e in (bad, neg):
        with pytest.raises(IndexError, match="node ids outside"):
            SparseConnect()(e.to(dev), so_topk, edge_weight=w.to(dev))
        with pytest.raises(IndexError, match="node ids outside"):
            SparseConnect()(e.to(dev), so_cl, edge_weight=w.to(dev))          # row-local attempt declines, general path reports
        with pytest.raises(IndexError, match="node ids outside"):
            srt = e[:, torch.argsort(e[0].clamp(min=0), stable=True)]
            SparseConnect()(srt.to(dev), so_cl, edge_weight=w.to(dev))
    # a large-K clustering takes the grouped path first: same refusal
    big_n = 140_000
    eb = torch.randint(0, big_n, (2, 1000), generator=g)
    eb[1, 3] = big_n + 1
    so_big = SelectOutput(cluster_index=torch.arange(big_n, device=dev), num_supernodes=big_n)
    with pytest.raises(IndexError, match="node ids outside"):
        SparseConnect()(eb.to(dev), so_big, edge_weight=None)
    # valid inputs still work afterwards (no sticky device state)
    out_ei, out_w = SparseConnect()(ei.to(dev), so_cl, edge_weight=w.to(dev))
    assert out_ei.size(1) > 0 and int(out_ei.max()) < n // 2
    with pytest.raises(IndexError):
        SelectOutput(cluster_index=torch.tensor([0, 3, 1], device=dev), num_supernodes=3)


@pytest.mark.parametrize("case", ["pairs", "upto8", "medium", "hub", "mostly_singletons_one_giant"])
def test_counting_assign_index_equals_stable_sort(dev, case):
    """The inverted assignment index (supernode -> its assignments, ascending) must be the stable sort by supernode
    whatever route builds it (reduce/base_reduce.py:146-153 reduces in that order): the counting route (few members
    per supernode) handles supernodes of 2, <= 8, <= 8192 and > 8192 members by four different code paths."""
    from tgp import kernels
    g = torch.Generator().manual_seed(11)
    n = 300_000
    if case == "pairs":
        k = n // 2
        cluster = torch.randperm(n, generator=g) // 2
    elif case == "upto8":
        k = n // 4
        cluster = torch.randint(0, k, (n,), generator=g)
    elif case == "medium":
        k = n // 3
        cluster = torch.randint(0, k, (n,), generator=g)
        cluster[torch.randperm(n, generator=g)[:5000]] = 17      # one supernode of ~5000 members (LDS bitonic)
        cluster[torch.randperm(n, generator=g)[:300]] = 4242     # and one of ~300
    elif case == "hub":
        k = n // 3
        cluster = torch.randint(0, k, (n,), generator=g)
        cluster[torch.randperm(n, generator=g)[:20000]] = 5      # > 8192 members: stable-compaction route
    else:
        k = n // 2
        cluster = torch.arange(n) % k
        cluster[: n // 4] = k - 1
    idx = kernels.build_assign_index(cluster.to(dev), k)
    order = torch.argsort(cluster, stable=True)
    counts = torch.bincount(cluster, minlength=k)
    row_ptr = torch.zeros(k + 1, dtype=torch.int64)
    row_ptr[1:] = torch.cumsum(counts, 0)
    assert torch.equal(idx._row_ptr.cpu().long(), row_ptr)
    assert torch.equal(idx.perm.cpu().long()[:n], order)


@pytest.mark.parametrize("block", range(6))
def test_fused_coalesce_route_equals_the_other_routes(dev, block):
    """The one-kernel row-local route (decoupled look-back, survivors at final offsets, long rows in-kernel) against
    the general radix route bit for bit -- edge_index AND weights (same reduction order) -- over random shapes: pair /
    triple / random / few-big / sparse-id clusterings, hub rows, unsorted lists (must decline), all reduce ops."""
    from tgp import kernels
    took = 0
    for case in range(block * 25, block * 25 + 25):
        n, k, cluster, ei, ew, op, rsl, sorted_rows = _coalesce_case(case)
        cl, eid, ewd = cluster.to(dev), ei.to(dev), None if ew is None else ew.to(dev)
        ref = kernels.coalesce_edges(eid, ewd, cl, k, op, rsl, route="general")
        try:
            got = kernels.coalesce_edges(eid, ewd, cl, k, op, rsl, route="fused")
        except RuntimeError as exc:
            assert "declined" in str(exc)
            continue
        took += 1
        assert sorted_rows or ei.size(1) <= 1 or bool((ei[0, 1:] >= ei[0, :-1]).all()), case
        assert torch.equal(got[0], ref[0]), case
        assert (got[1] is None and ref[1] is None) or torch.equal(got[1], ref[1]), case
        auto = kernels.coalesce_edges(eid, ewd, cl, k, op, rsl, assign_index=kernels.build_assign_index(cl, k))
        assert torch.equal(auto[0], ref[0]) and ((auto[1] is None) or torch.equal(auto[1], ref[1])), case
    assert took >= 5


def test_fused_coalesce_long_rows_and_many_tiles(dev):
    """Hub supernodes (rows of 65..1024 raw entries sorted by the whole workgroup), multi-pass tiles (32 rows of more
    than 1024 entries together) and a tile count far beyond the look-back window (64)."""
    from tgp import kernels
    g = torch.Generator().manual_seed(11)
    n = 60_000   # 30 000 supernode rows = 938 tiles: the most one launch takes (FZ_MAX_TILES = 1024)
    k = n // 2
    cluster = torch.randperm(n, generator=g) // 2
    deg = torch.randint(1, 40, (n,), generator=g)
    deg[torch.randint(0, n, (300,), generator=g)] = 400      # hubs: long rows
    row = torch.repeat_interleave(torch.arange(n), deg)
    col = torch.randint(0, n, (row.numel(),), generator=g)
    ei = torch.stack([row, col])
    ew = torch.rand(row.numel(), generator=g) + 0.1
    cl, eid, ewd = cluster.to(dev), ei.to(dev), ew.to(dev)
    ref = kernels.coalesce_edges(eid, ewd, cl, k, "sum", True, route="general")
    got = kernels.coalesce_edges(eid, ewd, cl, k, "sum", True, route="fused")
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    got = kernels.coalesce_edges(eid, None, cl, k, "sum", False, route="fused")
    ref = kernels.coalesce_edges(eid, None, cl, k, "sum", False, route="general")
    assert torch.equal(got[0], ref[0]) and got[1] is None


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


def test_edges_compact_entry_point(dev):
    """tgp_edges_compact through the C ABI: every alignment case (16-, 8-, 4-byte paths), fp32 and fp64 weights."""
    from tgp import _native as N
    L, st = N.lib(), N.stream_ptr(dev)
    g = torch.Generator().manual_seed(1)
    for cap, n, skew, wdt in [(1000, 777, 0, torch.float32), (1001, 1001, 1, torch.float64), (64, 1, 3, torch.float32),
                              (5000, 4096, 2, torch.float64)]:
        buf = torch.randint(0, 1 << 40, (2 * cap + skew,), generator=g).to(dev)
        row, col = buf[skew: skew + cap], buf[skew + cap: skew + 2 * cap]
        w = torch.rand(cap + skew, generator=g, dtype=wdt).to(dev)[skew:]
        eid = torch.arange(cap + skew, device=dev)[skew:]
        o_ei = torch.empty(2, n, dtype=torch.int64, device=dev)
        o_w = torch.empty(n, dtype=wdt, device=dev)
        o_id = torch.empty(n, dtype=torch.int64, device=dev)
        N.check(L.tgp_edges_compact(row.data_ptr(), col.data_ptr(), w.data_ptr(), w.element_size(), eid.data_ptr(), n,
                                    o_ei.data_ptr(), o_ei.data_ptr() + 8 * n, o_w.data_ptr(), o_id.data_ptr(), st), "compact")
        assert torch.equal(o_ei[0], row[:n]) and torch.equal(o_ei[1], col[:n])
        assert torch.equal(o_w, w[:n]) and torch.equal(o_id, eid[:n])


def test_float64_filter_edges_has_a_staged_fallback(dev, monkeypatch):
    """float64 edge weights had no count -> fill pair, so two refusals of the single-pass kernel (a look-back
    spin bound on a shared device) raised.  Now the staged route takes over: indices through the fp32 pair, the
    |w| > eps test on the gathered weights in double.  Forced here by making the single pass decline."""
    import tgp_oracle as O
    from tgp import kernels
    g = torch.Generator().manual_seed(3)
    n, e = 5000, 40000
    ei = torch.randint(0, n, (2, e), generator=g)
    ew = torch.rand(e, generator=g, dtype=torch.float64) - 0.3
    ew[::13] = 1e-9
    keep = torch.sort(torch.randperm(n, generator=g)[: n // 3])[0]
    want_ei, want_w = kernels.filter_edges(ei.to(dev), ew.to(dev), keep.to(dev), n, True)
    calls = {"n": 0}

    def declines(*a, **k):
        calls["n"] += 1
        return None
    monkeypatch.setattr(kernels, "_filter_edges_single", declines)
    got_ei, got_w, got_id = kernels.filter_edges(ei.to(dev), ew.to(dev), keep.to(dev), n, True, want_edge_id=True)
    assert calls["n"] >= 2 and got_w.dtype == torch.float64
    assert torch.equal(got_ei, want_ei) and torch.equal(got_w, want_w)
    assert torch.equal(ew.to(dev)[got_id], got_w)
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        r_ei, r_w = O.sparse_connect(ei, ew, keep, None, n, keep.numel())
    finally:
        torch.set_default_dtype(old)
    assert torch.equal(got_ei.cpu(), r_ei) and torch.equal(got_w.cpu(), r_w)


# ------------------------------------------------------------------ r5: float64 edge weights on the row-local coalesce
@pytest.mark.gpu
@pytest.mark.parametrize("op", ["sum", "mean", "min", "max", "mul"])
def test_float64_edge_weights_take_the_row_local_coalesce(dev, op):
    """connect/base_conn.py:86-89 with a double weight tensor: the row-sorted list now runs the sort-free row-local
    pipeline in double (it took the device-wide sort before).  Rows of every length class (<= 32, 33..64, 65..1024 raw
    entries), duplicates, self loops, sub-eps weights: indices equal the general float64 route's and the oracle's,
    weights to 1e-13; a list with a hub row (> 1024 raw entries) falls back to the general route by itself."""
    import tgp_oracle as O
    from tgp import kernels as K_
    g = torch.Generator().manual_seed(17)
    n, k = 6000, 1500
    cl = torch.randint(0, k, (n,), generator=g)
    cl[:k] = torch.arange(k)
    deg = torch.randint(1, 9, (n,), generator=g)
    members_of_7 = (cl == 7).nonzero().flatten()
    deg[members_of_7[:3]] = 120          # supernode row 7: a few hundred raw entries -> the long-row kernel
    members_of_9 = (cl == 9).nonzero().flatten()
    deg[members_of_9[:1]] = 50           # supernode row 9: 33..64 entries
    row = torch.repeat_interleave(torch.arange(n), deg)
    col = torch.randint(0, n, (row.numel(),), generator=g)
    col[::11] = row[::11]                # self loops
    ei = torch.stack([row, col])
    ew = torch.rand(row.numel(), generator=g, dtype=torch.float64) - 0.3
    ew[torch.rand(row.numel(), generator=g) < 0.05] = 1e-9
    eid, ewd, cld = ei.to(dev), ew.to(dev), cl.to(dev)
    ai = K_.build_assign_index(cld, k)
    for rsl in (True, False):
        r_ei, r_ew = _oracle64(O.sparse_connect, ei, ew, torch.arange(n), cl, n, k, reduce_op=op, remove_self_loops=rsl)
        got_rows = K_.coalesce_edges(eid, ewd, cld, k, op, rsl, assign_index=ai, route="rows")
        got_gen = K_.coalesce_edges(eid, ewd, cld, k, op, rsl, route="general")
        got_auto = K_.coalesce_edges(eid, ewd, cld, k, op, rsl, assign_index=ai)
        for got in (got_rows, got_gen, got_auto):
            assert got[1].dtype == torch.float64 and torch.equal(got[0].cpu(), r_ei)
            torch.testing.assert_close(got[1].cpu(), r_ew, rtol=1e-13, atol=1e-13)
        assert torch.equal(got_rows[1], got_auto[1])
    # which route the automatic choice took
    calls = []
    L = K_.N.lib()
    real = L.tgp_connect_coalesce_rows_count_published_f64

    class Spy:
        def __getattr__(self, name):
            if name == "tgp_connect_coalesce_rows_count_published_f64":
                return lambda *a: (calls.append(1), real(*a))[1]
            return getattr(L, name)
    old = K_.N.lib
    K_.N.lib = lambda: Spy()
    try:
        K_.coalesce_edges(eid, ewd, cld, k, op, True, assign_index=ai)
    finally:
        K_.N.lib = old
    assert calls == [1]
    # a hub row: the row-local count answers -5 (float64 has no hub kernels), the general route takes the call
    deg2 = deg.clone()
    deg2[members_of_7[:3]] = 600
    row2 = torch.repeat_interleave(torch.arange(n), deg2)
    col2 = torch.randint(0, n, (row2.numel(),), generator=g)
    ew2 = torch.rand(row2.numel(), generator=g, dtype=torch.float64)
    ei2 = torch.stack([row2, col2])
    r_ei, r_ew = _oracle64(O.sparse_connect, ei2, ew2, torch.arange(n), cl, n, k, reduce_op=op, remove_self_loops=True)
    got = K_.coalesce_edges(ei2.to(dev), ew2.to(dev), cld, k, op, True, assign_index=ai)
    assert torch.equal(got[0].cpu(), r_ei)
    torch.testing.assert_close(got[1].cpu(), r_ew, rtol=1e-13, atol=1e-13)
    with pytest.raises(RuntimeError):
        K_.coalesce_edges(ei2.to(dev), ew2.to(dev), cld, k, op, True, assign_index=ai, route="rows")
