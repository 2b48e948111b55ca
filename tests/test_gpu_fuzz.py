"""Seeded randomised differential tests: every native path against the CPU oracle on many small random shapes
(sizes, alignments, flag combinations, duplicate / self-loop / isolated-node patterns, sorted and unsorted
edge lists).  Indices must be bit-exact, fp32 within rtol = atol = 1e-5 (north_star)."""
import itertools
import random

import pytest
import torch

import tgp_oracle as O

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-5, atol=1e-5)


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def rand_graph(rng, g, n, e, sort_rows, with_loops, with_dups):
    if e == 0 or n == 0:
        return torch.zeros(2, 0, dtype=torch.long)
    r = torch.randint(0, n, (e,), generator=g)
    c = torch.randint(0, n, (e,), generator=g)
    if not with_loops:
        c = torch.where(r == c, (c + 1) % n, c)
    ei = torch.stack([r, c])
    if with_dups and e > 4:
        ei = torch.cat([ei, ei[:, : e // 3]], 1)  # repeated directed entries
    if sort_rows:  # stable sort by row only: columns stay in insertion order (what PyG loaders hand out)
        ei = ei[:, torch.sort(ei[0], stable=True)[1]]
    return ei


@pytest.mark.parametrize("seed", range(24))
def test_sparse_connect_fuzz(dev, seed):
    from tgp.connect import sparse_connect
    rng = random.Random(seed)
    g = torch.Generator().manual_seed(1000 + seed)
    n = rng.choice([1, 2, 7, 33, 150, 1000, 5000])
    e = rng.choice([0, 1, 5, 40, 700, 20_000])
    ei = rand_graph(rng, g, n, e, sort_rows=rng.random() < 0.6, with_loops=rng.random() < 0.5, with_dups=rng.random() < 0.5)
    weighted = rng.random() < 0.7
    ew = None
    if weighted:
        ew = torch.rand(ei.size(1), generator=g) + 0.05
        ew[torch.rand(ei.size(1), generator=g) < 0.1] = 0.0  # eps filter
    op = rng.choice(["sum", "mean", "min", "max", "mul"])
    flags = dict(remove_self_loops=rng.random() < 0.5, degree_norm=rng.random() < 0.4, edge_weight_norm=rng.random() < 0.4)
    # coalesce branch: every node in some supernode; clusters of very different sizes
    k = max(1, rng.choice([1, 2, n // 3 + 1, n]))
    cluster = torch.randint(0, k, (n,), generator=g)
    if rng.random() < 0.5:  # one hub supernode
        cluster[torch.rand(n, generator=g) < 0.3] = 0
    batch_pooled = torch.sort(torch.randint(0, 3, (k,), generator=g))[0]
    ref_ei, ref_ew = O.sparse_connect(ei, ew, torch.arange(n), cluster, n, k, reduce_op=op, batch_pooled=batch_pooled, **flags)
    got_ei, got_ew = sparse_connect(ei.to(dev), None if ew is None else ew.to(dev), node_index=torch.arange(n, device=dev),
                                    cluster_index=cluster.to(dev), num_nodes=n, num_supernodes=k, reduce_op=op,
                                    batch_pooled=batch_pooled.to(dev), **flags)
    assert torch.equal(got_ei.cpu(), ref_ei), (seed, "coalesce", op, flags)
    if ref_ew is None:
        assert got_ew is None
    else:
        torch.testing.assert_close(got_ew.cpu(), ref_ew, **TOL)
    # subgraph branch: a strict subset of the nodes is kept
    if n >= 2:
        kept = torch.sort(torch.randperm(n, generator=g)[: max(1, n // 2)])[0]
        if seed % 4 == 1:  # a subset in arbitrary order: new ids are positions in node_index, not ranks
            kept = kept[torch.randperm(kept.numel(), generator=g)]
        kk = kept.numel()
        bp = torch.sort(torch.randint(0, 3, (kk,), generator=g))[0]
        ref_ei, ref_ew = O.sparse_connect(ei, ew, kept, torch.arange(kk), n, kk, batch_pooled=bp, **flags)
        got_ei, got_ew = sparse_connect(ei.to(dev), None if ew is None else ew.to(dev), node_index=kept.to(dev),
                                        cluster_index=torch.arange(kk, device=dev), num_nodes=n, num_supernodes=kk,
                                        batch_pooled=bp.to(dev), **flags)
        assert torch.equal(got_ei.cpu(), ref_ei), (seed, "subgraph", flags)
        if ref_ew is None:
            assert got_ew is None
        else:
            torch.testing.assert_close(got_ew.cpu(), ref_ew, **TOL)


@pytest.mark.parametrize("seed", range(16))
def test_sparse_reduce_fuzz(dev, seed):
    from tgp.reduce import BaseReduce
    from tgp.select import SelectOutput
    rng = random.Random(seed)
    g = torch.Generator().manual_seed(2000 + seed)
    n = rng.choice([1, 3, 64, 500, 4000])
    f = rng.choice([1, 3, 4, 16, 33, 128, 200])
    x = torch.randn(n, f, generator=g)
    if rng.random() < 0.5:   # one-over-K style: every node assigned, clusters of any size
        k = max(1, rng.choice([1, n // 4 + 1, n]))
        ci = torch.randint(0, k, (n,), generator=g)
        ni = torch.arange(n)
    else:                    # TopK style: a subset of nodes, one per supernode, arbitrary supernode order
        k = max(1, n // 2)
        ni = torch.sort(torch.randperm(n, generator=g)[:k])[0]
        ci = torch.randperm(k, generator=g)
    w = torch.randn(ni.numel(), generator=g)
    # graph ids: members of one supernode share a graph (scatter_ with disagreeing duplicates is order-dependent
    # in torch itself), so graphs are assigned per supernode and handed down to the nodes
    graph_of_cluster = torch.randint(0, 3, (k,), generator=g)
    batch = torch.zeros(n, dtype=torch.long)
    batch[ni] = graph_of_cluster[ci]
    so = SelectOutput(node_index=ni.to(dev), cluster_index=ci.to(dev), num_nodes=n, num_supernodes=k, weight=w.to(dev))
    xp, bp = BaseReduce()(x.to(dev), so, batch=batch.to(dev))
    # ordering of the COO entries inside SelectOutput is node-sorted; the oracle takes the same views
    ref = O.reduce_sparse(x, so.node_index.cpu(), so.cluster_index.cpu(), so.weight.cpu(), k)
    assert torch.equal(xp.cpu(), ref), seed  # same summation order as the sequential CPU scatter: bit-identical
    assert torch.equal(bp.cpu(), O.reduce_batch_sparse(batch, so.node_index.cpu(), so.cluster_index.cpu(), k))


@pytest.mark.parametrize("seed", range(20))
def test_dense_pool_fuzz(dev, seed):
    from tgp.connect import DenseConnect
    from tgp.reduce import BaseReduce
    from tgp.select import SelectOutput
    rng = random.Random(seed)
    g = torch.Generator().manual_seed(3000 + seed)
    B = rng.choice([1, 2, 5, 70])
    N = rng.choice([1, 5, 31, 60, 64, 100, 257]) if B < 70 else rng.choice([8, 40, 60, 64])
    K = rng.choice([1, 3, 8, 20, 32, 33, 70])
    F = rng.choice([1, 4, 7, 32, 65])
    adj = (torch.rand(B, N, N, generator=g) < 0.3).float() * torch.rand(B, N, N, generator=g)
    x = torch.randn(B, N, F, generator=g)
    s = torch.softmax(torch.randn(B, N, K, generator=g), -1)
    if rng.random() < 0.5:  # padded graphs
        nb = torch.randint(1, N + 1, (B,), generator=g)
        mask = torch.arange(N).unsqueeze(0) < nb.unsqueeze(1)
        s, x = s * mask.unsqueeze(-1), x * mask.unsqueeze(-1)
        adj = adj * mask.unsqueeze(1) * mask.unsqueeze(2)
    flags = [rng.random() < 0.5 for _ in range(4)]  # remove_self_loops, degree_norm, adj_transpose, edge_weight_norm
    so = SelectOutput(s=s.to(dev))
    xp, _ = BaseReduce()(x.to(dev), so)
    torch.testing.assert_close(xp.cpu(), O.reduce_dense(s, x), **TOL)
    conn = DenseConnect(*flags)
    raw_ref = O.dense_connect(s, adj)
    torch.testing.assert_close(conn.dense_connect(adj=adj.to(dev), s=s.to(dev)).cpu(), raw_ref, **TOL)
    out, _ = conn(adj.to(dev), so)
    torch.testing.assert_close(out.cpu(), O.postprocess_dense(raw_ref.clone(), *flags), **TOL)


@pytest.mark.parametrize("seed", range(10))
def test_densify_and_block_diag_fuzz(dev, seed):
    from tgp.src import to_dense_adj, to_dense_batch
    from tgp.utils.ops import dense_to_block_diag
    rng = random.Random(seed)
    g = torch.Generator().manual_seed(4000 + seed)
    sizes = [rng.choice([1, 2, 9, 30]) for _ in range(rng.choice([1, 2, 5]))]
    n = sum(sizes)
    batch = torch.cat([torch.full((m,), i, dtype=torch.long) for i, m in enumerate(sizes)])
    off = 0
    eis = []
    for m in sizes:
        e = rng.choice([0, 3, 40])
        eis.append(rand_graph(rng, g, m, e, sort_rows=False, with_loops=True, with_dups=True) + off)
        off += m
    ei = torch.cat(eis, 1)
    ew = torch.rand(ei.size(1), generator=g)
    x = torch.randn(n, 5, generator=g)
    xd_ref, mask_ref = O.to_dense_batch(x, batch)
    xd, mask = to_dense_batch(x.to(dev), batch.to(dev))
    assert torch.equal(xd.cpu(), xd_ref) and torch.equal(mask.cpu(), mask_ref)
    ad_ref = O.to_dense_adj(ei, ew, batch)
    torch.testing.assert_close(to_dense_adj(ei.to(dev), batch.to(dev), ew.to(dev)).cpu(), ad_ref, **TOL)
    k = rng.choice([1, 4, 9])
    pooled = (torch.rand(len(sizes), k, k, generator=g) < 0.5).float() * torch.randn(len(sizes), k, k, generator=g)
    ref_ei, ref_ew = O.dense_to_block_diag(pooled)
    got_ei, got_ew = dense_to_block_diag(pooled.to(dev))
    assert torch.equal(got_ei.cpu(), ref_ei) and torch.equal(got_ew.cpu(), ref_ew)


def rand_batch(rng, g, feat):
    """A PyG-style batch of small random undirected graphs (row-sorted, no duplicates), positive weights."""
    sizes = [rng.choice([3, 7, 12, 25]) for _ in range(rng.choice([1, 2, 4]))]
    xs, eis, ews, bs, off = [], [], [], [], 0
    for gi, m in enumerate(sizes):
        a = torch.triu(torch.rand(m, m, generator=g) < rng.choice([0.15, 0.4, 0.8]), 1)
        a = a | a.t()
        ei = a.nonzero().t() + off
        w = torch.rand(m, m, generator=g)
        w = (w + w.t())[a] + 0.1
        eis.append(ei)
        ews.append(w)
        xs.append(torch.randn(m, feat, generator=g))
        bs.append(torch.full((m,), gi, dtype=torch.long))
        off += m
    return torch.cat(xs), torch.cat(eis, 1), torch.cat(ews), torch.cat(bs)


@pytest.mark.parametrize("seed", range(12))
def test_topk_and_cluster_poolers_end_to_end_fuzz(dev, seed):
    """get_pooler("topk") and the Graclus-style reduce/connect pipeline on random batches == oracle pooler functions
    (poolers/topk.py:120-190, poolers/graclus.py:91-156) including batch vectors and every Connect flag."""
    from tgp.connect import SparseConnect
    from tgp.poolers import get_pooler
    from tgp.reduce import BaseReduce
    from tgp.select import SelectOutput
    rng = random.Random(seed)
    g = torch.Generator().manual_seed(5000 + seed)
    feat = rng.choice([1, 4, 9])
    x, ei, ew, batch = rand_batch(rng, g, feat)
    use_w = rng.random() < 0.7
    cfg = dict(ratio=rng.choice([0.3, 0.5, 0.9, 2]), remove_self_loops=rng.random() < 0.7,
               degree_norm=rng.random() < 0.4, edge_weight_norm=rng.random() < 0.4,
               connect_red_op=rng.choice(["sum", "mean", "max"]))
    torch.manual_seed(seed)
    pooler = get_pooler("topk", in_channels=feat, **cfg).to(dev).eval()
    p = pooler.selector.weight.detach().cpu() if feat > 1 else None
    with torch.no_grad():
        out = pooler(x=x.to(dev), adj=ei.to(dev), edge_weight=ew.to(dev) if use_w else None, batch=batch.to(dev))
    ref = O.topk_pool(x, ei, ew if use_w else None, batch, p, ratio=cfg["ratio"], remove_self_loops=cfg["remove_self_loops"],
                      degree_norm=cfg["degree_norm"], edge_weight_norm=cfg["edge_weight_norm"], reduce_op=cfg["connect_red_op"])
    assert torch.equal(out.so.node_index.cpu(), ref["node_index"]) and torch.equal(out.so.cluster_index.cpu(), ref["cluster_index"])
    torch.testing.assert_close(out.x.cpu(), ref["x"], **TOL)
    assert torch.equal(out.edge_index.cpu(), ref["edge_index"]) and torch.equal(out.batch.cpu(), ref["batch"])
    if ref["edge_weight"] is None:
        assert out.edge_weight is None
    else:
        torch.testing.assert_close(out.edge_weight.cpu(), ref["edge_weight"], **TOL)
    # cluster pipeline: the oracle's greedy matching as the selector output
    cluster = O.greedy_matching(ei, ew if use_w else None, x.size(0))
    k = int(cluster.max()) + 1
    so = SelectOutput(cluster_index=cluster.to(dev), num_nodes=x.size(0), num_supernodes=k)
    xp, bp = BaseReduce()(x.to(dev), so, batch=batch.to(dev))
    conn = SparseConnect(reduce_op=cfg["connect_red_op"], remove_self_loops=cfg["remove_self_loops"],
                         degree_norm=cfg["degree_norm"], edge_weight_norm=cfg["edge_weight_norm"])
    ce, cw = conn(ei.to(dev), so, edge_weight=ew.to(dev) if use_w else None, batch_pooled=bp)
    refc = O.cluster_pool(x, ei, ew if use_w else None, batch, cluster, k, reduce_op=cfg["connect_red_op"],
                          remove_self_loops=cfg["remove_self_loops"], degree_norm=cfg["degree_norm"],
                          edge_weight_norm=cfg["edge_weight_norm"])
    assert torch.equal(xp.cpu(), refc["x"]) and torch.equal(bp.cpu(), refc["batch"]) and torch.equal(ce.cpu(), refc["edge_index"])
    if refc["edge_weight"] is None:
        assert cw is None
    else:
        torch.testing.assert_close(cw.cpu(), refc["edge_weight"], **TOL)


@pytest.mark.parametrize("seed", range(12))
def test_dense_poolers_end_to_end_fuzz(dev, seed):
    """get_pooler("diff" | "mincut" [+ "_u"]) from sparse inputs on random batches == the oracle's dense_pool
    (poolers/diffpool.py:145-260, poolers/mincut.py:150-289): assignments, pooled features, adjacency in every
    output format, losses."""
    from tgp.poolers import get_pooler
    rng = random.Random(seed)
    g = torch.Generator().manual_seed(6000 + seed)
    feat, k = rng.choice([3, 8]), rng.choice([2, 5])
    x, ei, ew, batch = rand_batch(rng, g, feat)
    alias = rng.choice(["diff", "mincut"])
    batched = rng.random() < 0.5
    cfg = dict(remove_self_loops=rng.random() < 0.7, degree_norm=rng.random() < 0.6, edge_weight_norm=rng.random() < 0.4,
               adj_transpose=rng.random() < 0.6, sparse_output=rng.random() < 0.5)
    if alias == "diff":
        cfg["normalize_loss"] = rng.random() < 0.5
    torch.manual_seed(seed)
    pooler = get_pooler(alias if batched else alias + "_u", in_channels=feat, k=k, **cfg).to(dev).eval()
    sd = pooler.state_dict()
    weights = [sd[n].cpu() for n in sd if n.endswith("weight")]
    biases = [sd[n].cpu() for n in sd if n.endswith("bias")]
    with torch.no_grad():
        out = pooler(x=x.to(dev), adj=ei.to(dev), edge_weight=ew.to(dev), batch=batch.to(dev))
    ref = O.dense_pool(alias, x, ei, ew, batch, weights, biases, batched=batched,
                       remove_self_loops=cfg["remove_self_loops"], degree_norm=cfg["degree_norm"],
                       edge_weight_norm=cfg["edge_weight_norm"], adj_transpose=cfg["adj_transpose"],
                       sparse_output=cfg["sparse_output"], normalize_loss=cfg.get("normalize_loss", False))
    torch.testing.assert_close(out.so.s.cpu(), ref["s"], **TOL)
    torch.testing.assert_close(out.x.cpu(), ref["x"], **TOL)
    if ref["edge_weight"] is not None:   # [2, E'] output: same edges, same order
        assert torch.equal(out.edge_index.cpu(), ref["edge_index"]), (alias, batched, cfg)
        torch.testing.assert_close(out.edge_weight.cpu(), ref["edge_weight"], **TOL)
    else:
        torch.testing.assert_close(out.edge_index.cpu(), ref["edge_index"], **TOL)
    if ref["batch"] is None:
        assert out.batch is None
    else:
        assert torch.equal(out.batch.cpu(), ref["batch"])
    assert set(out.loss) == set(ref["loss"])
    for key in ref["loss"]:
        torch.testing.assert_close(out.loss[key].cpu(), ref["loss"][key], **TOL)


@pytest.mark.parametrize("seed", range(6))
def test_grouped_coalesce_path_vs_oracle(dev, seed):
    """Unsorted edge lists with more than 65536 supernodes take the two-level path (radix sort by supernode row +
    in-row LDS sort); a hub supernode whose row exceeds the LDS sort makes it decline to the general path.  Both
    must equal the oracle (connect/base_conn.py:83-89 + utils/ops.py:338-419)."""
    from tgp.connect import sparse_connect
    rng = random.Random(seed)
    g = torch.Generator().manual_seed(7000 + seed)
    n, k = 180_000, rng.choice([70_000, 100_000, 180_000])
    e = 300_000
    ei = rand_graph(rng, g, n, e, sort_rows=False, with_loops=True, with_dups=True)
    ew = (torch.rand(ei.size(1), generator=g) + 0.05) if rng.random() < 0.7 else None
    if ew is not None:
        ew[torch.rand(ei.size(1), generator=g) < 0.05] = 0.0
    cluster = torch.randint(0, k, (n,), generator=g)
    if seed % 3 == 2:  # hub: one supernode row with far more than 1024 raw entries
        cluster[: n // 20] = 7
    op = rng.choice(["sum", "mean", "min", "max", "mul"])
    flags = dict(remove_self_loops=rng.random() < 0.5, degree_norm=rng.random() < 0.3)
    ref_ei, ref_ew = O.sparse_connect(ei, ew, torch.arange(n), cluster, n, k, reduce_op=op, **flags)
    got_ei, got_ew = sparse_connect(ei.to(dev), None if ew is None else ew.to(dev), node_index=torch.arange(n, device=dev),
                                    cluster_index=cluster.to(dev), num_nodes=n, num_supernodes=k, reduce_op=op, **flags)
    assert torch.equal(got_ei.cpu(), ref_ei), (seed, op, flags)
    if ref_ew is None:
        assert got_ew is None
    else:
        torch.testing.assert_close(got_ew.cpu(), ref_ew, **TOL)


@pytest.mark.parametrize("seed", range(8))
def test_sparse_norm_long_rows_vs_oracle(dev, seed):
    """A6 on lists whose rows straddle, fill and end exactly on the 1024-edge slabs of the segmented degree
    kernel (utils/ops.py:383-417); sorted and unsorted, several graphs, signed weights."""
    from tgp.utils.ops import postprocess_adj_pool_sparse
    rng = random.Random(seed)
    g = torch.Generator().manual_seed(4000 + seed)
    lens = []
    for _ in range(rng.choice([3, 12, 40])):
        lens.append(rng.choice([0, 1, 2, 63, 64, 65, 500, 1023, 1024, 1025, 2048, 3000, 5000]))
    if seed == 0:
        lens = [1024, 1024, 2048, 1, 1023]  # run ends exactly on slab edges
    if seed == 1:
        lens = [7000]  # a single row: every slab but the first is "through"
    n = len(lens)
    row = torch.repeat_interleave(torch.arange(n), torch.tensor(lens))
    E = row.numel()
    col = torch.randint(0, n, (E,), generator=g)
    ew = torch.rand(E, generator=g) + 0.05
    if seed % 2:
        ew = ew * torch.where(torch.rand(E, generator=g) < 0.3, -1.0, 1.0)
    if seed % 4 == 3:  # unsorted list -> atomic fallback
        p = torch.randperm(E, generator=g)
        row, col, ew = row[p], col[p], ew[p]
    ei = torch.stack([row, col])
    bp = torch.sort(torch.randint(0, 3, (n,), generator=g))[0]
    for dn, wn in ((True, False), (False, True), (True, True)):
        ref_ei, ref_ew = O.postprocess_sparse(ei, ew, n, degree_norm=dn, edge_weight_norm=wn, batch_pooled=bp)
        got_ei, got_ew = postprocess_adj_pool_sparse(ei.to(dev), ew.to(dev), n, degree_norm=dn, edge_weight_norm=wn,
                                                     batch_pooled=bp.to(dev))
        assert torch.equal(got_ei.cpu(), ref_ei)
        # signed weights can cancel inside a degree sum; the clamp at eps then amplifies rounding, so
        # compare the positive-weight cases tightly and the signed ones on the well-conditioned rows only
        if seed % 2 and dn:
            deg = torch.zeros(n, dtype=torch.float64).index_add_(0, ref_ei[0], ew.double()[ew.abs() > 1e-12])
            absdeg = torch.zeros(n, dtype=torch.float64).index_add_(0, ref_ei[0], ew.double().abs()[ew.abs() > 1e-12])
            ok = (deg.abs() > 1e-2 * absdeg)
            sel = ok[ref_ei[0]] & ok[ref_ei[1]]
            if wn:
                continue
            torch.testing.assert_close(got_ew.cpu()[sel], ref_ew[sel], rtol=1e-4, atol=1e-5)
        else:
            torch.testing.assert_close(got_ew.cpu(), ref_ew, **TOL)


@pytest.mark.parametrize("seed", range(16))
def test_native_topk_select_vs_oracle(dev, seed):
    """A12 ratio mode (select/topk_select.py:163-203): the radix-sort selection against the oracle's two-sort
    restatement of PyG's topk + the row sort of SelectOutput; ragged graphs, empty graph ids, ties, +-0, int ratio."""
    from tgp.select import TopkSelect
    rng = random.Random(seed)
    g = torch.Generator().manual_seed(7000 + seed)
    nb = rng.choice([1, 2, 5, 40, 80, 300])  # >= 64 graphs: one workgroup (or wave, if all <= 64 nodes) per graph
    pool = [0, 1, 2, 3, 17, 64] if seed % 4 == 2 else [0, 1, 2, 3, 17, 100, 700, 2500 if nb < 64 else 2048]
    sizes = [rng.choice(pool) for _ in range(nb)]
    if sum(sizes) == 0:
        sizes[0] = 5
    batch = torch.repeat_interleave(torch.arange(nb), torch.tensor(sizes))
    n = batch.numel()
    score = torch.randn(n, generator=g)
    if seed % 3 == 0:  # heavy ties (a stable descending sort keeps the lower node id first)
        score = torch.round(score * 2) / 2
        score[score == 0] = torch.where(torch.rand(int((score == 0).sum()), generator=g) < 0.5, 0.0, -0.0)
    ratio = rng.choice([0.5, 0.3, 0.999, 1, 3, 7])
    sel = TopkSelect(in_channels=None, ratio=ratio, act="linear").to(dev)
    if seed % 5 == 4 and n > 1:  # unsorted batch vector: graphs are not contiguous segments (radix path)
        perm = torch.randperm(n, generator=g)
        batch, score = batch[perm], score[perm]
    use_batch = None if nb == 1 and seed % 2 else batch.to(dev)
    so = sel(score.view(-1, 1).to(dev), batch=use_batch)
    if seed % 3 == 0:
        # PyG sorts with stable=False, so the order inside a tie is unspecified by the reference (torch's CPU sort
        # is not stable beyond a few hundred elements); the native kernel defines it as "lower node id first",
        # i.e. the oracle's arithmetic with stable sorts
        sizes_t = torch.tensor(sizes)
        kk = (torch.minimum(torch.full_like(sizes_t, int(ratio)), sizes_t) if ratio >= 1
              else (float(ratio) * sizes_t.to(torch.float32)).ceil().long())
        order = torch.sort(score, descending=True, stable=True)[1]
        b_sorted, b_perm = torch.sort(batch[order], stable=True)
        ptr = torch.cat([sizes_t.new_zeros(1), sizes_t.cumsum(0)])
        ref_perm = order[b_perm[(torch.arange(n) - ptr[b_sorted]) < kk[b_sorted]]]
    else:
        ref_perm = O.topk_perm(score, ratio, batch)           # graph-major, score-descending node list
    k = ref_perm.numel()
    ref_node, order = torch.sort(ref_perm)                     # select/base_select.py:58
    ref_cluster = torch.arange(k)[order]
    assert so.num_nodes == n and so.num_supernodes == k
    assert torch.equal(so.node_index.cpu(), ref_node)
    assert torch.equal(so.cluster_index.cpu(), ref_cluster)
    torch.testing.assert_close(so.weight.cpu(), score[ref_node], rtol=0, atol=0)
    ai = so.assign_index()                                     # supernode -> position of its assignment
    assert torch.equal(ai.perm[:k].cpu().long(), torch.argsort(ref_cluster))


@pytest.mark.parametrize("seed", range(8))
def test_native_graclus_matching_contract(dev, seed):
    """A14 (select/graclus_select.py:62-81): the native handshake matching must be a valid MAXIMAL matching over
    the non-loop edges, label = smaller id of the pair, consecutive cluster ids in representative order; on
    distinct weights it must equal the sequential greedy heavy-edge matching (unique answer)."""
    from tgp import kernels as KK
    from tgp.select import GraclusSelect
    rng = random.Random(seed)
    g = torch.Generator().manual_seed(9000 + seed)
    n = rng.choice([1, 2, 9, 200, 5000])
    e = rng.choice([0, 3, 50, 4 * n])
    r = torch.randint(0, n, (e,), generator=g)
    c = torch.randint(0, n, (e,), generator=g)
    if seed == 0:  # a path: smallest-id tie-breaking would need n/2 rounds here
        n = 4000
        r, c = torch.arange(n - 1), torch.arange(1, n)
    ei = torch.cat([torch.stack([r, c]), torch.stack([c, r])], 1)  # symmetric, maybe with loops / duplicates
    und_w = torch.rand(r.numel(), generator=g) + 0.1 if seed % 2 else None
    w = None if und_w is None else torch.cat([und_w, und_w])
    label = KK.graclus_match(ei.to(dev), None if w is None else w.to(dev), n).cpu()
    idx = torch.arange(n)
    partner_count = torch.bincount(label, minlength=n)
    assert partner_count.max() <= 2 and torch.all(label <= idx)
    assert torch.equal(label[label], label)                      # representatives label themselves
    matched = partner_count[label] == 2
    nl = ei[:, ei[0] != ei[1]]
    pairs = set(map(tuple, nl.t().tolist()))
    for i in (matched & (label != idx)).nonzero().view(-1).tolist():
        assert (i, int(label[i])) in pairs                        # partners are adjacent
    both_free = ~matched[nl[0]] & ~matched[nl[1]]
    assert not bool(both_free.any())                              # maximal: no edge between two free nodes
    if w is not None and n <= 200 and nl.size(1):                 # distinct weights: unique greedy answer
        order = torch.argsort(w[ei[0] != ei[1]], descending=True)
        ref = idx.clone()
        free = torch.ones(n, dtype=torch.bool)
        for k in order.tolist():
            a, b = int(nl[0, k]), int(nl[1, k])
            if free[a] and free[b]:
                free[a] = free[b] = False
                ref[a] = ref[b] = min(a, b)
        assert torch.equal(label, ref)
    so = GraclusSelect()(ei.to(dev), None if w is None else w.to(dev), num_nodes=n)
    ids = torch.unique(label)
    assert so.num_supernodes == ids.numel()
    assert torch.equal(so.cluster_index.cpu(), torch.searchsorted(ids, label))
    assert torch.equal(so.node_index.cpu(), idx)


def test_native_graclus_matching_monotone_path_is_maximal(dev):
    """a path with strictly increasing weights matches ONE pair per handshake round (n/2 rounds); the
    matching must still come out maximal -- here the unique greedy answer, a perfect matching -- not stop at a cap."""
    from tgp import kernels as KK
    from tgp.select import GraclusSelect, graclus_cluster
    n = 3000
    r, c = torch.arange(n - 1), torch.arange(1, n)
    w = torch.arange(1, n, dtype=torch.float32)
    ei = torch.cat([torch.stack([r, c]), torch.stack([c, r])], 1)
    ww = torch.cat([w, w])
    label = KK.graclus_match(ei.to(dev), ww.to(dev), n).cpu()
    ref = torch.arange(n)
    ref[1::2] = ref[0::2]  # pairs (0,1), (2,3), ...: the heaviest edge (n-2, n-1) first, then every second one
    assert torch.equal(label, ref)
    assert GraclusSelect()(ei.to(dev), ww.to(dev), num_nodes=n).num_supernodes == n // 2
    with pytest.warns(RuntimeWarning, match="not maximal"):
        capped = KK.graclus_match(ei.to(dev), ww.to(dev), n, max_rounds=10).cpu()
    assert int((capped != torch.arange(n)).sum()) < n // 2
    # the host-tensor stand-in has the same contract
    m = 400
    rm, cm = torch.arange(m - 1), torch.arange(1, m)
    em = torch.cat([torch.stack([rm, cm]), torch.stack([cm, rm])], 1)
    wm = torch.arange(1, m, dtype=torch.float32).repeat(2)
    assert torch.equal(graclus_cluster(em[0], em[1], wm, m), ref[:m])


@pytest.mark.parametrize("seed", range(4))
def test_native_graclus_matching_directed_and_asymmetric_input(dev, seed):
    """Edge lists that are not the symmetric, symmetrically weighted lists Graclus is defined on: entries without a
    reverse are ignored, a pair's weight is the larger of its two directions; the result must be the greedy
    heavy-edge matching of that symmetrised graph (distinct weights => unique) and maximal on it."""
    from tgp import kernels as KK
    g = torch.Generator().manual_seed(9500 + seed)
    n = [30, 200, 200, 3000][seed]
    e = 5 * n
    r = torch.randint(0, n, (e,), generator=g)
    c = torch.randint(0, n, (e,), generator=g)
    keep_rev = torch.rand(e, generator=g) < 0.7                       # 30 % of the pairs are one-directional
    ei = torch.cat([torch.stack([r, c]), torch.stack([c[keep_rev], r[keep_rev]])], 1)
    ei = torch.unique(ei, dim=1)                                       # coalesced: sorted by (row, col)
    if seed == 2:
        ei = ei[:, torch.randperm(ei.size(1), generator=g)]           # unsorted list: CSR rows in arbitrary column order
    w = torch.rand(ei.size(1), generator=g) + 0.1                      # a different weight on every directed entry
    label = KK.graclus_match(ei.to(dev), w.to(dev), n).cpu()
    # reference: symmetrise (max over the two directions, drop entries without reverse), sequential greedy
    key = {(int(a), int(b)): float(x) for a, b, x in zip(ei[0], ei[1], w)}
    und = {}
    for (a, b), x in key.items():
        if a != b and (b, a) in key:
            und[(min(a, b), max(a, b))] = max(x, key[(b, a)])
    ref = torch.arange(n)
    free = [True] * n
    for (a, b), _ in sorted(und.items(), key=lambda kv: -kv[1]):
        if free[a] and free[b]:
            free[a] = free[b] = False
            ref[a] = ref[b] = a
    assert torch.equal(label, ref)


def test_native_selectors_degenerate_inputs(dev):
    """Empty and degenerate inputs through the native selectors: no nodes, no edges, only self loops, scores with
    +-inf / NaN, a ratio that keeps everything, one-node graphs."""
    from tgp.poolers import get_pooler
    from tgp.select import GraclusSelect, TopkSelect
    # TopK: scores with infinities and NaN (torch.sort's descending order puts NaN first, then +inf)
    score = torch.tensor([0.5, float("inf"), -1.0, float("nan"), float("-inf"), 0.5, 2.0, -0.0, 0.0])
    batch = torch.tensor([0, 0, 0, 0, 0, 1, 1, 1, 1])
    so = TopkSelect(in_channels=None, ratio=0.5, act="linear").to(dev)(score.view(-1, 1).to(dev), batch=batch.to(dev))
    kept = set(so.node_index.cpu().tolist())
    assert kept == {3, 1, 0, 6, 5}, kept                                  # ceil(0.5*5)=3 of graph 0, 2 of graph 1
    assert so.cluster_index.cpu().tolist() == [2, 1, 0, 4, 3]             # rows sorted by node id; rank in score order
    # ratio >= number of nodes keeps everything; single-node graphs
    so = TopkSelect(in_channels=None, ratio=7, act="linear").to(dev)(torch.randn(3, 1, device=dev),
                                                                      batch=torch.tensor([0, 1, 2], device=dev))
    assert so.num_supernodes == 3 and so.node_index.cpu().tolist() == [0, 1, 2]
    # zero nodes
    so = TopkSelect(in_channels=4, ratio=0.5).to(dev)(torch.zeros(0, 4, device=dev),
                                                     batch=torch.zeros(0, dtype=torch.long, device=dev))
    assert so.num_nodes == 0 and so.num_supernodes == 0
    # Graclus: no edges / only self loops -> every node is its own cluster
    for ei in (torch.zeros(2, 0, dtype=torch.long), torch.tensor([[0, 1, 2], [0, 1, 2]])):
        so = GraclusSelect()(ei.to(dev), None, num_nodes=5)
        assert so.num_supernodes == 5 and so.cluster_index.cpu().tolist() == [0, 1, 2, 3, 4]
    # a two-node graph with one undirected edge collapses to one supernode; pooling it end to end works
    x = torch.randn(2, 3, device=dev)
    out = get_pooler("graclus")(x=x, adj=torch.tensor([[0, 1], [1, 0]], device=dev))
    assert out.x.shape == (1, 3) and out.edge_index.size(1) == 0
    torch.testing.assert_close(out.x[0], x.sum(0))


@pytest.mark.parametrize("dtype", [torch.float64, torch.bfloat16])
def test_operator_outputs_carry_the_input_dtype(dev, dtype):
    """A float64 / bf16 caller gets tensors of its own dtype back (as from the reference's ATen ops) and gradients of
    the input dtype.  bf16: the fp32 result rounded to bf16.  float64 (r4): the HBM-bound operators (sparse Reduce,
    SparseConnect) compute in fp64 -- equal to the fp64 arithmetic of the same formula at 1e-12 --, the dense GEMM
    path computes in fp32 and casts."""
    from tgp.connect import DenseConnect, SparseConnect
    from tgp.lift import BaseLift
    from tgp.reduce import BaseReduce
    from tgp.select import SelectOutput
    g = torch.Generator().manual_seed(3)
    n, f, k = 40, 6, 9
    x = torch.randn(n, f, generator=g)
    cluster = torch.randint(0, k, (n,), generator=g)
    cluster[:k] = torch.arange(k)
    w = torch.rand(n, generator=g) + 0.5
    so = SelectOutput(cluster_index=cluster.to(dev), num_nodes=n, num_supernodes=k, weight=w.to(dev))
    ref = BaseReduce()(x.to(dev), so)[0]
    xt = x.to(dev).to(dtype).requires_grad_(True)
    out = BaseReduce()(xt, so)[0]
    assert out.dtype == dtype
    ref_t = BaseReduce()(xt.detach().float(), so)[0]
    if dtype == torch.float64:
        exact = torch.zeros(k, f, dtype=torch.float64).index_add_(0, cluster, x.double() * w.double().view(-1, 1))
        torch.testing.assert_close(out.detach().cpu(), exact, rtol=1e-12, atol=1e-12)
        torch.testing.assert_close(out.float(), ref, rtol=1e-6, atol=1e-6)
    else:
        torch.testing.assert_close(out.float(), ref_t.to(dtype).float(), rtol=0, atol=0)
    out.sum().backward()
    assert xt.grad.dtype == dtype
    lifted = BaseLift(matrix_op="transpose")(out.detach(), so)
    assert lifted.dtype == dtype and lifted.shape == (n, f)
    # sparse Connect: pooled weights follow the input weights
    ei = torch.randint(0, n, (2, 200), generator=g).to(dev)
    ew = (torch.rand(200, generator=g) + 0.5).to(dev)
    e32, w32 = SparseConnect()(ei, so, edge_weight=ew)
    e_t, w_t = SparseConnect()(ei, so, edge_weight=ew.to(dtype))
    assert w_t.dtype == dtype and torch.equal(e_t, e32)
    if dtype == torch.float64:
        torch.testing.assert_close(w_t.float(), w32, rtol=1e-6, atol=1e-6)
    # dense Connect / Reduce: results follow S
    s = torch.softmax(torch.randn(2, 12, 4, generator=g), -1).to(dev)
    a = (torch.rand(2, 12, 12, generator=g) < 0.3).float().to(dev)
    sd = SelectOutput(s=s.to(dtype))
    ap, _ = DenseConnect()(a.to(dtype), sd)
    assert ap.dtype == dtype
    assert DenseConnect().dense_connect(a.to(dtype), s.to(dtype)).dtype == dtype
    xd = torch.randn(2, 12, 5, generator=g).to(dev).to(dtype)
    assert BaseReduce()(xd, sd)[0].dtype == dtype
    if dtype == torch.float64:
        ap32, _ = DenseConnect()(a, SelectOutput(s=s))
        torch.testing.assert_close(ap.float(), ap32, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("n,e,ratio", [(50, 0, 0.5), (50, 7, 0.5), (300, 4096, 0.5), (300, 4097, 0.9), (5_000, 70_000, 0.3),
                                       (200_000, 1_500_000, 0.5), (1_500_000, 3_000_000, 0.6), (2_000, 2_500_000, 1.0)])
@pytest.mark.parametrize("weighted", [True, False])
def test_subgraph_filter_matches_torch_across_chunk_counts(dev, n, e, ratio, weighted):
    """Induced subgraph + relabel + filters against the same selection written with torch ops, from an empty list
    and a single chunk to hundreds of chunks per workgroup, with the node bitmap in LDS or (N > 1.2 M) in global
    memory, sorted and unsorted node_index, all edges or none surviving."""
    import tgp.kernels as KK
    g = torch.Generator(device=dev).manual_seed(n + e)
    ei = torch.randint(0, n, (2, e), device=dev, generator=g)
    w = (torch.rand(e, device=dev, generator=g) - 0.2) if weighted else None
    if weighted and e:
        w[torch.rand(e, device=dev, generator=g) < 0.05] = 0.0  # eps filter
    k = max(1, int(n * ratio))
    keep = torch.sort(torch.randperm(n, device=dev, generator=g)[:k])[0]
    for node_index in (keep, None, keep.flip(0)):
        for rsl in (True, False):
            got_ei, got_w = KK.filter_edges(ei, w, node_index, n, rsl)
            mask = torch.ones(e, dtype=torch.bool, device=dev)
            src, dst = ei[0], ei[1]
            if node_index is not None:
                relabel = torch.full((n,), -1, dtype=torch.long, device=dev)
                relabel[node_index] = torch.arange(node_index.numel(), device=dev)
                mask &= (relabel[src] >= 0) & (relabel[dst] >= 0)
                src, dst = relabel[src], relabel[dst]
            if rsl:
                mask &= ei[0] != ei[1]
            if w is not None:
                mask &= w.abs() > 1e-8
            assert torch.equal(got_ei, torch.stack([src[mask], dst[mask]]))
            assert (got_w is None) == (w is None) and (w is None or torch.equal(got_w, w[mask]))
    # nothing survives
    none_ei, none_w = KK.filter_edges(ei, w, keep[:0], n, True)
    assert none_ei.shape == (2, 0) and (none_w is None or none_w.numel() == 0)
