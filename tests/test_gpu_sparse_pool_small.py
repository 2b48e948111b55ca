"""The one-launch sparse Reduce + Connect of small-graph batches (csrc/sparse_pool_small.hip; reference base_reduce.py:141-155, base_conn.py:79-89) and the exact-size output contract.

Regrouped by operator in round 6 from the per-round files test_gpu_round2..5.py; the test bodies are unchanged."""
import os
import socket
import pytest
import torch
import sys
import warnings

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


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


# ------------------------------------------------------------------------------------ output contract (r5)
def _er_batch(num_graphs, lo, hi, f, seed, dev):
    g = torch.Generator().manual_seed(seed)
    xs, eis, bs, off = [], [], [], 0
    for gi in range(num_graphs):
        n = int(torch.randint(lo, hi + 1, (1,), generator=g))
        a = torch.triu(torch.rand(n, n, generator=g) < 4.0 / n, 1)
        a = a | a.t()
        eis.append(a.nonzero().t() + off)
        xs.append(torch.randn(n, f, generator=g))
        bs.append(torch.full((n,), gi))
        off += n
    x, ei, batch = torch.cat(xs), torch.cat(eis, 1), torch.cat(bs)
    ew = torch.rand(ei.size(1), generator=g) + 0.1
    return x.to(dev), ei.to(dev), ew.to(dev), batch.to(dev)


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


@pytest.mark.parametrize("alias,kw", [("topk", dict(ratio=0.5)), ("graclus", {}), ("ndp", {})])
@pytest.mark.parametrize("shape", ["small_graphs", "small_graphs_own_tensors", "one_large_graph"])
def test_sparse_poolers_hand_out_contiguous_exact_size_edge_lists(dev, alias, kw, shape, monkeypatch):
    """SURVEY 8(b) "Ownership" / connect/base_conn.py:103-112: every output is a NEW tensor of the pooled size.  r4's
    one-launch operators returned ``edge_index`` as a [2, E'] view of a capacity-E buffer (torch.equal ignores strides, so
    parity could not see it): ``.view(-1)`` failed where the reference's tensor works and E * 20 bytes stayed pinned.
    Since r5: contiguous [2, E'] on every sparse pooler and both size regimes; the view layout is opt-in
    (``tgp.output_views``) and gives the same values.  Storage: <= 2 x the logical size -- except (r6) the one-launch
    pooling of a batch of small graphs, whose four outputs are disjoint pieces of ONE allocation of at most
    ``kernels._SPS_ARENA_BYTES`` (verdict r5 item 5: host time); ``TGP_SPS_ARENA=0`` gives them tensors of their own."""
    import tgp
    from tgp import kernels
    from tgp.poolers import get_pooler
    if shape == "small_graphs_own_tensors":
        monkeypatch.setattr(kernels, "_SPS_ARENA", False)
    if shape.startswith("small_graphs"):
        x, ei, ew, batch = _er_batch(200, 5, 60, 8, 7, dev)
    else:
        x, ei, ew, batch = _er_batch(1, 3000, 3000, 8, 8, dev)
    torch.manual_seed(3)
    pooler = get_pooler(alias, in_channels=8, **kw).to(dev).eval()
    with torch.no_grad():
        out = pooler(x=x, adj=ei, edge_weight=ew, batch=batch)
        so = out.so
        # (NDPPooling selects anew on every call, like the reference's; a graph whose sign partition cuts < 0.5 gets the
        #  reference's RANDOM partition, seeded from torch's generator: same seed, same draw)
        torch.manual_seed(11)
        with tgp.output_views():
            out_v = pooler(x=x, adj=ei, edge_weight=ew, batch=batch, so=so)
        torch.manual_seed(11)
        out2 = pooler(x=x, adj=ei, edge_weight=ew, batch=batch, so=so)
    for o in (out, out2):
        e = o.edge_index
        assert e.is_contiguous() and e.stride() == (e.size(1), 1)
        flat = e.view(-1)                                  # the reference's tensor allows this
        assert flat.numel() == 2 * e.size(1)
        shared = e.untyped_storage().data_ptr() == o.x.untyped_storage().data_ptr()
        if shared:  # the arena: one allocation, bounded, the outputs disjoint
            assert shape == "small_graphs" and alias != "ndp"
            assert e.untyped_storage().nbytes() <= kernels._SPS_ARENA_BYTES
            spans = sorted((t.data_ptr(), t.data_ptr() + t.numel() * t.element_size())
                           for t in (o.x, e, o.edge_weight, o.batch) if t is not None and t.numel())
            assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))
        else:
            assert e.untyped_storage().nbytes() <= max(2 * e.numel() * 8, 512)
            if o.edge_weight is not None:
                assert o.edge_weight.untyped_storage().nbytes() <= max(2 * o.edge_weight.numel() * 4, 512)
            assert o.x.untyped_storage().nbytes() <= max(2 * o.x.numel() * 4, 512)
        if o.edge_weight is not None:
            assert o.edge_weight.is_contiguous()
    assert torch.equal(out2.edge_index, out_v.edge_index) and torch.equal(out2.x, out_v.x)
    if out2.edge_weight is not None:
        assert torch.equal(out2.edge_weight, out_v.edge_weight)
    assert torch.equal(out2.batch, out_v.batch)


def test_new_edge_lists_need_no_lower_bounds_launch(dev):
    """r5: the one-launch sparse pooling searches for the per-graph edge ranges of an edge list it has not seen and
    leaves them for the next call (edge_ptr_out): results of the first (searching) and the second (handed-over) call are
    identical, and equal to the staged operators'."""
    import tgp
    from tgp import kernels
    from tgp.poolers import get_pooler
    x, ei, ew, batch = _er_batch(300, 3, 64, 12, 5, dev)
    for alias, kw in (("topk", dict(in_channels=12, ratio=0.5)), ("graclus", {})):
        torch.manual_seed(0)
        pooler = get_pooler(alias, **kw).to(dev).eval()
        e1 = ei.clone()
        with torch.no_grad():
            first = pooler(x=x, adj=e1, edge_weight=ew, batch=batch)
            assert kernels._edge_ptr_memo(e1, kernels._EDGE_PTR[id(e1)][3]()) is not None
            second = pooler(x=x, adj=e1, edge_weight=ew, batch=batch, so=first.so)
            staged_x, staged_b = pooler.reducer(x, first.so, batch=batch)
            staged_e, staged_w = pooler.connector(e1, first.so, edge_weight=ew, batch_pooled=staged_b)
        for o in (first, second):
            assert torch.equal(o.edge_index, staged_e) and torch.equal(o.edge_weight, staged_w)
            assert torch.equal(o.x, staged_x) and torch.equal(o.batch, staged_b)


# ------------------------------------------------------------------------ sparse poolers, training step (r5, late)
@pytest.mark.parametrize("alias,kw", [("topk", dict(ratio=0.5)), ("topk", dict(ratio=0.3, multiplier=2.0)), ("graclus", {})])
@pytest.mark.parametrize("weighted", [True, False])
def test_sparse_pooler_training_forward_is_the_one_launch_call(dev, alias, kw, weighted, monkeypatch):
    """A batch of small graphs in TRAINING: Reduce + Connect are the same single launch as in inference
    (SRCPooling.reduce_connect) with the sparse Reduce's backward attached to x' -- where r4 took the staged operators
    (6 launches for TopK, 10 for Graclus) whenever a gradient was required.  Outputs and every gradient (x, the TopK
    projection through the kept scores, the scores again through ``so.s``) equal the staged route's bit for bit: the
    forward sums are the same sums in the same order and the backward is the same node."""
    import tgp.src as S
    from tgp import kernels as K_
    from tgp.poolers import get_pooler
    x0, ei, ew, batch = _er_batch(150, 4, 60, 16, 21, dev)
    ew = ew if weighted else None
    torch.manual_seed(5)
    pooler = get_pooler(alias, in_channels=16, **kw).to(dev).train()
    calls = []
    real = K_.sparse_pool_small
    monkeypatch.setattr(K_, "sparse_pool_small", lambda *a, **k: (calls.append(1), real(*a, **k))[1])

    def step(x_needs_grad):
        pooler.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(x_needs_grad)
        out = pooler(x=x, adj=ei, edge_weight=ew, batch=batch)
        loss = out.x.square().sum()
        if alias == "topk":
            loss = loss + (out.so.s.coalesce().values() ** 2).sum() * 0.1
        loss.backward()
        return (out.x.detach(), out.edge_index, out.edge_weight, out.batch, x.grad,
                [p.grad.clone() for p in pooler.parameters() if p.grad is not None])

    import tgp.poolers as P
    for x_needs_grad in (True, False):
        if alias == "graclus" and not x_needs_grad:
            continue  # (nothing to differentiate: the inference call)
        calls.clear()
        monkeypatch.setattr(S, "_FOLD_TRAINING", True)
        monkeypatch.setattr(P, "_FOLD_TRAINING", False)  # (TopK's one-node path has its own test below)
        new = step(x_needs_grad)
        assert calls == [1]
        monkeypatch.setattr(S, "_FOLD_TRAINING", False)
        old = step(x_needs_grad)
        assert calls == [1]  # (the staged operators)
        assert torch.equal(new[0], old[0]) and torch.equal(new[1], old[1]) and torch.equal(new[3], old[3])
        assert (new[2] is None and old[2] is None) or torch.equal(new[2], old[2])
        if x_needs_grad:
            assert torch.equal(new[4], old[4])
        assert len(new[5]) == len(old[5]) and (alias != "topk" or len(new[5]) == 1)
        for a, b in zip(new[5], old[5]):
            assert torch.equal(a, b)
