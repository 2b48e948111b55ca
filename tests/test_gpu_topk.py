"""TopkSelect and TopK pooling (csrc/topk_select.hip; reference select/topk_select.py:163-203, poolers/topk.py): scores, selection, min_score mode, backward, the one-node training step.

Regrouped by operator in round 6 from the per-round files test_gpu_round2..5.py; the test bodies are unchanged."""
import math
import warnings
import pytest
import torch
import os
import sys

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


# ------------------------------------------------------------------------------ Graclus: all rounds of a batch in one launch
def _graph_batch(sizes, deg, seed, dev, weights="rand"):
    g = torch.Generator().manual_seed(seed)
    rows, cols, off = [], [], 0
    for n in sizes:
        if n >= 2:
            m = max(1, int(n * deg / 2))
            a = torch.randint(0, n, (m,), generator=g)
            b = torch.randint(0, n, (m,), generator=g)
            keep = a != b
            a, b = a[keep] + off, b[keep] + off
            rows += [a, b]
            cols += [b, a]
        off += n
    if rows:
        ei = torch.stack([torch.cat(rows), torch.cat(cols)])
        ei = torch.unique(ei[0] * off + ei[1])
        ei = torch.stack([ei // off, ei % off])
    else:
        ei = torch.zeros(2, 0, dtype=torch.long)
    if weights == "rand":
        half = torch.rand(off * off if off < 300 else 1, generator=g)
        lo, hi = torch.minimum(ei[0], ei[1]), torch.maximum(ei[0], ei[1])
        ew = (torch.sin((lo * 7919 + hi * 104729).double()) * 0.5 + 0.6).float()   # symmetric, many distinct values
        del half
    elif weights == "ties":
        ew = torch.ones(ei.size(1))
    else:
        ew = None
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    ptr = torch.zeros(len(sizes) + 1, dtype=torch.long)
    ptr[1:] = torch.cumsum(torch.tensor(sizes), 0)
    return ei.to(dev), (ew.to(dev) if ew is not None else None), batch.to(dev), ptr.to(dev), off


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


def test_sparse_connect_topk_edge_weight_gradient_and_normalisations(dev):
    """Filter path (kept-node subgraph): kept edges pass their gradient through, dropped edges get zero; with
    degree_norm / edge_weight_norm the differentiable normalisation sits on top (ops.py:383-417)."""
    from tgp.connect import SparseConnect
    from tgp.select import SelectOutput
    g = torch.Generator().manual_seed(6)
    n, e = 50, 400
    ei = torch.randint(0, n, (2, e), generator=g)
    w = torch.rand(e, generator=g) + 0.1
    kept = torch.sort(torch.randperm(n, generator=g)[:25])[0]
    so = SelectOutput(node_index=kept.to(dev), cluster_index=torch.arange(25, device=dev), num_nodes=n, num_supernodes=25)
    member = torch.zeros(n, dtype=torch.bool)
    member[kept] = True
    keep = member[ei[0]] & member[ei[1]] & (ei[0] != ei[1])
    relabel = torch.full((n,), -1, dtype=torch.long)
    relabel[kept] = torch.arange(25)
    for dn, ewn in ((False, False), (True, False), (True, True)):
        wd = w.to(dev).requires_grad_(True)
        bp = torch.zeros(25, dtype=torch.long, device=dev)
        out_ei, out_w = SparseConnect(degree_norm=dn, edge_weight_norm=ewn)(ei.to(dev), so, edge_weight=wd,
                                                                            batch_pooled=bp)
        wr = w.clone().requires_grad_(True)
        r, c, ww = relabel[ei[0][keep]], relabel[ei[1][keep]], wr[keep]
        if dn:
            deg = torch.zeros(25).index_add(0, r, ww)
            dis = deg.clamp(min=1e-8).pow(-0.5)
            ww = ww * dis[r] * dis[c]
        if ewn:
            ww = ww / ww.abs().max()
        assert torch.equal(out_ei.cpu(), torch.stack([r, c]))
        torch.testing.assert_close(out_w.detach().cpu(), ww.detach(), rtol=1e-5, atol=1e-6)
        coef = torch.randn(ww.numel(), generator=g)
        (out_w * coef.to(dev)).sum().backward()
        (ww * coef).sum().backward()
        torch.testing.assert_close(wd.grad.cpu(), wr.grad, rtol=1e-4, atol=1e-6)
        assert bool((wd.grad.cpu()[~keep] == 0).all())


def test_hierarchy_dense_sparse_output_into_topk_trains_edge_weights(dev):
    """a dense pooler with sparse_output=True feeding a TopK level -- the second level's pooled weights
    must carry a gradient back to the first level's parameters."""
    from tgp.poolers import get_pooler
    g = torch.Generator().manual_seed(7)
    n = 30
    a = torch.triu(torch.rand(n, n, generator=g) < 0.2, 1)
    ei = (a | a.t()).nonzero().t().to(dev)
    x = torch.randn(n, 8, generator=g).to(dev)
    p1 = get_pooler("mincut", in_channels=8, k=10, sparse_output=True).to(dev)
    p2 = get_pooler("topk", in_channels=8, ratio=0.5).to(dev)
    o1 = p1(x=x, adj=ei, batch=torch.zeros(n, dtype=torch.long, device=dev))
    assert o1.edge_weight.requires_grad
    o2 = p2(x=o1.x, adj=o1.edge_index, edge_weight=o1.edge_weight, batch=o1.batch)
    assert o2.edge_weight.requires_grad
    o2.edge_weight.sum().backward()
    grads = [p.grad for p in p1.parameters() if p.grad is not None]
    assert grads and any(float(gr.abs().sum()) > 0 for gr in grads)


# ------------------------------------------------------------------------------ TopkSelect, min_score mode (r3)
@pytest.mark.parametrize("seed", range(6))
def test_topk_min_score_mode_native_vs_oracle(dev, seed, monkeypatch):
    """min_score mode (select/topk_select.py:186-194): per-graph softmax + threshold + nonzero() as native kernels;
    node_index / cluster_index bit-exact vs the oracle, weights within 1e-5, no torch scatter / nonzero in the path."""
    import tgp_oracle as O
    from tgp.select import TopkSelect
    g = torch.Generator().manual_seed(seed)
    sizes = torch.randint(1, 90, (int(torch.randint(1, 40, (1,), generator=g)),), generator=g)
    if seed == 5:
        sizes = torch.tensor([5000, 3, 1])
    n = int(sizes.sum())
    batch = torch.repeat_interleave(torch.arange(sizes.numel()), sizes)
    f = 9
    x = torch.randn(n, f, generator=g)
    min_score = [0.02, 0.05, 0.5, 1e-4, 0.9, 2e-4][seed]
    sel = TopkSelect(in_channels=f, ratio=None, min_score=min_score).to(dev)
    calls = []
    real_nonzero = torch.Tensor.nonzero
    monkeypatch.setattr(torch.Tensor, "nonzero", lambda self, *a, **k: (calls.append("nonzero"), real_nonzero(self, *a, **k))[1])
    so = sel(x=x.to(dev), batch=batch.to(dev))
    monkeypatch.setattr(torch.Tensor, "nonzero", real_nonzero)
    assert calls == []
    ni, ci, w = O.topk_select(x, sel.weight.detach().cpu(), None, batch, min_score, "tanh")
    assert torch.equal(so.node_index.cpu(), ni) and torch.equal(so.cluster_index.cpu(), ci)
    torch.testing.assert_close(so.weight.cpu(), w, rtol=1e-5, atol=1e-7)
    # no batch vector: one graph
    so1 = sel(x=x.to(dev))
    ni1, _, w1 = O.topk_select(x, sel.weight.detach().cpu(), None, None, min_score, "tanh")
    assert torch.equal(so1.node_index.cpu(), ni1)
    torch.testing.assert_close(so1.weight.cpu(), w1, rtol=1e-5, atol=1e-7)


def test_topk_min_score_mode_gradients(dev):
    """The selection weights stay differentiable w.r.t. x and the projection (softmax Jacobian per graph)."""
    from tgp.select import TopkSelect
    g = torch.Generator().manual_seed(0)
    sizes = torch.tensor([30, 12, 47])
    batch = torch.repeat_interleave(torch.arange(3), sizes).to(dev)
    x = torch.randn(int(sizes.sum()), 6, generator=g).to(dev).requires_grad_(True)
    sel = TopkSelect(in_channels=6, ratio=None, min_score=0.02).to(dev)
    so = sel(x=x, batch=batch)
    up = torch.randn(so.weight.numel(), generator=g).to(dev)
    (so.weight * up).sum().backward()
    gx, gw = x.grad.clone(), sel.weight.grad.clone()
    x.grad = None
    sel.weight.grad = None
    score = (x * sel.weight).sum(-1)
    mx = torch.zeros(3, device=dev).scatter_reduce_(0, batch, score.detach(), "amax", include_self=False)
    e = (score - mx[batch]).exp()
    p = e / (torch.zeros(3, device=dev).index_add_(0, batch, e) + 1e-16)[batch]
    (p[so.node_index] * up).sum().backward()
    torch.testing.assert_close(gx, x.grad, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(gw, sel.weight.grad, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("n,F", [(1, 4), (100, 16), (3000, 32), (777, 7), (50, 300), (4096, 128), (20, 1024)])
@pytest.mark.parametrize("act", ["tanh", "linear"])
def test_topk_score_kernel_vs_torch(dev, n, F, act):
    """tgp_topk_score_f32 = act(x.w / ||w||) (topk_select.py:176-184) within 1e-6 of the torch expression; TopkSelect
    under no_grad takes it and selects the very nodes the differentiable route selects."""
    from tgp import kernels
    from tgp.select import TopkSelect
    g = torch.Generator().manual_seed(n * 31 + F)
    x = torch.randn(n, F, generator=g).to(dev)
    sel = TopkSelect(in_channels=F, ratio=0.5, act=act).to(dev)
    w = sel.weight.detach()
    want = (x.double() * w.double()).sum(-1) / w.double().norm(p=2, dim=-1)
    want = torch.tanh(want) if act == "tanh" else want
    got = kernels.topk_score(x, w, act == "tanh")
    assert torch.allclose(got.double(), want, rtol=1e-5, atol=1e-6)
    batch = torch.sort(torch.randint(0, 5, (n,), generator=g)).values.to(dev)
    with torch.no_grad():
        fused = sel(x, batch=batch)
    x2 = x.clone().requires_grad_(True)
    plain = sel(x2, batch=batch)
    assert plain.s.values().requires_grad
    a = torch.stack([fused.node_index, fused.cluster_index])
    b = torch.stack([plain.node_index, plain.cluster_index])
    if torch.equal(a, b):
        assert torch.allclose(fused.s.values(), plain.s.values().detach(), rtol=1e-5, atol=1e-6)
    else:  # a last-place difference may swap two nodes whose scores agree to rounding: same scores, sorted per graph
        assert torch.allclose(fused.s.values().sort().values, plain.s.values().detach().sort().values, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("act", ["tanh", "linear"])
@pytest.mark.parametrize("n,F", [(500, 16), (37, 7), (4096, 128)])
def test_topk_score_function_gradients_vs_torch(dev, act, n, F):
    """Fn.topk_score (forward: the fused kernel; backward in closed form) against autograd on the reference's expression
    act((x * w).sum(-1) / w.norm()) (topk_select.py:176-184), fp64."""
    from tgp import functions as Fn
    g = torch.Generator().manual_seed(n + F)
    x = torch.randn(n, F, generator=g).to(dev).requires_grad_(True)
    w = (torch.rand(1, F, generator=g) - 0.5).to(dev).requires_grad_(True)
    up = torch.randn(n, generator=g).to(dev)
    s = Fn.topk_score(x, w, act == "tanh")
    s.backward(up)
    xd, wd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    t = (xd * wd).sum(-1) / wd.norm(p=2, dim=-1)
    sd = torch.tanh(t) if act == "tanh" else t
    sd.backward(up.double())
    assert torch.allclose(s.detach().double(), sd.detach(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(x.grad.double(), xd.grad, rtol=1e-4, atol=1e-6)
    assert torch.allclose(w.grad.double(), wd.grad, rtol=1e-4, atol=1e-4 * float(wd.grad.abs().max()))


@pytest.mark.parametrize("sizes", [[50] * 40, [1, 2, 3, 700, 5], [5000]])
def test_topk_select_hands_reduce_its_transposed_index(dev, sizes):
    """Under autograd TopkSelect's compaction also emits the node -> assignment CSR offsets (perm = identity): equal to
    the index build_assign_index derives from node_index, and x.grad through the pooler equals the reference expression."""
    from tgp import kernels
    from tgp.poolers import get_pooler
    ei, ew, batch, ptr, n = _graph_batch(sizes, 3.0, 13, dev)
    x = torch.randn(n, 9, device=dev, requires_grad=True)
    pooler = get_pooler("topk", in_channels=9, ratio=0.4).to(dev)
    out = pooler(x=x, adj=ei, edge_weight=ew, batch=batch)
    lift = out.so._lift_index
    assert lift is not None and lift.perm is None
    want = kernels.build_assign_index(out.so.node_index, n)
    assert torch.equal(lift.row_ptr, want.row_ptr) and torch.equal(want.perm, torch.arange(want.nnz, device=dev, dtype=torch.int32))
    up = torch.randn_like(out.x)
    out.x.backward(up)
    # reference: x_pool = x[node_index] * score[node_index, None]; score = tanh(x w / |w|)
    xd = x.detach().double().requires_grad_(True)
    wd = pooler.selector.weight.detach().double()
    score = torch.tanh((xd * wd).sum(-1) / wd.norm(p=2, dim=-1))
    ni, ci = out.so.node_index, out.so.cluster_index      # row ci[j] of x_pool is node ni[j]
    (xd[ni] * score[ni].unsqueeze(1)).backward(up.double()[ci])
    assert torch.allclose(x.grad.double(), xd.grad, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("sizes", [[3000, 100, 8192, 2049], [2669, 506, 152, 1176], [5000]])
def test_topk_large_segment_sort_equals_the_device_wide_sort(dev, sizes):
    """Graphs of 2049 .. 8192 nodes are ranked one workgroup per graph (1024 threads, LDS bitonic network) instead of by
    the device-wide radix sort: identical node_index / cluster_index, ties included (lower node id first)."""
    from tgp import kernels
    from tgp.utils.ops import batch_info
    g = torch.Generator().manual_seed(sum(sizes))
    n = sum(sizes)
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes)).to(dev)
    score = torch.randn(n, generator=g)
    score[torch.randint(0, n, (n // 10,), generator=g)] = 0.25      # ties
    score = score.to(dev)
    info = batch_info(batch)
    k, koff = kernels.topk_plan(info.sizes, 0.3)
    k_total = int(koff[-1])
    seg = kernels.topk_select(score, batch, info.num_graphs, info.ptr, k, koff, k_total, segments_max_nodes=max(sizes))
    rad = kernels.topk_select(score, batch, info.num_graphs, info.ptr, k, koff, k_total, segments_max_nodes=0)
    assert torch.equal(seg[0], rad[0]) and torch.equal(seg[1].perm, rad[1].perm)


@pytest.mark.parametrize("ratio", [0.5, 0.1, 0.37, 3, 1.0])
def test_batch_facts_carry_the_topk_total(dev, ratio):
    """A TopK selector that reads the batch facts first gets sum_g k_g with them: equal to the plan kernel's last prefix sum
    (PyG's ceil(fp32(ratio) * n_g) / min(ratio, n_g))."""
    from tgp import kernels
    from tgp.utils.ops import batch_info
    g = torch.Generator().manual_seed(17)
    batch = torch.repeat_interleave(torch.arange(500), torch.randint(1, 90, (500,), generator=g)).to(dev)
    info = batch_info(batch, topk_ratio=float(ratio))
    k, koff = kernels.topk_plan(info.sizes, ratio)
    assert info.memo[("topk_total", float(ratio))] == int(koff[-1]) == int(k.sum())


def test_topk_select_directory_and_one_launch_subgraph_connect(dev):
    """r5: TopkSelect on a large graph leaves the kept-node bitmap + rank directory of its compaction pass on the
    SelectOutput; SparseConnect hands them to tgp_connect_subgraph_single, which then needs no memset / scatter /
    directory scan.  The directory is checked against node_index, the Connect against the route without it (bit for
    bit), against the oracle, and a bad endpoint still raises (now through the epoch-tagged status word)."""
    import tgp_oracle as O
    from tgp import kernels
    from tgp.connect import SparseConnect
    from tgp.select import TopkSelect
    g = torch.Generator().manual_seed(11)
    for n in (10_000, 131_072, 300_001):
        e = 6 * n
        a, b = torch.randint(0, n, (e,), generator=g), torch.randint(0, n, (e,), generator=g)
        ei = torch.stack([torch.cat([a, b]), torch.cat([b, a])])
        ew = torch.rand(ei.size(1), generator=g)
        ew[::17] = 0.0                                       # the |w| > eps filter has work
        x = torch.randn(n, 4, generator=g)
        torch.manual_seed(1)
        sel = TopkSelect(in_channels=4, ratio=0.37).to(dev)
        with torch.no_grad():
            so = sel(x=x.to(dev))
        md = so._assign_index.member_directory
        assert md is not None, "the device-wide route of tgp_topk_select writes the directory"
        nblk = md.numel() // 5
        bits = md[: 4 * nblk].cpu().view(torch.int32)
        member = torch.zeros(4 * nblk * 32, dtype=torch.bool)
        member[so.node_index.cpu()] = True
        want_bits = (member.view(-1, 32).long() << torch.arange(32)).sum(1)
        assert torch.equal(bits.long() & 0xFFFFFFFF, want_bits)
        rank = torch.cat([torch.zeros(1, dtype=torch.long), member.view(-1, 128).sum(1).cumsum(0)[:-1]])
        assert torch.equal(md[4 * nblk:].cpu().long(), rank)
        conn = SparseConnect()
        ei_d, ew_d = ei.to(dev), ew.to(dev)
        with kernels.output_views():
            pe, pw = conn(ei_d, so, edge_weight=ew_d)        # with the directory: ONE launch
        pe2, pw2 = kernels.filter_edges(ei_d, ew_d, so.node_index, n, True)   # without it
        assert torch.equal(pe, pe2) and torch.equal(pw, pw2)
        if n <= 131_072:
            r_ei, r_ew = O.sparse_connect(ei, ew, so.node_index.cpu(), None, n, int(so.num_supernodes))
            assert torch.equal(pe.cpu(), r_ei) and torch.equal(pw.cpu(), r_ew)
    bad = ei_d.clone()
    bad[1, 12345] = n + 7
    with pytest.raises(IndexError, match="outside"):
        conn(bad, so, edge_weight=ew_d)
    pe3, pw3 = conn(ei_d, so, edge_weight=ew_d)              # the buffers of the refused call left nothing behind
    assert torch.equal(pe3, pe2) and torch.equal(pw3, pw2)


@pytest.mark.parametrize("F", [4, 16, 32, 64, 100, 128, 256])
@pytest.mark.parametrize("use_tanh", [True, False])
def test_topk_pool_backward_kernel_vs_autograd_in_double(dev, F, use_tanh):
    """tgp_topk_pool_bwd_f32 against torch autograd of the same expression evaluated in float64 (select/topk_select.py:
    176-184 score, kept values as the weights of S, reduce/base_reduce.py:141-155 gate): every combination of present /
    absent incoming gradients and wanted outputs, a permuted supernode order, rows of dropped nodes exactly zero."""
    from tgp import kernels as K_
    g = torch.Generator().manual_seed(F + int(use_tanh))
    n, k = 3001, 1234
    x = torch.randn(n, F, generator=g)
    w = torch.randn(F, generator=g)
    node = torch.randperm(n, generator=g)[:k].sort().values
    cluster = torch.randperm(k, generator=g)
    gp = torch.randn(k, F, generator=g)
    gv = torch.randn(k, generator=g)

    def want(use_gp, use_gv):
        xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
        t = (xd @ wd) / wd.norm()
        s = torch.tanh(t) if use_tanh else t
        vals = s[node]
        xp = torch.zeros(k, F, dtype=torch.float64).index_add(0, cluster, vals[:, None] * xd[node])
        loss = 0
        if use_gp:
            loss = loss + (xp * gp.double()).sum()
        if use_gv:
            loss = loss + (vals * gv.double()).sum()
        loss.backward()
        return xd.grad, wd.grad, vals.detach().float()

    xg, wg = x.to(dev), w.to(dev)
    assert K_.topk_pool_bwd_fits(xg, wg) == (F % 4 == 0)
    if F % 4:
        return
    for use_gp, use_gv in ((True, True), (True, False), (False, True)):
        ex, ew_, vals = want(use_gp, use_gv)
        for want_gx, want_gw in ((True, True), (True, False), (False, True)):
            gx, gw = K_.topk_pool_bwd(xg, node.to(dev), cluster.to(dev), vals.to(dev), gp.to(dev) if use_gp else None,
                                      gv.to(dev) if use_gv else None, wg, use_tanh, want_gx, want_gw)
            assert (gx is not None) == want_gx and (gw is not None) == want_gw
            if want_gx:
                torch.testing.assert_close(gx.cpu().double(), ex, rtol=2e-5, atol=2e-5)
                dropped = torch.ones(n, dtype=torch.bool)
                dropped[node] = False
                assert not gx.cpu()[dropped].any()
            if want_gw:
                torch.testing.assert_close(gw.cpu().double(), ew_, rtol=2e-4, atol=2e-4 * float(ew_.abs().max()))
    # identity supernode order (cluster = None) and an empty selection
    ex, ew_, vals = want(True, True)
    gx, gw = K_.topk_pool_bwd(xg, node.to(dev), None, vals.to(dev), gp.to(dev)[cluster.to(dev)], gv.to(dev), wg,
                              use_tanh, True, True)
    torch.testing.assert_close(gx.cpu().double(), ex, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(gw.cpu().double(), ew_, rtol=2e-4, atol=2e-4 * float(ew_.abs().max()))
    empty = torch.empty(0, dtype=torch.long, device=dev)
    gx, gw = K_.topk_pool_bwd(xg, empty, None, torch.empty(0, device=dev), None, torch.empty(0, device=dev), wg, use_tanh,
                              True, True)
    assert not gx.any() and not gw.any()
    # an upstream gradient and a projection that sit at odd offsets of larger buffers (4-byte aligned only)
    big = torch.empty(k * F + 1, device=dev)
    big[1:] = gp.to(dev).reshape(-1)
    wbig = torch.empty(F + 1, device=dev)
    wbig[1:] = wg
    odd = K_.topk_pool_bwd(xg, node.to(dev), cluster.to(dev), vals.to(dev), big[1:].view(k, F), gv.to(dev), wbig[1:], use_tanh,
                           True, True)
    torch.testing.assert_close(odd[0].cpu().double(), ex, rtol=2e-5, atol=2e-5)
    again = K_.topk_pool_bwd(xg, node.to(dev), cluster.to(dev), vals.to(dev), gp.to(dev), gv.to(dev), wg, use_tanh, True, True)
    once = K_.topk_pool_bwd(xg, node.to(dev), cluster.to(dev), vals.to(dev), gp.to(dev), gv.to(dev), wg, use_tanh, True, True)
    assert torch.equal(again[0], once[0]) and torch.equal(again[1], once[1])  # fixed-order sums


@pytest.mark.parametrize("shape", ["small_graphs", "one_large_graph", "no_batch"])
@pytest.mark.parametrize("kw", [dict(ratio=0.5), dict(ratio=0.25, multiplier=1.5, nonlinearity="identity"), dict(ratio=7)])
def test_topk_pooler_training_step_as_one_autograd_node(dev, shape, kw, monkeypatch):
    """TopkPooling in training (poolers/topk.py:150-190): the forward is the inference call, one node carries the
    gradient.  Outputs equal the operator-by-operator graph's (the score's tanh is the fused kernel's instead of ATen's:
    a few ulp), gradients of x and the projection agree to fp32 accumulation-order tolerance -- including what reaches
    the projection through ``so.s`` (Lift) -- and the backward is ONE native call."""
    import tgp.poolers as P
    import tgp.src as S
    from tgp import kernels as K_
    from tgp.poolers import get_pooler
    if shape == "small_graphs":
        x0, ei, ew, batch = _er_batch(120, 4, 60, 32, 31, dev)
    else:
        x0, ei, ew, batch = _er_batch(1, 2500, 2500, 32, 32, dev)
        if shape == "no_batch":
            batch = None
    torch.manual_seed(9)
    pooler = get_pooler("topk", in_channels=32, **kw).to(dev).train()
    calls = []
    real = K_.topk_pool_bwd
    monkeypatch.setattr(K_, "topk_pool_bwd", lambda *a, **k: (calls.append(1), real(*a, **k))[1])

    def step(x_needs_grad):
        pooler.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(x_needs_grad)
        out = pooler(x=x, adj=ei, edge_weight=ew, batch=batch)
        lifted = pooler(x=out.x, so=out.so, lifting=True)
        loss = out.x.square().sum() + (lifted * x0).sum() * 0.3 + (out.so.s.coalesce().values() ** 3).sum() * 0.1
        loss.backward()
        return (out.x.detach(), out.edge_index, out.edge_weight, out.batch, out.so.s.detach().coalesce(), x.grad,
                pooler.selector.weight.grad.clone())

    for x_needs_grad in (True, False):
        calls.clear()
        monkeypatch.setattr(P, "_FOLD_TRAINING", True)
        monkeypatch.setattr(S, "_FOLD_TRAINING", True)
        new = step(x_needs_grad)
        assert calls == [1]
        monkeypatch.setattr(P, "_FOLD_TRAINING", False)
        monkeypatch.setattr(S, "_FOLD_TRAINING", False)
        old = step(x_needs_grad)
        assert calls == [1]
        assert torch.equal(new[4].indices(), old[4].indices())
        torch.testing.assert_close(new[4].values(), old[4].values(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(new[0], old[0], rtol=1e-5, atol=1e-6)
        assert torch.equal(new[1], old[1]) and torch.equal(new[2], old[2])
        assert (new[3] is None and old[3] is None) or torch.equal(new[3], old[3])
        if x_needs_grad:
            torch.testing.assert_close(new[5], old[5], rtol=1e-4, atol=1e-5 * max(1.0, float(old[5].abs().max())))
        else:
            assert new[5] is None and old[5] is None
        torch.testing.assert_close(new[6], old[6], rtol=2e-4, atol=2e-5 * max(1.0, float(old[6].abs().max())))


def test_topk_pool_backward_is_linear_in_the_upstream_gradients_at_full_size(dev):
    """tgp_topk_pool_bwd_f32 at BASELINE configs[3]'s size (N = 1M, F = 128, ratio 0.5): the gradients are linear in
    (dL/dx', dL/d values) -- bwd(a g1 + b g2) = a bwd(g1) + b bwd(g2) to fp32 rounding --, rows of dropped nodes are
    exactly zero, two runs agree bit for bit."""
    from tgp import kernels as K_
    g = torch.Generator(device=dev).manual_seed(11)
    n, F = 1_000_000, 128
    x = torch.randn(n, F, device=dev, generator=g)
    w = torch.randn(F, device=dev, generator=g)
    keep = torch.rand(n, device=dev, generator=g) < 0.5
    node = keep.nonzero().view(-1)
    k = node.numel()
    vals = torch.tanh((x[node] @ w) / w.norm())
    g1, g2 = torch.randn(k, F, device=dev, generator=g), torch.randn(k, F, device=dev, generator=g)
    v1, v2 = torch.randn(k, device=dev, generator=g), torch.randn(k, device=dev, generator=g)

    def bwd(gp, gv):
        return K_.topk_pool_bwd(x, node, None, vals, gp, gv, w, True, True, True)

    a, b = 0.75, -1.5
    x1, w1 = bwd(g1, v1)
    x2, w2 = bwd(g2, v2)
    xc, wc = bwd(a * g1 + b * g2, a * v1 + b * v2)
    ref_x, ref_w = a * x1 + b * x2, a * w1 + b * w2
    scale_x = float(ref_x.abs().max())
    assert float((xc - ref_x).abs().max()) <= 2e-5 * scale_x
    assert float((wc - ref_w).abs().max()) <= 2e-4 * float(ref_w.abs().max())
    assert not xc[~keep].any()
    again_x, again_w = bwd(g1, v1)
    assert torch.equal(again_x, x1) and torch.equal(again_w, w1)
