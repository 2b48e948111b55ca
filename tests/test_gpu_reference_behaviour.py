"""Compute-bearing behaviours the reference's own unit tests pin, re-stated against this package on the GPU
(the reference runs them on CPU tensors; here the same calls go through the HIP kernels).  Every test names the
reference test it mirrors (/root/reference/tests/...)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-5, atol=1e-5)


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def chain(n, dev, self_loops=False):
    a = torch.arange(n - 1, device=dev)
    ei = torch.stack([torch.cat([a, a + 1]), torch.cat([a + 1, a])])
    if self_loops:
        d = torch.arange(n, device=dev)
        ei = torch.cat([ei, torch.stack([d, d])], 1)
    return ei


def sparse_graph(dev, n=12, e=40, seed=42, graphs=1):
    """Random undirected weighted graph(s) without self loops (stands in for pooler_test_graph_sparse[_batch])."""
    g = torch.Generator().manual_seed(seed)
    eis, ews, bs = [], [], []
    for gi in range(graphs):
        a, b = torch.randint(0, n, (e,), generator=g), torch.randint(0, n, (e,), generator=g)
        keep = a != b
        key = torch.unique(torch.minimum(a, b)[keep] * n + torch.maximum(a, b)[keep])
        lo, hi = key // n, key % n
        w = torch.rand(lo.numel(), generator=g) + 0.1
        eis.append(torch.stack([torch.cat([lo, hi]), torch.cat([hi, lo])]) + gi * n)
        ews.append(torch.cat([w, w]))
        bs.append(torch.full((n,), gi, dtype=torch.long))
    x = torch.randn(n * graphs, 5, generator=g)
    return x.to(dev), torch.cat(eis, 1).to(dev), torch.cat(ews).to(dev), torch.cat(bs).to(dev)


def dense_batch(dev, B=3, N=8, F=4, seed=42):
    g = torch.Generator().manual_seed(seed)
    adj = (torch.rand(B, N, N, generator=g) < 0.4).float()
    adj = torch.maximum(adj, adj.transpose(1, 2))
    adj.diagonal(dim1=1, dim2=2).zero_()
    return torch.randn(B, N, F, generator=g).to(dev), adj.to(dev)


def round_robin_s(n, k, dev):
    s = torch.zeros(n, k, device=dev)
    s[torch.arange(n), torch.arange(n) % k] = 1.0
    return s


def to_sparse_unbatched(adj, S):
    B, N, K = S.shape
    b, r, c = adj.nonzero(as_tuple=True)
    ei = torch.stack([r + b * N, c + b * N])
    return ei, adj[b, r, c], S.reshape(B * N, K), torch.arange(B, device=S.device).repeat_interleave(N)


# ------------------------------------------------------------------ tests/reduce/test_base_reduce.py
def test_base_reduce_forward_paths(dev):  # :50-133
    from tgp.reduce import BaseReduce
    from tgp.select import SelectOutput
    x = torch.tensor([[1.0, 0.0], [0.0, 1.0], [2.0, 0.0], [0.0, 2.0]], device=dev)
    s = torch.tensor([[1.0, 0.0], [0.0, 1.0], [1.0, 0.0], [0.0, 1.0]], device=dev)
    so = SelectOutput(s=s, batch=torch.tensor([0, 0, 1, 1], device=dev))
    x_pool, batch_pool = BaseReduce()(x, so, batch=None)
    assert x_pool.shape == (4, 2) and batch_pool.tolist() == [0, 0, 1, 1]
    torch.testing.assert_close(x_pool, torch.tensor([[1.0, 0], [0, 1], [2, 0], [0, 2]], device=dev))
    xb = torch.tensor([[[1.0, 0.0], [0.0, 1.0]], [[2.0, 0.0], [0.0, 2.0]]], device=dev)
    sb = torch.eye(2, device=dev).repeat(2, 1, 1)
    xp, bp = BaseReduce()(xb, SelectOutput(s=sb), batch=None)
    assert xp.shape == (2, 2, 2) and bp is None
    torch.testing.assert_close(xp, xb)
    x2, s2 = torch.tensor([[1.0, 2.0], [3.0, 4.0]], device=dev), torch.eye(2, device=dev)
    xp, bp = BaseReduce()(x2, SelectOutput(s=s2), return_batched=True)
    assert xp.shape == (1, 2, 2) and bp is None
    xp, bp = BaseReduce()(x2, SelectOutput(s=s2), return_batched=False)
    assert xp.shape == (2, 2) and bp is None
    torch.testing.assert_close(xp, x2)


# ------------------------------------------------------------------ tests/lift/test_base_lift.py
def test_lift_dense_unbatched_values(dev):  # :228-237, 266-280, 325-389
    from tgp.lift import BaseLift
    from tgp.select import SelectOutput
    lm = torch.tensor([[1.0, 0.0], [0.5, 0.5], [0.0, 1.0], [1.0, 0.0]], device=dev)
    batch = torch.tensor([0, 0, 1, 1], device=dev)
    so = SelectOutput(s=lm, batch=batch)
    lift = BaseLift(matrix_op="transpose")
    xg = torch.tensor([[1.0, 10.0], [2.0, 20.0]], device=dev)            # K rows: shared by both graphs
    torch.testing.assert_close(lift(xg, so), lm.matmul(xg))
    x4 = torch.tensor([[1.0, 10.0], [2.0, 20.0], [3.0, 30.0], [4.0, 40.0]], device=dev)  # B*K rows
    want = torch.cat([lm[:2].matmul(x4[:2]), lm[2:].matmul(x4[2:])])
    torch.testing.assert_close(lift(x4, so), want)
    torch.testing.assert_close(lift(x4, so, batch_pooled=torch.tensor([0, 0, 1, 1], device=dev)), want)
    x3 = x4.view(2, 2, 2)                                                 # [B, K, F]
    torch.testing.assert_close(lift(x3, so), want)
    torch.testing.assert_close(lift(x3, so, batch_pooled=torch.tensor([0, 0, 1, 1], device=dev)), want)
    lm1 = torch.tensor([[1.0, 0.0], [0.25, 0.75]], device=dev)
    so1 = SelectOutput(s=lm1, batch=torch.zeros(2, dtype=torch.long, device=dev))
    x1 = torch.tensor([[[2.0, 4.0], [6.0, 8.0]]], device=dev)
    torch.testing.assert_close(lift(x1, so1), lm1.matmul(x1.squeeze(0)))
    eye = torch.eye(2, device=dev)
    sop = SelectOutput(s=eye, s_inv=eye.t())
    xp = torch.tensor([[5.0, 6.0], [7.0, 8.0]], device=dev)
    torch.testing.assert_close(BaseLift(matrix_op="precomputed")(xp, sop), xp)


def test_lift_dense_batched_values(dev):  # :296-313, 392-405
    from tgp.lift import BaseLift
    from tgp.select import SelectOutput
    s = torch.tensor([[[1.0, 0, 0], [0, 0, 1.0], [1.0, 0, 0]], [[0, 1.0, 0], [0, 1.0, 0], [0, 1.0, 0]]], device=dev)
    out = BaseLift(matrix_op="transpose")(torch.tensor([[10.0], [20.0], [30.0]], device=dev), SelectOutput(s=s))
    full = torch.tensor([[10.0], [0.0], [20.0], [0.0], [30.0], [0.0]], device=dev).view(2, 3, 1)
    torch.testing.assert_close(out, s.matmul(full))  # compacted rows are expanded through so.out_mask
    s2 = torch.tensor([[[1.0, 0], [0, 1.0], [1.0, 0]], [[0, 1.0], [1.0, 0], [0, 1.0]]], device=dev)
    xp = torch.tensor([[10.0], [20.0], [30.0], [40.0]], device=dev)
    torch.testing.assert_close(BaseLift(matrix_op="transpose")(xp, SelectOutput(s=s2)), s2.matmul(xp.view(2, 2, 1)))


# ------------------------------------------------------------------ tests/connect/test_dense_conn.py
def test_dense_connect_batched_flags(dev):  # :38-166, 198-231
    from tgp.connect import DenseConnect
    from tgp.select import SelectOutput
    x, adj = dense_batch(dev, B=3, N=8)
    k = 4
    s = torch.zeros(3, 8, k, device=dev)
    s[:, torch.arange(8), torch.arange(8) // 2] = 1.0
    so = SelectOutput(s=s)
    raw_like, _ = DenseConnect(edge_weight_norm=False)(adj, so)
    normed, _ = DenseConnect(edge_weight_norm=True)(adj, so)
    assert torch.isfinite(normed).all() and normed.abs().max() <= 1.0 + 1e-6
    for b in range(3):
        m = raw_like[b].abs().max()
        if m > 0:
            torch.testing.assert_close(normed[b], raw_like[b] / m, atol=1e-6, rtol=1e-6)
    no_loops, _ = DenseConnect(remove_self_loops=True)(adj, so)
    assert torch.equal(no_loops.diagonal(dim1=1, dim2=2), torch.zeros(3, k, device=dev))
    loops, _ = DenseConnect(remove_self_loops=False)(adj, so)
    assert not torch.allclose(loops[0].diagonal(), torch.zeros(k, device=dev))
    plain, _ = DenseConnect(degree_norm=False)(adj, so)
    deg, _ = DenseConnect(degree_norm=True)(adj, so)
    assert not torch.allclose(plain, deg) and torch.isfinite(deg).all()
    g = torch.Generator().manual_seed(0)
    nonsym = torch.rand(3, 8, 8, generator=g).to(dev)
    a, _ = DenseConnect(adj_transpose=False)(nonsym, so)
    b_, _ = DenseConnect(adj_transpose=True)(nonsym, so)
    assert a.shape == b_.shape and not torch.allclose(a, b_)
    s_lit = torch.tensor([[1.0, 0.0], [0.0, 1.0], [1.0, 0.0]], device=dev)
    adj_lit = torch.tensor([[0.0, 1.0, 2.0], [1.0, 0.0, 3.0], [2.0, 3.0, 0.0]], device=dev)
    out = DenseConnect().dense_connect(adj=adj_lit, s=s_lit)
    torch.testing.assert_close(out, DenseConnect._dense_connect(s_lit.unsqueeze(0), adj_lit.unsqueeze(0)))
    assert out.shape == (1, 2, 2) and out[0].tolist() == [[4.0, 4.0], [4.0, 0.0]]


def test_dense_connect_unbatched_outputs(dev):  # :260-348, 351-372, 425-441, 463-560
    from tgp.connect import DenseConnect
    from tgp.reduce import BaseReduce
    from tgp.select import SelectOutput
    x, ei, ew, _ = sparse_graph(dev)
    n, k = x.size(0), x.size(0) // 2
    so = SelectOutput(s=round_robin_s(n, k, dev))
    conn = DenseConnect(remove_self_loops=False, degree_norm=False, sparse_output=True)
    adj_pool, w = conn(edge_index=ei, edge_weight=ew, so=so)
    assert not adj_pool.is_sparse and adj_pool.size(0) == 2 and w is not None and w.numel() == adj_pool.size(1)
    coo = torch.sparse_coo_tensor(ei, ew, size=(n, n)).coalesce()
    adj_coo, w_coo = conn(edge_index=coo, edge_weight=None, so=so)
    assert adj_coo.is_sparse and w_coo is None and adj_coo.shape == (k, k)
    dense, w_d = DenseConnect(sparse_output=False)(edge_index=ei, edge_weight=ew, so=so)
    assert dense.shape == (1, k, k) and w_d is None
    # only self loops + remove_self_loops + degree_norm: an empty pooled graph, still a [2, E'] index
    so2 = SelectOutput(s=round_robin_s(2, 2, dev))
    empty, _ = DenseConnect(remove_self_loops=True, degree_norm=True, sparse_output=True)(
        edge_index=torch.zeros(2, 2, dtype=torch.long, device=dev), edge_weight=torch.ones(2, device=dev), so=so2)
    assert empty.size(0) == 2
    with pytest.raises(AssertionError, match="batch_pooled parameter is required"):
        DenseConnect(edge_weight_norm=True, sparse_output=True)(edge_index=ei, edge_weight=ew, so=so, batch_pooled=None)
    # no edges at all: zeros, single graph and multi graph
    so4 = SelectOutput(s=round_robin_s(4, 2, dev))
    none = torch.empty((2, 0), dtype=torch.long, device=dev)
    z1, w1 = DenseConnect(sparse_output=False, remove_self_loops=False)(edge_index=none, edge_weight=None, so=so4, batch=None)
    z2, w2 = DenseConnect(sparse_output=False, remove_self_loops=False)(
        edge_index=none, edge_weight=None, so=so4, batch=torch.tensor([0, 0, 1, 1], device=dev))
    assert w1 is None and w2 is None and z1.shape == (1, 2, 2) and z2.shape == (2, 2, 2)
    assert not z1.any() and not z2.any()
    # multi-graph batch: [B, K, K]; with edge_weight_norm and batch_pooled the block output is normalised per graph
    xb, eib, ewb, bb = sparse_graph(dev, graphs=3)
    sob = SelectOutput(s=round_robin_s(xb.size(0), 2, dev))
    multi, wm = DenseConnect(sparse_output=False)(edge_index=eib, edge_weight=ewb, so=sob, batch=bb)
    assert multi.shape == (3, 2, 2) and wm is None
    bp = torch.arange(3, device=dev).repeat_interleave(2)
    eo, wo = DenseConnect(edge_weight_norm=True, sparse_output=True)(edge_index=eib, edge_weight=ewb, so=sob, batch=bb,
                                                                      batch_pooled=bp)
    assert eo.size(0) == 2 and wo.numel() == eo.size(1) and wo.abs().max() <= 1.0 + 1e-6
    assert torch.equal(BaseReduce.reduce_batch(sob, bb), bp)


# ------------------------------------------------------------------ tests/connect/test_base_conn.py
def test_sparse_connect_paths(dev):  # :62-259
    from tgp.connect import SparseConnect, sparse_connect
    from tgp.select import SelectOutput
    x, ei, ew, _ = sparse_graph(dev)
    n, k = x.size(0), x.size(0) // 2
    kept = torch.arange(k, device=dev)
    so_top = SelectOutput(node_index=kept, cluster_index=torch.arange(k, device=dev), num_nodes=n, num_supernodes=k)
    adj, w = SparseConnect()(edge_index=ei, edge_weight=ew, so=so_top)
    assert adj.size(0) == 2 and w.numel() == adj.size(1) and (adj.numel() == 0 or int(adj.max()) < k)
    inside = (ei[0] < k) & (ei[1] < k)
    assert torch.equal(adj, ei[:, inside]) and torch.equal(w, ew[inside])  # subgraph keeps input order, relabel = identity
    cluster = torch.arange(n, device=dev) // 2
    so_cl = SelectOutput(cluster_index=cluster, num_nodes=n, num_supernodes=k)
    adj_c, w_c = SparseConnect()(edge_index=ei, edge_weight=ew, so=so_cl)
    assert adj_c.size(0) == 2 and int(adj_c.max()) < k and (adj_c[0] != adj_c[1]).all()
    key = adj_c[0] * k + adj_c[1]
    assert (key[1:] > key[:-1]).all()  # coalesced: sorted, unique
    chain_ei = chain(4, dev)
    cl = torch.tensor([0, 0, 1, 1], device=dev)
    a1, w1 = sparse_connect(edge_index=chain_ei, edge_weight=None, cluster_index=cl, num_nodes=4, num_supernodes=2,
                            degree_norm=True, remove_self_loops=True)
    assert w1 is not None and w1.size(0) == a1.size(1) and (w1 >= 0).all() and (w1 <= 1).all()
    a2, w2 = SparseConnect(degree_norm=True)(edge_index=chain_ei, edge_weight=None,
                                             so=SelectOutput(cluster_index=cl, num_nodes=4, num_supernodes=2))
    assert torch.equal(a1, a2) and torch.equal(w1, w2)
    coo = torch.sparse_coo_tensor(chain_ei, torch.ones(6, device=dev), size=(4, 4)).coalesce()
    a3, w3 = sparse_connect(edge_index=coo, edge_weight=None, cluster_index=cl, num_nodes=4, num_supernodes=2,
                            degree_norm=False, remove_self_loops=True)
    assert a3.is_sparse and w3 is None and a3.shape == (2, 2)
    a4, w4 = SparseConnect()(edge_index=coo, edge_weight=None, so=SelectOutput(cluster_index=cl, num_nodes=4, num_supernodes=2))
    assert a4.is_sparse and w4 is None and a4.shape == (2, 2)


# ------------------------------------------------------------------ tests/selection/test_mlp_select.py
def test_mlp_select_forward(dev):  # :7-79
    from tgp.select import MLPSelect, SelectOutput
    torch.manual_seed(0)
    sel = MLPSelect(in_channels=3, k=2, batched_representation=True).to(dev)
    out = sel(x=torch.randn(4, 3, device=dev), mask=None)
    assert isinstance(out, SelectOutput) and out.s.shape == (1, 4, 2) and out.in_mask is None
    out = sel(x=torch.randn(2, 4, 3, device=dev), mask=None)
    assert out.s.shape == (2, 4, 2) and out.in_mask is None
    torch.testing.assert_close(out.s.sum(-1), torch.ones(2, 4, device=dev), atol=1e-6, rtol=1e-6)
    mask = torch.tensor([[True, True, False, False], [True, False, True, False]], device=dev)
    out = sel(x=torch.randn(2, 4, 3, device=dev), mask=mask)
    assert torch.equal(out.in_mask, mask) and (out.s[~mask] == 0).all()
    torch.testing.assert_close(out.s[mask].sum(-1), torch.ones(4, device=dev), atol=1e-6, rtol=1e-6)
    batch = torch.tensor([0, 0, 0, 1, 1, 1], device=dev)
    un = MLPSelect(in_channels=[3, 4], k=2, batched_representation=False, act="relu").to(dev)
    out = un(x=torch.randn(6, 3, device=dev), batch=batch)
    assert out.s.shape == (6, 2) and torch.equal(out.batch, batch)


# ------------------------------------------------------------------ tests/poolers/test_diffpool.py, test_mincut.py
@pytest.mark.parametrize("alias,loss_keys", [("diff", {"link_loss", "entropy_loss"}), ("mincut", {"cut_loss", "ortho_loss"})])
def test_dense_poolers_reference_behaviour(dev, alias, loss_keys):
    from tgp.poolers import DiffPool, MinCutPooling, get_pooler
    cls = DiffPool if alias == "diff" else MinCutPooling
    p = cls(in_channels=16, k=5)
    assert p.batched is True
    if alias == "diff":  # test_diffpool.py:9-25
        assert p.link_loss_coeff == 1.0 and p.ent_loss_coeff == 1.0
        q = DiffPool(in_channels=16, k=5, link_loss_coeff=0.5, ent_loss_coeff=2.0, batched=False)
        assert (q.link_loss_coeff, q.ent_loss_coeff, q.batched) == (0.5, 2.0, False)
    else:                # test_mincut.py:9-27, 225-238
        q = MinCutPooling(in_channels=8, k=4, batched=False, cut_loss_coeff=0.25, ortho_loss_coeff=2.0)
        extra = q.extra_repr_args()
        assert extra["batched"] is False and extra["cut_loss_coeff"] == 0.25 and extra["ortho_loss_coeff"] == 2.0
    x, adj = dense_batch(dev)
    B, N, F = x.shape
    k = 3
    torch.manual_seed(0)
    pooler = cls(in_channels=F, k=k, batched=True).to(dev)
    out = pooler(x=x, adj=adj)                                  # batched forward
    assert out.x.shape == (B, k, F) and out.edge_index.shape == (B, k, k) and set(out.loss) == loss_keys
    lifted = pooler(x=out.x, so=out.so, lifting=True)           # lifting
    assert lifted.shape == x.shape
    sp = cls(in_channels=F, k=k, batched=True, sparse_output=True).to(dev)
    o = sp(x=x, adj=adj)                                        # batched, sparse output
    assert o.x.dim() == 2 and o.x.shape[1] == F and o.edge_index.dim() == 2 and o.edge_index.shape[0] == 2
    assert o.edge_weight.dim() == 1 and o.batch.shape[0] == o.x.shape[0] and set(o.loss) == loss_keys
    for train in (True, False):                                 # gradients reach the inputs in both modes
        pooler.train(train)
        xg = x.detach().clone().requires_grad_(True)
        sum(pooler(x=xg, adj=adj).loss.values()).backward()
        assert xg.grad is not None and torch.isfinite(xg.grad).all()
    pooler.eval()                                               # batched losses == sparse losses on the same data
    with torch.no_grad():
        out = pooler(x=x, adj=adj)
        ei, ew, s_flat, batch = to_sparse_unbatched(adj, out.so.s)
        sparse_loss = pooler.compute_sparse_loss(ei, ew, s_flat, batch)
    assert set(sparse_loss) == set(out.loss)
    for key in out.loss:
        torch.testing.assert_close(out.loss[key], sparse_loss[key], **TOL)
    g = torch.Generator().manual_seed(42)                       # unbatched: single graph, two graphs, sparse output
    un = cls(in_channels=16, k=5, batched=False).to(dev)
    x1 = torch.randn(10, 16, generator=g).to(dev)
    e1 = torch.randint(0, 10, (2, 30), generator=g).to(dev)
    o1 = un(x=x1, adj=e1)
    assert o1.x.shape == (1, 5, 16) and o1.edge_index.shape == (1, 5, 5) and set(o1.loss) == loss_keys
    x2 = torch.randn(25, 16, generator=g).to(dev)
    b2 = torch.tensor([0] * 10 + [1] * 15, device=dev)
    e2 = torch.cat([torch.randint(0, 10, (2, 20), generator=g), torch.randint(10, 25, (2, 30), generator=g)], 1).to(dev)
    o2 = un(x=x2, adj=e2, batch=b2)
    assert o2.x.shape == (2, 5, 16) and o2.edge_index.shape == (2, 5, 5) and o2.loss is not None
    us = cls(in_channels=16, k=5, batched=False, sparse_output=True).to(dev)
    o3 = us(x=x2, adj=e2, batch=b2)
    assert o3.x.dim() == 2 and o3.x.shape[1] == 16 and o3.batch.shape[0] == o3.x.shape[0]
    assert o3.edge_index.dim() == 2 and o3.edge_index.shape[0] == 2
    pu = get_pooler(alias + "_u", in_channels=16, k=5).to(dev)  # "_u" alias
    assert pu.batched is False
    ou = pu(x=x1, adj=e1)
    assert ou.x is not None and ou.loss is not None


# ------------------------------------------------------------------ tests/poolers/test_topk.py
def test_topk_reference_behaviour(dev):  # :22-112, 114-232
    from tgp.poolers import TopkPooling
    from tgp.select import SelectOutput, TopkSelect
    from tgp.src import PoolingOutput
    sel = TopkSelect(in_channels=1, ratio=0.5, min_score=None, act="linear", s_inv_op="transpose").to(dev)
    out = sel.forward(x=torch.arange(1.0, 6, device=dev).unsqueeze(-1), batch=None)
    assert out.node_index.sort(descending=True)[0].tolist() == [4, 3, 2]
    assert "ratio=0.5" in repr(sel) and "min_score" not in repr(sel)
    torch.manual_seed(0)
    sel = TopkSelect(in_channels=4, ratio=0.5, act="tanh").to(dev).eval()
    out = sel(x=torch.randn(6, 4, device=dev), batch=None)
    assert (out.num_supernodes, out.num_nodes, out.node_index.size(0)) == (3, 6, 3)
    assert torch.allclose(out.s.to_dense(), out.s_inv.to_dense().t())
    torch.manual_seed(1)
    sel = TopkSelect(in_channels=2, ratio=0.5, min_score=0.2, act="tanh").to(dev).eval()
    out = sel(x=0.01 * torch.randn(4, 2, device=dev), batch=torch.tensor([0, 0, 1, 1], device=dev))
    assert out.node_index.size(0) == 4 and "min_score=0.2" in repr(sel)  # softmax over 2 near-equal scores: both > 0.2
    torch.manual_seed(2)
    x = torch.randn(6, 5, device=dev)
    ei = chain(6, dev)
    pooler = TopkPooling(in_channels=5, ratio=0.5, min_score=None, nonlinearity="linear", lift="transpose",
                         s_inv_op="transpose", connect_red_op="sum", lift_red_op="sum").to(dev).eval()
    out = pooler(x=x, adj=ei, edge_weight=None, so=None, batch=None, attn=None, lifting=False)
    assert isinstance(out, PoolingOutput) and out.x.shape == (3, 5) and isinstance(out.edge_index, torch.Tensor)
    assert isinstance(out.so, SelectOutput)
    lifted = pooler(x=out.x, adj=None, so=out.so, batch=None, attn=None, lifting=True)
    assert lifted.shape == (6, 5)
    r, c = torch.meshgrid(torch.arange(4, device=dev), torch.arange(4, device=dev), indexing="ij")
    full = torch.stack([r.flatten(), c.flatten()])
    p1 = TopkPooling(in_channels=2, ratio=0.5, min_score=0.1, nonlinearity="tanh").to(dev).eval()
    o1 = p1(x=torch.ones(4, 2, device=dev), adj=full, edge_weight=None, so=None, batch=None, attn=None, lifting=False)
    assert o1.so.num_supernodes == 4 and o1.x.shape == (4, 2)  # uniform softmax 1/4 > 0.1: everything is kept
    sel1 = TopkSelect(in_channels=1, ratio=0.5, act="linear").to(dev)
    with pytest.raises(AssertionError):
        sel1(x=torch.randn(10, 2, device=dev))
    o = sel1(x=torch.randn(10, device=dev))
    assert (o.node_index.size(0), o.num_supernodes, o.num_nodes, o.s.size(0)) == (5, 5, 10, 10)


# ------------------------------------------------------------------ tests/poolers/test_graclus.py, test_ndp.py
def test_graclus_and_ndp_reference_behaviour(dev):
    from tgp.poolers import GraclusPooling, NDPPooling
    from tgp.select import SelectOutput
    from tgp.src import PoolingOutput
    x, ei, ew, _ = sparse_graph(dev, n=14)
    batch = torch.zeros(x.size(0), dtype=torch.long, device=dev)
    pooler = GraclusPooling(s_inv_op="inverse").eval()
    out = pooler(x=x, adj=ei, edge_weight=ew, batch=batch, lifting=False)       # test_graclus.py:10-27
    assert isinstance(out, PoolingOutput) and len(repr(out)) > 0 and isinstance(next(iter(out)), torch.Tensor)
    assert out.has_loss is False and out.get_loss_value() == 0.0
    assert pooler.get_forward_signature() is not None and pooler.data_transforms() is None
    cached = GraclusPooling(cached=True).eval()                                   # :30-47
    o1 = cached(x=x, adj=ei, edge_weight=ew, batch=batch, lifting=False)
    o2 = cached(x=x, adj=ei, edge_weight=ew, batch=batch, lifting=False)
    assert cached.cached is True and isinstance(cached._so_cached, SelectOutput) and o1.x.equal(o2.x)
    o3 = pooler(x=x, adj=ei, edge_weight=torch.ones(ei.size(1), 1, device=dev), batch=batch)   # :50-71 [E,1] weights
    assert o3.edge_index is not None and o3.edge_weight is not None
    with pytest.raises(RuntimeError):
        pooler(x=x, adj=ei, edge_weight=torch.ones(ei.size(1), 2, device=dev), batch=batch)
    ndp = NDPPooling().eval()                                                     # test_ndp.py:9-19 via the pooler
    on = ndp(x=x, adj=ei, edge_weight=ew, batch=batch)
    assert on.so.s.size(0) == x.size(0) and hasattr(on.so, "L") and 1 <= on.so.num_supernodes < x.size(0)
    assert on.x.shape == (on.so.num_supernodes, x.size(1))


# ------------------------------------------------------------------ tests/test_src.py
def test_dense_src_preprocessing_and_finalize(dev):  # :105-144, 199-256
    from tgp.poolers import MinCutPooling
    from tgp.select import SelectOutput
    from tgp.src import DenseSRCPooling
    x = torch.randn(4, 5, device=dev)
    ei = chain(4, dev)
    ew = torch.ones(ei.size(1), 1, device=dev)  # trailing feature dimension on the weights
    batch = torch.zeros(4, dtype=torch.long, device=dev)
    pooler = MinCutPooling(k=2, in_channels=5).to(dev)
    xb, adj, mask = pooler.preprocessing(edge_index=ei, edge_weight=ew, x=x, batch=batch)
    assert adj.dim() == 3 and xb.shape == (1, 4, 5) and mask.shape == (1, 4)
    cache = DenseSRCPooling(cache_preprocessing=True)
    _, first, _ = cache.preprocessing(x=x, edge_index=ei, edge_weight=ew.view(-1), batch=batch, use_cache=True)
    assert cache.preprocessing_cache is not None
    cache.preprocessing_cache = torch.full_like(first, 7.0)
    _, second, _ = cache.preprocessing(x=x, edge_index=ei, edge_weight=ew.view(-1), batch=batch, use_cache=True)
    torch.testing.assert_close(second, torch.full_like(first, 7.0))
    p = MinCutPooling(in_channels=2, k=3, sparse_output=True).to(dev)
    so2 = SelectOutput(s=torch.eye(2, device=dev).repeat(2, 1, 1))
    xo, eo, wo, bp = p._finalize_sparse_output(x_pool=torch.randn(2, 2, 2, device=dev), adj_pool=torch.eye(2, device=dev).repeat(2, 1, 1),
                                               batch=torch.tensor([0, 0, 1, 1], device=dev), batch_pooled=None, so=so2)
    assert bp.numel() == xo.size(0) and eo.size(0) == 2 and wo.numel() == eo.size(1)
    so1 = SelectOutput(s=torch.tensor([[[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]], device=dev))
    xo, _, _, bp = p._finalize_sparse_output(x_pool=torch.randn(1, 3, 2, device=dev), adj_pool=torch.eye(3, device=dev).unsqueeze(0),
                                             batch=None, batch_pooled=None, so=so1)
    assert bp is not None and (bp == 0).all() and bp.numel() == xo.size(0)
    so_sp = SelectOutput(cluster_index=torch.tensor([0, 1, 0], device=dev), num_supernodes=2)
    xo, eo, wo, bp = p._finalize_sparse_output(x_pool=torch.randn(1, 2, 2, device=dev),
                                               adj_pool=torch.tensor([[[1.0, 0.5], [0.5, 1.0]]], device=dev),
                                               batch=None, batch_pooled=None, so=so_sp)
    assert bp is None and xo.shape == (2, 2) and eo.shape[0] == 2 and wo.numel() == eo.shape[1]


def test_kron_connect_device_solve_matches_scipy_route(dev):
    """A9 / N4: the dense fp64 Kron reduction on the GPU gives the edge set and weights of the reference's scipy
    route (connect/kron_conn.py:117-146) - with and without a selector-provided Laplacian, COO in / COO out, a
    single kept node, and the MIS error path of tests/connect/test_kron_conn.py:230-255."""
    import warnings
    from tgp.connect import KronConnect
    from tgp.select import NDPSelect, SelectOutput
    x, ei, ew, _ = sparse_graph(dev, n=120, e=500, seed=7)
    n = x.size(0)
    dense, host = KronConnect(), KronConnect(dense_solve_max_nodes=0)
    so_ndp = NDPSelect()(edge_index=ei, edge_weight=ew, num_nodes=n)
    kept = torch.arange(0, n, 2, device=dev)
    so_plain = SelectOutput(node_index=kept, cluster_index=torch.arange(kept.numel(), device=dev), num_nodes=n,
                            num_supernodes=kept.numel())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for so in (so_ndp, so_plain):
            a1, w1 = dense(edge_index=ei, so=so, edge_weight=ew)
            a2, w2 = host(edge_index=ei, so=so, edge_weight=ew)
            assert a1.device.type == "cuda" and torch.equal(a1, a2) and w1.dtype == torch.float32
            torch.testing.assert_close(w1, w2, rtol=1e-5, atol=1e-6)
        coo = torch.sparse_coo_tensor(ei, ew, size=(n, n)).coalesce()
        ac, wc = dense(edge_index=coo, so=so_plain, edge_weight=None)
        assert ac.is_sparse and wc is None and ac.shape == (kept.numel(), kept.numel())
        one = SelectOutput(node_index=torch.tensor([3], device=dev), cluster_index=torch.tensor([0], device=dev),
                           num_nodes=n, num_supernodes=1)
        a_one, w_one = dense(edge_index=ei, so=one, edge_weight=ew)
        assert a_one.size(0) == 2 and a_one.size(1) == 0  # the 1x1 "-1" Laplacian leaves no off-diagonal edge
        lo, _ = KronConnect(sparse_threshold=0.0)(edge_index=ei, so=so_plain, edge_weight=ew)
        hi, _ = KronConnect(sparse_threshold=10.0)(edge_index=ei, so=so_plain, edge_weight=ew)
        assert lo.size(1) >= hi.size(1)
        bad = SelectOutput(num_nodes=3, num_supernodes=2, node_index=torch.arange(3, device=dev),
                           cluster_index=torch.tensor([0, 0, 1], device=dev), mis=torch.tensor([0, 5], device=dev))
        with pytest.raises(ValueError, match="MIS indices out of range"):
            dense(edge_index=torch.tensor([[0, 1, 1, 2], [1, 0, 2, 1]], device=dev), so=bad, edge_weight=None)
