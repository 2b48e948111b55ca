"""GPU parity: the HIP path (through the C ABI, via the tgp host mirror) against the golden vectors
produced by the reference (tests/golden/golden_v1.pt).

Bars: index / integer outputs bit-exact; fp32 rtol = atol = 1e-5 (north_star; the reference's own
tolerance in tests/poolers/test_dense_poolers_batched_vs_unbatched.py:124-171).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL = ATOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def D(v, dev):
    if isinstance(v, torch.Tensor):
        return v.to(dev)
    if isinstance(v, dict) and v.get("__coo__"):
        return torch.sparse_coo_tensor(v["indices"], v["values"], v["size"]).coalesce().to(dev)
    return v


def close(a, b, msg=""):
    a = a.detach().cpu()
    assert a.shape == b.shape, f"{msg}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    torch.testing.assert_close(a, b, rtol=RTOL, atol=ATOL, msg=lambda m: f"{msg}: {m}")


def exact(a, b, msg=""):
    if b is None:
        assert a is None, msg
        return
    a = a.detach().cpu()
    assert a.dtype == b.dtype, f"{msg}: dtype {a.dtype} vs {b.dtype}"
    assert a.shape == b.shape and torch.equal(a, b), f"{msg}: {a} vs {b}"


def check_output(out, e, name):
    if e["edge_index"] is None:
        assert out.edge_index is None
    elif isinstance(e["edge_index"], dict):  # torch COO adjacency
        got = out.edge_index.coalesce()
        exact(got.indices(), e["edge_index"]["indices"], name + ".coo.indices")
        close(got.values(), e["edge_index"]["values"], name + ".coo.values")
    elif e["edge_index"].dtype == torch.long:
        exact(out.edge_index, e["edge_index"], name + ".edge_index")
    else:
        close(out.edge_index, e["edge_index"], name + ".adj")
    if e["edge_weight"] is None:
        assert out.edge_weight is None, name + ".edge_weight"
    else:
        close(out.edge_weight, e["edge_weight"], name + ".edge_weight")
    exact(out.batch, e["batch"], name + ".batch")
    if e["x"] is not None:
        close(out.x, e["x"], name + ".x")
    if "loss" in e:
        for k, v in e["loss"].items():
            close(out.loss[k], v, f"{name}.loss.{k}")
    if "mask" in e:
        exact(out.mask, e["mask"], name + ".mask")


def check_so(so, e, name):
    assert so.num_nodes == e["num_nodes"] and so.num_supernodes == e["num_supernodes"], name
    if "node_index" in e:
        exact(so.node_index, e["node_index"], name + ".node_index")
        exact(so.cluster_index, e["cluster_index"], name + ".cluster_index")
        close(so.weight, e["weight"], name + ".weight")
    else:
        close(so.s, e["s"], name + ".s")


# ----------------------------------------------------------------------------------- TopK
def test_topk_poolers(golden, dev):
    from tgp.poolers import get_pooler
    names = [k for k in golden if k.startswith("topk_batch") or k == "c1_topk_er100"]
    assert len(names) == 15
    for name in names:
        c = golden[name]
        pooler = get_pooler("topk", **c["cfg"]).to(dev).eval()
        pooler.load_state_dict({k: v for k, v in c["params"].items()})
        i = c["inputs"]
        with torch.no_grad():
            out = pooler(x=D(i["x"], dev), adj=D(i["edge_index"], dev), edge_weight=D(i["edge_weight"], dev),
                         batch=D(i["batch"], dev))
        check_so(out.so, c["expected"]["so"], name)
        check_output(out, c["expected"], name)


def test_topk_coo_adjacency(golden, dev):
    from tgp.poolers import get_pooler
    c = golden["topk_coo_adj"]
    pooler = get_pooler("topk", **c["cfg"]).to(dev).eval()
    pooler.load_state_dict(c["params"])
    with torch.no_grad():
        out = pooler(x=D(c["inputs"]["x"], dev), adj=D(c["inputs"]["adj_coo"], dev),
                     batch=D(c["inputs"]["batch"], dev))
    assert out.edge_index.is_sparse
    check_output(out, c["expected"], "topk_coo")


# ----------------------------------------------------------------------------------- Graclus-style
def test_cluster_reduce_connect(golden, dev):
    """Reduce + Connect for one-over-K poolers given the reference's SelectOutput (selectors are
    non-deterministic in the reference)."""
    from tgp.connect import SparseConnect
    from tgp.reduce import BaseReduce
    from tgp.select import SelectOutput
    names = [k for k in golden if k.startswith("graclus_") and "precoarsen" not in k]
    assert len(names) == 14
    for name in names:
        c = golden[name]
        i, e, cfg = c["inputs"], c["expected"], dict(c["cfg"])
        so = SelectOutput(cluster_index=D(e["so"]["cluster_index"], dev), num_nodes=e["so"]["num_nodes"],
                          num_supernodes=e["so"]["num_supernodes"])
        x_pool, batch_pool = BaseReduce()(D(i["x"], dev), so, batch=D(i["batch"], dev))
        conn = SparseConnect(reduce_op=cfg.pop("connect_red_op"), **cfg)
        ei, ew = conn(D(i["edge_index"], dev), so, edge_weight=D(i["edge_weight"], dev), batch_pooled=batch_pool)
        close(x_pool, e["x"], name + ".x")
        exact(batch_pool, e["batch"], name + ".batch")
        exact(ei, e["edge_index"], name + ".edge_index")
        if e["edge_weight"] is None:
            assert ew is None, name
        else:
            close(ew, e["edge_weight"], name + ".edge_weight")


def test_many_to_one_clusters(golden, dev):
    from tgp.connect import SparseConnect, sparse_connect
    from tgp.reduce import BaseReduce
    from tgp.select import SelectOutput
    c = golden["cluster_many_to_one"]
    i, e = c["inputs"], c["expected"]
    so = SelectOutput(cluster_index=D(i["cluster_index"], dev), num_nodes=40, num_supernodes=i["num_supernodes"],
                      weight=D(i["weight"], dev))
    xp, bp = BaseReduce()(D(i["x"], dev), so, batch=torch.zeros(40, dtype=torch.long, device=dev))
    ei, ew = SparseConnect()(D(i["edge_index"], dev), so, edge_weight=D(i["edge_weight"], dev))
    close(xp, e["x"])
    exact(bp, e["batch"])
    exact(ei, e["edge_index"])
    close(ew, e["edge_weight"])
    c = golden["chain4_degree_norm_noweights"]
    ei, ew = sparse_connect(D(c["inputs"]["edge_index"], dev), None, cluster_index=D(c["inputs"]["cluster_index"], dev),
                            num_nodes=4, num_supernodes=2, degree_norm=True, remove_self_loops=True)
    exact(ei, c["expected"]["edge_index"])
    close(ew, c["expected"]["edge_weight"])


def test_graclus_pooler_end_to_end(golden, dev):
    """The in-package matching is a different (valid) matching than the fixture's; check the contract:
    clusters have 1-2 adjacent nodes, ids are consecutive, and Reduce/Connect agree with the oracle run on
    the SAME assignment."""
    import tgp_oracle as O
    from tgp.poolers import get_pooler
    c = golden["graclus_w_sum_default"]
    i = c["inputs"]
    pooler = get_pooler("graclus")
    out = pooler(x=D(i["x"], dev), adj=D(i["edge_index"], dev), edge_weight=D(i["edge_weight"], dev),
                 batch=D(i["batch"], dev))
    cl = out.so.cluster_index.cpu()
    n = i["x"].size(0)
    counts = torch.bincount(cl)
    assert counts.min() >= 1 and counts.max() <= 2 and counts.numel() == out.so.num_supernodes
    adj = torch.zeros(n, n, dtype=torch.bool)
    adj[i["edge_index"][0], i["edge_index"][1]] = True
    for k in (counts == 2).nonzero().view(-1).tolist():
        a, b = (cl == k).nonzero().view(-1).tolist()
        assert adj[a, b] or adj[b, a]
    ref = O.cluster_pool(i["x"], i["edge_index"], i["edge_weight"], i["batch"], cl, out.so.num_supernodes)
    close(out.x, ref["x"])
    exact(out.edge_index, ref["edge_index"])
    close(out.edge_weight, ref["edge_weight"])
    exact(out.batch, ref["batch"])


def test_graclus_precoarsening(golden, dev):
    from tgp.poolers import get_pooler
    from tgp.select import SelectOutput
    for wtag in ("w", "u"):
        c = golden[f"graclus_precoarsen_{wtag}"]
        i, e = c["inputs"], c["expected"]
        pooler = get_pooler("graclus")
        so = SelectOutput(cluster_index=D(e["so"]["cluster_index"], dev), num_nodes=e["so"]["num_nodes"],
                          num_supernodes=e["so"]["num_supernodes"])
        out = pooler._precoarsening_from_select_output(so=so, edge_index=D(i["edge_index"], dev),
                                                       edge_weight=D(i["edge_weight"], dev), batch=D(i["batch"], dev))
        check_output(out, e, "precoarsen")
        levels = pooler.multi_level_precoarsening(2, edge_index=D(i["edge_index"], dev),
                                                  edge_weight=D(i["edge_weight"], dev), batch=D(i["batch"], dev),
                                                  num_nodes=i["num_nodes"])
        assert len(levels) == 2 and levels[1].so.num_nodes == levels[0].so.num_supernodes


# ----------------------------------------------------------------------------------- NDP
def test_ndp_reduce_and_kron(golden, dev):
    import tgp_oracle as O
    from tgp.connect import KronConnect
    from tgp.reduce import BaseReduce
    from tgp.select import SelectOutput
    for wtag in ("w", "u"):
        c = golden[f"ndp_kron_{wtag}"]
        i, e = c["inputs"], c["expected"]
        n, ip = i["x"].size(0), i["idx_pos"]
        s = torch.sparse_coo_tensor(torch.stack([ip, torch.arange(ip.numel())]), torch.ones(ip.numel()),
                                    (n, ip.numel())).coalesce().to(dev)
        L = O.laplacian_scipy(i["edge_index"], i["edge_weight"], n)
        so = SelectOutput(s=s, L=L)
        xp, bp = BaseReduce()(D(i["x"], dev), so, batch=D(i["batch"], dev))
        close(xp, e["x"])
        exact(bp, e["batch"])
        ei, ew = KronConnect()(D(i["edge_index"], dev), so, edge_weight=D(i["edge_weight"], dev))
        exact(ei, e["edge_index"])
        close(ew, e["edge_weight"])
        with pytest.warns(UserWarning):
            ei2, ew2 = KronConnect()(D(i["edge_index"], dev), SelectOutput(s=s), edge_weight=D(i["edge_weight"], dev))
        e2 = golden[f"ndp_kron_nolap_{wtag}"]["expected"]
        exact(ei2, e2["edge_index"])
        close(ew2, e2["edge_weight"])


def test_ndp_pooler_runs(golden, dev):
    from tgp.poolers import get_pooler
    i = golden["ndp_kron_w"]["inputs"]
    out = get_pooler("ndp")(x=D(i["x"], dev), adj=D(i["edge_index"], dev), edge_weight=D(i["edge_weight"], dev),
                            batch=D(i["batch"], dev))
    k = out.so.num_supernodes
    assert 0 < k < i["x"].size(0) and out.x.shape == (k, i["x"].size(1))
    assert out.edge_index.max() < k and out.batch.numel() == k


# ----------------------------------------------------------------------------------- dense poolers
def test_dense_poolers(golden, dev):
    from tgp.poolers import get_pooler
    names = [k for k in golden if k.split("_")[0] in ("diff", "mincut")]
    assert len(names) == 38
    for name in names:
        c = golden[name]
        alias = name.split("_")[0]
        if "_unbatched_" in name or "_u_" in name:
            alias += "_u"
        pooler = get_pooler(alias, **c["cfg"]).to(dev).eval()
        pooler.load_state_dict(c["params"])
        i = c["inputs"]
        with torch.no_grad():
            if "adj" in i:
                out = pooler(x=D(i["x"], dev), adj=D(i["adj"], dev), mask=D(i["mask"], dev))
            else:
                out = pooler(x=D(i["x"], dev), adj=D(i["edge_index"], dev), edge_weight=D(i["edge_weight"], dev),
                             batch=D(i["batch"], dev))
        check_so(out.so, c["expected"]["so"], name)
        check_output(out, c["expected"], name)


def test_dense_poolers_train_mode_matches_eval(golden, dev):
    """With gradients enabled the Connect path switches to autograd Functions; values must not change,
    and gradients must reach the selector's parameters."""
    from tgp.poolers import get_pooler
    for name in ("diff_batched_default_w", "mincut_batched_default_w"):
        c = golden[name]
        pooler = get_pooler(name.split("_")[0], **c["cfg"]).to(dev)
        pooler.load_state_dict(c["params"])
        i = c["inputs"]
        out = pooler(x=D(i["x"], dev), adj=D(i["edge_index"], dev), edge_weight=D(i["edge_weight"], dev),
                     batch=D(i["batch"], dev))
        check_output(out, c["expected"], name + ".train")
        (out.x.sum() + out.edge_index.sum() + sum(out.loss.values())).backward()
        g = pooler.selector.mlp.lins[0].weight.grad
        assert g is not None and torch.isfinite(g).all() and g.abs().sum() > 0


# ----------------------------------------------------------------------------------- operators
def test_dense_ops_grid(golden, dev):
    from tgp.connect import DenseConnect
    from tgp.reduce import BaseReduce
    from tgp.select import SelectOutput
    c = golden["dense_ops_grid"]
    S, A, X = (D(c["inputs"][k], dev) for k in ("S", "A", "X"))
    e = c["expected"]
    xp, _ = BaseReduce()(X, SelectOutput(s=S))
    close(xp, e["x_pool"])
    close(DenseConnect().dense_connect(adj=A, s=S), e["raw"])
    for key, val in e.items():
        if not key.startswith("rsl"):
            continue
        f = {p[:-1]: bool(int(p[-1])) for p in key.split("_")}
        conn = DenseConnect(remove_self_loops=f["rsl"], degree_norm=f["dn"], adj_transpose=f["at"],
                            edge_weight_norm=f["ewn"])
        out, w = conn(A.clone(), SelectOutput(s=S))
        assert w is None
        close(out, val, key)
        # the transposed-view layout the preprocessing hands over must give the same numbers
        out_t, _ = conn(A.transpose(1, 2).contiguous().transpose(1, 2), SelectOutput(s=S))
        close(out_t, val, key + ".tview")


def test_dense_literals(golden, dev):
    from tgp.connect import DenseConnect
    from tgp.utils.ops import get_mask_from_dense_s, postprocess_adj_pool_dense, postprocess_adj_pool_sparse
    c = golden["dense_connect_literal"]
    out = DenseConnect().dense_connect(adj=D(c["inputs"]["adj"], dev), s=D(c["inputs"]["s"], dev))
    assert torch.equal(out.cpu(), torch.tensor([[[4.0, 4.0], [4.0, 0.0]]]))
    ei, ew = postprocess_adj_pool_sparse(torch.tensor([[0, 1], [1, 0]], device=dev),
                                         torch.tensor([0.0, 1.0], device=dev), num_nodes=2)
    exact(ei, golden["postprocess_sparse_eps_literal"]["expected"]["edge_index"])
    close(ew, golden["postprocess_sparse_eps_literal"]["expected"]["edge_weight"])
    m = get_mask_from_dense_s(torch.tensor([[1.0, 0.0], [0.0, 1.0], [0.3, 0.7]], device=dev),
                              batch=torch.tensor([0, 2, 2], device=dev))
    exact(m, golden["mask_from_dense_s_literal"]["expected"]["mask"])
    c = golden["postprocess_dense_all"]
    close(postprocess_adj_pool_dense(D(c["inputs"]["adj_pool"], dev).clone(), True, True, True, True),
          c["expected"]["out"])


def test_dense_reduce_unbatched(golden, dev):
    from tgp.reduce import BaseReduce
    from tgp.select import SelectOutput
    c = golden["dense_reduce_unbatched"]
    S, X, b = (D(c["inputs"][k], dev) for k in ("S", "X", "batch"))
    e = c["expected"]
    so = SelectOutput(s=S, batch=b)
    a, ab = BaseReduce()(X, so, batch=b)
    close(a, e["flat"])
    exact(ab, e["flat_batch"])
    close(BaseReduce()(X, so, batch=b, return_batched=True)[0], e["batched"])
    close(BaseReduce()(X, SelectOutput(s=S))[0], e["single"])
    close(BaseReduce()(X, SelectOutput(s=S), return_batched=True)[0], e["single_batched"])
    exact(so.out_mask, e["out_mask"])


def test_postprocess_sparse_grid(golden, dev):
    from tgp.utils.ops import postprocess_adj_pool_sparse
    c = golden["postprocess_sparse_grid"]
    i, e = c["inputs"], c["expected"]
    for key in [k[:-3] for k in e if k.endswith("_ei")]:
        f = {p[:-1]: bool(int(p[-1])) for p in key.split("_")}
        ei, ew = postprocess_adj_pool_sparse(D(i["edge_index"], dev), D(i["edge_weight"], dev) if f["w"] else None,
                                             num_nodes=i["num_nodes"], remove_self_loops=f["rsl"],
                                             degree_norm=f["dn"], edge_weight_norm=f["ewn"],
                                             batch_pooled=D(i["batch_pooled"], dev))
        exact(ei, e[key + "_ei"], key)
        if e[key + "_ew"] is None:
            assert ew is None, key
        else:
            close(ew, e[key + "_ew"], key)


def test_block_diag(golden, dev):
    from tgp.utils.ops import dense_to_block_diag
    c = golden["block_diag"]
    ei, ew = dense_to_block_diag(D(c["inputs"]["adj_pool"], dev))
    exact(ei, c["expected"]["edge_index"])
    close(ew, c["expected"]["edge_weight"])
    ei, ew = dense_to_block_diag(torch.zeros(2, 3, 3, device=dev))
    assert ei.shape == (2, 0) and ew.shape == (0,)
    with pytest.raises(ValueError, match="adj_pool must have shape"):
        dense_to_block_diag(torch.tensor([1.0, 2.0, 3.0], device=dev))


def test_dense_connect_unbatched_grid(golden, dev):
    from tgp.connect import DenseConnect
    from tgp.reduce import BaseReduce
    from tgp.select import SelectOutput
    c = golden["dense_connect_unbatched_grid"]
    i, e = c["inputs"], c["expected"]
    b = D(i["batch"], dev)
    so = SelectOutput(s=D(i["S"], dev), batch=b)
    bp = BaseReduce.reduce_batch(so, b)
    for sp in (False, True):
        for ewn in (False, True):
            o, w = DenseConnect(sparse_output=sp, edge_weight_norm=ewn)(
                D(i["edge_index"], dev), so, edge_weight=D(i["edge_weight"], dev), batch=b, batch_pooled=bp)
            key = f"sp{int(sp)}_ewn{int(ewn)}"
            if sp:
                exact(o, e[key + "_adj"], key)
                close(w, e[key + "_w"], key)
            else:
                close(o, e[key + "_adj"], key)
                assert w is None


def test_empty_graph_edge_cases(dev):
    """Empty / degenerate inputs the reference tests (tests/connect/test_dense_conn.py:506-535 and
    tests/reduce/test_base_reduce.py:39-47)."""
    from tgp.connect import DenseConnect, SparseConnect
    from tgp.reduce import BaseReduce
    from tgp.select import SelectOutput
    s = torch.softmax(torch.randn(4, 2, device=dev), -1)
    ei = torch.empty((2, 0), dtype=torch.long, device=dev)
    conn = DenseConnect(sparse_output=False, remove_self_loops=False)
    o, w = conn(edge_index=ei, edge_weight=None, so=SelectOutput(s=s), batch=None)
    assert w is None and torch.equal(o.cpu(), torch.zeros(1, 2, 2))
    o, _ = conn(edge_index=ei, edge_weight=None, so=SelectOutput(s=s), batch=torch.tensor([0, 0, 1, 1], device=dev))
    assert torch.equal(o.cpu(), torch.zeros(2, 2, 2))
    so = SelectOutput(cluster_index=torch.tensor([0, 0, 1, 1], device=dev))
    o, w = SparseConnect()(ei, so, edge_weight=None)
    assert o.shape == (2, 0) and w is None
    out = BaseReduce.reduce_batch(SelectOutput(s=torch.empty((0, 2), device=dev)),
                                  torch.empty((0,), dtype=torch.long, device=dev))
    assert out.shape == (0,)
    x = torch.randn(4, 3, device=dev)
    with pytest.raises(ValueError, match="return_batched=True is only supported"):
        BaseReduce()(x, so, return_batched=True)


# ----------------------------------------------------------------------------------- protocol behaviour
def test_pooler_protocol_forward_lift_cache(golden, dev):
    """What reference tests/test_poolers.py:25-136 exercises for every alias: forward -> lift -> caching ->
    clear_cache -> reset_parameters, on the five poolers of the path."""
    from tgp.poolers import get_pooler
    from tgp.src import PoolingOutput
    i = golden["topk_batch_w_default"]["inputs"]
    x, ei, ew, batch = (D(i[k], dev) for k in ("x", "edge_index", "edge_weight", "batch"))
    n, f = x.shape
    for alias, kw in (("topk", dict(in_channels=f, ratio=0.5)), ("graclus", {}), ("ndp", {}),
                      ("diff", dict(in_channels=f, k=4)), ("mincut", dict(in_channels=f, k=4)),
                      ("diff_u", dict(in_channels=f, k=4)), ("mincut_u", dict(in_channels=f, k=4))):
        pool = get_pooler(alias, **kw).to(dev).eval()
        pool.reset_parameters()
        with torch.no_grad():
            out = pool(x=x, adj=ei, edge_weight=ew, batch=batch)
            assert isinstance(out, PoolingOutput) and out.so.num_nodes in (n, out.so.s.size(-2))
            lifted = pool(x=out.x, so=out.so, batch=batch, batch_pooled=out.batch, lifting=True)
        if pool.is_dense and pool.batched:
            assert lifted.shape[-1] == f and lifted.dim() == 3
        else:
            assert lifted.shape == (n, f)
        assert torch.isfinite(lifted).all()
    # caching (reference tests/poolers/test_graclus.py:30-47): same SelectOutput and pooled graph reused
    pool = get_pooler("graclus", cached=True)
    a = pool(x=x, adj=ei, edge_weight=ew, batch=batch)
    b = pool(x=x, adj=ei, edge_weight=ew, batch=batch)
    assert b.so is a.so and b.edge_index is a.edge_index and torch.equal(a.x, b.x)
    pool.clear_cache()
    c = pool(x=x, adj=ei, edge_weight=ew, batch=batch)
    assert c.so is not a.so and torch.equal(c.x, a.x)
    # dense preprocessing cache is only used for single-graph inputs
    dp = get_pooler("diff", in_channels=f, k=3, cache_preprocessing=True).to(dev).eval()
    with torch.no_grad():
        dp(x=x, adj=ei, edge_weight=ew, batch=batch)
        assert dp.preprocessing_cache is None
        one = batch == 0
        keep = one[ei[0]]
        dp(x=x[one], adj=ei[:, keep], edge_weight=ew[keep], batch=batch[one])
        assert dp.preprocessing_cache is not None
        dp.clear_cache()
        assert dp.preprocessing_cache is None
