"""Round-5 GPU parity tests (all through the C ABI): the fp64 dense path on v_mfma_f64_16x16x4_f64 -- S^T X, S^T A S,
the per-graph products of the unbatched mode, their gradients -- against the oracle evaluated in float64, and
torch.autograd.gradcheck in double on the dense poolers (reference: reduce/base_reduce.py:158-190,
connect/dense_conn.py:111-208, which run model.double() inputs through torch.matmul in fp64)."""
import os
import sys
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _oracle64(fn, *a, **k):
    """The oracle evaluated in float64 (its `torch.ones` defaults follow the default dtype)."""
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        return fn(*a, **k)
    finally:
        torch.set_default_dtype(old)


def _dense_problem(B, N, K, F, seed, density=0.05):
    g = torch.Generator().manual_seed(seed)
    a = (torch.rand(B, N, N, generator=g) < density).double() * torch.rand(B, N, N, generator=g, dtype=torch.float64)
    a = a + a.transpose(1, 2)
    s = torch.softmax(torch.randn(B, N, K, generator=g, dtype=torch.float64), -1)
    x = torch.randn(B, N, F, generator=g, dtype=torch.float64)
    return s, a, x


def _close64(got, want, what):
    """rtol 1e-12 on the scale of the tensor (sums of both signs cancel: the bar is on max |want|)."""
    assert got.dtype == torch.float64, what
    scale = float(want.abs().max()) or 1.0
    err = float((got.cpu() - want).abs().max())
    assert err <= 1e-12 * scale, f"{what}: max abs err {err:.3e} on scale {scale:.3e}"


@pytest.mark.parametrize("shape", [(3, 333, 37, 19), (32, 1024, 128, 64), (2, 70, 5, 3), (1, 16, 64, 130)])
@pytest.mark.parametrize("transposed", [False, True])
def test_float64_dense_pool_vs_fp64_oracle(dev, shape, transposed):
    """Fused A3 + A7 + A8 on float64 tensors (tgp_dense_pool_f64): x_pool, raw S^T A S and the post-processed adjacency
    against the oracle in float64 at 1e-12 -- fp32 arithmetic would miss by 1e-7.  Odd sizes exercise the guarded tile
    edges and unaligned rows; `transposed` is the view DenseSRCPooling.preprocessing hands over (src.py:442-443)."""
    import tgp_oracle as O
    from tgp import kernels as K
    B, N, Kc, F = shape
    s, a, x = _dense_problem(B, N, Kc, F, seed=B * 7 + N)
    if B * N * N > 8e6:  # the big case once
        if transposed:
            pytest.skip("large case runs in the contiguous layout only")
    a_in = a.to(dev)
    if transposed:
        a_in = a.transpose(1, 2).contiguous().to(dev).transpose(1, 2)  # same values, transposed memory
        assert not a_in.is_contiguous() or N == 1
    flags = K.dense_flags(True, True, True, False)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        x_pool, raw, post = K.dense_pool(s.to(dev), a_in, x.to(dev), flags, want_raw=True)
    want_raw = _oracle64(O.dense_connect, s, a)
    want_post = _oracle64(O.postprocess_dense, want_raw.clone(), True, True, True, False)
    _close64(x_pool, s.transpose(1, 2) @ x, "x_pool")
    _close64(raw, want_raw, "raw S^T A S")
    _close64(post, want_post, "post-processed adjacency")


def test_float64_bmm_and_accumulate(dev):
    from tgp import kernels as K
    g = torch.Generator().manual_seed(5)
    for (G, M, Kd, Nc) in [(1, 1, 1, 1), (3, 65, 17, 33), (2, 130, 257, 64), (5, 7, 300, 129)]:
        a = torch.randn(G, M, Kd, generator=g, dtype=torch.float64)
        b = torch.randn(G, Kd, Nc, generator=g, dtype=torch.float64)
        _close64(K.bmm(a.to(dev), b.to(dev)), a @ b, f"bmm {G, M, Kd, Nc}")
        at = a.transpose(1, 2).contiguous()
        _close64(K.bmm(at.to(dev), b.to(dev), trans_a=True), a @ b, "bmm trans_a")
        acc0 = torch.randn(G, M, Nc, generator=g, dtype=torch.float64)
        acc = acc0.clone().to(dev)
        out = K.bmm(a.to(dev), b.to(dev), accumulate_into=acc)
        assert out is acc
        _close64(acc, acc0 + a @ b, "bmm accumulate")
        # a float32 operand beside a float64 one is promoted (torch.matmul would raise; this is lenient)
        _close64(K.bmm(a.float().to(dev), b.to(dev)), a.float().double() @ b, "mixed dtypes")
    # 2-D operands and a broadcast batch-1 operand
    a = torch.randn(40, 30, generator=g, dtype=torch.float64)
    b = torch.randn(4, 30, 20, generator=g, dtype=torch.float64)
    _close64(K.bmm(a.to(dev), b.to(dev)), a @ b, "broadcast A")


def test_float64_unbatched_products_vs_fp64_oracle(dev):
    """The un-padded batch in float64: per-graph S_b^T Y_b (segment GEMM, node range split across workgroups), the
    row-side product and the CSR SpMM -- reference base_reduce.py:170-190, dense_conn.py:140-208 in double."""
    import tgp_oracle as O
    from tgp import kernels as K
    g = torch.Generator().manual_seed(9)
    sizes = torch.tensor([1, 700, 33, 64, 129, 5])
    ptr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])
    n, Kc, F = int(sizes.sum()), 21, 10
    s = torch.softmax(torch.randn(n, Kc, generator=g, dtype=torch.float64), -1)
    y = torch.randn(n, F, generator=g, dtype=torch.float64)
    got = K.segment_gemm_tn(s.to(dev), y.to(dev), ptr.to(dev), int(sizes.max()))
    want = torch.stack([s[ptr[b]:ptr[b + 1]].t() @ y[ptr[b]:ptr[b + 1]] for b in range(sizes.numel())])
    _close64(got, want, "segment_gemm_tn")
    m = torch.randn(sizes.numel(), Kc, F, generator=g, dtype=torch.float64)
    got = K.segment_gemm_nn(s.to(dev), m.to(dev), ptr.to(dev), int(sizes.max()))
    want = torch.cat([s[ptr[b]:ptr[b + 1]] @ m[b] for b in range(sizes.numel())])
    _close64(got, want, "segment_gemm_nn")
    # SpMM on a sorted, coalesced list
    e = 5000
    key = torch.unique(torch.randint(0, n * n, (e,), generator=g))
    ei = torch.stack([key // n, key % n])
    w = torch.randn(ei.size(1), generator=g, dtype=torch.float64)
    got = K.spmm_sorted(ei.to(dev), w.to(dev), n, s.to(dev))
    want = torch.zeros(n, Kc, dtype=torch.float64).index_add_(0, ei[0], w.view(-1, 1) * s[ei[1]])
    _close64(got, want, "spmm")
    # the whole unbatched dense Connect in double against the oracle
    batch = torch.repeat_interleave(torch.arange(sizes.numel()), sizes)
    same = batch[ei[0]] == batch[ei[1]]
    ei2, w2 = ei[:, same], w[same].abs()
    from tgp.connect import DenseConnect
    from tgp.select import SelectOutput
    conn = DenseConnect(remove_self_loops=True, degree_norm=True, adj_transpose=False)
    so = SelectOutput(s=s.to(dev), batch=batch.to(dev))
    adj_pool, _ = conn(ei2.to(dev), so, edge_weight=w2.to(dev), batch=batch.to(dev))
    raw = _oracle64(O.dense_connect_unbatched, ei2, w2, batch, s)
    want = _oracle64(O.postprocess_dense, raw.clone(), True, True, False, False)
    _close64(adj_pool, want, "DenseConnect unbatched, float64")


def _tiny_batch(dev, seed=0, graphs=3, f=5):
    g = torch.Generator().manual_seed(seed)
    xs, eis, bs, off = [], [], [], 0
    for gi in range(graphs):
        n = int(torch.randint(5, 9, (1,), generator=g))
        a = torch.triu(torch.rand(n, n, generator=g) < 0.5, 1)
        a = a | a.t()
        eis.append(a.nonzero().t() + off)
        xs.append(torch.randn(n, f, generator=g, dtype=torch.float64))
        bs.append(torch.full((n,), gi))
        off += n
    x, ei, batch = torch.cat(xs), torch.cat(eis, 1), torch.cat(bs)
    ew = torch.rand(ei.size(1), generator=g, dtype=torch.float64) + 0.5
    return x.to(dev), ei.to(dev), ew.to(dev), batch.to(dev)


@pytest.mark.parametrize("alias", ["diff", "mincut", "diff_u", "mincut_u"])
def test_gradcheck_in_double_on_the_dense_poolers(dev, alias):
    """torch.autograd.gradcheck in float64 -- the standard way to validate a pooling layer -- through the whole pooler
    (select, Reduce, Connect, post-processing, both auxiliary losses), w.r.t. the node features AND the selector's
    parameters.  Passes on the reference (ATen fp64); failed here until r5 because S^T X / S^T A S narrowed to fp32."""
    from tgp.poolers import get_pooler
    torch.manual_seed(0)
    x, ei, ew, batch = _tiny_batch(dev)
    pooler = get_pooler(alias, in_channels=x.size(1), k=3).to(dev).double()
    lin = pooler.selector.mlp.lins[0]

    def fn(xin, w, b):
        saved = (lin.weight.data, lin.bias.data)
        # functional view of the parameters so that gradcheck perturbs them too
        del lin._parameters["weight"], lin._parameters["bias"]
        lin.weight, lin.bias = w, b
        try:
            out = pooler(x=xin, adj=ei, edge_weight=ew, batch=batch)
        finally:
            del lin.weight, lin.bias
            lin._parameters["weight"] = torch.nn.Parameter(saved[0])
            lin._parameters["bias"] = torch.nn.Parameter(saved[1])
        adj_out = out.edge_index if out.edge_index.is_floating_point() else out.edge_weight
        return (out.x, adj_out) + tuple(out.loss.values())

    w0 = lin.weight.detach().clone().requires_grad_(True)
    b0 = lin.bias.detach().clone().requires_grad_(True)
    xin = x.clone().requires_grad_(True)
    with warnings.catch_warnings():
        warnings.simplefilter("error", UserWarning)
        outs = fn(xin, w0, b0)
        assert all(o.dtype == torch.float64 for o in outs)
        assert torch.autograd.gradcheck(fn, (xin, w0, b0), eps=1e-6, atol=1e-6, rtol=1e-5, nondet_tol=0.0)


def test_float64_dense_pooler_forward_vs_fp64_oracle(dev):
    """get_pooler("diff") / ("mincut") in double end to end against the oracle's pooler functions in double: pooled
    features, adjacency and losses agree to 1e-11 (the softmax goes through ATen in both)."""
    import tgp_oracle as O
    from tgp.poolers import get_pooler
    torch.manual_seed(1)
    x, ei, ew, batch = _tiny_batch(dev, seed=3, graphs=9, f=6)
    for alias in ("diff", "mincut"):
        pooler = get_pooler(alias, in_channels=6, k=4).to(dev).double().eval()
        with torch.no_grad():
            out = pooler(x=x, adj=ei, edge_weight=ew, batch=batch)
        lin = pooler.selector.mlp.lins[0]
        ref = _oracle64(O.dense_pool, alias, x.cpu(), ei.cpu(), ew.cpu(), batch.cpu(), [lin.weight.detach().cpu()],
                        [lin.bias.detach().cpu()])
        torch.testing.assert_close(out.x.cpu(), ref["x"], rtol=1e-11, atol=1e-12)
        torch.testing.assert_close(out.edge_index.cpu(), ref["edge_index"], rtol=1e-11, atol=1e-12)
        for name, val in ref["loss"].items():
            torch.testing.assert_close(out.loss[name].cpu(), val, rtol=1e-10, atol=1e-12)


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


@pytest.mark.parametrize("alias,kw", [("topk", dict(ratio=0.5)), ("graclus", {}), ("ndp", {})])
@pytest.mark.parametrize("shape", ["small_graphs", "one_large_graph"])
def test_sparse_poolers_hand_out_contiguous_exact_size_edge_lists(dev, alias, kw, shape):
    """SURVEY 8(b) "Ownership" / connect/base_conn.py:103-112: every output is a NEW tensor of the pooled size.  r4's
    one-launch operators returned ``edge_index`` as a [2, E'] view of a capacity-E buffer (torch.equal ignores strides, so
    parity could not see it): ``.view(-1)`` failed where the reference's tensor works and E * 20 bytes stayed pinned.
    Default now: contiguous, storage <= 2 x the logical size, on every sparse pooler and both size regimes; the view
    layout is opt-in (``tgp.output_views``) and gives the same values."""
    import tgp
    from tgp.poolers import get_pooler
    if shape == "small_graphs":
        x, ei, ew, batch = _er_batch(200, 5, 60, 8, 7, dev)
    else:
        x, ei, ew, batch = _er_batch(1, 3000, 3000, 8, 8, dev)
    torch.manual_seed(3)
    pooler = get_pooler(alias, in_channels=8, **kw).to(dev).eval()
    with torch.no_grad():
        out = pooler(x=x, adj=ei, edge_weight=ew, batch=batch)
        so = out.so
        with tgp.output_views():
            out_v = pooler(x=x, adj=ei, edge_weight=ew, batch=batch, so=so)
        out2 = pooler(x=x, adj=ei, edge_weight=ew, batch=batch, so=so)
    for o in (out, out2):
        e = o.edge_index
        assert e.is_contiguous() and e.stride() == (e.size(1), 1)
        flat = e.view(-1)                                  # the reference's tensor allows this
        assert flat.numel() == 2 * e.size(1)
        assert e.untyped_storage().nbytes() <= max(2 * e.numel() * 8, 512)
        if o.edge_weight is not None:
            assert o.edge_weight.is_contiguous()
            assert o.edge_weight.untyped_storage().nbytes() <= max(2 * o.edge_weight.numel() * 4, 512)
        assert o.x.untyped_storage().nbytes() <= max(2 * o.x.numel() * 4, 512)
    assert torch.equal(out2.edge_index, out_v.edge_index) and torch.equal(out2.x, out_v.x)
    if out2.edge_weight is not None:
        assert torch.equal(out2.edge_weight, out_v.edge_weight)
    assert torch.equal(out2.batch, out_v.batch)


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


# ------------------------------------------------------------------------------------ fresh batches (r5)
def test_one_launch_batch_facts_equal_the_general_route(dev, monkeypatch):
    """tgp_batch_facts_sorted_i64 (one launch, pinned-word hand-over) against the two-kernel route and plain torch:
    CSR offsets, sizes, graph count, longest graph, non-empty graphs, TopkSelect's plan -- sorted vectors with empty graph
    ids, one graph, one node per graph; an unsorted vector / ids out of range / a long run of empty ids fall back."""
    import tgp.utils.ops as ops
    import tgp_oracle as O
    g = torch.Generator().manual_seed(2)
    cases = {
        "proteins": torch.repeat_interleave(torch.arange(2048), torch.randint(20, 61, (2048,), generator=g)),
        "with_empty_ids": torch.repeat_interleave(torch.tensor([0, 1, 4, 5, 9]), torch.tensor([3, 1, 70, 2, 300])),
        "one_graph": torch.zeros(5000, dtype=torch.long),
        "starts_late": torch.full((77,), 3),
        "node_per_graph": torch.arange(3000),
        "single_node": torch.zeros(1, dtype=torch.long),
    }
    for name, b in cases.items():
        bd = b.to(dev)
        ops._BATCH_INFO.clear()
        monkeypatch.setattr(ops, "_BATCH_FACTS_ONE_LAUNCH", True)
        a = ops.batch_info(bd, topk_ratio=0.5)
        assert a.is_sorted and a.memo.get(("topk", 0.5)) is not None, name
        ops._BATCH_INFO.clear()
        monkeypatch.setattr(ops, "_BATCH_FACTS_ONE_LAUNCH", False)
        r = ops.batch_info(bd, topk_ratio=0.5)
        sizes = torch.bincount(b)
        assert a.num_graphs == r.num_graphs == sizes.numel(), name
        assert torch.equal(a.sizes.cpu(), sizes) and torch.equal(r.sizes.cpu(), sizes), name
        ptr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])
        assert torch.equal(a.ptr.cpu(), ptr) and torch.equal(r.ptr.cpu(), ptr), name
        assert a.max_nodes == r.max_nodes == int(sizes.max()) and a.distinct == r.distinct == int((sizes > 0).sum()), name
        total, k, koff = a.memo[("topk", 0.5)]
        want_k = torch.ceil(torch.tensor(0.5, dtype=torch.float32) * sizes.float()).long()
        assert torch.equal(k.cpu(), want_k) and total == int(want_k.sum()), name
        assert torch.equal(koff.cpu(), torch.cat([torch.zeros(1, dtype=torch.long), want_k.cumsum(0)])), name
    monkeypatch.setattr(ops, "_BATCH_FACTS_ONE_LAUNCH", True)
    for name, b in {"unsorted": torch.tensor([0, 0, 2, 1, 2]), "long_gap": torch.tensor([0] * 10 + [500] * 10)}.items():
        ops._BATCH_INFO.clear()
        info = ops.batch_info(b.to(dev))
        sizes = torch.bincount(b)
        assert torch.equal(info.sizes.cpu(), sizes), name      # the general route took it
        assert info.is_sorted == (name != "unsorted")
    # back-to-back calls on one stream: the ticket / flag words are left clean by every call, also by a refused one
    ops._BATCH_INFO.clear()
    for _ in range(3):
        for b in (cases["proteins"], torch.tensor([3, 2, 1]), cases["with_empty_ids"]):
            ops._BATCH_INFO.clear()
            info = ops.batch_info(b.to(dev))
            assert torch.equal(info.sizes.cpu(), torch.bincount(b))


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


def test_to_dense_adj_with_multi_channel_edge_attributes(dev):
    """PyG to_dense_adj with edge_attr [E, C] (reference src.py:434): native scatter-add on the device (r5; a torch
    index_add_ form before), against the torch form on the host, both orientations, duplicates summed, max_num_nodes."""
    from tgp.src import to_dense_adj
    g = torch.Generator().manual_seed(8)
    sizes = torch.tensor([5, 9, 1, 7])
    batch = torch.repeat_interleave(torch.arange(4), sizes)
    ptr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])
    rows, cols = [], []
    for b in range(4):
        m = int(sizes[b])
        r = torch.randint(0, m, (3 * m,), generator=g) + ptr[b]
        c = torch.randint(0, m, (3 * m,), generator=g) + ptr[b]
        rows.append(r); cols.append(c)
    ei = torch.stack([torch.cat(rows), torch.cat(cols)])       # duplicates included
    attr = torch.randn(ei.size(1), 3, generator=g)
    for transposed in (False, True):
        for nmax in (None, 6):
            want = to_dense_adj(ei, batch, attr, max_num_nodes=nmax, transposed=transposed)          # host: torch form
            got = to_dense_adj(ei.to(dev), batch.to(dev), attr.to(dev), max_num_nodes=nmax, transposed=transposed)
            assert got.shape == want.shape
            torch.testing.assert_close(got.cpu(), want.contiguous(), rtol=1e-6, atol=1e-6)


# ------------------------------------------------------------------ r5: the dense poolers' training step in fewer launches
@pytest.mark.gpu
@pytest.mark.parametrize("M,K,F", [(1000, 20, 32), (64 * 3 + 5, 7, 16), (5000, 32, 64), (300, 13, 50), (122880, 20, 32),
                                   (63, 1, 1), (4097, 32, 33), (64, 4, 3), (129, 31, 17)])
@pytest.mark.parametrize("accumulate", [False, True])
def test_selector_backward_in_one_launch_vs_autograd(dev, M, K, F, accumulate):
    """tgp_mlp_select_bwd_f32 against torch autograd of softmax(x W^T + b) * mask in float64 (select/mlp_select.py:139-145):
    gx (also added in place to an existing gradient), gw, gb; twice the same bits (fixed-order partial sums)."""
    from tgp import kernels as K_
    g = torch.Generator().manual_seed(M + 7 * K + F)
    x = torch.randn(M, F, generator=g).to(dev)
    w = (torch.randn(K, F, generator=g) * 0.3).to(dev)
    b = torch.randn(K, generator=g).to(dev)
    mask = (torch.rand(M, generator=g) < 0.9).to(dev)
    gs = torch.randn(M, K, generator=g).to(dev)
    x64, w64, b64 = (t.double().requires_grad_(True) for t in (x, w, b))
    s64 = torch.softmax(x64 @ w64.t() + b64, -1) * mask.unsqueeze(-1)
    (s64 * gs.double()).sum().backward()
    s = K_.mlp_select(x, w, b, mask)
    torch.testing.assert_close(s, s64.detach().float(), rtol=1e-5, atol=1e-6)
    base = torch.randn(M, F, generator=g).to(dev) if accumulate else None
    gx, gw, gb = K_.mlp_select_bwd(s, gs, x, w, gx_accumulate=None if base is None else base.clone())
    want_gx = x64.grad.float() + (base if base is not None else 0)
    torch.testing.assert_close(gx, want_gx, rtol=1e-4, atol=1e-5 * max(1.0, float(want_gx.abs().max())))
    scale = max(1.0, float(w64.grad.abs().max()))
    torch.testing.assert_close(gw, w64.grad.float(), rtol=2e-4, atol=2e-5 * scale * max(1.0, (M / 1000) ** 0.5))
    torch.testing.assert_close(gb, b64.grad.float(), rtol=2e-4, atol=2e-5 * max(1.0, float(b64.grad.abs().max())))
    gx2, gw2, gb2 = K_.mlp_select_bwd(s, gs, x, w, gx_accumulate=None if base is None else base.clone())
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2) and torch.equal(gx, gx2)
    # only some gradients asked for
    gx3, gw3, gb3 = K_.mlp_select_bwd(s, gs, x, w, want_gx=False, want_gb=False)
    assert gx3 is None and gb3 is None and torch.equal(gw3, gw)
    gx4, gw4, gb4 = K_.mlp_select_bwd(s, gs, x, w, want_gw=False, want_gb=False)
    assert gw4 is None and gb4 is None
    if not accumulate:
        assert torch.equal(gx4, gx)


@pytest.mark.gpu
def test_selector_backward_is_run_to_run_identical_under_load(dev):
    """gW / gb are partial sums per workgroup added in a fixed order by a second launch: 200 calls interleaved with
    streaming copies on another stream and calls of other sizes must all give the first call's bits."""
    from tgp import kernels as K_
    g = torch.Generator().manual_seed(1)
    M, K, F = 122880, 20, 32
    x = torch.randn(M, F, generator=g).to(dev)
    w = (torch.randn(K, F, generator=g) * 0.3).to(dev)
    gs = torch.randn(M, K, generator=g).to(dev)
    s = torch.softmax(torch.randn(M, K, generator=g), -1).to(dev)
    big = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    x2, gs2, s2 = x[:5000], gs[:5000], s[:5000].contiguous()
    ref = K_.mlp_select_bwd(s, gs, x, w)
    ref2 = K_.mlp_select_bwd(s2, gs2, x2, w)
    side = torch.cuda.Stream()
    for it in range(200):
        if it % 3 == 0:
            with torch.cuda.stream(side):
                big.copy_(big.flip(0) if it % 6 == 0 else big)  # memory traffic from another stream
        got = K_.mlp_select_bwd(s, gs, x, w)
        got2 = K_.mlp_select_bwd(s2, gs2, x2, w)
        assert torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2]), it
        assert torch.equal(got2[1], ref2[1]) and torch.equal(got2[2], ref2[2]), it
    torch.cuda.synchronize()
    assert torch.equal(got[0], ref[0])


@pytest.mark.gpu
def test_selector_backward_outside_its_shapes_is_refused(dev):
    from tgp import kernels as K_, _native as N
    assert K_.mlp_select_bwd_fits(32, 64) and not K_.mlp_select_bwd_fits(33, 8) and not K_.mlp_select_bwd_fits(8, 65)
    s = torch.rand(10, 40, device=dev)
    with pytest.raises(N.TgpNativeError):
        K_.mlp_select_bwd(s, s.clone(), torch.rand(10, 8, device=dev), torch.rand(40, 8, device=dev))


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["mincut", "diff"])
def test_fused_function_hands_the_two_losses_out_as_scalars(dev, which):
    """functions.dense_pool_small(loss_scalars=True): the two auxiliary losses are 0-dim outputs (LossPair); their values
    and the gradients they send equal the [2,B]-terms / [2]-diff form's; a sum loss (expanded scalar gradients, read by
    the kernel as ONE value) equals the same loss with materialised gradients; a loss that uses only one of the two
    leaves the other's upstream gradient missing (NULL in the C call)."""
    from tgp import functions as Fn, kernels as K_
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_round3 import _ragged_dense_batch
    B, Nmax, K, F = 90, 40, 20, 32
    A, X, logits, mask = _ragged_dense_batch(B, Nmax, K, F, seed=11, dev=dev)
    flags = K_.dense_flags(True, True, True, False)
    scales = (0.37, 1.0 / int(mask.sum())) if which == "diff" else None

    def run(scalars, loss_of):
        l = logits.clone().requires_grad_(True)
        x = X.clone().requires_grad_(True)
        S = torch.softmax(l, -1) * mask.unsqueeze(-1)
        out = Fn.dense_pool_small(S, A, x, flags, which == "mincut", which == "mincut", scales, loss_scalars=scalars)
        loss_of(out).backward()
        return out, l.grad, x.grad

    def pair_of(out):
        return out[4] if which == "diff" else out[3]

    def old_loss(out):
        p = pair_of(out)
        both = p if which == "diff" else p.mean(dim=1)
        return out[0].sum() + out[2].sum() * 0.5 + both[0] * 1.7 - both[1] * 0.3

    def new_loss(out):
        p = pair_of(out)
        assert isinstance(p, Fn.LossPair) and p[0].dim() == 0 and p[1].dim() == 0
        return out[0].sum() + out[2].sum() * 0.5 + p[0] * 1.7 - p[1] * 0.3

    o_old, gl_old, gx_old = run(False, old_loss)
    o_new, gl_new, gx_new = run(True, new_loss)
    p_old, p_new = pair_of(o_old), pair_of(o_new)
    both_old = p_old if which == "diff" else p_old.mean(dim=1)
    torch.testing.assert_close(torch.stack(list(p_new)), both_old.detach(), rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(gl_new, gl_old, rtol=1e-4, atol=1e-6 * max(1.0, float(gl_old.abs().max())))
    torch.testing.assert_close(gx_new, gx_old, rtol=1e-5, atol=1e-6)
    # materialised upstream gradients of the same sum loss
    _, gl_m, gx_m = run(True, lambda out: (out[0] * torch.ones_like(out[0])).sum()
                        + (out[2] * torch.full_like(out[2], 0.5)).sum() + pair_of(out)[0] * 1.7 - pair_of(out)[1] * 0.3)
    torch.testing.assert_close(gl_new, gl_m, rtol=1e-5, atol=1e-7 * max(1.0, float(gl_m.abs().max())))
    torch.testing.assert_close(gx_new, gx_m, rtol=1e-6, atol=1e-7)
    # one loss only
    _, gl_one, _ = run(True, lambda out: pair_of(out)[1] * 2.0)
    _, gl_ref, _ = run(False, lambda out: (pair_of(out) if which == "diff" else pair_of(out).mean(dim=1))[1] * 2.0)
    torch.testing.assert_close(gl_one, gl_ref, rtol=1e-4, atol=1e-6 * max(1.0, float(gl_ref.abs().max())))


@pytest.mark.gpu
@pytest.mark.parametrize("alias", ["mincut", "diff"])
def test_dense_pooler_training_step_as_one_autograd_node(dev, alias, monkeypatch):
    """get_pooler('mincut' / 'diff') with a single-Linear selector on a PROTEINS-shaped sparse batch, training: Select +
    Reduce + Connect + losses run as functions._SelectPoolSmallFn (forward tgp_dense_pool_select_f32, backward
    tgp_dense_pool_small_bwd_f32 + tgp_mlp_select_bwd_f32); outputs, losses and every gradient equal the two-node form's
    (TGP_FOLD_TRAINING=0), and S stays differentiable for a caller's own use."""
    from tgp import kernels as K_
    from tgp import poolers as P
    from tgp.poolers import get_pooler
    g = torch.Generator().manual_seed(5)
    sizes = torch.randint(20, 61, (96,), generator=g)
    n = int(sizes.sum())
    batch = torch.repeat_interleave(torch.arange(96), sizes).to(dev)
    start = (torch.cumsum(sizes, 0) - sizes).to(dev)
    src = torch.arange(n, device=dev).repeat_interleave(2)
    dst = start[batch[src]] + (torch.rand(src.numel(), device=dev) * sizes.to(dev)[batch[src]]).long()
    keep = src != dst
    key = torch.unique(torch.cat([src[keep] * n + dst[keep], dst[keep] * n + src[keep]]))
    ei = torch.stack([key // n, key % n])
    x0 = torch.randn(n, 32, device=dev)
    torch.manual_seed(0)
    pooler = get_pooler(alias, in_channels=32, k=20).to(dev).train()
    calls = []
    real = K_.mlp_select_bwd
    monkeypatch.setattr(K_, "mlp_select_bwd", lambda *a, **k: (calls.append(k.get("gx_accumulate") is not None), real(*a, **k))[1])

    def step(extra_s):
        pooler.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        out = pooler(x=x, adj=ei, batch=batch)
        loss = out.x.square().sum() + out.edge_index.sum() + sum(out.loss.values())
        if extra_s:
            loss = loss + (out.so.s * out.so.s).sum() * 0.1
        loss.backward()
        return (out.x.detach(), out.edge_index.detach(), {k: v.detach() for k, v in out.loss.items()}, x.grad,
                [p.grad.clone() for p in pooler.parameters()])

    # node features that need no gradient (input data): only the parameters get one, and it is the same
    monkeypatch.setattr(P, "_FOLD_TRAINING", True)
    pooler.zero_grad(set_to_none=True)
    out = pooler(x=x0, adj=ei, batch=batch)
    (out.x.square().sum() + sum(out.loss.values())).backward()
    g_data = [p.grad.clone() for p in pooler.parameters()]
    pooler.zero_grad(set_to_none=True)
    xr = x0.clone().requires_grad_(True)
    out = pooler(x=xr, adj=ei, batch=batch)
    (out.x.square().sum() + sum(out.loss.values())).backward()
    for a, b in zip(g_data, [p.grad for p in pooler.parameters()]):
        assert torch.equal(a, b)
    for extra_s in (False, True):
        calls.clear()
        monkeypatch.setattr(P, "_FOLD_TRAINING", True)
        new = step(extra_s)
        assert calls == [True], calls  # one selector backward, accumulating into the pooling backward's gX
        monkeypatch.setattr(P, "_FOLD_TRAINING", False)
        old = step(extra_s)
        torch.testing.assert_close(new[0], old[0], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(new[1], old[1], rtol=1e-5, atol=1e-6)
        for k in old[2]:
            torch.testing.assert_close(new[2][k], old[2][k], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(new[3], old[3], rtol=2e-4, atol=1e-5 * max(1.0, float(old[3].abs().max())))
        for a, b in zip(new[4], old[4]):
            torch.testing.assert_close(a, b, rtol=5e-4, atol=2e-5 * max(1.0, float(b.abs().max())))


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


# ------------------------------------------------------------------ r5: the post-processing spread over K / 16 workgroups
@pytest.mark.gpu
@pytest.mark.parametrize("K", [68, 100, 128, 132, 200, 256])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_fused_dense_call_with_the_spread_post_processing(dev, K, dtype, monkeypatch):
    """tgp_dense_pool_f32 / _f64 on batches that take the tiled GEMM path with 64 < K <= 256: the second product leaves
    partial column sums, K / 16 workgroups per graph post-process (post_rows_kernel / dense64_post_rows_kernel).  Ragged
    row tiles (K not a multiple of 64 or 16), every combination of remove_self_loops / degree_norm / raw output, against
    the oracle (utils/ops.py:282-335); adj_transpose=False and edge_weight_norm keep the one-workgroup kernels and must
    agree as well."""
    import tgp_oracle as O
    from tgp import kernels as K_
    B, N, F = 3, 700, 24
    g = torch.Generator().manual_seed(K)
    A = ((torch.rand(B, N, N, generator=g) < 0.02).double() * torch.rand(B, N, N, generator=g, dtype=torch.float64))
    X = torch.randn(B, N, F, generator=g, dtype=torch.float64)
    S = torch.softmax(torch.randn(B, N, K, generator=g, dtype=torch.float64) * 2, -1)
    Ad, Xd, Sd = (t.to(dev, dtype) for t in (A, X, S))
    tol = dict(rtol=1e-5, atol=1e-6) if dtype == torch.float32 else dict(rtol=1e-11, atol=1e-12)
    raw_ref = _oracle64(O.dense_connect, S, A) if dtype == torch.float64 else O.dense_connect(S.float(), A.float()).double()
    xref = S.transpose(1, 2) @ X
    for rsl in (True, False):
        for dn in (True, False):
            for at, ewn in ((True, False), (False, False), (True, True)):
                for want_raw in (False, True):
                    flags = K_.dense_flags(rsl, dn, at, ewn)
                    xp, raw, ap = K_.dense_pool(Sd, Ad, Xd, flags, want_raw=want_raw, want_post=True)[:3]
                    ref = _oracle64(O.postprocess_dense, raw_ref, rsl, dn, at, ewn)
                    torch.testing.assert_close(ap.double().cpu(), ref, **tol)
                    torch.testing.assert_close(xp.double().cpu(), xref, rtol=tol["rtol"] * 10, atol=tol["atol"] * 100)
                    if want_raw:
                        torch.testing.assert_close(raw.double().cpu(), raw_ref, **tol)


# ------------------------------------------------------------------ r5: the dense poolers' forward straight from sparse inputs
@pytest.mark.gpu
def test_diffpool_inference_from_the_unpadded_batch_equals_the_densified_one(dev, monkeypatch):
    """get_pooler('diff') in inference on sparse inputs takes the same launch; its two losses (the link loss needs the
    dense adjacency, utils/losses.py:644-658) are computed from the adjacency the launch leaves as a side output."""
    from tgp import kernels as K_
    from tgp import poolers as P
    from tgp.poolers import get_pooler
    g = torch.Generator().manual_seed(41)
    B = 72
    sizes = torch.randint(6, 61, (B,), generator=g)
    n = int(sizes.sum())
    batch = torch.repeat_interleave(torch.arange(B), sizes)
    start = torch.cumsum(sizes, 0) - sizes
    row = torch.repeat_interleave(torch.arange(n), torch.randint(1, 6, (n,), generator=g))
    col = start[batch[row]] + (torch.rand(row.numel(), generator=g) * sizes[batch[row]]).long()
    ei, bd, x = torch.stack([row, col]).to(dev), batch.to(dev), torch.randn(n, 32, generator=g).to(dev)
    for at in (True, False):
        torch.manual_seed(0)
        pooler = get_pooler("diff", in_channels=32, k=20, adj_transpose=at).to(dev).eval()
        with torch.no_grad():
            monkeypatch.setattr(P, "_FOLD_SPARSE_INPUTS", True)
            new = pooler(x=x, adj=ei, batch=bd)
            monkeypatch.setattr(P, "_FOLD_SPARSE_INPUTS", False)
            old = pooler(x=x, adj=ei, batch=bd)
        torch.testing.assert_close(new.x, old.x, rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(new.edge_index, old.edge_index, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(new.so.s, old.so.s, rtol=1e-6, atol=1e-7)
        assert set(new.loss) == set(old.loss) == {"link_loss", "entropy_loss"}
        for k in old.loss:
            torch.testing.assert_close(new.loss[k], old.loss[k], rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("adj_transpose", [True, False])
@pytest.mark.parametrize("weighted", [False, True])
def test_mincut_forward_from_the_unpadded_batch_equals_the_densified_one(dev, adj_transpose, weighted, monkeypatch):
    """get_pooler('mincut') in inference on a sorted batch of small graphs given as PyG hands it over: ONE launch builds
    every graph's adjacency tile in LDS from its edges (tgp_dense_pool_select_sparse_f32) -- no to_dense_batch, no
    to_dense_adj.  Outputs, losses, S, mask and the pooled batch vector equal the densified path's (src.py:434-450 in
    front of the same fused call): duplicates summed, self loops, a graph without edges, a graph of more than 512
    entries, a column that leaves its row's graph.  An unsorted list keeps the densified path."""
    from tgp import kernels as K_
    from tgp import poolers as P
    from tgp.poolers import get_pooler
    g = torch.Generator().manual_seed(23)
    B = 80
    sizes = torch.randint(5, 61, (B,), generator=g)
    sizes[3] = 60
    n = int(sizes.sum())
    batch = torch.repeat_interleave(torch.arange(B), sizes)
    start = torch.cumsum(sizes, 0) - sizes
    deg = torch.randint(0, 7, (n,), generator=g)
    deg[batch == 5] = 0                                  # a graph without edges
    deg[batch == 3] = 12                                 # 60 nodes x 12 = 720 entries: the tail loop behind 512
    row = torch.repeat_interleave(torch.arange(n), deg)
    col = start[batch[row]] + (torch.rand(row.numel(), generator=g) * sizes[batch[row]]).long()
    col[::7] = row[::7]                                  # self loops
    col[1::13] = col[0::13][: col[1::13].numel()]        # (some duplicates)
    cross = (batch[row] == 10).nonzero().flatten()[:2]
    col[cross] = start[11] + 1                           # a column in the next graph
    ei = torch.stack([row, col]).to(dev)
    ew = (torch.rand(row.numel(), generator=g) + 0.1).to(dev) if weighted else None
    x = torch.randn(n, 32, generator=g).to(dev)
    bd = batch.to(dev)
    torch.manual_seed(0)
    calls = []
    real = K_.dense_pool_select_sparse
    monkeypatch.setattr(K_, "dense_pool_select_sparse", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    for sparse_output in (False, True):
        pooler = get_pooler("mincut", in_channels=32, k=20, adj_transpose=adj_transpose,
                            sparse_output=sparse_output).to(dev).eval()
        calls.clear()
        with torch.no_grad():
            monkeypatch.setattr(P, "_FOLD_SPARSE_INPUTS", True)
            new = pooler(x=x, adj=ei, edge_weight=ew, batch=bd)
            assert calls == [1]
            monkeypatch.setattr(P, "_FOLD_SPARSE_INPUTS", False)
            old = pooler(x=x, adj=ei, edge_weight=ew, batch=bd)
            assert calls == [1]
        torch.testing.assert_close(new.x, old.x, rtol=1e-6, atol=1e-6)
        if sparse_output:
            assert torch.equal(new.edge_index, old.edge_index) and torch.equal(new.batch, old.batch)
            torch.testing.assert_close(new.edge_weight, old.edge_weight, rtol=1e-5, atol=1e-6)
        else:
            torch.testing.assert_close(new.edge_index, old.edge_index, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(new.so.s, old.so.s, rtol=1e-6, atol=1e-7)
        assert torch.equal(new.so.in_mask, old.so.in_mask)
        for k in old.loss:
            torch.testing.assert_close(new.loss[k], old.loss[k], rtol=1e-5, atol=1e-6)
    # rows not sorted: the densified path takes the call, same result
    perm = torch.randperm(ei.size(1), generator=g).to(dev)
    ei2, ew2 = ei[:, perm].contiguous(), (None if ew is None else ew[perm].contiguous())
    calls.clear()
    with torch.no_grad():
        monkeypatch.setattr(P, "_FOLD_SPARSE_INPUTS", True)
        shuffled = pooler(x=x, adj=ei2, edge_weight=ew2, batch=bd)
        # (a NEW list is tried optimistically -- the kernel runs on clamped ranges while the facts kernel's verdict
        #  travels -- and its outputs are dropped; the verdict is remembered: the second call does not try)
        assert calls == [1] and K_._rows_sorted_memo(ei2) is False
        again = pooler(x=x, adj=ei2, edge_weight=ew2, batch=bd)
        assert calls == [1]
    torch.testing.assert_close(shuffled.x, old.x, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(again.x, old.x, rtol=1e-5, atol=1e-5)
    # a NEW sorted list: ranges and verdict from the one facts launch, no lower-bounds launch, remembered afterwards
    ei3 = ei.clone()
    calls.clear()
    with torch.no_grad():
        fresh = pooler(x=x, adj=ei3, edge_weight=ew, batch=bd.clone())
    assert calls == [1] and K_._rows_sorted_memo(ei3) is True
    torch.testing.assert_close(fresh.x, old.x, rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("alias", ["mincut", "diff"])
@pytest.mark.parametrize("adj_transpose", [True, False])
def test_dense_pooler_training_step_from_the_unpadded_batch(dev, alias, adj_transpose, monkeypatch):
    """get_pooler('mincut' / 'diff') in training on a sorted batch of small graphs given as sparse tensors: the forward is the
    launch that reads the un-padded batch (the padded x and the dense adjacency the backward kernels need are its side
    outputs: no to_dense_batch / to_dense_adj launches), the backward ends with the gather back to the un-padded rows.
    Outputs, losses and every gradient equal the densified path's."""
    from tgp import kernels as K_
    from tgp import poolers as P
    from tgp.poolers import get_pooler
    g = torch.Generator().manual_seed(31)
    B = 96
    sizes = torch.randint(8, 61, (B,), generator=g)
    n = int(sizes.sum())
    batch = torch.repeat_interleave(torch.arange(B), sizes)
    start = torch.cumsum(sizes, 0) - sizes
    deg = torch.randint(1, 6, (n,), generator=g)
    row = torch.repeat_interleave(torch.arange(n), deg)
    col = start[batch[row]] + (torch.rand(row.numel(), generator=g) * sizes[batch[row]]).long()
    ei, bd = torch.stack([row, col]).to(dev), batch.to(dev)
    ew = (torch.rand(row.numel(), generator=g) + 0.1).to(dev)
    x0 = torch.randn(n, 32, generator=g).to(dev)
    torch.manual_seed(0)
    pooler = get_pooler(alias, in_channels=32, k=20, adj_transpose=adj_transpose).to(dev).train()
    calls = []
    real = K_.dense_pool_select_sparse
    monkeypatch.setattr(K_, "dense_pool_select_sparse", lambda *a, **k: (calls.append(k.get("want_dense")), real(*a, **k))[1])

    def step(x_needs_grad):
        pooler.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(x_needs_grad)
        out = pooler(x=x, adj=ei, edge_weight=ew, batch=bd)
        loss = out.x.square().sum() + (out.edge_index * 0.5).sum() + sum(out.loss.values()) + (out.so.s ** 2).sum() * 0.1
        loss.backward()
        return (out.x.detach(), out.edge_index.detach(), {k: v.detach() for k, v in out.loss.items()}, x.grad,
                [p.grad.clone() for p in pooler.parameters()])

    for x_needs_grad in (True, False):
        calls.clear()
        monkeypatch.setattr(P, "_FOLD_SPARSE_INPUTS", True)
        new = step(x_needs_grad)
        assert calls == [True]
        monkeypatch.setattr(P, "_FOLD_SPARSE_INPUTS", False)
        old = step(x_needs_grad)
        assert calls == [True]
        torch.testing.assert_close(new[0], old[0], rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(new[1], old[1], rtol=1e-5, atol=1e-6)
        for k in old[2]:
            torch.testing.assert_close(new[2][k], old[2][k], rtol=1e-5, atol=1e-6)
        if x_needs_grad:
            torch.testing.assert_close(new[3], old[3], rtol=1e-4, atol=1e-5 * max(1.0, float(old[3].abs().max())))
        else:
            assert new[3] is None and old[3] is None
        for a, b in zip(new[4], old[4]):
            torch.testing.assert_close(a, b, rtol=2e-4, atol=1e-5 * max(1.0, float(b.abs().max())))


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


# ------------------------------------------------------------------- medium graphs, eight waves per graph (r5, late)
@pytest.mark.parametrize("B,N,K,F", [(9, 300, 64, 128), (10, 333, 40, 19), (9, 700, 32, 40), (12, 512, 64, 64)])
@pytest.mark.parametrize("waves", ["auto", "4", "8"])
def test_medium_graph_kernel_four_and_eight_waves_vs_oracle(dev, B, N, K, F, waves):
    """dense_pool_medium_kernel<MT, MINW, WAVES>: shapes whose S tile leaves one workgroup per CU take eight waves per
    graph (TGP_MEDIUM_WAVES forces either form, read once per process: a child process per setting); every form against
    the oracle (base_reduce.py:158-161, dense_conn.py:111-122, utils/ops.py:282-335) with ragged graph sizes."""
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import sys, torch
sys.path.insert(0, {os.path.join(ROOT, 'torch-geometric-pool_amd')!r}); sys.path.insert(0, {os.path.join(ROOT, 'oracle')!r})
import tgp_oracle as O
from tgp import kernels as K
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed({B * N + K})
B, N, Kc, F = {B}, {N}, {K}, {F}
sizes = torch.randint(N // 2, N + 1, (B,), generator=g); sizes[0] = N
mask = torch.arange(N)[None, :] < sizes[:, None]
A = (torch.rand(B, N, N, generator=g) < 0.02).float() * mask[:, :, None] * mask[:, None, :]
X = torch.randn(B, N, F, generator=g) * mask[..., None]
S = torch.softmax(torch.randn(B, N, Kc, generator=g), -1) * mask[..., None]
flags = K.dense_flags(True, True, False, False)
xp, raw, pooled = K.dense_pool(S.to(dev), A.to(dev), X.to(dev), flags=flags, want_raw=True, graph_sizes=sizes.to(dev))[:3]
raw_ref = O.dense_connect(S, A)
torch.testing.assert_close(xp.cpu(), O.reduce_dense(S, X), rtol=2e-4, atol=2e-4)
torch.testing.assert_close(raw.cpu(), raw_ref, rtol=2e-4, atol=2e-4)
torch.testing.assert_close(pooled.cpu(), O.postprocess_dense(raw_ref, True, True, False, False), rtol=2e-4, atol=2e-4)
print('ok')
"""
    env = dict(os.environ)
    env.pop("TGP_MEDIUM_WAVES", None)
    if waves != "auto":
        env["TGP_MEDIUM_WAVES"] = waves
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


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
