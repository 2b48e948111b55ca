"""Round-3 GPU parity tests: the native MLPSelect (select/mlp_select.py:105-147) against the
oracle's restatement, its gradients against torch autograd, and the selectors through ``get_pooler``."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


SHAPES = [  # (leading shape, F, K)
    ((2048, 60), 32, 20),      # C3: PROTEINS-shaped MinCut batch
    ((32, 1024), 64, 128),     # C2
    ((3, 50), 16, 8), ((1, 37), 7, 5), ((5, 33), 33, 31), ((2, 100), 130, 65), ((4, 64), 96, 256),
    ((2, 40), 300, 200),       # F * K beyond the LDS budget: W walks through LDS in slices
    ((777,), 12, 10),          # unbatched [N, F]
    ((2, 70), 24, 300),        # K > 256: tiled GEMM + softmax kernel
]


@pytest.mark.parametrize("lead,F,K", SHAPES)
@pytest.mark.parametrize("with_mask", [False, True])
def test_mlp_select_kernel_vs_oracle(dev, lead, F, K, with_mask):
    """S = softmax(X W^T + b) * mask, rtol = atol = 1e-5 (north_star: 1e-5 rel for fp32 outputs) and rows sum to 1."""
    import tgp_oracle as O
    from tgp import kernels as Kn
    g = torch.Generator().manual_seed(hash((lead, F, K)) % 1000)
    x = torch.randn(*lead, F, generator=g)
    w = torch.randn(K, F, generator=g) * (2.0 / F ** 0.5)
    b = torch.randn(K, generator=g)
    mask = (torch.rand(*lead, generator=g) < 0.8) if with_mask else None
    want = O.mlp_select(x, [w], [b], mask)
    got = Kn.mlp_select(x.to(dev), w.to(dev), b.to(dev), None if mask is None else mask.to(dev)).cpu()
    assert got.shape == want.shape
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    rel = ((got - want).abs() / want.abs().clamp_min(1e-30))[want > 1e-6]
    assert rel.numel() == 0 or float(rel.max()) < 2e-5, float(rel.max())
    if with_mask:
        assert bool((got[~mask] == 0).all())
    # no bias
    got = Kn.mlp_select(x.to(dev), w.to(dev), None, None).cpu()
    torch.testing.assert_close(got, O.mlp_select(x, [w], [torch.zeros(K)], None), rtol=1e-5, atol=1e-5)


def test_mlp_select_large_logits_and_misaligned_views(dev):
    """Logits of several hundred (softmax saturates: the max-subtraction must hold; one fp32 ulp of such a logit is
    3e-5, which is the relative accuracy ANY fp32 evaluation of the softmax can have -- the oracle is evaluated in
    float64 and the bound is 2e-4), and x / weight handed over as non-contiguous views (made contiguous at the
    boundary)."""
    import tgp_oracle as O
    from tgp import kernels as Kn
    g = torch.Generator().manual_seed(3)
    x = torch.randn(200, 40, generator=g) * 20
    w = torch.randn(24, 40, generator=g)
    b = torch.randn(24, generator=g) * 5
    want = O.mlp_select(x.double(), [w.double()], [b.double()]).float()
    got = Kn.mlp_select(x.to(dev), w.to(dev), b.to(dev), None).cpu()
    assert bool(torch.isfinite(got).all())
    torch.testing.assert_close(got, want, rtol=2e-4, atol=1e-6)
    torch.testing.assert_close(got.sum(-1), torch.ones(200), rtol=1e-5, atol=1e-5)
    xb = torch.randn(200, 80, generator=g)
    wb = torch.randn(24, 80, generator=g)
    got = Kn.mlp_select(xb.to(dev)[:, ::2], wb.to(dev)[:, 1::2], None, None).cpu()
    torch.testing.assert_close(got, O.mlp_select(xb[:, ::2], [wb[:, 1::2]], [torch.zeros(24)]), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("lead,F,K", [((6, 50), 16, 8), ((2, 64), 64, 128), ((300,), 20, 300)])
def test_mlp_select_gradients_vs_torch(dev, lead, F, K):
    """Backward of the one-pass selector (softmax gradient kernel + native GEMMs) against torch autograd of
    linear -> softmax -> mask."""
    from tgp import functions as Fn
    g = torch.Generator().manual_seed(1)
    x = torch.randn(*lead, F, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(K, F, generator=g) * 0.3).to(dev).requires_grad_(True)
    b = torch.randn(K, generator=g).to(dev).requires_grad_(True)
    mask = (torch.rand(*lead, generator=g) < 0.7).to(dev)
    up = torch.randn(*lead, K, generator=g).to(dev)
    s = Fn.mlp_select(x, w, b, mask)
    (s * up).sum().backward()
    got = [t.grad.clone() for t in (x, w, b)]
    for t in (x, w, b):
        t.grad = None
    ref = torch.softmax(torch.nn.functional.linear(x, w, b), -1) * mask.unsqueeze(-1)
    (ref * up).sum().backward()
    for a, t in zip(got, (x, w, b)):
        torch.testing.assert_close(a, t.grad, rtol=1e-4, atol=1e-5)


def test_mlp_select_module_uses_the_native_kernel(dev, monkeypatch):
    """MLPSelect.forward on a device batch never calls torch.softmax / F.linear for a single-Linear selector, and a
    multi-layer selector only for its hidden layers; outputs equal the oracle."""
    import tgp_oracle as O
    from tgp.select import MLPSelect
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 30, 12, generator=g)
    mask = torch.rand(4, 30, generator=g) < 0.9
    sel = MLPSelect(in_channels=12, k=6).to(dev)
    calls = []
    real_softmax = torch.softmax
    monkeypatch.setattr(torch, "softmax", lambda *a, **k: (calls.append("softmax"), real_softmax(*a, **k))[1])
    so = sel(x=x.to(dev), mask=mask.to(dev))
    assert calls == []
    monkeypatch.setattr(torch, "softmax", real_softmax)  # (the oracle below uses it)
    lin = sel.mlp.lins[0]
    want = O.mlp_select(x, [lin.weight.detach().cpu()], [lin.bias.detach().cpu()], mask)
    torch.testing.assert_close(so.s.cpu(), want, rtol=1e-5, atol=1e-5)
    sel2 = MLPSelect(in_channels=[12, 16], k=6, act="relu").to(dev)
    monkeypatch.setattr(torch, "softmax", lambda *a, **k: (calls.append("softmax"), real_softmax(*a, **k))[1])
    so2 = sel2(x=x.to(dev), mask=mask.to(dev))
    assert calls == []
    monkeypatch.setattr(torch, "softmax", real_softmax)
    ws = [l.weight.detach().cpu() for l in sel2.mlp.lins]
    bs = [l.bias.detach().cpu() for l in sel2.mlp.lins]
    torch.testing.assert_close(so2.s.cpu(), O.mlp_select(x, ws, bs, mask, act="relu"), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,N,K", [(7, 40, 6), (3, 200, 33), (64, 60, 20), (2, 300, 130)])
def test_mincut_loss_tail_kernel_vs_oracle(dev, B, N, K):
    """Both MinCut losses through the one-launch tail (inference path) equal the oracle's restatement of
    utils/losses.py:39-70 and the autograd-path values."""
    import tgp_oracle as O
    from tgp.utils import losses as L
    g = torch.Generator().manual_seed(B * 1000 + K)
    a = (torch.rand(B, N, N, generator=g) < 0.1).float()
    a = torch.maximum(a, a.transpose(1, 2))
    s = torch.softmax(torch.randn(B, N, K, generator=g), -1)
    raw = O.dense_connect(s, a)
    want_cut, want_ortho = O.mincut_loss(a, s, raw), O.orthogonality_loss(s)
    both = L.mincut_loss_terms(a.to(dev), s.to(dev), raw.to(dev)).mean(dim=1).cpu()
    torch.testing.assert_close(both[0], want_cut, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(both[1], want_ortho, rtol=1e-5, atol=1e-6)
    sg = s.to(dev).requires_grad_(True)
    torch.testing.assert_close(L.mincut_loss(a.to(dev), sg, raw.to(dev)).detach().cpu(), both[0], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(L.orthogonality_loss(sg).detach().cpu(), both[1], rtol=1e-5, atol=1e-6)


def test_all_gather_sparse_over_rccl_one_rank_group(dev):
    """The variable-size gather of pooled sparse outputs (merge rule tgp/data/collate.py:144-153) through
    REAL RCCL collectives on the device: a one-rank process group with ``force_collective`` runs the count exchange, the
    padded payload gathers and the offset merge; the merged result must equal the local one bit for bit."""
    import os
    import socket
    import torch.distributed as dist
    from tgp.connect import SparseConnect
    from tgp.distributed import all_gather_sparse
    from tgp.reduce import BaseReduce
    from tgp.select import TopkSelect
    created = False
    if not dist.is_initialized():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        created = True
    try:
        g = torch.Generator().manual_seed(0)
        sizes = torch.randint(20, 61, (300,), generator=g)
        n = int(sizes.sum())
        batch = torch.repeat_interleave(torch.arange(300), sizes).to(dev)
        start = (torch.cumsum(sizes, 0) - sizes).to(dev)
        src = torch.arange(n, device=dev).repeat_interleave(2)
        dst = start[batch[src]] + (torch.rand(src.numel(), device=dev) * sizes.to(dev)[batch[src]]).long()
        key = torch.unique(torch.cat([src * n + dst, dst * n + src]))
        ei = torch.stack([key // n, key % n])
        x = torch.randn(n, 16, device=dev)
        ew = torch.rand(ei.size(1), device=dev) + 0.5
        with torch.no_grad():
            so = TopkSelect(in_channels=16, ratio=0.5).to(dev)(x=x, batch=batch)
            xp, bp = BaseReduce()(x, so, batch=batch)
            pe, pw = SparseConnect()(ei, so, edge_weight=ew, batch_pooled=bp)
        mx, me, mw, mb = all_gather_sparse(xp, pe, pw, bp, 300, force_collective=True)
        assert mx.data_ptr() != xp.data_ptr()  # went through the gather buffers, not the early return
        assert torch.equal(mx, xp) and torch.equal(me, pe) and torch.equal(mw, pw) and torch.equal(mb, bp)
    finally:
        if created:
            dist.destroy_process_group()


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


def test_graclus_select_hands_its_csr_offsets_to_sparse_connect(dev):
    """GraclusSelect attaches the CSR offsets of the row-sorted list it walked; SparseConnect uses them only for that
    very tensor object, unmodified; results equal the route without them; a list with ids outside [0, N) still raises."""
    from tgp import kernels
    from tgp.connect import SparseConnect
    from tgp.select import GraclusSelect
    g = torch.Generator().manual_seed(2)
    n = 50_000
    a, b = torch.randint(0, n, (2, 200_000), generator=g)
    ei = torch.stack([torch.cat([a, b]), torch.cat([b, a])])
    ei = ei[:, torch.argsort(ei[0] * n + ei[1])].contiguous().to(dev)
    ew = torch.rand(ei.size(1), generator=g).to(dev) + 0.1
    so = GraclusSelect()(ei, ew, num_nodes=n)
    ptr = so.edge_csr_for(ei)
    assert ptr is not None and ptr.dtype == torch.int32 and ptr.numel() == n + 1
    assert so.edge_csr_for(ei.clone()) is None            # another object: not trusted
    want = kernels.coalesce_edges(ei, ew, so.cluster_index, so.num_supernodes, "sum", True, route="general")
    for route in ("staged", "fused"):
        got = kernels.coalesce_edges(ei, ew, so.cluster_index, so.num_supernodes, "sum", True, route=route,
                                     csr=(ptr, None))
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), route
    oi, ow = SparseConnect()(ei, so, edge_weight=ew)
    assert torch.equal(oi, want[0]) and torch.equal(ow, want[1])
    ei[0, 0] = ei[0, 0]                                   # in-place write bumps the version counter
    assert so.edge_csr_for(ei) is None
    import pickle
    pickle.loads(pickle.dumps(so))                        # (a weak reference would not pickle)
    # stale / wrong offsets (ids outside [0, N) clamp into them): refused, and the general route reports the ids
    bad = ei.clone()
    bad[0, -1] = n + 5
    bad_ptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    from tgp import _native as N
    N.check(N.lib().tgp_rowptr_from_sorted_i64(N.ptr(bad[0].contiguous()), bad.size(1), n, N.ptr(bad_ptr),
                                               N.stream_ptr(dev)), "rowptr")
    assert int(bad_ptr[n]) == bad.size(1) - 1             # clamped: no out-of-bounds write, visibly incomplete
    with pytest.raises(IndexError):
        kernels.coalesce_edges(bad, ew, so.cluster_index, so.num_supernodes, "sum", True, csr=(bad_ptr, None),
                               assign_index=kernels.build_assign_index(so.cluster_index, so.num_supernodes))


# ------------------------------------------------------------------------------ NDPSelect, one large graph (r3)
def _undirected(n, m, seed):
    g = torch.Generator().manual_seed(seed)
    a = torch.randint(0, n, (m,), generator=g)
    b = torch.randint(0, n, (m,), generator=g)
    keep = a != b
    a, b = a[keep], b[keep]
    key = torch.unique(torch.cat([a * n + b, b * n + a]))
    return torch.stack([key // n, key % n])


def test_ndp_large_graph_partition_contract_vs_scipy(dev):
    """A 50 000-node graph (beyond the one-workgroup kernel) is partitioned by the chip-wide LOBPCG (tgp_ndp_large_*):
    its Rayleigh quotient equals scipy's largest eigenvalue of Ls = I - D^-1/2 A D^-1/2 within 1e-6, the residual meets
    the tolerance, and the partition is the sign pattern of the iterate (or the reference's random fallback when the
    cut test says so), with both sides non-empty."""
    import numpy as np
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    from tgp import kernels as K
    n = 50_000
    ei = _undirected(n, 250_000, 3)
    w = torch.rand(ei.size(1) // 1, generator=torch.Generator().manual_seed(4)) + 0.5
    # symmetric weights: w(u,v) = w(v,u)
    key = torch.minimum(ei[0], ei[1]) * n + torch.maximum(ei[0], ei[1])
    w = ((key * 2654435761) % 1000).float() / 1000 + 0.5
    A = sp.coo_matrix((w.double().numpy(), (ei[0].numpy(), ei[1].numpy())), shape=(n, n)).tocsr()
    deg = np.asarray(A.sum(1)).reshape(-1)
    dis = np.where(deg > 0, 1.0 / np.sqrt(np.maximum(deg, 1e-300)), 0.0)
    Ls = sp.eye(n) - sp.diags(dis) @ A @ sp.diags(dis)
    lam_ref = float(spla.eigsh(Ls.tocsc(), k=1, which="LA", tol=1e-10, return_eigenvectors=False)[0])
    eid, wd = ei.to(dev), w.to(dev)
    indptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    K.rowptr_from_sorted(eid[0], n, indptr)
    keep = torch.zeros(n, dtype=torch.uint8, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    info, state = K.ndp_partition_large(indptr, eid[1], wd, 0, n, 7, keep, status, want_state=True)
    assert int(status.item()) == 0
    assert abs(state["lambda"] - lam_ref) <= 1e-6 * lam_ref, (state, lam_ref)
    assert state["residual_sq"] <= (1e-6 * state["lambda"]) ** 2 * 1.0001
    kb = keep.bool().cpu()
    assert 0 < int(kb.sum()) < n
    z = np.where(kb.numpy(), 1.0, -1.0)
    L = sp.diags(deg) - A
    cut = float(z @ (L @ z)) / (2.0 * A.sum())
    if int(info.item()) >= 0:   # spectral partition kept: its cut passed the reference's test
        assert cut >= 0.5 and abs(cut - state["cut"]) < 1e-9
    else:                       # random fallback (ndp_select.py:171-185): node 0 kept, node 1 dropped
        assert bool(kb[0]) and not bool(kb[1])


def test_ndp_select_single_large_graph_stays_on_device(dev, monkeypatch):
    """NDPSelect on one 30 000-node graph: no scipy eigen-solver is called (r2 sent such graphs to eigsh on the host);
    the selector's outputs have the reference's structure (kept nodes ascending, one-to-one S, weights 1)."""
    import scipy.sparse.linalg as spla
    from tgp.select import NDPSelect

    def boom(*a, **k):
        raise AssertionError("host eigen-solver called")
    monkeypatch.setattr(spla, "eigsh", boom)
    n = 30_000
    ei = _undirected(n, 120_000, 5).to(dev)
    so = NDPSelect()(ei, None, num_nodes=n)
    ni = so.node_index
    assert ni.is_cuda and 0 < ni.numel() < n and bool((ni[1:] > ni[:-1]).all())
    assert torch.equal(so.cluster_index, torch.arange(ni.numel(), device=dev))
    assert bool((so.weight == 1).all())
    # a batch that mixes small graphs with one large graph: the small ones keep the one-workgroup kernel
    sizes = [30, 45, 5000, 20]
    eis, bs, off = [], [], 0
    for gi, m in enumerate(sizes):
        eis.append(_undirected(m, 3 * m, 10 + gi) + off)
        bs.append(torch.full((m,), gi))
        off += m
    ei2, batch = torch.cat(eis, 1).to(dev), torch.cat(bs).to(dev)
    so2 = NDPSelect()(ei2, None, batch=batch, num_nodes=off)
    kept_per_graph = torch.bincount(batch[so2.node_index], minlength=len(sizes))
    assert bool((kept_per_graph > 0).all()) and bool((kept_per_graph < torch.tensor(sizes, device=dev)).all())


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


@pytest.mark.parametrize("transposed", [False, True])
def test_mincut_terms_from_inside_the_small_graph_kernel(dev, transposed):
    """Batches of small graphs: the pooling kernel itself leaves the per-graph tails of MinCut's losses (trace of the raw
    S^T A S, trace(S^T D S), the orthogonality norm); equal to the separate loss kernels and to the oracle, and the
    pooled outputs are unchanged by asking for them."""
    import tgp_oracle as O
    from tgp import kernels as Kn
    from tgp.utils import losses as L
    g = torch.Generator().manual_seed(7)
    B, N, K, F = 100, 57, 19, 30
    n_b = torch.randint(10, N + 1, (B,), generator=g)
    mask = torch.arange(N).unsqueeze(0) < n_b.unsqueeze(1)
    a = (torch.rand(B, N, N, generator=g) < 0.1).float() * torch.rand(B, N, N, generator=g)
    a = a * mask.unsqueeze(1) * mask.unsqueeze(2)           # directed, weighted, zero padding
    s = torch.softmax(torch.randn(B, N, K, generator=g), -1) * mask.unsqueeze(-1)
    x = torch.randn(B, N, F, generator=g) * mask.unsqueeze(-1)
    ad = a.to(dev)
    adj_arg = ad.transpose(1, 2).contiguous().transpose(1, 2) if transposed else ad  # same values, transposed memory
    flags = Kn.dense_flags(True, True, True, False)
    xp, raw, ap, terms = Kn.dense_pool(s.to(dev), adj_arg, x.to(dev), flags, want_raw=True, mincut_terms=True)
    assert terms is not None and terms.shape == (2, B)
    xp0, raw0, ap0 = Kn.dense_pool(s.to(dev), adj_arg, x.to(dev), flags, want_raw=True)
    assert torch.equal(xp, xp0) and torch.equal(raw, raw0) and torch.equal(ap, ap0)
    want = L.mincut_loss_terms(ad, s.to(dev), raw0)
    torch.testing.assert_close(terms, want, rtol=1e-5, atol=1e-6)
    rawc = O.dense_connect(s, a)
    torch.testing.assert_close(terms.mean(1).cpu(), torch.stack([O.mincut_loss(a, s, rawc), O.orthogonality_loss(s)]),
                               rtol=1e-5, atol=1e-6)


def test_mincut_pooler_small_graph_batch_losses_vs_oracle(dev):
    """get_pooler('mincut') forward (no grad) on 128 small graphs: the losses that come out of the fused path equal the
    autograd path's and the oracle's."""
    import tgp_oracle as O
    from tgp.poolers import get_pooler
    g = torch.Generator().manual_seed(3)
    sizes = torch.randint(12, 50, (128,), generator=g)
    n = int(sizes.sum())
    batch = torch.repeat_interleave(torch.arange(128), sizes)
    start = torch.cumsum(sizes, 0) - sizes
    src = torch.arange(n).repeat_interleave(2)
    dst = start[batch[src]] + (torch.rand(src.numel(), generator=g) * sizes[batch[src]]).long()
    key = torch.unique(torch.cat([src * n + dst, dst * n + src]))
    ei = torch.stack([key // n, key % n])
    ei = ei[:, ei[0] != ei[1]]
    x = torch.randn(n, 16, generator=g)
    pooler = get_pooler("mincut", in_channels=16, k=7).to(dev).eval()
    with torch.no_grad():
        out = pooler(x=x.to(dev), adj=ei.to(dev), batch=batch.to(dev))
    lin = pooler.selector.mlp.lins[0]
    ref = O.dense_pool("mincut", x, ei, None, batch, [lin.weight.detach().cpu()], [lin.bias.detach().cpu()])
    torch.testing.assert_close(out.x.cpu(), ref["x"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(out.edge_index.cpu(), ref["edge_index"], rtol=1e-5, atol=1e-5)
    for name in ("cut_loss", "ortho_loss"):
        torch.testing.assert_close(out.loss[name].cpu(), ref["loss"][name], rtol=1e-5, atol=1e-6)
    out_g = pooler(x=x.to(dev).requires_grad_(True), adj=ei.to(dev), batch=batch.to(dev))
    for name in ("cut_loss", "ortho_loss"):
        torch.testing.assert_close(out_g.loss[name].detach(), out.loss[name], rtol=1e-5, atol=1e-6)


def test_diffpool_loss_tail_and_sorted_dense_batch(dev):
    """DiffPool's inference losses through the one-launch tail equal the autograd path and the oracle; to_dense_batch
    for a sorted batch vector (no memsets) equals the scatter form, padding and mask included."""
    import tgp_oracle as O
    from tgp import kernels as Kn
    from tgp.utils import losses as L
    g = torch.Generator().manual_seed(9)
    B, N, K = 20, 33, 6
    a = (torch.rand(B, N, N, generator=g) < 0.15).float()
    a = torch.maximum(a, a.transpose(1, 2))
    s = torch.softmax(torch.randn(B, N, K, generator=g), -1)
    for normalize, coeff in ((False, 1.0), (True, 0.3)):
        scale = coeff / a.numel() if normalize else coeff
        both = Kn.diffpool_loss_tail(s.to(dev), a.to(dev), None, scale, 2.0 / (B * N)).cpu()
        torch.testing.assert_close(both[0], O.link_pred_loss(s, a, normalize) * coeff, rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(both[1], O.entropy_loss(s, B * N) * 2.0, rtol=1e-5, atol=1e-7)
    sg = s.to(dev).requires_grad_(True)
    torch.testing.assert_close(L.link_pred_loss(sg, a.to(dev), False).detach().cpu(),
                               Kn.diffpool_loss_tail(s.to(dev), a.to(dev), None, 1.0, 1.0).cpu()[0], rtol=1e-5, atol=1e-7)
    # sorted to_dense_batch vs the oracle (ragged sizes, an empty graph in the middle)
    sizes = torch.tensor([5, 0, 17, 1, 9])
    batch = torch.repeat_interleave(torch.arange(5), sizes)
    x = torch.randn(int(sizes.sum()), 7, generator=g)
    from tgp.src import to_dense_batch
    got, mask = to_dense_batch(x.to(dev), batch.to(dev), batch_size=5)
    want, wmask = O.to_dense_batch(x, batch)
    assert got.shape[0] == 5 and torch.equal(got.cpu()[:, : want.size(1)], want) and torch.equal(mask.cpu()[:, : want.size(1)], wmask)


# ------------------------------------------------------------------------------ edge cases of the r3 entry points
def test_round3_entry_points_on_empty_and_degenerate_inputs(dev):
    """Empty / ragged / degenerate inputs the reference's own tests exercise for the older operators, for the new ones:
    zero rows, a single row, empty graphs inside a batch, lists without edges, isolated nodes."""
    import tgp_oracle as O
    from tgp import kernels as Kn
    from tgp.select import MLPSelect, NDPSelect, TopkSelect
    g = torch.Generator().manual_seed(0)
    # MLPSelect: no rows, one row
    w, b = torch.randn(5, 3, generator=g).to(dev), torch.randn(5, generator=g).to(dev)
    assert Kn.mlp_select(torch.zeros(0, 3, device=dev), w, b, None).shape == (0, 5)
    one = torch.randn(1, 3, generator=g)
    torch.testing.assert_close(Kn.mlp_select(one.to(dev), w, b, None).cpu(),
                               O.mlp_select(one, [w.cpu()], [b.cpu()]), rtol=1e-5, atol=1e-6)
    assert Kn.softmax_bwd(torch.zeros(0, 5, device=dev), torch.zeros(0, 5, device=dev)).shape == (0, 5)
    # fully masked batch: S is exactly zero
    sel = MLPSelect(in_channels=3, k=4).to(dev)
    so = sel(x=torch.randn(2, 6, 3, generator=g).to(dev), mask=torch.zeros(2, 6, dtype=torch.bool, device=dev))
    assert float(so.s.detach().abs().max()) == 0.0
    # TopkSelect min_score: graphs of one node, an empty graph id in the middle of the batch vector
    batch = torch.tensor([0, 0, 0, 2, 3, 3]).to(dev)   # graph 1 owns no node
    x = torch.randn(6, 4, generator=g)
    tk = TopkSelect(in_channels=4, ratio=None, min_score=0.4).to(dev)
    so = tk(x=x.to(dev), batch=batch)
    ni, ci, wt = O.topk_select(x, tk.weight.detach().cpu(), None, batch.cpu(), 0.4, "tanh")
    assert torch.equal(so.node_index.cpu(), ni)
    torch.testing.assert_close(so.weight.cpu(), wt, rtol=1e-5, atol=1e-7)
    # coalesce routes: no edges, one edge, every edge a self loop
    cl = torch.tensor([0, 0, 1, 1, 2]).to(dev)
    for route in ("fused", "staged", "general"):
        ei0 = torch.zeros(2, 0, dtype=torch.long, device=dev)
        out = Kn.coalesce_edges(ei0, None, cl, 3, "sum", True, route=route)
        assert out[0].shape == (2, 0)
        ei1 = torch.tensor([[1], [4]], device=dev)
        out = Kn.coalesce_edges(ei1, torch.tensor([2.0], device=dev), cl, 3, "sum", True, route=route)
        assert out[0].tolist() == [[0], [2]] and out[1].tolist() == [2.0]
        loops = torch.tensor([[0, 1, 2, 3], [1, 0, 3, 2]], device=dev)   # all inside their clusters
        out = Kn.coalesce_edges(loops, None, cl, 3, "sum", True, route=route)
        assert out[0].shape == (2, 0)
    # NDPSelect: a large graph with isolated nodes and a second component (chip-wide route), still a valid partition
    n = 3000
    a = torch.arange(0, 2000 - 1)
    ei = torch.stack([torch.cat([a, a + 1]), torch.cat([a + 1, a])]).to(dev)    # a path on nodes 0..1999, rest isolated
    so = NDPSelect()(ei, None, num_nodes=n)
    assert 0 < so.num_supernodes < n and bool((so.node_index[1:] > so.node_index[:-1]).all())


# ----------------------------------------------------------------------------- fused backward of the small-graph kernel
def _ragged_dense_batch(B, Nmax, K, F, seed, dev, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    n_b = torch.randint(max(2, Nmax // 3), Nmax + 1, (B,), generator=g)
    n_b[0] = Nmax
    mask = torch.arange(Nmax).unsqueeze(0) < n_b.unsqueeze(1)
    A = (torch.rand(B, Nmax, Nmax, generator=g) < 0.15).float() * torch.rand(B, Nmax, Nmax, generator=g)
    A = A * mask.unsqueeze(1) * mask.unsqueeze(2)
    X = torch.randn(B, Nmax, F, generator=g) * mask.unsqueeze(-1)
    logits = torch.randn(B, Nmax, K, generator=g)
    return A.to(dev, dtype), X.to(dev, dtype), logits.to(dev, dtype), mask.to(dev)


def _dense_pool_reference(S, A, X, rsl, dn, at, eps_ops, eps_loss):
    """Plain torch restatement (differentiable) of Reduce, Connect, post-processing and MinCut's per-graph terms:
    base_reduce.py:158-161, dense_conn.py:111-122, utils/ops.py:282-335, utils/losses.py:39-70."""
    xp = S.transpose(1, 2) @ X
    raw = S.transpose(1, 2) @ A @ S
    ap = raw
    if rsl:
        ap = ap * (1 - torch.eye(ap.size(-1), device=ap.device, dtype=ap.dtype))
    if dn:
        d = ap.sum(-2 if at else -1, keepdim=True)
        d = torch.sqrt(d.clamp(min=eps_ops))
        ap = (ap / d) / d.transpose(-2, -1)
    num = torch.einsum("bii->b", raw)
    den = torch.einsum("bnk,bn,bnk->b", S, A.sum(-1), S)
    cut = -(num / (den + eps_loss))
    sts = S.transpose(1, 2) @ S
    sts = sts / torch.norm(sts, dim=(-2, -1), keepdim=True)
    k = S.size(-1)
    ortho = torch.norm(sts - torch.eye(k, device=S.device, dtype=S.dtype) / k ** 0.5, dim=(-2, -1))
    return xp, raw, ap, torch.stack([cut, ortho])


@pytest.mark.gpu
@pytest.mark.parametrize("Nmax,K,F", [(17, 5, 8), (40, 20, 32), (64, 32, 32), (60, 20, 3)])
@pytest.mark.parametrize("rsl,dn,at", [(True, True, True), (True, True, False), (False, True, True),
                                       (True, False, False), (False, False, True)])
@pytest.mark.parametrize("transposed_view", [False, True])
def test_small_graph_fused_backward_vs_autograd(dev, Nmax, K, F, rsl, dn, at, transposed_view):
    """tgp_dense_pool_small_bwd_f32 (one launch) against torch autograd of the fp64 restatement: gradients of S and X
    from random upstream gradients of x_pool, adj_pool, the raw S^T A S and both per-graph MinCut terms."""
    from tgp import functions as Fn, kernels as K_
    from tgp.utils import losses, ops
    B = 70
    A, X, logits, mask = _ragged_dense_batch(B, Nmax, K, F, seed=Nmax * 131 + K, dev=dev)
    flags = K_.dense_flags(rsl, dn, at, False)
    g = torch.Generator().manual_seed(7)
    w_x, w_a = torch.randn(B, K, F, generator=g).to(dev), torch.randn(B, K, K, generator=g).to(dev)
    w_r, w_t = torch.randn(B, K, K, generator=g).to(dev), torch.randn(2, B, generator=g).to(dev)

    # fp64 reference
    l64 = logits.double().requires_grad_(True)
    x64 = X.double().requires_grad_(True)
    S64 = torch.softmax(l64, -1) * mask.unsqueeze(-1)
    out = _dense_pool_reference(S64, A.double(), x64, rsl, dn, at, float(ops.eps), float(losses.eps))
    # DiffPool's two losses (utils/losses.py:644-658), scaled as poolers/diffpool.py:262-284 scales them
    w_d = torch.tensor([1.3, -0.7], device=dev)
    ref_diff = [0.37 * torch.norm(A.double() - S64 @ S64.transpose(1, 2), p=2),
                (-(S64 * torch.log(S64 + float(losses.eps))).sum()) / int(mask.sum())]
    ref_loss = sum((o * w.double()).sum() for o, w in zip(out, (w_x, w_r, w_a, w_t)))
    ref_loss = ref_loss + ref_diff[0] * w_d[0].double() + ref_diff[1] * w_d[1].double()
    ref_loss.backward()

    l32 = logits.clone().requires_grad_(True)
    x32 = X.clone().requires_grad_(True)
    S32 = torch.softmax(l32, -1) * mask.unsqueeze(-1)
    adj = A.transpose(1, 2).contiguous().transpose(1, 2) if transposed_view else A
    link_scale, ent_scale = 0.37, 1.0 / int(mask.sum())
    xp, raw, ap, terms, diff = Fn.dense_pool_small(S32, adj, x32, flags, True, True, (link_scale, ent_scale))
    for got, want, name in zip((xp, raw, ap, terms), out, ("x_pool", "raw", "adj_pool", "terms")):
        torch.testing.assert_close(got, want.float(), rtol=2e-4, atol=2e-5, msg=lambda m, n=name: f"{n}: {m}")
    torch.testing.assert_close(diff, torch.stack(ref_diff).float(), rtol=2e-4, atol=1e-5)
    loss = (xp * w_x).sum() + (raw * w_r).sum() + (ap * w_a).sum() + (terms * w_t).sum() + (diff * w_d).sum()
    loss.backward()
    scale_s = l64.grad.abs().max().item()
    torch.testing.assert_close(l32.grad, l64.grad.float(), rtol=2e-3, atol=2e-4 * max(scale_s, 1.0))
    torch.testing.assert_close(x32.grad, x64.grad.float(), rtol=2e-4, atol=2e-5 * max(x64.grad.abs().max().item(), 1.0))
    assert x32.grad[~mask].abs().max().item() == 0.0  # padded rows of X get exact zeros


@pytest.mark.gpu
def test_small_graph_fused_backward_partial_upstreams(dev):
    """Only some outputs feed the loss (x_pool alone; adj_pool alone; terms alone): the missing upstream gradients are
    NULL in the C call, the result equals autograd's."""
    from tgp import functions as Fn, kernels as K_
    from tgp.utils import losses, ops
    B, Nmax, K, F = 80, 33, 7, 16
    A, X, logits, mask = _ragged_dense_batch(B, Nmax, K, F, seed=5, dev=dev)
    flags = K_.dense_flags(True, True, True, False)
    for pick in ("x", "adj", "terms"):
        l64 = logits.double().requires_grad_(True)
        x64 = X.double().requires_grad_(True)
        out = _dense_pool_reference(torch.softmax(l64, -1) * mask.unsqueeze(-1), A.double(), x64, True, True, True,
                                    float(ops.eps), float(losses.eps))
        ref = {"x": out[0].square().sum(), "adj": out[2].square().sum(), "terms": out[3].mean(dim=1).sum()}[pick]
        ref.backward()
        l32 = logits.clone().requires_grad_(True)
        x32 = X.clone().requires_grad_(True)
        xp, raw, ap, terms, _ = Fn.dense_pool_small(torch.softmax(l32, -1) * mask.unsqueeze(-1), A, x32, flags,
                                                    False, pick == "terms")
        got = {"x": xp.square().sum(), "adj": ap.square().sum(), "terms": terms.mean(dim=1).sum()}[pick]
        got.backward()
        torch.testing.assert_close(l32.grad, l64.grad.float(), rtol=2e-3, atol=2e-4 * max(l64.grad.abs().max().item(), 1))
        if pick == "x":
            torch.testing.assert_close(x32.grad, x64.grad.float(), rtol=2e-4, atol=1e-4)
        else:
            assert x32.grad is None or x32.grad.abs().max().item() == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("alias", ["mincut", "diff"])
def test_dense_pooler_training_step_uses_the_fused_backward(dev, alias, monkeypatch):
    """get_pooler('mincut' / 'diff') on a PROTEINS-shaped sparse batch in training mode: the forward goes through the
    fused small-graph kernel, the backward through tgp_dense_pool_small_bwd_f32, and parameter / input gradients equal
    the operator-by-operator autograd path's (TGP_NO_SMALL_GRAPH_KERNEL-free check: the fused Function is switched off
    by patching dense_pool_is_small)."""
    from tgp import kernels as K_
    from tgp.poolers import get_pooler
    g = torch.Generator().manual_seed(3)
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
    real_bwd = K_.dense_pool_small_bwd
    monkeypatch.setattr(K_, "dense_pool_small_bwd", lambda *a, **k: (calls.append(1), real_bwd(*a, **k))[1])

    def step():
        pooler.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        out = pooler(x=x, adj=ei, batch=batch)
        loss = out.x.square().sum() + out.edge_index.square().sum() + sum(out.loss.values())
        loss.backward()
        return loss.detach(), x.grad, [p.grad.clone() for p in pooler.parameters()]

    fused = step()
    assert calls, "the fused backward did not run"
    monkeypatch.setattr(K_, "dense_pool_is_small", lambda *a: False)
    calls.clear()
    plain = step()
    assert not calls
    torch.testing.assert_close(fused[0], plain[0], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(fused[1], plain[1], rtol=2e-3, atol=2e-4 * max(plain[1].abs().max().item(), 1.0))
    for a, b in zip(fused[2], plain[2]):
        torch.testing.assert_close(a, b, rtol=2e-3, atol=2e-4 * max(b.abs().max().item(), 1.0))


# ----------------------------------------------------------------------------- NDPSelect: the list that needs no symmetrising
@pytest.mark.gpu
def test_ndp_symmetric_max_recognises_a_clean_list_and_refuses_the_rest(dev):
    """tgp_ndp_symmetric_max_f32: flag 0 and w = max(w, w_reverse) for a sorted, duplicate-free, loop-free, symmetric
    list; flag 1 for an unsorted list, a duplicate, a self loop, a missing reverse entry, an id out of range."""
    from tgp import kernels as K_
    g = torch.Generator().manual_seed(11)
    n = 500
    src = torch.randint(0, n, (3000,), generator=g)
    dst = torch.randint(0, n, (3000,), generator=g)
    keep = src != dst
    key = torch.unique(torch.cat([src[keep] * n + dst[keep], dst[keep] * n + src[keep]]))
    ei = torch.stack([key // n, key % n]).to(dev)
    w = torch.rand(ei.size(1), generator=g).to(dev)

    def run(e, ww):
        indptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
        K_.rowptr_from_sorted(e[0], n, indptr)
        out, flag = K_.ndp_symmetric_max(e, ww, n, indptr)
        return out, int(flag.item())

    out, flag = run(ei, w)
    assert flag == 0
    dense = torch.zeros(n, n, device=dev)
    dense[ei[0], ei[1]] = w
    torch.testing.assert_close(out, torch.maximum(dense, dense.t())[ei[0], ei[1]], rtol=0, atol=0)
    out1, flag1 = run(ei, None)
    assert flag1 == 0 and bool((out1 == 1).all())
    perm = torch.randperm(ei.size(1), generator=g).to(dev)
    assert run(ei[:, perm], w[perm])[1] == 1                                     # unsorted
    assert run(torch.cat([ei[:, :1], ei], 1), torch.cat([w[:1], w]))[1] == 1      # a duplicate
    loop = torch.tensor([[0], [0]], device=dev)
    assert run(torch.cat([loop, ei[:, ei[0] > 0]], 1), torch.cat([w[:1], w[ei[0] > 0]]))[1] == 1   # a self loop
    drop = torch.ones(ei.size(1), dtype=torch.bool, device=dev)
    drop[7] = False
    assert run(ei[:, drop], w[drop])[1] == 1                                     # reverse entry missing
    bad = ei.clone()
    bad[1, -1] = n + 3
    assert run(bad, w)[1] == 1                                                   # id out of range


@pytest.mark.gpu
def test_ndp_select_fast_and_general_preparation_agree(dev, monkeypatch):
    """NDPSelect on a batch of undirected graphs: the recognised-clean-list route and the two-coalesce route give the
    same SelectOutput (same kept nodes, same device adjacency for KronConnect)."""
    from tgp import kernels as K_
    from tgp.select import NDPSelect
    g = torch.Generator().manual_seed(4)
    sizes = torch.randint(10, 50, (40,), generator=g)
    n = int(sizes.sum())
    batch = torch.repeat_interleave(torch.arange(40), sizes)
    start = torch.cumsum(sizes, 0) - sizes
    src = torch.arange(n).repeat_interleave(2)
    dst = start[batch[src]] + (torch.rand(src.numel(), generator=g) * sizes[batch[src]]).long()
    keep = src != dst
    key = torch.unique(torch.cat([src[keep] * n + dst[keep], dst[keep] * n + src[keep]]))
    ei = torch.stack([key // n, key % n]).to(dev)
    w = torch.rand(ei.size(1), generator=g).to(dev)
    batch = batch.to(dev)
    sel = NDPSelect()
    torch.manual_seed(1)
    fast = sel(edge_index=ei, edge_weight=w, batch=batch, num_nodes=n)
    real = K_.ndp_symmetric_max
    monkeypatch.setattr(K_, "ndp_symmetric_max", lambda *a, **k: (real(*a, **k)[0], torch.ones(1, dtype=torch.int32, device=dev)))
    torch.manual_seed(1)
    general = sel(edge_index=ei, edge_weight=w, batch=batch, num_nodes=n)
    assert torch.equal(fast.node_index, general.node_index)
    for a, b in zip(fast._adj_device_csr, general._adj_device_csr):
        assert torch.equal(a, b)


# ----------------------------------------------------------------------------- native backwards of A8 and the entropy loss
@pytest.mark.gpu
@pytest.mark.parametrize("K", [7, 20, 128, 200, 513])
@pytest.mark.parametrize("rsl,dn,at", [(True, True, True), (True, True, False), (False, True, True),
                                       (False, True, False), (True, False, False)])
def test_postprocess_dense_backward_kernel_vs_autograd(dev, K, rsl, dn, at):
    """tgp_postprocess_dense_bwd_f32 against torch autograd of the elementwise form (utils/ops.py:282-335), including
    rows whose degree sum is below eps (the clamp blocks their gradient) and negative sums."""
    from tgp.utils import ops
    g = torch.Generator().manual_seed(K)
    B = 5
    raw = torch.rand(B, K, K, generator=g)
    raw[1] *= 0.0                       # a graph of all-zero sums
    raw[2, 0] = -raw[2, 0]              # a negative row
    raw[3, :, 1] = 0.0
    raw[3, 1, :] = 0.0                  # one isolated index
    w = torch.randn(B, K, K, generator=g).to(dev)
    r64 = raw.to(dev).double().requires_grad_(True)
    ref = ops._postprocess_dense_autograd(r64, rsl, dn, at, False)
    (ref * w.double()).sum().backward()
    r32 = raw.to(dev).requires_grad_(True)
    out = ops.postprocess_adj_pool_dense(r32, rsl, dn, at, False)
    torch.testing.assert_close(out, ref.float(), rtol=1e-5, atol=1e-5)
    (out * w).sum().backward()
    torch.testing.assert_close(r32.grad, r64.grad.float(), rtol=2e-4, atol=2e-5 * max(r64.grad.abs().max().item(), 1.0))


@pytest.mark.gpu
def test_entropy_loss_backward_kernel_vs_autograd(dev):
    from tgp.utils import losses
    g = torch.Generator().manual_seed(2)
    S = torch.softmax(torch.randn(6, 50, 9, generator=g), -1)
    S[0, 40:] = 0.0                      # padded rows
    s64 = S.to(dev).double().requires_grad_(True)
    ref = (-(s64 * torch.log(s64 + float(losses.eps))).sum()) / 123 * 0.7
    ref.backward()
    s32 = S.to(dev).requires_grad_(True)
    out = losses.entropy_loss(s32, 123) * 0.7
    out.backward()
    torch.testing.assert_close(out, ref.float(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(s32.grad, s64.grad.float(), rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,K", [(6, 150, 12), (32, 300, 128), (3, 90, 33)])
def test_mincut_terms_function_vs_autograd(dev, B, N, K):
    """_MinCutTermsFn (three forward kernels, one native backward tail) against fp64 autograd of utils/losses.py:39-70:
    values, and gradients with respect to S and to the raw S^T A S."""
    from tgp.utils import losses
    g = torch.Generator().manual_seed(B * 7 + K)
    A = (torch.rand(B, N, N, generator=g) < 0.1).float()
    A = torch.maximum(A, A.transpose(1, 2)).to(dev)
    S0 = torch.softmax(torch.randn(B, N, K, generator=g), -1).to(dev)
    wt = torch.randn(2, B, generator=g).to(dev)
    s64 = S0.double().requires_grad_(True)
    raw64 = (s64.transpose(1, 2) @ A.double() @ s64).detach().requires_grad_(True)
    num = torch.einsum("bii->b", raw64)
    den = torch.einsum("bnk,bn,bnk->b", s64, A.double().sum(-1), s64)
    cut = -(num / (den + float(losses.eps)))
    sts = s64.transpose(1, 2) @ s64
    sts = sts / torch.norm(sts, dim=(-2, -1), keepdim=True)
    ortho = torch.norm(sts - torch.eye(K, device=dev, dtype=torch.float64) / K ** 0.5, dim=(-2, -1))
    ref = torch.stack([cut, ortho])
    (ref * wt.double()).sum().backward()
    s32 = S0.clone().requires_grad_(True)
    raw32 = raw64.detach().float().requires_grad_(True)
    got = losses._MinCutTermsFn.apply(A, s32, raw32, None)
    torch.testing.assert_close(got, ref.float(), rtol=1e-5, atol=1e-6)
    (got * wt).sum().backward()
    torch.testing.assert_close(s32.grad, s64.grad.float(), rtol=2e-4, atol=2e-5 * max(s64.grad.abs().max().item(), 1.0))
    torch.testing.assert_close(raw32.grad, raw64.grad.float(), rtol=1e-5, atol=1e-7)


# ----------------------------------------------------------------------------- the selector folded into the small-graph kernel
@pytest.mark.gpu
@pytest.mark.parametrize("Nmax,K,F", [(17, 5, 8), (40, 20, 32), (64, 32, 32), (60, 20, 3)])
@pytest.mark.parametrize("with_mask,with_bias,transposed_view", [(True, True, False), (False, False, True), (True, False, True)])
def test_dense_pool_select_fold_vs_separate_kernels(dev, Nmax, K, F, with_mask, with_bias, transposed_view):
    """tgp_dense_pool_select_f32 (MLPSelect's Linear + softmax + mask inside the pooling kernel): S against the oracle's
    mlp_select in fp64, pooled outputs and MinCut terms against the two-launch path fed with that S."""
    import tgp_oracle as O
    from tgp import kernels as K_
    B = 70
    A, X, _, mask = _ragged_dense_batch(B, Nmax, K, F, seed=Nmax + 3 * K, dev=dev)
    g = torch.Generator().manual_seed(K)
    W = (torch.randn(K, F, generator=g) * 0.7).to(dev)
    bias = torch.randn(K, generator=g).to(dev) if with_bias else None
    m = mask if with_mask else None
    adj = A.transpose(1, 2).contiguous().transpose(1, 2) if transposed_view else A
    flags = K_.dense_flags(True, True, True, False)
    s, xp, raw, ap, terms = K_.dense_pool_select(X, adj, W, bias, m, flags, want_raw=True, mincut_terms=True)
    ref_mask = m.cpu() if m is not None else torch.ones(B, Nmax, dtype=torch.bool)
    want = O.mlp_select(X.cpu().double(), [W.cpu().double()], [None if bias is None else bias.cpu().double()], ref_mask)
    torch.testing.assert_close(s.cpu(), want.float(), rtol=1e-5, atol=1e-6)
    xp2, raw2, ap2, terms2 = K_.dense_pool(s, adj, X, flags, want_raw=True, mincut_terms=True)
    torch.testing.assert_close(xp, xp2, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(raw, raw2, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(ap, ap2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(terms, terms2, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("alias", ["mincut", "diff"])
def test_pooler_inference_with_the_folded_selector_equals_the_separate_path(dev, alias, monkeypatch):
    """get_pooler('mincut' / 'diff') in eval mode on a PROTEINS-shaped sparse batch: Select + Reduce + Connect as one
    launch gives the same PoolingOutput (so.s, x, adjacency, losses) as MLPSelect's kernel followed by the pooling kernel."""
    from tgp import kernels as K_
    from tgp.poolers import get_pooler
    g = torch.Generator().manual_seed(9)
    sizes = torch.randint(20, 61, (80,), generator=g)
    n = int(sizes.sum())
    batch = torch.repeat_interleave(torch.arange(80), sizes).to(dev)
    start = (torch.cumsum(sizes, 0) - sizes).to(dev)
    src = torch.arange(n, device=dev).repeat_interleave(2)
    dst = start[batch[src]] + (torch.rand(src.numel(), device=dev) * sizes.to(dev)[batch[src]]).long()
    keep = src != dst
    key = torch.unique(torch.cat([src[keep] * n + dst[keep], dst[keep] * n + src[keep]]))
    ei = torch.stack([key // n, key % n])
    x = torch.randn(n, 32, device=dev)
    torch.manual_seed(0)
    pooler = get_pooler(alias, in_channels=32, k=20).to(dev).eval()
    calls = []
    real, real_sparse = K_.dense_pool_select, K_.dense_pool_select_sparse
    monkeypatch.setattr(K_, "dense_pool_select", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    # (r5: MinCut on sparse inputs takes the form of the same kernel that reads the un-padded batch)
    monkeypatch.setattr(K_, "dense_pool_select_sparse", lambda *a, **k: (calls.append(2), real_sparse(*a, **k))[1])
    with torch.no_grad():
        folded = pooler(x=x, adj=ei, batch=batch)
    assert calls == [2], "the folded kernel did not run"  # (r5: both poolers read the un-padded batch in inference)
    monkeypatch.setattr(type(pooler), "_select_reduce_connect", lambda self, *a: None)
    monkeypatch.setattr(type(pooler), "_select_reduce_connect_sparse", lambda self, *a: None)
    with torch.no_grad():
        plain = pooler(x=x, adj=ei, batch=batch)
    torch.testing.assert_close(folded.so.s, plain.so.s, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(folded.x, plain.x, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(folded.edge_index, plain.edge_index, rtol=1e-5, atol=1e-5)
    for k in plain.loss:
        torch.testing.assert_close(folded.loss[k], plain.loss[k], rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("transposed", [False, True])
def test_densify_together_equals_the_separate_functions(dev, transposed):
    """to_dense_batch zero-filling the adjacency buffer in its own launch + to_dense_adj scattering into it (what the
    dense poolers' preprocessing runs) against the two stand-alone functions: same x, mask, adjacency (duplicates
    summed), gradients of x and of the edge weights."""
    from tgp.connect import DenseConnect
    from tgp.reduce import BaseReduce
    from tgp.src import DenseSRCPooling, to_dense_adj, to_dense_batch
    g = torch.Generator().manual_seed(6)
    sizes = torch.tensor([5, 1, 17, 3, 9])
    n = int(sizes.sum())
    batch = torch.repeat_interleave(torch.arange(5), sizes).to(dev)
    start = (torch.cumsum(sizes, 0) - sizes).to(dev)
    src = torch.arange(n, device=dev).repeat_interleave(3)
    dst = start[batch[src]] + (torch.rand(src.numel(), device=dev) * sizes.to(dev)[batch[src]]).long()
    ei = torch.stack([src, dst])                       # duplicates and self loops included
    ew = torch.rand(ei.size(1), generator=g).to(dev).requires_grad_(True)
    x = torch.randn(n, 7, generator=g).to(dev).requires_grad_(True)
    pool = DenseSRCPooling(reducer=BaseReduce(), connector=DenseConnect(), adj_transpose=transposed)
    xd, adj, mask = pool.preprocessing(x=x, edge_index=ei, edge_weight=ew, batch=batch)
    wx, wa = torch.randn_like(xd), torch.randn_like(adj)
    ((xd * wx).sum() + (adj * wa).sum()).backward()
    gx, gw = x.grad.clone(), ew.grad.clone()
    x.grad = None
    ew.grad = None
    adj2 = to_dense_adj(ei, batch, ew, None, None, transposed=transposed)
    xd2, mask2 = to_dense_batch(x, batch, None, None)
    ((xd2 * wx).sum() + (adj2 * wa).sum()).backward()
    assert torch.equal(xd, xd2) and torch.equal(mask, mask2)
    torch.testing.assert_close(adj, adj2, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(gx, x.grad, rtol=0, atol=0)
    torch.testing.assert_close(gw, ew.grad, rtol=1e-6, atol=1e-6)


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


@pytest.mark.parametrize("sizes,deg,weights", [
    ([39] * 64, 3.7, "rand"), ([5, 1, 0, 17, 300, 2, 1024, 64, 1, 1], 4.0, "rand"), ([620, 7, 1000], 12.0, "ties"),
    ([30] * 200, 2.0, None), ([1, 1, 1], 1.0, "rand"), ([900], 30.0, "rand"),
])
def test_graclus_per_graph_rounds_equal_the_device_wide_rounds(dev, sizes, deg, weights):
    """tgp_graclus_match_graphs (one workgroup per graph, every round in one launch) gives the very labels the
    device-wide propose / match rounds give (graphs do not interact; same edge key)."""
    from tgp import kernels
    ei, ew, batch, ptr, n = _graph_batch(sizes, deg, 11, dev, weights)
    want = kernels.graclus_match(ei, ew, n)
    got = kernels.graclus_match(ei, ew, n, graph_ptr=ptr, max_graph_nodes=max(sizes))
    assert torch.equal(got, want)
    # maximal matching, in range, symmetric (the contract test_native_graclus_matching_contract states)
    lab = got.cpu()
    pair_free = torch.bincount(lab, minlength=n)[lab] == 1
    r, c = ei.cpu()
    assert not bool((pair_free[r] & pair_free[c] & (r != c)).any())


def test_graclus_per_graph_rounds_refuse_what_they_cannot_do(dev, monkeypatch):
    """An entry that leaves its graph, or a graph longer than the declared bound, raises the status word and the
    device-wide rounds run from a fresh start: same labels as without the graph offsets."""
    from tgp import kernels, _native as N
    ei, ew, batch, ptr, n = _graph_batch([40, 50, 60], 4.0, 5, dev)
    cross = torch.tensor([[3, 95], [95, 3]], device=dev)
    ei2 = torch.cat([ei, cross], 1)
    ew2 = torch.cat([ew, torch.tensor([9.0, 9.0], device=dev)])
    want = kernels.graclus_match(ei2, ew2, n)
    got = kernels.graclus_match(ei2, ew2, n, graph_ptr=ptr, max_graph_nodes=60)
    assert torch.equal(got, want) and int(got[95]) == 3
    # a wrong (too small) declared bound is caught by the kernel itself
    big = N.lib().tgp_graclus_match_max_graph_nodes()
    ei3, ew3, _, ptr3, n3 = _graph_batch([big + 1, 10], 3.0, 6, dev)
    want = kernels.graclus_match(ei3, ew3, n3)
    got = kernels.graclus_match(ei3, ew3, n3, graph_ptr=ptr3, max_graph_nodes=10)
    assert torch.equal(got, want)
    # and a batch with such a graph never reaches the per-graph entry when the bound is honest
    called = []
    real = N.lib().tgp_graclus_match_graphs
    monkeypatch.setattr(N.lib(), "tgp_graclus_match_graphs", lambda *a: called.append(1) or real(*a))
    kernels.graclus_match(ei3, ew3, n3, graph_ptr=ptr3, max_graph_nodes=big + 1)
    assert not called


def test_graclus_pooler_with_a_batch_vector_takes_the_per_graph_rounds(dev, monkeypatch):
    from tgp import _native as N
    from tgp.poolers import get_pooler
    ei, ew, batch, ptr, n = _graph_batch([39] * 32, 3.7, 3, dev)
    x = torch.randn(n, 8, device=dev)
    pooler = get_pooler("graclus").to(dev)
    called = []
    real = N.lib().tgp_graclus_match_graphs
    real_fused = N.lib().tgp_graclus_match_graphs_fused  # r4: graphs of at most 64 nodes take the one-launch selector
    monkeypatch.setattr(N.lib(), "tgp_graclus_match_graphs", lambda *a: called.append(1) or real(*a))
    monkeypatch.setattr(N.lib(), "tgp_graclus_match_graphs_fused", lambda *a: called.append(2) or real_fused(*a))
    out_b = pooler(x=x, adj=ei, edge_weight=ew, batch=batch)
    from tgp import kernels as _k
    assert called == ([2] if _k._GRACLUS_FUSED else [1])  # (TGP_GRACLUS_FUSED=0: the staged per-graph route)
    so_plain = pooler.select(edge_index=ei, edge_weight=ew, num_nodes=n)
    assert torch.equal(out_b.so.cluster_index, so_plain.cluster_index)
    # an unsorted batch vector keeps the device-wide rounds
    called.clear()
    perm = torch.randperm(n, device=dev)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(n, device=dev)
    pooler(x=x[perm], adj=inv[ei], edge_weight=ew, batch=batch[perm])
    assert not called


@pytest.mark.parametrize("n", [0, 1, 31, 1024, 1025, 70_001, 2_100_000])
def test_graclus_relabel_kernel_equals_unique_inverse(dev, n):
    """tgp_graclus_relabel_i64 = torch.unique(label, return_inverse=True) (graclus_select.py:68) for matching labels
    (label[i] = min(i, partner))."""
    from tgp import _native as N
    g = torch.Generator().manual_seed(n)
    perm = torch.randperm(n, generator=g)
    label = torch.arange(n)
    m = (n // 3) * 2
    a, b = perm[:m:2], perm[1:m:2]
    lo = torch.minimum(a, b)
    label[a] = lo
    label[b] = lo
    ids, inverse = torch.unique(label, sorted=True, return_inverse=True)
    label = label.to(dev)
    L = N.lib()
    index = torch.full((2, n), -1, dtype=torch.int64, device=dev)
    k = torch.full((1,), -1, dtype=torch.int64, device=dev)
    ws = N.workspace(L.tgp_graclus_relabel_workspace_bytes(n), dev)
    N.check(L.tgp_graclus_relabel_i64(N.ptr(label), n, N.ptr(ws), ws.numel(), N.ptr(index), N.ptr(k), None, None,
                                      None, N.stream_ptr(dev)), "relabel")
    assert int(k) == ids.numel()
    assert torch.equal(index[0].cpu(), torch.arange(n)) and torch.equal(index[1].cpu(), inverse)
    # the same call with the supernode -> members index of the matching: equal to what the general builder derives
    from tgp import kernels
    row_ptr = torch.full((n + 1,), -7, dtype=torch.int32, device=dev)
    perm = torch.full((max(n, 1),), -7, dtype=torch.int32, device=dev)
    index2 = torch.empty_like(index)
    N.check(L.tgp_graclus_relabel_i64(N.ptr(label), n, N.ptr(ws), ws.numel(), N.ptr(index2), N.ptr(k), N.ptr(row_ptr),
                                      N.ptr(perm), None, N.stream_ptr(dev)), "relabel")
    assert torch.equal(index2, index)
    if n:
        want = kernels.build_assign_index(index[1], int(k))
        assert torch.equal(row_ptr[:int(k) + 1], want.row_ptr) and torch.equal(perm[:n], want.perm)


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


@pytest.mark.parametrize("n,m,weights", [(200_000, 1_000_000, "unit"), (200_000, 600_000, "rand"), (5000, 20_000, "unit"),
                                         (70_000, 70_000, "rand")])
def test_graclus_tail_rounds_equal_the_device_wide_rounds(dev, n, m, weights, monkeypatch):
    """tgp_graclus_match_tail (the last rounds over a list of the free nodes, one workgroup) gives the labels the
    device-wide rounds give; it is actually taken on these graphs."""
    from tgp import kernels, _native as N
    ei = _undirected(n, m, n + m).to(dev)
    if weights == "rand":
        lo, hi = torch.minimum(ei[0], ei[1]), torch.maximum(ei[0], ei[1])
        ew = (torch.sin((lo * 7919 + hi * 104729).double()) * 0.5 + 0.6).float()
    else:
        ew = torch.ones(ei.size(1), device=dev)
    want = kernels.graclus_match(ei, ew, n, max_rounds=1 << 20)      # (a bounded loop never takes the tail)
    seen = []
    real = N.lib().tgp_graclus_match_tail
    monkeypatch.setattr(N.lib(), "tgp_graclus_match_tail", lambda *a: seen.append(1) or real(*a))
    got = kernels.graclus_match(ei, ew, n)
    assert seen and torch.equal(got, want)


# ------------------------------------------------------------------------------ batch facts in one read-back
def _batch_info_torch(batch):
    sizes = torch.bincount(batch)
    return dict(num_graphs=sizes.numel(), sizes=sizes.tolist(), is_sorted=bool((batch[1:] >= batch[:-1]).all()),
                max_nodes=int(sizes.max()), distinct=int((sizes > 0).sum()))


@pytest.mark.parametrize("case", ["sorted", "unsorted", "gaps", "one", "single_node", "long_runs", "big"])
def test_batch_facts_kernel_equals_the_torch_ops(dev, case):
    """utils.ops.batch_info through tgp_batch_facts_i64 (one read-back) = bincount / comparison / max on the same vector;
    ids beyond N or negative ids decline to the torch route."""
    from tgp.utils.ops import batch_info
    g = torch.Generator().manual_seed(3)
    if case == "sorted":
        batch = torch.repeat_interleave(torch.arange(300), torch.randint(1, 70, (300,), generator=g))
    elif case == "unsorted":
        batch = torch.randint(0, 50, (5000,), generator=g)
    elif case == "gaps":      # empty graphs in the middle and a sorted vector
        batch = torch.sort(torch.randint(0, 40, (200,), generator=g) * 3).values
    elif case == "one":
        batch = torch.zeros(1000, dtype=torch.long)
    elif case == "single_node":
        batch = torch.tensor([0])
    elif case == "long_runs":  # runs longer than a wave and crossing workgroup boundaries
        batch = torch.repeat_interleave(torch.arange(7), torch.tensor([1, 63, 64, 65, 700, 256, 1]))
    else:
        batch = torch.repeat_interleave(torch.arange(20000), torch.randint(20, 61, (20000,), generator=g))
    want = _batch_info_torch(batch)
    info = batch_info(batch.to(dev))
    assert info.num_graphs == want["num_graphs"] and info.is_sorted == want["is_sorted"]
    assert info.max_nodes == want["max_nodes"] and info.distinct == want["distinct"]
    assert info.sizes.tolist() == want["sizes"] and info.sizes_host == want["sizes"]
    assert info.ptr.tolist() == [0] + torch.cumsum(torch.tensor(want["sizes"]), 0).tolist()


def test_batch_facts_kernel_declines_ids_it_cannot_count(dev):
    from tgp.utils.ops import batch_info
    info = batch_info(torch.tensor([0, 7], device=dev))     # more graph ids than nodes: legal, torch route
    assert info.num_graphs == 8 and info.sizes.tolist() == [1, 0, 0, 0, 0, 0, 0, 1] and info.distinct == 2
    with pytest.raises(RuntimeError):
        batch_info(torch.tensor([0, -1], device=dev))       # (bincount's own error, as before)


def test_graclus_per_graph_route_with_an_unchecked_unsorted_edge_list(dev):
    """The one-launch route goes on before the row order of a NEW edge_index object is known (the check rides on the
    offsets kernel); a list that turns out not to be sorted is redone on the general route: same labels as ever, and the
    memo remembers the answer for the object."""
    from tgp import kernels
    ei, ew, batch, ptr, n = _graph_batch([39] * 64, 3.7, 21, dev)
    perm = torch.randperm(ei.size(1), device=dev)
    ei_u, ew_u = ei[:, perm].contiguous(), ew[perm].contiguous()
    want = kernels.graclus_match(ei, ew, n)
    assert kernels._rows_sorted_memo(ei_u) is None
    got = kernels.graclus_match(ei_u, ew_u, n, graph_ptr=ptr, max_graph_nodes=39)
    assert torch.equal(got, want) and kernels._rows_sorted_memo(ei_u) is False
    fresh = ei.clone()
    assert kernels._rows_sorted_memo(fresh) is None
    (index, k, _, ones), row_ptr = kernels.graclus_match(fresh, ew, n, graph_ptr=ptr, max_graph_nodes=39,
                                                         relabel=True, return_row_ptr=True)
    assert torch.equal(ones, torch.ones(n, device=dev))
    # r4: graphs of at most 64 nodes take the one-launch selector, which validates the row order itself and builds no
    # offsets (row_ptr None); the staged per-graph route (TGP_GRACLUS_FUSED=0) returns them
    assert kernels._rows_sorted_memo(fresh) is True and (row_ptr is None or row_ptr.numel() == n + 1)
    ids, inverse = torch.unique(want, return_inverse=True)
    assert k == ids.numel() and torch.equal(index[1], inverse)


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


def test_graclus_reduce_backward_with_the_identity_lift_index(dev):
    """GraclusSelect's SelectOutput needs no transposed index for the backward of Reduce (node i owns assignment i):
    dX = S dX' equals the gather it is, and Lift gives the same rows."""
    from tgp.lift import lift_index_of
    from tgp.poolers import get_pooler
    ei, ew, batch, ptr, n = _graph_batch([33] * 20, 3.5, 8, dev)
    x = torch.randn(n, 12, device=dev, requires_grad=True)
    pooler = get_pooler("graclus").to(dev)
    out = pooler(x=x, adj=ei, edge_weight=ew, batch=batch)
    idx = lift_index_of(out.so)
    assert idx.one_to_one and idx.perm is None
    up = torch.randn_like(out.x)
    out.x.backward(up)
    want = up[out.so.cluster_index] * out.so.weight.unsqueeze(1)
    assert torch.allclose(x.grad, want, rtol=1e-6, atol=1e-6)
    lifted = pooler(x=out.x.detach(), so=out.so, lifting=True)
    assert torch.allclose(lifted, out.x.detach()[out.so.cluster_index], rtol=1e-6, atol=1e-6)


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


@pytest.mark.parametrize("max_iter", [1, 5, 37])
def test_ndp_large_steps_stop_at_the_step_budget(dev, max_iter):
    """The fused LOBPCG step (tgp_ndp_large_steps) honours max_iter when the tolerance is out of reach: exactly that
    many updates, the host loop ends, the partition is still a sign split of the current iterate (or the reference's
    random fallback when its cut is below 0.5)."""
    from tgp import kernels as K
    n = 3000
    ei = _undirected(n, 12000, 5).to(dev)
    indptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    K.rowptr_from_sorted(ei[0], n, indptr)
    keep = torch.full((n,), 7, dtype=torch.uint8, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    info, state = K.ndp_partition_large(indptr, ei[1], None, 0, n, 3, keep, status, max_iter=max_iter, tol=1e-12,
                                        want_state=True)
    assert int(status) == 0 and state["steps"] == max_iter
    assert set(keep.unique().tolist()) <= {0, 1} and 0 < int(keep.sum()) < n
    assert state["random"] or int(info) == max_iter


@pytest.mark.parametrize("n,hubs,deg,lds", [(3000, 3, 900, False), (70_000, 5, 4000, True), (3000, 70, 400, False)])
def test_graclus_matching_with_hub_rows(dev, n, hubs, deg, lds):
    """Rows beyond 256 entries are scanned by whole waves taken from a list (work stealing): the matching is still valid
    and maximal, and on distinct weights it is the sequential greedy heavy-edge matching (unique), hubs next to each
    other included (70 hubs = more long rows than one wave has lanes); n >= 65536 takes the LDS-bitmap kernel."""
    from tgp import kernels as KK
    g = torch.Generator().manual_seed(n + hubs)
    a = torch.randint(0, n, (2 * n,), generator=g)
    b = torch.randint(0, n, (2 * n,), generator=g)
    h = torch.arange(hubs).repeat_interleave(deg)
    t = torch.randint(hubs, n, (hubs * deg,), generator=g)
    r, c = torch.cat([a, h]), torch.cat([b, t])
    keep = r != c
    key = torch.unique(torch.minimum(r, c)[keep] * n + torch.maximum(r, c)[keep])
    lo, hi = key // n, key % n
    und_w = torch.rand(lo.numel(), generator=g) + 0.1
    und_w = und_w + torch.arange(lo.numel()) * 1e-9       # distinct
    ei = torch.cat([torch.stack([lo, hi]), torch.stack([hi, lo])], 1)
    w = torch.cat([und_w, und_w])
    order = torch.argsort(ei[0] * n + ei[1])
    ei, w = ei[:, order].contiguous(), w[order].contiguous()
    label = KK.graclus_match(ei.to(dev), w.to(dev), n).cpu()
    idx = torch.arange(n)
    cnt = torch.bincount(label, minlength=n)
    assert cnt.max() <= 2 and torch.all(label <= idx) and torch.equal(label[label], label)
    matched = cnt[label] == 2
    assert not bool((~matched[ei[0]] & ~matched[ei[1]]).any())      # maximal
    ref = idx.clone()
    free = torch.ones(n, dtype=torch.bool)
    for k in torch.argsort(und_w, descending=True).tolist():        # sequential greedy on the undirected pairs
        x, y = int(lo[k]), int(hi[k])
        if free[x] and free[y]:
            free[x] = free[y] = False
            ref[x] = ref[y] = min(x, y)
    assert torch.equal(label, ref)
