"""Backward of the native Reduce / Connect operators: gradients from the HIP
autograd Functions against plain torch autograd of the same fp32 math on the same device."""
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-4, atol=1e-5)


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def test_dense_reduce_connect_gradients(dev):
    from tgp.connect import DenseConnect
    from tgp.reduce import BaseReduce
    from tgp.select import SelectOutput
    g = torch.Generator(device=dev).manual_seed(0)
    B, N, K, F = 3, 70, 12, 9
    S0 = torch.softmax(torch.randn(B, N, K, device=dev, generator=g), -1)
    A0 = torch.rand(B, N, N, device=dev, generator=g)
    X0 = torch.randn(B, N, F, device=dev, generator=g)
    wx = torch.randn(B, K, F, device=dev, generator=g)
    wa = torch.randn(B, K, K, device=dev, generator=g)

    def run(native):
        S, A, X = (t.clone().requires_grad_(True) for t in (S0, A0, X0))
        if native:
            so = SelectOutput(s=S)
            xp, _ = BaseReduce()(X, so)
            ap, _ = DenseConnect(remove_self_loops=True, degree_norm=True, adj_transpose=True)(A, so)
        else:
            xp = S.transpose(1, 2) @ X
            raw = S.transpose(1, 2) @ A @ S
            raw = raw * (1 - torch.eye(K, device=dev))
            d = torch.sqrt(raw.sum(-2, keepdim=True).clamp(min=1e-8))
            ap = (raw / d) / d.transpose(-2, -1)
        ((xp * wx).sum() + (ap * wa).sum()).backward()
        return xp.detach(), ap.detach(), S.grad, A.grad, X.grad

    got, ref = run(True), run(False)
    for a, b, name in zip(got, ref, ("x_pool", "adj_pool", "dS", "dA", "dX")):
        torch.testing.assert_close(a, b, msg=lambda m: f"{name}: {m}", **TOL)


def test_sparse_reduce_gradients(dev):
    from tgp.reduce import BaseReduce
    from tgp.select import SelectOutput
    g = torch.Generator(device=dev).manual_seed(1)
    n, k, f = 500, 120, 16
    cluster = torch.randint(0, k, (n,), device=dev, generator=g)
    cluster[:k] = torch.arange(k, device=dev)
    x0 = torch.randn(n, f, device=dev, generator=g)
    w0 = torch.rand(n, device=dev, generator=g) + 0.5
    go = torch.randn(k, f, device=dev, generator=g)

    x, w = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
    so = SelectOutput(cluster_index=cluster, num_nodes=n, num_supernodes=k, weight=w)
    xp, _ = BaseReduce()(x, so)
    (xp * go).sum().backward()

    x2, w2 = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
    ref = torch.zeros(k, f, device=dev).index_add_(0, cluster, x2 * w2.view(-1, 1))
    (ref * go).sum().backward()
    torch.testing.assert_close(xp.detach(), ref.detach(), **TOL)
    torch.testing.assert_close(x.grad, x2.grad, **TOL)
    torch.testing.assert_close(w.grad, w2.grad, **TOL)


def test_topk_pooler_trains(dev):
    """TopK's projection vector receives a gradient through the score-weighted Reduce."""
    from tgp.poolers import get_pooler
    g = torch.Generator(device=dev).manual_seed(2)
    x = torch.randn(60, 8, device=dev, generator=g)
    ei = torch.randint(0, 60, (2, 300), device=dev, generator=g)
    pool = get_pooler("topk", in_channels=8, ratio=0.5).to(dev)
    out = pool(x=x, adj=ei)
    out.x.pow(2).sum().backward()
    grad = pool.selector.weight.grad
    assert grad is not None and torch.isfinite(grad).all() and grad.abs().sum() > 0


# --------------------------------------------------------------------------- un-padded (unbatched) mode
def _ragged_batch(dev, sizes):
    sizes = torch.tensor(sizes)
    batch = torch.repeat_interleave(torch.arange(sizes.numel()), sizes).to(dev)
    ptr = torch.cat([sizes.new_zeros(1), sizes.cumsum(0)]).to(dev)
    return sizes, batch, ptr


@pytest.mark.parametrize("Kd,Nc", [(12, 8), (13, 7), (128, 64)])
def test_segment_gemm_nn_forward(dev, Kd, Nc):
    """C[rows of b] = A[rows of b] M_b against a per-graph loop (lift/base_lift.py:205-215)."""
    from tgp import kernels as KK
    g = torch.Generator(device=dev).manual_seed(Kd)
    sizes, batch, ptr = _ragged_batch(dev, [70, 1, 300, 64, 129])
    a = torch.randn(int(sizes.sum()), Kd, device=dev, generator=g)
    m = torch.randn(sizes.numel(), Kd, Nc, device=dev, generator=g)
    got = KK.segment_gemm_nn(a, m, ptr, int(sizes.max()))
    ref = torch.cat([(a[int(ptr[b]):int(ptr[b + 1])].double() @ m[b].double()).float() for b in range(sizes.numel())])
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-4)


def test_unbatched_reduce_connect_gradients(dev):
    """S [N,K] + batch vector + sparse (non-symmetric, weighted, duplicated) A: X' = S_b^T X_b,
    A' = S_b^T A_b S_b (reduce/base_reduce.py:170-182, connect/dense_conn.py:140-208) and their gradients
    w.r.t. S, X and the edge weights against torch autograd on the densified problem."""
    from tgp.connect import DenseConnect
    from tgp.reduce import BaseReduce
    from tgp.select import SelectOutput
    g = torch.Generator(device=dev).manual_seed(3)
    sizes, batch, ptr = _ragged_batch(dev, [40, 7, 65, 33])
    n, K, F, B = int(sizes.sum()), 6, 5, sizes.numel()
    rows, cols = [], []
    for b in range(B):
        lo, nb = int(ptr[b]), int(sizes[b])
        e = 4 * nb
        rows.append(lo + torch.randint(0, nb, (e,), device=dev, generator=g))
        cols.append(lo + torch.randint(0, nb, (e,), device=dev, generator=g))
    ei = torch.stack([torch.cat(rows), torch.cat(cols)])
    ei = torch.cat([ei, ei[:, :50]], 1)  # duplicates are summed
    ei = ei[:, torch.randperm(ei.size(1), device=dev, generator=g)]
    w0 = torch.rand(ei.size(1), device=dev, generator=g) + 0.1
    S0 = torch.softmax(torch.randn(n, K, device=dev, generator=g), -1)
    X0 = torch.randn(n, F, device=dev, generator=g)
    wx = torch.randn(B * K, F, device=dev, generator=g)
    wa = torch.randn(B, K, K, device=dev, generator=g)

    def run(native):
        S, X, w = (t.clone().requires_grad_(True) for t in (S0, X0, w0))
        if native:
            so = SelectOutput(s=S, batch=batch)
            xp, _ = BaseReduce()(X, so, batch=batch)
            ap, _ = DenseConnect(remove_self_loops=True, degree_norm=True, adj_transpose=False)(
                ei, so, edge_weight=w, batch=batch)
        else:
            A = torch.zeros(n, n, device=dev).index_put((ei[0], ei[1]), w, accumulate=True)
            xs, as_ = [], []
            for b in range(B):
                lo, hi = int(ptr[b]), int(ptr[b + 1])
                xs.append(S[lo:hi].t() @ X[lo:hi])
                as_.append(S[lo:hi].t() @ A[lo:hi, lo:hi] @ S[lo:hi])
            xp, raw = torch.cat(xs), torch.stack(as_)
            raw = raw * (1 - torch.eye(K, device=dev))
            d = torch.sqrt(raw.sum(-1, keepdim=True).clamp(min=1e-8))
            ap = (raw / d) / d.transpose(-2, -1)
        ((xp * wx).sum() + (ap * wa).sum()).backward()
        return xp.detach(), ap.detach(), S.grad, X.grad, w.grad

    got, ref = run(True), run(False)
    for a, b, name in zip(got, ref, ("x_pool", "adj_pool", "dS", "dX", "dw")):
        torch.testing.assert_close(a, b, msg=lambda m: f"{name}: {m}", **TOL)


def test_unbatched_sparse_output_gradients(dev):
    """DenseConnect(sparse_output=True) on unbatched inputs: the block-diagonal edge weights carry gradients
    back to S through the export and the sparse normalisations (connect/dense_conn.py:329-354)."""
    from tgp.connect import DenseConnect
    from tgp.select import SelectOutput
    g = torch.Generator(device=dev).manual_seed(4)
    sizes, batch, ptr = _ragged_batch(dev, [20, 31])
    n, K, B = int(sizes.sum()), 4, 2
    A = torch.zeros(n, n, device=dev)
    for b in range(B):
        lo, hi = int(ptr[b]), int(ptr[b + 1])
        blk = (torch.rand(hi - lo, hi - lo, device=dev, generator=g) < 0.2).float()
        A[lo:hi, lo:hi] = torch.triu(blk, 1) + torch.triu(blk, 1).t()
    ei = A.nonzero().t().contiguous()
    S0 = torch.softmax(torch.randn(n, K, device=dev, generator=g), -1)
    bp = torch.arange(B, device=dev).repeat_interleave(K)

    S = S0.clone().requires_grad_(True)
    out_ei, out_w = DenseConnect(remove_self_loops=True, degree_norm=True, edge_weight_norm=True, sparse_output=True)(
        ei, SelectOutput(s=S, batch=batch), batch=batch, batch_pooled=bp)
    coef = torch.randn(out_w.numel(), device=dev, generator=g)
    (out_w * coef).sum().backward()

    S2 = S0.clone().requires_grad_(True)
    raw = torch.stack([S2[int(ptr[b]):int(ptr[b + 1])].t() @ A[int(ptr[b]):int(ptr[b + 1]), int(ptr[b]):int(ptr[b + 1])]
                       @ S2[int(ptr[b]):int(ptr[b + 1])] for b in range(B)])
    raw = raw * (1 - torch.eye(K, device=dev))
    d = raw.sum(-1, keepdim=True).clamp(min=1e-8).pow(-0.5)  # row sums of the block-diagonal list
    nrm = raw * d * d.transpose(-2, -1)
    nrm = nrm / nrm.abs().amax(dim=(1, 2), keepdim=True)
    b_i, r_i, c_i = out_ei[0] // K, out_ei[0] % K, out_ei[1] % K
    ref_w = nrm[b_i, r_i, c_i]
    (ref_w * coef).sum().backward()
    torch.testing.assert_close(out_w.detach(), ref_w.detach(), **TOL)
    torch.testing.assert_close(S.grad, S2.grad, **TOL)


def test_lift_gradients(dev):
    """BaseLift in its three layouts (lift/base_lift.py:102-111, 138-247): sparse S, dense batched S, dense
    [N,K] S with a multi-graph batch vector; gradients w.r.t. x_pool (and the assignment where it is dense)."""
    from tgp.lift import BaseLift
    from tgp.select import SelectOutput
    g = torch.Generator(device=dev).manual_seed(5)
    # sparse
    n, k, f = 300, 70, 6
    cluster = torch.randint(0, k, (n,), device=dev, generator=g)
    cluster[:k] = torch.arange(k, device=dev)
    w = torch.rand(n, device=dev, generator=g) + 0.5
    so = SelectOutput(cluster_index=cluster, num_nodes=n, num_supernodes=k, weight=w)
    xp = torch.randn(k, f, device=dev, generator=g, requires_grad=True)
    go = torch.randn(n, f, device=dev, generator=g)
    out = BaseLift(matrix_op="transpose")(xp, so)
    (out * go).sum().backward()
    xp2 = xp.detach().clone().requires_grad_(True)
    ref = xp2[cluster] * w.view(-1, 1)
    (ref * go).sum().backward()
    torch.testing.assert_close(out.detach(), ref.detach(), **TOL)
    torch.testing.assert_close(xp.grad, xp2.grad, **TOL)
    # dense batched
    B, N, K = 3, 50, 7
    S = torch.softmax(torch.randn(B, N, K, device=dev, generator=g), -1).requires_grad_(True)
    xp = torch.randn(B, K, f, device=dev, generator=g, requires_grad=True)
    go = torch.randn(B, N, f, device=dev, generator=g)
    out = BaseLift(matrix_op="transpose")(xp, SelectOutput(s=S))
    (out * go).sum().backward()
    S2, xp2 = S.detach().clone().requires_grad_(True), xp.detach().clone().requires_grad_(True)
    ((S2 @ xp2) * go).sum().backward()
    torch.testing.assert_close(out.detach(), (S2 @ xp2).detach(), **TOL)
    torch.testing.assert_close(S.grad, S2.grad, **TOL)
    torch.testing.assert_close(xp.grad, xp2.grad, **TOL)
    # dense [N,K] with a batch vector
    sizes, batch, ptr = _ragged_batch(dev, [40, 3, 66])
    n, Bn = int(sizes.sum()), 3
    S = torch.softmax(torch.randn(n, K, device=dev, generator=g), -1).requires_grad_(True)
    xp = torch.randn(Bn * K, f, device=dev, generator=g, requires_grad=True)
    go = torch.randn(n, f, device=dev, generator=g)
    out = BaseLift(matrix_op="transpose")(xp, SelectOutput(s=S, batch=batch), batch=batch)
    (out * go).sum().backward()
    S2, xp2 = S.detach().clone().requires_grad_(True), xp.detach().clone().requires_grad_(True)
    ref = torch.cat([S2[int(ptr[b]):int(ptr[b + 1])] @ xp2.view(Bn, K, f)[b] for b in range(Bn)])
    (ref * go).sum().backward()
    torch.testing.assert_close(out.detach(), ref.detach(), **TOL)
    torch.testing.assert_close(S.grad, S2.grad, **TOL)
    torch.testing.assert_close(xp.grad, xp2.grad, **TOL)


@pytest.mark.parametrize("alias", ["diff", "mincut"])
def test_unbatched_pooler_parameter_gradients_match_batched(dev, alias):
    """The `_u` poolers train like the batched ones: same parameters, same graphs => same parameter gradients
    through Select -> Reduce -> Connect -> losses (reference pin 11 compares the forward values only)."""
    from tgp.poolers import get_pooler
    g = torch.Generator().manual_seed(6)
    sizes = [30, 45, 38]
    xs, eis, bs, off = [], [], [], 0
    for gi, n in enumerate(sizes):
        a = torch.triu(torch.rand(n, n, generator=g) < 0.15, 1)
        a = a | a.t()
        eis.append(a.nonzero().t() + off)
        xs.append(torch.randn(n, 16, generator=g))
        bs.append(torch.full((n,), gi))
        off += n
    x, ei, b = torch.cat(xs).to(dev), torch.cat(eis, 1).to(dev), torch.cat(bs).to(dev)
    pb = get_pooler(alias, in_channels=16, k=8).to(dev)
    pu = get_pooler(alias + "_u", in_channels=16, k=8).to(dev)
    pu.load_state_dict(pb.state_dict())
    grads = []
    for p in (pb, pu):
        out = p(x=x, adj=ei, batch=b)
        adj = out.edge_index
        loss = out.x.pow(2).sum() + adj.pow(2).sum() + sum(out.loss.values())
        loss.backward()
        grads.append([q.grad.clone() for q in p.parameters()])
    for ga, gb in zip(*grads):
        torch.testing.assert_close(ga, gb, rtol=1e-3, atol=1e-4)


def test_to_dense_batch_and_linear_gradients(dev):
    """Sparse -> padded-dense preprocessing (src.py:448-450) and the selector's Linear layer
    (select/mlp_select.py:67) under autograd: native forward, gather / split-range backward."""
    from tgp import functions as Fn
    from tgp.src import to_dense_batch
    g = torch.Generator(device=dev).manual_seed(7)
    sizes, batch, ptr = _ragged_batch(dev, [5, 1, 9, 4])
    n, f = int(sizes.sum()), 6
    x0 = torch.randn(n, f, device=dev, generator=g)
    for max_nodes in (None, 6):  # 6 truncates the 9-node graph: dropped nodes get zero gradient
        x = x0.clone().requires_grad_(True)
        out, mask = to_dense_batch(x, batch, max_num_nodes=max_nodes)
        go = torch.randn(out.shape, device=dev, generator=g)
        (out * go).sum().backward()
        nmax = out.size(1)
        local = torch.arange(n, device=dev) - ptr[batch]
        keep = local < nmax
        ref = torch.zeros(n, f, device=dev)
        ref[keep] = go[batch[keep], local[keep]]
        assert out.grad_fn is not None and not mask.requires_grad
        torch.testing.assert_close(x.grad, ref, **TOL)
        assert torch.equal(out[batch[keep], local[keep]], x0[keep])
    # linear: [B,N,F] input, weight [K,F], bias
    X0 = torch.randn(3, 700, 10, device=dev, generator=g)
    W0 = torch.randn(12, 10, device=dev, generator=g)
    b0 = torch.randn(12, device=dev, generator=g)
    go = torch.randn(3, 700, 12, device=dev, generator=g)
    res = []
    for native in (True, False):
        X, W, b = (t.clone().requires_grad_(True) for t in (X0, W0, b0))
        y = Fn.linear(X, W, b) if native else torch.nn.functional.linear(X, W, b)
        (y * go).sum().backward()
        res.append((y.detach(), X.grad, W.grad, b.grad))
    for a, r in zip(*res):
        torch.testing.assert_close(a, r, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("F", [1, 3, 4, 64, 128, 130, 300, 2100])
def test_row_dot_and_gradients(dev, F):
    """TopkSelect scoring x.w (select/topk_select.py:176) and its two gradients against torch."""
    from tgp import functions as Fn
    g = torch.Generator(device=dev).manual_seed(F)
    n = 3001
    x0 = torch.randn(n, F, device=dev, generator=g)
    w0 = torch.randn(1, F, device=dev, generator=g)
    go = torch.randn(n, device=dev, generator=g)
    x, w = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
    out = Fn.row_dot(x, w)
    (out * go).sum().backward()
    x2, w2 = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
    ref = (x2 * w2).sum(-1)
    (ref * go).sum().backward()
    torch.testing.assert_close(out.detach(), ref.detach(), rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(x.grad, x2.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(w.grad, w2.grad, rtol=1e-4, atol=1e-3)
    xs = x0[:, : max(1, F // 2)]  # strided rows
    torch.testing.assert_close(Fn.row_dot(xs, w0[:, : xs.size(1)]), (xs * w0[:, : xs.size(1)]).sum(-1), rtol=1e-5,
                               atol=1e-4)


@pytest.mark.parametrize("rsl,dn,at", [(True, True, True), (True, True, False), (False, True, True), (True, False, False),
                                       (False, True, False)])
def test_postprocess_dense_closed_form_backward(dev, rsl, dn, at):
    """A8 (utils/ops.py:282-335) under autograd: native forward + closed-form backward against torch autograd of the
    elementwise form, including a graph whose degree sums sit below eps (clamped: zero gradient through the sum)."""
    from tgp.utils.ops import _postprocess_dense_autograd, postprocess_adj_pool_dense
    g = torch.Generator(device=dev).manual_seed(3)
    a0 = torch.rand(5, 9, 9, device=dev, generator=g) + 0.05
    a0[2] = 0.0                      # empty graph: every degree is clamped at eps
    a0[3, :, 4] = 0.0
    a0[3, 4, :] = 0.0                # an isolated supernode
    go = torch.randn(5, 9, 9, device=dev, generator=g)
    a = a0.clone().requires_grad_(True)
    out = postprocess_adj_pool_dense(a, rsl, dn, at, False)
    (out * go).sum().backward()
    b = a0.clone().requires_grad_(True)
    ref = _postprocess_dense_autograd(b, rsl, dn, at, False)
    (ref * go).sum().backward()
    torch.testing.assert_close(out.detach(), ref.detach(), **TOL)
    torch.testing.assert_close(a.grad, b.grad, rtol=1e-4, atol=1e-4)


def test_orthogonality_loss_closed_form_backward(dev):
    from tgp.utils.losses import orthogonality_loss
    g = torch.Generator(device=dev).manual_seed(4)
    S0 = torch.softmax(torch.randn(6, 40, 7, device=dev, generator=g), -1)
    S = S0.clone().requires_grad_(True)
    loss = orthogonality_loss(S)
    loss.backward()
    S2 = S0.clone().requires_grad_(True)
    sts = S2.transpose(1, 2) @ S2
    sts = sts / torch.norm(sts, dim=(-2, -1), keepdim=True)
    ref = torch.norm(sts - torch.eye(7, device=dev) / 7 ** 0.5, dim=(-2, -1)).mean()
    ref.backward()
    torch.testing.assert_close(loss.detach(), ref.detach(), **TOL)
    torch.testing.assert_close(S.grad, S2.grad, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("transposed_view", [False, True])
def test_dense_connect_gradients_large_graph_path(dev, transposed_view):
    """DenseConnect under autograd on graphs too large for the one-workgroup-per-graph kernels (N > 512 or K > 64):
    U = A S is a tensor shared between the forward, its backward and the link loss (functions.ASProducts); A may be
    the transposed view the dense preprocessing hands over (src.py:442-443)."""
    from tgp.connect import DenseConnect
    from tgp.select import SelectOutput
    from tgp.utils.losses import link_pred_loss
    g = torch.Generator(device=dev).manual_seed(8)
    B, N, K = 2, 600, 70
    S0 = torch.softmax(torch.randn(B, N, K, device=dev, generator=g), -1)
    A0 = (torch.rand(B, N, N, device=dev, generator=g) < 0.05).float() * torch.rand(B, N, N, device=dev, generator=g)
    wa = torch.randn(B, K, K, device=dev, generator=g)

    def run(native):
        S = S0.clone().requires_grad_(True)
        A = A0.transpose(1, 2).contiguous().transpose(1, 2) if transposed_view else A0.clone()
        assert A.is_contiguous() != transposed_view
        if native:
            ap, _ = DenseConnect(remove_self_loops=True, degree_norm=True, adj_transpose=True)(A, SelectOutput(s=S))
            ll = link_pred_loss(S, A, normalize_loss=False)
        else:
            raw = S.transpose(1, 2) @ A @ S
            raw = raw * (1 - torch.eye(K, device=dev))
            d = torch.sqrt(raw.sum(-2, keepdim=True).clamp(min=1e-8))
            ap = (raw / d) / d.transpose(-2, -1)
            ll = torch.norm(A - S @ S.transpose(1, 2), p=2)
        ((ap * wa).sum() + ll).backward()
        return ap.detach(), ll.detach(), S.grad

    got, ref = run(True), run(False)
    torch.testing.assert_close(got[0], ref[0], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(got[1], ref[1], rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(got[2], ref[2], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("shape", [(2048, 60, 32, 20), (64, 1024, 64, 128), (700, 50, 7, 3)])
def test_linear_weight_gradient_tall_skinny(dev, shape):
    """dW = dY^T X of the selector's Linear over many node rows (select/mlp_select.py:67): the slab-wise path for
    narrow layers and the split-range segment product otherwise, against torch."""
    from tgp import functions as Fn
    B, N, F, K = shape
    g = torch.Generator(device=dev).manual_seed(B + K)
    X0 = torch.randn(B, N, F, device=dev, generator=g)
    W0 = torch.randn(K, F, device=dev, generator=g) / F ** 0.5
    b0 = torch.randn(K, device=dev, generator=g)
    go = torch.randn(B, N, K, device=dev, generator=g)
    res = []
    for native in (True, False):
        X, W, b = (t.clone().requires_grad_(True) for t in (X0, W0, b0))
        y = Fn.linear(X, W, b) if native else torch.nn.functional.linear(X, W, b)
        (y * go).sum().backward()
        res.append((X.grad, W.grad, b.grad))
    scale = float(res[1][1].abs().max())
    torch.testing.assert_close(res[0][0], res[1][0], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(res[0][1], res[1][1], rtol=1e-4, atol=1e-5 * max(scale, 1.0) * 10)
    torch.testing.assert_close(res[0][2], res[1][2], rtol=1e-4, atol=1e-2)
