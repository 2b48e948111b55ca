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
