"""Fused auxiliary losses of the dense poolers: forward against the CPU oracle
(oracle/tgp_oracle.py restates utils/losses.py), backward against torch autograd of the reference formulas
on the same device.  fp32 tolerance: rtol = atol = 1e-5 forward (north_star), 1e-4 on gradients."""
import math

import pytest
import torch

import tgp_oracle as O

pytestmark = pytest.mark.gpu
FWD = dict(rtol=1e-5, atol=1e-5)
BWD = dict(rtol=1e-4, atol=1e-5)
SHAPES = [(1, 5, 3), (3, 50, 7), (2, 64, 8), (4, 200, 20), (2, 333, 17), (2, 1024, 128), (64, 60, 20)]


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _inputs(dev, B, N, K, seed=0, weighted=True):
    g = torch.Generator().manual_seed(seed)
    S = torch.softmax(torch.randn(B, N, K, generator=g), -1)
    A = (torch.rand(B, N, N, generator=g) < 0.2).float()
    if weighted:
        A = A * torch.rand(B, N, N, generator=g)
    return S, A


@pytest.mark.parametrize("B,N,K", SHAPES)
def test_link_loss_matches_oracle(dev, B, N, K):
    from tgp.utils.losses import link_pred_loss
    S, A = _inputs(dev, B, N, K, seed=N)
    for normalize in (False, True):
        want = O.link_pred_loss(S.double(), A.double(), normalize).float()
        got = link_pred_loss(S.to(dev), A.to(dev), normalize_loss=normalize).cpu()
        torch.testing.assert_close(got, want, **FWD)


def test_link_loss_padded_rows_and_strided(dev):
    """Padded node rows (S = 0, A = 0) contribute nothing; per-graph squares add up."""
    from tgp import kernels
    S, A = _inputs(dev, 3, 40, 6, seed=5)
    S[:, 30:] = 0
    A[:, 30:, :] = 0
    A[:, :, 30:] = 0
    sq = kernels.link_loss_sq(S.to(dev), A.to(dev)).cpu()
    want = ((A[:, :30, :30] - S[:, :30] @ S[:, :30].transpose(1, 2)) ** 2).sum((1, 2))
    torch.testing.assert_close(sq, want, **FWD)
    # exact fit => exactly representable zero residual stays tiny, gradient finite
    from tgp.utils.losses import link_pred_loss
    S1 = torch.eye(4).repeat(2, 1, 1).to(dev).requires_grad_(True)
    A1 = torch.eye(4).repeat(2, 1, 1).to(dev)
    loss = link_pred_loss(S1, A1, normalize_loss=False)
    loss.backward()
    assert float(loss.detach()) == 0.0 and torch.isfinite(S1.grad).all()


@pytest.mark.parametrize("B,N,K", [(3, 50, 7), (2, 128, 16), (2, 333, 17)])
def test_link_loss_gradients(dev, B, N, K):
    from tgp.utils.losses import link_pred_loss
    S0, A0 = _inputs(dev, B, N, K, seed=1)

    def run(native):
        S = S0.to(dev).requires_grad_(True)
        A = A0.to(dev).requires_grad_(True)
        if native:
            loss = link_pred_loss(S, A, normalize_loss=False)
        else:
            loss = torch.norm(A - S @ S.transpose(1, 2), p=2)
        loss.backward()
        return loss.detach(), S.grad, A.grad

    for a, b in zip(run(True), run(False)):
        torch.testing.assert_close(a, b, **BWD)


@pytest.mark.parametrize("B,N,K", SHAPES)
def test_entropy_and_cut_and_ortho_match_oracle(dev, B, N, K):
    from tgp.utils.losses import entropy_loss, mincut_loss, orthogonality_loss
    S, A = _inputs(dev, B, N, K, seed=N + 1)
    Sd, Ad = S.to(dev), A.to(dev)
    n_nodes = B * N - 3
    torch.testing.assert_close(entropy_loss(Sd, n_nodes).cpu(), O.entropy_loss(S.double(), n_nodes).float(), **FWD)
    raw = S.transpose(1, 2) @ A @ S
    torch.testing.assert_close(mincut_loss(Ad, Sd, raw.to(dev)).cpu(), O.mincut_loss(A, S, raw), **FWD)
    torch.testing.assert_close(orthogonality_loss(Sd).cpu(), O.orthogonality_loss(S), **FWD)
    with pytest.raises(ValueError):
        orthogonality_loss(Sd, batch_reduction="max")


def test_entropy_cut_ortho_gradients(dev):
    from tgp.utils.losses import entropy_loss, mincut_loss, orthogonality_loss
    B, N, K = 3, 70, 9
    S0, A0 = _inputs(dev, B, N, K, seed=2)
    eps = 1e-8

    def run(native):
        S = S0.to(dev).requires_grad_(True)
        A = A0.to(dev).requires_grad_(True)
        raw = S.transpose(1, 2) @ A @ S
        if native:
            loss = entropy_loss(S, B * N) + 3.0 * mincut_loss(A, S, raw) + 2.0 * orthogonality_loss(S, "sum")
        else:
            ent = (-(S * torch.log(S + eps)).sum(-1)).sum() / (B * N)
            num = torch.einsum("ijj->i", raw)
            den = torch.einsum("bnk,bn,bnk->b", S, A.sum(-1), S)
            cut = (-(num / (den + eps))).mean()
            sts = S.transpose(1, 2) @ S
            sts = sts / torch.norm(sts, dim=(-2, -1), keepdim=True)
            ortho = torch.norm(sts - torch.eye(K, device=dev) / math.sqrt(K), dim=(-2, -1)).sum()
            loss = ent + 3.0 * cut + 2.0 * ortho
        loss.backward()
        return loss.detach(), S.grad, A.grad

    for a, b in zip(run(True), run(False)):
        torch.testing.assert_close(a, b, **BWD)


def test_diffpool_trains_with_fused_losses(dev):
    """End to end: DiffPool forward + loss + backward reaches the selector's weights."""
    from tgp.poolers import get_pooler
    torch.manual_seed(0)
    B, N, F, K = 4, 48, 8, 6
    pooler = get_pooler("diff", in_channels=F, k=K).to(dev)
    x = torch.randn(B, N, F, device=dev)
    adj = (torch.rand(B, N, N, device=dev) < 0.2).float()
    adj = ((adj + adj.transpose(1, 2)) > 0).float()
    out = pooler(x=x, adj=adj)
    total = out.x.sum() + sum(out.get_loss_value())
    total.backward()
    grads = [p.grad for p in pooler.parameters()]
    assert grads and all(g is not None and torch.isfinite(g).all() and g.abs().sum() > 0 for g in grads)


@pytest.mark.parametrize("K", [1, 3, 4, 20, 64, 128, 130, 300])
def test_edge_dot_and_gradient(dev, K):
    """ss[e] = <S[row_e], S[col_e]> (utils/losses.py:73-127, 661-708) and dS against the torch gathers, for
    unsorted edges with duplicates and self loops, every lane-group width of the kernel."""
    from tgp import functions as Fn
    g = torch.Generator(device=dev).manual_seed(K)
    n, E = 500, 7001
    ei = torch.randint(0, n, (2, E), device=dev, generator=g)
    ei[:, :40] = ei[:, 40:80]          # duplicates
    ei[1, 100:140] = ei[0, 100:140]    # self loops
    S0 = torch.softmax(torch.randn(n, K, device=dev, generator=g), -1)
    go = torch.randn(E, device=dev, generator=g)
    S = S0.clone().requires_grad_(True)
    out = Fn.edge_dot(S, ei)
    (out * go).sum().backward()
    S2 = S0.clone().requires_grad_(True)
    ref = (S2[ei[0]] * S2[ei[1]]).sum(-1)
    (ref * go).sum().backward()
    torch.testing.assert_close(out.detach(), ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(S.grad, S2.grad, rtol=1e-4, atol=1e-5)
    with torch.no_grad():
        torch.testing.assert_close(Fn.edge_dot(S0, ei), ref.detach(), rtol=1e-5, atol=1e-6)


def test_losses_with_graph_sizes_match_padded_run(dev):
    """Link-prediction residual and MinCut denominator on a ragged zero-padded batch: skipping the padding by the
    per-graph sizes must give the same numbers (utils/losses.py:39-56, 644-652)."""
    from tgp import kernels as KK
    g = torch.Generator(device=dev).manual_seed(31)
    sizes = torch.tensor([200, 5, 0, 17, 128, 33, 1, 199, 64, 130], device=dev)
    B, N, K = sizes.numel(), 200, 12
    mask = torch.arange(N, device=dev).unsqueeze(0) < sizes.unsqueeze(1)
    A = (torch.rand(B, N, N, device=dev, generator=g) < 0.1).float() * mask.unsqueeze(1) * mask.unsqueeze(2)
    S = torch.softmax(torch.randn(B, N, K, device=dev, generator=g), -1) * mask.unsqueeze(-1)
    torch.testing.assert_close(KK.link_loss_sq(S, A, sizes), KK.link_loss_sq(S, A), rtol=1e-6, atol=1e-4)
    for a, b in zip(KK.cut_terms(A, S, sizes), KK.cut_terms(A, S)):
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-6)
    ref = torch.stack([torch.norm(A[b] - S[b] @ S[b].t()) ** 2 for b in range(B)])
    torch.testing.assert_close(KK.link_loss_sq(S, A, sizes), ref, rtol=1e-4, atol=1e-3)
