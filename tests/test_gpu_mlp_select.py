"""MLPSelect's last layer (csrc/mlp_select.hip; reference select/mlp_select.py:105-147): forward, backward, the folds into the pooling kernels.

Regrouped by operator in round 6 from the per-round files test_gpu_round2..5.py; the test bodies are unchanged."""
import pytest
import torch
import os
import sys
import warnings

pytestmark = pytest.mark.gpu

SHAPES = [  # (leading shape, F, K)
    ((2048, 60), 32, 20),      # C3: PROTEINS-shaped MinCut batch
    ((32, 1024), 64, 128),     # C2
    ((3, 50), 16, 8), ((1, 37), 7, 5), ((5, 33), 33, 31), ((2, 100), 130, 65), ((4, 64), 96, 256),
    ((2, 40), 300, 200),       # F * K beyond the LDS budget: W walks through LDS in slices
    ((777,), 12, 10),          # unbatched [N, F]
    ((2, 70), 24, 300),        # K > 256: tiled GEMM + softmax kernel
]


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


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
