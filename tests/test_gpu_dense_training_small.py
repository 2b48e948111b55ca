"""Training of the dense poolers on batches of small graphs (csrc/dense_graph_kernels.h: one-launch forward + backward; functions._DensePoolSmallFn / _SelectPoolSmallFn / _SelectPoolSparseFn).

Regrouped by operator in round 6 from the per-round files test_gpu_round2..5.py; the test bodies are unchanged."""
import pytest
import torch
import os
import sys
import warnings

pytestmark = pytest.mark.gpu


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


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["mincut", "diff"])
def test_fused_function_hands_the_two_losses_out_as_scalars(dev, which):
    """functions.dense_pool_small(loss_scalars=True): the two auxiliary losses are 0-dim outputs (LossPair); their values
    and the gradients they send equal the [2,B]-terms / [2]-diff form's; a sum loss (expanded scalar gradients, read by
    the kernel as ONE value) equals the same loss with materialised gradients; a loss that uses only one of the two
    leaves the other's upstream gradient missing (NULL in the C call)."""
    from tgp import functions as Fn, kernels as K_
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


@pytest.mark.gpu
@pytest.mark.parametrize("hidden", [None, 12])
@pytest.mark.parametrize("train", [False, True])
def test_diffpool_padded_small_graphs_losses_from_the_pooling_launch(dev, monkeypatch, hidden, train):
    """r6: DiffPool on PADDED inputs that take the one-wave-per-graph kernel (x [B,N,F], adj [B,N,N] -- what the second
    layer of a hierarchical model gets): both losses come from the launch's per-graph records and ONE tail launch
    (tgp_dense_pool_small_diff_f32 + tgp_diffpool_stats_tail_f32), with the selector folded in (single Linear) or S handed
    over (hidden layer); inference and training.  Against the oracle in float64 (utils/losses.py:644-658, 476-483)."""
    import tgp_oracle as O
    from tgp import kernels as K_
    from tgp.poolers import get_pooler
    g = torch.Generator().manual_seed(41)
    B, Nn, F, k = 80, 40, 16, 12  # (the one-wave-per-graph kernel takes batches of >= 64 graphs)
    a = (torch.rand(B, Nn, Nn, generator=g) < 0.15).float() * (torch.rand(B, Nn, Nn, generator=g) + 0.2)
    a = a + a.transpose(1, 2)
    x = torch.randn(B, Nn, F, generator=g)
    chans = F if hidden is None else [F, hidden]
    torch.manual_seed(3)
    pooler = get_pooler("diff", in_channels=chans, k=k, **({} if hidden is None else {"act": "tanh"})).to(dev)
    pooler.train(train)
    lins = pooler.selector.mlp.lins
    tails = []
    real = K_.diffpool_stats_tail
    monkeypatch.setattr(K_, "diffpool_stats_tail", lambda *a_, **k_: (tails.append(1), real(*a_, **k_))[1])
    xg = x.to(dev).requires_grad_(train)
    with torch.set_grad_enabled(train):
        out = pooler(x=xg, adj=a.to(dev))
    assert tails == [1]  # the losses took the records path
    ws = [l.weight.detach().cpu().double().requires_grad_(train) for l in lins]
    bs = [l.bias.detach().cpu().double().requires_grad_(train) for l in lins]
    xr = x.double().requires_grad_(train)
    ref = O.dense_pool("diff", xr, a.double(), None, None, ws, bs, act=None if hidden is None else "tanh")
    torch.testing.assert_close(out.x.detach().cpu().double(), ref["x"].detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(out.edge_index.detach().cpu().double(), ref["edge_index"].detach(), rtol=1e-5, atol=1e-6)
    for name, want in ref["loss"].items():
        torch.testing.assert_close(out.loss[name].detach().cpu().double(), want.detach(), rtol=1e-5, atol=1e-7,
                                   msg=lambda m, n=name: f"{n}: {m}")
    if train:
        (out.x.sum() * 0.3 + out.edge_index.square().sum() + 0.7 * out.loss["link_loss"] + 1.3 * out.loss["entropy_loss"]).backward()
        (ref["x"].sum() * 0.3 + ref["edge_index"].square().sum() + 0.7 * ref["loss"]["link_loss"]
         + 1.3 * ref["loss"]["entropy_loss"]).backward()
        pairs = [(xg.grad, xr.grad)] + [(l.weight.grad, w.grad) for l, w in zip(lins, ws)] + \
            [(l.bias.grad, b.grad) for l, b in zip(lins, bs)]
        for got, want in pairs:
            torch.testing.assert_close(got.cpu().double(), want, rtol=2e-4, atol=1e-5 * max(1.0, float(want.abs().max())))
