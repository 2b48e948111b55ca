"""Dense Reduce / Connect / post-processing and the dense losses' kernels (csrc/dense.hip, gemm_mfma.h, gemm_f64.hip, dense_post.h, losses.hip; reference base_reduce.py:158-161, dense_conn.py:111-122, utils/ops.py:282-335, utils/losses.py).

Regrouped by operator in round 6 from the per-round files test_gpu_round2..5.py; the test bodies are unchanged."""
import math
import warnings
import pytest
import torch
import os
import socket
import sys

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


# ------------------------------------------------------------------------------ one-launch sparse pooling of small graphs
def _small_batch(num_graphs, lo, hi, f, seed, dev, deg=4, dup=False):
    """PyG-style batch: sorted batch vector, row-major sorted undirected edge list (optionally with duplicate entries)."""
    g = torch.Generator().manual_seed(seed)
    sizes = torch.randint(lo, hi + 1, (num_graphs,), generator=g)
    n = int(sizes.sum())
    batch = torch.repeat_interleave(torch.arange(num_graphs), sizes)
    start = torch.cumsum(sizes, 0) - sizes
    src = torch.arange(n).repeat_interleave(max(deg // 2, 1))
    dst = start[batch[src]] + (torch.rand(src.numel(), generator=g) * sizes[batch[src]]).long()
    keep = src != dst
    src, dst = src[keep], dst[keep]
    key = torch.cat([src * n + dst, dst * n + src])
    key = torch.sort(key)[0] if dup else torch.unique(key)
    ei = torch.stack([key // n, key % n])
    x = torch.randn(n, f, generator=g)
    ew = torch.rand(ei.size(1), generator=g) + 0.25
    ew[torch.rand(ei.size(1), generator=g) < 0.05] = 0.0  # some weights the eps filter drops
    return x.to(dev), ei.to(dev), ew.to(dev), batch.to(dev), sizes


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


# ------------------------------------------------------------------------------------------- runtime eps
def test_dense_connect_unbatched_filters_small_edges_with_patched_eps(dev, monkeypatch):
    """Mirror of reference tests/connect/test_dense_conn.py:444-463: ops.eps is read at call time."""
    from tgp.connect import DenseConnect
    from tgp.select import SelectOutput
    from tgp.utils import ops as ops_module
    ei = torch.tensor([[0, 1, 1], [1, 0, 2]], device=dev)
    ew = torch.tensor([0.5, 0.5, 2.0], device=dev)
    so = SelectOutput(s=torch.eye(3, device=dev))
    conn = DenseConnect(sparse_output=True, remove_self_loops=False, degree_norm=False)
    adj0, w0 = conn(edge_index=ei, edge_weight=ew, so=so)
    assert w0.numel() == 3
    monkeypatch.setattr(ops_module, "eps", 1.0)
    adj, w = conn(edge_index=ei, edge_weight=ew, so=so)
    assert adj.size(0) == 2 and w.numel() == 1 and bool(torch.all(w > 1.0))


# ------------------------------------------------------------------------------------------- small fixes
def test_orthogonality_loss_accepts_a_single_2d_assignment(dev):
    """reference utils/losses.py:59-70 uses transpose(-2,-1) and norm(dim=(-2,-1)): S [N,K] is valid input."""
    from tgp.utils.losses import orthogonality_loss
    g = torch.Generator().manual_seed(1)
    s = torch.softmax(torch.randn(37, 5, generator=g), -1)
    sts = s.t() @ s
    ref = torch.norm(sts / torch.norm(sts) - torch.eye(5) / math.sqrt(5))
    got = orthogonality_loss(s.to(dev))
    assert got.dim() == 0
    torch.testing.assert_close(got.cpu(), ref, rtol=1e-5, atol=1e-6)
    sd = s.to(dev).requires_grad_(True)
    orthogonality_loss(sd).backward()
    sr = s.clone().requires_grad_(True)
    t = sr.t() @ sr
    torch.norm(t / torch.norm(t) - torch.eye(5) / math.sqrt(5)).backward()
    torch.testing.assert_close(sd.grad.cpu(), sr.grad, rtol=1e-4, atol=1e-6)


def test_float64_inputs_are_computed_in_float64(dev):
    """r5 (was: "announced as fp32 arithmetic"): a float64 feature matrix through the dense Reduce gives the fp64 product,
    no warning, as torch.matmul does in the reference (reduce/base_reduce.py:158-161)."""
    from tgp.reduce import BaseReduce
    from tgp.select import SelectOutput
    s = torch.softmax(torch.randn(1, 4, 2, device=dev, dtype=torch.float64), -1)
    so = SelectOutput(s=s)
    x = torch.randn(1, 4, 3, dtype=torch.float64, device=dev)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        out, _ = BaseReduce()(x, so)
    assert out.dtype == torch.float64
    torch.testing.assert_close(out, s.transpose(1, 2) @ x, rtol=1e-13, atol=1e-13)


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


def test_float64_dense_postprocessing_and_block_diag_run_in_fp64(dev):
    """postprocess_adj_pool_dense on a float64 [B,K,K] tensor (utils/ops.py:282-335), all 16 flag combinations, and
    dense_to_block_diag (utils/ops.py:53-82) in fp64: 1e-12 against the oracle in fp64."""
    import tgp_oracle as O
    from tgp import functions as Fn
    from tgp.utils.ops import postprocess_adj_pool_dense
    g = torch.Generator().manual_seed(77)
    for (B, K) in ((5, 17), (2, 130)):
        a = torch.rand(B, K, K, generator=g, dtype=torch.float64) * (torch.rand(B, K, K, generator=g) < 0.4)
        a[0, :, 3] = 0.0  # an empty column: the clamp(min=eps) branch
        for rsl in (True, False):
            for dn in (True, False):
                for at in (True, False):
                    for ewn in (True, False):
                        got = postprocess_adj_pool_dense(a.to(dev), rsl, dn, at, ewn)
                        ref = O.postprocess_dense(a, rsl, dn, at, ewn)
                        assert got.dtype == torch.float64
                        torch.testing.assert_close(got.cpu(), ref, rtol=1e-12, atol=1e-12)
        ei, w = Fn.block_diag_edges(a.to(dev))
        r_ei, r_w = O.dense_to_block_diag(a)
        assert w.dtype == torch.float64 and torch.equal(ei.cpu(), r_ei) and torch.equal(w.cpu(), r_w)


def test_float64_dense_gemm_path_runs_in_fp64_without_a_warning(dev):
    """r5: the dense GEMM path has an fp64 form (v_mfma_f64_16x16x4_f64): a float64 DiffPool input is computed in double
    and nothing is announced (replaces r4's test that asserted the fp32 narrowing; the value checks live in
    test_float64_dense_pool_vs_fp64_oracle in this file)."""
    import warnings
    from tgp.poolers import get_pooler
    x, ei, ew, batch, _ = _small_batch(8, 20, 40, 8, 1, dev)
    pooler = get_pooler("diff", in_channels=8, k=4).to(dev).double().eval()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        with torch.no_grad():
            out = pooler(x=x.double(), adj=ei, edge_weight=ew.double(), batch=batch)
    assert out.x.dtype == torch.float64 and out.edge_index.dtype == torch.float64


def test_bmm_accumulate_in_the_epilogue(dev):
    """C += op(A) B in the GEMM epilogue (tgp_bmm_accumulate_f32): the bits of the separate product + add, for both
    operand layouts, ragged shapes and the big-tile path; DenseConnect's two-term dS uses it."""
    from tgp import kernels as K
    g = torch.Generator(device=dev).manual_seed(0)
    for (G, M, Kd, Nc, ta) in ((3, 70, 33, 20, False), (2, 1024, 1024, 128, False), (4, 100, 257, 65, True)):
        a = torch.randn((G, Kd, M) if ta else (G, M, Kd), device=dev, generator=g)
        b = torch.randn(G, Kd, Nc, device=dev, generator=g)
        c0 = torch.randn(G, M, Nc, device=dev, generator=g)
        want = c0 + K.bmm(a, b, trans_a=ta)
        got = K.bmm(a, b, trans_a=ta, accumulate_into=c0.clone())
        assert torch.equal(got, want)


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
