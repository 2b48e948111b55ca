"""Tensors beyond 2^31 elements (the 288 GB of one MI355X make them ordinary): every kernel that turns
(graph, row, column) into an element offset must do it in 64 bits.  Each case builds one tensor of > 2^31
elements on the device (8-10 GB) and checks the slices that sit beyond the 32-bit boundary against small
torch computations of the same slice."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _release():
    yield
    torch.cuda.empty_cache()


def _enough_memory(dev, gib):
    free, _ = torch.cuda.mem_get_info(dev)
    if free < gib * 2 ** 30:
        pytest.skip(f"needs {gib} GiB of free device memory")


def test_dense_pool_batch_beyond_2g_elements(dev):
    """[B,N,N] adjacency of 2.4e9 elements through the tiled path, the losses and the block-diagonal export."""
    _enough_memory(dev, 40)
    import tgp.kernels as K
    B, N, Kc, F = 36, 8192, 64, 32
    assert B * N * N > 2 ** 31
    g = torch.Generator(device=dev).manual_seed(5)
    adj = torch.empty(B, N, N, device=dev)
    for b in range(B):
        adj[b] = (torch.rand(N, N, device=dev, generator=g) < 0.01).float()
    s = torch.softmax(torch.randn(B, N, Kc, device=dev, generator=g), dim=-1)
    x = torch.randn(B, N, F, device=dev, generator=g)
    xp, raw, _ = K.dense_pool(s, adj, x, want_raw=True, want_post=False)
    sq = K.link_loss_sq(s, adj)
    deg, q, den = K.cut_terms(adj, s)
    for b in (0, 15, B - 1):
        sd, ad = s[b].double(), adj[b].double()
        torch.testing.assert_close(raw[b], (sd.T @ ad @ sd).float(), rtol=1e-4, atol=1e-3)
        torch.testing.assert_close(xp[b], (sd.T @ x[b].double()).float(), rtol=1e-4, atol=1e-3)
        torch.testing.assert_close(sq[b], ((ad - sd @ sd.T) ** 2).sum().float(), rtol=1e-4, atol=1e-2)
        torch.testing.assert_close(deg[b], ad.sum(1).float(), rtol=1e-6, atol=1e-4)
        torch.testing.assert_close(q[b], (sd * sd).sum(1).float(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(den[b], (ad.sum(1) * (sd * sd).sum(1)).sum().float(), rtol=1e-4, atol=1e-2)
    # per-graph products with the adjacency as the left operand (DenseConnect's backward)
    u = K.bmm(adj, s)
    ut = K.bmm(adj, s, trans_a=True)
    for b in (0, B - 1):
        torch.testing.assert_close(u[b], (adj[b].double() @ s[b].double()).float(), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(ut[b], (adj[b].double().T @ s[b].double()).float(), rtol=1e-4, atol=1e-4)


def test_densify_and_export_beyond_2g_elements(dev):
    """to_dense_adj writes a [B,N,N] tensor of 2.4e9 elements; block_diag_edges reads it back."""
    _enough_memory(dev, 40)
    import tgp.kernels as K
    B, N = 36, 8192
    g = torch.Generator(device=dev).manual_seed(6)
    per = 20_000
    rows = torch.randint(0, N, (B, per), device=dev, generator=g)
    cols = torch.randint(0, N, (B, per), device=dev, generator=g)
    off = (torch.arange(B, device=dev) * N).view(B, 1)
    ei = torch.stack([(rows + off).view(-1), (cols + off).view(-1)])
    w = torch.rand(B * per, device=dev, generator=g) + 0.5
    batch = torch.arange(B, device=dev).repeat_interleave(N)
    ptr = torch.arange(B + 1, device=dev) * N
    dense = K.to_dense_adj(ei, w, batch, ptr, B, N, transposed=False)
    assert dense.numel() > 2 ** 31
    for b in (0, B - 1):
        ref = torch.zeros(N * N, device=dev).index_add_(0, rows[b] * N + cols[b], w[b * per:(b + 1) * per]).view(N, N)
        torch.testing.assert_close(dense[b], ref, rtol=1e-6, atol=1e-6)
    # the export indexes B*K*K in 32 bits (pooled adjacencies are small): an oversized call fails loudly ...
    from tgp._native import TgpNativeError
    with pytest.raises(TgpNativeError, match="2\\^31"):
        K.block_diag_edges(dense)
    # ... and the last graphs (base pointer beyond 8 GiB) export exactly
    tail = 4
    out_ei, out_w = K.block_diag_edges(dense[B - tail:])
    flat = out_ei[0] * (tail * N) + out_ei[1]
    assert bool((flat[1:] > flat[:-1]).all())
    last = out_ei[0] >= (tail - 1) * N
    r, c = dense[B - 1].nonzero(as_tuple=True)
    assert torch.equal(out_ei[0][last] - (tail - 1) * N, r) and torch.equal(out_ei[1][last] - (tail - 1) * N, c)
    torch.testing.assert_close(out_w[last], dense[B - 1][r, c], rtol=0, atol=0)
    torch.testing.assert_close(out_w.double().sum(), dense[B - tail:].double().sum(), rtol=1e-9, atol=0)


def test_sparse_reduce_and_lift_beyond_2g_elements(dev):
    """[N,F] features of 2.3e9 elements: TopK gather-reduce, cluster segment sums, lift and the dense-batch pair."""
    _enough_memory(dev, 60)
    import tgp.kernels as K
    N, F = 18_000_000, 128
    assert N * F > 2 ** 31
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn(N, F, device=dev, generator=g)
    # (1) TopK-style gather of half the nodes with a weight
    perm = torch.randperm(N, device=dev, generator=g)[: N // 2]
    wt = torch.rand(N // 2, device=dev, generator=g)
    idx = K.build_assign_index(torch.arange(N // 2, device=dev), N // 2)
    out = K.reduce_sparse(x, perm, wt, idx)
    probe = torch.tensor([0, 1, N // 4, N // 2 - 2, N // 2 - 1], device=dev)
    torch.testing.assert_close(out[probe], x[perm[probe]] * wt[probe].unsqueeze(1), rtol=1e-6, atol=1e-6)
    hi = perm.argmax()  # the row with the largest source offset
    torch.testing.assert_close(out[hi], x[perm[hi]] * wt[hi], rtol=1e-6, atol=1e-6)
    del out
    # (2) clusters of ~4 nodes, summed
    Kc = N // 4
    cl = torch.randint(0, Kc, (N,), device=dev, generator=g)
    idx = K.build_assign_index(cl, Kc)
    out = K.reduce_sparse(x, torch.arange(N, device=dev), None, idx)
    for c in (0, Kc // 2, Kc - 1):
        members = (cl == c).nonzero().view(-1)
        torch.testing.assert_close(out[c], x[members].sum(0), rtol=1e-5, atol=1e-5)
    col_ref = x[:, :4].double().sum(0)
    torch.testing.assert_close(out[:, :4].double().sum(0), col_ref, rtol=1e-6, atol=1e-3)
    del out
    # (3) to_dense_batch / from_dense_batch round trip over graphs of 2250 nodes
    per = 2250
    B = N // per
    batch = torch.arange(B, device=dev).repeat_interleave(per)
    ptr = torch.arange(B + 1, device=dev) * per
    dense, mask = K.to_dense_batch(x, batch, ptr, B, per)
    assert bool(mask.all())
    assert torch.equal(dense.view(N, F)[-1000:], x[-1000:]) and torch.equal(dense.view(N, F)[:1000], x[:1000])
    back = K.from_dense_batch(dense, batch, ptr, per)
    assert torch.equal(back[-1000:], x[-1000:]) and torch.equal(back[N // 2: N // 2 + 1000], x[N // 2: N // 2 + 1000])


def test_row_operators_beyond_2g_elements(dev):
    """One [N,128] matrix of 2.3e9 elements through every operator that addresses rows of a node matrix:
    scores, their weight gradient, edge / pair dot products, the sparse-adjacency product and both per-graph
    products on the un-padded batch."""
    _enough_memory(dev, 60)
    import tgp.kernels as K
    N, F = 18_000_000, 128
    g = torch.Generator(device=dev).manual_seed(8)
    s = torch.randn(N, F, device=dev, generator=g)
    hi = torch.arange(N - 4096, N, device=dev)  # rows whose element offsets exceed 2^31
    assert int(hi[0]) * F > 2 ** 31
    # scores and their weight gradient
    w = torch.randn(F, device=dev, generator=g)
    score = K.row_dot(s, w)
    torch.testing.assert_close(score[hi], (s[hi].double() @ w.double()).float(), rtol=1e-5, atol=1e-5)
    gsel = torch.zeros(N, device=dev)
    gsel[hi] = torch.randn(hi.numel(), device=dev, generator=g)
    torch.testing.assert_close(K.weighted_colsum(s, gsel), (gsel[hi].double() @ s[hi].double()).float(), rtol=1e-5, atol=1e-4)
    # edge / pair dots between the top rows and the bottom rows
    E = 1_000_000
    lo_idx = torch.randint(0, 4096, (E,), device=dev, generator=g)
    hi_idx = torch.randint(N - 4096, N, (E,), device=dev, generator=g)
    ref = (s[lo_idx[:2000]].double() * s[hi_idx[:2000]].double()).sum(1).float()
    torch.testing.assert_close(K.edge_dot(s, torch.stack([lo_idx, hi_idx]))[:2000], ref, rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(K.pair_dot(s, lo_idx, s, hi_idx)[:2000], ref, rtol=1e-5, atol=1e-4)
    # T = A S with a row-sorted edge list whose columns are the high rows
    rows = torch.sort(torch.randint(0, N, (E,), device=dev, generator=g)).values
    ei = torch.stack([rows, hi_idx])
    ew = torch.rand(E, device=dev, generator=g)
    t = K.spmm_sorted(ei, ew, N, s)
    last_row = int(rows[-1])
    sel = rows == last_row
    torch.testing.assert_close(t[last_row], (ew[sel].double().unsqueeze(1) * s[hi_idx[sel]].double()).sum(0).float(), rtol=1e-5, atol=1e-5)
    first_row = int(rows[0])
    sel = rows == first_row
    torch.testing.assert_close(t[first_row], (ew[sel].double().unsqueeze(1) * s[hi_idx[sel]].double()).sum(0).float(), rtol=1e-5, atol=1e-5)
    del t
    # per-graph products on the un-padded batch: graphs of 2250 nodes
    per = 2250
    B = N // per
    ptr = torch.arange(B + 1, device=dev) * per
    y = s[:, :16].contiguous()
    c = K.segment_gemm_tn(s, y, ptr, per)
    for b in (0, B - 1):
        sl = slice(b * per, (b + 1) * per)
        torch.testing.assert_close(c[b], (s[sl].double().T @ y[sl].double()).float(), rtol=1e-4, atol=1e-3)
    bm = torch.randn(B, F, 8, device=dev, generator=g)
    z = K.segment_gemm_nn(s, bm, ptr, per)
    for b in (0, B - 1):
        sl = slice(b * per, (b + 1) * per)
        torch.testing.assert_close(z[sl], (s[sl].double() @ bm[b].double()).float(), rtol=1e-4, atol=1e-3)
