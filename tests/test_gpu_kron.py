"""A9 / N4: the block-batched Kron reduction kernel (tgp_kron_batched_{count,fill}) against the oracle's restatement of
the reference's scipy route (connect/kron_conn.py:117-165) on whole-batch Laplacians: edge_index bit-exact (row-major
order of the pooled batch), weights within 1e-5 relative."""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def make_batch(sizes, seed, weighted=True, density=None, connected=True):
    """Undirected graphs; every component keeps at least one node so that L[-,-] is regular; kept = random half
    with node 0 of every graph (and of every chain link) kept."""
    g = torch.Generator().manual_seed(seed)
    eis, ews, bs, keep, off = [], [], [], [], 0
    for gi, n in enumerate(sizes):
        p = density if density is not None else min(1.0, 4.0 / max(n, 2))
        a = torch.triu(torch.rand(n, n, generator=g) < p, 1)
        if connected and n > 1:  # a spanning chain: one component
            idx = torch.arange(n - 1)
            a[idx, idx + 1] = True
        r, c = a.nonzero(as_tuple=True)
        w = (torch.rand(r.numel(), generator=g) + 0.1) if weighted else torch.ones(r.numel())
        eis.append(torch.stack([torch.cat([r, c]), torch.cat([c, r])]) + off)
        ews.append(torch.cat([w, w]))
        bs.append(torch.full((n,), gi, dtype=torch.long))
        k = torch.rand(n, generator=g) < 0.5
        k[0] = True
        keep.append(k)
        off += n
    ei, ew, batch, keep = torch.cat(eis, 1), torch.cat(ews), torch.cat(bs), torch.cat(keep)
    return ei, ew, batch, keep.nonzero().view(-1)


def oracle_kron(ei, ew, n, idx_pos, thr=1e-2):
    import tgp_oracle as O
    L = O.laplacian_scipy(ei, ew.double(), n).astype(np.float64)
    return L, O.kron_connect(L, idx_pos, thr)


def so_of(idx_pos, n, dev, L=None):
    from tgp.select import SelectOutput
    k = idx_pos.numel()
    kw = {} if L is None else {"L": L}
    return SelectOutput(node_index=idx_pos.to(dev), cluster_index=torch.arange(k, device=dev), num_nodes=n,
                        num_supernodes=k, **kw)


def native_only(conn):
    """Make every non-native route of KronConnect fail loudly."""
    def boom(*a, **k):
        raise AssertionError("KronConnect left the block-batched native route")
    conn._kron_on_device = boom
    return conn


def check(ei_out, ew_out, ref, dev):
    ei_ref, ew_ref = ref
    assert ei_out.device.type == "cuda" and ew_out.dtype == torch.float32
    assert torch.equal(ei_out.cpu(), ei_ref), (ei_out.shape, ei_ref.shape)
    torch.testing.assert_close(ew_out.cpu(), ew_ref, rtol=1e-5, atol=0)


@pytest.mark.parametrize("weighted", [True, False])
def test_kron_batched_small_graphs_vs_oracle(dev, weighted):
    from tgp.connect import KronConnect
    g = torch.Generator().manual_seed(3)
    sizes = torch.randint(2, 61, (300,), generator=g).tolist() + [1, 1, 2, 64, 65, 127, 128]
    ei, ew, batch, idx_pos = make_batch(sizes, seed=11, weighted=weighted)
    n = batch.numel()
    L, ref = oracle_kron(ei, ew, n, idx_pos)
    conn = native_only(KronConnect())
    # with the selector's Laplacian (NDP) ...
    out = conn(ei.to(dev), so_of(idx_pos, n, dev, L=L.astype(np.float32) if not weighted else L),
               edge_weight=ew.to(dev), batch=batch.to(dev))
    check(out[0], out[1], ref, dev)
    # ... and without (Laplacian formed from the edge list in the kernel)
    with pytest.warns(UserWarning, match="Laplacian not provided"):
        out = conn(ei.to(dev), so_of(idx_pos, n, dev), edge_weight=ew.to(dev), batch=batch.to(dev))
    check(out[0], out[1], ref, dev)


def test_kron_batched_workspace_resident_graphs(dev):
    """Graphs beyond the LDS capacity (129 .. 1024 nodes) run the same elimination on a workspace slab."""
    from tgp.connect import KronConnect
    sizes = [40, 129, 30, 400, 257, 12, 1024]
    ei, ew, batch, idx_pos = make_batch(sizes, seed=5)
    n = batch.numel()
    L, ref = oracle_kron(ei, ew, n, idx_pos)
    out = native_only(KronConnect())(ei.to(dev), so_of(idx_pos, n, dev, L=L), edge_weight=ew.to(dev),
                                     batch=batch.to(dev))
    check(out[0], out[1], ref, dev)


def test_kron_from_edge_list_unsorted_duplicates_self_loops(dev):
    """No Laplacian on the SelectOutput: L = D - A is formed in the kernel from an edge list in any order, with
    duplicate entries (summed) and self loops (ignored), as get_laplacian + to_scipy_sparse_matrix do."""
    from tgp.connect import KronConnect
    sizes = [17, 33, 5, 48]
    ei, ew, batch, idx_pos = make_batch(sizes, seed=9)
    n = batch.numel()
    g = torch.Generator().manual_seed(1)
    dup = torch.randint(0, ei.size(1), (40,), generator=g)
    loops = torch.randint(0, n, (10,), generator=g)
    ei2 = torch.cat([ei, ei[:, dup], torch.stack([loops, loops])], 1)
    ew2 = torch.cat([ew, ew[dup], torch.rand(10, generator=g)])
    perm = torch.randperm(ei2.size(1), generator=g)
    ei2, ew2 = ei2[:, perm], ew2[perm]
    _, ref = oracle_kron(ei2, ew2, n, idx_pos)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = native_only(KronConnect())(ei2.to(dev), so_of(idx_pos, n, dev), edge_weight=ew2.to(dev),
                                         batch=batch.to(dev))
    check(out[0], out[1], ref, dev)
    # no batch vector: the whole input is one graph (n = 103 nodes, still one LDS-resident block)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = native_only(KronConnect())(ei2.to(dev), so_of(idx_pos, n, dev), edge_weight=ew2.to(dev))
    check(out[0], out[1], ref, dev)


def test_kron_thresholds(dev):
    from tgp.connect import KronConnect
    ei, ew, batch, idx_pos = make_batch([30, 45, 22], seed=2)
    n = batch.numel()
    for thr in (0.0, 1e-2, 0.2, 10.0):
        L, ref = oracle_kron(ei, ew, n, idx_pos, thr)
        out = native_only(KronConnect(sparse_threshold=thr))(ei.to(dev), so_of(idx_pos, n, dev, L=L),
                                                             edge_weight=ew.to(dev), batch=batch.to(dev))
        check(out[0], out[1], ref, dev)


def test_kron_singular_block_is_damped(dev):
    """A component made of dropped nodes only makes L[-,-] exactly singular; the reference's intent (kron_conn.py:
    131-135) is the Marquardt-Levenberg damping 1e-6 I: compare with a dense fp64 evaluation of exactly that."""
    from tgp.connect import KronConnect
    # graph 0: path 0-1-2 plus a separate dropped pair 3-4; graph 1: triangle + isolated dropped node
    ei = torch.tensor([[0, 1, 1, 2, 3, 4, 5, 6, 6, 7, 5, 7], [1, 0, 2, 1, 4, 3, 6, 5, 7, 6, 7, 5]])
    ew = torch.tensor([1.0, 1.0, 2.0, 2.0, 1.0, 1.0, 1.0, 1.0, 3.0, 3.0, 0.5, 0.5])
    batch = torch.tensor([0, 0, 0, 0, 0, 1, 1, 1, 1])
    idx_pos = torch.tensor([0, 2, 5, 7])
    n = 9
    import tgp_oracle as O
    L = torch.from_numpy(O.laplacian_scipy(ei, ew.double(), n).toarray())
    neg = torch.tensor([1, 3, 4, 6, 8])
    lnn = L[neg][:, neg] + 1e-6 * torch.eye(neg.numel(), dtype=torch.float64)
    lnew = L[idx_pos][:, idx_pos] - L[idx_pos][:, neg] @ torch.linalg.solve(lnn, L[neg][:, idx_pos])
    a = -lnew
    a = a * (a.abs() > 1e-2)
    a.fill_diagonal_(0)
    nz = a.nonzero()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = native_only(KronConnect())(ei.to(dev), so_of(idx_pos, n, dev), edge_weight=ew.to(dev),
                                         batch=batch.to(dev))
    assert torch.equal(out[0].cpu(), nz.t())
    torch.testing.assert_close(out[1].cpu(), a[nz[:, 0], nz[:, 1]].float(), rtol=1e-5, atol=0)


def test_kron_declines_oversized_graph_and_falls_back(dev):
    """A graph beyond tgp_kron_batched_max_graph_nodes(): the library route takes the batch; same result."""
    from tgp import kernels as K
    from tgp.connect import KronConnect
    big = K.kron_max_graph_nodes() + 40
    ei, ew, batch, idx_pos = make_batch([20, big], seed=4)
    n = batch.numel()
    L, ref = oracle_kron(ei, ew, n, idx_pos)
    out = KronConnect()(ei.to(dev), so_of(idx_pos, n, dev, L=L), edge_weight=ew.to(dev), batch=batch.to(dev))
    check(out[0], out[1], ref, dev)


def test_ndp_pooler_2048_graph_batch_connect_stays_on_device(dev, monkeypatch):
    """get_pooler("ndp") on a PROTEINS-sized batch (2048 graphs, n ~ 40): Connect never touches
    the host solvers; result equals the oracle's Kron reduction of the same SelectOutput."""
    import scipy.sparse.linalg as spla
    from tgp.poolers import get_pooler
    g = torch.Generator().manual_seed(0)
    sizes = torch.randint(20, 61, (2048,), generator=g).tolist()
    ei, ew, batch, _ = make_batch(sizes, seed=21)
    n = batch.numel()
    x = torch.randn(n, 8, generator=g)
    pool = get_pooler("ndp").to(dev)
    so = pool.select(edge_index=ei.to(dev), edge_weight=ew.to(dev), batch=batch.to(dev), num_nodes=n)

    def boom(*a, **k):
        raise AssertionError("host / library solver called from KronConnect")
    monkeypatch.setattr(spla, "spsolve", boom)
    monkeypatch.setattr(torch.linalg, "solve", boom)
    ei_p, ew_p = pool.connect(edge_index=ei.to(dev), so=so, edge_weight=ew.to(dev), batch=batch.to(dev))
    out = pool(x=x.to(dev), adj=ei.to(dev), edge_weight=ew.to(dev), batch=batch.to(dev))  # whole call, fresh selection
    monkeypatch.undo()
    import tgp_oracle as O
    ref = O.kron_connect(so.L.astype(np.float64), so.node_index.cpu())
    check(ei_p, ew_p, ref, dev)
    k = out.so.num_supernodes
    assert out.x.shape == (k, 8) and out.batch.numel() == k and int(out.edge_index.max()) < k
