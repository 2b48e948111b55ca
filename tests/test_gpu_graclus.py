"""GraclusSelect (csrc/graclus_match.hip; reference select/graclus_select.py:62-81): matching rounds, per-graph rounds, one-launch select, relabelling, hub rows.

Regrouped by operator in round 6 from the per-round files test_gpu_round2..5.py; the test bodies are unchanged."""
import pytest
import torch
import os
import socket

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


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


# ------------------------------------------------------------------------------ NDPSelect, one large graph (r3)
def _undirected(n, m, seed):
    g = torch.Generator().manual_seed(seed)
    a = torch.randint(0, n, (m,), generator=g)
    b = torch.randint(0, n, (m,), generator=g)
    keep = a != b
    a, b = a[keep], b[keep]
    key = torch.unique(torch.cat([a * n + b, b * n + a]))
    return torch.stack([key // n, key % n])


# ------------------------------------------------------------------------------ hub rows in the row-local coalesce
def _hub_graph(n, pairs, hubs, hub_deg, seed):
    """Undirected, row-major sorted, duplicate-free edge list with `hubs` nodes of ~hub_deg neighbours each."""
    g = torch.Generator().manual_seed(seed)
    a = torch.randint(0, n, (pairs,), generator=g)
    b = torch.randint(0, n, (pairs,), generator=g)
    hub_ids = torch.randperm(n, generator=g)[:hubs]
    ha = hub_ids.repeat_interleave(hub_deg)
    hb = torch.randint(0, n, (hubs * hub_deg,), generator=g)
    a, b = torch.cat([a, ha]), torch.cat([b, hb])
    keep = a != b
    a, b = a[keep], b[keep]
    key = torch.unique(torch.cat([a * n + b, b * n + a]))
    return torch.stack([key // n, key % n]), hub_ids


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


def _graclus_select_outputs(ei, ew, n, gptr, gmax):
    from tgp import kernels
    (index, k, assign, ones), _ = kernels.graclus_match(ei, ew, n, return_row_ptr=True, graph_ptr=gptr,
                                                        max_graph_nodes=gmax, relabel=True)
    return index, k, assign.row_ptr[:k + 1].clone(), assign.perm[:n].clone(), ones


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


def test_graclus_pooler_on_a_hub_graph_stays_on_the_rowlocal_route(dev):
    """Whole `graclus` forward on a graph with hub nodes: the Connect no longer falls to the radix route for one long
    supernode row, and the result equals the oracle's Reduce + Connect given the same clustering."""
    import tgp_oracle as O
    from tgp import kernels
    from tgp.poolers import get_pooler
    n = 60_000
    ei, _ = _hub_graph(n, 150_000, 4, 30_000, 7)
    x = torch.randn(n, 16, generator=torch.Generator().manual_seed(1))
    pooler = get_pooler("graclus").to(dev).eval()
    ei_d = ei.to(dev)
    with torch.no_grad():
        out = pooler(x=x.to(dev), adj=ei_d)
    hub = kernels._HUB_ROWS.get(id(ei_d))
    assert hub is not None and hub[0]() is ei_d  # the row-local route met the hub rows and took them itself
    ref = O.cluster_pool(x, ei, None, None, out.so.cluster_index.cpu(), out.so.num_supernodes)
    assert torch.equal(out.edge_index.cpu(), ref["edge_index"])
    torch.testing.assert_close(out.x.cpu(), ref["x"], rtol=1e-5, atol=1e-5)


def test_graclus_optimistic_route_with_unsorted_rows(dev):
    """A batch of small graphs whose edge list nobody has looked at yet takes the one-launch matching optimistically
    (the row-order check rides on the offsets kernel).  With UNSORTED rows the offsets are no CSR: the per-graph kernel
    must refuse before reading through them (status 4) and the call must still return a valid maximal matching."""
    from tgp import kernels as K
    x, ei, ew, batch, sizes = _small_batch(200, 10, 60, 4, 33, dev)
    g = torch.Generator(device=dev).manual_seed(1)
    perm = torch.randperm(ei.size(1), device=dev, generator=g)
    ei_u, ew_u = ei[:, perm].contiguous(), ew[perm].abs() + 0.1  # a fresh tensor object: no row-order memo yet
    n = x.size(0)
    ptr = torch.zeros(sizes.numel() + 1, dtype=torch.long)
    ptr[1:] = torch.cumsum(sizes, 0)
    assert K._rows_sorted_memo(ei_u) is None
    label = K.graclus_match(ei_u, ew_u, n, graph_ptr=ptr.to(dev), max_graph_nodes=int(sizes.max()))
    assert K._rows_sorted_memo(ei_u) is False  # found out on the way, remembered
    lab = label.cpu()
    cnt = torch.bincount(lab, minlength=n)
    assert int(cnt.max()) <= 2 and bool((lab <= torch.arange(n)).all())
    r, c = ei_u.cpu()
    paired = cnt[lab] == 2
    partner_ok = torch.zeros(n, dtype=torch.bool)
    same = lab[r] == lab[c]
    partner_ok[r[same & (r != c)]] = True
    assert bool((partner_ok | ~paired).all())            # every pair is an edge
    free = ~paired
    assert not bool((free[r] & free[c] & (r != c)).any())  # maximal: no edge between two single nodes


@pytest.mark.parametrize("weighted", [True, False])
@pytest.mark.parametrize("shape", [(300, 5, 64, 4, False), (2048, 20, 60, 4, False), (257, 1, 30, 6, True),
                                   (64, 40, 64, 16, False)])
def test_one_launch_graclus_select_equals_the_staged_route(dev, weighted, shape, monkeypatch):
    """GraclusSelect of a sorted batch of small graphs (select/graclus_select.py:62-81) as ONE launch -- matching, consecutive
    cluster ids, supernode -> members index -- against the staged route (offsets, CSR gather, symmetrise, one-launch rounds,
    relabel kernels): the same pairs (both use the key on global ids), hence the same index / K / members index, bit for
    bit.  Ragged sizes incl. one-node graphs, isolated nodes, duplicate entries, weight ties (unweighted)."""
    from tgp import kernels
    from tgp.utils.ops import batch_info
    B, lo, hi, deg, dup = shape
    x, ei, ew, batch, sizes = _small_batch(B, lo, hi, 8, 100 + B, dev, deg=deg, dup=dup)
    ew = ew if weighted else None
    n = x.size(0)
    info = batch_info(batch)
    monkeypatch.setattr(kernels, "_GRACLUS_FUSED", False)
    ref = _graclus_select_outputs(ei, ew, n, info.ptr, info.max_nodes)
    monkeypatch.setattr(kernels, "_GRACLUS_FUSED", True)
    for _ in range(3):  # stale status words of earlier calls must read as "not ready"
        got = _graclus_select_outputs(ei, ew, n, info.ptr, info.max_nodes)
        assert got[1] == ref[1]
        for a, b in zip((got[0], got[2], got[3], got[4]), (ref[0], ref[2], ref[3], ref[4])):
            assert torch.equal(a, b)
    # the cluster ids describe a maximal matching: every cluster has one or two members, two members share an edge
    idx, k = got[0], got[1]
    sizes_c = torch.bincount(idx[1], minlength=k)
    assert int(sizes_c.min()) >= 1 and int(sizes_c.max()) <= 2


def test_one_launch_graclus_select_refusals_fall_back(dev, monkeypatch):
    """What the one-launch kernel refuses still gives the staged route's answer through the same call: a directed list
    (entries without a reverse are dropped by both), an UNSORTED list (refused: rows not ascending), a batch with a graph
    beyond 64 nodes (not attempted), an edge that leaves its graph (refused by both per-graph kernels: device-wide rounds)."""
    from tgp import kernels
    from tgp.utils.ops import batch_info
    x, ei, ew, batch, sizes = _small_batch(200, 10, 50, 8, 7, dev)
    n = x.size(0)
    info = batch_info(batch)

    def both(e, w, gptr, gmax):
        monkeypatch.setattr(kernels, "_GRACLUS_FUSED", False)
        ref = _graclus_select_outputs(e.clone(), w, n, gptr, gmax)
        monkeypatch.setattr(kernels, "_GRACLUS_FUSED", True)
        got = _graclus_select_outputs(e.clone(), w, n, gptr, gmax)
        assert got[1] == ref[1]
        for a, b in zip((got[0], got[2], got[3], got[4]), (ref[0], ref[2], ref[3], ref[4])):
            assert torch.equal(a, b)
        return got

    keep = torch.rand(ei.size(1), generator=torch.Generator().manual_seed(3)).to(dev) < 0.8   # directed: reverses missing
    both(ei[:, keep].contiguous(), ew[keep].contiguous(), info.ptr, info.max_nodes)
    perm = torch.randperm(ei.size(1), generator=torch.Generator().manual_seed(4)).to(dev)       # unsorted rows
    both(ei[:, perm].contiguous(), ew[perm].contiguous(), info.ptr, info.max_nodes)
    leak = ei.clone()
    leak[1, 0] = n - 1                                                                         # leaves its graph
    both(leak, ew, info.ptr, info.max_nodes)
    x2, ei2, ew2, batch2, _ = _small_batch(40, 30, 100, 8, 9, dev)                             # graphs beyond 64 nodes
    info2 = batch_info(batch2)
    n2 = x2.size(0)
    monkeypatch.setattr(kernels, "_GRACLUS_FUSED", True)
    (index, k, assign, ones), _ = kernels.graclus_match(ei2, ew2, n2, return_row_ptr=True, graph_ptr=info2.ptr,
                                                        max_graph_nodes=info2.max_nodes, relabel=True)
    assert index.shape == (2, n2) and 0 < k <= n2
