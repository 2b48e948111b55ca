"""Seeded random shapes through the r6 training nodes of the dense poolers (functions._PoolLargeFn: the densifying route;
functions._PoolUnbatchedFn: the un-padded rows route and the unbatched poolers): odd K and F (the scalar copy / general
post-processing paths, operand rows that are not multiples of four floats), graphs of two or three nodes beside graphs
of a few hundred, isolated nodes, DIRECTED edge lists (A != A^T: the general route with V = A^T S), duplicate entries.
Values and gradients against the CPU oracle in float64 under autograd (reference poolers/mincut.py:220-237,
diffpool.py:208-218, utils/losses.py:39-70,126-201,319-359,476-483,644-708; connect/dense_conn.py:111-208)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _random_batch(seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
    n_graphs = r(1, 5)
    sizes = [r(2, 6) if r(0, 3) == 0 else r(40, 230) for _ in range(n_graphs)]
    sizes[r(0, n_graphs - 1)] = r(70, 260)  # at least one graph beyond the one-wave kernels
    k = [3, 7, 20, 33, 65, 130, 80, 112][r(0, 7)]
    f = [1, 3, 5, 17, 32][r(0, 4)]
    directed, weighted, duplicates = r(0, 1) == 1, r(0, 2) > 0, r(0, 3) == 0
    xs, eis, bs, off = [], [], [], 0
    for gi, n in enumerate(sizes):
        a = torch.rand(n, n, generator=g) < min(6.0 / n, 0.6)
        a.fill_diagonal_(False)
        if not directed:
            a = torch.triu(a, 1)
            a = a | a.t()
        if n > 4:
            a[n // 2, :] = False  # an isolated node (no row, no column)
            a[:, n // 2] = False
        if not a.any():
            a[0, 1] = a[1, 0] = True
        e = a.nonzero().t()
        if duplicates and e.size(1) > 2:
            e = torch.cat([e, e[:, : max(1, e.size(1) // 7)]], 1)
            e = e[:, torch.argsort(e[0] * n + e[1], stable=True)]
        eis.append(e + off)
        xs.append(torch.randn(n, f, generator=g))
        bs.append(torch.full((n,), gi))
        off += n
    x, ei, batch = torch.cat(xs), torch.cat(eis, 1), torch.cat(bs)
    ew = (torch.rand(ei.size(1), generator=g) + 0.25) if weighted else None
    return sizes, k, f, x, ei, ew, batch


def _check(dev, alias, seed, oracle_alias=None, node=None, made=None):
    import tgp_oracle as O
    from tgp.poolers import get_pooler
    sizes, k, f, x, ei, ew, batch = made or _random_batch(seed)
    what = f"seed {seed}: sizes {sizes} K {k} F {f} E {ei.size(1)} weighted {ew is not None}"
    torch.manual_seed(seed)
    pooler = get_pooler(alias, in_channels=f, k=k).to(dev).train()
    lin = pooler.selector.mlp.lins[0]
    xg = x.to(dev).requires_grad_(True)
    out = pooler(x=xg, adj=ei.to(dev), edge_weight=None if ew is None else ew.to(dev), batch=batch.to(dev))
    if node is not None:
        seen, stack, names = set(), [out.x.grad_fn], []
        while stack:
            fn = stack.pop()
            if fn is None or fn in seen:
                continue
            seen.add(fn)
            names.append(type(fn).__name__)
            stack.extend(nx for nx, _ in fn.next_functions)
        assert any(node in n for n in names), (what, names)
    g = torch.Generator().manual_seed(seed + 1)
    wx, wa = torch.randn(out.x.shape, generator=g), torch.randn(out.edge_index.shape, generator=g)
    l1, l2 = list(out.loss.values())
    (((out.x * wx.to(dev)).sum() + (out.edge_index * wa.to(dev)).sum()) + 0.7 * l1 + 1.3 * l2).backward()

    xr = x.double().requires_grad_(True)
    wr = lin.weight.detach().cpu().double().requires_grad_(True)
    br = lin.bias.detach().cpu().double().requires_grad_(True)
    ref = O.dense_pool(oracle_alias or alias, xr, ei, (torch.ones(ei.size(1)) if ew is None else ew).double(), batch,
                       [wr], [br], **({"batched": False} if alias.endswith("_u") else {}))
    r1, r2 = list(ref["loss"].values())
    (((ref["x"] * wx.double()).sum() + (ref["edge_index"] * wa.double()).sum()) + 0.7 * r1 + 1.3 * r2).backward()

    def close(got, want, name, rtol, atol):
        scale = max(float(want.detach().abs().max()), 1e-3)
        torch.testing.assert_close(got.detach().cpu().double(), want.detach(), rtol=rtol, atol=atol * scale,
                                   msg=lambda m: f"{what}: {name}: {m}")

    close(out.x, ref["x"], "x_pool", 1e-5, 1e-5)
    close(out.edge_index, ref["edge_index"], "adj_pool", 1e-5, 1e-5)
    close(l1, r1, "loss 1", 1e-5, 1e-5)
    close(l2, r2, "loss 2", 1e-5, 1e-5)
    close(xg.grad, xr.grad, "dX", 3e-4, 3e-5)
    close(lin.weight.grad, wr.grad, "dW", 3e-4, 3e-5)
    close(lin.bias.grad, br.grad, "db", 3e-4, 3e-5)


@pytest.mark.parametrize("seed", list(range(100, 112)))
@pytest.mark.parametrize("alias", ["mincut", "diff"])
def test_fuzz_densifying_route(dev, monkeypatch, alias, seed):
    import tgp.poolers as P
    monkeypatch.setattr(P, "_ROWS_ROUTE_DENSITY", 0.0)
    _check(dev, alias, seed, node="_PoolLargeFn")


@pytest.mark.parametrize("seed", list(range(200, 212)))
@pytest.mark.parametrize("alias", ["mincut", "diff"])
def test_fuzz_rows_route(dev, monkeypatch, alias, seed):
    import tgp.poolers as P
    monkeypatch.setattr(P, "_ROWS_ROUTE_DENSITY", 2.0)  # every sparse input takes the rows route
    _check(dev, alias, seed, node="_PoolUnbatchedFn")


@pytest.mark.parametrize("seed", list(range(300, 312)))
@pytest.mark.parametrize("alias", ["mincut_u", "diff_u"])
def test_fuzz_unbatched_poolers(dev, alias, seed):
    _check(dev, alias, seed, oracle_alias=alias[:-2])


def _edgeless_batch(sizes, edgeless, f, k, seed):
    g = torch.Generator().manual_seed(seed)
    xs, eis, bs, off = [], [], [], 0
    for gi, n in enumerate(sizes):
        a = torch.triu(torch.rand(n, n, generator=g) < 6.0 / n, 1)
        a = a | a.t()
        if gi in edgeless:
            a[:] = False
        eis.append(a.nonzero().t() + off)
        xs.append(torch.randn(n, f, generator=g))
        bs.append(torch.full((n,), gi))
        off += n
    return sizes, k, f, torch.cat(xs), torch.cat(eis, 1), None, torch.cat(bs)


@pytest.mark.parametrize("sizes,edgeless", [([90, 70, 120], {1}), ([90, 70], {0, 1}), ([90, 70, 120], {0}), ([90, 70, 120], {2})])
@pytest.mark.parametrize("alias", ["mincut", "diff", "mincut_u", "diff_u"])
@pytest.mark.parametrize("density", [0.0, 2.0])
def test_graphs_without_edges(dev, monkeypatch, alias, sizes, edgeless, density):
    """A graph without a single edge inside a batch, and a batch without any edge (E = 0): zero degree vectors, zero
    S^T A S, MinCut's 0 / (0 + eps); both routes of the batched poolers and the unbatched ones, values and gradients."""
    import tgp.poolers as P
    monkeypatch.setattr(P, "_ROWS_ROUTE_DENSITY", density)
    _check(dev, alias, 7, oracle_alias=alias.replace("_u", ""), made=_edgeless_batch(sizes, edgeless, 8, 12, 3))
