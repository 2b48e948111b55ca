"""Unbatched dense poolers (mincut_u / diff_u: S [Ntot,K], sparse A) outside autograd, r6: Reduce, Connect and both
losses from one S^T [A S | X | S] product behind the CSR SpMM (poolers._unbatched_fused) against the CPU oracle's
restatement of the reference's per-edge / per-graph forms (utils/losses.py:73-127, 204-240, 661-708;
connect/dense_conn.py:140-208; reduce/base_reduce.py:170-182), and batched == unbatched (reference
tests/poolers/test_dense_poolers_batched_vs_unbatched.py:80-174)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _batch(sizes, f, deg, seed, weighted=True, duplicates=False):
    g = torch.Generator().manual_seed(seed)
    xs, eis, bs, off = [], [], [], 0
    for gi, n in enumerate(sizes):
        a = torch.triu(torch.rand(n, n, generator=g) < deg / n, 1)
        a = a | a.t()
        e = a.nonzero().t() + off
        if duplicates:  # every fifth entry twice: the Connect sums duplicates, the link loss' sum_e w_e^2 does not
            e = torch.cat([e, e[:, ::5]], 1)
        eis.append(e)
        xs.append(torch.randn(n, f, generator=g))
        bs.append(torch.full((n,), gi))
        off += n
    x, ei, batch = torch.cat(xs), torch.cat(eis, 1), torch.cat(bs)
    ew = (torch.rand(ei.size(1), generator=g) + 0.25) if weighted else None
    return x, ei, ew, batch


def _launches(fn):
    from torch.profiler import ProfilerActivity, profile
    fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    return sum(1 for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA)


CASES = [  # alias, graph sizes, K, F, weighted, duplicates
    ("mincut_u", [40, 25, 33], 8, 16, True, False),
    ("mincut_u", [130, 97, 160], 40, 24, False, False),
    ("mincut_u", [70, 90], 72, 12, True, True),
    ("diff_u", [40, 25, 33], 8, 16, True, False),
    ("diff_u", [130, 97, 160], 40, 24, False, False),
    ("diff_u", [70, 90], 72, 12, True, True),
    ("mincut_u", [77], 10, 6, True, False),   # one graph (batch given)
    ("diff_u", [77], 10, 6, False, True),
]


@pytest.mark.parametrize("alias,sizes,k,f,weighted,duplicates", CASES)
def test_unbatched_forward_matches_oracle(dev, alias, sizes, k, f, weighted, duplicates):
    import tgp_oracle as O
    from tgp.poolers import get_pooler
    x, ei, ew, batch = _batch(sizes, f, 6.0, seed=sum(sizes) + k, weighted=weighted, duplicates=duplicates)
    pooler = get_pooler(alias, in_channels=f, k=k).to(dev).eval()
    lin = pooler.selector.mlp.lins[0]
    args = dict(x=x.to(dev), adj=ei.to(dev), edge_weight=None if ew is None else ew.to(dev), batch=batch.to(dev))
    with torch.no_grad():
        out = pooler(**args)
        n_launch = _launches(lambda: pooler(**args))
    ref = O.dense_pool(alias[:-2], x, ei, ew, batch, [lin.weight.detach().cpu()], [lin.bias.detach().cpu()],
                       batched=False)
    torch.testing.assert_close(out.x.cpu(), ref["x"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(out.edge_index.cpu(), ref["edge_index"], rtol=1e-5, atol=1e-5)
    assert torch.equal(out.batch.cpu(), ref["batch"])
    for name, want in ref["loss"].items():
        torch.testing.assert_close(out.loss[name].cpu(), want, rtol=2e-5, atol=1e-6, msg=lambda m: f"{name}: {m}")
    # the verdict's bar for this path: at most 12 launches per forward on a list that is already coalesced
    if not duplicates:
        assert n_launch <= 12, n_launch


@pytest.mark.parametrize("alias", ["mincut", "diff"])
def test_batched_equals_unbatched(dev, alias):
    """Reference pin 11: so.s, x, adj and the losses of the batched and the unbatched mode agree at 1e-5."""
    from tgp.poolers import get_pooler
    sizes, k, f = [50, 50, 50], 6, 9  # equal sizes: the padded batch has no padding, as in the reference's test
    x, ei, ew, batch = _batch(sizes, f, 5.0, seed=4, weighted=False)
    torch.manual_seed(0)
    pb = get_pooler(alias, in_channels=f, k=k, adj_transpose=False).to(dev).eval()
    pu = get_pooler(alias + "_u", in_channels=f, k=k, adj_transpose=False).to(dev).eval()
    pu.load_state_dict(pb.state_dict())
    with torch.no_grad():
        ob = pb(x=x.to(dev), adj=ei.to(dev), batch=batch.to(dev))
        ou = pu(x=x.to(dev), adj=ei.to(dev), batch=batch.to(dev))
    torch.testing.assert_close(ob.so.s.reshape(-1, k), ou.so.s, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(ob.x, ou.x, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(ob.edge_index, ou.edge_index, rtol=1e-5, atol=1e-5)
    for name in ob.loss:
        torch.testing.assert_close(ob.loss[name], ou.loss[name], rtol=1e-4, atol=1e-5, msg=lambda m: f"{name}: {m}")


def test_unbatched_training_keeps_the_differentiable_path(dev):
    """Under autograd the composed differentiable operators still take the call (the fused forward has no backward)."""
    from tgp.poolers import get_pooler
    x, ei, ew, batch = _batch([40, 30], 8, 5.0, seed=1)
    pooler = get_pooler("mincut_u", in_channels=8, k=5).to(dev).train()
    out = pooler(x=x.to(dev), adj=ei.to(dev), edge_weight=ew.to(dev), batch=batch.to(dev))
    (out.x.sum() + out.edge_index.sum() + sum(out.loss.values())).backward()
    lin = pooler.selector.mlp.lins[0]
    assert lin.weight.grad is not None and torch.isfinite(lin.weight.grad).all()
