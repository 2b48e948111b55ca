"""C-ABI entry points on empty / degenerate inputs and cross-operator behaviour (hierarchies of poolers, dtype rules).

Regrouped by operator in round 6 from the per-round files test_gpu_round2..5.py; the test bodies are unchanged."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


# ------------------------------------------------------------------------------ edge cases of the r3 entry points
def test_round3_entry_points_on_empty_and_degenerate_inputs(dev):
    """Empty / ragged / degenerate inputs the reference's own tests exercise for the older operators, for the new ones:
    zero rows, a single row, empty graphs inside a batch, lists without edges, isolated nodes."""
    import tgp_oracle as O
    from tgp import kernels as Kn
    from tgp.select import MLPSelect, NDPSelect, TopkSelect
    g = torch.Generator().manual_seed(0)
    # MLPSelect: no rows, one row
    w, b = torch.randn(5, 3, generator=g).to(dev), torch.randn(5, generator=g).to(dev)
    assert Kn.mlp_select(torch.zeros(0, 3, device=dev), w, b, None).shape == (0, 5)
    one = torch.randn(1, 3, generator=g)
    torch.testing.assert_close(Kn.mlp_select(one.to(dev), w, b, None).cpu(),
                               O.mlp_select(one, [w.cpu()], [b.cpu()]), rtol=1e-5, atol=1e-6)
    assert Kn.softmax_bwd(torch.zeros(0, 5, device=dev), torch.zeros(0, 5, device=dev)).shape == (0, 5)
    # fully masked batch: S is exactly zero
    sel = MLPSelect(in_channels=3, k=4).to(dev)
    so = sel(x=torch.randn(2, 6, 3, generator=g).to(dev), mask=torch.zeros(2, 6, dtype=torch.bool, device=dev))
    assert float(so.s.detach().abs().max()) == 0.0
    # TopkSelect min_score: graphs of one node, an empty graph id in the middle of the batch vector
    batch = torch.tensor([0, 0, 0, 2, 3, 3]).to(dev)   # graph 1 owns no node
    x = torch.randn(6, 4, generator=g)
    tk = TopkSelect(in_channels=4, ratio=None, min_score=0.4).to(dev)
    so = tk(x=x.to(dev), batch=batch)
    ni, ci, wt = O.topk_select(x, tk.weight.detach().cpu(), None, batch.cpu(), 0.4, "tanh")
    assert torch.equal(so.node_index.cpu(), ni)
    torch.testing.assert_close(so.weight.cpu(), wt, rtol=1e-5, atol=1e-7)
    # coalesce routes: no edges, one edge, every edge a self loop
    cl = torch.tensor([0, 0, 1, 1, 2]).to(dev)
    for route in ("fused", "staged", "general"):
        ei0 = torch.zeros(2, 0, dtype=torch.long, device=dev)
        out = Kn.coalesce_edges(ei0, None, cl, 3, "sum", True, route=route)
        assert out[0].shape == (2, 0)
        ei1 = torch.tensor([[1], [4]], device=dev)
        out = Kn.coalesce_edges(ei1, torch.tensor([2.0], device=dev), cl, 3, "sum", True, route=route)
        assert out[0].tolist() == [[0], [2]] and out[1].tolist() == [2.0]
        loops = torch.tensor([[0, 1, 2, 3], [1, 0, 3, 2]], device=dev)   # all inside their clusters
        out = Kn.coalesce_edges(loops, None, cl, 3, "sum", True, route=route)
        assert out[0].shape == (2, 0)
    # NDPSelect: a large graph with isolated nodes and a second component (chip-wide route), still a valid partition
    n = 3000
    a = torch.arange(0, 2000 - 1)
    ei = torch.stack([torch.cat([a, a + 1]), torch.cat([a + 1, a])]).to(dev)    # a path on nodes 0..1999, rest isolated
    so = NDPSelect()(ei, None, num_nodes=n)
    assert 0 < so.num_supernodes < n and bool((so.node_index[1:] > so.node_index[:-1]).all())
