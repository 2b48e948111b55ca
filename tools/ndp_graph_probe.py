"""One graph of the ownership test's batch under the microscope: dense spectrum of Ls against what NDPSelect kept."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
from tgp.select import NDPSelect  # noqa: E402


def _er_batch(num_graphs, lo, hi, f, seed, dev):
    """The batch of tests/test_gpu_sparse_pool_small.py: Erdos-Renyi graphs of lo..hi nodes, ~4 neighbours per node."""
    g = torch.Generator().manual_seed(seed)
    xs, eis, bs, off = [], [], [], 0
    for gi in range(num_graphs):
        n = int(torch.randint(lo, hi + 1, (1,), generator=g))
        a = torch.triu(torch.rand(n, n, generator=g) < 4.0 / n, 1)
        a = a | a.t()
        eis.append(a.nonzero().t() + off)
        xs.append(torch.randn(n, f, generator=g))
        bs.append(torch.full((n,), gi))
        off += n
    x, ei, batch = torch.cat(xs), torch.cat(eis, 1), torch.cat(bs)
    ew = torch.rand(ei.size(1), generator=g) + 0.1
    return x.to(dev), ei.to(dev), ew.to(dev), batch.to(dev)


dev = torch.device("cuda:0")
gsel = [int(a) for a in sys.argv[1:]] or [166]
x, ei, ew, batch = _er_batch(200, 5, 60, 8, 7, dev)
torch.manual_seed(3)
so = NDPSelect()(edge_index=ei, edge_weight=ew, batch=batch, num_nodes=x.size(0))
info = so._partition_info.cpu()
keep = torch.zeros(x.size(0), dtype=torch.bool)
keep[so.node_index.cpu()] = True
n = x.size(0)
A = torch.zeros(n, n, dtype=torch.float64)
A[ei[0].cpu(), ei[1].cpu()] = ew.double().cpu()
A = torch.maximum(A, A.t())
A.fill_diagonal_(0)
for g in gsel:
    idx = (batch == g).nonzero().view(-1).cpu()
    a = A[idx][:, idx]
    m = idx.numel()
    deg = a.sum(1)
    dis = torch.where(deg > 0, deg.clamp(min=1e-300).rsqrt(), torch.zeros_like(deg))
    ls = torch.eye(m, dtype=torch.float64) - dis[:, None] * a * dis[None, :]
    vals, vecs = torch.linalg.eigh(ls)
    L = torch.diag(deg) - a
    vol = float(deg.sum())
    cut = lambda mask: float(torch.where(mask, 1.0, -1.0).double() @ (L @ torch.where(mask, 1.0, -1.0).double())) / (2 * vol)
    kp = keep[idx]
    print(f"graph {g}: {m} nodes, isolated {int((deg == 0).sum())}, info {int(info[g])}, top eigenvalues {vals[-4:].tolist()}")
    print(f"   cut of the top eigenvector's sign pattern {cut(vecs[:, -1] >= 0):.4f}; cut of what was kept {cut(kp):.4f}")
    for k in range(1, 5):
        v = vecs[:, -k]
        off = min(int((kp != (v >= 0)).sum()), int((kp != (v < 0)).sum()))
        print(f"   kept vs sign pattern of eigenvector #{k} (lambda {float(vals[-k]):.6f}): {off} nodes off; min |v| {float(v.abs().min()):.1e}")
