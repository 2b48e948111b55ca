"""Where the rows route's link loss (sum A^2 - 2 trace(S^T A S) + |S^T S|^2) loses digits: the pooler's value against the
same three terms evaluated in float64 from the pooler's OWN S (isolates the terms' arithmetic from S's) and against the
direct form |A - S S^T| in float64.

    python3 tools/link_loss_precision.py
"""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "torch-geometric-pool_amd"))
import tgp.poolers as P  # noqa: E402
from tgp.poolers import get_pooler  # noqa: E402

dev = torch.device("cuda:0")
P._ROWS_ROUTE_DENSITY = 2.0
for sizes, k, f, seed in (([350, 420, 380], 128, 16, 328), ([350, 420, 380], 72, 16, 372), ([1024] * 8, 128, 64, 1), ([400, 300], 40, 24, 240)):
    g = torch.Generator().manual_seed(seed)
    xs, eis, bs, off = [], [], [], 0
    for gi, n in enumerate(sizes):
        a = torch.triu(torch.rand(n, n, generator=g) < 8.0 / n, 1)
        a = a | a.t()
        eis.append(a.nonzero().t() + off)
        xs.append(torch.randn(n, f, generator=g))
        bs.append(torch.full((n,), gi))
        off += n
    x, ei, batch = torch.cat(xs).to(dev), torch.cat(eis, 1).to(dev), torch.cat(bs).to(dev)
    torch.manual_seed(seed)
    pooler = get_pooler("diff", in_channels=f, k=k).to(dev).eval()
    with torch.no_grad():
        out = pooler(x=x, adj=ei, batch=batch)
    got = float(out.loss["link_loss"])
    s = out.so.s.double()  # [B, Nmax, K], padded rows zero
    B, Nm = s.size(0), s.size(1)
    A = torch.zeros(B, Nm, Nm, dtype=torch.float64, device=dev)
    ptr = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)), device=dev)
    b = batch[ei[0]]
    A[b, ei[0] - ptr[b], ei[1] - ptr[b]] = 1.0
    direct = float(torch.linalg.norm((A - s @ s.transpose(1, 2)).reshape(-1)))
    sw2 = float(ei.size(1))
    tr = float(torch.einsum("bnk,bnm,bmk->", s, A, s))
    gsq = float(((s.transpose(1, 2) @ s) ** 2).sum())
    ident = (sw2 - 2 * tr + gsq) ** 0.5
    print(f"sizes {sizes[:3]}{'...' if len(sizes) > 3 else ''} K={k}: pooler {got:.6f}   fp64 direct from its S {direct:.6f} "
          f"(rel {abs(got - direct) / direct:.2e})   fp64 identity {ident:.6f}   terms: sum A^2 {sw2:.1f}, 2 tr {2 * tr:.3f}, |G|^2 {gsq:.3f}")
