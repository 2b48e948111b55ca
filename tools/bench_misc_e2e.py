"""Less common call patterns at scale: sparse_output dense poolers, lifting, multi-level precoarsening."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
from tgp.poolers import get_pooler  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def wall(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def sparse_graph(n, deg, graphs):
    src = torch.arange(n, device=dev).repeat_interleave(deg // 2)
    per = n // graphs
    dst = (src // per) * per + torch.randint(0, per, (src.numel(),), device=dev, generator=g)
    key = torch.unique(torch.cat([src * n + dst, dst * n + src]))
    return torch.stack([key // n, key % n]), torch.arange(n, device=dev) // per


with torch.no_grad():
    n, F = 32 * 1024, 64
    ei, batch = sparse_graph(n, 16, 32)
    x = torch.randn(n, F, device=dev, generator=g)
    for alias in ("diff", "mincut", "diff_u"):
        p = get_pooler(alias, in_channels=F, k=128, sparse_output=True).to(dev).eval()
        print(f"{alias:9s} sparse_output=True B=32 N=1024 K=128        fwd {wall(lambda: p(x=x, adj=ei, batch=batch)):8.3f} ms", flush=True)
        out = p(x=x, adj=ei, batch=batch)
        print(f"{alias:9s} lifting (x_pool -> nodes)                   fwd {wall(lambda: p(x=out.x, so=out.so, batch=batch, batch_pooled=out.batch, lifting=True)):8.3f} ms", flush=True)
    n, F = 1_000_000, 128
    ei, batch = sparse_graph(n, 10, 8)
    x = torch.randn(n, F, device=dev, generator=g)
    for alias, kw in (("topk", dict(in_channels=F, ratio=0.5)), ("graclus", {})):
        p = get_pooler(alias, **kw).to(dev).eval()
        out = p(x=x, adj=ei, batch=batch)
        print(f"{alias:9s} lifting N=1M F=128                          fwd {wall(lambda: p(x=out.x, so=out.so, lifting=True)):8.3f} ms", flush=True)
    p = get_pooler("graclus")
    print(f"graclus   multi_level_precoarsening(2) N=1M E=10M          {wall(lambda: p.multi_level_precoarsening(2, edge_index=ei, batch=batch, num_nodes=n), iters=5):8.3f} ms", flush=True)
