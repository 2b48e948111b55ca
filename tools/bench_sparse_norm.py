"""A6 at scale: degree / max-abs normalisation of a pooled edge list (utils/ops.py:383-417) on E = 10 M sorted edges."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
from tgp.utils.ops import postprocess_adj_pool_sparse  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
n = 550_000
a = torch.randint(0, n, (5_000_000,), device=dev, generator=g)
b = torch.randint(0, n, (5_000_000,), device=dev, generator=g)
ei = torch.stack([torch.cat([a, b]), torch.cat([b, a])])
ei = ei[:, torch.argsort(ei[0] * n + ei[1])]
ew = torch.rand(ei.size(1), device=dev, generator=g) + 0.1
bp = torch.sort(torch.randint(0, 8, (n,), device=dev, generator=g))[0]


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


E = ei.size(1)
for name, kw in (("filter only (self loops + eps)", dict(remove_self_loops=True)),
                 ("+ degree_norm", dict(remove_self_loops=True, degree_norm=True)),
                 ("+ edge_weight_norm", dict(remove_self_loops=True, edge_weight_norm=True, batch_pooled=bp)),
                 ("+ both", dict(remove_self_loops=True, degree_norm=True, edge_weight_norm=True, batch_pooled=bp))):
    us = timed(lambda: postprocess_adj_pool_sparse(ei, ew, n, **kw))
    print(f"{name:34s} {us:8.1f} us   ({E * 20 * 2 / us / 1e6:.2f} TB/s on read+write of the list)")
