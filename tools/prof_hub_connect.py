#!/usr/bin/env python3
"""The coalesce Connect on a 1M-node graph with ten 100 000-entry hub rows, a few calls (for rocprofv3 --kernel-trace)."""
import sys
import torch
sys.path.insert(0, "torch-geometric-pool_amd")
from tgp.connect import SparseConnect
from tgp.select import GraclusSelect
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
n, hubs, deg = 1_000_000, 10, 100_000
a = torch.randint(0, n, (3 * n,), device=dev, generator=g); b = torch.randint(0, n, (3 * n,), device=dev, generator=g)
h = torch.arange(hubs, device=dev).repeat_interleave(deg)
t = torch.randint(hubs, n, (hubs * deg,), device=dev, generator=g)
aa, bb = torch.cat([a, h]), torch.cat([b, t])
keep = aa != bb
aa, bb = aa[keep], bb[keep]
key = torch.unique(torch.cat([aa * n + bb, bb * n + aa]))
ei = torch.stack([key // n, key % n])
ew = torch.ones(ei.size(1), device=dev)
so = GraclusSelect()(ei, ew, num_nodes=n)
conn = SparseConnect()
for _ in range(6):
    out = conn(ei, so, edge_weight=ew)
torch.cuda.synchronize()
print("edges out", out[0].size(1))
