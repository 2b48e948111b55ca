"""Diagnostic: per-round kernel times of the native Graclus matching (run under rocprofv3 --kernel-trace)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
from tgp import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
n, graphs = 1_000_000, 8
src = torch.arange(n, device=dev).repeat_interleave(5)
per = n // graphs
dst = (src // per) * per + torch.randint(0, per, (src.numel(),), device=dev, generator=g)
key = torch.unique(torch.cat([src * n + dst, dst * n + src]))
ei = torch.stack([key // n, key % n])
for _ in range(3):
    label = K.graclus_match(ei, None, n)
torch.cuda.synchronize()
print("matched nodes", int((torch.bincount(label, minlength=n)[label] == 2).sum()), "of", n)
