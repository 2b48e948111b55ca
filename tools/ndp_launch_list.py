"""Kernels of one NDP pooler forward on the SAME tensor objects (the memos of the package hit), in launch order."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from e2e_launches import CASES, batch_graphs, resolve_sizes, count_kernels  # noqa: E402
from tgp.poolers import get_pooler  # noqa: E402

dev = torch.device("cuda:0")
alias, kw, sizes, deg, f = CASES["ndp_c3"]
x, ei, batch = batch_graphs(resolve_sizes(sizes), deg, f)
pooler = get_pooler(alias, **kw).to(dev).eval()


def same():
    with torch.no_grad():
        pooler(x=x, adj=ei, batch=batch)


for _ in range(5):
    same()
count_kernels(same, list_kernels=True)
