"""Whole pooler forwards when every call sees NEW batch / edge_index tensor objects (a DataLoader's mini-batches): the
per-tensor memos (batch facts, row-sortedness, TopK plan) miss on every call.  Compare with tools/e2e_launches.py, which
pools the same tensors again and again (full-batch training).

    python tools/e2e_fresh_batch.py [case ...]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from e2e_launches import CASES, batch_graphs, resolve_sizes  # noqa: E402
from tgp.poolers import get_pooler  # noqa: E402

dev = torch.device("cuda:0")


def wall(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name in ([a for a in sys.argv[1:] if a in CASES] or ["mincut_c3", "diff_c3", "topk_c3", "graclus_c3", "ndp_c3"]):
    alias, kw, sizes, deg, f = CASES[name]
    sizes = resolve_sizes(sizes)
    x, ei, batch = batch_graphs(sizes, deg, f)
    pooler = get_pooler(alias, **kw).to(dev).eval()

    def same():
        with torch.no_grad():
            pooler(x=x, adj=ei, batch=batch)

    def fresh():
        with torch.no_grad():
            pooler(x=x, adj=ei.clone(), batch=batch.clone())

    def clones_only():
        ei.clone(), batch.clone()

    a, b, c = wall(same), wall(fresh), wall(clones_only)
    print(f"{name:12s} same tensors {a:7.3f} ms   new tensor objects per call {b - c:7.3f} ms (clones excluded)", flush=True)
