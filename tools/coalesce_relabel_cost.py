"""Verdict r5 item 4, the measured half: what it costs to PRODUCE the pre-relabelled column array `cluster[col]` that a
look-up-free `cr_gather_sort_kernel` would read, at C4 (row-sorted, N = 1 M, E = 10 M, Graclus assignment).  The gain
side is the `no_table` row of profiles/r05_coalesce_variants.md (the gather-sort kernel with NO look-up at all).

    python3 tools/coalesce_relabel_cost.py
"""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "torch-geometric-pool_amd"))
from tgp import kernels  # noqa: E402
from tgp.select import GraclusSelect  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
n = 1_000_000
a = torch.randint(0, n, (5_000_000,), device=dev, generator=g)
b = torch.randint(0, n, (5_000_000,), device=dev, generator=g)
keep = a != b
a, b = a[keep], b[keep]
ei = torch.stack([torch.cat([a, b]), torch.cat([b, a])])
ei = ei[:, torch.argsort(ei[0] * n + ei[1])].contiguous()
ew = torch.rand(ei.size(1), device=dev, generator=g) + 0.5
so = GraclusSelect()(ei, ew, num_nodes=n)
k = so.num_supernodes
cluster = so.cluster_index
cluster32 = cluster.to(torch.int32)
col = ei[1]
col32 = col.to(torch.int32)
E = ei.size(1)


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps * 1e3


out32 = torch.empty(E, dtype=torch.int32, device=dev)
out64 = torch.empty(E, dtype=torch.int64, device=dev)
idx = so.assign_index()
csr = (so.edge_csr_for(ei), None)
rows = [
    ("Connect call as shipped (look-ups inside cr_gather_sort_kernel)",
     timed(lambda: kernels.coalesce_edges(ei, ew, cluster, k, "sum", True, assign_index=idx, csr=csr))),
    ("relabel pass: int32 table [N] gathered by int64 columns -> int32 [E] (torch.index_select, out=)",
     timed(lambda: torch.index_select(cluster32, 0, col, out=out32))),
    ("relabel pass: int64 table gathered by int64 columns -> int64 [E]",
     timed(lambda: torch.index_select(cluster, 0, col, out=out64))),
    ("streaming floor of that pass: int64 [E] -> int32 [E] conversion (80 MB read + 40 MB written, no gather)",
     timed(lambda: out32.copy_(col))),
]
print(f"C4: N = {n}, E = {E}, K = {k}")
for name, us in rows:
    print(f"{us:9.1f} us   {name}")
