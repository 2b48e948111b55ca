#!/usr/bin/env python3
"""Launch one hot kernel repeatedly (for rocprofv3 --pmc / --kernel-trace runs).
usage: python tools/run_kernel.py {gemm_c2|gemm_c5|reduce_topk|reduce_ndp|reduce_graclus|coalesce_c4|coalesce_c4_sorted|
                                   subgraph_topk|c3|topk_batch|graclus_batch} [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-geometric-pool_amd"))
from tgp import kernels  # noqa: E402
from tgp.select import SelectOutput  # noqa: E402

def marker():
    """A kernel no measured call launches, right before the measured repetitions (tools/pmc_summary.py cuts there)."""
    torch.cuda.synchronize()
    kernels.entropy_sum(torch.ones(8, 8, device="cuda:0"))


which = sys.argv[1] if len(sys.argv) > 1 else "gemm_c2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
if which.startswith("gemm"):
    B, N, K = (32, 1024, 128) if which == "gemm_c2" else (2, 8192, 512)
    A = (torch.rand(B, N, N, device=dev, generator=g) < 0.01).float()
    S = torch.softmax(torch.randn(B, N, K, device=dev, generator=g), -1)
    marker()
    for _ in range(reps):
        kernels.bmm(A, S)
elif which == "reduce_topk":
    n, f = 1_000_000, 128
    x = torch.randn(n, f, device=dev, generator=g)
    keep = torch.sort(torch.randperm(n, device=dev, generator=g)[: n // 2])[0]
    k = keep.numel()
    so = SelectOutput(node_index=keep, num_nodes=n, cluster_index=torch.randperm(k, device=dev, generator=g),
                      num_supernodes=k, weight=torch.rand(k, device=dev, generator=g))
    so._set_one_to_one_index()  # as TopkSelect attaches it
    idx = so.assign_index()
    marker()
    for _ in range(reps):
        kernels.reduce_sparse(x, so.node_index, so.weight, idx)
elif which in ("reduce_ndp", "reduce_graclus"):
    n, f = 1_000_000, 128
    x = torch.randn(n, f, device=dev, generator=g)
    if which == "reduce_ndp":
        keep = (torch.rand(n, device=dev, generator=g) < 0.5).nonzero().view(-1)
        so = SelectOutput(node_index=keep, num_nodes=n, cluster_index=torch.arange(keep.numel(), device=dev),
                          num_supernodes=keep.numel())
        so._set_one_to_one_index()
    else:
        pair = torch.randperm(n, device=dev, generator=g)
        cluster = torch.empty(n, dtype=torch.long, device=dev)
        cluster[pair] = torch.arange(n, device=dev) // 2
        so = SelectOutput(cluster_index=cluster, num_nodes=n, num_supernodes=n // 2)
    idx = so.assign_index()
    marker()
    for _ in range(reps):
        kernels.reduce_sparse(x, so.node_index, so.weight, idx)
elif which in ("topk_batch", "graclus_batch"):  # bench.py's batched sparse workloads: the one-launch kernel
    sys.path.insert(0, ROOT)
    import bench
    wl = bench.TopkBatch(bench.Ctx(dev, 0, 1, None), which=which)
    wl.compute()
    marker()
    for _ in range(reps):
        wl.compute()
elif which == "coalesce_c4":
    n = 1_000_000
    a = torch.randint(0, n, (5_000_000,), device=dev, generator=g)
    b = torch.randint(0, n, (5_000_000,), device=dev, generator=g)
    ei = torch.stack([torch.cat([a, b]), torch.cat([b, a])])
    ew = torch.ones(ei.size(1), device=dev)
    pair = torch.randperm(n, device=dev, generator=g)
    cluster = torch.empty(n, dtype=torch.long, device=dev)
    cluster[pair] = torch.arange(n, device=dev) // 2
    marker()
    for _ in range(reps):
        kernels.coalesce_edges(ei, ew, cluster, n // 2, "sum", True)
elif which == "coalesce_c4_sorted":  # bench.py's c4_graclus Connect: row-sorted edges, Graclus assignment, index cached
    from tgp.connect import SparseConnect
    from tgp.select import GraclusSelect
    n = 1_000_000
    a = torch.randint(0, n, (5_000_000,), device=dev, generator=g)
    b = torch.randint(0, n, (5_000_000,), device=dev, generator=g)
    keep = a != b
    a, b = a[keep], b[keep]
    ei = torch.stack([torch.cat([a, b]), torch.cat([b, a])])
    ei = ei[:, torch.argsort(ei[0] * n + ei[1])].contiguous()
    ew = torch.ones(ei.size(1), device=dev)
    so = GraclusSelect()(ei, ew, num_nodes=n)
    conn = SparseConnect()
    marker()
    for _ in range(reps):
        conn(ei, so, edge_weight=ew)
elif which == "subgraph_topk":
    from tgp.connect import SparseConnect
    n = 1_000_000
    a = torch.randint(0, n, (5_000_000,), device=dev, generator=g)
    b = torch.randint(0, n, (5_000_000,), device=dev, generator=g)
    keep = a != b
    a, b = a[keep], b[keep]
    ei = torch.stack([torch.cat([a, b]), torch.cat([b, a])])
    ei = ei[:, torch.argsort(ei[0] * n + ei[1])].contiguous()
    ew = torch.ones(ei.size(1), device=dev)
    # (as bench.py's topk_connect: the SelectOutput of the real selector, with the bitmap + rank directory it attaches)
    from tgp.select import TopkSelect
    torch.manual_seed(0)
    with torch.no_grad():
        so = TopkSelect(in_channels=8, ratio=0.5).to(dev)(x=torch.randn(n, 8, device=dev, generator=g))
    conn = SparseConnect()
    marker()
    for _ in range(reps):
        conn(ei, so, edge_weight=ew)
elif which == "c3":
    B, N, K, F = 2048, 60, 20, 32
    S = torch.softmax(torch.randn(B, N, K, device=dev, generator=g), -1)
    A = (torch.rand(B, N, N, device=dev, generator=g) < 0.1).float()
    X = torch.randn(B, N, F, device=dev, generator=g)
    marker()
    for _ in range(reps):
        kernels.dense_pool(S, A, X, kernels.dense_flags(True, True, True, False), want_raw=True)
torch.cuda.synchronize()
print("done", which, reps)
