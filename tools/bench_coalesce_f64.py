"""Coalesce Connect at C4 (row-sorted, N = 1 M, E = 10 M, GraclusSelect's assignment) with float32 and float64 edge weights:
the row-local pipeline against the sort-based float64 route it replaced (r5).   python tools/bench_coalesce_f64.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
from tgp import kernels as K
from tgp.select import GraclusSelect
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
n = 1_000_000
a = torch.randint(0, n, (5_000_000,), device=dev, generator=g); b = torch.randint(0, n, (5_000_000,), device=dev, generator=g)
keep = a != b; a, b = a[keep], b[keep]
ei = torch.stack([torch.cat([a, b]), torch.cat([b, a])]); ei = ei[:, torch.argsort(ei[0] * n + ei[1])].contiguous()
ew = torch.rand(ei.size(1), device=dev, generator=g) + 0.5
so = GraclusSelect()(ei, ew, num_nodes=n)
k = so.num_supernodes; idx = so.assign_index(); csr = (so.edge_csr_for(ei), None)
ew64 = ew.double()
def timed(fn):
    for _ in range(3): fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t0.record()
    for _ in range(20): fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / 20 * 1e3
print("fp32, row-local route, us per call:   ", timed(lambda: K.coalesce_edges(ei, ew, so.cluster_index, k, "sum", True, assign_index=idx, csr=csr)))
print("fp64, row-local route (r5):           ", timed(lambda: K.coalesce_edges(ei, ew64, so.cluster_index, k, "sum", True, assign_index=idx, csr=csr)))
print("fp64, sort-based route (r4):          ", timed(lambda: K.coalesce_edges(ei, ew64, so.cluster_index, k, "sum", True, route="general")))
r = K.coalesce_edges(ei, ew64, so.cluster_index, k, "sum", True, assign_index=idx, csr=csr)
q = K.coalesce_edges(ei, ew64, so.cluster_index, k, "sum", True, route="general")
print("same edges / largest weight difference:", torch.equal(r[0], q[0]), float((r[1] - q[1]).abs().max()))
