"""Graclus matching on a 1M-node graph with a few very long rows (hubs of a power-law graph): rows beyond 256 entries are
scanned by whole waves that take them from a list.  python tools/bench_graclus_hubs.py"""
import sys, time, torch
sys.path.insert(0, "torch-geometric-pool_amd"); sys.path.insert(0, "tools")
from tgp import kernels as K
from e2e_launches import wall
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
n = 1_000_000
a = torch.randint(0, n, (3 * n,), device=dev, generator=g); b = torch.randint(0, n, (3 * n,), device=dev, generator=g)
for hubs, deg in ((0, 0), (10, 100_000), (100, 10_000)):
    aa, bb = a, b
    if hubs:
        h = torch.arange(hubs, device=dev).repeat_interleave(deg)
        t = torch.randint(hubs, n, (hubs * deg,), device=dev, generator=g)
        aa, bb = torch.cat([a, h]), torch.cat([b, t])
    keep = aa != bb
    aa, bb = aa[keep], bb[keep]
    key = torch.unique(torch.cat([aa * n + bb, bb * n + aa]))
    ei = torch.stack([key // n, key % n])
    ew = torch.ones(ei.size(1), device=dev)
    f = lambda: K.graclus_match(ei, ew, n)
    print(f"hubs {hubs} x deg {deg}: E = {ei.size(1)}, graclus_match {wall(f, 5):.3f} ms", flush=True)
