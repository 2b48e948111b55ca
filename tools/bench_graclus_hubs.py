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
    # whole `graclus` forward (GraclusSelect + Reduce + coalesce Connect): r3 sent the whole list to the radix route for
    # one supernode row beyond 1024 entries (3.2 ms with ten hubs against 0.92 ms without); r4 sorts those rows alone
    from tgp.poolers import get_pooler
    x = torch.randn(n, 128, device=dev, generator=g)
    pooler = get_pooler("graclus").to(dev).eval()
    def fwd():
        with torch.no_grad():
            return pooler(x=x, adj=ei, edge_weight=ew)
    out = fwd()
    print(f"    whole graclus forward {wall(fwd, 5):.3f} ms, pooled edges {out.edge_index.size(1)}, "
          f"hub kernels {'on' if id(ei) in K._HUB_ROWS else 'off'}", flush=True)
