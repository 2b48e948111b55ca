"""End-to-end pooler calls (Select + Reduce + Connect [+ losses, + backward]) at scale: wall time per call.
Finds host overhead / syncs / slow helper ops around the native kernels."""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
from tgp.poolers import get_pooler  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
only = sys.argv[1:] or None


def wall(fn, iters=10):
    for _ in range(3):
        fn()
    gc.collect()  # keep CPython's full collections (~40 ms with torch imported) out of a 10-iteration window
    gc.freeze()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def sparse_graph(n, deg, graphs):
    src = torch.arange(n, device=dev).repeat_interleave(deg // 2)
    per = n // graphs
    dst = (src // per) * per + torch.randint(0, per, (src.numel(),), device=dev, generator=g)
    ei = torch.cat([torch.stack([src, dst]), torch.stack([dst, src])], 1)
    key = torch.unique(ei[0] * n + ei[1])  # sorted, duplicate-free: what PyG datasets hand out
    ei = torch.stack([key // n, key % n])
    batch = torch.arange(n, device=dev) // per
    return ei, batch


def run(name, pooler, x, ei, ew, batch, train):
    pooler = pooler.to(dev)
    params = [p for p in pooler.parameters()]

    def step():
        xx = x.detach().requires_grad_(train)
        if train:
            out = pooler(x=xx, adj=ei, edge_weight=ew, batch=batch)
            loss = out.x.sum()
            if out.loss:
                loss = loss + sum(out.loss.values())
            loss.backward()
            for p in params:
                p.grad = None
        else:
            with torch.no_grad():
                pooler(x=xx, adj=ei, edge_weight=ew, batch=batch)
    print(f"{name:58s} {'fwd+bwd' if train else 'fwd    '} {wall(step):9.3f} ms", flush=True)


cases = []
n, F = 1_000_000, 128
ei, batch = sparse_graph(n, 10, 8)
x = torch.randn(n, F, device=dev, generator=g)
ew = ((ei[0] * ei[1] + ei[0] + ei[1]) % 1000).float() / 1000 + 0.1  # symmetric: w(i,j) = w(j,i), an undirected graph
cases.append(("topk N=1M E=10M F=128", lambda: get_pooler("topk", in_channels=F, ratio=0.5), x, ei, ew, batch))
cases.append(("graclus N=1M E=10M F=128", lambda: get_pooler("graclus"), x, ei, ew, batch))
n2, F2 = 32 * 1024, 64
ei2, batch2 = sparse_graph(n2, 16, 32)
x2 = torch.randn(n2, F2, device=dev, generator=g)
for nm in ("diff", "mincut", "diff_u", "mincut_u"):
    cases.append((f"{nm} B=32 N=1024 K=128 F=64 (sparse input)", (lambda nm=nm: get_pooler(nm, in_channels=F2, k=128)),
                  x2, ei2, None, batch2))
# PROTEINS-like batch: 2048 graphs of 20..60 nodes, K=20, F=32 (BASELINE configs[2] shape)
sizes3 = torch.randint(20, 61, (2048,), device=dev, generator=g)
batch3 = torch.repeat_interleave(torch.arange(2048, device=dev), sizes3)
ptr3 = torch.cat([sizes3.new_zeros(1), sizes3.cumsum(0)])
n3 = batch3.numel()
src3 = torch.arange(n3, device=dev).repeat_interleave(2)
dst3 = ptr3[batch3[src3]] + (torch.rand(src3.numel(), device=dev, generator=g) * sizes3[batch3[src3]]).long()
key3 = torch.unique(torch.cat([src3 * n3 + dst3, dst3 * n3 + src3]))
ei3 = torch.stack([key3 // n3, key3 % n3])
x3 = torch.randn(n3, 32, device=dev, generator=g)
for nm in ("mincut", "diff", "mincut_u"):
    cases.append((f"{nm} small graphs B=2048 n~40 K=20 F=32", (lambda nm=nm: get_pooler(nm, in_channels=32, k=20)),
                  x3, ei3, None, batch3))
cases.append(("topk small graphs B=2048 n~40 F=32", lambda: get_pooler("topk", in_channels=32, ratio=0.5), x3, ei3, None, batch3))
cases.append(("graclus small graphs B=2048 n~40 F=32", lambda: get_pooler("graclus"), x3, ei3, None, batch3))
# NDP: spectral partition (tgp_ndp_partition) + Reduce + block-batched Kron reduction, all on the device; and its
# two-level precoarsening (what NDP is normally used for: a transform applied once per dataset)
cases.append(("ndp small graphs B=2048 n~40 F=32", lambda: get_pooler("ndp"), x3, ei3, None, batch3))
if not only or any("precoarsen" in o for o in only):
    ndp = get_pooler("ndp").to(dev)
    ms = wall(lambda: ndp.multi_level_precoarsening(levels=2, edge_index=ei3, batch=batch3, num_nodes=n3), iters=5)
    print(f"{'ndp 2-level precoarsening, small graphs B=2048 n~40':58s} {'fwd    '} {ms:9.3f} ms", flush=True)
for name, mk, xx, e, w, b in cases:
    if only and not any(o in name for o in only):
        continue
    for train in ((False, True) if not os.environ.get("E2E_MODE") else (os.environ["E2E_MODE"] == "train",)):
        try:
            run(name, mk(), xx, e, w, b, train)
        except Exception as ex:  # noqa: BLE001
            print(f"{name:58s} {'fwd+bwd' if train else 'fwd    '} FAILED: {type(ex).__name__}: {ex}", flush=True)
