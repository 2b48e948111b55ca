"""Dense pooler forward on pre-batched inputs captured in a HIP graph (torch.cuda.CUDAGraph): the native entry points
neither synchronise nor allocate outside torch's allocator, so a whole pooler call replays as one graph launch."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
from tgp.poolers import get_pooler  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def wall(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


for alias, B, N, K, F in (("mincut", 2048, 60, 20, 32), ("diff", 2048, 60, 20, 32), ("diff", 32, 1024, 128, 64),
                          ("mincut", 512, 126, 32, 64)):
    pooler = get_pooler(alias, in_channels=F, k=K).to(dev).eval()
    x = torch.randn(B, N, F, device=dev, generator=g)
    adj = (torch.rand(B, N, N, device=dev, generator=g) < 0.1).float()
    adj = torch.maximum(adj, adj.transpose(1, 2)).contiguous()
    with torch.no_grad():
        eager = wall(lambda: pooler(x=x, adj=adj))
        ref = pooler(x=x, adj=adj)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                pooler(x=x, adj=adj)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = pooler(x=x, adj=adj)
        replay = wall(graph.replay)
        graph.replay()
        torch.cuda.synchronize()
        ok = torch.allclose(out.x, ref.x, rtol=1e-5, atol=1e-5) and torch.allclose(out.edge_index, ref.edge_index, rtol=1e-5, atol=1e-5)
        ok = ok and all(torch.allclose(out.loss[k], ref.loss[k], rtol=1e-4, atol=1e-6) for k in ref.loss)
    print(f"{alias:7s} B={B:5d} N={N:5d} K={K:4d} F={F:3d}: eager {eager:7.3f} ms   graph replay {replay:7.3f} ms   same={ok}", flush=True)
