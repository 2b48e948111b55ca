"""Where the un-padded rows route of the batched dense poolers stops paying against the densifying route: whole MinCut
forward and training step (device time between events, 40 calls) on batches of equal-size graphs at several densities,
both routes forced in turn.

    python3 tools/rows_route_crossover.py
"""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "torch-geometric-pool_amd"))
import tgp.poolers as P  # noqa: E402
from tgp.poolers import get_pooler  # noqa: E402

dev = torch.device("cuda:0")


def batch(B, n, deg, f, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    rows, cols = [], []
    for b in range(B):
        a = torch.rand(n, n, device=dev, generator=g) < deg / (2.0 * n)
        a = a | a.t()
        a.fill_diagonal_(False)
        e = a.nonzero().t()
        rows.append(e[0] + b * n)
        cols.append(e[1] + b * n)
    ei = torch.stack([torch.cat(rows), torch.cat(cols)])
    return torch.randn(B * n, f, device=dev, generator=g), ei, torch.arange(B, device=dev).repeat_interleave(n)


def timed(fn, reps=40):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps


print(f"{'shape':34s} {'density':>8s} | {'fwd dense':>9s} {'fwd rows':>9s} | {'step dense':>10s} {'step rows':>9s}   (ms)")
for B, n, k, f, degs in ((32, 1024, 128, 64, (30, 60, 100, 160)), (512, 150, 32, 32, (8, 16, 32, 48)),
                         (64, 400, 64, 32, (16, 32, 64, 100)), (8, 4096, 256, 64, (120, 240)), (16, 2048, 128, 64, (60, 120, 240)),
                         (128, 512, 32, 32, (16, 64, 128))):
    for deg in degs:
        x, ei, bt = batch(B, n, deg, f)
        dens = ei.size(1) / float(B * n * n)
        torch.manual_seed(0)
        pooler = get_pooler("mincut", in_channels=f, k=k).to(dev)
        res = []
        for train in (False, True):
            for route in (0.0, 2.0):
                P._ROWS_ROUTE_DENSITY = route
                pooler.train(train)
                xin = x.clone().requires_grad_(train)

                def step():
                    if not train:
                        with torch.no_grad():
                            return pooler(x=xin, adj=ei, batch=bt)
                    pooler.zero_grad(set_to_none=True)
                    out = pooler(x=xin, adj=ei, batch=bt)
                    (out.x.sum() + out.edge_index.sum() + sum(out.loss.values())).backward()
                res.append(min(timed(step) for _ in range(3)))
        print(f"B={B:4d} n={n:5d} K={k:4d} F={f:3d} deg={deg:4d} {dens:8.4f} | {res[0]:9.3f} {res[1]:9.3f} | {res[2]:10.3f} {res[3]:9.3f}",
              flush=True)
