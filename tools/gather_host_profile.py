#!/usr/bin/env python3
"""Where the host time of a sparse step WITH the variable-size gather goes on a one-rank RCCL group (nothing hides host
time there): wall time per step with and without SparseGather, cProfile of 2000 steps."""
import cProfile
import os
import pstats
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "torch-geometric-pool_amd"))
import bench  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29655")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1)
ctx = bench.Ctx(dev, 0, 1, dist)
wl = bench.TopkBatch(ctx, force_collective=True)
for fn, name in ((wl.compute, "compute only"), (wl.step, "compute + gather")):
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 2000 * 1e6:.1f} us per step (wall)")
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    wl.step()
pr.disable()
wl.sg.flush()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
dist.destroy_process_group()
