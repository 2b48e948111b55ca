#!/usr/bin/env python3
"""Where the host time of the one-launch sparse pooling call goes (tgp_sparse_pool_small_f32 through
SRCPooling.reduce_connect): wall time per call, cProfile of 2000 calls, and the kernel's own duration from HIP events."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "torch-geometric-pool_amd"))
import bench  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "topk"
dev = torch.device("cuda:0")
ctx = bench.Ctx(dev, 0, 1, None)
wl = bench.TopkBatch(ctx, which="topk_batch" if which == "topk" else "graclus_batch")
for _ in range(50):
    wl.compute()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    wl.compute()
torch.cuda.synchronize()
print(f"{which}: {(time.perf_counter() - t0) / 2000 * 1e6:.1f} us per call (wall)")
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    wl.compute()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
