"""Kernels of ONE step of bench.py's e2e_train_mincut_c3 workload in launch order (torch.profiler), with the autograd
node / Python frame that launched each.      python tools/train_step_sequence.py"""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "torch-geometric-pool_amd"))
import bench  # noqa: E402


w = bench.PoolerTrainStep(bench.Ctx(torch.device("cuda:0"), 0, 1, None))
for _ in range(5):
    w.step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    w.step()
    torch.cuda.synchronize()
evs = prof.events()
kern = sorted([e for e in evs if e.device_type == torch.autograd.DeviceType.CUDA], key=lambda e: e.time_range.start)
cpu = [e for e in evs if e.device_type == torch.autograd.DeviceType.CPU]
print(f"{len(kern)} device activities in one step")
t0 = kern[0].time_range.start
for e in kern:
    # innermost CPU op whose time range launched it: match through the correlation of linked kernels
    owner = ""
    for c in cpu:
        if any(k is e or (k.name == e.name and k.time_range.start == e.time_range.start) for k in getattr(c, "kernels", [])):
            owner = c.name
    print(f"  +{(e.time_range.start - t0):8.1f} us  {e.device_time:7.1f} us  {e.name[:90]:90s}  <- {owner[:50]}")
