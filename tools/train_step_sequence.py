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
owners = {}
for c in cpu:
    for k in getattr(c, "kernels", []):
        key = (k.name, int(k.duration))
        span = c.time_range.end - c.time_range.start
        if key not in owners or span < owners[key][0]:
            owners[key] = (span, c.name)
for e in kern:
    owner = owners.get((e.name, int(e.device_time)), (0, ""))[1]
    print(f"  +{(e.time_range.start - t0):8.1f} us  {e.device_time:7.1f} us  {e.name[:90]:90s}  <- {owner[:60]}")
