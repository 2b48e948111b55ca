"""Soak of the look-back kernels: 30 000 one-launch small-batch calls (refusals counted), 3 000 C4 coalesce / subgraph Connect
steps (sampled against the first result).  python tools/soak_lookback.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "torch-geometric-pool_amd"))
import bench
from tgp import kernels
dev = torch.device("cuda:0")
for which in ("topk_batch", "graclus_batch"):
    wl = bench.TopkBatch(bench.Ctx(dev, 0, 1, None), which=which)
    declined = 0
    t0 = time.perf_counter()
    n = 30000
    for i in range(n):
        with torch.no_grad():
            out = wl.pool.reduce_connect(wl.x, wl.ei, wl.ew, wl.so, wl.batch)
        if out is None:
            declined += 1
            kernels._SPS_DECLINED.clear()
    torch.cuda.synchronize()
    print(f"{which}: {n} calls, {declined} declined, {(time.perf_counter()-t0)/n*1e6:.1f} us per call", flush=True)
# the big-graph look-back kernels
wl = bench.GraclusC4(bench.Ctx(dev, 0, 1, None))
ref = wl.step()
t0 = time.perf_counter()
bad = 0
for i in range(3000):
    out = wl.step()
    if i % 500 == 0 and not (torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])):
        bad += 1
torch.cuda.synchronize()
print(f"c4_graclus: 3000 steps, {bad} mismatching samples, {(time.perf_counter()-t0)/3000*1e3:.3f} ms per step")
wl = bench.TopkConnect(bench.Ctx(dev, 0, 1, None))
ref = wl.step()
bad = 0
for i in range(3000):
    out = wl.step()
    if i % 500 == 0 and not (torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])):
        bad += 1
torch.cuda.synchronize()
print(f"topk_connect: 3000 steps, {bad} mismatching samples")
