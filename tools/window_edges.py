"""Where do the ~80 us of fixed overhead of a short timed window go?  Wall clock vs HIP events around 20 C2 steps."""
import os, sys, time, gc
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "torch-geometric-pool_amd"))
import bench
dev = torch.device("cuda:0")
class A: unfused = False
ctx = bench.Ctx(dev, 0, 1, None)
wl = bench.DenseDiffPool("c2", ctx, unfused=False, force_collective=False)
for _ in range(5): wl.step()
gc.collect(); gc.freeze()
for _ in range(1000): wl.step()
for trial in range(6):
    ctx.sync()
    e0, e1, ef = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    wl.step()
    ef.record()
    t_first_launch = time.perf_counter()
    for _ in range(19): wl.step()
    e1.record()
    t_launched = time.perf_counter()
    ctx.sync()
    t1 = time.perf_counter()
    print(f"wall {1e3*(t1-t0):.3f} ms; events {e0.elapsed_time(e1):.3f} ms; first step (events) {e0.elapsed_time(ef)*1e3:.1f} us; "
          f"host: first step launched after {1e6*(t_first_launch-t0):.0f} us, all launched after {1e6*(t_launched-t0):.0f} us", flush=True)
