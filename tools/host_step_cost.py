"""Host-side cost of one benchmark step (Python + launch), measured while the GPU queue is never the limit:
per-step wall time of N launches WITHOUT a synchronise, for a kernel that is much shorter than the launch path."""
import cProfile, os, pstats, sys, time, gc
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "torch-geometric-pool_amd"))
import bench
dev = torch.device("cuda:0")
ctx = bench.Ctx(dev, 0, 1, None)
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
wl = bench.SmallGraphsMinCut(ctx) if which == "c3" else bench.DenseDiffPool("c2", ctx, unfused=False, force_collective=False)
for _ in range(20): wl.step()
gc.collect(); gc.freeze(); ctx.sync()
for trial in range(3):
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(30): wl.step()   # 30 steps: far below the queue depth, the host never waits for the GPU
    t1 = time.perf_counter(); ctx.sync(); t2 = time.perf_counter()
    print(f"{which}: host {1e6*(t1-t0)/30:.1f} us per step (launch only); with the GPU {1e6*(t2-t0)/30:.1f} us per step")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): wl.step()
pr.disable(); ctx.sync()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
