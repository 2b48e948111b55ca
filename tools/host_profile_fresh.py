"""Where the host time of the batched sparse poolers' Reduce + Connect goes on NEW tensor objects (bench.py's
topk_batch_fresh / graclus_batch_fresh step): cProfile over the step, the 25 functions with the most own time.

    python tools/host_profile_fresh.py [topk_batch_fresh|graclus_batch_fresh] [calls]
"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "torch-geometric-pool_amd"))
import bench  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "topk_batch_fresh"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
ctx = bench.Ctx(torch.device("cuda:0"), 0, 1, None)
if which.startswith("e2e:"):  # a whole pooler forward on new edge_index / batch objects (tools/e2e_fresh_batch.py's case)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from e2e_launches import CASES, batch_graphs, resolve_sizes
    from tgp.poolers import get_pooler
    alias, kw, sizes, deg, f = CASES[which[4:]]
    sizes = resolve_sizes(sizes)
    x, ei, batch = batch_graphs(sizes, deg, f)
    pooler = get_pooler(alias, **kw).to(ctx.dev).eval()
    copies = [(ei.clone(), batch.clone()) for _ in range(24)]

    class _Fwd:
        turn = 0

        def step(self):
            self.turn = (self.turn + 1) % 24
            e, b = copies[self.turn]
            with torch.no_grad():
                # (new Python objects over the same memory, as a DataLoader's batches are new objects)
                return pooler(x=x, adj=e.view(2, -1), batch=b.view(-1))
    wl = _Fwd()
else:
    wl = bench.TopkBatch(ctx, which=which)
for _ in range(200):
    wl.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(calls):
    wl.step()
torch.cuda.synchronize()
print(f"{which}: {(time.perf_counter() - t0) / calls * 1e6:.1f} us per step (wall, {calls} steps)")
pr = cProfile.Profile()
pr.enable()
for _ in range(calls):
    wl.step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime")
rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:28]
print(f"{'own us/step':>12s} {'cum us/step':>12s} {'calls/step':>10s}  function")
for (fn, line, name), (cc, nc, tt, ct, _) in rows:
    print(f"{tt / calls * 1e6:12.2f} {ct / calls * 1e6:12.2f} {nc / calls:10.1f}  {os.path.basename(fn)}:{line} {name}")
