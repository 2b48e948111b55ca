"""Host time of one eager training step (forward + bench loss + backward) of a dense pooler on the C2-shaped sparse batch:
cProfile, the 30 functions with the most own time.   python3 tools/host_profile_train.py [mincut_c2|diff_c2|...] [steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
from e2e_launches import CASES, batch_graphs, resolve_sizes  # noqa: E402
from tgp.poolers import get_pooler  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "mincut_c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
alias, kw, sizes, deg, f = CASES[name]
x, ei, batch = batch_graphs(resolve_sizes(sizes), deg, f)
x.requires_grad_(True)
pooler = get_pooler(alias, **kw).to(torch.device("cuda:0")).train()


def step():
    pooler.zero_grad(set_to_none=True)
    x.grad = None
    out = pooler(x=x, adj=ei, batch=batch)
    loss = out.x.sum() + out.edge_index.sum()
    for v in out.loss.values():
        loss = loss + v
    loss.backward()


for _ in range(100):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print(f"{name}: {(time.perf_counter() - t0) / steps * 1e6:.1f} us per step (wall, {steps} steps)")
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:32]
print(f"{'own us/step':>12s} {'cum us/step':>12s} {'calls/step':>10s}  function")
for (fn, line, fname), (cc, nc, tt, ct, _) in rows:
    print(f"{tt / steps * 1e6:12.2f} {ct / steps * 1e6:12.2f} {nc / steps:10.1f}  {os.path.basename(fn)}:{line} {fname}")
