"""Where the HOST time of a pooler training step goes (cProfile over 300 steps, GPU work left asynchronous).

    python tools/profile_host_step.py [mincut_c3|diff_c3|...] [--forward] [--infer [--fresh]] [--tottime]

--infer: the pooler's forward alone under no_grad, eval mode (any alias of e2e_launches.CASES, sparse poolers included).
"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from e2e_launches import CASES, batch_graphs  # noqa: E402
from tgp.poolers import get_pooler  # noqa: E402

dev = torch.device("cuda:0")
name = next((a for a in sys.argv[1:] if a in CASES), "mincut_c3")
alias, kw, sizes, deg, f = CASES[name]
if sizes is None:
    g = torch.Generator().manual_seed(0)
    sizes = torch.randint(20, 61, (2048,), generator=g).tolist()
x, ei, batch = batch_graphs(sizes, deg, f)
x.requires_grad_(True)
pooler = get_pooler(alias, **kw).to(dev).train()
if "--infer" in sys.argv:
    pooler.eval()


def step():
    if "--infer" in sys.argv:
        with torch.no_grad():
            if "--fresh" in sys.argv:  # new tensor objects per call: the per-tensor memos miss (a DataLoader's batches)
                pooler(x=x, adj=ei.clone(), batch=batch.clone())
            else:
                pooler(x=x, adj=ei, batch=batch)
        return
    pooler.zero_grad(set_to_none=True)
    x.grad = None
    out = pooler(x=x, adj=ei, batch=batch)
    loss = out.x.sum() + out.edge_index.sum() + sum(out.loss.values())
    if "--forward" not in sys.argv:
        loss.backward()


for _ in range(20):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime" if "--tottime" in sys.argv else "cumulative").print_stats(45)
