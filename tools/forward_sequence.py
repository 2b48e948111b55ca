"""Kernels of ONE forward of a pooler case of tools/e2e_launches.py in launch order (torch.profiler).
    python tools/forward_sequence.py ndp_c3 [graclus_c4 ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from e2e_launches import CASES, batch_graphs  # noqa: E402
from tgp.poolers import get_pooler  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

dev = torch.device("cuda:0")
for name in [a for a in sys.argv[1:] if a in CASES] or ["ndp_c3"]:
    alias, kw, sizes, deg, f = CASES[name]
    if sizes is None:
        g = torch.Generator().manual_seed(0)
        sizes = torch.randint(20, 61, (2048,), generator=g).tolist()
    x, ei, batch = batch_graphs(sizes, deg, f)
    pooler = get_pooler(alias, **kw).to(dev).eval()

    def fwd():
        with torch.no_grad():
            return pooler(x=x, adj=ei, batch=batch)

    for _ in range(5):
        fwd()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fwd()
        torch.cuda.synchronize()
    evs = sorted([e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA],
                 key=lambda e: e.time_range.start)
    t0 = evs[0].time_range.start
    print(f"== {name}: {len(evs)} device activities, GPU-busy {sum(e.device_time for e in evs):.0f} us, span "
          f"{evs[-1].time_range.end - t0:.0f} us")
    for e in evs:
        print(f"  +{e.time_range.start - t0:8.1f} us  {e.device_time:7.1f} us  {e.name[:110]}")
