import os, sys, torch
sys.path.insert(0, "torch-geometric-pool_amd"); sys.path.insert(0, "tools")
from e2e_launches import batch_graphs, wall
from tgp.poolers import get_pooler
dev = torch.device("cuda:0")
for name, sizes, deg, f, k in (("c2", [1024] * 32, 10, 64, 128), ("c3", None, 4, 32, 20)):
    if sizes is None:
        g = torch.Generator().manual_seed(0); sizes = torch.randint(20, 61, (2048,), generator=g).tolist()
    x, ei, batch = batch_graphs(sizes, deg, f)
    for alias in ("mincut_u", "diff_u"):
        pooler = get_pooler(alias, in_channels=f, k=k).to(dev).eval()
        def fwd():
            with torch.no_grad():
                return pooler(x=x, adj=ei, batch=batch)
        ms = wall(fwd)
        from torch.profiler import ProfilerActivity, profile
        fwd(); torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            fwd(); torch.cuda.synchronize()
        evs = sorted([e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA], key=lambda e: e.time_range.start)
        print(f"== {alias} {name}: {ms:.3f} ms/forward, {len(evs)} kernels, GPU-busy {sum(e.device_time for e in evs)/1e3:.3f} ms")
        for e in evs:
            print(f"   {e.device_time:7.1f} us  {e.name[:110]}")
