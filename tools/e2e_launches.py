"""Whole pooler forwards on sparse PyG-style inputs: wall time per call (eager), kernels launched per call
(torch.profiler), and the same call replayed from a HIP graph where it can be captured.

    python tools/e2e_launches.py [diff_c2] [mincut_c3] [diff_c3] [--list-kernels]
"""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
from tgp.poolers import get_pooler  # noqa: E402

dev = torch.device("cuda:0")


def batch_graphs(sizes, deg, f, seed=0):
    """Sorted, duplicate-free undirected edge list over a batch of graphs with the given sizes."""
    g = torch.Generator(device=dev).manual_seed(seed)
    sizes = torch.as_tensor(sizes, device=dev)
    n = int(sizes.sum())
    batch = torch.repeat_interleave(torch.arange(sizes.numel(), device=dev), sizes)
    start = torch.cumsum(sizes, 0) - sizes
    src = torch.arange(n, device=dev).repeat_interleave(max(deg // 2, 1))
    dst = start[batch[src]] + (torch.rand(src.numel(), device=dev, generator=g) * sizes[batch[src]]).long()
    keep = src != dst
    src, dst = src[keep], dst[keep]
    key = torch.unique(torch.cat([src * n + dst, dst * n + src]))
    ei = torch.stack([key // n, key % n])
    x = torch.randn(n, f, device=dev, generator=g)
    return x, ei, batch


def wall(fn, iters=50):
    for _ in range(5):
        fn()
    gc.collect()
    gc.freeze()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / iters * 1e3)
    return sorted(ts)[len(ts) // 2]


def count_kernels(fn, list_kernels=False):
    from torch.profiler import ProfilerActivity, profile
    fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    if list_kernels:
        for e in sorted(evs, key=lambda e: e.time_range.start):
            print(f"     {e.device_time:8.1f} us  {e.name[:110]}")
        print(f"     GPU-busy {sum(e.device_time for e in evs) / 1e3:.3f} ms")
    return len(evs)


CASES = {
    "diff_c2": ("diff", dict(in_channels=64, k=128), [1024] * 32, 10, 64),
    "mincut_c2": ("mincut", dict(in_channels=64, k=128), [1024] * 32, 10, 64),
    "mincut_c3": ("mincut", dict(in_channels=32, k=20), None, 4, 32),
    "diff_c3": ("diff", dict(in_channels=32, k=20), None, 4, 32),
    # medium graphs (TU-dataset-like: 512 graphs of 100..200 nodes, ~8 entries per row): beyond the one-wave kernels,
    # denser than the rows route's bound
    "mincut_med": ("mincut", dict(in_channels=32, k=32), "med", 8, 32),
    "diff_med": ("diff", dict(in_channels=32, k=32), "med", 8, 32),
    "mincut_u_c3": ("mincut_u", dict(in_channels=32, k=20), None, 4, 32),
    "diff_u_c3": ("diff_u", dict(in_channels=32, k=20), None, 4, 32),
    "mincut_u_c2": ("mincut_u", dict(in_channels=64, k=128), [1024] * 32, 10, 64),
    "topk_c3": ("topk", dict(in_channels=32, ratio=0.5), None, 4, 32),
    "graclus_c3": ("graclus", dict(), None, 4, 32),
    "ndp_c3": ("ndp", dict(), None, 4, 32),
    "topk_c4": ("topk", dict(in_channels=128, ratio=0.5), [1_000_000], 10, 128),
    "graclus_c4": ("graclus", dict(), [1_000_000], 10, 128),
}


def resolve_sizes(sizes):
    """None: 2048 PROTEINS-shaped graphs of 20..60 nodes; "med": 512 graphs of 100..200 nodes; a list: as given."""
    if sizes is None:
        return torch.randint(20, 61, (2048,), generator=torch.Generator().manual_seed(0)).tolist()
    if isinstance(sizes, str) and sizes == "med":
        return torch.randint(100, 201, (512,), generator=torch.Generator().manual_seed(0)).tolist()
    return sizes


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    list_kernels = "--list-kernels" in sys.argv
    for name in (args or list(CASES)):
        alias, kw, sizes, deg, f = CASES[name]
        sizes = resolve_sizes(sizes)
        x, ei, batch = batch_graphs(sizes, deg, f)
        pooler = get_pooler(alias, **kw).to(dev).eval()

        def fwd():
            with torch.no_grad():
                return pooler(x=x, adj=ei, batch=batch)

        ms = wall(fwd)
        try:
            nk = count_kernels(fwd, list_kernels)
        except Exception as exc:  # profiler not available: report and go on
            nk = f"n/a ({type(exc).__name__})"
        line = f"{name:12s} eager {ms:7.3f} ms/forward   kernels/forward {nk}"
        try:
            gph = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                fwd()
            torch.cuda.current_stream().wait_stream(s)
            with torch.cuda.graph(gph):
                fwd()
            line += f"   graph replay {wall(gph.replay):7.3f} ms"
        except Exception as exc:
            msg = str(exc).splitlines()[0] if str(exc) else ""
            if "captur" in msg:  # a count -> fill entry read its count back: a host synchronisation inside the capture
                line += "   (count -> fill host read: not capturable)"
            else:
                line += f"   graph capture failed: {type(exc).__name__}: {msg[:80]}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
