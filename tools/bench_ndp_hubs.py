"""NDPSelect's chip-wide partition (tgp_ndp_large_*) on a 1M-node graph with a few very long rows (hubs of a power-law
graph): the mat-vec and the one-time kernels spread rows beyond NL_HUB entries over whole workgroups.
python tools/bench_ndp_hubs.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-geometric-pool_amd"))
from tgp import kernels as K
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
n = 1_000_000
a = torch.randint(0, n, (3 * n,), device=dev, generator=g); b = torch.randint(0, n, (3 * n,), device=dev, generator=g)
for hubs, deg in ((0, 0), (10, 100_000), (100, 10_000)):
    aa, bb = a, b
    if hubs:
        h = torch.arange(hubs, device=dev).repeat_interleave(deg)
        t = torch.randint(hubs, n, (hubs * deg,), device=dev, generator=g)
        aa, bb = torch.cat([a, h]), torch.cat([b, t])
    keep = aa != bb
    aa, bb = aa[keep], bb[keep]
    key = torch.unique(torch.cat([aa * n + bb, bb * n + aa]))
    ei = torch.stack([key // n, key % n])
    indptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    K.rowptr_from_sorted(ei[0], n, indptr)
    for rep in range(2):
        k8 = torch.zeros(n, dtype=torch.uint8, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        info, state = K.ndp_partition_large(indptr, ei[1], None, 0, n, 1, k8, status, want_state=True)
        torch.cuda.synchronize()
        dt = 1e3 * (time.perf_counter() - t0)
        print(f"hubs {hubs} x deg {deg}: E = {ei.size(1)}, partition {dt:8.2f} ms, {state['steps']} steps "
              f"({dt / max(state['steps'], 1):.3f} ms per step), lambda {state['lambda']:.6f}, cut {state['cut']:.4f}, "
              f"kept {int(k8.sum())}", flush=True)
