"""Time tgp_mlp_select_bwd_f32 (the selector's backward in one launch) and its parts.   python tools/bench_mlp_select_bwd.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
from tgp import kernels as K  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for M, Kc, F in [(122880, 20, 32), (122880, 32, 64), (1 << 20, 20, 32)]:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, F, generator=g).to(dev)
    w = torch.randn(Kc, F, generator=g).to(dev)
    gs = torch.randn(M, Kc, generator=g).to(dev)
    s = torch.softmax(torch.randn(M, Kc, generator=g), -1).to(dev)
    acc = torch.zeros(M, F, device=dev)
    full = timed(lambda: K.mlp_select_bwd(s, gs, x, w))
    accum = timed(lambda: K.mlp_select_bwd(s, gs, x, w, gx_accumulate=acc))
    only_gx = timed(lambda: K.mlp_select_bwd(s, gs, x, w, want_gw=False, want_gb=False))
    only_gw = timed(lambda: K.mlp_select_bwd(s, gs, x, w, want_gx=False))
    old = timed(lambda: (K.softmax_bwd(s, gs),))
    mb = M * 4.0 * (2 * Kc + 2 * F) / 1e6
    print(f"M={M} K={Kc} F={F}: all three {full:6.1f} us ({mb / full:5.2f} TB/s of {mb:.0f} MB), accumulating {accum:6.1f}, "
          f"gx only {only_gx:6.1f}, gw+gb only {only_gw:6.1f}; softmax_bwd kernel alone (old first step) {old:6.1f} us")
