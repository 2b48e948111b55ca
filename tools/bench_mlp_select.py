"""MLPSelect's last Linear + softmax + mask (kernels.mlp_select) at the C2 and C3 shapes: time per call.
python tools/bench_mlp_select.py   (TGP_HIP_LIB picks a build)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
from tgp import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for M, F, Kc in ((32 * 1024, 64, 128), (2048 * 60, 32, 20), (8192 * 2, 128, 512 // 4), (1 << 20, 128, 64)):
    x = torch.randn(M, F, device=dev, generator=g)
    w = torch.randn(Kc, F, device=dev, generator=g) * 0.2
    b = torch.randn(Kc, device=dev, generator=g) * 0.1
    for _ in range(5):
        K.mlp_select(x, w, b, None)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(100):
        K.mlp_select(x, w, b, None)
    t1.record()
    torch.cuda.synchronize()
    us = t0.elapsed_time(t1) / 100 * 1e3
    print(f"mlp_select M={M:8d} F={F:4d} K={Kc:4d}: {us:8.1f} us   {(M * (F + Kc) * 4) / us / 1e6:6.2f} TB/s   {2.0 * M * F * Kc / us / 1e6:6.1f} TFLOP/s")
