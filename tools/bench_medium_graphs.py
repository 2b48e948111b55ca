"""Fused dense Reduce + Connect on batches of small / medium graphs (TU-dataset-like shapes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
from tgp import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


flags = K.dense_flags(True, True, True, False)
shapes = [(2048, 60, 20, 32), (2048, 60, 20, 64), (2048, 60, 40, 32), (2048, 64, 32, 32), (2048, 100, 20, 32),
          (2048, 126, 32, 64), (1024, 200, 50, 64), (512, 300, 64, 128), (256, 500, 100, 64), (64, 60, 20, 32),
          (16, 60, 20, 32), (128, 620, 150, 89)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for B, N, Kc, F in shapes:
    s = torch.softmax(torch.randn(B, N, Kc, device=dev, generator=g), -1)
    a = torch.rand(B, N, N, device=dev, generator=g)
    x = torch.randn(B, N, F, device=dev, generator=g)
    us = timed(lambda: K.dense_pool(s, a, x, flags))
    nbytes = 4.0 * B * (N * N + N * Kc + N * F + Kc * Kc + Kc * F)
    fl = 2.0 * B * (N * N * Kc + N * Kc * Kc + N * Kc * F)
    print(f"B={B:5d} N={N:4d} K={Kc:4d} F={F:4d}: {us:9.1f} us  {nbytes / us / 1e6:6.2f} TB/s  {fl / us / 1e6:7.1f} TFLOP/s  "
          f"{B * N / us:8.1f} Mnodes/s", flush=True)
