"""Fused dense Reduce + Connect at awkward shapes (N, K, F not multiples of 4): how much the guarded path costs."""
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
for B, N, Kc, F in ((32, 1024, 128, 64), (32, 1023, 128, 64), (32, 1022, 128, 64), (32, 1024, 127, 64),
                    (32, 1024, 128, 63), (32, 1000, 100, 50), (32, 999, 99, 49), (2, 8192, 512, 128), (2, 8191, 512, 128)):
    s = torch.softmax(torch.randn(B, N, Kc, device=dev, generator=g), -1)
    a = torch.rand(B, N, N, device=dev, generator=g)
    x = torch.randn(B, N, F, device=dev, generator=g)
    us = timed(lambda: K.dense_pool(s, a, x, flags))
    fl = 2.0 * B * (N * N * Kc + N * Kc * Kc + N * Kc * F)
    print(f"B={B:3d} N={N:5d} K={Kc:4d} F={F:4d}: {us:9.1f} us  {fl / us / 1e6:7.1f} TFLOP/s", flush=True)
