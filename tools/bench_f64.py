"""fp64 dense path (r5): time the fused S^T X + S^T A S + post call and the plain product on float64 tensors, print
TFLOP/s against the measured fp64 matrix rate (tools/micro/mfma_f64_rate)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-geometric-pool_amd"))
from tgp import kernels as K  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    for (B, N, Kc, F) in [(32, 1024, 128, 64), (2, 8192, 512, 128), (256, 200, 32, 32)]:
        a = (torch.rand(B, N, N, device=dev, generator=g) < 0.01).double()
        s = torch.softmax(torch.randn(B, N, Kc, device=dev, generator=g, dtype=torch.float64), -1)
        x = torch.randn(B, N, F, device=dev, generator=g, dtype=torch.float64)
        flags = K.dense_flags(True, True, True, False)
        t = timed(lambda: K.dense_pool(s, a, x, flags))
        flop = 2.0 * B * (N * N * Kc + Kc * N * (Kc + F))
        t1 = timed(lambda: K.bmm(a, s))
        t2 = timed(lambda: torch.bmm(a, s))
        print(f"B={B} N={N} K={Kc} F={F}: fused call {t * 1e3:.3f} ms = {flop / t * 1e-12:.1f} TFLOP/s; "
              f"U = A S alone {t1 * 1e3:.3f} ms = {2.0 * B * N * N * Kc / t1 * 1e-12:.1f} TFLOP/s "
              f"(torch.bmm / rocBLAS fp64 on the same operands: {t2 * 1e3:.3f} ms)")


if __name__ == "__main__":
    main()
