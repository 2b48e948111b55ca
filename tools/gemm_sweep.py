"""Fixed cost vs steady-state rate of the fp32 MFMA GEMM: time C[b] = A[b] S[b] for growing inner size."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
from tgp import kernels  # noqa: E402

dev = torch.device("cuda:0")
B, M, Nc = 32, 1024, 128
if len(sys.argv) > 1:
    B, M, Nc = (int(v) for v in sys.argv[1:4])
for Kd in (32, 64, 128, 256, 512, 1024, 2048, 4096):
    A = torch.rand(B, M, Kd, device=dev)
    S = torch.rand(B, Kd, Nc, device=dev)
    for _ in range(3):
        kernels.bmm(A, S)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(20):
        a.record()
        kernels.bmm(A, S)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    t = ts[len(ts) // 2]
    print(f"Kd={Kd:5d}  {t * 1e3:8.1f} us   {2.0 * B * M * Nc * Kd / t / 1e9:7.1f} TFLOP/s")
