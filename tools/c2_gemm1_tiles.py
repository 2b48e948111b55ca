"""C2's first product U = A S (2*32*1024^2*128 flop) under the tile shapes the launcher can be forced to
(TGP_GEMM_BM / TGP_GEMM_BN are read once per process: run this script once per setting)."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
from tgp import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
B, N, Kc = 32, 1024, 128
a = (torch.rand(B, N, N, device=dev, generator=g) < 0.01).float()
s = torch.softmax(torch.randn(B, N, Kc, device=dev, generator=g), -1)
for _ in range(20):
    K.bmm(a, s)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        K.bmm(a, s)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 100)
ms = sorted(ts)[2]
print(f"BM={os.environ.get('TGP_GEMM_BM', '-')} BN={os.environ.get('TGP_GEMM_BN', '-')}: {ms * 1e3:.1f} us per launch = "
      f"{2.0 * B * N * N * Kc / ms / 1e9:.1f} TFLOP/s")
