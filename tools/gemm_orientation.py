"""First product of the dense Connect in its two orientations: U = A S (rows = nodes) vs T = S^T A (rows = clusters)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
from tgp import kernels as K
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
def timed(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for B, N, Kc in ((32, 1024, 128), (2, 8192, 512), (64, 512, 64)):
    S = torch.softmax(torch.randn(B, N, Kc, device=dev, generator=g), -1)
    A = torch.rand(B, N, N, device=dev, generator=g)
    fl = 2.0 * B * N * N * Kc
    t1 = timed(lambda: K.bmm(A, S))
    t2 = timed(lambda: K.bmm(S, A, trans_a=True))
    print(f"B={B} N={N} K={Kc}: U=A S {t1:8.1f} us {fl/t1/1e6:6.1f} TF | T=S^T A {t2:8.1f} us {fl/t2/1e6:6.1f} TF", flush=True)
