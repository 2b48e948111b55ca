"""Dense poolers on a RAGGED batch (graph sizes log-normal: many small graphs, a few large ones, all padded to the
largest): whole forward and the fused Reduce + Connect kernel with and without the per-graph size hint."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
from tgp import kernels as K  # noqa: E402
from tgp.poolers import get_pooler  # noqa: E402

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


for B, mean, cap, Kc, F in ((512, 40, 400, 20, 32), (1024, 30, 300, 16, 64), (256, 120, 500, 50, 64)):
    sizes = torch.exp(torch.randn(B, device=dev, generator=g) * 0.7 + torch.log(torch.tensor(float(mean)))).long().clamp(4, cap)
    sizes[0] = cap
    N = int(sizes.max())
    mask = torch.arange(N, device=dev).unsqueeze(0) < sizes.unsqueeze(1)
    A = (torch.rand(B, N, N, device=dev, generator=g) < 0.05).float()
    A = torch.maximum(A, A.transpose(1, 2)) * mask.unsqueeze(1) * mask.unsqueeze(2)
    A = A.contiguous()
    X = torch.randn(B, N, F, device=dev, generator=g) * mask.unsqueeze(-1)
    S = torch.softmax(torch.randn(B, N, Kc, device=dev, generator=g), -1) * mask.unsqueeze(-1)
    flags = K.dense_flags(True, True, True, False)
    t0 = timed(lambda: K.dense_pool(S, A, X, flags))
    t1 = timed(lambda: K.dense_pool(S, A, X, flags, graph_sizes=sizes))
    real = float((sizes.double() ** 2).sum() * 4 / 1e6)
    print(f"B={B} sizes mean {float(sizes.float().mean()):.0f} max {N} K={Kc} F={F}: padded A {B * N * N * 4 / 1e6:.0f} MB, real "
          f"{real:.0f} MB | fused kernel {t0:7.1f} us -> with sizes {t1:7.1f} us", flush=True)
    # the same batch as a sparse PyG-style input through the pooler
    batch = torch.repeat_interleave(torch.arange(B, device=dev), sizes)
    ptr = torch.cat([sizes.new_zeros(1), sizes.cumsum(0)])
    nz = A.nonzero()
    ei = torch.stack([ptr[nz[:, 0]] + nz[:, 1], ptr[nz[:, 0]] + nz[:, 2]])
    x = X[mask]
    pooler = get_pooler("mincut", in_channels=F, k=Kc).to(dev).eval()
    with torch.no_grad():
        for _ in range(3):
            pooler(x=x, adj=ei, batch=batch)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            pooler(x=x, adj=ei, batch=batch)
        torch.cuda.synchronize()
    print(f"    mincut pooler forward on the sparse batch: {(time.perf_counter() - t) / 10 * 1e3:.3f} ms", flush=True)
