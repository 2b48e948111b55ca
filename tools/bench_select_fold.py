"""The fused selector + pooling call on padded dense inputs WITH a node mask (kernels.dense_pool_select; C3 shape:
2048 graphs, N = 60, K = 20, F = 32): time per call.   python tools/bench_select_fold.py   (TGP_HIP_LIB picks a build)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
from tgp import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
B, N, Kc, F = 2048, 60, 20, 32
sizes = torch.randint(20, 61, (B,), device=dev, generator=g)
mask = torch.arange(N, device=dev)[None, :] < sizes[:, None]
x = torch.randn(B, N, F, device=dev, generator=g) * mask[..., None]
adj = (torch.rand(B, N, N, device=dev, generator=g) < 0.1).float() * mask[:, :, None] * mask[:, None, :]
w = torch.randn(Kc, F, device=dev, generator=g) * 0.3
b = torch.randn(Kc, device=dev, generator=g) * 0.1
flags = K.dense_flags(True, True, False, False)
for with_mask in (True, False):
    m = mask if with_mask else None
    for _ in range(10):
        K.dense_pool_select(x, adj, w, b, m, flags, want_raw=True, mincut_terms=True)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(200):
        K.dense_pool_select(x, adj, w, b, m, flags, want_raw=True, mincut_terms=True)
    t1.record()
    torch.cuda.synchronize()
    print(f"dense_pool_select, mask {'given' if with_mask else 'None '}: {t0.elapsed_time(t1) / 200 * 1e3:7.1f} us per call")
