"""S = softmax(X W^T + b) (tgp_mlp_select_f32) at the C2 shape and neighbours, back to back (HIP events, 200 calls) and
one call at a time behind an L2-sized memset (cold caches).  TGP_MLP_SPLIT_TILES=0: mlp_select_mfma_kernel everywhere.

    python3 tools/bench_mlp_select.py
"""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "torch-geometric-pool_amd"))
from tgp import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
tag = "mfma kernel (r5)" if os.environ.get("TGP_MLP_SPLIT_TILES") == "0" else "split kernel"
scratch = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for M, F, kk in ((32768, 64, 128), (32768, 64, 64), (32768, 32, 128), (8192, 64, 128), (131072, 64, 128), (32768, 128, 96)):
    x = torch.randn(M, F, device=dev, generator=g)
    w = torch.randn(kk, F, device=dev, generator=g) * 0.2
    b = torch.randn(kk, device=dev, generator=g)
    ref = torch.softmax(x.double() @ w.double().t() + b.double(), -1)
    out = K.mlp_select(x, w, b, None)
    err = float((out.double() - ref).abs().max())
    for _ in range(20):
        K.mlp_select(x, w, b, None)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(200):
        K.mlp_select(x, w, b, None)
    t1.record()
    torch.cuda.synchronize()
    warm = t0.elapsed_time(t1) / 200 * 1e3
    cold = []
    for _ in range(10):
        scratch.zero_()
        a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        K.mlp_select(x, w, b, None)
        c.record()
        torch.cuda.synchronize()
        cold.append(a.elapsed_time(c) * 1e3)
    cold.sort()
    mb = M * 4 * (F + kk) / 1e6
    print(f"{tag:18s} M={M:7d} F={F:4d} K={kk:4d}: back to back {warm:6.1f} us ({mb / warm * 1e-3 * 1e3:5.2f} GB/ms)   "
          f"behind a 512 MB memset {cold[len(cold) // 2]:6.1f} us   max err vs fp64 {err:.1e}", flush=True)
