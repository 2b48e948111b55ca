"""T = A S of the dense poolers' rows route (tgp_spmm_csr_f32) at the C2 shape: 32 graphs x 1024 nodes, ~10 entries per
row, K = 128 (and K = 64 / 256).  TGP_SPMM_REDUCE_ROUTE=1 = the r5 route (the sparse Reduce's gather-sum), default =
the XCD-grouped row kernel; TGP_SPMM_ROWS_ITER = row groups per workgroup.

    python3 tools/bench_spmm.py
"""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "torch-geometric-pool_amd"))
from tgp import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def batch(B, n, deg):
    rows, cols = [], []
    for b in range(B):
        a = torch.rand(n, n, device=dev, generator=g) < deg / n
        a = a | a.t()
        a.fill_diagonal_(False)
        e = a.nonzero().t()
        rows.append(e[0] + b * n)
        cols.append(e[1] + b * n)
    return torch.stack([torch.cat(rows), torch.cat(cols)])


def timed(fn, reps=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps * 1e3


tag = "reduce route (r5)" if os.environ.get("TGP_SPMM_REDUCE_ROUTE") == "1" else \
    f"row kernel, ITER={os.environ.get('TGP_SPMM_ROWS_ITER', '1')}"
for B, n, deg, kk in ((32, 1024, 5.0, 128), (32, 1024, 5.0, 64), (32, 1024, 5.0, 256), (2048, 40, 2.8, 64), (4, 8192, 8.0, 128),
                      (2048, 40, 2.8, 20), (2048, 40, 2.8, 32), (32, 1024, 5.0, 16), (32, 1024, 5.0, 8)):
    ei = batch(B, n, deg)
    N = B * n
    w = torch.rand(ei.size(1), device=dev, generator=g) + 0.5
    s = torch.softmax(torch.randn(N, kk, device=dev, generator=g), -1)
    rp = K.csr_offsets(ei, N)
    out = K.spmm_csr(rp, ei, w, N, s)
    ref = torch.zeros(N, kk, device=dev, dtype=torch.float64).index_add_(0, ei[0], s[ei[1]].double() * w.double()[:, None])
    err = float((out.double() - ref).abs().max())
    us = timed(lambda: K.spmm_csr(rp, ei, w, N, s))
    gathered = ei.size(1) * kk * 4 / 1e6
    print(f"{tag:24s} B={B:5d} n={n:5d} E={ei.size(1):8d} K={kk:4d}: {us:7.1f} us  ({gathered:6.1f} MB of S rows gathered = "
          f"{gathered / us * 1e-3 * 1e3:6.2f} GB/ms)  max err vs fp64 {err:.2e}", flush=True)
