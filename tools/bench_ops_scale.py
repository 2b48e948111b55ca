"""Every secondary native operator at a large shape: time + achieved bandwidth over its algorithmic bytes.
Purpose: catch operators that are correct but pathologically slow at scale (contended atomics, serial loops)."""
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


def report(name, us, nbytes=None, flops=None):
    extra = ""
    if nbytes:
        extra += f"  {nbytes / us / 1e6:6.2f} TB/s"
    if flops:
        extra += f"  {flops / us / 1e6:6.1f} TFLOP/s"
    print(f"{name:64s} {us:9.1f} us{extra}", flush=True)


def sorted_batch(num_graphs, lo, hi):
    sizes = torch.randint(lo, hi + 1, (num_graphs,), device=dev, generator=g)
    batch = torch.repeat_interleave(torch.arange(num_graphs, device=dev), sizes)
    ptr = torch.cat([sizes.new_zeros(1), sizes.cumsum(0)])
    return sizes, batch, ptr


# ---- reduce_batch (A2 batch vector) --------------------------------------------------------------
n = 1_000_000
batch = torch.sort(torch.randint(0, 64, (n,), device=dev, generator=g))[0]
k = n // 2
node_index = torch.arange(n, device=dev)
cluster = torch.div(node_index, 2, rounding_mode="floor")
report("reduce_batch_sparse n=1M k=500k", timed(lambda: K.reduce_batch_sparse(batch, node_index, cluster, k)),
       n * 24 + k * 8)

# ---- densify (A11) -------------------------------------------------------------------------------
for B, lo, hi, F, deg in ((32, 1024, 1024, 64, 16), (2048, 20, 60, 32, 4), (2, 8192, 8192, 128, 16)):
    sizes, bt, ptr = sorted_batch(B, lo, hi)
    N = bt.numel()
    x = torch.randn(N, F, device=dev, generator=g)
    nmax = int(sizes.max())
    report(f"to_dense_batch B={B} n~{hi} F={F}", timed(lambda: K.to_dense_batch(x, bt, ptr, B, nmax)),
           N * F * 4 + B * nmax * F * 4)
    src = torch.arange(N, device=dev).repeat_interleave(deg)
    off = (torch.rand(src.numel(), device=dev, generator=g) * sizes[bt[src]]).long()
    dst = ptr[bt[src]] + off
    ei = torch.stack([src, dst])
    ew = torch.rand(src.numel(), device=dev, generator=g)
    for tr in (False, True):
        report(f"to_dense_adj B={B} n~{hi} E={ei.size(1)} transposed={tr}",
               timed(lambda: K.to_dense_adj(ei, ew, bt, ptr, B, nmax, tr)), ei.size(1) * 20 + B * nmax * nmax * 4)
    # unbatched dense connect (A9): T = A S (CSR SpMM), then S_b^T T_b per graph
    Kc = 128 if hi >= 1024 else 20
    s = torch.softmax(torch.randn(N, Kc, device=dev, generator=g), -1)
    report(f"spmm_sorted N={N} E={ei.size(1)} K={Kc}", timed(lambda: K.spmm_sorted(ei, ew, N, s)),
           ei.size(1) * 12 + ei.size(1) * Kc * 4 + N * Kc * 4, 2 * ei.size(1) * Kc)
    t = K.spmm_sorted(ei, ew, N, s)
    report(f"segment_gemm_tn S^T T  N={N} K={Kc}", timed(lambda: K.segment_gemm_tn(s, t, ptr, nmax)),
           2 * N * Kc * 4, 2 * N * Kc * Kc)
    report(f"segment_gemm_tn S^T X  N={N} K={Kc} F={F}", timed(lambda: K.segment_gemm_tn(s, x, ptr, nmax)),
           N * (Kc + F) * 4, 2 * N * Kc * F)

# ---- block-diagonal export (A10) and dense post-processing (A8) ---------------------------------------
for B, Kc in ((32, 128), (2048, 20), (2, 512), (64, 1024)):
    a = torch.rand(B, Kc, Kc, device=dev, generator=g)
    a = a * (torch.rand(B, Kc, Kc, device=dev, generator=g) < 0.5)
    nnz = int((a.abs() > 1e-8).sum())
    report(f"block_diag_edges B={B} K={Kc} nnz={nnz}", timed(lambda: K.block_diag_edges(a)), 2 * a.numel() * 4 + nnz * 20)
    for fl, nm in ((K.dense_flags(True, True, False, False), "loops+deg"),
                   (K.dense_flags(True, True, False, True), "loops+deg+max")):
        report(f"postprocess_dense B={B} K={Kc} {nm}", timed(lambda: K.postprocess_dense(a, fl)), 2 * a.numel() * 4)
