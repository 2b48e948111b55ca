"""CSR SpMM T = A S two ways: tgp_spmm_csr_f32 (dispatches by row width) vs the sparse Reduce kernel with identity order."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
from tgp import kernels as K
from tgp import _native as N
dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(0)
def timed(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for n, deg, Kc in ((32768, 16, 128), (80000, 4, 20), (1_000_000, 10, 16), (16384, 16, 50)):
    src = torch.arange(n, device=dev).repeat_interleave(deg)
    dst = torch.randint(0, n, (src.numel(),), device=dev, generator=g)
    ei = torch.stack([src, dst]); w = torch.rand(src.numel(), device=dev, generator=g)
    S = torch.randn(n, Kc, device=dev, generator=g)
    t1 = timed(lambda: K.spmm_sorted(ei, w, n, S))
    row_ptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    L = N.lib(); st = N.stream_ptr(dev)
    L.tgp_rowptr_from_sorted_i64(N.ptr(ei[0].contiguous()), ei.size(1), n, N.ptr(row_ptr), st)
    perm = torch.arange(ei.size(1), dtype=torch.int32, device=dev)
    idx = K.AssignIndex(row_ptr, perm, ei.size(1), n)
    col = ei[1].contiguous()
    t2 = timed(lambda: K.reduce_sparse(S, col, w, idx))
    ref = K.spmm_sorted(ei, w, n, S); got = K.reduce_sparse(S, col, w, idx)
    print(f"N={n} E={ei.size(1)} K={Kc}: spmm_csr {t1:7.1f} us | reduce kernel {t2:7.1f} us | max diff {(ref-got).abs().max().item():.2e}", flush=True)
