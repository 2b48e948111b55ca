"""gW = dY^T X of the selector at C2 scale (n = 32768 rows, 128 x 64): the slab-wise small-bmm + sum against the
split-range segment product."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
from tgp import kernels as K, functions as Fn
dev = torch.device("cuda:0")
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(2_000_000)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3
for n, p, q in [(32768, 128, 64), (32768, 128, 128), (16384, 512, 128), (122880, 20, 32)]:
    a = torch.randn(n, p, device=dev); b = torch.randn(n, q, device=dev)
    ref = a.double().t() @ b.double()
    r1 = Fn._tall_skinny_tn(a, b)
    r2 = K.segment_gemm_tn(a, b, Fn._whole_range(n, dev), n)[0]
    r3 = K.bmm(a.view(1, n, p), b.view(1, n, q), trans_a=True)[0]
    print(n, p, q, "slab %.1f us  seg %.1f us  bmm %.1f us  torch %.1f us  colsum %.1f us" % (
        t(lambda: Fn._tall_skinny_tn(a, b)), t(lambda: K.segment_gemm_tn(a, b, Fn._whole_range(n, dev), n)),
        t(lambda: K.bmm(a.view(1, n, p), b.view(1, n, q), trans_a=True)), t(lambda: a.t() @ b), t(lambda: a.sum(0))),
        "err", float((r1 - ref).abs().max()), float((r2 - ref).abs().max()), float((r3 - ref).abs().max()))

print("batched TN over row slabs, B operand a column block of a wider buffer (ld 452), then sum over the slabs")
n, p, q, ld = 32768, 128, 68, 452
a = torch.randn(n, p, device=dev); wide = torch.randn(n, ld, device=dev)
ref = a.double().t() @ wide[:, 256:256 + q].double()
for G in (16, 32, 64, 128, 256, 512):
    out = torch.empty(G, p, q, device=dev)
    def run():
        K.bmm_into(a.view(G, n // G, p), wide.view(G, n // G, ld)[:, :, 256:256 + q], out, trans_a=True)
        return out.sum(0)
    r = run()
    print(G, "%.1f us" % t(run), "gemm only %.1f us" % t(lambda: K.bmm_into(a.view(G, n // G, p), wide.view(G, n // G, ld)[:, :, 256:256 + q], out, trans_a=True)),
          "err", float((r - ref).abs().max()))
