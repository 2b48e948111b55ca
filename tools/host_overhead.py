"""Host-side cost (us per call, GPU work negligible) of the native wrappers next to a plain torch op."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
from tgp import kernels as K
from tgp import _native as N
dev = torch.device("cuda:0")
a = torch.randn(4, 8, 8, device=dev); b = torch.randn(4, 8, 8, device=dev)
batch = torch.zeros(100, dtype=torch.long, device=dev); ni = torch.arange(100, device=dev); ci = torch.arange(100, device=dev) // 2
def t(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6
print("K.bmm tiny           host us/call:", round(t(lambda: K.bmm(a, b)), 1))
print("torch.bmm tiny       host us/call:", round(t(lambda: torch.bmm(a, b)), 1))
print("K.reduce_batch_sparse host us/call:", round(t(lambda: K.reduce_batch_sparse(batch, ni, ci, 50)), 1))
print("torch.empty          host us/call:", round(t(lambda: torch.empty(1024, device=dev)), 1))
print("N.stream_ptr         host us/call:", round(t(lambda: N.stream_ptr(dev)), 1))
print("N.workspace(4096)    host us/call:", round(t(lambda: N.workspace(4096, dev)), 1))
print("N.require_device     host us/call:", round(t(lambda: N.require_device(a, b)), 1))
print("N.f32c               host us/call:", round(t(lambda: N.f32c(a)), 1))
print("N.ptr                host us/call:", round(t(lambda: N.ptr(a)), 1))
L = N.lib()
print("ctypes tgp_version   host us/call:", round(t(lambda: L.tgp_version()), 1))
