"""Diagnostic: per-wave phase times of dense_pool_medium_kernel (stamps build: `make -C torch-geometric-pool_amd/csrc stamps`).
Slots: 0 start, 1 S in LDS, 2 strips done (per wave), 3 partial sums added, 4 end.  Usage: medium_stamps.py B N K F"""
import ctypes
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
lib = ctypes.CDLL(os.path.join(ROOT, "torch-geometric-pool_amd", "lib", "libtgp_hip_stamps.so"))
p, i64, ci, sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_size_t
lib.tgp_dense_pool_workspace_bytes.restype = sz
lib.tgp_dense_pool_workspace_bytes.argtypes = [i64] * 4
lib.tgp_dense_pool_f32.argtypes = [p, p, p, i64, i64, i64, i64, ci, ctypes.c_float, p, p, p, p, p, sz, p]
lib.tgp_debug_set_gemm_stamps.argtypes = [p]
B, N, K, F = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (1024, 200, 50, 64)
dev = torch.device("cuda:0")
torch.manual_seed(0)
S = torch.softmax(torch.randn(B, N, K, device=dev), -1)
A = (torch.rand(B, N, N, device=dev) < 0.1).float()
X = torch.randn(B, N, F, device=dev)
xp, ap = torch.empty(B, K, F, device=dev), torch.empty(B, K, K, device=dev)
ws = torch.empty(lib.tgp_dense_pool_workspace_bytes(B, N, K, F), dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream(dev).cuda_stream


def run():
    rc = lib.tgp_dense_pool_f32(S.data_ptr(), A.data_ptr(), X.data_ptr(), B, N, K, F, 1 | 2 | 8, 1e-8, None, xp.data_ptr(), None,
                                ap.data_ptr(), ws.data_ptr(), ws.numel(), stream)
    assert rc == 0


for _ in range(3):
    run()
stamps = torch.zeros(B * 4 * 16, dtype=torch.int64, device=dev)
assert lib.tgp_debug_set_gemm_stamps(stamps.data_ptr()) == 0
torch.cuda.synchronize()
run()
torch.cuda.synchronize()
st = stamps.view(B, 4, 16)[:, :, :5].cpu().double() / 100.0
st = st - st[:, :, 0].min()
names = ["start", "S in LDS", "strips done", "partials added", "end"]
print(f"graphs {B} (N={N} K={K} F={F}); span {float(st[:, :, 4].max()):.2f} us")
for i, nm in enumerate(names):
    col = st[:, :, i].reshape(-1)
    d = (st[:, :, i] - st[:, :, i - 1]).reshape(-1) if i else col
    print(f"  {nm:16s} at {col.quantile(0.5):7.2f} us (p10 {col.quantile(0.1):7.2f}, p90 {col.quantile(0.9):7.2f})"
          f"   phase {d.quantile(0.5):6.2f} us (p10 {d.quantile(0.1):6.2f}, p90 {d.quantile(0.9):6.2f})")
per_wave = (st[:, :, 2] - st[:, :, 1])
print("  strip phase by wave id (median us):", [round(float(per_wave[:, w].quantile(0.5)), 2) for w in range(4)])
starts = st[:, 0, 0]
print(f"  workgroup start times: p50 {starts.quantile(0.5):.2f}, p90 {starts.quantile(0.9):.2f}, max {starts.max():.2f} us")
