"""Diagnostic: per-wave phase times of dense_pool_small_kernel at the C3 shape.  Needs the stamps build
(`make -C torch-geometric-pool_amd/csrc stamps`).  Slots: 0 start, 1 operands in registers / A tile in LDS,
2 X' done + stored, 3 U and A' MFMAs done, 4 post-processing done, 5 end."""
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
B, N, K, F = (int(sys.argv[1]) if len(sys.argv) > 1 else 2048), 60, 20, 32
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
stamps = torch.zeros(B * 16, dtype=torch.int64, device=dev)
assert lib.tgp_debug_set_gemm_stamps(stamps.data_ptr()) == 0
torch.cuda.synchronize()
run()
torch.cuda.synchronize()
st = stamps.view(-1, 16)[:, :6].cpu().double() / 100.0
st = st - st[:, 0].min()
names = ["start", "operands ready", "X' stored", "U, A' MFMAs", "post-processing", "end"]
print(f"waves {st.size(0)}; span {float(st[:, 5].max()):.2f} us")
for i, nm in enumerate(names):
    col = st[:, i]
    d = (st[:, i] - st[:, i - 1]) if i else col
    print(f"  {nm:18s} at {col.quantile(0.5):6.2f} us (p10 {col.quantile(0.1):6.2f}, p90 {col.quantile(0.9):6.2f})"
          f"   phase {d.quantile(0.5):6.2f} us (p90 {d.quantile(0.9):6.2f})")

# the two load groups of a workgroup separately (waves 0-3 request first, waves 4-7 queue behind them)
grp = (torch.arange(st.size(0)) % 8) >= 4
for gi, gname in ((False, "waves 0-3"), (True, "waves 4-7")):
    sub = st[grp == gi]
    print(f"  {gname}: " + ", ".join(f"{nm} {sub[:, i].quantile(0.5):5.2f}" for i, nm in enumerate(names)))
