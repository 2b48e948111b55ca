"""Diagnostic (stamps build): per-workgroup phase times of the link-loss kernel (gemm_f32_mfma_kernel MODE 1) at the C2 shape:
start, prologue (entry -> first barrier), k-loop, and how many workgroups a CU runs one after the other."""
import ctypes
import os

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
lib = ctypes.CDLL(os.path.join(ROOT, "torch-geometric-pool_amd", "lib", "libtgp_hip_stamps.so"))
p, i64, sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_size_t
lib.tgp_link_loss_workspace_bytes.restype = sz
lib.tgp_link_loss_workspace_bytes.argtypes = [i64, i64, i64]
lib.tgp_link_loss_f32.argtypes = [p, p, i64, i64, i64, p, p, p, sz, p]
lib.tgp_debug_set_gemm_stamps.argtypes = [p]
B, N, K = 32, 1024, 128
dev = torch.device("cuda:0")
torch.manual_seed(0)
A = (torch.rand(B, N, N, device=dev) < 0.01).float()
A = ((A + A.transpose(1, 2)) > 0).float()
S = torch.softmax(torch.randn(B, N, K, device=dev), -1)
sq = torch.empty(B, device=dev)
ws = torch.empty(int(lib.tgp_link_loss_workspace_bytes(B, N, K)), dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream(dev).cuda_stream


def run():
    assert lib.tgp_link_loss_f32(S.data_ptr(), A.data_ptr(), B, N, K, None, sq.data_ptr(), ws.data_ptr(), ws.numel(), stream) == 0


for _ in range(3):
    run()
nwg = B * 16 * 16
stamps = torch.zeros(nwg * 16, dtype=torch.int64, device=dev)
assert lib.tgp_debug_set_gemm_stamps(stamps.data_ptr()) == 0
torch.cuda.synchronize()
run()
torch.cuda.synchronize()
st = stamps.view(-1, 16).cpu()
st = st[(st[:, 0] > 0) & (st[:, 2] > 0)]  # (tiles below the diagonal leave before the first stamp pair is complete)
t = st[:, :3].double() / 100.0
t = t - t[:, 0].min()


def q(v):
    v = v.sort()[0]
    n = v.numel()
    return " ".join(f"{float(v[int(f * (n - 1))]):7.2f}" for f in (0, 0.1, 0.5, 0.9, 1.0))


print(f"workgroups that multiplied: {st.size(0)}; last k-loop ends at {float(t[:, 2].max()):.2f} us")
print("                      min     p10     p50     p90     max   (us)")
print("start              ", q(t[:, 0]))
print("prologue duration  ", q(t[:, 1] - t[:, 0]))
print("k-loop duration    ", q(t[:, 2] - t[:, 1]))
hw, xcc = st[:, 4], st[:, 5] & 0xF
cu = (xcc << 16) | (hw & 0xFF00)
per_cu = {}
for i in range(st.size(0)):
    per_cu.setdefault(int(cu[i]), []).append(i)
counts = torch.tensor([len(v) for v in per_cu.values()])
print(f"distinct CUs {len(per_cu)}; workgroups per CU min / median / max {int(counts.min())} / {int(counts.median())} / {int(counts.max())}")
busy = torch.tensor([sum(float(t[i, 2] - t[i, 0]) for i in v) for v in per_cu.values()])
print(f"sum of workgroup lifetimes per CU: median {float(busy.median()):.1f} us, max {float(busy.max()):.1f} us (4 run side by side)")
