"""Diagnostic: where the fixed per-launch cost of the fp32 MFMA GEMM goes.  Needs the stamps build
(`make -C torch-geometric-pool_amd/csrc stamps`).  Prints, for U = A S at the C2 shape, the distribution over
workgroups of: start time, prologue, k-loop, epilogue, finish time; grouped by XCD."""
import ctypes
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
lib = ctypes.CDLL(os.path.join(ROOT, "torch-geometric-pool_amd", "lib", "libtgp_hip_stamps.so"))
p, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
lib.tgp_bmm_f32.argtypes = [p, p, p, i64, i64, i64, i64, ci, i64, i64, i64, i64, i64, i64, p]
lib.tgp_debug_set_gemm_stamps.argtypes = [p]

B, N, K = (32, 1024, 128) if len(sys.argv) < 4 else (int(v) for v in sys.argv[1:4])
dev = torch.device("cuda:0")
torch.manual_seed(0)
stream = torch.cuda.current_stream(dev).cuda_stream
if os.environ.get("STAGE2"):
    # the second product of the dense path, S^T [U | X] over a slice of N: C[b] = A[b]^T Bm[b], A stored [Kd, M]
    Kd, M, Nc = (int(v) for v in os.environ["STAGE2"].split(","))
    A = torch.rand(B, Kd, M, device=dev)
    S = torch.rand(B, Kd, Nc, device=dev)
    U = torch.empty(B, M, Nc, device=dev)

    def run():
        rc = lib.tgp_bmm_f32(A.data_ptr(), S.data_ptr(), U.data_ptr(), B, M, Nc, Kd, 1, M, Nc, Nc, Kd * M, Kd * Nc, M * Nc, stream)
        assert rc == 0
    N, K = M, Nc
else:
    A = (torch.rand(B, N, N, device=dev) < 0.01).float()
    S = torch.softmax(torch.randn(B, N, K, device=dev), -1)
    U = torch.empty(B, N, K, device=dev)

    def run():
        rc = lib.tgp_bmm_f32(A.data_ptr(), S.data_ptr(), U.data_ptr(), B, N, K, N, 0, N, K, K, N * N, N * K, N * K, stream)
        assert rc == 0


for _ in range(3):
    run()
nwg_max = B * ((N + 63) // 64) * ((K + 63) // 64)
stamps = torch.zeros(nwg_max * 16, dtype=torch.int64, device=dev)
assert lib.tgp_debug_set_gemm_stamps(stamps.data_ptr()) == 0
if os.environ.get("REVERSE"):
    assert lib.tgp_debug_set_gemm_reverse(1) == 0
torch.cuda.synchronize()
run()
torch.cuda.synchronize()
st = stamps.view(-1, 16).cpu()
st = st[st[:, 0] > 0]
t = st[:, :4].double() / 100.0  # us (100 MHz counter)
t0 = t[:, 0].min()
t = t - t0
xcc = st[:, 5] & 0xF
print(f"workgroups {st.size(0)}; kernel span {float(t[:, 3].max()):.2f} us")


def q(v):
    v = v.sort()[0]
    n = v.numel()
    return " ".join(f"{float(v[int(f * (n - 1))]):7.2f}" for f in (0, 0.1, 0.5, 0.9, 1.0))


print("                      min     p10     p50     p90     max   (us)")
print("start              ", q(t[:, 0]))
print("prologue duration  ", q(t[:, 1] - t[:, 0]))
print("k-loop duration    ", q(t[:, 2] - t[:, 1]))
print("epilogue duration  ", q(t[:, 3] - t[:, 2]))
print("finish             ", q(t[:, 3]))
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print(f"xcd {x}: n={int(m.sum()):4d} start p50 {float(t[m, 0].median()):6.2f} loop p50 {float((t[m, 2] - t[m, 1]).median()):6.2f} "
          f"finish max {float(t[m, 3].max()):6.2f}")

# co-residency: workgroups that ran on the same CU (xcc, se, sh, cu from HW_ID)
hw = st[:, 4]
cu_key = (xcc << 16) | (hw & 0xFF00)
loop = (t[:, 2] - t[:, 1])
groups = {}
for i in range(st.size(0)):
    groups.setdefault(int(cu_key[i]), []).append(i)
sizes = {}
for k, v in groups.items():
    sizes[len(v)] = sizes.get(len(v), 0) + 1
print("CUs by number of co-resident workgroups:", sizes, "distinct CUs:", len(groups))
by_n = {}
for k, v in groups.items():
    by_n.setdefault(len(v), []).extend(float(loop[i]) for i in v)
for n, vals in sorted(by_n.items()):
    vv = torch.tensor(vals)
    print(f"  {n} WG/CU: loop mean {float(vv.mean()):.2f} min {float(vv.min()):.2f} max {float(vv.max()):.2f}")
# logical tile position of the slowest / fastest workgroups
order = loop.argsort()
print("fastest block ids:", [int(i) for i in order[:8]], "slowest:", [int(i) for i in order[-8:]])

# progress inside the loop: first half / third quarter / last quarter, by residency generation
half = (st[:, 6].double() / 100.0 - t0) - t[:, 1]
q3 = (st[:, 7].double() - st[:, 6].double()) / 100.0
q4 = t[:, 2] - (st[:, 7].double() / 100.0 - t0)
ids = torch.arange(st.size(0))
for name, m in (("gen1 (id < 256)", ids < 256), ("gen2 (id >= 256)", ids >= 256)):
    print(f"{name}: first half {float(half[m].mean()):.2f}  third quarter {float(q3[m].mean()):.2f}  last quarter {float(q4[m].mean()):.2f} us")
pairs = [v for v in groups.values() if len(v) == 2]
print("sample CU pairs (block ids, loop us):", [(p_[0], p_[1], round(float(loop[p_[0]]), 1), round(float(loop[p_[1]]), 1)) for p_ in pairs[:6]])


# phases of one k-step (t = nk/2), wave 0 of every workgroup: durations in ns
phs = st[:, 8:15].double() * 10.0
names = ["MFMA pairs 0..STORE_AT (+ first operand fetch)", "LDS stage stores (incl. wait for the tile)", "MFMA pairs ..LOAD_AT",
         "global loads issue", "MFMA pairs ..15", "barrier wait"]
for name, m in (("gen1", ids < 256), ("gen2", ids >= 256)):
    if int(m.sum()) == 0:
        continue
    print(name, "k-step phases (mean ns):", ", ".join(f"{n}: {float((phs[m, i + 1] - phs[m, i]).mean()):.0f}" for i, n in enumerate(names)),
          f"| total {float((phs[m, 6] - phs[m, 0]).mean()):.0f}")
