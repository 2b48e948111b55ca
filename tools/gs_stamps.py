"""Diagnostic: phase times inside cr_gather_sort_kernel (needs `make -C torch-geometric-pool_amd/csrc stamps`)."""
import ctypes, os, sys
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
os.environ["TGP_HIP_LIB"] = os.path.join(ROOT, "torch-geometric-pool_amd", "lib", "libtgp_hip_stamps.so")
sys.path.insert(0, os.path.join(ROOT, "torch-geometric-pool_amd"))
from tgp import _native, kernels
from tgp.select import GraclusSelect
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
n = 1_000_000
a = torch.randint(0, n, (5_000_000,), device=dev, generator=g); b = torch.randint(0, n, (5_000_000,), device=dev, generator=g)
keep = a != b; a, b = a[keep], b[keep]
ei = torch.stack([torch.cat([a, b]), torch.cat([b, a])]); ei = ei[:, torch.argsort(ei[0] * n + ei[1])]
ew = torch.ones(ei.size(1), device=dev)
so = GraclusSelect()(ei, ew, num_nodes=n)
k = so.num_supernodes
idx = so.assign_index()
for _ in range(2):
    kernels.coalesce_edges(ei, ew, so.cluster_index, k, "sum", True, assign_index=idx)
nwg = (k + 31) // 32
st = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
lib = _native.lib()
lib.tgp_debug_set_gs_stamps.argtypes = [ctypes.c_void_p]
assert lib.tgp_debug_set_gs_stamps(st.data_ptr()) == 0
torch.cuda.synchronize()
kernels.coalesce_edges(ei, ew, so.cluster_index, k, "sum", True, assign_index=idx)
torch.cuda.synchronize()
s = st.view(-1, 8).cpu().double() / 100.0
names = ["(a) segments -> LDS", "(b) gather", "(c1) rows<=32", "(c1b) rows 33..64 + tail", "-", "row offsets"]
print("workgroups", nwg, "mean us per workgroup:")
for i, nm in enumerate(names):
    print(f"  {nm:22s} {float(s[:, i].mean()):7.2f}  (max {float(s[:, i].max()):7.2f})")
def timed(mode):
    assert lib.tgp_debug_set_gs_ablate(mode) == 0
    for _ in range(3):
        kernels.coalesce_edges(ei, ew, so.cluster_index, k, "sum", True, assign_index=idx)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(20):
        kernels.coalesce_edges(ei, ew, so.cluster_index, k, "sum", True, assign_index=idx)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 20 * 1e3
assert lib.tgp_debug_set_gs_stamps(0) == 0
base = timed(0)
print(f"whole call us: full {base:.1f}; no sort {timed(1):.1f}; no edge/table loads {timed(2):.1f}; neither {timed(3):.1f}; no table {timed(4):.1f}; no col/w {timed(8):.1f}; nt table {timed(16):.1f}; dummy 4-byte {timed(256):.1f}; dummy 2-byte {timed(512):.1f}; dummy 1-byte {timed(1024):.1f}")
lib.tgp_debug_set_gs_ablate(0)
t0 = s[:, 7].min()
print("start spread", float((s[:, 7] - t0).max()), "us; total per WG", float(s[:, :6].sum(1).mean()))
