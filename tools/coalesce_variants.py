"""One variant of the coalesce Connect's gather-sort kernel at C4 (row-sorted, N = 1 M, E = 10 M, Graclus assignment),
launched a few times so that a profiler pass (`rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum`, or --kernel-trace) sees it.

    python3 tools/coalesce_variants.py VARIANT [--time]

VARIANT: full | col32 (the int32 columns handed over with the CSR offsets: product code) | and, with the diagnostic
build (full_diag / col32_diag = the same two on that build: its instrumentation costs time, compare within it), (`make -C torch-geometric-pool_amd/csrc stamps`): no_table (key = node id / 2, no look-up), dummy4 / dummy2 /
dummy1 (that key plus a dropped look-up into 4- / 2- / 1-byte entries: table footprint 4 / 2 / 1 MB), nt_table
(non-temporal look-ups), no_edges (synthetic node ids, look-ups stay).  Ablated variants give wrong edges: timing and
counters only.  --time: median-free event timing of the whole Connect call instead (20 calls)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
ABLATE = {"full_diag": 0, "col32_diag": 0, "no_table": 4, "dummy4": 256, "dummy2": 512, "dummy1": 1024, "nt_table": 16, "no_edges": 8}
variant = sys.argv[1] if len(sys.argv) > 1 else "full"
if variant in ABLATE:
    os.environ["TGP_HIP_LIB"] = os.path.join(ROOT, "torch-geometric-pool_amd", "lib", "libtgp_hip_stamps.so")
sys.path.insert(0, os.path.join(ROOT, "torch-geometric-pool_amd"))
from tgp import _native, kernels  # noqa: E402
from tgp.select import GraclusSelect  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
n = 1_000_000
a = torch.randint(0, n, (5_000_000,), device=dev, generator=g)
b = torch.randint(0, n, (5_000_000,), device=dev, generator=g)
keep = a != b
a, b = a[keep], b[keep]
ei = torch.stack([torch.cat([a, b]), torch.cat([b, a])])
ei = ei[:, torch.argsort(ei[0] * n + ei[1])].contiguous()
ew = torch.rand(ei.size(1), device=dev, generator=g) + 0.5
so = GraclusSelect()(ei, ew, num_nodes=n)
k = so.num_supernodes
idx = so.assign_index()
csr = (so.edge_csr_for(ei), None)  # what SparseConnect hands over: the offsets GraclusSelect built for this very list
assert csr[0] is not None
if variant in ("col32", "col32_diag"):
    csr = (csr[0], ei[1].to(torch.int32).contiguous())
lib = _native.lib()
if variant in ABLATE:
    lib.tgp_debug_set_gs_ablate.argtypes = [ctypes.c_int]
    assert lib.tgp_debug_set_gs_ablate(ABLATE[variant]) == 0


def call():
    return kernels.coalesce_edges(ei, ew, so.cluster_index, k, "sum", True, assign_index=idx, csr=csr)


for _ in range(3):
    call()
torch.cuda.synchronize()
if "--time" in sys.argv:
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20):
        call()
    t1.record()
    torch.cuda.synchronize()
    print(f"{variant:9s} whole Connect call {t0.elapsed_time(t1) / 20 * 1e3:7.1f} us   E = {ei.size(1)}  K = {k}")
else:
    for _ in range(5):
        call()
    torch.cuda.synchronize()
