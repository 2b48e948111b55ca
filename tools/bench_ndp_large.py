"""NDPSelect (+ Reduce) on one N = 1M, E = 10M graph (BASELINE configs[3], the NDP half): the chip-wide LOBPCG
partition (tgp_ndp_large_*), wall time, steps, Rayleigh quotient, cut; no host eigen-solver may be called."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-geometric-pool_amd"))
import scipy.sparse.linalg as spla
from tgp import kernels as K
from tgp.reduce import BaseReduce
from tgp.select import NDPSelect


def boom(*a, **k):
    raise AssertionError("host eigen-solver called")


spla.eigsh = boom
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
a = torch.randint(0, n, (5 * n,), device=dev, generator=g)
b = torch.randint(0, n, (5 * n,), device=dev, generator=g)
keep = a != b
a, b = a[keep], b[keep]
ei = torch.stack([torch.cat([a, b]), torch.cat([b, a])])
ei = ei[:, torch.argsort(ei[0] * n + ei[1])].contiguous()
x = torch.randn(n, 128, device=dev, generator=g)
sel, red = NDPSelect(), BaseReduce()
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    so = sel(ei, None, num_nodes=n)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    xp, _ = red(x, so)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    info = int(so._partition_info[0])
    print(f"N={n} E={ei.size(1)}: NDPSelect {1e3 * (t1 - t0):8.2f} ms (LOBPCG steps {info if info >= 0 else 'random fallback'}), "
          f"kept {so.num_supernodes}, Reduce {1e3 * (t2 - t1):6.3f} ms", flush=True)
# the partition kernels alone, with their state
indptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
K.rowptr_from_sorted(ei[0], n, indptr)
k8 = torch.zeros(n, dtype=torch.uint8, device=dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
info, state = K.ndp_partition_large(indptr, ei[1], None, 0, n, 1, k8, status, want_state=True)
torch.cuda.synchronize()
print(f"partition kernels alone {1e3 * (time.perf_counter() - t0):8.2f} ms: {state}")
