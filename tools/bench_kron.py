"""KronConnect on a PROTEINS-sized batch (2048 graphs, n ~ U[20,60]) and with a few large graphs mixed in: the
block-batched kernel vs the whole-batch scipy route of the reference (host), same SelectOutput."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-geometric-pool_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from test_gpu_kron import make_batch, so_of
import tgp_oracle as O
from tgp.connect import KronConnect

dev = torch.device("cuda:0")
NO_REF = "--no-reference" in sys.argv  # skip the ~70 s host route (profiling runs)
for label, extra in (("2048 graphs n~U[20,60]", []), ("+ graphs of 200, 400, 620, 1000 nodes", [200, 400, 620, 1000])):
    g = torch.Generator().manual_seed(0)
    sizes = torch.randint(20, 61, (2048,), generator=g).tolist() + extra
    ei, ew, batch, idx_pos = make_batch(sizes, seed=1)
    n = batch.numel()
    L = O.laplacian_scipy(ei, ew.double(), n).astype(np.float32)
    so = so_of(idx_pos, n, dev, L=L)
    eid, ewd, bd = ei.to(dev), ew.to(dev), batch.to(dev)
    conn = KronConnect()
    for _ in range(3):
        out = conn(eid, so, edge_weight=ewd, batch=bd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        out = conn(eid, so, edge_weight=ewd, batch=bd)
    torch.cuda.synchronize()
    t_dev = (time.perf_counter() - t0) / reps
    if NO_REF:
        print(f"{label}: N={n} E={ei.size(1)} kept={idx_pos.numel()} edges_out={out[0].size(1)}  native {t_dev*1e3:.3f} ms/call")
        continue
    t0 = time.perf_counter()
    ref = O.kron_connect(L, idx_pos)
    t_host = time.perf_counter() - t0
    print(f"{label}: N={n} E={ei.size(1)} kept={idx_pos.numel()} edges_out={out[0].size(1)}  native {t_dev*1e3:.3f} ms/call"
          f"  | reference route (scipy spsolve, host) {t_host*1e3:.1f} ms  same edges: {torch.equal(out[0].cpu(), ref[0])}")
