import os, sys, torch
sys.path.insert(0, "/root/repo/torch-geometric-pool_amd"); sys.path.insert(0, "/root/repo/tests")
from test_gpu_kron import make_batch
from tgp.select import NDPSelect
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
sizes = torch.randint(20, 61, (2048,), generator=g).tolist()
ei, ew, batch, _ = make_batch(sizes, seed=21)
so = NDPSelect()(edge_index=ei.to(dev), edge_weight=ew.to(dev), batch=batch.to(dev), num_nodes=batch.numel())
info = so._partition_info.cpu().float()
it = info[info >= 0]
print("graphs", info.numel(), "random fallback", int((info < 0).sum()), "iterations: median", float(it.median()), "p90", float(it.quantile(0.9)), "max", float(it.max()), "at cap (3000)", int((it >= 3000).sum()))
