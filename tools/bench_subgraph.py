import os, sys, torch
sys.path.insert(0, "/root/repo/torch-geometric-pool_amd")
from tgp.connect import sparse_connect
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
n = 1_000_000
a = torch.randint(0, n, (5_000_000,), device=dev, generator=g); b = torch.randint(0, n, (5_000_000,), device=dev, generator=g)
ei = torch.stack([torch.cat([a, b]), torch.cat([b, a])]); ew = torch.rand(ei.size(1), device=dev, generator=g)
keep = torch.sort(torch.randperm(n, device=dev, generator=g)[: n // 2])[0]
k = keep.numel()
def run():
    return sparse_connect(ei, ew, node_index=keep, cluster_index=torch.arange(k, device=dev), num_nodes=n, num_supernodes=k)
for _ in range(3): out = run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): out = run()
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 20
E2 = out[0].size(1)
alg = ei.size(1) * 20 + n + k * 8 + E2 * 20
print(f"subgraph connect: {ms*1e3:.1f} us, E'={E2}, algorithmic {alg/1e6:.0f} MB -> {alg/ms/1e9:.2f} TB/s")
