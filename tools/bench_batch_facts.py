import os, sys, torch
sys.path.insert(0, "torch-geometric-pool_amd")
from tgp.utils import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
sizes = torch.randint(20, 61, (2048,), generator=g)
batch = torch.repeat_interleave(torch.arange(2048), sizes).to(dev)
copies = [batch.clone() for _ in range(64)]
for c in copies[:8]:
    ops.batch_info(c, topk_ratio=0.5)
torch.cuda.synchronize()
import time
t = time.perf_counter()
for c in copies[8:]:
    ops.batch_info(c, topk_ratio=0.5)
torch.cuda.synchronize()
print(f"batch_info on a new batch vector: {(time.perf_counter() - t) / 56 * 1e6:.1f} us per call (launch + host wait)")
