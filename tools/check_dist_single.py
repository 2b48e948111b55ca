"""torchrun --nproc-per-node 1 smoke of the RCCL gather path (world_size 1, forced through the collectives)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-geometric-pool_amd"))
from tgp import distributed as D  # noqa: E402

dev = torch.device(f"cuda:{int(os.environ.get('LOCAL_RANK', 0))}")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
D._world = lambda group=None: 2 if os.environ.get("FAKE_WORLD") else dist.get_world_size()  # exercise the collective code
x = torch.randn(5, 7, 3, device=dev)
a = torch.randn(5, 7, 7, device=dev)
try:
    gx, ga = D.all_gather_dense([x, a])
    print("dense gather ok", gx.shape, ga.shape)
except Exception as e:  # world=1 buffers sized for FAKE_WORLD=2 cannot be gathered; report and continue
    print("dense gather (fake world) raised:", type(e).__name__, str(e)[:120])
D._world = lambda group=None: dist.get_world_size()
buf = torch.empty(1 * 5, 7, 3, device=dev)
dist.all_gather_into_tensor(buf, x)
assert torch.equal(buf, x)
t = torch.tensor([1.5], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
# bucketed, asynchronous PackedGather through RCCL (one rank, collective forced), interleaved with compute
os.environ["TGP_FORCE_COLLECTIVE"] = "1"
pg = D.PackedGather(bucket_steps=4)
got = []
for j in range(10):
    y = (torch.full((32, 128, 64), float(j), device=dev), torch.full((32, 128, 128), float(-j), device=dev))
    pg.start(list(y))
    torch.mm(torch.randn(512, 512, device=dev), torch.randn(512, 512, device=dev))  # something to overlap with
    got.extend(pg.take_ready())
got.extend(pg.flush())
torch.cuda.synchronize()
assert len(got) == 10 and all(float(g[0][0, 0, 0]) == j and float(g[1][-1, -1, -1]) == -j for j, g in enumerate(got))
assert got[3][0].shape == (32, 128, 64) and got[3][1].shape == (32, 128, 128)
print("bucketed PackedGather over RCCL ok")
print("rccl collectives ok on", torch.cuda.get_device_name(dev))
dist.destroy_process_group()
