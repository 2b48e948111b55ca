"""Training step (forward + backward) of a dense pooler on pre-batched inputs, eager vs captured as HIP graphs with
torch.cuda.make_graphed_callables: the native autograd Functions neither synchronise nor allocate outside torch's
allocator, so both passes replay without host work."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
import tgp  # noqa: E402
from tgp.poolers import get_pooler  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


class Step(torch.nn.Module):
    def __init__(self, pooler):
        super().__init__()
        self.pooler = pooler

    def forward(self, x, adj):
        out = self.pooler(x=x, adj=adj)
        loss = out.x.pow(2).mean() + out.edge_index.pow(2).mean()
        for v in out.loss.values():
            loss = loss + v
        return loss


def wall(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


tgp.freeze_gc()
CASES = (("mincut", 2048, 60, 20, 32), ("diff", 2048, 60, 20, 32), ("mincut", 512, 126, 32, 64), ("diff", 32, 1024, 128, 64))
if os.environ.get("GRAPHED_CASE"):
    CASES = (CASES[int(os.environ["GRAPHED_CASE"])],)
for alias, B, N, K, F in CASES:
    x = torch.randn(B, N, F, device=dev, generator=g, requires_grad=True)
    adj = (torch.rand(B, N, N, device=dev, generator=g) < 0.1).float()
    adj = torch.maximum(adj, adj.transpose(1, 2)).contiguous()
    eager = Step(get_pooler(alias, in_channels=F, k=K).to(dev))
    graphed_mod = Step(get_pooler(alias, in_channels=F, k=K).to(dev))
    graphed_mod.load_state_dict(eager.state_dict())

    def eager_step():
        eager(x, adj).backward()
        x.grad = None
        for p in eager.parameters():
            p.grad = None

    t_eager = wall(eager_step)
    try:
        graphed = torch.cuda.make_graphed_callables(graphed_mod, (x, adj), num_warmup_iters=3)

        def graphed_step():
            graphed(x, adj).backward()
            x.grad = None
            for p in graphed_mod.parameters():
                p.grad = None

        t_graph = wall(graphed_step)
        # same numbers?
        le = eager(x, adj)
        le.backward()
        ge = [p.grad.clone() for p in eager.parameters()]
        for p in eager.parameters():
            p.grad = None
        lg = graphed(x, adj)
        lg.backward()
        gg = [p.grad.clone() for p in graphed_mod.parameters()]
        same = torch.allclose(le, lg, rtol=1e-4, atol=1e-5) and all(torch.allclose(a, b, rtol=1e-3, atol=1e-4) for a, b in zip(ge, gg))
        print(f"{alias:7s} B={B:5d} N={N:5d} K={K:4d} F={F:3d}: eager fwd+bwd {t_eager:7.3f} ms   graphed {t_graph:7.3f} ms   same={same}", flush=True)
    except Exception as ex:  # noqa: BLE001
        print(f"{alias:7s} B={B:5d} N={N:5d}: eager {t_eager:7.3f} ms   capture FAILED: {type(ex).__name__}: {str(ex)[:200]}", flush=True)
