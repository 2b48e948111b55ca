"""Forward + backward of whole pooler calls on sparse PyG-style inputs: wall time per step, kernels per step, and the
kernels that take the time (torch.profiler).

    python tools/e2e_train_step.py [mincut_c3] [diff_c3] [diff_c2] [mincut_c2] [--top 25]
"""
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from e2e_launches import CASES, batch_graphs, resolve_sizes, wall  # noqa: E402
from tgp.poolers import get_pooler  # noqa: E402

dev = torch.device("cuda:0")


def main():
    args = [a for a in sys.argv[1:] if a in CASES]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 0
    seq = "--sequence" in sys.argv  # kernels of one step in launch order
    for name in (args or ["mincut_c3", "diff_c3", "mincut_c2", "diff_c2"]):
        alias, kw, sizes, deg, f = CASES[name]
        sizes = resolve_sizes(sizes)
        x, ei, batch = batch_graphs(sizes, deg, f)
        x.requires_grad_(True)
        pooler = get_pooler(alias, **kw).to(dev).train()

        def fwd():
            with torch.no_grad():
                return pooler(x=x, adj=ei, batch=batch)

        fresh = "--fresh" in sys.argv  # new edge_index / batch objects every step (mini-batch training: memos miss)

        def step():
            pooler.zero_grad(set_to_none=True)
            x.grad = None
            out = pooler(x=x, adj=ei.clone(), batch=batch.clone()) if fresh else pooler(x=x, adj=ei, batch=batch)
            if "--bench-loss" in sys.argv and out.loss:  # the loss of bench.py's e2e_train_* lines
                (out.x.sum() + out.edge_index.sum() + sum(out.loss.values())).backward()
                return
            loss = out.x.square().sum()
            if out.edge_weight is not None and out.edge_weight.requires_grad:
                loss = loss + out.edge_weight.square().sum()
            elif torch.is_tensor(out.edge_index) and out.edge_index.is_floating_point():
                loss = loss + out.edge_index.square().sum()
            if out.loss:
                loss = loss + sum(out.loss.values())
            loss.backward()

        ms_f, ms = wall(fwd), wall(step, iters=20)
        from torch.profiler import ProfilerActivity, profile
        step()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step()
            torch.cuda.synchronize()
        evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
        busy = sum(e.device_time for e in evs) / 1e3
        print(f"{name:12s} forward {ms_f:7.3f} ms   forward+backward {ms:7.3f} ms   kernels/step {len(evs)}   "
              f"GPU-busy {busy:7.3f} ms", flush=True)
        if seq:
            for e in sorted(evs, key=lambda e: e.time_range.start):
                print(f"      {e.device_time:7.1f} us  {e.name[:130]}")
        if top:
            agg = defaultdict(lambda: [0, 0.0])
            for e in evs:
                agg[e.name][0] += 1
                agg[e.name][1] += e.device_time
            for nm, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
                print(f"      {t:8.1f} us  x{n:<3d} {nm[:120]}")


if __name__ == "__main__":
    main()
