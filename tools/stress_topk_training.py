"""Randomised cross-check of TopkPooling's one-node training path (functions._TopkPoolTrainFn + tgp_topk_pool_bwd_f32)
against the operator-by-operator autograd graph: random batches (small graphs, one large graph, no batch vector), ratios,
widths, activations, multipliers, with and without edge weights; outputs and every gradient compared.
    python tools/stress_topk_training.py [cases]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
import tgp.poolers as P  # noqa: E402
import tgp.src as S  # noqa: E402
from tgp.poolers import get_pooler  # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = random.Random(0)
bad = 0
for case in range(cases):
    g = torch.Generator().manual_seed(case)
    F = rng.choice([4, 8, 16, 32, 64, 128, 20, 256])
    kind = rng.choice(["small", "small", "large", "nobatch"])
    if kind == "small":
        sizes = torch.randint(1, 65, (rng.randint(2, 300),), generator=g)
    else:
        sizes = torch.tensor([rng.randint(100, 5000)])
    xs, eis, bs, off = [], [], [], 0
    for gi, n in enumerate(sizes.tolist()):
        m = rng.randint(0, 4 * n)
        if m:
            e = torch.randint(0, n, (2, m), generator=g)
            e = torch.unique(e[0] * n + e[1])
            eis.append(torch.stack([e // n, e % n]) + off)
        bs.append(torch.full((n,), gi))
        off += n
    n_tot = off
    x0 = torch.randn(n_tot, F, generator=g).to(dev)
    ei = (torch.cat(eis, 1) if eis else torch.zeros(2, 0, dtype=torch.long)).to(dev)
    ew = (torch.rand(ei.size(1), generator=g) + 0.1).to(dev) if rng.random() < 0.5 else None
    batch = None if kind == "nobatch" else torch.cat(bs).to(dev)
    kw = dict(ratio=rng.choice([0.1, 0.3, 0.5, 0.8, 1, 3]), nonlinearity=rng.choice(["tanh", "identity"]),
              multiplier=rng.choice([1.0, 1.0, 2.5]))
    torch.manual_seed(case)
    pooler = get_pooler("topk", in_channels=F, **kw).to(dev).train()
    x_grad = rng.random() < 0.8

    def step(fold):
        P._FOLD_TRAINING = fold
        S._FOLD_TRAINING = fold
        pooler.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(x_grad)
        out = pooler(x=x, adj=ei, edge_weight=ew, batch=batch)
        lifted = pooler(x=out.x, so=out.so, lifting=True)
        loss = out.x.square().sum() + (lifted * x0).sum() * 0.3 + (out.so.s.coalesce().values() ** 3).sum() * 0.1
        loss.backward()
        return out.x.detach(), out.edge_index, out.edge_weight, x.grad, pooler.selector.weight.grad.clone()

    a, b = step(True), step(False)
    ok = a[0].shape == b[0].shape and torch.equal(a[1], b[1])
    ok = ok and torch.allclose(a[0], b[0], rtol=1e-4, atol=1e-5)
    ok = ok and ((a[2] is None and b[2] is None) or torch.equal(a[2], b[2]))
    if x_grad:
        ok = ok and torch.allclose(a[3], b[3], rtol=1e-3, atol=1e-4 * max(1.0, float(b[3].abs().max())))
    ok = ok and torch.allclose(a[4], b[4], rtol=2e-3, atol=2e-4 * max(1.0, float(b[4].abs().max())))
    if not ok:
        bad += 1
        print(f"case {case}: MISMATCH kind={kind} F={F} kw={kw} weighted={ew is not None} x_grad={x_grad}")
print(f"{cases} cases, {bad} mismatches")
