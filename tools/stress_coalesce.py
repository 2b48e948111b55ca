#!/usr/bin/env python3
"""Randomised stress of the coalesce Connect routes against each other (row-local / grouped / general radix route must agree
bit for bit) and of the assignment index against a stable argsort.  usage: python tools/stress_coalesce.py [cases]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
from tgp import kernels  # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
bad = 0
for case in range(cases):
    rng = random.Random(case)
    g = torch.Generator().manual_seed(case)
    n = rng.choice([3, 50, 700, 5_000, 40_000, 200_000])
    e = rng.choice([0, 1, 7, n, 4 * n, 12 * n])
    shape = rng.choice(["pairs", "random", "few_big", "many_empty"])
    if shape == "pairs":
        k = max(1, n // 2)
        cluster = (torch.randperm(n, generator=g) // 2).clamp(max=k - 1)
    elif shape == "random":
        k = max(1, rng.choice([n // 3, n // 10, n]))
        cluster = torch.randint(0, k, (n,), generator=g)
    elif shape == "few_big":
        k = max(1, min(n, rng.choice([2, 5, 40])))
        cluster = torch.randint(0, k, (n,), generator=g)
    else:
        k = 2 * n + 5
        cluster = torch.randint(0, max(1, n // 4), (n,), generator=g) * 3
    ei = torch.randint(0, n, (2, e), generator=g)
    if rng.random() < 0.7 and e > 0:
        ei = ei[:, torch.argsort(ei[0], stable=True)]
    ew = (torch.rand(e, generator=g) - 0.3) if rng.random() < 0.7 else None
    if ew is not None and e:
        ew[torch.rand(e, generator=g) < 0.1] = 0.0
    op = rng.choice(["sum", "mean", "min", "max", "mul"])
    rsl = rng.random() < 0.5
    cl = cluster.to(dev)
    idx = kernels.build_assign_index(cl, k)
    order = torch.argsort(cluster, stable=True)
    ok = torch.equal(idx.perm[:n].cpu().long(), order)
    a = kernels.coalesce_edges(ei.to(dev), None if ew is None else ew.to(dev), cl, k, op, rsl, assign_index=idx)
    b = kernels.coalesce_edges(ei.to(dev), None if ew is None else ew.to(dev), cl, k, op, rsl)
    ok = ok and torch.equal(a[0], b[0]) and ((a[1] is None and b[1] is None) or torch.equal(a[1], b[1]))
    if e:
        # reference semantics in plain torch (connect/base_conn.py:83-89): unique (row, col) pairs of the relabelled list
        r, c = cluster[ei[0]], cluster[ei[1]]
        key = r * k + c
        uk = torch.unique(key)
        if rsl:
            uk = uk[(uk // k) != (uk % k)]
        if ew is None:
            ok = ok and torch.equal(a[0].cpu(), torch.stack([uk // k, uk % k]))
    if not ok:
        bad += 1
        print("MISMATCH case", case, n, e, shape, k, op, rsl, flush=True)
print(f"{cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
