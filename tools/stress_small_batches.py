#!/usr/bin/env python3
"""Randomised cross-check of the one-launch paths for batches of small graphs (round 4) against the staged operators:
TopK / Graclus poolers with and without the one-launch kernels (tgp_sparse_pool_small_f32, tgp_graclus_match_graphs_fused,
the per-graph offsets hand-over), random batch shapes, degrees, duplicate entries, zero weights, missing weights.
usage: python tools/stress_small_batches.py [cases]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
from tgp import kernels  # noqa: E402
from tgp.poolers import get_pooler  # noqa: E402
from tgp.src import SRCPooling  # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0


def batch_of(rng, g):
    B = rng.choice([2, 3, 17, 256, 257, 1000, 2048])
    lo = rng.choice([1, 2, 10, 30])
    hi = rng.choice([lo, 20, 40, 64])
    hi = max(hi, lo)
    deg = rng.choice([0, 2, 4, 8, 14])
    sizes = torch.randint(lo, hi + 1, (B,), generator=g)
    n = int(sizes.sum())
    batch = torch.repeat_interleave(torch.arange(B), sizes)
    start = torch.cumsum(sizes, 0) - sizes
    src = torch.arange(n).repeat_interleave(max(deg // 2, 1))
    dst = start[batch[src]] + (torch.rand(src.numel(), generator=g) * sizes[batch[src]]).long()
    keep = (src != dst) if rng.random() < 0.8 else torch.ones_like(src, dtype=torch.bool)  # sometimes self loops stay
    if deg == 0:
        keep = torch.zeros_like(keep)
    src, dst = src[keep], dst[keep]
    key = torch.cat([src * n + dst, dst * n + src])
    key = torch.sort(key)[0] if rng.random() < 0.3 else torch.unique(key)
    ei = torch.stack([key // n, key % n])
    f = rng.choice([1, 4, 7, 32, 64])
    x = torch.randn(n, f, generator=g)
    ew = torch.rand(ei.size(1), generator=g) + 0.25
    ew[torch.rand(ei.size(1), generator=g) < 0.1] = 0.0
    if rng.random() < 0.3:
        ew = None
    return x.to(dev), ei.to(dev), None if ew is None else ew.to(dev), batch.to(dev), f


def same(a, b):
    if a is None or b is None:
        return a is None and b is None
    return a.shape == b.shape and torch.equal(a, b)


for case in range(cases):
    rng = random.Random(case)
    g = torch.Generator().manual_seed(case)
    torch.manual_seed(case)
    x, ei, ew, batch, f = batch_of(rng, g)
    if ei.size(1) == 0:
        continue
    alias = rng.choice(["topk", "graclus"])
    kw = dict(in_channels=f, ratio=rng.choice([0.25, 0.5, 0.8])) if alias == "topk" else {}
    kw.update(remove_self_loops=rng.random() < 0.7, degree_norm=rng.random() < 0.3, edge_weight_norm=rng.random() < 0.3)
    pooler = get_pooler(alias, **kw).to(dev).eval()
    with torch.no_grad():
        fused = pooler(x=x, adj=ei, edge_weight=ew, batch=batch)
        # the staged reference: no one-launch kernels at all
        real_rc, gf, gp = SRCPooling.reduce_connect, kernels._GRACLUS_FUSED, kernels._SPS_GIVE_PTRS
        SRCPooling.reduce_connect = lambda self, *a, **k: None
        kernels._GRACLUS_FUSED, kernels._SPS_GIVE_PTRS = False, False
        try:
            staged = pooler(x=x, adj=ei.clone(), edge_weight=ew, batch=batch)
        finally:
            SRCPooling.reduce_connect = real_rc
            kernels._GRACLUS_FUSED, kernels._SPS_GIVE_PTRS = gf, gp
    ok = (same(fused.x, staged.x) and same(fused.edge_index, staged.edge_index)
          and same(fused.edge_weight, staged.edge_weight) and same(fused.batch, staged.batch)
          and same(fused.so.cluster_index, staged.so.cluster_index))
    if not ok:
        bad += 1
        print(f"case {case}: MISMATCH alias={alias} kw={kw} B={int(batch.max()) + 1} n={x.size(0)} E={ei.size(1)} f={f}",
              flush=True)
print(f"{cases} cases, {bad} mismatches")
