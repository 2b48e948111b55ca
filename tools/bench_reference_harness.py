"""The reference's own timing harness configuration (examples/time_and_mem_test.py:44-55, :296, :335-440) on this build:
a batch of 4 graphs with 50..3000 nodes (alternating Erdos-Renyi p = 0.01 and Barabasi-Albert m = 1), F = 16, poolers
in train mode with ratio 0.1 / k = 10 % of the average graph size, 2 warm-up steps, then 10 iterations of
forward (timed), loss = out.x.sum() + auxiliary losses, backward (timed), with the peak device memory of each pass.
The reference publishes no output of that script; this prints the table it would print, for the five poolers in scope.

    python tools/bench_reference_harness.py [--seed 42] [--iterations 10]
"""
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))
from tgp.poolers import get_pooler  # noqa: E402

NUM_GRAPHS, MIN_SIZE, MAX_SIZE, F_DIM = 4, 50, 3000, 16


def erdos_renyi(n, p, rng):
    m = rng.binomial(n * (n - 1) // 2, p)
    r = rng.integers(0, n, size=2 * m + 16)
    c = rng.integers(0, n, size=2 * m + 16)
    keep = r < c
    pairs = np.unique(np.stack([r[keep], c[keep]], 1), axis=0)[:m]
    return pairs


def barabasi_albert(n, rng):  # m = 1: every new node attaches to one earlier node, chosen by degree
    targets = [0, 1]
    pairs = [(0, 1)]
    for v in range(2, n):
        u = targets[rng.integers(0, len(targets))]
        pairs.append((u, v))
        targets += [u, v]
    return np.asarray(pairs)


def make_batch(seed, dev):
    random.seed(seed)
    rng = np.random.default_rng(seed)
    xs, eis, batch, off = [], [], [], 0
    sizes = []
    for i in range(NUM_GRAPHS):
        n = random.randint(MIN_SIZE, MAX_SIZE)
        pairs = erdos_renyi(n, 0.01, rng) if i % 2 == 0 else barabasi_albert(n, rng)
        both = np.concatenate([pairs, pairs[:, ::-1]], 0)
        both = both[np.lexsort((both[:, 1], both[:, 0]))]
        eis.append(torch.from_numpy(both.T.copy()).long() + off)
        xs.append(torch.from_numpy(rng.standard_normal((n, F_DIM)).astype(np.float32)))
        batch.append(torch.full((n,), i, dtype=torch.long))
        off += n
        sizes.append(n)
    ei = torch.cat(eis, 1).to(dev)
    return torch.cat(xs).to(dev), ei, torch.ones(ei.size(1), device=dev), torch.cat(batch).to(dev), sizes


def main():
    seed = int(sys.argv[sys.argv.index("--seed") + 1]) if "--seed" in sys.argv else 42
    iters = int(sys.argv[sys.argv.index("--iterations") + 1]) if "--iterations" in sys.argv else 10
    dev = torch.device("cuda:0")
    x, ei, ew, batch, sizes = make_batch(seed, dev)
    k = max(1, int(x.size(0) / NUM_GRAPHS * 0.1))
    print(f"batch: {NUM_GRAPHS} graphs of {sizes} nodes ({x.size(0)} nodes, {ei.size(1)} directed edges), F = {F_DIM}, "
          f"k = {k}, ratio = 0.1, {iters} iterations after 2 warm-ups, train mode")
    print(f"{'pooler':10s} {'forward ms':>11s} {'backward ms':>12s} {'fwd peak MB':>12s} {'bwd peak MB':>12s}")
    cfgs = {"topk": dict(in_channels=F_DIM, ratio=0.1), "graclus": dict(), "ndp": dict(),
            "diff": dict(in_channels=F_DIM, k=k), "mincut": dict(in_channels=F_DIM, k=k)}
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    for name, kw in cfgs.items():
        if only is not None and name != only:
            continue
        torch.manual_seed(seed)
        pooler = get_pooler(name, **kw).to(dev).train()

        def fwd(xin):
            return pooler(x=xin, adj=ei, edge_weight=ew, batch=batch)

        def loss_of(out):
            loss = out.x.sum()
            if out.loss:
                loss = loss + sum(out.loss.values())
            return loss

        for _ in range(2):
            pooler.zero_grad(set_to_none=True)
            xin = x.clone().requires_grad_(True)
            loss_of(fwd(xin)).backward()
        torch.cuda.synchronize()
        tf, tb, mf, mb = [], [], 0, 0
        for _ in range(iters):
            pooler.zero_grad(set_to_none=True)
            xin = x.clone().requires_grad_(True)
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats(dev)
            base = torch.cuda.memory_allocated(dev)
            t0 = time.perf_counter()
            out = fwd(xin)
            torch.cuda.synchronize()
            tf.append(time.perf_counter() - t0)
            mf = max(mf, torch.cuda.max_memory_allocated(dev) - base)
            loss = loss_of(out)
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats(dev)
            base = torch.cuda.memory_allocated(dev)
            t0 = time.perf_counter()
            loss.backward()
            torch.cuda.synchronize()
            tb.append(time.perf_counter() - t0)
            mb = max(mb, torch.cuda.max_memory_allocated(dev) - base)
            del out, loss, xin
        print(f"{name:10s} {np.mean(tf) * 1e3:11.3f} {np.mean(tb) * 1e3:12.3f} {mf / 2**20:12.1f} {mb / 2**20:12.1f}", flush=True)


if __name__ == "__main__":
    main()
