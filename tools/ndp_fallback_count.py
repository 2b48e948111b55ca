"""How many graphs of the ownership test's batch take NDPSelect's random fallback (cut < 0.5), with and without the Lanczos
warm start of the one-wave kernel (child process with TGP_NDP_LANCZOS=0)."""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-geometric-pool_amd"))


def _er_batch(num_graphs, lo, hi, f, seed, dev):
    """The batch of tests/test_gpu_sparse_pool_small.py: Erdos-Renyi graphs of lo..hi nodes, ~4 neighbours per node."""
    g = torch.Generator().manual_seed(seed)
    xs, eis, bs, off = [], [], [], 0
    for gi in range(num_graphs):
        n = int(torch.randint(lo, hi + 1, (1,), generator=g))
        a = torch.triu(torch.rand(n, n, generator=g) < 4.0 / n, 1)
        a = a | a.t()
        eis.append(a.nonzero().t() + off)
        xs.append(torch.randn(n, f, generator=g))
        bs.append(torch.full((n,), gi))
        off += n
    x, ei, batch = torch.cat(xs), torch.cat(eis, 1), torch.cat(bs)
    ew = torch.rand(ei.size(1), generator=g) + 0.1
    return x.to(dev), ei.to(dev), ew.to(dev), batch.to(dev)


def run():
    from tgp.select import NDPSelect
    dev = torch.device("cuda:0")
    x, ei, ew, batch = _er_batch(200, 5, 60, 8, 7, dev)
    outs = []
    for seed in (3, 4):
        torch.manual_seed(seed)
        so = NDPSelect()(edge_index=ei, edge_weight=ew, batch=batch, num_nodes=x.size(0))
        info = so._partition_info.cpu()
        outs.append((so.num_supernodes, int((info == -1).sum()), int(info.max()), (info == -1).nonzero().view(-1).tolist()))
    return outs


if __name__ == "__main__":
    print(os.environ.get("TGP_NDP_LANCZOS", "1"), run())
    if len(sys.argv) == 1:
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, TGP_NDP_LANCZOS="0"))
