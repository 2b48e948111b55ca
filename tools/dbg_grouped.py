import os, sys, random
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-geometric-pool_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from test_gpu_fuzz import rand_graph, O
from tgp.connect import sparse_connect
dev = torch.device("cuda:0")
for seed in (1, 3):
    rng = random.Random(seed)
    g = torch.Generator().manual_seed(7000 + seed)
    n, k = 180_000, rng.choice([70_000, 100_000, 180_000])
    e = 300_000
    ei = rand_graph(rng, g, n, e, sort_rows=False, with_loops=True, with_dups=True)
    ew = (torch.rand(ei.size(1), generator=g) + 0.05) if rng.random() < 0.7 else None
    if ew is not None:
        ew[torch.rand(ei.size(1), generator=g) < 0.05] = 0.0
    cluster = torch.randint(0, k, (n,), generator=g)
    op = rng.choice(["sum", "mean", "min", "max", "mul"])
    flags = dict(remove_self_loops=rng.random() < 0.5, degree_norm=rng.random() < 0.3)
    print(seed, k, op, flags)
    ref_ei, ref_ew = O.sparse_connect(ei, ew, torch.arange(n), cluster, n, k, reduce_op=op, **flags)
    got_ei, got_ew = sparse_connect(ei.to(dev), ew.to(dev), node_index=torch.arange(n, device=dev),
                                    cluster_index=cluster.to(dev), num_nodes=n, num_supernodes=k, reduce_op=op, **flags)
    bad = (~torch.isclose(got_ew.cpu(), ref_ew, rtol=1e-5, atol=1e-6)).nonzero().flatten()
    cr, cc = cluster[ei[0]], cluster[ei[1]]
    for i in bad.tolist():
        r, c = ref_ei[0, i].item(), ref_ei[1, i].item()
        rowlen = int((cr == r).sum())
        sel = (cr == r) & (cc == c)
        print("  row", r, "col", c, "rowlen", rowlen, "run", int(sel.sum()), "weights", ew[sel].tolist(), "got", got_ew[i].item(), "ref", ref_ew[i].item())
        cols = cc[cr == r]
        u, cnts = torch.unique(cols, return_counts=True)
        print("     runs in row:", sorted(cnts.tolist(), reverse=True)[:6], "self in row", int((cols == r).sum()))
