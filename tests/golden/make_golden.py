#!/usr/bin/env python3
"""Generate the golden input/output vectors for the SRC hot path.

TEST INFRASTRUCTURE ONLY — run in the BUILD container, never on the GPU box.

It imports the *real* reference (tgp 1.0.1) from /root/reference over the
build-authored PyG stand-in (tests/golden/pyg_shim.py), runs the reference's own
operator / pooler code on small seeded inputs, and stores plain tensors (inputs,
module parameters, expected outputs) in ``tests/golden/golden_v1.pt``.  Only that
data file travels; neither the reference nor the shim is needed to consume it.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Every case is a dict {"inputs": {...}, "params": {...}, "cfg": {...}, "expected": {...}}
of tensors / python scalars.  Scores fed to TopK are tie-free (random normal fp32)
because PyG's first sort is not stable (hard parts).
"""
import os
import sys
import types
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import pyg_shim  # noqa: E402

pyg_shim.install()


def _greedy_matching(row, col, weight=None, num_nodes=None):
    """Deterministic stand-in for torch_cluster.graclus_cluster (input generator only):
    visit nodes in index order, match each unmatched node with its unmatched
    neighbour of largest weight (first one on ties); cluster id = min of the pair."""
    n = int(num_nodes)
    row, col = row.tolist(), col.tolist()
    w = [1.0] * len(row) if weight is None else weight.tolist()
    nbrs = [[] for _ in range(n)]
    for r, c, ww in zip(row, col, w):
        if r != c:
            nbrs[r].append((c, ww))
    cluster = [-1] * n
    for u in range(n):
        if cluster[u] >= 0:
            continue
        best, best_w = -1, -float("inf")
        for v, ww in nbrs[u]:
            if cluster[v] < 0 and ww > best_w:
                best, best_w = v, ww
        cluster[u] = u
        if best >= 0:
            cluster[best] = u
    return torch.tensor(cluster, dtype=torch.long)


_tc = types.ModuleType("torch_cluster")
_tc.graclus_cluster = _greedy_matching
sys.modules["torch_cluster"] = _tc

sys.path.insert(0, "/root/reference")
import tgp  # noqa: E402
from tgp.connect import DenseConnect, KronConnect, SparseConnect, sparse_connect  # noqa: E402
from tgp.poolers import get_pooler  # noqa: E402
from tgp.reduce import BaseReduce  # noqa: E402
from tgp.select import SelectOutput  # noqa: E402
from tgp.utils.ops import (  # noqa: E402
    dense_to_block_diag,
    get_mask_from_dense_s,
    postprocess_adj_pool_dense,
    postprocess_adj_pool_sparse,
)

assert tgp.__version__ == "1.0.1", tgp.__version__
CASES = {}


# ----------------------------------------------------------------------------- helpers
def er_graph(n, p, gen, weighted=False, offset=0):
    """Undirected Erdos-Renyi graph without self loops, both directions listed,
    row-major sorted (the layout PyG's generators return)."""
    upper = torch.triu(torch.rand(n, n, generator=gen) < p, diagonal=1)
    a = upper | upper.t()
    ei = a.nonzero().t().contiguous()
    ew = None
    if weighted:
        wmat = torch.rand(n, n, generator=gen) + 0.1
        wmat = torch.triu(wmat, 1)
        wmat = wmat + wmat.t()
        ew = wmat[ei[0], ei[1]].contiguous()
    return ei + offset, ew


def batched_graphs(sizes, p, gen, feat, weighted):
    eis, ews, xs, bs, off = [], [], [], [], 0
    for g, n in enumerate(sizes):
        ei, ew = er_graph(n, p, gen, weighted, off)
        eis.append(ei)
        if weighted:
            ews.append(ew)
        xs.append(torch.randn(n, feat, generator=gen))
        bs.append(torch.full((n,), g, dtype=torch.long))
        off += n
    return (torch.cat(xs), torch.cat(eis, 1), torch.cat(ews) if weighted else None,
            torch.cat(bs))


def t(v):
    if isinstance(v, torch.Tensor):
        if v.is_sparse:
            v = v.coalesce()
            return {"__coo__": True, "indices": v.indices().clone(),
                    "values": v.values().clone(), "size": list(v.size())}
        return v.detach().clone().contiguous()
    return v


def so_dict(so):
    d = {"num_nodes": so.num_nodes, "num_supernodes": so.num_supernodes}
    if so.is_sparse:
        d.update(node_index=t(so.node_index), cluster_index=t(so.cluster_index),
                 weight=t(so.weight))
    else:
        d.update(s=t(so.s))
        if so.in_mask is not None:
            d["in_mask"] = t(so.in_mask)
        d["out_mask"] = t(so.out_mask)
    if so.batch is not None:
        d["batch"] = t(so.batch)
    return d


def pool_dict(out):
    d = {"x": t(out.x), "edge_index": t(out.edge_index), "edge_weight": t(out.edge_weight),
         "batch": t(out.batch), "so": so_dict(out.so)}
    if out.loss is not None:
        d["loss"] = {k: t(v) for k, v in out.loss.items()}
    if out.mask is not None:
        d["mask"] = t(out.mask)
    return d


def params_of(module):
    return {k: t(v) for k, v in module.state_dict().items()}


def add(name, inputs, expected, cfg=None, params=None):
    assert name not in CASES, name
    CASES[name] = {"inputs": {k: t(v) for k, v in inputs.items()},
                   "params": params or {}, "cfg": cfg or {}, "expected": expected}


# ----------------------------------------------------------------------------- TopK
def gen_topk():
    # C1: BASELINE.json configs[0] — TopK ratio=0.5 on one 100-node ER graph.
    torch.manual_seed(42)
    gen = torch.Generator().manual_seed(42)
    ei, _ = er_graph(100, 0.1, gen)
    x = torch.randn(100, 16, generator=gen)
    ew = torch.ones(ei.size(1))
    pooler = get_pooler("topk", in_channels=16, ratio=0.5).eval()
    with torch.no_grad():
        out = pooler(x=x, adj=ei, edge_weight=ew)
    add("c1_topk_er100", dict(x=x, edge_index=ei, edge_weight=ew, batch=None),
        pool_dict(out), cfg=dict(in_channels=16, ratio=0.5), params=params_of(pooler))

    sizes = [12, 9, 15]
    for weighted in (True, False):
        for flags in (dict(), dict(degree_norm=True), dict(edge_weight_norm=True),
                      dict(degree_norm=True, edge_weight_norm=True, remove_self_loops=False),
                      dict(multiplier=2.5), dict(ratio=4), dict(min_score=0.05)):
            gen = torch.Generator().manual_seed(7)
            x, ei, ew, batch = batched_graphs(sizes, 0.35, gen, 6, weighted)
            # add two self loops + one duplicated edge so the filters have work to do
            extra = torch.tensor([[3, 14, 0], [3, 14, 1]])
            ei = torch.cat([ei, extra], 1)
            if ew is not None:
                ew = torch.cat([ew, torch.tensor([0.7, 1.3, 0.25])])
            cfg = dict(in_channels=6, ratio=0.5)
            cfg.update(flags)
            torch.manual_seed(11)
            pooler = get_pooler("topk", **cfg).eval()
            with torch.no_grad():
                out = pooler(x=x, adj=ei, edge_weight=ew, batch=batch)
            tag = "_".join(f"{k}-{v}" for k, v in flags.items()) or "default"
            add(f"topk_batch_{'w' if weighted else 'u'}_{tag}",
                dict(x=x, edge_index=ei, edge_weight=ew, batch=batch), pool_dict(out),
                cfg=cfg, params=params_of(pooler))

    # torch COO adjacency in -> torch COO adjacency out (base_conn.py:71-76,103-110)
    gen = torch.Generator().manual_seed(9)
    x, ei, ew, batch = batched_graphs([10, 8], 0.4, gen, 4, True)
    coo = torch.sparse_coo_tensor(ei, ew, (18, 18)).coalesce()
    torch.manual_seed(12)
    pooler = get_pooler("topk", in_channels=4, ratio=0.5).eval()
    with torch.no_grad():
        out = pooler(x=x, adj=coo, batch=batch)
    add("topk_coo_adj", dict(x=x, adj_coo=coo, batch=batch), pool_dict(out),
        cfg=dict(in_channels=4, ratio=0.5), params=params_of(pooler))


# ----------------------------------------------------------------------------- Graclus-style
def gen_cluster_connect():
    sizes = [14, 10, 11]
    for weighted in (True, False):
        gen = torch.Generator().manual_seed(21)
        x, ei, ew, batch = batched_graphs(sizes, 0.3, gen, 5, weighted)
        n = x.size(0)
        for reduce_op in ("sum", "mean", "max", "min", "mul"):
            for flags in (dict(), dict(degree_norm=True),
                          dict(edge_weight_norm=True, remove_self_loops=False)):
                if reduce_op != "sum" and flags:
                    continue
                pooler = get_pooler("graclus", connect_red_op=reduce_op, **flags)
                with torch.no_grad():
                    out = pooler(x=x, adj=ei, edge_weight=ew, batch=batch)
                tag = "_".join(f"{k}-{v}" for k, v in flags.items()) or "default"
                add(f"graclus_{'w' if weighted else 'u'}_{reduce_op}_{tag}",
                    dict(x=x, edge_index=ei, edge_weight=ew, batch=batch), pool_dict(out),
                    cfg=dict(connect_red_op=reduce_op, **flags))
        # precoarsening (no features): src.py:630-692
        pooler = get_pooler("graclus")
        pre = pooler.precoarsening(edge_index=ei, edge_weight=ew, batch=batch, num_nodes=n)
        add(f"graclus_precoarsen_{'w' if weighted else 'u'}",
            dict(edge_index=ei, edge_weight=ew, batch=batch, num_nodes=n), pool_dict(pre))
        levels = pooler.multi_level_precoarsening(2, edge_index=ei, edge_weight=ew,
                                                  batch=batch, num_nodes=n)
        add(f"graclus_precoarsen2_{'w' if weighted else 'u'}",
            dict(edge_index=ei, edge_weight=ew, batch=batch, num_nodes=n),
            {"levels": [pool_dict(l) for l in levels]})

    # many-to-one clusters with non-unit assignment weights, directed multigraph input
    gen = torch.Generator().manual_seed(33)
    n, k, e = 40, 7, 300
    ei = torch.randint(0, n, (2, e), generator=gen)
    ew = torch.randn(e, generator=gen)
    cluster = torch.randint(0, k, (n,), generator=gen)
    cluster[:k] = torch.arange(k)
    wgt = torch.rand(n, generator=gen) + 0.5
    x = torch.randn(n, 9, generator=gen)
    so = SelectOutput(cluster_index=cluster, num_nodes=n, num_supernodes=k, weight=wgt)
    xp, bp = BaseReduce()(x, so, batch=torch.zeros(n, dtype=torch.long))
    adj, w = SparseConnect()(ei, so, edge_weight=ew)
    add("cluster_many_to_one", dict(x=x, edge_index=ei, edge_weight=ew, cluster_index=cluster,
                                    weight=wgt, num_supernodes=k),
        dict(x=t(xp), batch=t(bp), edge_index=t(adj), edge_weight=t(w), so=so_dict(so)))

    # reference known-answer: 4-chain, cluster=[0,0,1,1], degree_norm without weights
    # (tests/connect/test_base_conn.py:145-198)
    ei = torch.tensor([[0, 1, 1, 2, 2, 3], [1, 0, 2, 1, 3, 2]])
    adj, w = sparse_connect(edge_index=ei, edge_weight=None,
                            cluster_index=torch.tensor([0, 0, 1, 1]), num_nodes=4,
                            num_supernodes=2, degree_norm=True, remove_self_loops=True)
    add("chain4_degree_norm_noweights", dict(edge_index=ei, cluster_index=torch.tensor([0, 0, 1, 1])),
        dict(edge_index=t(adj), edge_weight=t(w)))


# ----------------------------------------------------------------------------- NDP / Kron
def gen_ndp():
    from torch_geometric.utils import get_laplacian, to_scipy_sparse_matrix
    for weighted in (True, False):
        gen = torch.Generator().manual_seed(5)
        x, ei, ew, batch = batched_graphs([11, 13], 0.35, gen, 4, weighted)
        n = x.size(0)
        keep = torch.rand(n, generator=gen) < 0.5
        keep[0] = True
        keep[12] = True
        idx_pos = keep.nonzero().view(-1)
        eil, ewl = get_laplacian(ei, ew, normalization=None, num_nodes=n)
        L = to_scipy_sparse_matrix(eil, ewl, num_nodes=n).tocsr()
        S = torch.sparse_coo_tensor(torch.stack([idx_pos, torch.arange(idx_pos.numel())]),
                                    torch.ones(idx_pos.numel()), (n, idx_pos.numel())).coalesce()
        so = SelectOutput(s=S, L=L)
        xp, bp = BaseReduce()(x, so, batch=batch)
        adj, w = KronConnect()(ei, so, edge_weight=ew)
        add(f"ndp_kron_{'w' if weighted else 'u'}",
            dict(x=x, edge_index=ei, edge_weight=ew, batch=batch, idx_pos=idx_pos),
            dict(x=t(xp), batch=t(bp), edge_index=t(adj), edge_weight=t(w), so=so_dict(so)))
        # without a stored Laplacian (kron_conn.py 'Laplacian not provided' branch)
        so2 = SelectOutput(s=S)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            adj2, w2 = KronConnect()(ei, so2, edge_weight=ew)
        add(f"ndp_kron_nolap_{'w' if weighted else 'u'}",
            dict(edge_index=ei, edge_weight=ew, idx_pos=idx_pos, num_nodes=n),
            dict(edge_index=t(adj2), edge_weight=t(w2)))


# ----------------------------------------------------------------------------- dense poolers
def gen_dense():
    sizes = [9, 6, 12]
    for alias in ("diff", "mincut"):
        for tag, flags in (
            ("default", dict()),
            ("sparse_out", dict(sparse_output=True)),
            ("noT_ewn", dict(adj_transpose=False, edge_weight_norm=True)),
            ("raw", dict(remove_self_loops=False, degree_norm=False)),
            ("mlp2", dict(in_channels=[5, 7], act="relu")),
        ):
            for weighted in (True, False):
                gen = torch.Generator().manual_seed(3)
                x, ei, ew, batch = batched_graphs(sizes, 0.4, gen, 5, weighted)
                cfg = dict(in_channels=5, k=4)
                cfg.update(flags)
                torch.manual_seed(1)
                pooler = get_pooler(alias, **cfg).eval()
                with torch.no_grad():
                    out = pooler(x=x, adj=ei, edge_weight=ew, batch=batch)
                add(f"{alias}_batched_{tag}_{'w' if weighted else 'u'}",
                    dict(x=x, edge_index=ei, edge_weight=ew, batch=batch), pool_dict(out),
                    cfg=cfg, params=params_of(pooler))
        # unbatched mode (alias_u): sparse A, dense S [N,K], python loop over graphs
        for tag, flags in (("default", dict()), ("sparse_out", dict(sparse_output=True)),
                           ("sparse_out_ewn", dict(sparse_output=True, edge_weight_norm=True))):
            for weighted in (True, False):
                gen = torch.Generator().manual_seed(4)
                x, ei, ew, batch = batched_graphs(sizes, 0.4, gen, 5, weighted)
                cfg = dict(in_channels=5, k=4)
                cfg.update(flags)
                torch.manual_seed(2)
                pooler = get_pooler(alias + "_u", **cfg).eval()
                with torch.no_grad():
                    out = pooler(x=x, adj=ei, edge_weight=ew, batch=batch)
                add(f"{alias}_unbatched_{tag}_{'w' if weighted else 'u'}",
                    dict(x=x, edge_index=ei, edge_weight=ew, batch=batch), pool_dict(out),
                    cfg=cfg, params=params_of(pooler))
        # single graph, no batch vector, both modes
        gen = torch.Generator().manual_seed(6)
        ei, ew = er_graph(10, 0.4, gen, True)
        x = torch.randn(10, 5, generator=gen)
        for mode in ("", "_u"):
            torch.manual_seed(3)
            pooler = get_pooler(alias + mode, in_channels=5, k=3).eval()
            with torch.no_grad():
                out = pooler(x=x, adj=ei, edge_weight=ew)
            add(f"{alias}{mode}_single_graph", dict(x=x, edge_index=ei, edge_weight=ew, batch=None),
                pool_dict(out), cfg=dict(in_channels=5, k=3), params=params_of(pooler))
        # already-dense padded inputs + explicit mask (src.py:475-482)
        gen = torch.Generator().manual_seed(8)
        B, N, F = 3, 8, 5
        a = (torch.rand(B, N, N, generator=gen) < 0.4).float() * torch.rand(B, N, N, generator=gen)
        a = a + a.transpose(1, 2)
        mask = torch.ones(B, N, dtype=torch.bool)
        mask[1, 6:] = False
        mask[2, 5:] = False
        a = a * mask.unsqueeze(1) * mask.unsqueeze(2)
        xd = torch.randn(B, N, F, generator=gen) * mask.unsqueeze(-1)
        torch.manual_seed(4)
        pooler = get_pooler(alias, in_channels=F, k=3).eval()
        with torch.no_grad():
            out = pooler(x=xd, adj=a, mask=mask)
        add(f"{alias}_dense_inputs_mask", dict(x=xd, adj=a, mask=mask), pool_dict(out),
            cfg=dict(in_channels=F, k=3), params=params_of(pooler))


# ----------------------------------------------------------------------------- operator level
def gen_operators():
    gen = torch.Generator().manual_seed(13)
    # DenseConnect literals: tests/connect/test_dense_conn.py:210-232  ->  S^T A S = [[4,4],[4,0]]
    s = torch.tensor([[1.0, 0.0], [0.0, 1.0], [1.0, 0.0]])
    adj = torch.tensor([[0.0, 1.0, 2.0], [1.0, 0.0, 3.0], [2.0, 3.0, 0.0]])
    out = DenseConnect().dense_connect(adj=adj, s=s)
    add("dense_connect_literal", dict(s=s, adj=adj), dict(adj_pool=t(out)))

    B, N, K, F = 4, 17, 5, 6
    S = torch.softmax(torch.randn(B, N, K, generator=gen), -1)
    A = (torch.rand(B, N, N, generator=gen) < 0.3).float()
    A = A * torch.rand(B, N, N, generator=gen)
    A[3] = 0  # a graph without edges (test_dense_conn.py:506-535)
    X = torch.randn(B, N, F, generator=gen)
    xp, bp = BaseReduce()(X, SelectOutput(s=S), batch=None)
    exp = {"x_pool": t(xp), "raw": t(DenseConnect._dense_connect(S, A))}
    for rsl in (True, False):
        for dn in (True, False):
            for at in (True, False):
                for ewn in (True, False):
                    conn = DenseConnect(remove_self_loops=rsl, degree_norm=dn,
                                        adj_transpose=at, edge_weight_norm=ewn)
                    o, _ = conn(A.clone(), SelectOutput(s=S))
                    exp[f"rsl{int(rsl)}_dn{int(dn)}_at{int(at)}_ewn{int(ewn)}"] = t(o)
    add("dense_ops_grid", dict(S=S, A=A, X=X), exp)

    # dense unbatched reduce paths 3/4 (base_reduce.py:170-190)
    batch = torch.tensor([0] * 6 + [1] * 4 + [2] * 7)
    S2 = torch.softmax(torch.randn(17, K, generator=gen), -1)
    X2 = torch.randn(17, F, generator=gen)
    so = SelectOutput(s=S2, batch=batch)
    a, ab = BaseReduce()(X2, so, batch=batch)
    b, _ = BaseReduce()(X2, so, batch=batch, return_batched=True)
    c, cb = BaseReduce()(X2, SelectOutput(s=S2), batch=None)
    d, _ = BaseReduce()(X2, SelectOutput(s=S2), batch=None, return_batched=True)
    add("dense_reduce_unbatched", dict(S=S2, X=X2, batch=batch),
        dict(flat=t(a), flat_batch=t(ab), batched=t(b), single=t(c), single_batch=cb,
             single_batched=t(d), out_mask=t(so.out_mask)))

    # postprocess_adj_pool_sparse grid
    n, e = 12, 60
    ei = torch.randint(0, n, (2, e), generator=gen)
    ew = torch.randn(e, generator=gen)
    ew[::7] = 0.0
    ew[3] = 5e-9
    bpool = torch.tensor([0] * 5 + [1] * 7)
    exp = {}
    for rsl in (True, False):
        for dn in (True, False):
            for ewn in (True, False):
                for has_w in (True, False):
                    oi, ow = postprocess_adj_pool_sparse(
                        ei, ew if has_w else None, num_nodes=n, remove_self_loops=rsl,
                        degree_norm=dn, edge_weight_norm=ewn, batch_pooled=bpool)
                    key = f"rsl{int(rsl)}_dn{int(dn)}_ewn{int(ewn)}_w{int(has_w)}"
                    exp[key + "_ei"] = t(oi)
                    exp[key + "_ew"] = t(ow)
    add("postprocess_sparse_grid", dict(edge_index=ei, edge_weight=ew, batch_pooled=bpool, num_nodes=n), exp)

    # eps filter literal: tests/utils/test_ops.py:254-269
    oi, ow = postprocess_adj_pool_sparse(torch.tensor([[0, 1], [1, 0]]), torch.tensor([0.0, 1.0]), num_nodes=2)
    add("postprocess_sparse_eps_literal", dict(), dict(edge_index=t(oi), edge_weight=t(ow)))

    # dense_to_block_diag
    Ap = torch.randn(3, 4, 4, generator=gen)
    Ap[Ap.abs() < 0.6] = 0
    bi, bw = dense_to_block_diag(Ap)
    add("block_diag", dict(adj_pool=Ap), dict(edge_index=t(bi), edge_weight=t(bw)))

    # get_mask_from_dense_s literal: tests/utils/test_ops.py:209-218
    m = get_mask_from_dense_s(torch.tensor([[1.0, 0.0], [0.0, 1.0], [0.3, 0.7]]), batch=torch.tensor([0, 2, 2]))
    add("mask_from_dense_s_literal", dict(), dict(mask=t(m)))

    # postprocess_adj_pool_dense is applied in place on the diagonal (ops.py:308)
    Ad = torch.rand(2, 5, 5, generator=gen)
    o = postprocess_adj_pool_dense(Ad.clone(), True, True, True, True)
    add("postprocess_dense_all", dict(adj_pool=Ad), dict(out=t(o)))

    # DenseConnect on sparse A + dense S [N,K] (A7'), dense and block-diagonal outputs
    xb, eib, ewb, bb = batched_graphs([7, 5, 9], 0.45, gen, 3, True)
    S3 = torch.softmax(torch.randn(21, 4, generator=gen), -1)
    so3 = SelectOutput(s=S3, batch=bb)
    exp = {}
    for sp in (False, True):
        for ewn in (False, True):
            conn = DenseConnect(sparse_output=sp, edge_weight_norm=ewn)
            bp3 = BaseReduce.reduce_batch(so3, bb)
            o, w = conn(eib, so3, edge_weight=ewb, batch=bb, batch_pooled=bp3)
            exp[f"sp{int(sp)}_ewn{int(ewn)}_adj"] = t(o)
            exp[f"sp{int(sp)}_ewn{int(ewn)}_w"] = t(w)
    add("dense_connect_unbatched_grid", dict(edge_index=eib, edge_weight=ewb, batch=bb, S=S3), exp)


# ----------------------------------------------------------------------------- assign_all_nodes
def gen_assign_all():
    """SelectOutput.assign_all_nodes / utils.ops.get_assignments (base_select.py:381-486, ops.py:1222-1440) on
    graphs where every node is reached by propagation, so no random fallback is involved."""
    from tgp.utils.ops import get_assignments
    gen = torch.Generator().manual_seed(11)
    # two chains of 9 and 7 nodes with a few chords; kept nodes on both
    def chain(n, off):
        a = torch.arange(n - 1) + off
        ei = torch.stack([torch.cat([a, a + 1]), torch.cat([a + 1, a])])
        return ei
    ei = torch.cat([chain(9, 0), chain(7, 9), torch.tensor([[2, 6, 11, 14], [6, 2, 14, 11]])], 1)
    batch = torch.cat([torch.zeros(9, dtype=torch.long), torch.ones(7, dtype=torch.long)])
    kept = torch.tensor([1, 7, 10, 15])
    for it in (1, 2, 5):
        if it < 5:
            continue  # fewer rounds leave nodes to the random fallback: unpinned
        a = get_assignments(kept, edge_index=ei, max_iter=it, batch=batch)
        add(f"get_assignments_chains_it{it}", dict(kept=kept, edge_index=ei, batch=batch), {"assignments": t(a)},
            cfg=dict(max_iter=it))
    # vote counting + tie break: node 4 hears cluster 1 twice and cluster 2 twice -> smallest id wins
    ei2 = torch.tensor([[0, 0, 1, 1, 2, 3, 0, 1, 4], [4, 4, 4, 4, 5, 5, 2, 3, 6]])
    a2 = get_assignments(torch.tensor([0, 1]), edge_index=ei2, max_iter=3, num_nodes=7)
    add("get_assignments_votes", dict(kept=torch.tensor([0, 1]), edge_index=ei2), {"assignments": t(a2)},
        cfg=dict(max_iter=3, num_nodes=7))
    # through SelectOutput, with per-node weights and an extra attribute
    g = torch.Generator().manual_seed(12)
    w = torch.rand(16, generator=g)
    so = SelectOutput(cluster_index=torch.arange(4), node_index=kept, num_nodes=16, num_supernodes=4, tag="kept")
    full = so.assign_all_nodes(adj=ei, weight=w, batch=batch, closest_node_assignment=True)
    add("assign_all_nodes_chains", dict(kept=kept, edge_index=ei, batch=batch, weight=w),
        {"so": so_dict(full), "tag": full.tag})


def main():
    gen_assign_all_last = gen_assign_all
    gen_topk()
    gen_cluster_connect()
    gen_ndp()
    gen_dense()
    gen_operators()
    gen_assign_all_last()
    out = os.path.join(HERE, "golden_v1.pt")
    torch.save({"tgp_version": tgp.__version__, "torch": str(torch.__version__),
                "numpy": str(np.__version__), "cases": CASES}, out)
    print(f"wrote {len(CASES)} cases -> {out} ({os.path.getsize(out) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
