"""Pin the CPU oracle (oracle/tgp_oracle.py) before anything trusts it.

(i) literal known-answers held by the reference's own tests,
(ii) every case of tests/golden/golden_v1.pt, i.e. outputs of the reference's own
     code run in the build container (tests/golden/make_golden.py).
Bars: integer / index outputs bit-exact; fp32 rtol = atol = 1e-5 (the tolerance the
reference uses itself: tests/poolers/test_dense_poolers_batched_vs_unbatched.py:124-171).
"""
import pytest
import torch

import tgp_oracle as O

RTOL = ATOL = 1e-5


def close(a, b, msg=""):
    assert a.shape == b.shape, f"{msg}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    torch.testing.assert_close(a, b, rtol=RTOL, atol=ATOL, msg=lambda m: f"{msg}: {m}")


def exact(a, b, msg=""):
    if a is None or b is None:
        assert a is None and b is None, msg
        return
    assert a.dtype == b.dtype and torch.equal(a, b), f"{msg}: {a} vs {b}"


def check_pool(got, exp, name):
    exact(got["edge_index"] if got["edge_index"].dtype == torch.long else None,
          exp["edge_index"] if exp["edge_index"].dtype == torch.long else None, name + ".edge_index")
    if exp["edge_index"].dtype != torch.long:
        close(got["edge_index"], exp["edge_index"], name + ".adj")
    if exp["edge_weight"] is None:
        assert got["edge_weight"] is None, name
    else:
        close(got["edge_weight"], exp["edge_weight"], name + ".edge_weight")
    exact(got["batch"], exp["batch"], name + ".batch")
    if exp["x"] is not None:
        close(got["x"], exp["x"], name + ".x")


# --------------------------------------------------------------------- literals
def test_literal_topk_selection_known_answer():
    # reference tests/poolers/test_topk.py:22-34
    x = torch.arange(1.0, 6.0).unsqueeze(-1)
    ni, ci, w = O.topk_select(x, None, 0.5, None, act="linear")
    assert torch.equal(ni.sort(descending=True)[0], torch.tensor([4, 3, 2]))


def test_literal_dense_connect():
    # reference tests/connect/test_dense_conn.py:210-232  ->  [[4,4],[4,0]]
    s = torch.tensor([[1.0, 0.0], [0.0, 1.0], [1.0, 0.0]])
    adj = torch.tensor([[0.0, 1.0, 2.0], [1.0, 0.0, 3.0], [2.0, 3.0, 0.0]])
    out = O.dense_connect(s.unsqueeze(0), adj.unsqueeze(0))
    assert torch.equal(out, torch.tensor([[[4.0, 4.0], [4.0, 0.0]]]))


def test_literal_eps_filter():
    # reference tests/utils/test_ops.py:254-269
    ei, ew = O.postprocess_sparse(torch.tensor([[0, 1], [1, 0]]), torch.tensor([0.0, 1.0]), 2)
    assert ei.shape == (2, 1) and torch.equal(ew, torch.tensor([1.0]))


def test_literal_mask_from_dense_s():
    # reference tests/utils/test_ops.py:209-218
    m = O.out_mask_dense(torch.tensor([[1.0, 0.0], [0.0, 1.0], [0.3, 0.7]]), torch.tensor([0, 2, 2]))
    assert m.shape == (3, 2) and torch.equal(m[1], torch.tensor([False, False]))


def test_literal_reduce_batch_dense():
    # reference tests/reduce/test_base_reduce.py:8-36, 50-72
    out = O.reduce_batch_dense(torch.tensor([0, 0, 1, 1]), 2)
    assert torch.equal(out, torch.tensor([0, 0, 1, 1]))
    x = torch.tensor([[1.0, 0.0], [0.0, 1.0], [2.0, 0.0], [0.0, 2.0]])
    s = torch.tensor([[1.0, 0.0], [0.0, 1.0], [1.0, 0.0], [0.0, 1.0]])
    xp = O.reduce_dense(s, x, torch.tensor([0, 0, 1, 1]))
    assert xp.shape == (4, 2)


def test_literal_edge_weight_norm_and_self_loops():
    # reference tests/connect/test_dense_conn.py:59-107
    g = torch.Generator().manual_seed(0)
    a = torch.rand(3, 6, 6, generator=g)
    s = torch.softmax(torch.randn(3, 6, 3, generator=g), -1)
    raw = O.postprocess_dense(O.dense_connect(s, a), True, True, True, False)
    nrm = O.postprocess_dense(O.dense_connect(s, a), True, True, True, True)
    for b in range(3):
        torch.testing.assert_close(nrm[b], raw[b] / raw[b].abs().max(), rtol=1e-6, atol=1e-6)
        assert torch.equal(raw[b].diagonal(), torch.zeros(3))


def test_literal_empty_graph_dense_connect_unbatched():
    # reference tests/connect/test_dense_conn.py:506-535
    s = torch.softmax(torch.randn(4, 2), -1)
    ei = torch.empty((2, 0), dtype=torch.long)
    assert torch.equal(O.dense_connect_unbatched(ei, None, None, s), torch.zeros(1, 2, 2))
    assert torch.equal(O.dense_connect_unbatched(ei, None, torch.tensor([0, 0, 1, 1]), s),
                       torch.zeros(2, 2, 2))


# --------------------------------------------------------------------- golden: TopK
def _topk_cases(golden):
    return [k for k in golden if k.startswith("topk_batch") or k == "c1_topk_er100"]


def test_golden_topk(golden):
    names = _topk_cases(golden)
    assert len(names) == 15
    for name in names:
        c = golden[name]
        i, cfg, e = c["inputs"], dict(c["cfg"]), c["expected"]
        cfg.pop("in_channels")
        got = O.topk_pool(i["x"], i["edge_index"], i["edge_weight"], i["batch"],
                          c["params"].get("selector.weight"), **cfg)
        exact(got["node_index"], e["so"]["node_index"], name)
        exact(got["cluster_index"], e["so"]["cluster_index"], name)
        close(got["weight"], e["so"]["weight"], name)
        check_pool(got, e, name)


def test_golden_topk_coo(golden):
    c = golden["topk_coo_adj"]
    i, e = c["inputs"], c["expected"]
    coo = i["adj_coo"]
    got = O.topk_pool(i["x"], coo["indices"], coo["values"], i["batch"],
                      c["params"]["selector.weight"], ratio=0.5)
    exact(got["edge_index"], e["edge_index"]["indices"], "coo")
    close(got["edge_weight"], e["edge_index"]["values"], "coo")
    assert e["edge_weight"] is None


# --------------------------------------------------------------------- golden: cluster connect
def test_golden_graclus(golden):
    names = [k for k in golden if k.startswith("graclus_") and "precoarsen" not in k]
    assert len(names) == 14
    for name in names:
        c = golden[name]
        i, e = c["inputs"], c["expected"]
        cfg = dict(c["cfg"])
        n = i["x"].size(0)
        cl = O.greedy_matching(i["edge_index"], i["edge_weight"], n)
        exact(cl, e["so"]["cluster_index"], name + ".cluster")
        got = O.cluster_pool(i["x"], i["edge_index"], i["edge_weight"], i["batch"], cl,
                             e["so"]["num_supernodes"], reduce_op=cfg.pop("connect_red_op"), **cfg)
        check_pool(got, e, name)


def test_golden_graclus_precoarsening(golden):
    for wtag in ("w", "u"):
        c = golden[f"graclus_precoarsen_{wtag}"]
        i, e = c["inputs"], c["expected"]
        cl = O.greedy_matching(i["edge_index"], i["edge_weight"], i["num_nodes"])
        got = O.cluster_pool(None, i["edge_index"], i["edge_weight"], i["batch"], cl,
                             e["so"]["num_supernodes"])
        check_pool(got, e, "pre")
        lv = golden[f"graclus_precoarsen2_{wtag}"]["expected"]["levels"]
        check_pool(got, lv[0], "pre2.l0")
        n1 = lv[0]["so"]["num_supernodes"]
        cl1 = O.greedy_matching(got["edge_index"], got["edge_weight"], n1)
        got1 = O.cluster_pool(None, got["edge_index"], got["edge_weight"], got["batch"], cl1,
                              lv[1]["so"]["num_supernodes"])
        check_pool(got1, lv[1], "pre2.l1")


def test_golden_many_to_one_and_chain(golden):
    c = golden["cluster_many_to_one"]
    i, e = c["inputs"], c["expected"]
    got = O.cluster_pool(i["x"], i["edge_index"], i["edge_weight"],
                         torch.zeros(40, dtype=torch.long), i["cluster_index"],
                         i["num_supernodes"], weight=i["weight"])
    check_pool(got, e, "many_to_one")
    c = golden["chain4_degree_norm_noweights"]
    ei, ew = O.sparse_connect(c["inputs"]["edge_index"], None, torch.arange(4),
                              c["inputs"]["cluster_index"], 4, 2, degree_norm=True)
    exact(ei, c["expected"]["edge_index"])
    close(ew, c["expected"]["edge_weight"])
    assert bool(((ew >= 0) & (ew <= 1)).all())  # reference tests/connect/test_base_conn.py:145-198


# --------------------------------------------------------------------- golden: NDP / Kron
def test_golden_ndp_kron(golden):
    for wtag in ("w", "u"):
        c = golden[f"ndp_kron_{wtag}"]
        i, e = c["inputs"], c["expected"]
        n = i["x"].size(0)
        ip = i["idx_pos"]
        ni, ci, w = O.sort_assignment(ip, torch.arange(ip.numel()), None)
        close(O.reduce_sparse(i["x"], ni, ci, w, ip.numel()), e["x"], "ndp.x")
        exact(O.reduce_batch_sparse(i["batch"], ni, ci, ip.numel()), e["batch"], "ndp.batch")
        L = O.laplacian_scipy(i["edge_index"], i["edge_weight"], n)
        ei, ew = O.kron_connect(L, ip)
        exact(ei, e["edge_index"], "kron.ei")
        close(ew, e["edge_weight"], "kron.ew")
        c2 = golden[f"ndp_kron_nolap_{wtag}"]
        exact(ei, c2["expected"]["edge_index"], "kron.nolap.ei")
        close(ew, c2["expected"]["edge_weight"], "kron.nolap.ew")


# --------------------------------------------------------------------- golden: dense poolers
def _mlp_params(params):
    ws = [v for k, v in sorted(params.items()) if k.endswith("weight")]
    bs = [v for k, v in sorted(params.items()) if k.endswith("bias")]
    return ws, bs


def _run_dense(name, c):
    alias = name.split("_")[0]
    i, cfg, e = c["inputs"], dict(c["cfg"]), c["expected"]
    cfg.pop("in_channels")
    cfg.pop("k")
    ws, bs = _mlp_params(c["params"])
    batched = "_unbatched_" not in name and "_u_" not in name
    if "adj" in i:
        got = O.dense_pool(alias, i["x"], i["adj"], None, None, ws, bs, mask=i["mask"], **cfg)
    else:
        got = O.dense_pool(alias, i["x"], i["edge_index"], i["edge_weight"], i["batch"], ws, bs,
                           batched=batched, **cfg)
    close(got["s"], e["so"]["s"], name + ".s")
    exact(got["mask"], e["mask"], name + ".mask")
    for k, v in e["loss"].items():
        close(got["loss"][k], v, name + "." + k)
    check_pool(got, e, name)


def test_golden_dense_poolers(golden):
    names = [k for k in golden if k.split("_")[0] in ("diff", "mincut")]
    assert len(names) == 38
    for name in names:
        _run_dense(name, golden[name])


# --------------------------------------------------------------------- golden: operators
def test_golden_dense_ops_grid(golden):
    c = golden["dense_ops_grid"]
    S, A, X = c["inputs"]["S"], c["inputs"]["A"], c["inputs"]["X"]
    e = c["expected"]
    close(O.reduce_dense(S, X), e["x_pool"])
    raw = O.dense_connect(S, A)
    close(raw, e["raw"])
    assert torch.equal(raw[3], torch.zeros_like(raw[3]))
    for key, val in e.items():
        if not key.startswith("rsl"):
            continue
        f = {p[:-1]: bool(int(p[-1])) for p in key.split("_")}
        close(O.postprocess_dense(raw, f["rsl"], f["dn"], f["at"], f["ewn"]), val, key)


def test_golden_dense_reduce_unbatched(golden):
    c = golden["dense_reduce_unbatched"]
    S, X, b = c["inputs"]["S"], c["inputs"]["X"], c["inputs"]["batch"]
    e = c["expected"]
    close(O.reduce_dense(S, X, b), e["flat"])
    close(O.reduce_dense(S, X, b, True), e["batched"])
    close(O.reduce_dense(S, X, None), e["single"])
    close(O.reduce_dense(S, X, None, True), e["single_batched"])
    exact(O.reduce_batch_dense(b, S.size(1)), e["flat_batch"])
    exact(O.out_mask_dense(S, b), e["out_mask"])


def test_golden_postprocess_sparse_grid(golden):
    c = golden["postprocess_sparse_grid"]
    i, e = c["inputs"], c["expected"]
    for key in [k[:-3] for k in e if k.endswith("_ei")]:
        f = {p[:-1]: bool(int(p[-1])) for p in key.split("_")}
        ei, ew = O.postprocess_sparse(i["edge_index"], i["edge_weight"] if f["w"] else None,
                                      i["num_nodes"], f["rsl"], f["dn"], f["ewn"], i["batch_pooled"])
        exact(ei, e[key + "_ei"], key)
        if e[key + "_ew"] is None:
            assert ew is None, key
        else:
            close(ew, e[key + "_ew"], key)


def test_golden_block_diag_and_dense_post(golden):
    c = golden["block_diag"]
    ei, ew = O.dense_to_block_diag(c["inputs"]["adj_pool"])
    exact(ei, c["expected"]["edge_index"])
    close(ew, c["expected"]["edge_weight"])
    c = golden["postprocess_dense_all"]
    close(O.postprocess_dense(c["inputs"]["adj_pool"], True, True, True, True), c["expected"]["out"])


def test_golden_dense_connect_unbatched_grid(golden):
    c = golden["dense_connect_unbatched_grid"]
    i, e = c["inputs"], c["expected"]
    raw = O.dense_connect_unbatched(i["edge_index"], i["edge_weight"], i["batch"], i["S"])
    bp = O.reduce_batch_dense(i["batch"], i["S"].size(1))
    for ewn in (False, True):
        close(O.postprocess_dense(raw, True, True, False, ewn), e[f"sp0_ewn{int(ewn)}_adj"])
        ei, ew = O.dense_to_block_diag(raw)
        ei, ew = O.postprocess_sparse(ei, ew, raw.size(0) * raw.size(1), True, True, ewn, bp)
        exact(ei, e[f"sp1_ewn{int(ewn)}_adj"])
        close(ew, e[f"sp1_ewn{int(ewn)}_w"])


def test_get_assignments_and_assign_all_nodes(golden):
    """A0 `SelectOutput.assign_all_nodes` (base_select.py:381-486): oracle loops == reference dump == product
    (host-side torch logic, runs on any device)."""
    import tgp_oracle as O
    from tgp.select import SelectOutput
    from tgp.utils.ops import get_assignments
    for name in ("get_assignments_chains_it5", "get_assignments_votes"):
        c = golden[name]
        n = c["cfg"].get("num_nodes") or c["inputs"]["batch"].numel()
        want = c["expected"]["assignments"]
        assert torch.equal(O.get_assignments(c["inputs"]["kept"], c["inputs"]["edge_index"], c["cfg"]["max_iter"], n), want)
        got = get_assignments(c["inputs"]["kept"], edge_index=c["inputs"]["edge_index"], max_iter=c["cfg"]["max_iter"],
                              batch=c["inputs"].get("batch"), num_nodes=c["cfg"].get("num_nodes"))
        assert torch.equal(got, want)
    c = golden["assign_all_nodes_chains"]
    i = c["inputs"]
    so = SelectOutput(cluster_index=torch.arange(4), node_index=i["kept"], num_nodes=16, num_supernodes=4, tag="kept")
    full = so.assign_all_nodes(adj=i["edge_index"], weight=i["weight"], batch=i["batch"])
    e = c["expected"]["so"]
    assert torch.equal(full.cluster_index, e["cluster_index"]) and torch.equal(full.node_index, e["node_index"])
    assert torch.equal(full.weight, e["weight"]) and full.tag == c["expected"]["tag"]
    assert (full.num_nodes, full.num_supernodes) == (e["num_nodes"], e["num_supernodes"])
    # reference tests/selection/test_base_select.py:380-405, 442-466, 469-494
    small = SelectOutput(cluster_index=torch.tensor([0, 1]), node_index=torch.tensor([0, 2]), num_nodes=4, num_supernodes=2)
    ei = torch.tensor([[0, 1, 2, 3], [1, 0, 3, 2]])
    with pytest.raises(ValueError, match=r"Weight tensor size \(3\) must match the number of nodes \(4\)"):
        small.assign_all_nodes(adj=ei, weight=torch.ones(3))
    out = small.assign_all_nodes(adj=ei)
    assert (out.num_nodes, out.num_supernodes) == (4, 2) and out.cluster_index.tolist() == [0, 0, 1, 1]
    coo = torch.sparse_coo_tensor(ei, torch.ones(4), size=(4, 4)).coalesce()
    assert small.assign_all_nodes(adj=coo).cluster_index.tolist() == [0, 0, 1, 1]
    with pytest.raises(AssertionError):
        small.assign_all_nodes(adj=None)
    assert small.assign_all_nodes(adj=ei) is not small and out.assign_all_nodes(adj=ei) is out  # all kept: unchanged
