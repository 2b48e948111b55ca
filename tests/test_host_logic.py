"""Host-side mirror of the reference's operator / factory interface (no GPU needed): names, argument
handling, error types and repr strings the reference's tests pin."""
import pytest
import torch

from tgp.connect import DenseConnect, KronConnect, SparseConnect
from tgp.lift import BaseLift
from tgp.poolers import DiffPool, MinCutPooling, TopkPooling, get_pooler, pooler_map
from tgp.reduce import BaseReduce, Reduce
from tgp.select import MLPSelect, SelectOutput, TopkSelect, cluster_to_s, topk
from tgp.src import PoolingOutput
from tgp.utils import (
    check_and_filter_edge_weights,
    connectivity_to_edge_index,
    connectivity_to_torch_coo,
    get_mask_from_dense_s,
    is_dense_adj,
)


# ------------------------------------------------------------------ factory (reference poolers/__init__.py:91-147)
def test_get_pooler_aliases_and_kwargs_filtering():
    assert set(pooler_map) == {"topk", "graclus", "ndp", "diff", "mincut"}
    p = get_pooler("TopK", in_channels=8, ratio=0.25, k=99, not_an_arg=1)
    assert isinstance(p, TopkPooling) and p.selector.ratio == 0.25
    assert get_pooler("diff_u", in_channels=4, k=3).batched is False
    assert get_pooler("diff", in_channels=4, k=3).batched is True
    assert get_pooler("mincut_u", in_channels=4, k=3, batched=True).batched is True  # setdefault only
    with pytest.raises(ValueError, match="Unknown pooler_name"):
        get_pooler("does_not_exist")
    with pytest.raises(ValueError, match="Unknown pooler_name"):
        get_pooler("nope_u")
    with pytest.raises(TypeError, match=r"Missing required argument\(s\) for pooler 'mincut' \(MinCutPooling\): in_channels, k"):
        get_pooler("mincut")
    with pytest.raises(ValueError):
        TopkPooling(in_channels=4, ratio=None, min_score=None)  # reference tests/poolers/test_topk.py:11-19


def test_state_dict_names_match_reference():
    assert list(get_pooler("topk", in_channels=5).state_dict()) == ["selector.weight"]
    assert list(get_pooler("diff", in_channels=5, k=3).state_dict()) == ["selector.mlp.lins.0.weight",
                                                                         "selector.mlp.lins.0.bias"]
    assert len(get_pooler("mincut", in_channels=[5, 7], k=3, act="relu").state_dict()) == 4


def test_pooler_properties():
    p = get_pooler("graclus")
    assert p.is_sparse and not p.is_dense and p.is_precoarsenable and not p.has_loss and not p.is_trainable
    d = get_pooler("diff", in_channels=4, k=2)
    assert d.is_dense and d.has_loss and d.is_trainable and not d.is_precoarsenable
    assert "in_channels" in DiffPool.get_signature().args and "lifting" in MinCutPooling.get_forward_signature().args


# ------------------------------------------------------------------ repr strings (reference tests pin these)
def test_repr_strings():
    assert repr(BaseReduce()) == "BaseReduce()"  # tests/reduce/test_base_reduce.py:136-138
    assert repr(SparseConnect()) == ("SparseConnect(reduce_op=sum, remove_self_loops=True, "
                                     "edge_weight_norm=False, degree_norm=False)")
    assert repr(DenseConnect()) == ("DenseConnect(remove_self_loops=True, degree_norm=True, adj_transpose=True, "
                                    "edge_weight_norm=False, sparse_output=False)")
    assert repr(KronConnect()) == "KronConnect(sparse_threshold=0.01)"
    assert repr(BaseLift()) == "BaseLift(matrix_op=precomputed, reduce_op=sum)"
    sel = TopkSelect(in_channels=1, ratio=0.5, act="linear")
    assert "ratio=0.5" in repr(sel) and "min_score" not in repr(sel)
    assert "min_score=0.1" in repr(TopkSelect(in_channels=3, min_score=0.1))
    assert repr(MLPSelect(4, 3)) == "MLPSelect(in_channels=[4], k=3, act=None, dropout=0.0, s_inv_op=transpose)"
    assert repr(get_pooler("topk", in_channels=3)).startswith("TopkPooling(\n\tselect=TopkSelect(")
    with pytest.raises(TypeError, match="sparse_output must be a bool"):
        DenseConnect(sparse_output="yes")


# ------------------------------------------------------------------ SelectOutput (reference select/base_select.py)
def test_select_output_from_cluster_index():
    so = SelectOutput(cluster_index=torch.tensor([1, 0, 1, 2]))
    assert so.is_sparse and so.num_nodes == 4 and so.num_supernodes == 3
    assert torch.equal(so.node_index, torch.arange(4))  # tests/selection/test_base_select.py:62-64
    assert torch.equal(so.weight, torch.ones(4))
    assert torch.equal(so.s_inv.to_dense(), so.s.to_dense().t())
    assert so.out_mask is None and repr(so) == "SelectOutput(num_nodes=4, num_supernodes=3)"
    ei, w = cluster_to_s(torch.tensor([1, 0]), as_edge_index=True)
    assert torch.equal(ei, torch.tensor([[0, 1], [1, 0]])) and w is None


def test_select_output_is_stored_node_sorted():
    so = SelectOutput(node_index=torch.tensor([7, 2, 5]), num_nodes=9, cluster_index=torch.arange(3),
                      num_supernodes=3, weight=torch.tensor([0.7, 0.2, 0.5]))
    assert torch.equal(so.node_index, torch.tensor([2, 5, 7]))
    assert torch.equal(so.cluster_index, torch.tensor([1, 2, 0]))
    assert torch.equal(so.weight, torch.tensor([0.2, 0.5, 0.7]))


def test_select_output_dense_and_masks():
    s = torch.rand(2, 5, 3)
    mask = torch.ones(2, 5, dtype=torch.bool)
    so = SelectOutput(s=s, in_mask=mask, foo=torch.ones(2))
    assert so.is_dense and so.out_mask.shape == (2, 3) and "in_mask" in so._extra_args and "foo" in so._extra_args
    with pytest.raises(ValueError, match="must be 2D"):
        SelectOutput(s=s, in_mask=torch.ones(5, dtype=torch.bool))
    with pytest.raises(ValueError, match="only supported for batched dense"):
        SelectOutput(s=torch.rand(5, 3), in_mask=mask)
    with pytest.raises(AssertionError):
        SelectOutput(s=s, cluster_index=torch.arange(5))
    with pytest.raises(ValueError):
        SelectOutput(s="nope")
    m = get_mask_from_dense_s(torch.tensor([[1.0, 0.0], [0.0, 1.0], [0.3, 0.7]]), batch=torch.tensor([0, 2, 2]))
    assert m.shape == (3, 2) and torch.equal(m[1], torch.tensor([False, False]))  # tests/utils/test_ops.py:209-218
    with pytest.raises(ValueError, match="s must have shape"):
        get_mask_from_dense_s(torch.ones(1, 1, 1, 1))
    c = so.clone()
    assert c is not so and torch.equal(c.s, so.s)


def test_topk_select_known_answer():
    # reference tests/poolers/test_topk.py:22-34
    sel = TopkSelect(in_channels=1, ratio=0.5, act="linear")
    out = sel.forward(x=torch.arange(1.0, 6.0).unsqueeze(-1), batch=None)
    assert torch.equal(out.node_index.sort(descending=True)[0], torch.tensor([4, 3, 2]))
    score = torch.tensor([0.1, 0.9, 0.5, 0.4, 0.8, 0.3])
    batch = torch.tensor([0, 0, 0, 1, 1, 1])
    assert torch.equal(topk(score, 0.5, batch), torch.tensor([1, 2, 4, 3]))
    assert torch.equal(topk(score, 1, batch), torch.tensor([1, 4]))


def test_reduce_batch_dense_cases():
    # reference tests/reduce/test_base_reduce.py:8-47
    so = SelectOutput(s=torch.randn(4, 2))
    assert torch.equal(BaseReduce.reduce_batch(so, torch.tensor([0, 0, 1, 1])), torch.tensor([0, 0, 1, 1]))
    assert torch.equal(Reduce.reduce_batch(SelectOutput(s=torch.randn(3, 2)), torch.zeros(3, dtype=torch.long)),
                       torch.tensor([0, 0]))
    assert Reduce.reduce_batch(so, None) is None
    out = BaseReduce.reduce_batch(SelectOutput(s=torch.empty((0, 2))), torch.empty((0,), dtype=torch.long))
    assert out.shape == (0,)


# ------------------------------------------------------------------ PoolingOutput (reference src.py:19-116)
def test_pooling_output_record():
    so = SelectOutput(s=torch.rand(1, 4, 2))
    out = PoolingOutput(x=torch.zeros(1, 2, 3), edge_index=torch.zeros(1, 2, 2), so=so, loss={"a": torch.tensor(1.0)})
    x, ei, ew, batch, mask, so2, loss = out
    assert ew is None and batch is None and so2 is so and mask.shape == (1, 2) and out.has_loss
    assert out.get_loss_value("a") == 1.0 and out.get_loss_value() == [torch.tensor(1.0)]
    assert PoolingOutput().get_loss_value() == 0 and PoolingOutput().mask is None
    assert repr(out) == ("PoolingOutput(so=[4, 2], x=[1, 2, 3], edge_index=[1, 2, 2], edge_weight=None, "
                         "batch=None, mask=[1, 2], loss=['a'])")
    assert out.as_data().num_nodes == 2


# ------------------------------------------------------------------ connectivity helpers (reference utils/ops.py)
def test_connectivity_converters_and_errors():
    ei = torch.tensor([[0, 1], [1, 0]])
    assert connectivity_to_edge_index(ei, torch.ones(2, 1))[1].shape == (2,)
    with pytest.raises(ValueError, match="Dense adjacency matrices are not supported by connectivity_to_edge_index"):
        connectivity_to_edge_index(torch.rand(3, 3))
    with pytest.raises(ValueError, match="dtype torch.long"):
        connectivity_to_edge_index(ei.to(torch.int32))
    with pytest.raises(RuntimeError, match=r"Edge weights must be of shape \[E\] or \[E, 1\]"):
        check_and_filter_edge_weights(torch.ones(2, 2))
    with pytest.raises(ValueError, match="Edge index must be of type Tensor or SparseTensor"):
        connectivity_to_torch_coo([[0, 1], [1, 0]])
    coo = connectivity_to_torch_coo(ei, None, 2)
    assert coo.is_sparse and torch.equal(coo.to_dense(), torch.tensor([[0.0, 1.0], [1.0, 0.0]]))
    i2, w2 = connectivity_to_edge_index(coo)
    assert torch.equal(i2, ei) and torch.equal(w2, torch.ones(2))
    assert is_dense_adj(torch.rand(2, 3, 3)) and is_dense_adj(torch.rand(3, 3))
    assert not is_dense_adj(ei) and not is_dense_adj(coo)


def test_connect_argument_errors():
    so_dense = SelectOutput(s=torch.rand(1, 4, 2))
    with pytest.raises(ValueError, match="batch sizes do not match"):
        DenseConnect._prepare_batched_dense_inputs(torch.randn(2, 4, 2), torch.randn(3, 4, 4))
    with pytest.raises(ValueError, match="SelectOutput is required"):
        DenseConnect()(torch.rand(1, 4, 4), None)
    with pytest.raises(ValueError, match="DenseConnect expects a dense assignment matrix"):
        DenseConnect()(torch.rand(1, 4, 4), SelectOutput(cluster_index=torch.tensor([0, 0, 1, 1])))
    with pytest.raises(AssertionError, match="batch_pooled parameter is required"):
        SparseConnect(edge_weight_norm=True)(torch.tensor([[0, 1], [1, 0]]),
                                             SelectOutput(cluster_index=torch.tensor([0, 1])))
    with pytest.raises(ValueError, match="Dense SelectOutput.s must be 2D"):
        so4 = SelectOutput(s=torch.rand(4, 2))
        so4.s = torch.rand(1, 1, 4, 2)
        BaseReduce()(torch.rand(4, 3), so4)
    assert so_dense.num_supernodes == 2


def test_batch_info_memo_is_keyed_by_object_and_version():
    """The per-batch-vector memo (graph count, sizes, offsets, longest graph) must never serve stale facts:
    a different tensor object or an in-place write misses; trailing empty graphs are honoured."""
    import gc

    import torch
    from tgp.utils.ops import batch_info, graph_ptr, is_multi_graph_batch, max_graph_size, num_graphs_of

    b = torch.tensor([0, 0, 1, 1, 1, 3])
    info = batch_info(b)
    assert (info.num_graphs, info.max_nodes, info.distinct) == (4, 3, 3)
    assert batch_info(b) is info  # same object, same version: memo hit
    assert info.sizes.tolist() == [2, 3, 0, 1] and info.ptr.tolist() == [0, 2, 5, 5, 6]
    b[5] = 1  # in-place write bumps the version counter
    info2 = batch_info(b)
    assert info2 is not info and (info2.num_graphs, info2.max_nodes, info2.distinct) == (2, 4, 2)
    sizes, ptr = graph_ptr(b, 5)  # caller-supplied batch_size beyond max + 1
    assert sizes.tolist() == [2, 4, 0, 0, 0] and ptr.tolist() == [0, 2, 6, 6, 6, 6]
    # a new tensor that happens to reuse the Python id of a dead one must miss
    for _ in range(50):
        t = torch.randint(0, 7, (11,))
        got = batch_info(t)
        assert got.num_graphs == int(t.max()) + 1 and got.max_nodes == int(torch.bincount(t).max())
        del t, got
        gc.collect()
    assert not is_multi_graph_batch(torch.tensor([2, 2, 2])) and num_graphs_of(torch.tensor([2, 2, 2])) == 3
    assert is_multi_graph_batch(torch.tensor([0, 2])) and max_graph_size(torch.tensor([0, 2, 2])) == 2
    assert num_graphs_of(None) == 1 and num_graphs_of(torch.zeros(0, dtype=torch.long)) == 1


def test_select_output_s_inv_is_lazy_but_behaves_like_the_eager_attribute():
    """S_inv (reference base_select.py:290-300) is materialised on first access; assignment, .apply() and clone()
    keep the reference's observable behaviour."""
    import torch
    from tgp.select import SelectOutput

    so = SelectOutput(cluster_index=torch.tensor([0, 1, 1, 2]), num_nodes=4, num_supernodes=3)
    assert so._s_inv is None and so.s_inv_is_transpose_of_s          # nothing built yet
    t = so.s_inv
    assert torch.equal(t.to_dense(), so.s.to_dense().t()) and so.s_inv is t and so.s_inv_is_transpose_of_s
    so2 = so.clone()
    assert so2.s_inv is not t and torch.equal(so2.s_inv.to_dense(), t.to_dense()) and so2.s_inv_is_transpose_of_s
    so.apply(lambda v: v * 2)                                         # S changed: S_inv follows it
    assert torch.equal(so.s_inv.to_dense(), so.s.to_dense().t())
    custom = torch.ones(3, 4).to_sparse()
    so.s_inv = custom                                                 # a user-provided inverse is kept as is
    assert so.s_inv is custom and not so.s_inv_is_transpose_of_s
    so.apply(lambda v: v * 3)
    assert torch.equal(so.s_inv.to_dense(), torch.full((3, 4), 3.0))
    dense = SelectOutput(s=torch.rand(2, 5, 3), s_inv_op="inverse")
    assert not dense.s_inv_is_transpose_of_s and dense.s_inv.shape == (2, 3, 5)
    dense.set_s_inv("transpose")
    assert dense.s_inv_is_transpose_of_s and torch.equal(dense.s_inv, dense.s.transpose(-1, -2))


def test_roofline_traffic_json_is_the_pmc_summary_it_names():
    """profiles/roofline_traffic.json (what bench.py reports as roofline.traffic) is generated by tools/pmc_summary.py
    from the same PMC pass as the markdown summary it names in `_source`: every value must be that file's per-call
    total, and every key must be one bench.py asks for."""
    import json
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    blob = json.load(open(os.path.join(root, "profiles", "roofline_traffic.json")))
    md = open(os.path.join(root, blob["_source"])).read()
    totals = {m.group(1): float(m.group(2)) for m in
              re.finditer(r"^\| (\S+) \| \*\*kernels of one measured call.*\*\*([0-9.]+)\*\* \|$", md, re.M)}
    assert totals, "no per-call totals found in " + blob["_source"]
    src = open(os.path.join(root, "tools", "pmc_summary.py")).read()
    keys = dict(re.findall(r'^    "(\w+)": "([^"]+)",$', src, re.M))
    bench_src = open(os.path.join(root, "bench.py")).read()
    seen = 0
    for work, total_mb in totals.items():
        key = keys.get(work, work)
        assert key in blob, f"{key} (workload {work}) missing from roofline_traffic.json"
        assert round(blob[key] / 1e6, 1) == pytest.approx(total_mb, abs=0.051), (key, blob[key], total_mb)
        stem = key.split(":")[0]
        assert stem in bench_src, f"bench.py never asks for traffic key {key}"
        seen += 1
    assert seen == len([k for k in blob if not k.startswith("_")])
