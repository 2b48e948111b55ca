"""Host-side behaviours the reference's own unit tests pin (containers, validation, error types / messages, repr
strings, cache flags), re-stated against this package.  Every test names the reference test it mirrors
(/root/reference/tests/...); none of them launches a kernel, so they run without a GPU."""
from unittest.mock import patch

import pytest
import torch

from tgp.connect import Connect, DenseConnect, KronConnect, SparseConnect, sparse_connect
from tgp.lift import BaseLift, Lift
from tgp.reduce import BaseReduce, Reduce
from tgp.select import MLPSelect, Select, SelectOutput, cluster_to_s
from tgp.src import BasePrecoarseningMixin, DenseSRCPooling, PoolingOutput, Precoarsenable, SRCPooling
from tgp.utils import ops
from tgp.utils.ops import (apply_dense_node_mask, connectivity_to_edge_index, connectivity_to_sparsetensor,
                           connectivity_to_torch_coo, dense_to_block_diag, expand_compacted_rows, get_assignments,
                           get_mask_from_dense_s, propagate_assignments_sparse)


# ------------------------------------------------------------------ tests/selection/test_base_select.py
def test_cluster_to_s_as_edge_index():  # :10-23
    ci, ni = torch.tensor([0, 1, 0, 2, 1]), torch.tensor([2, 0, 4, 3, 1])
    w = torch.tensor([0.5, 1.5, 2.5, 3.5, 4.5])
    ei, rw = cluster_to_s(ci, node_index=ni, weight=w, as_edge_index=True)
    assert ei.shape == (2, 5) and torch.equal(ei[0], ni) and torch.equal(ei[1], ci) and torch.equal(rw, w)


def test_selectoutput_from_cluster_index_and_default_s_inv():  # :26-67
    ci = torch.tensor([0, 1, 0])
    out = SelectOutput(s=None, node_index=None, num_nodes=3, cluster_index=ci, num_supernodes=2, weight=torch.ones(3))
    assert out.s.is_sparse and out.is_sparse and out.is_expressive
    a, b = out.s_inv.coalesce(), out.s.t().coalesce()
    assert torch.equal(a.indices(), b.indices()) and torch.allclose(a.values(), b.values())
    assert (out.num_nodes, out.num_supernodes) == (3, 2)
    assert torch.equal(out.node_index, torch.arange(3)) and torch.equal(out.cluster_index, ci)
    assert torch.equal(out.weight, torch.ones(3))


def test_selectoutput_s_inv_repr_clone_apply_and_moves():  # :70-120, 184-241
    with pytest.raises(ValueError):
        SelectOutput(s=torch.tensor([[[1.0, 2.0], [3.0, 4.0]]])).set_s_inv("not_a_method")
    s = torch.eye(2).unsqueeze(0)
    out = SelectOutput(s=s, s_inv=s.transpose(1, 2))
    assert "SelectOutput(" in repr(out) and "num_nodes=2" in repr(out) and "num_supernodes=2" in repr(out)
    c = out.clone()
    c.s[0, 0, 0] = 5.0
    assert not torch.equal(c.s, out.s)
    assert torch.equal(out.clone().apply(lambda t: t * 2).s, s * 2)
    assert out.clone().cpu().s.device.type == "cpu"
    assert isinstance(out.clone().detach(), SelectOutput) and isinstance(out.clone().detach_(), SelectOutput)
    out.s_inv = None
    out.cpu()
    so = SelectOutput(s=torch.eye(3).unsqueeze(0), batch=torch.tensor([0]), in_mask=torch.tensor([[True, True, False]]))
    meta = so.clone().to("meta")
    assert meta.s.device.type == meta.batch.device.type == meta.in_mask.device.type == "meta"
    so2 = SelectOutput(s=torch.eye(2).unsqueeze(0), batch=torch.tensor([0]))
    so2.cpu()
    assert so2.batch.device.type == "cpu"
    so3 = SelectOutput(s=torch.eye(2).unsqueeze(0))
    so3.requires_grad_(True)
    assert so3.s.requires_grad and so3.s_inv.requires_grad


def test_selectoutput_cuda_moves_batch(monkeypatch):  # :207-234
    class _Batch:
        def __init__(self):
            self.calls = []

        def cuda(self, device=None, non_blocking=False):
            self.calls.append((device, non_blocking))
            return self

    so = SelectOutput(s=torch.eye(2).unsqueeze(0))
    so.batch = _Batch()
    monkeypatch.setattr(SelectOutput, "apply", lambda self, func: self)
    assert so.cuda(device=1, non_blocking=True) is so and so.batch.calls == [(1, True)]
    so.batch = None
    assert so.cuda(device=0) is so and so.batch is None


def test_selectoutput_apply_reaches_extra_args():  # :123-154
    theta = torch.arange(4, dtype=torch.float32).view(2, 2)
    so = SelectOutput(s=torch.eye(2), theta=theta.clone(), nested=[theta.clone(), {"inner": theta.clone()}],
                      non_tensor="keep-me")
    so.apply(lambda t: t + 1)
    assert torch.equal(so.s, torch.eye(2) + 1) and torch.equal(so.theta, theta + 1)
    assert torch.equal(so.nested[0], theta + 1) and torch.equal(so.nested[1]["inner"], theta + 1)
    assert so.non_tensor == "keep-me"
    base = torch.eye(2)
    so = SelectOutput(s=base.unsqueeze(0), tuple_extra=(base.clone(), {"inner": base.clone()}))
    so._extra_args.add("missing_attr")
    so.apply(lambda t: t + 2)
    assert torch.equal(so.tuple_extra[0], base + 2) and torch.equal(so.tuple_extra[1]["inner"], base + 2)


def test_selectoutput_in_mask_and_out_mask_rules():  # :276-349
    s = torch.eye(4)
    with pytest.raises(ValueError, match="in_mask must be 2D"):
        SelectOutput(s=s, in_mask=torch.tensor([True, True, False, True]))
    assert SelectOutput(s=s.unsqueeze(0).expand(2, 4, 4), in_mask=torch.ones(2, 4, dtype=torch.bool)).in_mask.shape == (2, 4)
    with pytest.raises(ValueError, match="only supported for batched dense"):
        SelectOutput(s=s, in_mask=torch.ones(1, 4, dtype=torch.bool))
    with pytest.raises(ValueError, match="must have shape"):
        SelectOutput(s=s.unsqueeze(0).expand(2, 4, 4), in_mask=torch.ones(2, 5, dtype=torch.bool))
    s5 = torch.tensor([[1.0, 0, 0, 0], [0.5, 0.5, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1], [0, 1, 0, 0]])
    b = torch.tensor([0, 0, 0, 1, 1])
    mask = SelectOutput(s=s5, batch=b).out_mask
    assert torch.equal(mask, get_mask_from_dense_s(s5, b))
    assert mask.tolist() == [[True, True, True, False], [False, True, False, True]]
    assert SelectOutput(s=torch.eye(2)).out_mask.shape == (1, 2)
    assert SelectOutput(s=torch.ones((1, 2, 2, 2))).out_mask is None
    assert SelectOutput(s=torch.eye(3).unsqueeze(0), in_mask=torch.zeros((1, 3), dtype=torch.bool)).is_expressive is False


def test_selectoutput_invalid_init_select_base_and_weights():  # :352-377
    with pytest.raises(ValueError):
        SelectOutput(s="invalid_s_value")
    with pytest.raises(NotImplementedError):
        Select().forward(x=torch.randn(1, 1), edge_index=None)
    assert repr(Select()) == "Select()"
    sp = torch.sparse_coo_tensor(torch.tensor([[0, 1, 2], [1, 2, 0]]), torch.ones(3), size=(3, 3)).coalesce()
    assert SelectOutput(s=sp, weight=torch.tensor([0.5, 1.5, 2.5]), num_supernodes=3).num_nodes == 3


def test_assign_all_nodes_extra_args_follow():  # :408-439, 497-526
    so = SelectOutput(cluster_index=torch.tensor([0, 1]), node_index=torch.tensor([0, 2]), num_nodes=4, num_supernodes=2,
                      custom_attr="test_value", another_attr=42)
    assert {"custom_attr", "another_attr"} <= so._extra_args
    so._extra_args.add("non_existent_attr")
    new = so.assign_all_nodes(adj=torch.tensor([[0, 1, 2, 3], [1, 0, 3, 2]]), closest_node_assignment=True)
    assert new.custom_attr == "test_value" and new.another_attr == 42 and not hasattr(new, "non_existent_attr")
    assert (new.num_nodes, new.num_supernodes) == (4, 2)


# ------------------------------------------------------------------ tests/selection/test_mlp_select.py
def test_mlp_select_validation_and_repr():  # :56-87
    sel = MLPSelect(in_channels=[3, 4], k=2, batched_representation=False, act="relu", dropout=0.0, s_inv_op="transpose")
    assert "MLPSelect(" in repr(sel) and "in_channels=[3, 4]" in repr(sel) and "k=2" in repr(sel)
    sel.reset_parameters()
    with pytest.raises(AssertionError, match=r"x must be of shape \[N, F\]"):
        MLPSelect(in_channels=3, k=2, batched_representation=False)(x=torch.randn(1, 4, 3))


# ------------------------------------------------------------------ tests/utils/test_ops.py
def test_ops_validation_errors():  # :199-251, 272-363
    with pytest.raises(ValueError, match="adj_pool must have shape"):
        dense_to_block_diag(torch.tensor([1.0, 2.0, 3.0]))
    with pytest.raises(ValueError, match="s must have shape"):
        get_mask_from_dense_s(torch.ones(1, 1, 1, 1))
    mk = get_mask_from_dense_s(torch.tensor([[1.0, 0], [0, 1], [0.3, 0.7]]), batch=torch.tensor([0, 2, 2]))
    assert mk.shape == (3, 2) and mk[1].tolist() == [False, False]
    with pytest.raises(ValueError, match="expects x to be 3D"):
        apply_dense_node_mask(torch.randn(2, 3), torch.ones(2, 3, dtype=torch.bool))
    with pytest.raises(ValueError, match="expects mask shape"):
        apply_dense_node_mask(torch.randn(2, 3, 4), torch.ones(2, 2, dtype=torch.bool))
    with pytest.raises(ValueError, match="at least 1D"):
        expand_compacted_rows(torch.tensor(1.0), torch.tensor([True]), expected_rows=1)
    with pytest.raises(ValueError, match="must contain exactly 2 entries"):
        expand_compacted_rows(torch.randn(1, 3), None, expected_rows=2)
    with pytest.raises(ValueError, match="must contain exactly 2 entries"):
        expand_compacted_rows(torch.randn(1, 3), torch.tensor([True]), expected_rows=2)
    with pytest.raises(ValueError, match="x_compact has 2 rows"):
        expand_compacted_rows(torch.randn(2, 3), torch.tensor([True, False, False]), expected_rows=3)
    with pytest.raises(ValueError, match="Edge index must be of type Tensor or SparseTensor"):
        connectivity_to_torch_coo("invalid-edge-index", num_nodes=2)
    adj = torch.eye(4)
    for fn in (connectivity_to_edge_index, connectivity_to_torch_coo, connectivity_to_sparsetensor):
        for a in (adj, adj.unsqueeze(0)):
            with pytest.raises(ValueError, match="Dense adjacency matrices are not supported"):
                fn(a)
    bad1, badf = torch.tensor([0, 1, 2]), torch.tensor([[0, 1], [1, 0]], dtype=torch.float32)
    with patch("tgp.utils.ops.HAS_TORCH_SPARSE", False):
        for fn in (connectivity_to_edge_index, connectivity_to_torch_coo, connectivity_to_sparsetensor):
            with pytest.raises(ValueError, match="got a Tensor with 1 dimensions"):
                fn(bad1)
            with pytest.raises(ValueError, match=r"dtype=torch\.float32"):
                fn(badf)
        with pytest.raises(ImportError, match="Cannot convert connectivity to sparse tensor: torch_sparse is not installed"):
            connectivity_to_sparsetensor(torch.tensor([[0, 1], [1, 0]]))


def test_connectivity_to_torch_coo_defensive_else_branch(monkeypatch):  # :279-291
    class Dummy:
        pass

    calls = iter([True, False])
    monkeypatch.setattr(ops, "is_sparsetensor", lambda _x: next(calls))
    with pytest.raises(ValueError, match="Edge index must be a Tensor or SparseTensor."):
        ops.connectivity_to_torch_coo(Dummy(), num_nodes=2)


def test_assignment_propagation_corner_cases():  # :385-419
    a, m = torch.tensor([2, 0]), torch.tensor([True, False])
    oa, mp, om = propagate_assignments_sparse(a, torch.tensor([[0], [1]]), torch.tensor([0]), m, 1)
    assert torch.equal(oa, a) and mp.shape == (2, 0) and torch.equal(om, m)
    sp = torch.sparse_coo_tensor(torch.tensor([[0, 1], [1, 0]]), torch.ones(2), size=(2, 2)).coalesce()
    r = get_assignments([0], edge_index=sp, max_iter=1, num_nodes=2)
    assert r.shape == (2, 2) and r[0].tolist() == [0, 1]


# ------------------------------------------------------------------ tests/reduce/test_base_reduce.py
def test_reduce_batch_and_reduce_validation():  # :8-47, 75-90, 136-138
    so = SelectOutput(s=torch.randn(4, 2))
    out = BaseReduce.reduce_batch(so, torch.tensor([0, 0, 1, 1]))
    assert out.tolist() == [0, 0, 1, 1]
    assert BaseReduce.reduce_batch(SelectOutput(s=torch.randn(3, 2)), torch.zeros(3, dtype=torch.long)).tolist() == [0, 0]
    assert BaseReduce.reduce_batch(SelectOutput(s=torch.empty((0, 2))), torch.empty((0,), dtype=torch.long)).shape == (0,)
    sp = torch.sparse_coo_tensor(torch.tensor([[0, 1, 2], [0, 0, 1]]), torch.ones(3), size=(3, 2)).coalesce()
    with pytest.raises(ValueError, match="return_batched=True is only supported"):
        BaseReduce()(torch.randn(3, 2), SelectOutput(s=sp), return_batched=True)
    with pytest.raises(ValueError, match=r"Dense SelectOutput.s must be 2D \[N, K\]"):
        BaseReduce()(torch.randn(2, 3), SelectOutput(s=torch.randn(1, 2, 3, 4)))
    assert repr(Reduce()) == "Reduce()" and repr(BaseReduce()) == "BaseReduce()"


# ------------------------------------------------------------------ tests/connect/test_base_conn.py, test_dense_conn.py
def test_connect_base_classes_and_sparse_connect_errors():  # test_base_conn.py:16-60, 262-297
    assert "Connect()" in repr(Connect())
    with pytest.raises(NotImplementedError):
        Connect().forward(edge_index=torch.tensor([[0, 1], [1, 0]]), so=SelectOutput(s=torch.eye(3)))
    ei, ew = torch.tensor([[0, 1], [1, 0]]), torch.ones(2)
    with pytest.raises(RuntimeError) as e1:
        sparse_connect(edge_index=ei, edge_weight=ew, node_index=None, cluster_index=None)
    with pytest.raises(RuntimeError) as e2:  # dense assignment: neither index exists
        SparseConnect()(edge_index=ei, edge_weight=ew, so=SelectOutput(s=torch.randn(2, 1)))
    from tgp._native import TgpNativeError
    assert not isinstance(e1.value, TgpNativeError) and not isinstance(e2.value, TgpNativeError)
    r = repr(SparseConnect(reduce_op="mean", remove_self_loops=False, edge_weight_norm=True, degree_norm=True))
    for token in ("SparseConnect", "reduce_op=mean", "remove_self_loops=False", "edge_weight_norm=True", "degree_norm=True"):
        assert token in r
    so = SelectOutput(node_index=torch.tensor([0, 1]), cluster_index=torch.arange(2), num_nodes=2, num_supernodes=2)
    with pytest.raises(AssertionError, match="batch_pooled parameter is required"):
        SparseConnect(edge_weight_norm=True)(edge_index=ei, edge_weight=ew, so=so, batch_pooled=None)


def test_dense_connect_validation():  # test_dense_conn.py:17-36, 168-189, 189-258, 416-441
    with pytest.raises(TypeError, match="sparse_output must be a bool"):
        DenseConnect(sparse_output=1)
    r = repr(DenseConnect(remove_self_loops=False, degree_norm=True, adj_transpose=False, edge_weight_norm=True,
                          sparse_output=False))
    for token in ("DenseConnect", "remove_self_loops=False", "degree_norm=True", "adj_transpose=False",
                  "edge_weight_norm=True", "sparse_output=False"):
        assert token in r
    with pytest.raises(ValueError, match="SelectOutput is required"):
        DenseConnect()(torch.randn(1, 4, 4), so=None)
    sp = torch.sparse_coo_tensor(torch.tensor([[0, 1], [0, 1]]), torch.ones(2), size=(2, 2)).coalesce()
    with pytest.raises(ValueError, match="dense assignment matrix"):
        DenseConnect()(torch.randn(1, 2, 2), SelectOutput(s=sp))
    s2, a2 = DenseConnect._prepare_batched_dense_inputs(torch.randn(4, 2), torch.randn(4, 4))
    assert s2.shape == (1, 4, 2) and a2.shape == (1, 4, 4)
    s3, a3 = DenseConnect._prepare_batched_dense_inputs(torch.randn(2, 4, 2), torch.randn(2, 4, 4))
    assert s3.shape == (2, 4, 2) and a3.shape == (2, 4, 4)
    with pytest.raises(ValueError, match="batch sizes do not match"):
        DenseConnect._prepare_batched_dense_inputs(torch.randn(2, 4, 2), torch.randn(3, 4, 4))
    with pytest.raises(ValueError, match="Expected batched dense inputs with 3 dimensions"):
        DenseConnect._prepare_batched_dense_inputs(torch.randn(4), torch.randn(4, 4))

    class Dummy:
        def __init__(self, s):
            self.s = s

    with pytest.raises(TypeError, match="SelectOutput.s must be a torch.Tensor"):
        DenseConnect()(torch.randn(1, 2, 2), Dummy([[1.0, 0.0], [0.0, 1.0]]))
    ei = torch.tensor([[0, 1], [1, 0]])
    with pytest.raises(ValueError, match="Edge index must be of type"):
        DenseConnect(remove_self_loops=False, degree_norm=False, sparse_output=True)(
            edge_index=[[0, 1], [1, 0]], edge_weight=None, so=SelectOutput(s=torch.eye(2).unsqueeze(0)))
    for bad in (torch.randn(2, 4, 2), torch.randn(1, 2, 2, 2)):
        with pytest.raises(ValueError, match="SelectOutput.s must have shape"):
            DenseConnect(sparse_output=True)(edge_index=ei, edge_weight=None, so=SelectOutput(s=bad))
    assert "KronConnect" in repr(KronConnect())


# ------------------------------------------------------------------ tests/lift/test_base_lift.py
def test_lift_validation():  # :199-226, 240-263, 283-293, 408-413
    assert len(repr(Lift())) > 0
    r = repr(BaseLift(matrix_op="transpose", reduce_op="mean"))
    assert "BaseLift" in r and "matrix_op=transpose" in r and "reduce_op=mean" in r
    with pytest.raises(RuntimeError):
        BaseLift(matrix_op="invalid_op")(torch.randn(3, 2), SelectOutput(torch.randn(3, 3)))
    with pytest.raises(ValueError, match="Inconsistent per-graph blocks"):
        BaseLift._lift_dense_multi_graph(torch.tensor([[1.0, 0.0], [0.0, 1.0], [1.0, 1.0]]), torch.randn(6, 2),
                                         torch.tensor([0, 0, 1]), torch.tensor([0, 0, 1, 1, 2, 2]))
    so = SelectOutput(s=torch.tensor([[1.0, 0.0], [0.0, 1.0], [1.0, 0.0], [0.0, 1.0]]), batch=torch.tensor([0, 0, 1, 1]))
    with pytest.raises(ValueError, match="Unexpected pooled feature shape"):
        BaseLift(matrix_op="transpose")(torch.randn(3, 2), so)
    with pytest.raises(ValueError, match="batch_pooled has an unexpected length"):
        BaseLift(matrix_op="transpose")(torch.randn(4, 2), so, batch_pooled=torch.tensor([0, 0, 1]))
    with pytest.raises(ValueError, match="batch_pooled has an unexpected length"):
        BaseLift(matrix_op="transpose")(torch.randn(2, 2, 3), so, batch_pooled=torch.tensor([0, 0, 1]))


# ------------------------------------------------------------------ tests/test_src.py
class _DummyReducer(torch.nn.Module):
    @staticmethod
    def reduce_batch(select_output, batch):
        return torch.zeros(select_output.num_supernodes, dtype=batch.dtype, device=batch.device)

    def reset_parameters(self):
        pass


class _DummyConnector(torch.nn.Module):
    def forward(self, so, edge_index, edge_weight=None, **kwargs):
        return edge_index, edge_weight

    def reset_parameters(self):
        pass


class _DummyPrecoarseningPooler(BasePrecoarseningMixin, SRCPooling):
    def __init__(self):
        super().__init__(selector=None, reducer=_DummyReducer(), lifter=None, connector=_DummyConnector())


class _CountingPrecoarsenable(Precoarsenable):
    def __init__(self):
        self.calls = 0

    def precoarsening(self, edge_index=None, edge_weight=None, *, batch=None, num_nodes=None, **kwargs):
        self.calls += 1
        so = SelectOutput(cluster_index=torch.zeros(batch.numel(), dtype=torch.long), num_supernodes=1)
        return PoolingOutput(edge_index=edge_index, edge_weight=edge_weight, batch=batch, so=so)


def test_pooling_output_and_srcpooling_basics():  # :80-102
    assert SRCPooling().compute_loss() is None
    loss = {"quality": torch.tensor(1.0), "kl": torch.tensor(2.0)}
    out = PoolingOutput(loss=loss)
    assert out.get_loss_value() == [loss["quality"], loss["kl"]] and out.get_loss_value("quality") is loss["quality"]
    assert PoolingOutput(x=torch.randn(3, 2)).as_data().num_nodes == 3
    assert PoolingOutput(so=SelectOutput(s=torch.randn(4, 5))).as_data().num_nodes == 5
    assert PoolingOutput().as_data().num_nodes is None


def test_ensure_batched_inputs_edge_cases_and_cache_flags():  # :147-196
    pooler = DenseSRCPooling(cache_preprocessing=True)
    with pytest.raises(ValueError, match="edge_index cannot be None"):
        pooler._ensure_batched_inputs(x=torch.randn(2, 3), edge_index=None, edge_weight=None, batch=None, mask=None)
    dense_adj = torch.eye(2).unsqueeze(0)
    x_out, _, _ = pooler._ensure_batched_inputs(x=torch.randn(2, 3), edge_index=dense_adj, edge_weight=None, batch=None,
                                                mask=None, use_cache=False)
    assert x_out.shape == (1, 2, 3) and pooler.preprocessing_cache is None
    multi = DenseSRCPooling(cache_preprocessing=True)
    multi._ensure_batched_inputs(x=torch.randn(2, 2, 3), edge_index=dense_adj.repeat(2, 1, 1), edge_weight=None,
                                 batch=torch.tensor([0, 0, 1, 1]), mask=None, use_cache=True)
    assert multi.preprocessing_cache is None
    single = DenseSRCPooling(cache_preprocessing=True)
    _, adj_single, _ = single._ensure_batched_inputs(x=torch.randn(1, 2, 3), edge_index=dense_adj, edge_weight=None,
                                                     batch=torch.zeros(2, dtype=torch.long), mask=None, use_cache=True)
    assert single.preprocessing_cache is not None and torch.equal(single.preprocessing_cache, adj_single)


def test_precoarsening_plumbing():  # :259-318
    pre = _CountingPrecoarsenable()
    with pytest.raises(ValueError, match="'levels' must be >= 1"):
        pre.multi_level_precoarsening(levels=0, edge_index=torch.tensor([[0], [0]]))
    outs = pre.multi_level_precoarsening(levels=2, edge_index=torch.tensor([[0, 1], [1, 0]]), edge_weight=torch.ones(2),
                                         batch=torch.zeros(2, dtype=torch.long), num_nodes=2)
    assert len(outs) == 2 and pre.calls == 2
    pooler = _DummyPrecoarseningPooler()
    so = SelectOutput(cluster_index=torch.tensor([0, 0, 0]), num_supernodes=1)
    out = pooler._precoarsening_from_select_output(so=so, edge_index=torch.tensor([[0, 1], [1, 2]]),
                                                   edge_weight=torch.ones(2), batch=None)
    assert so.batch.tolist() == [0, 0, 0] and out.batch.numel() == 1 and int(out.batch[0]) == 0
    so = SelectOutput(cluster_index=torch.tensor([0, 0, 0]), num_supernodes=1, batch=torch.tensor([1, 1, 1]))
    out = pooler._precoarsening_from_select_output(so=so, edge_index=torch.tensor([[0, 1], [1, 2]]),
                                                   edge_weight=torch.ones(2), batch=None)
    assert so.batch.tolist() == [1, 1, 1] and int(out.batch[0]) == 0


def test_ndp_select_public_helpers():
    """NDPSelect.sign_partition / eval_cut (reference select/ndp_select.py:154-185)."""
    import torch
    from tgp.select import NDPSelect
    pos, neg = NDPSelect.sign_partition(torch.tensor([0.5, -1.0, 0.0, -0.1]))
    assert pos.tolist() == [0, 2] and neg.tolist() == [1, 3]
    pos, neg = NDPSelect.sign_partition(7)
    assert 0 in pos.tolist() and 1 in neg.tolist() and len(pos) + len(neg) == 7
    # a path 0-1-2: L = D - A, z = (+1, -1, +1): both edges are cut, volume 4 -> z^T L z / (2 vol) = 8 / 8
    L = torch.tensor([[1.0, -1.0, 0.0], [-1.0, 2.0, -1.0], [0.0, -1.0, 1.0]])
    z = torch.tensor([[1.0], [-1.0], [1.0]])
    assert float(NDPSelect.eval_cut(4.0, L, z)) == 1.0
    assert float(NDPSelect.eval_cut(4.0, L.to_sparse(), z)) == 1.0
