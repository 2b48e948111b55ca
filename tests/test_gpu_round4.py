"""Round-4 GPU parity tests (all through the C ABI): the dense in-place RCCL gather, the one-launch sparse pooling of
batches of small graphs, long rows in the coalesce Connect, float64 value types of the HBM-bound operators."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture()
def one_rank_rccl(dev):
    """A one-rank RCCL process group (created here unless the process already has one)."""
    import torch.distributed as dist
    created = False
    if not dist.is_initialized():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        created = True
    yield dist
    if created:
        dist.destroy_process_group()


def test_packed_gather_in_place_over_rccl_one_rank_group(dev, one_rank_rccl):
    """SURVEY 8(e), dense outputs: ``PackedGather.slots()`` hands the pooling call slices of the all-gather send buffer
    (``reduce_connect(out_x=, out_adj=)``: the kernels write INTO it, no pack copy) and the bucket goes out as one REAL
    RCCL collective on a one-rank group.  Gathered == a plain local call, bit for bit, for every step of two buckets
    (one full, one flushed partly filled), and the outputs really are the send buffer's memory."""
    from tgp.connect import DenseConnect
    from tgp.distributed import PackedGather
    from tgp.reduce import BaseReduce
    from tgp.select import SelectOutput
    from tgp.src import DenseSRCPooling
    g = torch.Generator(device=dev).manual_seed(3)
    B, N, K, F = 6, 256, 32, 24
    pool = DenseSRCPooling(reducer=BaseReduce(), connector=DenseConnect(), adj_transpose=True)
    pg = PackedGather(bucket_steps=2, force_collective=True)
    want, got = [], []
    for step in range(3):
        A = (torch.rand(B, N, N, device=dev, generator=g) < 0.05).float()
        A = torch.maximum(A, A.transpose(1, 2)).contiguous()
        X = torch.randn(B, N, F, device=dev, generator=g)
        S = torch.softmax(torch.randn(B, N, K, device=dev, generator=g), -1)
        so = SelectOutput(s=S)
        with torch.no_grad():
            x_ref, _, a_ref = pool.reduce_connect(X, A, so)
            ox, oa = pg.slots([(B, K, F), (B, K, K)], device=dev)
            x_pool, _, adj_pool = pool.reduce_connect(X, A, so, out_x=ox, out_adj=oa)
        assert x_pool.data_ptr() == ox.data_ptr() and adj_pool.data_ptr() == oa.data_ptr()
        want.append((x_ref.clone(), a_ref.clone()))
        pg.start([x_pool, adj_pool])
        got.extend(pg.take_ready())
    got.extend(pg.flush())
    assert len(got) == 3
    for (xw, aw), (xg, ag) in zip(want, got):
        assert xg.data_ptr() != xw.data_ptr()
        assert torch.equal(xg, xw) and torch.equal(ag, aw)
