"""tgp_edge_facts_sorted_i64 (csrc/densify.hip): the one-launch check of a NEW row-sorted edge list in front of the
sparse-input dense pooling call (reference src.py:434-450 assumes a sorted batch vector and PyG-ordered edges)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _facts(ei, batch):
    from tgp import kernels as K
    from tgp.utils.ops import batch_info
    handle = K.edge_facts_launch(ei, batch)
    assert handle is not None
    state, tag, buf = handle
    flags = state.wait_facts(tag)[1]
    return flags, buf


def test_edge_ranges_of_a_sorted_batch(dev):
    batch = torch.tensor([0, 0, 0, 1, 1, 3, 3, 3], device=dev)  # graph 2 is empty
    ei = torch.tensor([[0, 1, 2, 3, 4, 5, 6, 7], [1, 0, 1, 4, 3, 6, 5, 6]], device=dev)
    flags, buf = _facts(ei, batch)
    assert flags == 0
    assert buf[:5].tolist() == [0, 3, 5, 5, 8]


def test_unsorted_rows_are_reported(dev):
    batch = torch.tensor([0, 0, 1, 1], device=dev)
    ei = torch.tensor([[2, 0, 1, 3], [3, 1, 0, 2]], device=dev)
    flags, _ = _facts(ei, batch)
    assert flags & 1


def test_last_graph_id_beyond_the_node_count_is_rejected_without_writing(dev):
    """batch = [0, 0, 50] passed the per-entry checks; the tail then wrote edge_ptr[3 .. 51] into a
    buffer of n + 2 = 5 entries.  Now the flags report a malformed batch vector and nothing beyond the buffer is touched."""
    from tgp import _native as N
    n = 3
    batch = torch.tensor([0, 0, 50], device=dev)
    ei = torch.tensor([[0, 1], [1, 0]], device=dev)  # (the last entry's graph is 0: the tail would run from 1 to 51)
    # call the entry directly on a guarded buffer: n + 2 entries, then 64 canaries
    from tgp import kernels as K
    st = N.stream_ptr(dev)
    state = K._sps_state(dev, st, 0)
    guard = torch.full((n + 2 + 64,), -7, dtype=torch.long, device=dev)
    tag = state.next_facts_tag()
    N.check(N.lib().tgp_edge_facts_sorted_i64(N.ptr(ei[0].contiguous()), 2, N.ptr(batch), n, N.ptr(guard),
                                              state.ticket.data_ptr() + 8, state.facts_slot(tag), tag, st),
            "tgp_edge_facts_sorted_i64")
    flags = state.wait_facts(tag)[1]
    assert flags & 1
    assert (guard[n + 2:] == -7).all()
