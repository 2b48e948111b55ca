from .ops import (
    apply_dense_node_mask,
    build_pooled_batch,
    check_and_filter_edge_weights,
    connectivity_to_edge_index,
    connectivity_to_sparsetensor,
    connectivity_to_torch_coo,
    dense_to_block_diag,
    expand_compacted_rows,
    get_mask_from_dense_s,
    is_dense_adj,
    is_multi_graph_batch,
    postprocess_adj_pool_dense,
    postprocess_adj_pool_sparse,
    pseudo_inverse,
    rank3_diag,
    rank3_trace,
)
from .signature import Signature, foo_signature

__all__ = [
    "apply_dense_node_mask", "build_pooled_batch", "check_and_filter_edge_weights",
    "connectivity_to_edge_index", "connectivity_to_sparsetensor", "connectivity_to_torch_coo",
    "dense_to_block_diag", "expand_compacted_rows", "get_mask_from_dense_s", "is_dense_adj",
    "is_multi_graph_batch", "postprocess_adj_pool_dense", "postprocess_adj_pool_sparse", "pseudo_inverse",
    "rank3_diag", "rank3_trace", "Signature", "foo_signature",
]
