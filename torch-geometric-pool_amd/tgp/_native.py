"""ctypes binding of the tgp HIP library (``include/tgp_hip.h``).

There is deliberately **no CPU or PyTorch fallback** behind these wrappers: if
``lib/libtgp_hip.so`` is missing, or an operator is handed tensors that do not live on a
ROCm device, the call raises.  PyTorch is used for device memory (outputs and workspaces
come from its caching allocator) and for the current HIP stream, nothing else.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch
from torch import Tensor

_HERE = os.path.dirname(os.path.abspath(__file__))
# TGP_HIP_LIB: load another build of the same library (kernel experiments); never a different implementation
LIB_PATH = os.environ.get("TGP_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libtgp_hip.so")

# flag bits / enums, mirrored from include/tgp_hip.h
REMOVE_SELF_LOOPS = 1
DEGREE_NORM = 2
EDGE_WEIGHT_NORM = 4
SUM_AXIS_ROWS = 8
EPS_FILTER = 16
ADJ_TRANSPOSED = 32
NODE_FILTER = 64
WANT_EDGE_ID = 128
HUGE_ROWS = 256
REDUCE_OPS = {"sum": 0, "add": 0, "mean": 1, "min": 2, "max": 3, "mul": 4}

_c_i64, _c_int, _c_sz, _c_p, _c_f = ctypes.c_int64, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_float

# name -> (restype, argtypes); must list every symbol the header declares
SIGNATURES = {
    "tgp_version": (_c_int, []),
    "tgp_last_error": (ctypes.c_char_p, []),
    "tgp_device_cu_count": (_c_int, []),
    "tgp_assign_index_workspace_bytes": (_c_sz, [_c_i64, _c_i64]),
    "tgp_assign_index_build": (_c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_p, _c_p, _c_sz, _c_p]),
    "tgp_reduce_sparse_f32": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_i64,
                                       _c_i64, _c_p, _c_p]),
    "tgp_reduce_batch_i64": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_int, _c_p, _c_p]),
    "tgp_connect_subgraph_workspace_bytes": (_c_sz, [_c_i64, _c_i64]),
    "tgp_connect_subgraph_count": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_int, _c_f, _c_p,
                                            _c_sz, _c_p, _c_p]),
    "tgp_connect_subgraph_fill": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_int, _c_f, _c_p, _c_i64, _c_p,
                                           _c_p, _c_p, _c_p, _c_p]),
    "tgp_connect_subgraph_single_workspace_bytes": (_c_sz, [_c_i64]),
    "tgp_connect_subgraph_single_status_words": (_c_i64, [_c_i64]),
    "tgp_connect_subgraph_single": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_int, _c_f, _c_p, _c_sz, _c_p,
                                             _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_p, ctypes.c_uint32, _c_p]),
    "tgp_connect_coalesce_workspace_bytes": (_c_sz, [_c_i64, _c_i64, _c_i64]),
    "tgp_connect_coalesce_count": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_int, _c_int, _c_f,
                                            _c_p, _c_sz, _c_p, _c_p]),
    "tgp_connect_coalesce_fill": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_i64, _c_p, _c_p,
                                           _c_p, _c_p]),
    "tgp_connect_coalesce_rows_workspace_bytes": (_c_sz, [_c_i64, _c_i64, _c_i64]),
    "tgp_connect_coalesce_rows_huge_workspace_bytes": (_c_sz, [_c_i64, _c_i64, _c_i64]),
    "tgp_connect_coalesce_rows_count": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_p, _c_p, _c_p,
                                                 _c_int, _c_int, _c_f, _c_p, _c_sz, _c_p, _c_p]),
    "tgp_count_publish": (_c_int, [_c_p, _c_p, ctypes.c_uint32, _c_p]),
    "tgp_connect_coalesce_rows_count_status_words": (_c_i64, [_c_i64, _c_i64]),
    "tgp_connect_coalesce_rows_count_published": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_p,
                                                           _c_p, _c_p, _c_int, _c_int, _c_f, _c_p, _c_sz, _c_p, _c_p,
                                                           _c_i64, _c_p, ctypes.c_uint32, _c_p]),
    "tgp_connect_coalesce_rows_fill": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_int, _c_i64, _c_p, _c_p, _c_p,
                                                _c_p]),
    "tgp_connect_coalesce_rows_workspace_bytes_f64": (_c_sz, [_c_i64, _c_i64, _c_i64]),
    "tgp_connect_coalesce_rows_count_published_f64": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_p,
                                                               _c_p, _c_p, _c_int, _c_int, ctypes.c_double, _c_p, _c_sz,
                                                               _c_p, _c_p, _c_i64, _c_p, ctypes.c_uint32, _c_p]),
    "tgp_connect_coalesce_rows_fill_f64": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_p, _c_p, _c_p, _c_p]),
    "tgp_connect_coalesce_fused_workspace_bytes": (_c_sz, [_c_i64, _c_i64, _c_i64]),
    "tgp_connect_coalesce_fused_count": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_p,
                                                  _c_p, _c_int, _c_int, _c_f, _c_p, _c_p, _c_sz, _c_p, _c_p]),
    "tgp_connect_coalesce_fused_fill": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_p, _c_p, _c_p]),
    "tgp_sparse_pool_small_max_graph_nodes": (_c_int, []),
    "tgp_sparse_pool_small_status_words": (_c_i64, [_c_i64, _c_int]),
    "tgp_graph_lower_bounds_i64": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_p, _c_p]),
    "tgp_sparse_pool_small_f32": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_p, _c_p,
                                           _c_p, _c_i64, _c_i64, _c_int, _c_int, _c_int, _c_f, _c_p, _c_p, _c_p, _c_p, _c_p,
                                           _c_p, _c_i64, _c_p, ctypes.c_uint32, _c_p]),
    "tgp_connect_coalesce_grouped_workspace_bytes": (_c_sz, [_c_i64, _c_i64, _c_i64]),
    "tgp_connect_coalesce_grouped_count": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_int, _c_int,
                                                    _c_f, _c_p, _c_sz, _c_p, _c_p]),
    "tgp_postprocess_sparse_workspace_bytes": (_c_sz, [_c_i64, _c_i64, _c_i64]),
    "tgp_postprocess_sparse_norm_f32": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_int, _c_f, _c_p, _c_i64,
                                                 _c_p, _c_sz, _c_p]),
    "tgp_dense_pool_workspace_bytes": (_c_sz, [_c_i64, _c_i64, _c_i64, _c_i64]),
    "tgp_dense_pool_f32": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_f, _c_p, _c_p, _c_p,
                                    _c_p, _c_p, _c_sz, _c_p]),
    "tgp_dense_pool_is_small": (_c_int, [_c_i64, _c_i64, _c_i64, _c_i64]),
    "tgp_dense_pool_mincut_f32": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_f, _c_f, _c_p,
                                           _c_p, _c_p, _c_p, _c_p, _c_sz, _c_p]),
    "tgp_dense_pool_select_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_f, _c_f,
                                           _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p]),
    "tgp_edge_facts_sorted_i64": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_p, _c_p, _c_p, ctypes.c_uint64, _c_p]),
    "tgp_dense_pool_select_sparse_f32": (_c_int, [_c_p, _c_i64, _c_p, _c_p, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i64,
                                                  _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_f, _c_f, _c_p, _c_p, _c_p,
                                                  _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p]),
    "tgp_diffpool_stats_tail_f32": (_c_int, [_c_p, _c_i64, _c_f, _c_f, _c_p, _c_p]),
    "tgp_dense_pool_small_diff_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_int,
                                               _c_f, _c_f, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p]),
    "tgp_dense_pool_small_bwd_f32": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_f, _c_f, _c_p,
                                              _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_f, _c_f, _c_f, _c_int,
                                              _c_p, _c_p, _c_p]),
    "tgp_postprocess_dense_workspace_bytes": (_c_sz, [_c_i64, _c_i64]),
    "tgp_postprocess_dense_bwd_f32": (_c_int, [_c_p, _c_p, _c_i64, _c_i64, _c_int, _c_f, _c_p, _c_p]),
    "tgp_postprocess_dense_f32": (_c_int, [_c_p, _c_p, _c_i64, _c_i64, _c_int, _c_f, _c_p, _c_sz, _c_p]),
    "tgp_bmm_f32": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_i64, _c_i64,
                             _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
    "tgp_bmm_accumulate_f32": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_i64, _c_i64,
                                        _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
    "tgp_segment_gemm_tn_workspace_bytes": (_c_sz, [_c_i64, _c_i64, _c_i64, _c_i64]),
    "tgp_segment_gemm_tn_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64,
                                         _c_p, _c_sz, _c_p]),
    "tgp_segment_gemm_nn_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
    "tgp_topk_plan": (_c_int, [_c_p, _c_i64, ctypes.c_double, _c_p, _c_p, _c_p]),
    "tgp_topk_select_workspace_bytes": (_c_sz, [_c_i64]),
    "tgp_topk_select": (_c_int, [_c_p, _c_p, _c_i64, _c_i64, _c_p, _c_p, _c_p, _c_i64, _c_p, _c_sz, _c_p, _c_p, _c_p,
                                 _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p]),
    "tgp_topk_select_directory_blocks": (_c_i64, [_c_i64]),
    "tgp_one_to_one_index_build": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_p, _c_p, _c_p]),
    "tgp_reduce_one_to_one_f32": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_p, _c_int, _c_i64, _c_p, _c_p]),
    "tgp_topk_minscore_workspace_bytes": (_c_sz, [_c_i64, _c_i64]),
    "tgp_topk_minscore_count": (_c_int, [_c_p, _c_p, _c_i64, _c_i64, _c_f, _c_f, _c_p, _c_p, _c_sz, _c_p, _c_p]),
    "tgp_topk_minscore_fill": (_c_int, [_c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_p, _c_p]),
    "tgp_graclus_match_workspace_bytes": (_c_sz, [_c_i64, _c_i64]),
    "tgp_graclus_match_start": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_p, _c_sz, _c_p, _c_int, _c_p]),
    "tgp_graclus_relabel_max_nodes": (_c_i64, []),
    "tgp_graclus_relabel_workspace_bytes": (_c_sz, [_c_i64]),
    "tgp_graclus_relabel_i64": (_c_int, [_c_p, _c_i64, _c_p, _c_sz, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p]),
    "tgp_graclus_match_tail": (_c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_p, _c_p, _c_p]),
    "tgp_graclus_match_max_graph_nodes": (_c_int, []),
    "tgp_graclus_match_graphs_fused_max_graph_nodes": (_c_int, []),
    "tgp_graclus_match_graphs_fused_status_words": (_c_i64, [_c_i64]),
    "tgp_graclus_match_graphs_fused": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_p,
                                                _c_p, _c_p, _c_p, _c_i64, _c_p, ctypes.c_uint32, _c_p]),
    "tgp_graclus_match_graphs": (_c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_p, _c_i64, _c_i64, _c_p, _c_p, _c_p]),
    "tgp_graclus_match_rounds": (_c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_int, _c_p, _c_p, _c_p]),
    "tgp_batch_facts_i64": (_c_int, [_c_p, _c_i64, _c_p, _c_p, ctypes.c_double, _c_p]),
    "tgp_batch_facts_sorted_i64": (_c_int, [_c_p, _c_i64, _c_p, _c_p, ctypes.c_double, _c_p, _c_p, _c_p, _c_p,
                                            ctypes.c_uint64, _c_p]),
    "tgp_topk_score_f32": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_p, _c_int, _c_p, _c_p]),
    "tgp_row_dot_f32": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_p, _c_p, _c_p]),
    "tgp_weighted_colsum_workspace_bytes": (_c_sz, [_c_i64]),
    "tgp_weighted_colsum_f32": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_p, _c_p, _c_p, _c_sz, _c_p]),
    "tgp_topk_pool_bwd_fits": (_c_int, [_c_i64]),
    "tgp_topk_pool_bwd_workspace_bytes": (_c_sz, [_c_i64]),
    "tgp_topk_pool_bwd_f32": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_p, _c_p, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_int,
                                       _c_p, _c_p, _c_p, _c_sz, _c_p]),
    "tgp_pair_dot_f32": (_c_int, [_c_p, _c_p, _c_i64, _c_p, _c_p, _c_i64, _c_p, _c_p]),
    "tgp_edge_dot_f32": (_c_int, [_c_p, _c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_p, _c_p]),
    "tgp_link_loss_workspace_bytes": (_c_sz, [_c_i64, _c_i64, _c_i64]),
    "tgp_link_loss_f32": (_c_int, [_c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_p, _c_p, _c_p, _c_sz, _c_p]),
    "tgp_entropy_sum_workspace_bytes": (_c_sz, [_c_i64]),
    "tgp_entropy_bwd_f32": (_c_int, [_c_p, _c_i64, _c_f, _c_p, _c_f, _c_p, _c_p]),
    "tgp_entropy_sum_f32": (_c_int, [_c_p, _c_i64, _c_f, _c_p, _c_p, _c_sz, _c_p]),
    "tgp_cut_terms_f32": (_c_int, [_c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_p]),
    "tgp_mincut_loss_terms_bwd_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_f, _c_p, _c_p, _c_p, _c_p]),
    "tgp_dense_pool_train_workspace_bytes": (_c_sz, [_c_i64, _c_i64, _c_i64, _c_i64]),
    "tgp_dense_pool_train_fwd_f32": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_f, _c_p, _c_i64,
                                              _c_p, _c_p, _c_p, _c_p, _c_p, _c_sz, _c_p]),
    "tgp_mincut_terms_fused_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_f, _c_p, _c_p, _c_p,
                                            _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p]),
    "tgp_segment_gemm_tn3_workspace_bytes": (_c_sz, [_c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64]),
    "tgp_segment_gemm_tn3_post_workspace_bytes": (_c_sz, [_c_i64, _c_i64, _c_i64, _c_i64, _c_i64]),
    "tgp_pool_rows_fwd_workspace_bytes": (_c_sz, [_c_i64, _c_i64, _c_i64, _c_i64, _c_i64]),
    "tgp_pool_rows_bwd_f32": (_c_int, [_c_p] * 13 + [_c_i64] * 6 + [_c_int, _c_f, _c_f, _c_int, _c_int, _c_f, _c_f, _c_f, _c_p,
                                       _c_int, _c_p, _c_p, _c_int, _c_p, _c_p, _c_p] + [_c_p] * 10 + [_c_p]),
    "tgp_pool_rows_fwd_f32": (_c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_p, _c_i64, _c_i64,
                                       _c_i64, _c_int, _c_int, _c_f, _c_f, _c_int, _c_p, _c_f, _c_f, _c_f, _c_p, _c_p, _c_p,
                                       _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_sz, _c_p]),
    "tgp_segment_gemm_tn3_post_f32": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i64,
                                               _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_f, _c_p, _c_sz, _c_p]),
    "tgp_segment_gemm_tn3_f32": (_c_int, [_c_p, _c_p, _c_i64, _c_p, _c_i64, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_i64,
                                          _c_i64, _c_i64, _c_i64, _c_int, _c_p, _c_sz, _c_p]),
    "tgp_edge_row_stats_f32": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_p, _c_p, _c_p]),
    "tgp_diffpool_unbatched_tail_f32": (_c_int, [_c_p, _c_p, _c_i64, _c_i64, _c_p, _c_f, _c_p, _c_int, _c_f, _c_f, _c_p,
                                                 _c_p, _c_p]),
    "tgp_dense_pool_train_rhs_f32": (_c_int, [_c_p, _c_p, _c_int, _c_p, _c_p, _c_p, _c_p, _c_p, _c_f, _c_p, _c_f, _c_f,
                                              _c_p, _c_int, _c_int, _c_p, _c_i64, _c_i64, _c_i64, _c_p, _c_p, _c_p,
                                              _c_p]),
    "tgp_softmax_bwd_ex_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_p, _c_f, _c_f, _c_p, _c_i64, _c_i64,
                                        _c_i64, _c_p, _c_p]),
    "tgp_segment_gemm_nn_ld_f32": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64,
                                            _c_i64, _c_i64, _c_p]),
    "tgp_segment_gemm_tn_ld_f32": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
    "tgp_dense_symmetry_f32": (_c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_p, ctypes.c_uint64, _c_p]),
    "tgp_edge_symmetry_f32": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_p, _c_i64, _c_p, _c_p, ctypes.c_uint64, _c_p]),
    "tgp_slab_sum_split_f32": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_p, _c_p, _c_p]),
    "tgp_adj_symmetry_f32": (_c_int, [_c_p, _c_p, _c_i64, _c_p, _c_p, _c_i64, _c_p, _c_p, _c_p, ctypes.c_uint64, _c_p]),
    "tgp_copy_cols2_f32": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
    "tgp_copy_cols3_f32": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_i64, _c_i64, _c_i64,
                                    _c_i64, _c_p]),
    "tgp_mincut_loss_terms_f32": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_f, _c_p, _c_p]),
    "tgp_rowptr_from_sorted_flag_i64": (_c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_p, _c_p]),
    "tgp_rowptr_from_sorted_i64": (_c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_p]),
    "tgp_spmm_csr_f32": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_p, _c_p]),
    "tgp_spmm_csr_stats_f32": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_p]),
    "tgp_spmm_csr_entropy_f32": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_p, _c_f, _c_p, _c_p, _c_p]),
    "tgp_to_dense_adj_f32": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_p, _c_p, _c_i64, _c_i64, _c_int, _c_int, _c_p, _c_p]),
    "tgp_to_dense_adj_channels_f32": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_p, _c_p, _c_i64, _c_i64, _c_int, _c_p,
                                               _c_p]),
    "tgp_from_dense_adj_f32": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_p, _c_p, _c_i64, _c_i64, _c_int, _c_p, _c_p]),
    "tgp_from_dense_batch_f32": (_c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_p, _c_i64, _c_i64, _c_p, _c_p]),
    "tgp_to_dense_batch_sorted_f32": (_c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_i64, _c_p, _c_p, _c_p, _c_i64, _c_p]),
    "tgp_entropy_partials_f32": (_c_int, [_c_p, _c_i64, _c_f, _c_p, _c_sz, _c_p, _c_p]),
    "tgp_diffpool_loss_tail_f32": (_c_int, [_c_p, _c_i64, _c_p, _c_int, _c_f, _c_f, _c_p, _c_p]),
    "tgp_to_dense_batch_f32": (_c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_p, _c_i64, _c_i64, _c_p, _c_p, _c_p]),
    "tgp_block_diag_workspace_bytes": (_c_sz, [_c_i64, _c_i64]),
    "tgp_block_diag_count": (_c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_int, _c_f, _c_p, _c_sz, _c_p, _c_p]),
    "tgp_block_diag_fill": (_c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_int, _c_f, _c_p, _c_i64, _c_p, _c_p, _c_p,
                                     _c_p]),
    "tgp_ndp_symmetric_max_f32": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_p, _c_p, _c_p, _c_p]),
    "tgp_ndp_max_graph_nodes": (_c_int, []),
    "tgp_ndp_partition": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_i64, ctypes.c_uint64, _c_int,
                                   ctypes.c_double, _c_p, _c_p, _c_p, _c_p]),
    "tgp_ndp_large_workspace_bytes": (_c_sz, [_c_i64]),
    "tgp_ndp_large_start": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_int, _c_p, _c_sz, _c_p, _c_p]),
    "tgp_ndp_large_steps": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_int, ctypes.c_double, _c_p, _c_sz, _c_p,
                                     _c_p, _c_p]),
    "tgp_ndp_large_finish": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, ctypes.c_uint64, _c_p, _c_sz, _c_p, _c_p,
                                      _c_p]),
    "tgp_ndp_large_state": (_c_int, [_c_p, _c_i64, _c_p, _c_p]),
    "tgp_kron_batched_workspace_bytes": (_c_sz, [_c_i64, _c_i64, _c_i64, _c_i64, _c_i64]),
    "tgp_kron_batched_max_graph_nodes": (_c_int, []),
    "tgp_kron_batched_count": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_i64, _c_i64, _c_p, _c_i64, _c_i64,
                                        _c_i64, _c_i64, _c_i64, _c_p, _c_i64, ctypes.c_double, _c_p, _c_p, _c_sz, _c_p,
                                        _c_p]),
    "tgp_kron_batched_fill": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p, _c_i64, _c_p, _c_p,
                                       _c_p, _c_p, _c_p]),
    "tgp_mlp_select_max_fused_k": (_c_int, []),
    "tgp_mlp_select_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_p, _c_p]),
    "tgp_softmax_rows_f32": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_p]),
    "tgp_softmax_bwd_f32": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_p]),
    "tgp_mlp_select_bwd_fits": (_c_int, [_c_i64, _c_i64]),
    "tgp_mlp_select_bwd_workspace_bytes": (_c_sz, [_c_i64, _c_i64, _c_i64]),
    "tgp_mlp_select_bwd_f32": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_p, _c_int, _c_p, _c_p, _c_p,
                                        _c_sz, _c_p]),
    "tgp_reduce_sparse_f64": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_p, _c_p]),
    "tgp_connect_subgraph_single_f64": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_int, ctypes.c_double,
                                                 _c_p, _c_sz, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_p,
                                                 ctypes.c_uint32, _c_p]),
    "tgp_connect_coalesce_workspace_bytes_f64": (_c_sz, [_c_i64, _c_i64, _c_i64]),
    "tgp_connect_coalesce_count_f64": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_int, _c_int,
                                                ctypes.c_double, _c_p, _c_sz, _c_p, _c_p]),
    "tgp_connect_coalesce_fill_f64": (_c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_i64, _c_p, _c_p, _c_p,
                                               _c_p]),
    "tgp_postprocess_sparse_workspace_bytes_f64": (_c_sz, [_c_i64, _c_i64, _c_i64]),
    "tgp_postprocess_sparse_norm_f64": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_int, ctypes.c_double, _c_p, _c_i64,
                                                 _c_p, _c_sz, _c_p]),
    "tgp_block_diag_count_f64": (_c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_int, ctypes.c_double, _c_p, _c_sz, _c_p, _c_p]),
    "tgp_block_diag_fill_f64": (_c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_int, ctypes.c_double, _c_p, _c_i64, _c_p, _c_p,
                                         _c_p, _c_p]),
    "tgp_postprocess_dense_workspace_bytes_f64": (_c_sz, [_c_i64, _c_i64]),
    "tgp_postprocess_dense_f64": (_c_int, [_c_p, _c_p, _c_i64, _c_i64, _c_int, ctypes.c_double, _c_p, _c_sz, _c_p]),
    "tgp_edges_compact": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_p]),
    "tgp_copy_arrays": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_p]),
    "tgp_result_wait_pack_cols": (_c_int, [_c_p, ctypes.c_uint32, _c_p, _c_p, _c_p]),
    "tgp_mask_index_scratch_words": (_c_i64, [_c_i64]),
    "tgp_mask_index_count": (_c_int, [_c_p, _c_i64, _c_p, _c_p, _c_p, ctypes.c_uint32, _c_p]),
    "tgp_mask_index_fill": (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p]),
    "tgp_bmm_f64": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_i64, _c_i64, _c_i64, _c_i64,
                             _c_i64, _c_i64, _c_int, _c_p]),
    "tgp_dense_pool_workspace_bytes_f64": (_c_sz, [_c_i64, _c_i64, _c_i64, _c_i64]),
    "tgp_dense_pool_f64": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, ctypes.c_double, _c_p, _c_p,
                                    _c_p, _c_p, _c_sz, _c_p]),
    "tgp_segment_gemm_tn_workspace_bytes_f64": (_c_sz, [_c_i64, _c_i64, _c_i64, _c_i64]),
    "tgp_segment_gemm_tn_f64": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p, _c_sz,
                                         _c_p]),
    "tgp_segment_gemm_nn_f64": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
    "tgp_spmm_csr_f64": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_p, _c_p]),
    "tgp_gather_pack_bytes": (_c_i64, [_c_i64, _c_i64, _c_i64, _c_int]),
    "tgp_gather_pack_f32": (_c_int, [_c_p, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_int,
                                     _c_i64, _c_p, _c_p]),
    "tgp_gather_unpack_f32": (_c_int, [_c_p, _c_i64, _c_i64, _c_int, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_p,
                                       _c_p, _c_p, _c_p, _c_p, _c_p, ctypes.c_uint64, _c_p]),
    "tgp_gather_max_bucket_steps": (_c_int, []),
    "tgp_gather_pack_bucket_f32": (_c_int, [_c_p, _c_p, _c_int, _c_i64, _c_p, _c_p]),
    "tgp_gather_unpack_bucket_f32": (_c_int, [_c_p, _c_i64, _c_i64, _c_int, _c_i64, _c_int, _c_p, _c_p, _c_p]),
    "tgp_debug_sort_workspace_bytes": (_c_sz, [_c_i64]),
    "tgp_debug_sort_pairs_u64": (_c_int, [_c_p, _c_p, _c_i64, _c_int, _c_p, _c_p, _c_p, _c_sz, _c_p]),
}

_lib: Optional[ctypes.CDLL] = None


class TgpNativeError(RuntimeError):
    """Raised when the HIP library is missing or reports a non-zero status."""


def lib() -> ctypes.CDLL:
    """Load (once) and return the C-ABI library; raises loudly when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TgpNativeError(
                f"tgp HIP extension not built: {LIB_PATH} is missing. Run `python -c 'import "
                "__graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here == ABI mismatch: fail loudly
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib().tgp_last_error().decode(errors="replace")
        raise TgpNativeError(f"{what} failed with status {status}: {msg}")


def require_device(*tensors: Optional[Tensor]) -> torch.device:
    """All given tensors must live on one ROCm device; returns it."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise TgpNativeError(
                "tgp (MI355X build) runs Reduce/Connect only on ROCm device tensors; got a "
                f"{t.device} tensor. Move inputs to 'cuda' - there is no CPU fallback.  (The reference's operators also "
                "accept host tensors, reduce/base_reduce.py:141-155: this build does not.  Remedy: "
                "`pooler.to('cuda')` and `x.to('cuda')`, `edge_index.to('cuda')`, `batch.to('cuda')` before the call; the "
                "results come back on the device.)")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise TgpNativeError(f"tensors on different devices: {dev} vs {t.device}")
    if dev is None:
        raise TgpNativeError("no device tensor given")
    return dev


def ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(dev: torch.device) -> int:
    """Handle of torch's CURRENT stream on ``dev`` (what the library launches on).  The raw accessor costs ~0.3 us;
    building a torch.cuda.Stream object per call cost ~6 us, a quarter of the whole launch path of a 21 us kernel."""
    if _raw_stream is not None:
        return _raw_stream(dev.index if dev.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(dev).cuda_stream


def workspace(nbytes: int, dev: torch.device) -> Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)


def f64c(t: Tensor) -> Tensor:
    """fp64 + contiguous view/copy (the float64 value type of the HBM-bound operators)."""
    if t.dtype != torch.float64:
        t = t.to(torch.float64)
    return t if t.is_contiguous() else t.contiguous()


def f32c(t: Tensor) -> Tensor:
    """fp32 + contiguous view/copy (the dtype the path computes in)."""
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    return t if t.is_contiguous() else t.contiguous()


def i64c(t: Tensor) -> Tensor:
    if t.dtype != torch.int64:
        t = t.to(torch.int64)
    return t if t.is_contiguous() else t.contiguous()
