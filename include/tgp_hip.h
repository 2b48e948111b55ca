/*
 * tgp_hip.h — C ABI of the MI355X-native SRC pooling hot path (Reduce + Connect).
 *
 * The reference (tgp-team/torch-geometric-pool 1.0.1) has no FFI: its operator API is a
 * Python class protocol whose arithmetic is delegated to ATen / torch_geometric /
 * torch_scatter.  This header defines the boundary *underneath* that protocol: one entry
 * point per row of SURVEY.md section 8(a), each citing the reference lines it replaces.
 * The host-side mirror of the protocol (torch-geometric-pool_amd/tgp) binds these symbols
 * with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / PyTorch caching allocator) unless
 *     the name ends in `_host`; index tensors are int64 (reference: utils/ops.py:472-476),
 *     features / weights are fp32; `[2,E]` edge lists are given as two row pointers.
 *   - fp32 is the default value type (the reference's default dtype: torch.ones(E) at ops.py:385,547;
 *     base_select.py:64; north_star's 1e-5 tolerance is stated in it).  r4: the HBM-bound operators also exist for
 *     float64 values (`*_f64`: sparse Reduce, subgraph / coalesce Connect, block-diagonal export, sparse and dense
 *     post-processing; see the section near the end), because the reference's ATen ops compute model.double() inputs
 *     in fp64.  r5: the dense GEMM path has fp64 forms too (tgp_bmm_f64, tgp_dense_pool_f64, tgp_segment_gemm_*_f64,
 *     tgp_spmm_csr_f64 on the fp64 matrix instruction); only the fused small-graph kernels and the fused loss kernels
 *     are fp32-only, and the host mirror does not route float64 tensors to them.  Two algorithms whose RESULT depends on wider arithmetic run in fp64
 *     inside their kernels regardless of the I/O type: the Kron reduction (tgp_kron_batched_*) and NDPSelect's
 *     eigen-iteration (tgp_ndp_*); KronConnect also accepts fp64 Laplacian values (`val64`).
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on it and
 *     never synchronises, allocates or frees.  Scratch memory comes from the caller
 *     (`ws`, sized by the matching *_workspace_bytes()).
 *   - data-dependent output sizes use a count -> fill pair: *_count leaves the number of
 *     surviving edges in `*d_count` (device int64); the caller reads it (the one host
 *     sync the reference's own `.item()` calls also pay), allocates exact outputs and
 *     calls *_fill with the SAME workspace.
 *   - `eps` arguments: the reference reads its module-level `eps` (tgp/__init__.py:6, 1e-8) at CALL time
 *     (utils/ops.py:72,318,377,395; utils/losses.py:498; tests monkeypatch it), so it is an argument of every
 *     entry point that filters or clamps with it, never a compiled-in constant.
 *   - *d_count < 0 is a refusal, not a size: -1 = this fast path declines (preconditions not met: the caller
 *     takes its general route; -3 / -5: declines that name the entry point to call instead, see there), -2 = bad input (a node id outside [0, num_nodes) or a cluster id outside
 *     [0, num_supernodes): never dereferenced; the reference's index ops raise for it too).
 *   - return value 0 = ok; otherwise a negative tgp_status and tgp_last_error() holds a
 *     thread-local message.  No exceptions cross the boundary; no global mutable state.
 */
#ifndef TGP_HIP_H
#define TGP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TGP_ABI_VERSION 10042 /* 1.0.1 of the reference, ABI revision 41 (r6: tgp_mask_index_*, node_rank on tgp_kron_batched_count / _fill; the dense poolers' training step at C2 scale: tgp_dense_pool_train_*, tgp_mincut_terms_fused_f32, tgp_softmax_bwd_ex_f32, tgp_copy_cols2_f32; tgp_result_wait_pack_cols; tgp_spmm_csr_stats_f32 / _entropy_f32; tgp_segment_gemm_tn3_post_f32) */

enum tgp_status {
  TGP_OK = 0,
  TGP_ERR_INVALID = -1,   /* bad argument (null pointer, negative size, unknown enum) */
  TGP_ERR_WORKSPACE = -2, /* workspace too small */
  TGP_ERR_LAUNCH = -3,    /* HIP reported an error at launch */
  TGP_ERR_RANGE = -4      /* size beyond what the int32 internal indices can address */
};

/* coalesce reduction, reference: utils/typing.py ConnectionType, connect/base_conn.py:87 */
enum tgp_reduce_op { TGP_SUM = 0, TGP_MEAN = 1, TGP_MIN = 2, TGP_MAX = 3, TGP_MUL = 4 };

/* flag bits shared by the Connect / post-processing entry points */
enum tgp_flags {
  TGP_REMOVE_SELF_LOOPS = 1, /* utils/ops.py:370-371, 307-308 */
  TGP_DEGREE_NORM = 2,       /* utils/ops.py:383-401, 311-319 */
  TGP_EDGE_WEIGHT_NORM = 4,  /* utils/ops.py:404-417, 322-333 */
  TGP_SUM_AXIS_ROWS = 8,     /* dense: degree = sum over axis -2 (adj_transpose=True, ops.py:312-314) */
  TGP_EPS_FILTER = 16,       /* sparse: drop |w| <= 1e-8 when weights are given (ops.py:374-380) */
  TGP_ADJ_TRANSPOSED = 32,   /* dense: A is handed over as the transposed view (src.py:442-443) */
  TGP_NODE_FILTER = 64,      /* subgraph_fill: node_index was given to the matching _count call */
  TGP_WANT_EDGE_ID = 128,    /* subgraph_count + _fill: also stage / emit the input position of every kept edge */
  TGP_HUGE_ROWS = 256        /* coalesce_rows_count: supernode rows beyond 1024 raw entries (hubs) are sorted device-wide
                                instead of declining the call (workspace: ..._rows_huge_workspace_bytes) */
};

int tgp_version(void);
const char* tgp_last_error(void);
/* number of compute units of the current device (sizing persistent grids); <0 on error */
int tgp_device_cu_count(void);

/* ------------------------------------------------------------------------------------
 * A1  BaseReduce.forward, sparse path  (reduce/base_reduce.py:141-155)
 *     x_pool[c,:] = sum_{i: cluster_index[i]==c} weight[i] * x[node_index[i],:]
 * The inverted index (assignments grouped by supernode, ascending i inside a group) is a
 * function of the SelectOutput alone, so it is built once and may be cached by the caller.
 * ---------------------------------------------------------------------------------- */
size_t tgp_assign_index_workspace_bytes(int64_t nnz, int64_t num_supernodes);
int tgp_assign_index_build(const int64_t* cluster_index, int64_t nnz, int64_t num_supernodes,
                           int32_t* row_ptr /* [num_supernodes+1] */, int32_t* perm /* [nnz] */,
                           void* ws, size_t ws_bytes, void* stream);
int tgp_reduce_sparse_f32(const float* x, int64_t num_nodes, int64_t num_features, int64_t x_row_stride,
                          const int64_t* node_index, const float* weight /* may be NULL = ones */,
                          const int32_t* row_ptr /* NULL = one assignment per supernode (nnz == K): TopK, NDP */,
                          const int32_t* perm /* NULL = identity */, int64_t nnz,
                          int64_t num_supernodes, float* x_pool /* [K,F] contiguous */, void* stream);

/* A1 for selectors whose supernodes own exactly ONE node (TopK, NDP; r4), with the PACKED index: pack[c] = low word the
 * int32 source row node_index[a], high word the fp32 weight[a] of supernode c's single assignment a -- one streamed
 * 8-byte load per supernode instead of perm[c] + two random gathers.  x_pool[c,:] = 0 + weight * x[row,:], the bits of
 * tgp_reduce_sparse_f32.  tgp_one_to_one_index_build derives perm and pack from (node_index, cluster_index, weight) in
 * one launch (cluster_index a permutation of 0..k-1); tgp_topk_select emits the pack itself.  F % 4 == 0, F >= 32 and
 * 16-byte aligned rows (other shapes: tgp_reduce_sparse_f32). */
int tgp_one_to_one_index_build(const int64_t* node_index, const int64_t* cluster_index,
                               const float* weight /* NULL = ones */, int64_t k, int32_t* perm /* [k] */,
                               uint64_t* pack /* [k] */, void* stream);
int tgp_reduce_one_to_one_f32(const float* x, int64_t num_nodes, int64_t num_features, int64_t x_row_stride,
                              const uint64_t* pack, int has_weight, int64_t num_supernodes, float* x_pool, void* stream);

/* A2  Reduce.reduce_batch, sparse branch (reduce/base_reduce.py:37-41):
 *     out = arange(K); out[cluster_index[i]] = batch[node_index[i]]                      */
int tgp_reduce_batch_i64(const int64_t* batch, const int64_t* node_index, const int64_t* cluster_index,
                         int64_t nnz, int64_t num_supernodes,
                         int every_cluster_has_a_node /* 1: the arange is never visible (selectors that create a
                                                         supernode per kept node / matched pair): one launch */,
                         int64_t* batch_pool, void* stream);

/* ------------------------------------------------------------------------------------
 * A5 + A6  sparse_connect, TopK branch (connect/base_conn.py:79-82 -> PyG subgraph with
 * relabel_nodes=True) fused with the edge filters of postprocess_adj_pool_sparse
 * (utils/ops.py:370-380).  Output keeps INPUT edge order; endpoints are relabelled to
 * their position in the ascending `node_index`.  node_index == NULL turns the node
 * filter/relabel off (plain remove_self_loops + eps filter on an edge list).
 * ---------------------------------------------------------------------------------- */
size_t tgp_connect_subgraph_workspace_bytes(int64_t num_edges, int64_t num_nodes);
int tgp_connect_subgraph_count(const int64_t* row, const int64_t* col, const float* edge_weight /* NULL ok */,
                               int64_t num_edges, const int64_t* node_index /* NULL ok */, int64_t num_kept,
                               int64_t num_nodes, int flags, float eps, void* ws, size_t ws_bytes,
                               int64_t* d_count, void* stream);
int tgp_connect_subgraph_fill(const int64_t* row, const int64_t* col, const float* edge_weight,
                              int64_t num_edges, int64_t num_nodes, int flags, float eps, const void* ws,
                              int64_t num_out, int64_t* out_row, int64_t* out_col, float* out_weight,
                              int64_t* out_edge_id /* NULL ok: input position of every kept edge (the map the
                                                      backward of the weight pass-through needs); requires
                                                      TGP_WANT_EDGE_ID in `flags` of _count and _fill */,
                              void* stream);

/* The same operator in ONE pass (r4): survivors are written once, in their final int64 form at their final offsets of
 * CAPACITY-num_edges outputs (out_row / out_col / out_weight / out_edge_id: the first `total` entries are the result,
 * identical to the count -> fill pair above), through an epoch-tagged decoupled look-back over the 4096-edge chunks
 * (`status`: >= tgp_connect_subgraph_single_status_words(E) 64-bit words of device memory, caller-owned, never
 * cleared, one buffer per stream; 0 < epoch < 2^29 different for every call on it).  `*result` (device-accessible, e.g.
 * pinned host memory the caller polls) receives {epoch << 34 | refused << 31 | total} when the last chunk is done;
 * refused = an endpoint outside [0, num_nodes) was met (the count -> fill pair reports -2 for it), or a look-back
 * spin bound was hit (the device is shared: use the pair).  The two are told apart by word [1] of `status`: a chunk that
 * met a bad endpoint stores `epoch` into its low 32 bits (r5; r4 used a zeroed word of the workspace).  num_edges > 0.
 * member_bits_in / rank128_in (r5, NULL ok, both or neither): the kept-node bitmap [4 * blocks] and its rank directory
 * [blocks] (kept nodes with id < 128 b), blocks = tgp_topk_select_directory_blocks(num_nodes), as tgp_topk_select
 * writes them for the selection whose ASCENDING node_index is passed here.  The call is then ONE launch: no memset, no
 * scatter of the kept nodes into bitmap + relabel table, no directory scan in every workgroup (r4: 118 us per call at
 * N = 1M / E = 10M of which the stage kernel was 76).  Needs bitmap + directory to fit LDS (num_nodes <= ~1.2 M);
 * otherwise they are ignored. */
size_t tgp_connect_subgraph_single_workspace_bytes(int64_t num_nodes);
int64_t tgp_connect_subgraph_single_status_words(int64_t num_edges);
int tgp_connect_subgraph_single(const int64_t* row, const int64_t* col, const float* edge_weight /* NULL ok */,
                                int64_t num_edges, const int64_t* node_index /* NULL = filters only */, int64_t k,
                                int64_t num_nodes, int flags, float eps, void* ws, size_t ws_bytes, int64_t* out_row,
                                int64_t* out_col, float* out_weight, int64_t* out_edge_id /* NULL ok */,
                                const uint32_t* member_bits_in, const uint32_t* rank128_in,
                                uint64_t* status, int64_t status_words, uint64_t* result, uint32_t epoch, void* stream);

/* ------------------------------------------------------------------------------------
 * A4 + A6  sparse_connect, one-over-K branch (connect/base_conn.py:83-89 -> PyG coalesce):
 * relabel endpoints by cluster_index, sort by (row, col) (stable), merge duplicates with
 * `reduce_op`, then the filters of utils/ops.py:370-380.  Output is row-major sorted and
 * unique.  edge_weight == NULL keeps the result unweighted (out_weight unused).
 * ---------------------------------------------------------------------------------- */
size_t tgp_connect_coalesce_workspace_bytes(int64_t num_edges, int64_t num_nodes, int64_t num_supernodes);
int tgp_connect_coalesce_count(const int64_t* row, const int64_t* col, const float* edge_weight /* NULL ok */,
                               int64_t num_edges, const int64_t* cluster_index, int64_t num_nodes,
                               int64_t num_supernodes, int reduce_op, int flags, float eps, void* ws, size_t ws_bytes,
                               int64_t* d_count, void* stream);
int tgp_connect_coalesce_fill(const void* ws, int64_t num_edges, int64_t num_nodes, int64_t num_supernodes, int has_weight,
                              int flags, int64_t num_out, int64_t* out_row, int64_t* out_col,
                              float* out_weight, void* stream);

/* A4 + A6 fast path for ROW-SORTED edge lists (the PyG default): no global sort.  Edges are grouped by
 * supernode row through the CSR of the input and the supernode->member index of the assignment
 * (assign_row_ptr / assign_perm = what tgp_assign_index_build returns for cluster_index), then every short
 * row segment is ordered by column, merged and filtered inside LDS.  Same output as the sort-based pair
 * above.  If the rows are not sorted *d_count is set to -1 and the caller uses tgp_connect_coalesce_{count,fill}
 * instead.  A supernode row of more than 1024 raw entries (a hub of a power-law graph): *d_count = -5 unless
 * TGP_HUGE_ROWS is in `flags` (r4) -- then the entries of such rows alone go through a device-wide stable sort (at most
 * 4096 of them per call, else -1) while every other row stays in LDS, and `ws` must hold
 * tgp_connect_coalesce_rows_huge_workspace_bytes(); the matching _fill call takes the same workspace. */
size_t tgp_connect_coalesce_rows_workspace_bytes(int64_t num_edges, int64_t num_nodes, int64_t num_supernodes);
size_t tgp_connect_coalesce_rows_huge_workspace_bytes(int64_t num_edges, int64_t num_nodes, int64_t num_supernodes);
int tgp_connect_coalesce_rows_count(const int64_t* row, const int64_t* col, const float* edge_weight /* NULL ok */,
                                    int64_t num_edges, const int64_t* cluster_index, int64_t num_nodes,
                                    int64_t num_supernodes, const int32_t* assign_row_ptr,
                                    const int32_t* assign_perm,
                                    const int32_t* csr_ptr /* NULL, or CSR offsets [N+1] of this row-sorted list */,
                                    int reduce_op, int flags, float eps, void* ws,
                                    size_t ws_bytes, int64_t* d_count, void* stream);
int tgp_connect_coalesce_rows_fill(const void* ws, int64_t num_edges, int64_t num_nodes, int64_t num_supernodes,
                                   int has_weight /* bit 0: weights; bit 1: the count call ran with TGP_HUGE_ROWS */,
                                   int64_t num_out, int64_t* out_row, int64_t* out_col,
                                   float* out_weight, void* stream);

/* The host read of any count -> fill pair below without a device-to-host copy: stores
 * {epoch << 34 | *d_count as a 34-bit two's complement number} with system scope into `*result` (pinned host memory the
 * caller polls; epoch as for tgp_sparse_pool_small_f32).  Stands in for the reference's `.item()` reads
 * (connect/base_conn.py:79-82 via torch_geometric.utils.subgraph, utils/ops.py:370-380). */
int tgp_count_publish(const int64_t* d_count, uint64_t* result, uint32_t epoch, void* stream);

/* tgp_connect_coalesce_rows_count with its survivor scan as ONE launch that also hands the count to the host (r4):
 * the scan takes the survivors in front of every block of 4096 supernode rows from an epoch-tagged decoupled look-back
 * (`status`: >= ..._status_words(K, N) 64-bit words of device memory, caller-owned, never cleared, one buffer per stream;
 * 0 < epoch < 2^29 different for every call on it; the member segments of the front end take their offsets from a second
 * look-back over the same buffer -- member degree sums, their scan and the CSR check are ONE launch on this route) and its last workgroup stores {epoch << 34 | *d_count as a 34-bit
 * two's complement number} with system scope into `*result` (pinned host memory the caller polls; same word format as
 * tgp_count_publish).  *d_count is written as well; tgp_connect_coalesce_rows_fill follows as for the plain count.
 * `csr_col` (NULL ok, only with csr_ptr): the int32 copy of `col` that GraclusSelect's CSR holds for this very list --
 * half the column stream of the gather kernel; `col` is still needed (hub rows).
 * Measured alternatives (profiles/r04_coalesce_tail_experiments.md): the look-back inside the FILL (64-row tiles) and the
 * fill inside the count call (host wait at the end) are both slower. */
int64_t tgp_connect_coalesce_rows_count_status_words(int64_t num_supernodes, int64_t num_nodes);
int tgp_connect_coalesce_rows_count_published(const int64_t* row, const int64_t* col, const int32_t* csr_col /* NULL ok */,
                                              const float* edge_weight /* NULL ok */, int64_t num_edges,
                                              const int64_t* cluster_index, int64_t num_nodes, int64_t num_supernodes,
                                              const int32_t* assign_row_ptr, const int32_t* assign_perm,
                                              const int32_t* csr_ptr /* NULL ok */, int reduce_op, int flags, float eps,
                                              void* ws, size_t ws_bytes, int64_t* d_count, uint64_t* status,
                                              int64_t status_words, uint64_t* result, uint32_t epoch, void* stream);
/* r5: the same row-local pipeline for float64 edge weights (they took the general sort-based route before: 0.6 ms
 * against 0.24 at C4).  Values are staged, merged (sum / mean / min / max / mul) and eps-filtered in double, as the
 * reference's coalesce does for a double tensor (connect/base_conn.py:86-89).  Workspace of ..._workspace_bytes_f64;
 * TGP_HUGE_ROWS is refused (float32 only): a list with a supernode row beyond 1024 raw entries answers -5 and the caller
 * runs tgp_connect_coalesce_count_f64.  tgp_connect_coalesce_rows_fill_f64 follows a count >= 0. */
size_t tgp_connect_coalesce_rows_workspace_bytes_f64(int64_t num_edges, int64_t num_nodes, int64_t num_supernodes);
int tgp_connect_coalesce_rows_count_published_f64(const int64_t* row, const int64_t* col, const int32_t* csr_col /* NULL ok */,
                                                  const double* edge_weight, int64_t num_edges,
                                                  const int64_t* cluster_index, int64_t num_nodes, int64_t num_supernodes,
                                                  const int32_t* assign_row_ptr, const int32_t* assign_perm,
                                                  const int32_t* csr_ptr /* NULL ok */, int reduce_op, int flags,
                                                  double eps, void* ws, size_t ws_bytes, int64_t* d_count,
                                                  uint64_t* status, int64_t status_words, uint64_t* result,
                                                  uint32_t epoch, void* stream);
int tgp_connect_coalesce_rows_fill_f64(const void* ws, int64_t num_edges, int64_t num_nodes, int64_t num_supernodes,
                                       int64_t num_out, int64_t* out_row, int64_t* out_col, double* out_weight,
                                       void* stream);

/* A4 + A6, row-sorted input, as ONE heavy kernel + a widening fill (r3).  Every workgroup derives its rows' member
 * edge ranges, LDS slots and survivor counts locally; the survivors in front of it come from a decoupled look-back, so
 * weights are written once, at their final place, into `out_weight_cap` (capacity num_edges: the first *d_count
 * entries are the result), and tgp_connect_coalesce_fused_fill only widens the columns and generates the rows.
 * Optional CSR of the SAME edge list (csr_ptr [N+1], csr_col [E], int32; e.g. the one GraclusSelect builds of a
 * symmetric input): the pass over `row` (CSR offsets + sortedness check) is skipped and 4-byte columns are streamed;
 * row / col may then be NULL.  *d_count: >= 0 the output size; -1 declined (unsorted rows, a supernode row of more
 * than 1024 raw entries, ...: use the radix routes); -3 declined by this kernel's tile limit only (more than 256
 * members in a tile of rows): tgp_connect_coalesce_rows_{count,fill} takes the call.  Output identical to the other
 * coalesce routes. */
size_t tgp_connect_coalesce_fused_workspace_bytes(int64_t num_edges, int64_t num_nodes, int64_t num_supernodes);
int tgp_connect_coalesce_fused_count(const int64_t* row, const int64_t* col, const int32_t* csr_ptr /* NULL ok */,
                                     const int32_t* csr_col /* NULL ok */, const float* edge_weight /* NULL ok */,
                                     int64_t num_edges, const int64_t* cluster_index, int64_t num_nodes,
                                     int64_t num_supernodes, const int32_t* assign_row_ptr,
                                     const int32_t* assign_perm, int reduce_op, int flags, float eps,
                                     float* out_weight_cap /* [num_edges] when weighted */, void* ws, size_t ws_bytes,
                                     int64_t* d_count, void* stream);
int tgp_connect_coalesce_fused_fill(const void* ws, int64_t num_edges, int64_t num_nodes, int64_t num_supernodes,
                                    int64_t num_out, int64_t* out_row, int64_t* out_col, void* stream);

/* A1 + A2 + (A5 | A4) + A6 filters for a BATCH OF SMALL GRAPHS in ONE launch (r4): sparse Reduce
 * (reduce/base_reduce.py:14-53,141-155) and sparse Connect (connect/base_conn.py:79-89 + the self-loop / |w| > eps
 * filters of utils/ops.py:370-380) of a PyG-style batch whose `batch` vector is sorted and whose graphs have at most
 * tgp_sparse_pool_small_max_graph_nodes() nodes: one wave per graph.  `graph_ptr` [B+1] = node offsets of the graphs.
 *   mode 0 (kept-node selection: TopK, NDP-shaped S): node_index ascending, one supernode per kept node, numbered
 *          graph-major; induced subgraph, endpoints relabelled to their position in node_index, input edge order kept.
 *   mode 1 (every node in exactly one cluster: Graclus, ...): node_index = 0..N-1, cluster ids contiguous per graph and
 *          ascending over the graphs; endpoints mapped through cluster_index, duplicates merged with `reduce_op` in input
 *          order, output in (row, col) order (PyG coalesce).
 * Outputs: x_pool [K,F], batch_pool [K] (NULL ok), and the surviving edges written ONCE at their final offsets of the
 * capacity-E buffers out_row / out_col / out_weight (NULL iff edge_weight is NULL): the first `total` entries are the
 * result, identical to tgp_reduce_sparse_f32 + tgp_reduce_batch_i64 + tgp_connect_{subgraph,coalesce_*}_{count,fill}.
 * `status` (>= tgp_sparse_pool_small_status_words(B, mode) 64-bit words of device memory, caller-owned, kept between
 * calls on ONE stream, never cleared: every word carries `epoch` in its bits 34.., 0 < epoch < 2^29, a different value
 * for every call on that buffer) holds the look-back state.  `*result` receives ONE word when the last graph is done:
 * bits 34.. = epoch, bit 31 = refused (a precondition above does not hold, checked on the device -- an edge leaving its
 * graph, unsorted rows, a graph too large, more than 512 edges in a mode-1 graph, ...: outputs are then unspecified
 * and the caller takes the staged entry points), bits 0..30 = total.  It is stored with system scope, so `result` may
 * point into pinned host memory that the caller polls for the call's epoch (the call's one host wait: no copy kernel,
 * no stream synchronise); with a device pointer the caller copies the word back after the launch.
 * `edge_ptr` / `assign_ptr` (NULL ok; [B+1] int64): the first edge / first assignment (mode 0) of every graph, when the
 * caller has them -- tgp_graph_lower_bounds_i64 of graph_ptr in `row` (worth keeping per edge list and batch vector) and
 * TopkSelect's keep-count prefix.  The kernel then skips its own searches (three dependent rounds + a boundary pass: 5.8
 * of 13.4 us on 2048 PROTEINS-shaped graphs); nothing is trusted -- the ranges must tile the arrays, every edge and kept
 * node is range-checked as without them, a wrong table is a refusal. */
int tgp_sparse_pool_small_max_graph_nodes(void);
int64_t tgp_sparse_pool_small_status_words(int64_t num_graphs, int mode);
/* out[g] = first position of the ascending `values` [n] that is >= graph_ptr[g], g = 0 .. num_graphs. */
int tgp_graph_lower_bounds_i64(const int64_t* values, int64_t n, const int64_t* graph_ptr, int64_t num_graphs,
                               int64_t* out /* [num_graphs + 1] */, void* stream);
int tgp_sparse_pool_small_f32(const float* x, int64_t num_nodes, int64_t num_features, int64_t x_row_stride,
                              const int64_t* graph_ptr /* [B+1] */, int64_t num_graphs,
                              const int64_t* edge_ptr /* NULL ok */, const int64_t* assign_ptr /* NULL ok */,
                              int64_t* edge_ptr_out /* NULL ok; [B+1]: r5 -- a call WITHOUT edge_ptr leaves the edge
                                                       ranges it searched for here (valid when the call is not refused):
                                                       a new edge list costs no tgp_graph_lower_bounds_i64 launch, the
                                                       same list pooled again gets them back as edge_ptr */,
                              const int64_t* row,
                              const int64_t* col, const float* edge_weight /* NULL ok */, int64_t num_edges,
                              const int64_t* node_index, const int64_t* cluster_index,
                              const float* weight /* NULL = ones */, int64_t nnz, int64_t num_supernodes, int mode,
                              int reduce_op, int flags, float eps, float* x_pool, int64_t* batch_pool /* NULL ok */,
                              int64_t* out_row, int64_t* out_col, float* out_weight, uint64_t* status,
                              int64_t status_words, uint64_t* result /* device-ACCESSIBLE: may be pinned host memory */,
                              uint32_t epoch, void* stream);

/* A4 for edge lists in ANY order, two-level: a stable radix sort by supernode row only (log2 K bits instead of the
 * 2 log2 K bits of the (row, col) key above) carrying (cluster column, weight) as payload, then the same in-row
 * sort / merge as the row-sorted path.  Same output as the other two pairs.  *d_count = -1 if a supernode row
 * exceeds the in-LDS sort (1024 raw entries): use tgp_connect_coalesce_{count,fill} then.  The fill half is
 * tgp_connect_coalesce_rows_fill with this workspace. */
size_t tgp_connect_coalesce_grouped_workspace_bytes(int64_t num_edges, int64_t num_nodes, int64_t num_supernodes);
int tgp_connect_coalesce_grouped_count(const int64_t* row, const int64_t* col, const float* edge_weight /* NULL ok */,
                                       int64_t num_edges, const int64_t* cluster_index, int64_t num_nodes,
                                       int64_t num_supernodes, int reduce_op, int flags, float eps, void* ws,
                                       size_t ws_bytes, int64_t* d_count, void* stream);

/* A6 (rest)  degree / per-graph max normalisation of a pooled edge list, in place
 * (utils/ops.py:383-417).  edge_weight must be initialised (ones when the list was
 * unweighted, ops.py:384-385).  ws: tgp_postprocess_sparse_workspace_bytes().           */
size_t tgp_postprocess_sparse_workspace_bytes(int64_t num_edges, int64_t num_nodes, int64_t num_graphs);
int tgp_postprocess_sparse_norm_f32(const int64_t* row, const int64_t* col, float* edge_weight,
                                    int64_t num_edges, int64_t num_nodes, int flags, float eps,
                                    const int64_t* batch_pooled /* needed for EDGE_WEIGHT_NORM */,
                                    int64_t num_graphs, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * A3 + A7 + A8  dense pooling of a padded batch
 *   x_pool  [B,K,F] = S^T X                  (reduce/base_reduce.py:158-161)
 *   adj_raw [B,K,K] = S^T A S                (connect/dense_conn.py:111-122)
 *   adj_pool[B,K,K] = postprocess(adj_raw)   (utils/ops.py:282-335)
 * S [B,N,K], X [B,N,F] contiguous fp32; A [B,N,N] contiguous, or its transposed view when
 * TGP_ADJ_TRANSPOSED is set (then element (b,i,j) is read from A[b*N*N + j*N + i]).
 * x / x_pool, adj_raw, adj_pool may each be NULL to skip that product.  MinCut
 * (poolers/mincut.py:226-237) asks for adj_raw AND adj_pool; DiffPool only adj_pool.
 * graph_sizes (optional): graph b's real nodes are its first graph_sizes[b] rows and everything beyond them in
 * S, A, X is zero (the layout of to_dense_batch / to_dense_adj, src.py:434-450); kernels then stop at the real
 * size instead of the padded one.  NULL = every row may be non-zero.
 * ---------------------------------------------------------------------------------- */
size_t tgp_dense_pool_workspace_bytes(int64_t B, int64_t N, int64_t K, int64_t F);
int tgp_dense_pool_f32(const float* S, const float* A, const float* X, int64_t B, int64_t N, int64_t K,
                       int64_t F, int flags, float eps, const int64_t* graph_sizes /* [B] or NULL */, float* x_pool,
                       float* adj_raw, float* adj_pool, void* ws, size_t ws_bytes, void* stream);

/* The same call for batches the one-wave-per-graph kernel takes (tgp_dense_pool_is_small(B,N,K,F) != 0: N <= 64,
 * K, F <= 32, B >= 64), which can also leave the per-graph tails of MinCut's two auxiliary losses in
 * mincut_terms [2,B] while S, A and the raw S^T A S sit in registers / LDS (poolers/mincut.py:226-237 computes them
 * between Reduce and Connect): [0,b] = -trace(S^T A S) / (trace(S^T D S) + loss_eps) (utils/losses.py:39-56),
 * [1,b] = || S^T S / ||S^T S||_F - I / sqrt(K) ||_F (utils/losses.py:59-70). */
int tgp_dense_pool_is_small(int64_t B, int64_t N, int64_t K, int64_t F);
int tgp_dense_pool_mincut_f32(const float* S, const float* A, const float* X, int64_t B, int64_t N, int64_t K,
                              int64_t F, int flags, float eps, float loss_eps, float* x_pool, float* adj_raw,
                              float* adj_pool, float* mincut_terms, void* ws, size_t ws_bytes, void* stream);

/* The same call with the selector folded in (A13 + A3 + A7 + A8 in ONE launch, same batches): S = softmax(X W^T + b) * mask
 * (select/mlp_select.py:105-147 for a single Linear; W [K,F], bias [K] or NULL, mask [B,N] bytes or NULL) is formed by the
 * graph's wave from the X it has just loaded, written once to S_out [B,N,K] (SelectOutput.s) and used from registers. */
int tgp_dense_pool_select_f32(const float* X, const float* A, const float* W, const float* bias,
                              const unsigned char* mask, int64_t B, int64_t N, int64_t K, int64_t F, int flags,
                              float eps, float loss_eps, float* S_out, float* x_pool, float* adj_raw, float* adj_pool,
                              float* mincut_terms /* [2,B] or NULL */,
                              int64_t* batch_pool /* [B*K] or NULL: arange(B).repeat_interleave(K), utils/ops.py:152-169 */,
                              void* stream);

/* r5: the same call on the batch as a PyG loader hands it over -- x [Ntot,F] un-padded, a ROW-SORTED edge list
 * (row, col, w or NULL = ones; E entries), batch [Ntot] sorted, node_ptr / edge_ptr [B+1] = first node / first edge of
 * every graph.  The graph's wave builds its adjacency tile in LDS from its edges (duplicates summed, entries beyond N
 * dropped, a column of another graph placed by its local id there: to_dense_adj, src.py:434-443; adj_transpose != 0:
 * the transposed scatter of src.py:442-443), reads its own rows of x, and writes S_out [B,N,K] (zero rows behind the
 * graph's nodes) and mask_out [B,N] (to_dense_batch's mask): neither to_dense_batch nor to_dense_adj runs and no [B,N,N]
 * tensor exists -- three launches -> one in front of a dense pooler on sparse inputs.  N = the longest graph (<= 64).
 * The caller guarantees the row order and the ranges (tgp_graph_lower_bounds_i64, or tgp_edge_facts_sorted_i64 below,
 * whose flag word may be read AFTER this launch: what is read through edge_ptr is clamped to [0, E], columns outside
 * [0, Ntot) are dropped).  x_dense_out / adj_dense_out (optional): the zero-padded x and the adjacency tiles as the kernel
 * holds them, for a training step -- the backward kernels read the padded tensors; what to_dense_batch / to_dense_adj
 * would have written, without their launches and without the zero fill. */
int tgp_dense_pool_select_sparse_f32(const float* x, int64_t Ntot, const int64_t* row, const int64_t* col, const float* w,
                                     int64_t E, const int64_t* batch, const int64_t* node_ptr, const int64_t* edge_ptr,
                                     const float* W, const float* bias, int64_t B, int64_t N, int64_t K, int64_t F,
                                     int flags, int adj_transpose, float eps, float loss_eps, float* S_out,
                                     unsigned char* mask_out, float* x_pool, float* adj_raw, float* adj_pool,
                                     float* mincut_terms /* [2,B] or NULL */, int64_t* batch_pool /* [B*K] or NULL */,
                                     float* x_dense_out /* [B,N,F] or NULL */, float* adj_dense_out /* [B,N,N] or NULL */,
                                     float* diff_stats /* r6, [B,4] or NULL: (sum A^2, trace(S^T A S), |S^T S|_F^2,
                                                          sum -S log(S + loss_eps)) per graph: DiffPool's two losses
                                                          (utils/losses.py:644-658, 476-483) without the dense adjacency */,
                                     void* stream);
/* r6: the same records from the one-wave-per-graph kernel on PADDED inputs (x [B,N,F], adj [B,N,N]; what the second
 * pooling layer of a hierarchical model is handed): S [B,N,K] given (W = NULL), or the selector folded in (S = NULL: W
 * [K,F], bias, mask [B,N] bytes or NULL; S_out [B,N,K] is written).  Batches tgp_dense_pool_is_small accepts. */
int tgp_dense_pool_small_diff_f32(const float* S, const float* A, const float* X, const float* W, const float* bias,
                                  const unsigned char* mask, int64_t B, int64_t N, int64_t K, int64_t F, int flags,
                                  float eps, float loss_eps, float* S_out, float* x_pool, float* adj_raw, float* adj_pool,
                                  float* diff_stats /* [B,4] */, int64_t* batch_pool /* [B*K] or NULL */, void* stream);
/* r6: DiffPool's link-prediction and entropy losses from those records: out2[0] = sqrt(max(sum_b (a2 - 2 tr + fro), 0)) *
 * link_scale, out2[1] = (sum_b ent) * ent_scale; one launch (the residual product, its partial sum, the entropy pass and
 * the tail were four). */
int tgp_diffpool_stats_tail_f32(const float* stats /* [B,4], 16-byte aligned */, int64_t B, float link_scale,
                                float ent_scale, float* out2, void* stream);

/* Facts of a NEW edge list for that call, one launch, no host round trip in front of the consumer: edge_ptr [N+2]
 * (entries [0, B] written: first entry whose source node belongs to a graph >= g) and one flag word in pinned host
 * memory (result[2]; result[0] = tag stored last): 1 = rows not grouped by ascending source node, 2 = a source id outside
 * [0, N), 4 = more than 64 consecutive graph ids without an entry.  ticket: two zeroed uint32 words owned by the caller
 * per (device, stream), left zero (not the pair of tgp_batch_facts_sorted_i64).  E > 0, N > 0, batch sorted. */
int tgp_edge_facts_sorted_i64(const int64_t* row, int64_t E, const int64_t* batch, int64_t N, int64_t* edge_ptr,
                              uint32_t* ticket, uint64_t* result, uint64_t tag, void* stream);

/* Backward of that call for the same batches (what autograd derives operator by operator from base_reduce.py:158-161,
 * dense_conn.py:111-122, utils/ops.py:282-335 and utils/losses.py:39-70: ~80 launches in a MinCut training step), ONE
 * launch: from the upstream gradients of x_pool [B,K,F], of the post-processed adj_pool [B,K,K], of the raw S^T A S
 * [B,K,K] and of the two per-graph terms [2,B] (each may be NULL = zero) it writes gS [B,N,K] and, when gX is given,
 * gX [B,N,F].  flags as in the forward call (TGP_EDGE_WEIGHT_NORM is refused); A receives no gradient.
 * DiffPool's two batch-wide losses (utils/losses.py:644-658) ride along when diff_losses [2] (their forward values,
 * tgp_diffpool_loss_tail_f32's output) and g_link / g_ent (device scalars: the upstream gradients of
 * link = link_scale * ||A - S S^T||_F and ent = ent_scale * sum(-S log(S + ent_eps)); each may be NULL) are given.
 * r5: g_mean_cut / g_mean_ortho (device scalars, each may be NULL): upstream gradients of the MEANS over the batch of
 * the two per-graph terms -- what poolers/mincut.py:226-237 returns as cut_loss / ortho_loss -- added to g_terms[.,b]
 * as *g / B inside the kernel; grad_bcast bit 0 / bit 1: g_x_pool / g_adj_pool points at ONE value that stands for
 * every element (the gradient of `x_pool.sum()` arrives from autograd as an expanded scalar; no [B,K,F] copy is made). */
int tgp_dense_pool_small_bwd_f32(const float* S, const float* A, const float* X, int64_t B, int64_t N, int64_t K,
                                 int64_t F, int flags, float eps, float loss_eps, const float* g_x_pool,
                                 const float* g_adj_pool, const float* g_adj_raw, const float* g_terms,
                                 const float* g_mean_cut, const float* g_mean_ortho, const float* g_link,
                                 const float* g_ent, const float* diff_losses, float link_scale, float ent_scale,
                                 float ent_eps, int grad_bcast, float* gS, float* gX, void* stream);

/* Generic batched fp32 GEMM on the matrix cores, C[b] = op(A[b]) B[b] with B[b] [Kd,Nc] row-major.
 * trans_a = 0: A[b] is [M,Kd] row-major; 1: A[b] is stored [Kd,M] (C = A^T B).  Used by
 * BaseLift (lift/base_lift.py:138-247: S_inv^T X_pool) and exposed for callers' own products. */
int tgp_bmm_f32(const float* A, const float* Bm, float* C, int64_t batch, int64_t M, int64_t Nc, int64_t Kd,
                int trans_a, int64_t lda, int64_t ldb, int64_t ldc, int64_t sA, int64_t sB, int64_t sC,
                void* stream);
/* C += op(A) Bm, same arguments (r4): a second product accumulates in the GEMM epilogue into the buffer the first one
 * wrote (two-term gradients of the dense Connect under autograd) */
int tgp_bmm_accumulate_f32(const float* A, const float* Bm, float* C, int64_t batch, int64_t M, int64_t Nc, int64_t Kd,
                int trans_a, int64_t lda, int64_t ldb, int64_t ldc, int64_t sA, int64_t sB, int64_t sC,
                void* stream);

/* A3' / A7'  un-padded batch (reduce/base_reduce.py:170-182, connect/dense_conn.py:195-206):
 * C[b] = S_b^T Y_b where graph b owns node rows ptr[b]..ptr[b+1] of S [Ntot,K] and Y [Ntot,F].
 * One launch instead of the reference's Python loop over graphs. max_nodes = max_b (ptr[b+1]-ptr[b]).
 * Long graphs are split along their node range across workgroups; the partial products live in the
 * workspace and are added in a fixed order. */
size_t tgp_segment_gemm_tn_workspace_bytes(int64_t B, int64_t K, int64_t F, int64_t max_nodes);
int tgp_segment_gemm_tn_f32(const float* S, const float* Y, const int64_t* ptr, float* C, int64_t B,
                            int64_t Ntot, int64_t K, int64_t F, int64_t max_nodes, void* ws, size_t ws_bytes,
                            void* stream);

/* Row-side counterpart on the same un-padded batch: C[rows of graph b] = A[rows of graph b] Bm[b] with
 * A [Ntot,Kd], Bm [B,Kd,Nc], C [Ntot,Nc].  BaseLift on dense [N,K] assignments with a batch vector
 * (lift/base_lift.py:138-247 loops over graphs) and the backward of tgp_segment_gemm_tn_f32
 * (dS_b = Y_b G_b^T, dY_b = S_b G_b). */
int tgp_segment_gemm_nn_f32(const float* A, const float* Bm, const int64_t* ptr, float* C, int64_t B,
                            int64_t Ntot, int64_t Kd, int64_t Nc, int64_t max_nodes, void* stream);

/* ----------------------------------------------------------------------------------
 * A12  TopkSelect, ratio mode (select/topk_select.py:163-203 -> PyG topk(score, ratio, batch), then the
 * row sort of SelectOutput, select/base_select.py:58).  Graph g keeps its k[g] highest-scoring nodes; kept
 * node i becomes supernode koff[g] + (its rank in the graph's descending score order), ties by lower node id.
 *   ptr  [B+1] exclusive prefix sums of the graph sizes,  k [B],  koff [B+1] exclusive prefix sums of k.
 * Outputs (k_total = koff[B] entries, ordered by ascending node id = the reference's row-sorted COO):
 *   node_index, cluster_index (int64) and assign_perm (int32): assign_perm[c] = position of supernode c's
 *   single assignment, i.e. the inverted index the sparse Reduce consumes (tgp_assign_index_build's perm).
 * batch may be NULL when B == 1.  segments_max_nodes > 0 promises that the batch vector is sorted (graph g owns nodes
 * ptr[g] .. ptr[g+1]) and that no graph has more nodes than that: graphs are then sorted one per wave / workgroup
 * instead of by a device-wide radix sort (0 = no promise). */
int tgp_topk_plan(const int64_t* sizes, int64_t B, double ratio, int64_t* k /* [B] */, int64_t* koff /* [B+1] */,
                  void* stream); /* k[g] = ceil(fp32(ratio) * n_g) (ratio < 1) or min(ratio, n_g); koff = prefix sums */
size_t tgp_topk_select_workspace_bytes(int64_t N);
int tgp_topk_select(const float* score, const int64_t* batch, int64_t N, int64_t B, const int64_t* ptr,
                    const int64_t* k, const int64_t* koff, int64_t segments_max_nodes, void* ws, size_t ws_bytes,
                    int64_t* node_index, int64_t* cluster_index, int32_t* assign_perm,
                    float* values /* optional [k_total]: score[node_index], the weights of S */,
                    int32_t* lift_row_ptr /* optional [N+1]: CSR offsets of the node -> assignment index (the
                                             transposed index Reduce's backward and Lift walk; its perm is the identity) */,
                    uint64_t* assign_pack /* optional [k_total]: the packed one-to-one index of
                                             tgp_reduce_one_to_one_f32, {node id, score} of supernode c */,
                    uint32_t* member_bits /* optional, with rank128 and directory_written (r5): by-products for the
                                             subgraph Connect of the SAME selection -- the kept-node bitmap
                                             [4 * tgp_topk_select_directory_blocks(N)] and */,
                    uint32_t* rank128 /* the rank directory [..._directory_blocks(N)]: kept nodes with id < 128 b; hand
                                         both to tgp_connect_subgraph_single (member_bits_in / rank128_in) */,
                    int* directory_written /* HOST int: 1 when this call wrote them (the device-wide route, i.e. large
                                              graphs; the per-graph sort routes of small-graph batches do not) */,
                    void* stream);
int64_t tgp_topk_select_directory_blocks(int64_t N);

/* ----------------------------------------------------------------------------------
 * A14  GraclusSelect's matching (select/graclus_select.py:62-81 -> torch_cluster 1.6.3 graclus_cluster, absent
 * from the reference tree; its published algorithm: pair every node with its heaviest unmatched neighbour).
 * Data-parallel handshake rounds over the CSR of the edge list (row_ptr / perm = tgp_assign_index_build of the
 * source ids; perm may be NULL when the list is already sorted by source, with row_ptr from
 * tgp_rowptr_from_sorted_i64); label[i] = min(i, partner) or i.  _start gathers the CSR and resets the state,
 * _rounds runs `rounds` propose/match rounds and sets matched[r] = 1 if round r formed a pair (device memory):
 * the matching is maximal once a round matches nothing.  _start also symmetrises the list (a pair weighs the larger
 * of its two directions, entries without a reverse are ignored): pairs form by mutual proposal. */
size_t tgp_graclus_match_workspace_bytes(int64_t num_nodes, int64_t num_edges);
int tgp_graclus_match_start(const int64_t* row, const int64_t* col, const float* weight /* NULL = ones */, const int32_t* row_ptr,
                            const int32_t* perm, int64_t num_nodes, int64_t num_edges, void* ws, size_t ws_bytes,
                            int64_t* label, int init_state /* 1; 0 only in front of tgp_graclus_match_graphs, which sets
                                                              the labels itself and keeps its free flags in LDS */,
                            void* stream);
int tgp_graclus_match_rounds(const int32_t* row_ptr, int64_t num_nodes, int64_t num_edges, void* ws, int rounds,
                             unsigned int* matched, int64_t* label, void* stream);
/* The remaining rounds in two launches once at most 16384 free nodes are left (after tgp_graclus_match_rounds, same
 * workspace): one workgroup runs them over the list of those nodes (same pairs: a proposal depends on the free set
 * only).  *d_status = 1: the matching is now maximal; 0: too many free nodes, nothing was changed. */
int tgp_graclus_match_tail(const int32_t* row_ptr, int64_t num_nodes, int64_t num_edges, void* ws, int64_t* label,
                           int* d_status, void* stream);
/* All rounds of every graph in ONE launch (instead of tgp_graclus_match_rounds, after tgp_graclus_match_start) for a
 * batch whose graphs own contiguous node ranges graph_ptr[b] .. graph_ptr[b+1] of at most
 * tgp_graclus_match_max_graph_nodes() nodes: one workgroup per graph, the matching the device-wide rounds give.
 * max_graph_nodes: the caller's bound on the longest graph (64 or less: one wave per graph).
 * *d_status != 0: not applicable (1: a graph longer than the bound, 2: an entry that leaves its graph); the caller
 * then runs tgp_graclus_match_start + _rounds again. */
int tgp_graclus_match_max_graph_nodes(void);
int tgp_graclus_match_graphs(const int32_t* row_ptr, int64_t num_nodes, int64_t num_edges, void* ws,
                             const int64_t* graph_ptr, int64_t num_graphs, int64_t max_graph_nodes, int64_t* label,
                             int* d_status, void* stream);
/* r4: the whole GraclusSelect of a sorted batch of SMALL graphs in ONE launch -- matching (select/graclus_select.py:66, the
 * pairs of tgp_graclus_match_graphs), consecutive cluster ids (`:68-70`, as tgp_graclus_relabel_i64) and the supernode ->
 * members index -- from the row-sorted edge list itself (no CSR, no workspace): one wave per graph of at most
 * tgp_graclus_match_graphs_fused_max_graph_nodes() (= 64) nodes and 512 entries.  index [2, N] int64 (row 0 = 0..N-1, row
 * 1 = cluster id), assign_row_ptr [N + 1] / assign_perm [N] int32 (the first K + 1 offsets are written), ones [N] fp32,
 * label [N] int64 (NULL ok: the pair's smaller node id).  `status` / `epoch` / `*result` as for tgp_sparse_pool_small_f32:
 * *result = {epoch << 34 | K}, or {epoch << 34 | 1 << 31 | ...} when the kernel refused (a graph beyond 64 nodes or 512
 * entries, an entry that leaves its graph, rows not ascending): run tgp_graclus_match_start + _graphs / _rounds then. */
int tgp_graclus_match_graphs_fused_max_graph_nodes(void);
int64_t tgp_graclus_match_graphs_fused_status_words(int64_t num_graphs);
int tgp_graclus_match_graphs_fused(const int64_t* row, const int64_t* col, const float* edge_weight /* NULL ok */,
                                   int64_t num_nodes, int64_t num_edges, const int64_t* graph_ptr, int64_t num_graphs,
                                   const int64_t* edge_ptr /* NULL ok: tgp_graph_lower_bounds_i64 of graph_ptr in row */,
                                   int64_t* label /* NULL ok */, int64_t* index, int32_t* assign_row_ptr,
                                   int32_t* assign_perm, float* ones, uint64_t* status, int64_t status_words,
                                   uint64_t* result, uint32_t epoch, void* stream);
/* Matching labels -> consecutive cluster ids, the torch.unique(cluster, return_inverse=True) of the reference's
 * select/graclus_select.py:66-70 without a sort (representatives label[r] == r keep their order, which is the order
 * unique() gives: the label of a pair is its smaller node id).  index_out [2, N] int64: row 0 = 0..N-1, row 1 = ids;
 * *d_k = number of ids.  N <= tgp_graclus_relabel_max_nodes(). */
int64_t tgp_graclus_relabel_max_nodes(void);
size_t tgp_graclus_relabel_workspace_bytes(int64_t num_nodes);
int tgp_graclus_relabel_i64(const int64_t* label, int64_t num_nodes, void* ws, size_t ws_bytes, int64_t* index_out,
                            int64_t* d_k,
                            int32_t* assign_row_ptr /* optional [N + 1] (the first K + 1 entries are written) */,
                            int32_t* assign_perm /* optional [N]; both or neither: the supernode -> members index the
                                                    sparse Reduce and the coalesce Connect walk (what
                                                    tgp_assign_index_build derives from the ids), valid when every
                                                    label is shared by at most two nodes -- a matching */,
                            float* ones /* optional [N]: filled with 1.0f, the values of the assignment matrix */,
                            void* stream);

/* TopkSelect scoring (select/topk_select.py:176, score = (x * w).sum(-1)): out[i] = <x[i,:], w>, one pass over
 * x [N,F] (row stride ldx); and the matching weight gradient out[f] = sum_i g[i] x[i,f] (fixed-order two-level
 * sum, deterministic). */
int tgp_row_dot_f32(const float* x, int64_t N, int64_t F, int64_t ldx, const float* w, float* out, void* stream);
/* The whole ratio-mode score of select/topk_select.py:176-184 in that one pass, for callers that need no gradient:
 * out[i] = act(<x[i,:], w> / ||w||_2), act 0 = identity ("linear"), 1 = tanh. */
int tgp_topk_score_f32(const float* x, int64_t N, int64_t F, int64_t ldx, const float* w, int act, float* out,
                       void* stream);
size_t tgp_weighted_colsum_workspace_bytes(int64_t F);
int tgp_weighted_colsum_f32(const float* x, int64_t N, int64_t F, int64_t ldx, const float* g, float* out, void* ws,
                            size_t ws_bytes, void* stream);
/* Backward of TopK pooling's trained path in one pass over the KEPT rows (r5; the reference gets it from ATen autograd
 * through poolers/topk.py:150-190 = select/topk_select.py:176-184 (t = x w / ||w||, s = act(t)), the kept scores as
 * the values of S (select/base_select.py:19-71) and reduce/base_reduce.py:141-155 (x'[c_a] = s_a x[i_a])).
 * Inputs: x [N,F] (row stride ldx, 16-byte aligned rows), node / cluster / values [K] = the assignments of the
 * one-to-one S (cluster NULL: 0..K-1), g_xpool = dL/dx' [K,F] contiguous or NULL, g_values = dL/d(values of S) [K] or
 * NULL, w [F], act 0 = identity, 1 = tanh.  Outputs: gx [N,F] contiguous, written entirely (rows of nodes that were
 * not kept are zero; NULL: not wanted), gw [F] (NULL: not wanted; fixed-order sums, run-to-run identical).
 * F % 4 == 0, F <= 256 (tgp_topk_pool_bwd_fits); other widths keep the operator-by-operator backward. */
int tgp_topk_pool_bwd_fits(int64_t F);
size_t tgp_topk_pool_bwd_workspace_bytes(int64_t F);
int tgp_topk_pool_bwd_f32(const float* x, int64_t N, int64_t F, int64_t ldx, const int64_t* node,
                          const int64_t* cluster, const float* values, int64_t K, const float* g_xpool,
                          const float* g_values, const float* w, int act, float* gx, float* gw, void* ws,
                          size_t ws_bytes, void* stream);

/* ss[e] = <S[row_e,:], S[col_e,:]> for every edge: the entries of S S^T that the sparse (unbatched) losses
 * read (utils/losses.py:73-127 sparse_mincut_loss, :661-708 sparse_link_pred_loss: (S[src] * S[dst]).sum(-1)),
 * without the two [E,K] gathers.  Any edge order; S [N,K] row-major. */
int tgp_edge_dot_f32(const int64_t* row, const int64_t* col, int64_t E, const float* S, int64_t N, int64_t K,
                     float* out, void* stream);
/* Two-matrix form, out[e] = <A[ia_e,:], Bm[ib_e,:]> (both [*,K] row-major): the assignment-weight gradient of the
 * sparse Reduce, dw_i = <x[node_i], dX'[cluster_i]> (autograd of reduce/base_reduce.py:146-153). */
int tgp_pair_dot_f32(const int64_t* ia, const int64_t* ib, int64_t E, const float* A, const float* Bm, int64_t K,
                     float* out, void* stream);

/* ----------------------------------------------------------------------------------
 * N3  auxiliary losses of the dense poolers, fused.
 *
 * tgp_link_loss_f32: sq[b] = || A[b] - S[b] S[b]^T ||_F^2 (utils/losses.py:644-708 link_pred_loss
 *   materialises S S^T [B,N,N]; here its tiles exist only in the MFMA accumulators and the GEMM
 *   epilogue reduces the squared residual).  S [B,N,K], A [B,N,N], sq [B].  The caller takes
 *   sqrt(sum_b sq[b]) (and / numel when normalize_loss).
 * tgp_entropy_sum_f32: out[0] = sum over all n elements of -S log(S + 1e-8)
 *   (utils/losses.py:476-483 entropy_loss before the / num_nodes).
 * tgp_cut_terms_f32: deg[b,i] = sum_j A[b,i,j]; q[b,i] = sum_k S[b,i,k]^2; den[b] = sum_i deg q
 *   = trace(S^T D S) of utils/losses.py:39-81 mincut_loss without forming D.  deg, q [B,N] are
 *   returned because the backward pass needs them.
 * ---------------------------------------------------------------------------------- */
size_t tgp_link_loss_workspace_bytes(int64_t B, int64_t N, int64_t K);
int tgp_link_loss_f32(const float* S, const float* A, int64_t B, int64_t N, int64_t K,
                      const int64_t* graph_sizes /* [B] or NULL, as in tgp_dense_pool_f32 */, float* sq, void* ws,
                      size_t ws_bytes, void* stream);
size_t tgp_entropy_sum_workspace_bytes(int64_t n);
int tgp_entropy_sum_f32(const float* S, int64_t n, float eps, float* out, void* ws, size_t ws_bytes, void* stream);
/* Backward of that sum (utils/losses.py:476-483 under autograd): out[i] = -(log(S[i] + eps) + S[i] / (S[i] + eps)) *
 * g[0] * scale, g a device scalar (the upstream gradient), one elementwise pass. */
int tgp_entropy_bwd_f32(const float* S, int64_t n, float eps, const float* g, float scale, float* out, void* stream);
/* DiffPool's losses (poolers/diffpool.py:262-284) from the native partial results in one launch:
 * out2[0] = sqrt(sum_b sq[b]) * link_scale (sq from tgp_link_loss_f32), out2[1] = (sum of ent_partial) * ent_scale
 * (ent_partial from tgp_entropy_partials_f32: *n_partial_out block sums of -S log(S + eps) in `ws`). */
int tgp_entropy_partials_f32(const float* S, int64_t n, float eps, void* ws, size_t ws_bytes, int* n_partial_out,
                             void* stream);
int tgp_diffpool_loss_tail_f32(const float* sq, int64_t B, const float* ent_partial, int n_partial, float link_scale,
                               float ent_scale, float* out2, void* stream);
int tgp_cut_terms_f32(const float* A, const float* S, int64_t B, int64_t N, int64_t K,
                      const int64_t* graph_sizes /* [B] or NULL */, float* deg, float* q, float* den, void* stream);
/* Per-graph tails of MinCut's losses in one launch: out[0,b] = -trace(raw[b]) / (den[b] + eps) (utils/losses.py:39-56),
 * out[1,b] = || G_b / ||G_b||_F - I / sqrt(K) ||_F (utils/losses.py:59-70); raw, gram [B,K,K], den [B], out [2,B]. */
int tgp_mincut_loss_terms_f32(const float* raw, const float* den, const float* gram, int64_t B, int64_t K, float eps,
                              float* out, void* stream);
/* Backward of those two tails in one launch: from the upstream gradients g_terms [2,B] it writes g_raw [B,K,K] (the
 * gradient with respect to raw: -(g_cut / (den + eps)) on the diagonal), c1 [B] (the gradient with respect to den:
 * g_cut trace(raw) / (den + eps)^2; d den / dS = 2 D S) and W [B,K,K] (the gradient with respect to gram = S^T S;
 * dS = S (W + W^T)). */
int tgp_mincut_loss_terms_bwd_f32(const float* raw, const float* den, const float* gram, const float* g_terms, int64_t B,
                                  int64_t K, float eps, float* g_raw, float* c1, float* W, void* stream);

/* ------------------------------------------------------------------------------------
 * N1 (r6)  The dense poolers' TRAINING step for graphs beyond the one-wave / one-workgroup kernels (C2: B = 32,
 * N = 1024, K = 128, F = 64), as the pieces ONE autograd node is made of (host side: functions._PoolLargeFn).  What the
 * reference gets from ATen autograd over poolers/mincut.py:220-237 / diffpool.py:208-218 (harness:
 * examples/time_and_mem_test.py:396-401).
 *
 * The operand buffer of the backward, acat [B][N][3K+F+4], has the column blocks  [U | X | 1 0 0 0 | S | V]:
 *   gS = acat [RU ; RX ; 0 ; RS ; RV]  is ONE product; for a symmetric A (V = U) its first 2K+F+4 columns / rows suffice
 *   and V is never formed; dY^T [X | 1 0 0 0] is the selector's weight AND bias gradient; dY is written over the V block,
 *   which lies behind S, so gX = [S | dY] [g_x ; W] is one product as well.
 *
 * tgp_dense_pool_train_fwd_f32: tgp_dense_pool_f32's three launches (U = A S; S^T [U | X | S] split over N; slab
 *   combine + post-processing), with U written where the caller says -- row stride ldu >= K: column block 0 of acat --
 *   and, when `gram` is not NULL, G = S^T S [B,K,K] as a third right-hand side of the second product.  adj_raw is
 *   required, adj_pool optional; flags as tgp_dense_pool_f32 (TGP_ADJ_TRANSPOSED: A is stored transposed).
 * tgp_mincut_terms_fused_f32: MinCut's per-graph loss tails out [2,B] (as tgp_mincut_loss_terms_f32) with
 *   den[b] = sum_i deg[b,i] q[b,i] formed in the same launch (deg, q [B,N] from tgp_cut_terms_f32) and kept for the
 *   backward together with stats [B,4] = (trace(raw), |G|_F^2, trace(G), |G / |G| - I / sqrt(K)|_F) per graph (optional).
 *   ptr (optional, [B+1]): deg / q are those of an UN-padded batch, graph b owns entries ptr[b] .. ptr[b+1] (N unused).
 *   q may be NULL: den[b] = sum_i deg[b,i] (the caller's deg already carries the factor, e.g. (A q)_i).
 *   means (optional, [2]): the batch means of the two terms, formed by the workgroup that arrives last (terms added in
 *   graph order: independent of the arrival order); needs `ticket`, one zeroed uint32 word per (device, stream) that the
 *   call leaves zero.
 * tgp_dense_pool_train_rhs_f32: the right-hand sides into rcat [B][3K+F+4][K], rows [RU ; RX ; four zero rows ; RS ; RV]:
 *   with gR = g_raw_a + g_raw_b (either may be NULL) + the loss' diagonal term, RV = gR, RU = gR^T (`symmetric` bit 0:
 *   RU = gR + gR^T, RV not written; bit 1: RU = gR, RV = gR^T -- the buffer's first block holds A S while raw = S^T A^T S), RX = g_x^T (g_x [B,K,F], or one value when gx_bcast), RS by mode:
 *   0: zeros;
 *   1 (MinCut): RS = W + W^T, gR -= (g_cut / (den + eps)) I, c1[b] = g_cut trace(raw) / (den + eps)^2 (the per-graph
 *      scalars from the forward's `stats`: the launch is purely elementwise, 32 x 32 tiles), with
 *      g_cut = *g_la * scale, g_ortho = *g_lb * scale (0-dim device values or NULL; scale = 1 / B for the batch means);
 *   2 (DiffPool): c = link_scale^2 *g_la / *link_loss (0 when the loss is 0), gR -= c I, RS = 2 c G.
 *   gw (optional, [B][2K][F], needs the selector weight W [K][F]): gw[b] = [g_x[b] ; W].
 * tgp_softmax_bwd_ex_f32: tgp_softmax_bwd_f32 on dS + extra + 2 c1[m / rows_per_graph] deg[m] S
 *   - *ent_g ent_scale (log(S + eps) + S / (S + eps))   (extra, c1/deg, ent_g: each optional); dy with row stride ld_dy;
 *   batch (optional, [M]): the graph of row m is batch[m] (un-padded batch) instead of m / rows_per_graph.
 * tgp_copy_cols2_f32: dst[r, col_a:col_a+wa] = a[r,:], dst[r, col_b:col_b+wb] = b[r,:] (row stride ld) in one pass;
 *   one_col >= 0: also dst[r, one_col:one_col+4] = [1 0 0 0].
 * tgp_adj_symmetry_f32: is the [B,Nmax,Nmax] adjacency the edge list (row, col) was scattered into symmetric?  One
 *   thread per entry compares adj[b,r,c] with adj[b,c,r]; the verdict arrives in pinned host words (protocol of
 *   tgp_edge_facts_sorted_i64: result word 0 = tag stored last, word 2 = 1 when some entry differs from its mirror).
 * ---------------------------------------------------------------------------------- */
size_t tgp_dense_pool_train_workspace_bytes(int64_t B, int64_t N, int64_t K, int64_t F);
int tgp_dense_pool_train_fwd_f32(const float* S, const float* A, const float* X, int64_t B, int64_t N, int64_t K,
                                 int64_t F, int flags, float eps, float* U, int64_t ldu, float* x_pool, float* adj_raw,
                                 float* adj_pool, float* gram, void* ws, size_t ws_bytes, void* stream);
/* (edge_row_ptr / edge_col / edge_w, optional, un-padded batches only: den = sum over a graph's entries of
 *  w_e q[col_e] = sum_j indeg_j q_j, the in-degree form of the batched poolers' S^T A^T S -- deg is then not read) */
int tgp_mincut_terms_fused_f32(const float* raw, const float* gram, const float* deg, const float* q, int64_t B,
                               int64_t N, int64_t K, float eps, float* den, float* out, float* stats, const int64_t* ptr,
                               uint32_t* ticket, float* means, const int32_t* edge_row_ptr, const int64_t* edge_col,
                               const float* edge_w, void* stream);
int tgp_dense_pool_train_rhs_f32(const float* g_raw_a, const float* g_raw_b, int mode, const float* stats, const float* den,
                                 const float* gram, const float* g_la, const float* g_lb, float scale,
                                 const float* link_loss, float link_scale, float eps, const float* g_x, int gx_bcast,
                                 int symmetric, const float* W, int64_t B, int64_t K, int64_t F, float* rcat, float* c1,
                                 float* gw, void* stream);
int tgp_softmax_bwd_ex_f32(const float* s, const float* ds, const float* extra, const float* c1, const float* deg,
                           int64_t rows_per_graph, const float* ent_g, float ent_scale, float ent_eps, float* dy,
                           int64_t ld_dy, int64_t M, int64_t K, const int64_t* batch, void* stream);
/* ------------------------------------------------------------------------------------
 * N3 (r6)  The UNBATCHED dense poolers' forward (mincut_u / diff_u: S [Ntot,K], sparse A) from the products their
 * Connect forms anyway -- no per-edge dot products, no index_add scatters (utils/losses.py:73-127, 204-240, 661-708;
 * connect/dense_conn.py:140-208; reduce/base_reduce.py:170-182):
 *   sum_{e in g} w_e <S_row, S_col> = trace(S_g^T (A S)_g) = trace(raw_g)       (sparse_mincut_loss' numerator)
 *   |A - S S^T|_F^2 = sum_e w_e^2 - 2 sum_g trace(raw_g) + sum_g |S_g^T S_g|_F^2  (sparse_link_pred_loss)
 * tgp_segment_gemm_tn3_f32: C_j[b] = S_b^T Y_j,b for up to three right-hand sides Y_j [Ntot,F_j] in one grid (+ one
 *   combine launch): S^T [A S | X | S] = raw pooled adjacency, pooled features, per-graph Gram matrices.  transpose0:
 *   the first output (K x K) is written transposed by the combine launch (S^T A^T S from the slabs of S^T (A S)).
 * tgp_edge_row_stats_f32: deg[i] = sum of w over CSR row i (entry count when w is NULL), q[i] = |S_i|^2.
 * tgp_diffpool_unbatched_tail_f32: out2 = (sqrt(max(sw2 - 2 sum_b trace(raw_b) + sum_b |gram_b|^2, 0)) link_scale,
 *   (sum of ent_partial) ent_scale); sw2 = sum_e w_e^2 from *sw2_dev when not NULL, else sw2_host; stats [B,2] scratch.
 * ---------------------------------------------------------------------------------- */
size_t tgp_segment_gemm_tn3_workspace_bytes(int64_t B, int64_t K, int64_t F0, int64_t F1, int64_t F2, int64_t max_nodes);
int tgp_segment_gemm_tn3_f32(const float* S, const float* Y0, int64_t F0, const float* Y1, int64_t F1, const float* Y2,
                             int64_t F2, const int64_t* ptr, float* C0, float* C1, float* C2, int64_t B, int64_t Ntot,
                             int64_t K, int64_t max_nodes, int transpose0, void* ws, size_t ws_bytes, void* stream);
/* r6: the same products (Y0 must be [Ntot,K]: the first output is raw = S^T Y0 [B,K,K]) and raw's post-processing
 * (utils/ops.py:282-335; flags = TGP_REMOVE_SELF_LOOPS | TGP_DEGREE_NORM | ..., eps its degree clamp) -> adj_pool [B,K,K]
 * in one call: for 64 < K <= 176 the post-processing launch sums the slabs of all three products itself (two launches in
 * all); otherwise the combine launch and the post-processing kernels of that size follow the product. */
size_t tgp_segment_gemm_tn3_post_workspace_bytes(int64_t B, int64_t K, int64_t F1, int64_t F2, int64_t max_nodes);
int tgp_segment_gemm_tn3_post_f32(const float* S, const float* Y0, const float* Y1, int64_t F1, const float* Y2, int64_t F2,
                                  const int64_t* ptr, float* raw, float* C1, float* C2, float* adj_pool, int64_t B,
                                  int64_t Ntot, int64_t K, int64_t max_nodes, int transpose0, int flags, float eps,
                                  void* ws, size_t ws_bytes, void* stream);
/* r6 (late): the forward of the un-padded rows route as ONE call (host time: four wrapper calls -> one).  Strings together
 * tgp_mlp_select_f32 (W != NULL: S [Ntot,K] is written; W == NULL: S is read), tgp_spmm_csr_{,stats_,entropy_}f32 (T = A S
 * [Ntot,K]; rowstat = [2,Ntot] deg | q for mode 1, [Ntot] entropy shares for mode 2), tgp_segment_gemm_tn3_post_f32 (raw,
 * x_pool [B,K,F], gram (modes 1, 2), adj_pool) and the loss tail: mode 1 tgp_mincut_terms_fused_f32 (den [B], terms [2,B],
 * stats [B,4], means [2] + ticket or NULL; transposed: the in-degree denominator from the entries), mode 2
 * tgp_diffpool_unbatched_tail_f32 (dstats [B,2], out2 [2]; sw2 as for that entry).  Same launches, same results. */
size_t tgp_pool_rows_fwd_workspace_bytes(int64_t B, int64_t K, int64_t F, int64_t max_nodes, int64_t Ntot);
int tgp_pool_rows_fwd_f32(const float* x, int64_t Ntot, int64_t F, const float* W, const float* bias, float* S,
                          const int32_t* row_ptr, const int64_t* col, const float* w, int64_t nnz, const int64_t* ptr,
                          int64_t B, int64_t K, int64_t max_nodes, int transposed, int post_flags, float eps,
                          float loss_eps, int mode, const float* sw2_dev, float sw2_host, float link_scale,
                          float ent_scale, float* T, float* raw, float* x_pool, float* gram, float* adj_pool,
                          float* rowstat, float* den, float* terms, float* stats, float* means, uint32_t* ticket,
                          float* dstats, float* out2, void* ws, size_t ws_bytes, void* stream);
/* ... and its backward for the common case -- selector folded in (W [K,F]), A = A^T (T serves as U and V) -- as ONE call:
 * tgp_postprocess_dense_bwd_f32 (g_adj, an expanded scalar when g_adj_bcast), tgp_dense_pool_train_rhs_f32,
 * tgp_copy_cols3_f32, tgp_segment_gemm_nn_ld_f32 (gS, gX), tgp_softmax_bwd_ex_f32, tgp_segment_gemm_tn_ld_f32 over
 * `slabs` row ranges slab_ptr [slabs+1] and tgp_slab_sum_split_f32.  Saved by the forward: S, T, raw, gram, stats, den, deg,
 * link_loss ([1], mode 2).  Upstream gradients (each may be NULL): g_adj, g_raw [B,K,K], g_x [B,K,F] (one value when
 * g_x_bcast), g_s [Ntot,K], g_la / g_lb (0-dim: the two losses).  Buffers: ga [B,K,K] (with g_adj), rcat [B,3K+F+4,K], c1
 * [B] (mode 1), gwcat [B,2K,F] (with gX), acat [Ntot,3K+F+4], gS [Ntot,K], part [slabs,K,F+4] (with gW / gbias).
 * Outputs: gX [Ntot,F], gW [K,F], gbias [K] (each optional). */
int tgp_pool_rows_bwd_f32(const float* S, const float* T, const float* X, const float* W, const float* raw,
                          const float* gram, const float* stats, const float* den, const float* deg,
                          const float* link_loss, const int64_t* ptr, const int64_t* batch, const int64_t* slab_ptr,
                          int64_t slabs, int64_t Ntot, int64_t B, int64_t K, int64_t F, int64_t max_nodes, int post_flags,
                          float eps, float loss_eps, int mode, int transposed, float inv_b, float link_scale,
                          float ent_scale, const float* g_adj, int g_adj_bcast, const float* g_raw, const float* g_x,
                          int g_x_bcast, const float* g_s, const float* g_la, const float* g_lb, float* ga, float* rcat,
                          float* c1, float* gwcat, float* acat, float* gS, float* gX, float* part, float* gW, float* gbias,
                          void* stream);
int tgp_edge_row_stats_f32(const int32_t* row_ptr, const float* w, const float* S, int64_t N, int64_t K, float* deg,
                           float* q, void* stream);
/* the two segment products with explicit row strides (operands that are column blocks of a wider buffer: the unbatched
 * training step): nn: C[rows of b] = A[rows of b, 0:Kd] Bm[b] (A row stride lda; Bm [B][Kd][Nc], row stride ldb, batch
 * stride sB; C row stride ldc);  tn: C[b] = A[rows of b]^T Y[rows of b] ([B][M][Nc], no node-range split). */
int tgp_segment_gemm_nn_ld_f32(const float* A, int64_t lda, const float* Bm, int64_t ldb, int64_t sB, const int64_t* ptr,
                               float* C, int64_t ldc, int64_t B, int64_t Ntot, int64_t Kd, int64_t Nc, int64_t max_nodes,
                               void* stream);
int tgp_segment_gemm_tn_ld_f32(const float* A, int64_t lda, const float* Y, int64_t ldy, const int64_t* ptr, float* C,
                               int64_t B, int64_t Ntot, int64_t M, int64_t Nc, void* stream);
/* is the COALESCED row-sorted list (row, col, w) with CSR offsets row_ptr symmetric: has every entry (r, c, w) a mirror
 * entry (c, r, w)?  Verdict as tgp_adj_symmetry_f32 (pinned result word 2 = 1: no). */
/* ... and for a dense [B,N,N] adjacency: every entry equals its mirror image?  (one pass, 32 x 32 tile pairs) */
int tgp_dense_symmetry_f32(const float* adj, int64_t B, int64_t N, uint32_t* ticket, uint64_t* result, uint64_t tag,
                           void* stream);
int tgp_edge_symmetry_f32(const int64_t* row, const int64_t* col, const float* w, int64_t E, const int32_t* row_ptr,
                          int64_t N, uint32_t* ticket, uint64_t* result, uint64_t tag, void* stream);
int tgp_diffpool_unbatched_tail_f32(const float* raw, const float* gram, int64_t B, int64_t K, const float* sw2_dev,
                                    float sw2_host, const float* ent_partial, int n_partial, float link_scale,
                                    float ent_scale, float* stats, float* out2, void* stream);

/* part [slabs][K][W] (W >= F + 1: dY^T [X | 1 0 0 0] per row slab) -> gw [K][F] (columns 0..F-1) and gb [K] (column F), the
 * slabs added in order; either output may be NULL. */
int tgp_slab_sum_split_f32(const float* part, int64_t slabs, int64_t K, int64_t F, int64_t W, float* gw, float* gb,
                           void* stream);
int tgp_adj_symmetry_f32(const int64_t* row, const int64_t* col, int64_t E, const int64_t* batch, const int64_t* ptr,
                         int64_t Nmax, const float* adj, uint32_t* ticket, uint64_t* result, uint64_t tag, void* stream);
int tgp_copy_cols2_f32(const float* a, int64_t wa, const float* b, int64_t wb, int64_t rows, float* dst, int64_t ld,
                       int64_t col_a, int64_t col_b, int64_t one_col, void* stream);
/* the same with a third source block c [rows, wc] -> dst[:, col_c : col_c + wc] */
int tgp_copy_cols3_f32(const float* a, int64_t wa, const float* b, int64_t wb, const float* c, int64_t wc, int64_t rows,
                       float* dst, int64_t ld, int64_t col_a, int64_t col_b, int64_t col_c, int64_t one_col, void* stream);

/* A7'  sparse A times dense S (connect/dense_conn.py:165,204: torch.sparse.mm), A in CSR built from a
 * row-sorted coalesced edge list: T[i,:] = sum_{e in row i} w[e] * S[col[e],:].  w may be NULL. */
int tgp_rowptr_from_sorted_i64(const int64_t* rows, int64_t n, int64_t num_rows, int32_t* row_ptr, void* stream);
/* the same for a list whose order has not been checked: *d_unsorted (zeroed by the entry) = 1 when a row id descends;
 * the offsets of such a list are meaningless but stay in [0, n] (callers go on and read the flag with their next
 * read-back instead of synchronising for the check first) */
int tgp_rowptr_from_sorted_flag_i64(const int64_t* rows, int64_t n, int64_t num_rows, int32_t* row_ptr, int* d_unsorted,
                                    void* stream);
int tgp_spmm_csr_f32(const int32_t* row_ptr, const int64_t* col, const float* w, int64_t num_rows, int64_t nnz,
                     const float* S, int64_t K, float* T, void* stream);
/* r6: the same product with MinCut's degree term riding along (utils/losses.py:73-127): deg[i] = sum of row i's weights
 * (entry count when w is NULL), q[i] = |S_i|^2 -- the outputs of tgp_edge_row_stats_f32 without its launch. */
int tgp_spmm_csr_stats_f32(const int32_t* row_ptr, const int64_t* col, const float* w, int64_t num_rows, int64_t nnz,
                           const float* S, int64_t K, float* T, float* deg, float* q, void* stream);
/* r6: ... and with DiffPool's entropy loss riding along (utils/losses.py:476-483): partial[0 .. *n_partial) = per-workgroup
 * shares of sum(-S log(S + eps)) over S [num_rows,K] (partial: >= num_rows floats; the sum of the shares is what
 * tgp_diffpool_unbatched_tail_f32 takes as its entropy partials).  *n_partial = -1: the shape does not take the row
 * kernel -- T is computed, the shares are not. */
int tgp_spmm_csr_entropy_f32(const int32_t* row_ptr, const int64_t* col, const float* w, int64_t num_rows, int64_t nnz,
                             const float* S, int64_t K, float* T, float eps, float* partial, int* n_partial, void* stream);

/* A8 alone: post-process a [B,K,K] pooled adjacency (src may equal dst).                */
size_t tgp_postprocess_dense_workspace_bytes(int64_t B, int64_t K);
int tgp_postprocess_dense_f32(const float* src, float* dst, int64_t B, int64_t K, int flags, float eps, void* ws,
                              size_t ws_bytes, void* stream);
/* Its backward (what autograd derives from utils/ops.py:282-335): g_raw [B,K,K] from the raw S^T A S and the upstream
 * gradient of the post-processed tensor; one launch, K <= 4096, TGP_EDGE_WEIGHT_NORM refused (not differentiated here).
 * flags bit 16 (r6): g_post is ONE value that stands for every element (the gradient of a plain sum: an expanded scalar). */
int tgp_postprocess_dense_bwd_f32(const float* raw, const float* g_post, int64_t B, int64_t K, int flags, float eps,
                                  float* g_raw, void* stream);

/* A11  DenseSRCPooling.preprocessing (src.py:374-452 -> PyG to_dense_adj / to_dense_batch): the step
 * right before the timed path.  batch must be sorted; ptr[B+1] = exclusive prefix of the graph sizes.
 * adj [B,Nmax,Nmax] / out [B,Nmax,F] / mask [B,Nmax] are zero-filled inside.  transposed != 0 writes
 * A^T (what src.py:442-443 produces as a view).  Duplicated edges are summed (float atomics).  adj_is_zeroed != 0: the
 * caller has zero-filled adj already (tgp_to_dense_batch_sorted_f32 can do it in its own launch). */
int tgp_to_dense_adj_f32(const int64_t* row, const int64_t* col, const float* edge_weight /* NULL = ones */,
                         int64_t num_edges, const int64_t* batch, const int64_t* ptr, int64_t B, int64_t Nmax,
                         int transposed, int adj_is_zeroed, float* adj, void* stream);
/* the same for multi-channel edge attributes [E, C] (PyG to_dense_adj with a 2-D edge_attr): adj [B,Nmax,Nmax,C],
 * zero-filled inside, duplicates summed per channel (r5) */
int tgp_to_dense_adj_channels_f32(const int64_t* row, const int64_t* col, const float* edge_attr /* [E, C] */,
                                  int64_t num_edges, int64_t num_channels, const int64_t* batch, const int64_t* ptr,
                                  int64_t B, int64_t Nmax, int transposed, float* adj, void* stream);
/* inverse gather of tgp_to_dense_adj_f32 (its backward w.r.t. the edge weights, which the reference gets from ATen
 * autograd over PyG's scatter, src.py:434): grad_weight[e] = grad_adj[slot of e], 0 for dropped entries */
int tgp_from_dense_adj_f32(const float* grad_adj, const int64_t* row, const int64_t* col, int64_t num_edges,
                           const int64_t* batch, const int64_t* ptr, int64_t B, int64_t Nmax, int transposed,
                           float* grad_weight, void* stream);
/* inverse gather of tgp_to_dense_batch_f32 (its backward): x[i,:] = dense[batch[i], i - ptr[batch[i]], :] */
int tgp_from_dense_batch_f32(const float* dense, int64_t N, int64_t F, const int64_t* batch, const int64_t* ptr,
                             int64_t B, int64_t Nmax, float* x, void* stream);
/* Facts about a batch vector for ONE host read (what PyG's to_dense_batch / the reference's src.py:434-450 obtain with
 * batch.max().item() and a bincount): sizes [N + 1] int64 = node count per graph id (the caller keeps sizes[:B]),
 * facts int64[5] = {B - 1, bit 0: not sorted | bit 1: an id outside [0, N] (sizes are then meaningless), nodes of the
 * longest graph, number of non-empty graphs, sum over graphs of TopkSelect's k_g for topk_ratio (tgp_topk_plan's
 * arithmetic; 0 when topk_ratio <= 0)}. */
int tgp_batch_facts_i64(const int64_t* batch, int64_t N, int64_t* sizes, int64_t* facts, double topk_ratio, void* stream);
/* r5: the same facts for a SORTED vector in ONE launch, no memset, no copy back: ptr [N + 2] (CSR offsets: graph g owns
 * nodes ptr[g] .. ptr[g + 1]), sizes [N + 1], and for topk_ratio > 0 TopkSelect's plan k [N + 1] / koff [N + 2] (NULL:
 * not wanted).  `ticket`: two uint32 words owned by the caller per (device, stream), zero at entry, left zero.  `result`:
 * six words of pinned host memory -- {tag, B - 1, flags, longest graph, non-empty graphs, sum_g k_g}, word 0 stored last
 * (the caller polls it).  flags != 0 (1: not sorted, 2: an id outside [0, N], 4: more than 64 consecutive ids without a
 * node): nothing else is meaningful, take tgp_batch_facts_i64. */
int tgp_batch_facts_sorted_i64(const int64_t* batch, int64_t N, int64_t* ptr, int64_t* sizes, double topk_ratio,
                               int64_t* k, int64_t* koff, uint32_t* ticket, uint64_t* result, uint64_t tag,
                               void* stream);
/* to_dense_batch for a SORTED batch vector (graph b = nodes ptr[b] .. ptr[b+1]): output-parallel, padding and mask
 * written by the same kernel (no memsets in front).  zero_buf / zero_count (optional): a second float buffer to zero-fill
 * in the same launch (the adjacency tgp_to_dense_adj_f32 scatters into next). */
int tgp_to_dense_batch_sorted_f32(const float* x, int64_t num_nodes, int64_t F, const int64_t* ptr, int64_t B,
                                  int64_t Nmax, float* out, uint8_t* mask, float* zero_buf /* NULL ok */,
                                  int64_t zero_count, void* stream);
int tgp_to_dense_batch_f32(const float* x, int64_t num_nodes, int64_t num_features, const int64_t* batch,
                           const int64_t* ptr, int64_t B, int64_t Nmax, float* out, uint8_t* mask, void* stream);

/* A10  dense_to_block_diag (utils/ops.py:53-82): entries with |a| > 1e-8 in (b,row,col)
 * order, offset by b*K; optional valid-supernode mask [B*K] (src.py:526-552) drops and
 * renumbers supernodes (relabel[B*K] int64: new id or -1, NULL = keep all).             */
size_t tgp_block_diag_workspace_bytes(int64_t B, int64_t K);
int tgp_block_diag_count(const float* adj_pool, int64_t B, int64_t K, const int64_t* relabel /* NULL ok */,
                         int flags, float eps, void* ws, size_t ws_bytes, int64_t* d_count, void* stream);
int tgp_block_diag_fill(const float* adj_pool, int64_t B, int64_t K, const int64_t* relabel, int flags,
                        float eps, const void* ws, int64_t num_out, int64_t* out_row, int64_t* out_col,
                        float* out_weight, void* stream);

/* ------------------------------------------------------------------------------------
 * A14  NDPSelect._spectral_partition (select/ndp_select.py:187-256) for every graph of a batch, one workgroup each:
 *      largest eigenvector of Ls = I - D^-1/2 A D^-1/2 by a one-vector LOBPCG iteration (fp64, vectors and matrix in LDS), sign partition,
 *      cut = z^T L z / (2 vol), random +-1 partition (node 0 kept, node 1 dropped, rest from `seed`) when cut < 0.5.
 *      Input: CSR over all nodes (`indptr` [N+1], `col`, `w` or NULL = ones) of the SYMMETRIC adjacency without self
 *      loops (the caller symmetrises with max: to_undirected(reduce="max"), ndp_select.py:198-202); `graph_ptr` [B+1].
 *      Output: keep[v] = 1 for the positive side; info[g] = iterations used, -1 = random fallback, -2 = the graph is
 *      beyond tgp_ndp_max_graph_nodes(): left unpartitioned (keep = 0 on its nodes), the caller partitions those few
 *      graphs itself (pass max_graph_nodes = min(longest graph, tgp_ndp_max_graph_nodes()));
 *      *d_status: 0 ok, bit 1 = an entry that couples two graphs.
 * ---------------------------------------------------------------------------------- */
/* The symmetrisation in front of it (ndp_select.py:198-202: duplicates summed, self loops dropped, max with the
 * transpose) when the list already is row-major sorted, duplicate-free, loop-free and pattern-symmetric: w_out[e] =
 * max(w[e], w[reverse of e]) and *d_flag = 0; any violation (checked per entry; `indptr` = tgp_rowptr_from_sorted_i64 of
 * `row`) sets *d_flag = 1 and the caller takes the general coalesce route. */
int tgp_ndp_symmetric_max_f32(const int64_t* row, const int64_t* col, const float* w /* NULL = ones */, int64_t E,
                              int64_t num_nodes, const int32_t* indptr, float* w_out, int* d_flag, void* stream);
int tgp_ndp_max_graph_nodes(void);
int tgp_ndp_partition(const int32_t* indptr, const int64_t* col, const float* w /* NULL ok */, int64_t num_nodes,
                      int64_t nnz, const int64_t* graph_ptr, int64_t num_graphs, int64_t max_graph_nodes,
                      uint64_t seed, int max_iter, double tol, uint8_t* keep, int32_t* info, int* d_status,
                      void* stream);

/* One LARGE graph (beyond tgp_ndp_max_graph_nodes(); e.g. N = 1M, E = 10M of BASELINE configs[3]) on the whole
 * chip: the same LOBPCG iteration with every vector pass as a grid-wide kernel over fp64 work vectors in `ws`
 * (tgp_ndp_large_workspace_bytes(n): seven fp64 vectors + one fp32), scalars handed from kernel to kernel in device
 * memory.  The graph is the node range [p0, p1) of the same symmetric, self-loop-free CSR.
 *   start : degrees, start vector, first product.
 *   steps : `steps` LOBPCG steps (no-ops once |Ls x - lambda x| <= tol * lambda or max_iter steps are done);
 *           d_progress[0] = done flag, [1] = steps taken so far: the host reads it once per batch of steps.
 *   finish: sign partition, cut test, the reference's random fallback (ndp_select.py:171-185, 250-252);
 *           keep[p0 .. p1) is written, *info = steps used or -1 (random fallback).
 *   state : (diagnostics; synchronises the stream) lambda, |residual|^2, steps, random flag, cut. */
size_t tgp_ndp_large_workspace_bytes(int64_t num_graph_nodes);
int tgp_ndp_large_start(const int32_t* indptr, const int64_t* col, const float* w /* NULL ok */, int64_t p0,
                        int64_t p1, int max_iter, void* ws, size_t ws_bytes, int* d_status, void* stream);
int tgp_ndp_large_steps(const int32_t* indptr, const int64_t* col, const float* w, int64_t p0, int64_t p1, int steps,
                        double tol, void* ws, size_t ws_bytes, int32_t* d_progress, int* d_status, void* stream);
int tgp_ndp_large_finish(const int32_t* indptr, const int64_t* col, const float* w, int64_t p0, int64_t p1,
                         uint64_t seed, void* ws, size_t ws_bytes, uint8_t* keep, int32_t* info, void* stream);
int tgp_ndp_large_state(const void* ws, int64_t num_graph_nodes, double* out5, void* stream);

/* ------------------------------------------------------------------------------------
 * A9  KronConnect.forward (connect/kron_conn.py:117-165), block-batched: the batch Laplacian is block diagonal, so
 *     every graph's Kron reduction  L' = L[+,+] - L[+,-] L[-,-]^-1 L[-,+]  is independent.  The graph's dense
 *     Laplacian is formed in fp64 -- in LDS by one workgroup per graph up to 128 nodes; in a workspace slab, reduced
 *     panel by panel by many workgroups per graph, up to tgp_kron_batched_max_graph_nodes() -- the dropped nodes are
 *     eliminated, then A = -L', |A| > threshold, zero diagonal,
 *     fp32 cast; edges come out in row-major order of the pooled batch (what the reference's CSR -> COO gives).
 *     Entries: CSR over ALL nodes of the batch (`indptr` [N+1], `col` [nnz], values fp32 or fp64 or NULL = ones,
 *     optional `perm` = CSR slot -> entry).  from_adjacency = 0: the entries ARE the Laplacian (SelectOutput.L);
 *     from_adjacency = 1: they are edge weights and L = D - A is formed in the kernel (get_laplacian,
 *     kron_conn.py:88-92; self loops skipped, duplicates summed).  `graph_ptr` [B+1]: node offsets of the graphs
 *     (sorted batch vector); `node_index`: kept nodes, ascending (pooled id = position).  An exactly singular
 *     L[-,-] block is redone with the reference's 1e-6 I damping (kron_conn.py:131-135), per graph.
 *     *d_count = -1: declined (node_index not ascending, a graph beyond the size limit, or an entry that couples two
 *     graphs) - the caller keeps its generic route.  from_adjacency bit 1 (value 2, TGP_KRON_SKIP_OVERSIZE): graphs
 *     beyond the size limit do not decline the call, they are left out (no scratch, no edges) and the caller reduces
 *     those few graphs itself (pass max_graph_nodes = min(longest graph, tgp_kron_batched_max_graph_nodes())).
 * ---------------------------------------------------------------------------------- */
size_t tgp_kron_batched_workspace_bytes(int64_t num_nodes, int64_t num_graphs, int64_t max_graph_nodes,
                                        int64_t cap_dense, int64_t cap_big);
int tgp_kron_batched_max_graph_nodes(void);
/* cap_dense / cap_big: sizes (elements) of the k x k result buffer and of the fp64 scratch for graphs beyond the LDS
 * capacity.  Exact figures from the caller (sum over graphs of n_g^2; sum of n_g * (n_g | 1) over the graphs of more
 * than 128 nodes) keep the workspace at what the batch needs; -1 = the worst case from max_graph_nodes alone
 * (num_nodes * max_graph_nodes).  The real totals are checked on the device (too small: *d_count = -1).
 * num_big: how many graphs have more than 128 (and at most tgp_kron_batched_max_graph_nodes()) nodes, an upper
 * bound, -1 = unknown (num_graphs): it sizes the launches of the multi-workgroup elimination those graphs take.
 * node_rank (r6, optional, [num_nodes + 1] on the device): node_rank[i] = kept nodes in front of node i (so
 * node_rank[num_nodes] = num_kept), from a caller that compacted the kept nodes itself (tgp_mask_index_fill) and thereby
 * knows node_index to be ascending and in range: the call then skips its own flag scatter and scan (three launches). */
int tgp_kron_batched_count(const int32_t* indptr, const int64_t* col, const float* val32, const double* val64,
                           const int32_t* perm, int from_adjacency, int64_t num_nodes, int64_t nnz,
                           const int64_t* graph_ptr, int64_t num_graphs, int64_t max_graph_nodes, int64_t cap_dense,
                           int64_t cap_big, int64_t num_big, const int64_t* node_index, int64_t num_kept,
                           double threshold, const uint32_t* node_rank, void* ws, size_t ws_bytes, int64_t* d_count,
                           void* stream);
int tgp_kron_batched_fill(const void* ws, int64_t num_nodes, int64_t num_graphs, int64_t max_graph_nodes,
                          int64_t cap_dense, int64_t cap_big, int64_t num_big, const int64_t* graph_ptr,
                          int64_t num_out, int64_t* out_row, int64_t* out_col, float* out_weight,
                          const uint32_t* node_rank /* what the count call was given, or NULL */, void* stream);

/* A12, min_score mode (select/topk_select.py:186-194): prob = per-graph softmax of `score` (PyG utils.softmax: max
 * subtraction, +1e-16 in the denominator), kept = prob > min(max_g(prob) - tol, min_score) (PyG topk; tol 1e-7), kept
 * nodes in ascending order (nonzero()).  `ptr` [B+1]: node offsets of the graphs of the SORTED batch vector.  count
 * writes prob [N] and leaves the number of kept nodes in *d_count; fill writes node_index [num_out]. */
size_t tgp_topk_minscore_workspace_bytes(int64_t num_nodes, int64_t num_graphs);
int tgp_topk_minscore_count(const float* score, const int64_t* ptr, int64_t num_nodes, int64_t num_graphs,
                            float min_score, float tol, float* prob, void* ws, size_t ws_bytes, int64_t* d_count,
                            void* stream);
int tgp_topk_minscore_fill(const void* ws, const int64_t* ptr, int64_t num_nodes, int64_t num_graphs, int64_t num_out,
                           int64_t* node_index, void* stream);

/* ------------------------------------------------------------------------------------
 * A13  MLPSelect's last layer, one pass over the node features
 *   S[m,:] = softmax(X[m,:] W^T + b) * mask[m]       (select/mlp_select.py:139-145; single Linear(F,K): :67)
 * x [M,F], weight [K,F] (torch.nn.Linear layout), bias [K] or NULL, mask [M] bytes (0 = padded row) or NULL,
 * s_out [M,K]; all contiguous fp32.  K <= tgp_mlp_select_max_fused_k() (256: eight 32-column accumulator tiles per
 * wave); wider selectors run tgp_bmm_f32 for the logits and tgp_softmax_rows_f32 (bias + softmax + mask, in place).
 * tgp_softmax_bwd_f32: dY = S * (dS - <dS,S>_row), the gradient w.r.t. the logits (0 on masked rows, whose S is 0).
 * ---------------------------------------------------------------------------------- */
int tgp_mlp_select_max_fused_k(void);
int tgp_mlp_select_f32(const float* x, const float* weight, const float* bias, const unsigned char* mask, int64_t M,
                       int64_t F, int64_t K, float* s_out, void* stream);
int tgp_softmax_rows_f32(float* y, const float* bias, const unsigned char* mask, int64_t M, int64_t K, void* stream);
int tgp_softmax_bwd_f32(const float* s, const float* ds, float* dy, int64_t M, int64_t K, void* stream);
/* r5: the whole backward of that layer (K <= 32, F <= 64: tgp_mlp_select_bwd_fits) in one pass over S, dS and X: it
 * forms dY = S (dS - <dS,S>) and writes gx [M,F] = dY W (accumulate_gx != 0: adds to what gx holds -- the gradient the
 * pooling backward already left there), gw [K,F] = dY^T X and gb [K] = column sums of dY (gx / gw / gb may be NULL; gb
 * needs gw).  What autograd derives from select/mlp_select.py:139-145 as softmax backward + two matmul backwards + a bias
 * reduction + the accumulation of the two gradients of X: eight launches; here one, plus -- when gw is asked for -- a
 * second tiny one that adds the workgroups' partial gw / gb (in `ws`, tgp_mlp_select_bwd_workspace_bytes) in a fixed
 * order.  No float atomics: results are run-to-run identical. */
int tgp_mlp_select_bwd_fits(int64_t F, int64_t K);
size_t tgp_mlp_select_bwd_workspace_bytes(int64_t M, int64_t F, int64_t K);
int tgp_mlp_select_bwd_f32(const float* s, const float* ds, const float* x, const float* weight, int64_t M, int64_t F,
                           int64_t K, float* gx, int accumulate_gx, float* gw, float* gb, void* ws, size_t ws_bytes,
                           void* stream);

/* ------------------------------------------------------------------------------------
 * test hooks for the shared primitives (device-wide stable LSD radix sort, block scan)
 * ---------------------------------------------------------------------------------- */
size_t tgp_debug_sort_workspace_bytes(int64_t n);
int tgp_debug_sort_pairs_u64(const uint64_t* keys_in, const uint32_t* vals_in, int64_t n, int key_bits,
                             uint64_t* keys_out, uint32_t* vals_out, void* ws, size_t ws_bytes, void* stream);

/* Output contract of the single-call operators (r5).  tgp_sparse_pool_small_f32 and tgp_connect_subgraph_single write
 * survivors at their final offsets of CAPACITY-sized buffers; the reference hands out new tensors of exactly the pooled
 * size (connect/base_conn.py:103-112, SURVEY 8(b) "Ownership").  Once the count has arrived the caller allocates exact
 * outputs and this ONE launch moves the first n entries of the capacity arrays there (weight: 4- or 8-byte values or
 * NULL, edge_id: NULL ok), so edge_index is a contiguous [2, n] tensor and nothing pins E-sized scratch. */
int tgp_edges_compact(const int64_t* row, const int64_t* col, const void* weight, int weight_bytes,
                      const int64_t* edge_id, int64_t n, int64_t* out_row, int64_t* out_col, void* out_weight,
                      int64_t* out_edge_id, void* stream);
/* The same kernel for up to eight unrelated arrays (host arrays of `count` device pointers / byte counts, each a multiple
 * of 4): the merged outputs of a gathered step (tgp/data/collate.py:144-153) leave the receive buffer as exact-size
 * tensors in one launch. */
int tgp_copy_arrays(const void* const* src, void* const* dst, const int64_t* bytes, int count, void* stream);
/* r6: a byte mask [n] as the sorted list of its non-zero positions -- the `nonzero` with which NDPSelect turns its +-1
 * partition into the kept nodes, and the [2, k] index / value arrays of the one-node-per-supernode S built around it
 * (select/ndp_select.py:257-262, select/base_select.py:142-161).  tgp_mask_index_count: per-tile counts into `scratch`
 * (tgp_mask_index_scratch_words(n) 32-bit words, word 0 an arrival ticket that is zero between calls: clear the buffer
 * once), then {epoch << 34 | k} in the PINNED word `result` (k = -3 in 34-bit two's complement when `*declined` (device,
 * optional) is non-zero).  tgp_mask_index_fill, once the host has read k: pos_out[k] the positions in increasing order,
 * rank_out[k] = 0..k-1 and ones_out[k] = 1 (both optional); perm_out[k] = 0..k-1 (int32) and pack_out[k] = {uint32
 * position, fp32 1.0} (both optional): the one-to-one inverted index tgp_one_to_one_index_build would make of the three
 * arrays, for the Reduce of the same selection; node_rank_out[n + 1] (optional): non-zero bytes in front of every
 * position, total at [n] (what tgp_kron_batched_count takes as node_rank).  n < 2^31. */
int64_t tgp_mask_index_scratch_words(int64_t n);
int tgp_mask_index_count(const uint8_t* mask, int64_t n, const int32_t* declined, uint32_t* scratch, uint64_t* result,
                         uint32_t epoch, void* stream);
int tgp_mask_index_fill(const uint8_t* mask, int64_t n, const uint32_t* scratch, int64_t k, int64_t* pos_out,
                        int64_t* rank_out, float* ones_out, int32_t* perm_out, uint64_t* pack_out,
                        uint32_t* node_rank_out, void* stream);
/* r6: the host wait of tgp_sparse_pool_small_f32 + the launch that makes its edge_index contiguous, in one call (the
 * reference's own host reads: `.item()` in utils/ops.py:370-380 / the nonzero of connect/base_conn.py:79-89).  Spins on
 * the PINNED result word until call `epoch` has stored it; unless the kernel refused the input (bit 31 of the word) the
 * n = word & 0x7fffffff columns the kernel left in `col_scratch` are moved behind the n rows at the front of `out_rows`
 * (capacity >= 2 n: edge_index' [2, n] contiguous afterwards).  The caller reads the word from the pinned memory. */
int tgp_result_wait_pack_cols(const uint64_t* result, uint32_t epoch, const int64_t* col_scratch, int64_t* out_rows,
                              void* stream);

/* ------------------------------------------------------------------------------------
 * float64 value types of the HBM-bound operators (r4).  The reference's ATen ops compute model.double() inputs in fp64
 * (reduce/base_reduce.py:141-155, utils/ops.py:282-419); these entry points do the same for the operators that only
 * move and add values -- sparse Reduce, the edge-list Connect (subgraph / coalesce / block-diagonal export) and both
 * post-processings -- with the argument meaning of their fp32 twins above (`eps` as a double).  Same summation orders,
 * products rounded before the add.  (The dense GEMM path's fp64 forms follow in the next section, r5.)
 * ---------------------------------------------------------------------------------- */
int tgp_reduce_sparse_f64(const double* x, int64_t num_nodes, int64_t num_features, int64_t x_row_stride,
                          const int64_t* node_index, const double* weight /* NULL = ones */,
                          const int32_t* row_ptr /* NULL = one assignment per supernode */,
                          const int32_t* perm /* NULL = identity */, int64_t nnz, int64_t num_supernodes, double* x_pool,
                          void* stream);
int tgp_connect_subgraph_single_f64(const int64_t* row, const int64_t* col, const double* edge_weight /* NULL ok */,
                                    int64_t num_edges, const int64_t* node_index, int64_t k, int64_t num_nodes, int flags,
                                    double eps, void* ws /* tgp_connect_subgraph_single_workspace_bytes */,
                                    size_t ws_bytes, int64_t* out_row, int64_t* out_col, double* out_weight,
                                    int64_t* out_edge_id, const uint32_t* member_bits_in, const uint32_t* rank128_in,
                                    uint64_t* status, int64_t status_words, uint64_t* result, uint32_t epoch,
                                    void* stream);
size_t tgp_connect_coalesce_workspace_bytes_f64(int64_t num_edges, int64_t num_nodes, int64_t num_supernodes);
int tgp_connect_coalesce_count_f64(const int64_t* row, const int64_t* col, const double* edge_weight /* NULL ok */,
                                   int64_t num_edges, const int64_t* cluster_index, int64_t num_nodes,
                                   int64_t num_supernodes, int reduce_op, int flags, double eps, void* ws,
                                   size_t ws_bytes, int64_t* d_count, void* stream);
int tgp_connect_coalesce_fill_f64(const void* ws, int64_t num_edges, int64_t num_nodes, int64_t num_supernodes,
                                  int has_weight, int flags, int64_t num_out, int64_t* out_row, int64_t* out_col,
                                  double* out_weight, void* stream);
size_t tgp_postprocess_sparse_workspace_bytes_f64(int64_t num_edges, int64_t num_nodes, int64_t num_graphs);
int tgp_postprocess_sparse_norm_f64(const int64_t* row, const int64_t* col, double* edge_weight /* in place */,
                                    int64_t num_edges, int64_t num_nodes, int flags, double eps,
                                    const int64_t* batch_pooled, int64_t num_graphs, void* ws, size_t ws_bytes,
                                    void* stream);
int tgp_block_diag_count_f64(const double* adj, int64_t B, int64_t K, const int64_t* relabel, int flags, double eps,
                             void* ws /* tgp_block_diag_workspace_bytes */, size_t ws_bytes, int64_t* d_count,
                             void* stream);
int tgp_block_diag_fill_f64(const double* adj, int64_t B, int64_t K, const int64_t* relabel, int flags, double eps,
                            const void* ws, int64_t num_out, int64_t* out_row, int64_t* out_col, double* out_weight,
                            void* stream);
size_t tgp_postprocess_dense_workspace_bytes_f64(int64_t B, int64_t K);
int tgp_postprocess_dense_f64(const double* src, double* dst /* may alias src */, int64_t B, int64_t K, int flags,
                              double eps, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * float64 dense path (r5): S^T X, S^T A S and the products of their gradients on v_mfma_f64_16x16x4_f64.  The reference
 * computes model.double() inputs with torch.matmul in fp64 (reduce/base_reduce.py:158-161, connect/dense_conn.py:111-122;
 * unbatched: base_reduce.py:170-190, dense_conn.py:140-208); the host mirror routes float64 tensors here, so the dense
 * poolers pass torch.autograd.gradcheck in double.  Argument meaning of the fp32 twins above; same association
 * (U = A S, then S^T [U | X] over slices of the node range, slabs added in slice order).
 *   tgp_bmm_f64            C[b] (+)= op(A[b]) Bm[b]            (accumulate != 0: C += product)
 *   tgp_dense_pool_f64     fused A3 + A7 + A8 of a padded batch (flags: TGP_* of tgp_dense_pool_f32, incl. TGP_ADJ_TRANSPOSED)
 *   tgp_segment_gemm_*_f64 the per-graph products of an un-padded batch
 *   tgp_spmm_csr_f64       T = A S for a row-sorted coalesced edge list (w NULL = ones)
 * ---------------------------------------------------------------------------------- */
int tgp_bmm_f64(const double* A, const double* Bm, double* C, int64_t batch, int64_t M, int64_t Nc, int64_t Kd,
                int trans_a, int64_t lda, int64_t ldb, int64_t ldc, int64_t sA, int64_t sB, int64_t sC, int accumulate,
                void* stream);
size_t tgp_dense_pool_workspace_bytes_f64(int64_t B, int64_t N, int64_t K, int64_t F);
int tgp_dense_pool_f64(const double* S, const double* A /* NULL ok */, const double* X /* NULL ok */, int64_t B,
                       int64_t N, int64_t K, int64_t F, int flags, double eps, double* x_pool, double* adj_raw,
                       double* adj_pool, void* ws, size_t ws_bytes, void* stream);
size_t tgp_segment_gemm_tn_workspace_bytes_f64(int64_t B, int64_t K, int64_t F, int64_t max_nodes);
int tgp_segment_gemm_tn_f64(const double* S, const double* Y, const int64_t* ptr, double* C, int64_t B, int64_t Ntot,
                            int64_t K, int64_t F, int64_t max_nodes, void* ws, size_t ws_bytes, void* stream);
int tgp_segment_gemm_nn_f64(const double* A, const double* Bm, const int64_t* ptr, double* C, int64_t B, int64_t Ntot,
                            int64_t Kd, int64_t Nc, int64_t max_nodes, void* stream);
int tgp_spmm_csr_f64(const int32_t* row_ptr, const int64_t* col, const double* w, int64_t num_rows, int64_t nnz,
                     const double* S, int64_t K, double* T, void* stream);

/* ------------------------------------------------------------------------------------
 * pack / unpack of the variable-size all-gather of pooled sparse outputs (r4).  A rank's pooled graphs
 * (x [K,F], batch [K] int64 or NULL, edge_index rows [E] int64, edge_weight [E] or NULL) go into one byte buffer of
 * `capacity` bytes behind a 128-byte header {magic, K, E, B, F, w_words, needed_bytes, x_words}; when needed_bytes >
 * capacity only the header is written (every rank then sees how much room the largest rank needs).  r5: the VALUES
 * travel as 4-byte words -- `F` and `x_row_stride` count words per feature row, `x_words` / `w_words` the words per
 * element (1 = fp32 / int32, 2 = fp64 / int64; w_words 0 = no weights) -- so float64 features and weights cross ranks
 * bit for bit (they were narrowed to fp32 before).  The buffers of all ranks are all-gathered as they are;
 * tgp_gather_unpack_f32 reads `world` of them ([world, capacity] bytes) and writes the merged tensors, node ids of rank
 * r shifted by the supernodes, graph ids by the graphs of the ranks before it (tgp/data/collate.py:144-153).  Output
 * sizes = sums over the headers (the caller reads them once).
 * ---------------------------------------------------------------------------------- */
int64_t tgp_gather_pack_bytes(int64_t num_supernodes, int64_t num_edges, int64_t feature_words, int w_words);
int tgp_gather_pack_f32(const float* x, int64_t x_row_stride, const int64_t* batch /* NULL ok */, const int64_t* row,
                        const int64_t* col, const float* edge_weight /* NULL ok */, int64_t num_supernodes,
                        int64_t num_edges, int64_t num_graphs, int64_t feature_words, int w_words, int x_words,
                        int64_t capacity, void* out, void* stream);
int tgp_gather_unpack_f32(const void* gathered, int64_t capacity,
                          int64_t rank_stride /* bytes between two ranks' buffers (>= capacity: several steps' slots may be
                                                 gathered as one bucket) */,
                          int world, int64_t max_words /* 4-byte words of the largest payload: sizes the grid */,
                          int64_t k_cap, int64_t e_cap /* rows the outputs can hold */,
                          int64_t feature_words, int w_words, int x_words /* what THIS rank packed: every header must
                                                                             say the same */,
                          float* x_out, int64_t* batch_out /* NULL ok */, int64_t* row_out, int64_t* col_out,
                          float* weight_out /* NULL ok */,
                          uint64_t* result /* NULL, or 5 words of device-accessible (pinned host) memory: {tag, K total,
                                              E total, largest needed_bytes, status}, word 0 stored last.  status bits:
                                              1 = every header valid, 2 = every rank agrees on feature_words / w_words /
                                              x_words, 4 = the totals fit k_cap / e_cap; a payload that does not fit
                                              capacity, or a status other than 7, makes the launch a no-op apart from
                                              this report */,
                          uint64_t tag, void* stream);
/* The same for a whole BUCKET of steps in one launch each (r4): step j of at most tgp_gather_max_bucket_steps() (= 8) is
 * packed into out + j * capacity; after the single collective over the bucket ([world][n * capacity] bytes, rank_stride =
 * n * capacity or more) one launch unpacks all its steps.  ptrs / dims are HOST arrays read during the call:
 *   pack:   ptrs [n][5] = {x, batch (NULL ok), row, col, edge_weight (NULL ok)}, dims [n][7] = {x_row_stride,
 *           num_supernodes, num_edges, num_graphs, feature_words, w_words, x_words};
 *   unpack: ptrs [n][6] = {x_out, batch_out (NULL ok), row_out, col_out, weight_out (NULL ok), result (NULL ok)},
 *           dims [n][6] = {k_cap, e_cap, tag, feature_words, w_words, x_words}. */
int tgp_gather_max_bucket_steps(void);
int tgp_gather_pack_bucket_f32(const void* const* ptrs, const int64_t* dims, int num_steps, int64_t capacity, void* out,
                               void* stream);
int tgp_gather_unpack_bucket_f32(const void* gathered, int64_t capacity, int64_t rank_stride, int world,
                                 int64_t max_words, int num_steps, void* const* ptrs, const int64_t* dims, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TGP_HIP_H */
