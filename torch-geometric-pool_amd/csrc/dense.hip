// A3 + A7 + A8: dense pooling of a padded batch on the fp32 matrix cores.
//
//   U        = A  S            [B,N,K]   (2*B*N*N*K flop: the dominant product)
//   [A'|X']  = S^T [U | X]     [B,K,K+F] (split over N, partial slabs, fixed-order combine)
//   adj_pool = postprocess(A') (diag=0, D^-1/2 . D^-1/2, /max|.|) fused into the slab combine
//
// The reference computes (S^T A) S with two torch.matmul calls (connect/dense_conn.py:120-122)
// and S^T X with a third (reduce/base_reduce.py:159); fp32 in, fp32 out, rtol 1e-5.  That
// tolerance on length-N dot products rules out bf16 / split-bf16 inputs, so the products run on
// v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chains, 64 FLOP/clk/SIMD = 157 TFLOP/s peak).
// A S is associated the other way round from the reference (A(S) first): same flops, but the
// intermediate is [N,K] (1/8 .. 1/16 of A) and A is streamed from HBM exactly once in full
// 128-byte row segments.
//
// Components: gemm_mfma.h (the tiled fp32-MFMA GEMM), dense_post.h (slab combine + A8 post-processing),
// dense_graph_kernels.h (one wave / one workgroup per graph, batched small products); this file holds the launch
// plan and the C ABI entry points.
#include <stdlib.h>

#include "common.h"
#include "gemm_mfma.h"
#include "dense_post.h"
#include "dense_graph_kernels.h"

namespace tgp {

static const int kStage2Tile = getenv("TGP_STAGE2_TILE") ? atoi(getenv("TGP_STAGE2_TILE")) : 0;
static const int kStage2TileM = getenv("TGP_STAGE2_TILE_M") ? atoi(getenv("TGP_STAGE2_TILE_M")) : 0;
struct DensePlan {
  int splits, tile, tile_m;
  int k_per_split;
  size_t u_floats, aslab_floats, xslab_floats, post_floats, colsum_floats;
};

static DensePlan dense_plan(int64_t B, int64_t N, int64_t K, int64_t F, bool with_gram = false) {
  DensePlan p;
  // second product: output is only K x (K+F) per graph -> split N so the chip is filled.  Small K: 64 x 64
  // tiles (256-thread workgroups) need fewer splits for the same number of workgroups, i.e. longer k-loops
  // per workgroup and less slab traffic.
  p.tile = (kStage2Tile == 64 || kStage2Tile == 128) ? kStage2Tile : 64;
  p.tile_m = (kStage2TileM == 64 || kStage2TileM == 128) ? kStage2TileM : p.tile;
  const int64_t T = p.tile, TM = p.tile_m;
  const int64_t tiles = ((K + TM - 1) / TM) * ((with_gram ? 2 : 1) * ((K + T - 1) / T) + ((F + T - 1) / T));
  const int64_t base = B * (tiles > 0 ? tiles : 1);
  // Split count: every workgroup of this grid is resident at once (<= 4 per CU), so the launch ends with the
  // busiest CU.  Cost model: (k-steps per workgroup + ~3 steps of prologue and epilogue) x (time of a k-step with n
  // co-resident workgroups).
  // E.g. C2: 192 tiles -> 4 splits = 768 workgroups = exactly 3 per CU (3 splits = 576 leaves a quarter of
  // the CUs with 3 and the rest with 2).  Slabs cost traffic too, hence the small bias against more splits.
  const int64_t max_splits = (N + 4 * BK - 1) / (4 * BK);  // keep >= 4 k-steps per workgroup
  const int64_t cus = 256;
  int64_t splits = 1;
  double best = 1e30;
  for (int64_t sp = 1; sp <= 32 && sp <= (max_splits > 0 ? max_splits : 1); ++sp) {
    const int64_t wgs = base * sp;
    const int64_t ksteps = ((N + sp - 1) / sp + BK - 1) / BK;
    const int64_t per_cu = (wgs + cus - 1) / cus, resident = per_cu < 4 ? per_cu : 4;
    // measured with the stamps build: a k-step takes ~(0.8 + 0.4 n) us when n workgroups share a CU
    const double cost = static_cast<double>((per_cu + 3) / 4) * static_cast<double>(ksteps + 3) *
                        static_cast<double>(2 + resident) * (1.0 + 0.01 * sp);
    if (cost < best) { best = cost; splits = sp; }
  }
  static const int force_splits = getenv("TGP_STAGE2_SPLITS") ? atoi(getenv("TGP_STAGE2_SPLITS")) : 0;
  if (force_splits > 0 && force_splits <= max_splits) splits = force_splits;
  int64_t kps = (N + splits - 1) / splits;
  kps = (kps + BK - 1) / BK * BK;
  splits = (N + kps - 1) / kps;
  if (splits < 1) splits = 1;
  p.splits = static_cast<int>(splits);
  p.k_per_split = static_cast<int>(kps);
  p.u_floats = static_cast<size_t>(B) * N * K;
  p.aslab_floats = static_cast<size_t>(B) * splits * K * K;
  p.xslab_floats = static_cast<size_t>(B) * splits * K * F;
  p.post_floats = post_ws_floats(B, K);
  p.colsum_floats = static_cast<size_t>(B) * splits * ((K + 63) / 64) * K;  // post_rows_kernel's degree partials
  return p;
}

}  // namespace tgp

using namespace tgp;

extern "C" size_t tgp_dense_pool_workspace_bytes(int64_t B, int64_t N, int64_t K, int64_t F) {
  if (B <= 0 || N <= 0 || K <= 0) return 256;
  const DensePlan p = dense_plan(B, N, K, F > 0 ? F : 0);
  return align_up(p.u_floats * 4) + align_up(p.aslab_floats * 4) + align_up(p.xslab_floats * 4) +
         align_up(p.post_floats * 4) + align_up(p.colsum_floats * 4) + 256;
}

static bool dense_pool_small_ok(int64_t B, int64_t N, int64_t K, int64_t F) {
  static const int no_small = getenv("TGP_NO_SMALL_GRAPH_KERNEL") ? 1 : 0;
  return !no_small && N <= SG_N && K <= SG_K && F <= SG_K && B >= 64;  // any N, K, F: padded batches are ragged
}

extern "C" int tgp_dense_pool_is_small(int64_t B, int64_t N, int64_t K, int64_t F) {
  return dense_pool_small_ok(B, N, K, F) ? 1 : 0;
}

static int dense_pool_impl(const float* S, const float* A, const float* X, int64_t B, int64_t N, int64_t K, int64_t F,
                           int flags, float eps, const int64_t* graph_sizes, float* x_pool, float* adj_raw,
                           float* adj_pool, float* mincut_terms, float loss_eps, void* ws, size_t ws_bytes,
                           void* stream_);
static int dense_pool_tiled(const float* S, const float* A, const float* X, int64_t B, int64_t N, int64_t K, int64_t F,
                            int flags, float eps, float* x_pool, float* adj_raw, float* adj_pool, float* U, int64_t ldu,
                            float* gram, const DensePlan& p, Carver& cv, hipStream_t stream);

extern "C" int tgp_dense_pool_f32(const float* S, const float* A, const float* X, int64_t B, int64_t N,
                                  int64_t K, int64_t F, int flags, float eps, const int64_t* graph_sizes,
                                  float* x_pool,
                                  float* adj_raw, float* adj_pool, void* ws, size_t ws_bytes, void* stream_) {
  return dense_pool_impl(S, A, X, B, N, K, F, flags, eps, graph_sizes, x_pool, adj_raw, adj_pool, nullptr, 0.f, ws,
                         ws_bytes, stream_);
}

extern "C" int tgp_dense_pool_mincut_f32(const float* S, const float* A, const float* X, int64_t B, int64_t N,
                                         int64_t K, int64_t F, int flags, float eps, float loss_eps, float* x_pool,
                                         float* adj_raw, float* adj_pool, float* mincut_terms, void* ws,
                                         size_t ws_bytes, void* stream_) {
  TGP_REQUIRE(mincut_terms && A, TGP_ERR_INVALID, "tgp_dense_pool_mincut_f32: mincut_terms / A is null");
  TGP_REQUIRE(dense_pool_small_ok(B, N, K, F), TGP_ERR_INVALID,
              "tgp_dense_pool_mincut_f32: only batches the one-wave-per-graph kernel takes (tgp_dense_pool_is_small)");
  return dense_pool_impl(S, A, X, B, N, K, F, flags, eps, nullptr, x_pool, adj_raw, adj_pool, mincut_terms, loss_eps,
                         ws, ws_bytes, stream_);
}

extern "C" int tgp_dense_pool_select_f32(const float* X, const float* A, const float* W, const float* bias,
                                         const unsigned char* mask, int64_t B, int64_t N, int64_t K, int64_t F,
                                         int flags, float eps, float loss_eps, float* S_out, float* x_pool,
                                         float* adj_raw, float* adj_pool, float* mincut_terms, int64_t* batch_pool,
                                         void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(B >= 0 && N >= 0 && K >= 0 && F >= 0, TGP_ERR_INVALID, "tgp_dense_pool_select_f32: negative size");
  if (B == 0 || N == 0 || K == 0) return TGP_OK;
  TGP_REQUIRE(X && W && S_out && F > 0, TGP_ERR_INVALID, "tgp_dense_pool_select_f32: X, W and S_out are required");
  TGP_REQUIRE(dense_pool_small_ok(B, N, K, F), TGP_ERR_INVALID,
              "tgp_dense_pool_select_f32: only batches the one-wave-per-graph kernel takes (tgp_dense_pool_is_small)");
  TGP_REQUIRE(!mincut_terms || A, TGP_ERR_INVALID, "tgp_dense_pool_select_f32: mincut_terms needs A");
  const bool want_a = A && (adj_raw || adj_pool);
  SmallArgs q{nullptr, want_a ? A : nullptr, X, static_cast<int>(B), static_cast<int>(N), static_cast<int>(K),
              static_cast<int>(F), flags, eps, x_pool, want_a ? adj_raw : nullptr, want_a ? adj_pool : nullptr,
              want_a ? mincut_terms : nullptr, loss_eps, W, bias, mask, S_out,
              reinterpret_cast<long long*>(batch_pool)};
  const int grid = static_cast<int>((B + SG_WAVES - 1) / SG_WAVES);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_pool_small_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(SG_WAVES * SG_WAVE_FLOATS * sizeof(float)));
  hipLaunchKernelGGL(dense_pool_small_kernel<false>, dim3(grid), dim3(64 * SG_WAVES),
                     SG_WAVES * SG_WAVE_FLOATS * sizeof(float), stream, q);
  return check_launch("tgp_dense_pool_select_f32");
}

// The same call on the batch as a PyG loader hands it over (r5): x [Ntot,F] un-padded, a ROW-SORTED edge list with the
// per-graph ranges node_ptr / edge_ptr [B+1] -- the graph's wave builds its adjacency tile in LDS from its edges, so
// neither to_dense_batch nor to_dense_adj runs and no [B,N,N] tensor exists (src.py:434-450 + the call above: three
// launches -> one).  N = the longest graph (<= 64).  The caller has checked that the rows are sorted and the ranges are
// the lower bounds of node_ptr in the row array; `batch` [Ntot] serves columns that leave their row's graph.
extern "C" int tgp_dense_pool_select_sparse_f32(const float* x, int64_t Ntot, const int64_t* row, const int64_t* col,
                                                const float* w, int64_t E, const int64_t* batch, const int64_t* node_ptr,
                                                const int64_t* edge_ptr, const float* W, const float* bias, int64_t B,
                                                int64_t N, int64_t K, int64_t F, int flags, int adj_transpose, float eps,
                                                float loss_eps, float* S_out, unsigned char* mask_out, float* x_pool,
                                                float* adj_raw, float* adj_pool, float* mincut_terms,
                                                int64_t* batch_pool, float* x_dense_out, float* adj_dense_out,
                                                float* diff_stats, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(B >= 0 && N >= 0 && K >= 0 && F >= 0 && E >= 0 && Ntot >= 0, TGP_ERR_INVALID,
              "tgp_dense_pool_select_sparse_f32: negative size");
  TGP_REQUIRE(!diff_stats || adj_pool || adj_raw, TGP_ERR_INVALID,
              "tgp_dense_pool_select_sparse_f32: diff_stats come with the Connect product (adj_pool or adj_raw)");
  if (B == 0 || N == 0 || K == 0) return TGP_OK;
  TGP_REQUIRE(x && W && S_out && F > 0 && node_ptr && edge_ptr && batch && (E == 0 || (row && col)), TGP_ERR_INVALID,
              "tgp_dense_pool_select_sparse_f32: x, W, S_out, batch, node_ptr, edge_ptr (and the edge list) are required");
  TGP_REQUIRE(dense_pool_small_ok(B, N, K, F), TGP_ERR_INVALID,
              "tgp_dense_pool_select_sparse_f32: only batches the one-wave-per-graph kernel takes (tgp_dense_pool_is_small)");
  TGP_REQUIRE(!(flags & TGP_ADJ_TRANSPOSED), TGP_ERR_INVALID,
              "tgp_dense_pool_select_sparse_f32: TGP_ADJ_TRANSPOSED describes a dense layout (use adj_transpose)");
  SmallArgs q{nullptr, nullptr, x, static_cast<int>(B), static_cast<int>(N), static_cast<int>(K), static_cast<int>(F),
              flags, eps, x_pool, adj_raw, adj_pool, mincut_terms, loss_eps, W, bias, nullptr, S_out,
              reinterpret_cast<long long*>(batch_pool), reinterpret_cast<const long long*>(E > 0 ? row : nullptr),
              reinterpret_cast<const long long*>(col), w, reinterpret_cast<const long long*>(node_ptr),
              reinterpret_cast<const long long*>(edge_ptr), reinterpret_cast<const long long*>(batch),
              adj_transpose ? 1 : 0, mask_out, static_cast<long long>(E), static_cast<long long>(Ntot), adj_dense_out,
              x_dense_out, diff_stats};
  const int grid = static_cast<int>((B + SG_WAVES - 1) / SG_WAVES);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_pool_small_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(SG_WAVES * SG_WAVE_FLOATS * sizeof(float)));
  hipLaunchKernelGGL(dense_pool_small_kernel<true>, dim3(grid), dim3(64 * SG_WAVES),
                     SG_WAVES * SG_WAVE_FLOATS * sizeof(float), stream, q);
  return check_launch("tgp_dense_pool_select_sparse_f32");
}

// r6: the one-wave-per-graph kernel on PADDED inputs with DiffPool's per-graph records (SmallArgs.diff_stats) -- the form a
// second pooling layer of a hierarchical model meets (x [B,N,F], adj [B,N,N] dense).  S given (W == NULL) or the selector
// folded in (S == NULL: S_out [B,N,K] is written).  The losses then need tgp_diffpool_stats_tail_f32 only.
extern "C" int tgp_dense_pool_small_diff_f32(const float* S, const float* A, const float* X, const float* W,
                                             const float* bias, const unsigned char* mask, int64_t B, int64_t N,
                                             int64_t K, int64_t F, int flags, float eps, float loss_eps, float* S_out,
                                             float* x_pool, float* adj_raw, float* adj_pool, float* diff_stats,
                                             int64_t* batch_pool, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(B >= 0 && N >= 0 && K >= 0 && F >= 0, TGP_ERR_INVALID, "tgp_dense_pool_small_diff_f32: negative size");
  if (B == 0 || N == 0 || K == 0) return TGP_OK;
  TGP_REQUIRE(A && X && F > 0 && diff_stats && (adj_raw || adj_pool) && ((S && !W) || (W && S_out && !S)), TGP_ERR_INVALID,
              "tgp_dense_pool_small_diff_f32: A, X, diff_stats, a Connect output and either S or (W, S_out) are required");
  TGP_REQUIRE(dense_pool_small_ok(B, N, K, F), TGP_ERR_INVALID,
              "tgp_dense_pool_small_diff_f32: only batches the one-wave-per-graph kernel takes (tgp_dense_pool_is_small)");
  SmallArgs q{S, A, X, static_cast<int>(B), static_cast<int>(N), static_cast<int>(K), static_cast<int>(F), flags, eps,
              x_pool, adj_raw, adj_pool, nullptr, loss_eps, W, W ? bias : nullptr, W ? mask : nullptr, W ? S_out : nullptr,
              reinterpret_cast<long long*>(batch_pool)};
  q.diff_stats = diff_stats;
  const int grid = static_cast<int>((B + SG_WAVES - 1) / SG_WAVES);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_pool_small_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(SG_WAVES * SG_WAVE_FLOATS * sizeof(float)));
  hipLaunchKernelGGL(dense_pool_small_kernel<false>, dim3(grid), dim3(64 * SG_WAVES),
                     SG_WAVES * SG_WAVE_FLOATS * sizeof(float), stream, q);
  return check_launch("tgp_dense_pool_small_diff_f32");
}

extern "C" int tgp_dense_pool_small_bwd_f32(const float* S, const float* A, const float* X, int64_t B, int64_t N,
                                            int64_t K, int64_t F, int flags, float eps, float loss_eps,
                                            const float* g_x_pool, const float* g_adj_pool, const float* g_adj_raw,
                                            const float* g_terms, const float* g_mean_cut, const float* g_mean_ortho,
                                            const float* g_link, const float* g_ent, const float* diff_losses,
                                            float link_scale, float ent_scale, float ent_eps, int grad_bcast,
                                            float* gS, float* gX, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(B >= 0 && N >= 0 && K >= 0 && F >= 0, TGP_ERR_INVALID, "tgp_dense_pool_small_bwd_f32: negative size");
  if (B == 0 || N == 0 || K == 0) return TGP_OK;
  TGP_REQUIRE(S && A && gS, TGP_ERR_INVALID, "tgp_dense_pool_small_bwd_f32: S, A and gS are required");
  TGP_REQUIRE(dense_pool_small_ok(B, N, K, F), TGP_ERR_INVALID,
              "tgp_dense_pool_small_bwd_f32: only batches the one-wave-per-graph kernel takes (tgp_dense_pool_is_small)");
  TGP_REQUIRE(!(flags & TGP_EDGE_WEIGHT_NORM), TGP_ERR_INVALID,
              "tgp_dense_pool_small_bwd_f32: edge_weight_norm is not differentiated by this entry");
  TGP_REQUIRE(!gX || X || !g_x_pool, TGP_ERR_INVALID, "tgp_dense_pool_small_bwd_f32: gX needs X");
  TGP_REQUIRE(!(g_link || g_ent) || diff_losses, TGP_ERR_INVALID,
              "tgp_dense_pool_small_bwd_f32: g_link / g_ent need diff_losses");
  TGP_REQUIRE((grad_bcast & ~3) == 0, TGP_ERR_INVALID, "tgp_dense_pool_small_bwd_f32: unknown grad_bcast bits");
  SmallBwdArgs q{S, A, X, static_cast<int>(B), static_cast<int>(N), static_cast<int>(K), static_cast<int>(F), flags,
                 eps, loss_eps, X ? g_x_pool : nullptr, g_adj_pool, g_adj_raw, g_terms, g_mean_cut, g_mean_ortho,
                 grad_bcast, g_link, g_ent, (g_link || g_ent) ? diff_losses : nullptr, link_scale, ent_scale, ent_eps, gS,
                 F > 0 ? gX : nullptr};
  const int grid = static_cast<int>((B + SG_WAVES - 1) / SG_WAVES);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_pool_small_bwd_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(SG_WAVES * SG_WAVE_FLOATS * sizeof(float)));
  hipLaunchKernelGGL(dense_pool_small_bwd_kernel, dim3(grid), dim3(64 * SG_WAVES),
                     SG_WAVES * SG_WAVE_FLOATS * sizeof(float), stream, q);
  return check_launch("tgp_dense_pool_small_bwd_f32");
}

static int dense_pool_impl(const float* S, const float* A, const float* X, int64_t B, int64_t N, int64_t K, int64_t F,
                           int flags, float eps, const int64_t* graph_sizes, float* x_pool, float* adj_raw,
                           float* adj_pool, float* mincut_terms, float loss_eps, void* ws, size_t ws_bytes,
                           void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(B >= 0 && N >= 0 && K >= 0 && F >= 0, TGP_ERR_INVALID, "tgp_dense_pool_f32: negative size");
  if (B == 0 || K == 0) return TGP_OK;
  const bool want_x = X && x_pool && F > 0;
  const bool want_a = A && (adj_raw || adj_pool);
  TGP_REQUIRE(S || N == 0, TGP_ERR_INVALID, "tgp_dense_pool_f32: S is null");
  TGP_REQUIRE(N < (1ll << 31) && K <= 16000 && F < (1ll << 31) && B < 65536, TGP_ERR_RANGE,
              "tgp_dense_pool_f32: dimension too large");
  if (N == 0) {
    if (want_x) (void)hipMemsetAsync(x_pool, 0, sizeof(float) * B * K * F, stream);
    if (adj_raw) (void)hipMemsetAsync(adj_raw, 0, sizeof(float) * B * K * K, stream);
    if (adj_pool) (void)hipMemsetAsync(adj_pool, 0, sizeof(float) * B * K * K, stream);
    return check_launch("tgp_dense_pool_f32");
  }
  if (dense_pool_small_ok(B, N, K, F)) {
    SmallArgs q{S, want_a ? A : nullptr, want_x ? X : nullptr, static_cast<int>(B), static_cast<int>(N),
                static_cast<int>(K), static_cast<int>(F), flags, eps, want_x ? x_pool : nullptr,
                want_a ? adj_raw : nullptr, want_a ? adj_pool : nullptr, want_a ? mincut_terms : nullptr, loss_eps,
                nullptr, nullptr, nullptr, nullptr, nullptr};
    const int grid = static_cast<int>((B + SG_WAVES - 1) / SG_WAVES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_pool_small_kernel<false, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              static_cast<int>(SG_WAVES * SG_WAVE_FLOATS * sizeof(float)));
    hipLaunchKernelGGL((dense_pool_small_kernel<false, true>), dim3(grid), dim3(64 * SG_WAVES),
                       SG_WAVES * SG_WAVE_FLOATS * sizeof(float), stream, q);
    return check_launch("tgp_dense_pool_f32(small)");
  }
  // One workgroup per graph while S and the K x K result fit LDS (measured faster than the tiled path from 8 graphs
  // up, also with a single resident workgroup per CU; below that the tiled GEMM path splits the work finer).
  static const int no_medium = getenv("TGP_NO_MEDIUM_GRAPH_KERNEL") ? 1 : 0;
  static const int kMediumMinGraphs = getenv("TGP_MEDIUM_MIN_GRAPHS") ? atoi(getenv("TGP_MEDIUM_MIN_GRAPHS")) : 8;
  static const int kMediumMaxLds = getenv("TGP_MEDIUM_MAX_LDS") ? atoi(getenv("TGP_MEDIUM_MAX_LDS")) : 152 * 1024;
  if (!no_medium && K <= 64 && B >= kMediumMinGraphs && reinterpret_cast<uintptr_t>(S) % 4 == 0) {
    const int64_t npad = (N + 31) / 32 * 32;
    const size_t lds = K <= 32 ? medium_lds_bytes<1>(npad) : medium_lds_bytes<2>(npad);
    const bool fits32 = static_cast<int64_t>(N) * (N > F ? N : F) * 4 < (1ll << 31) - 4096;
    if (lds <= static_cast<size_t>(kMediumMaxLds) && fits32) {
      MediumArgs q{S, want_a ? A : nullptr, want_x ? X : nullptr, static_cast<int>(B), static_cast<int>(N),
                   static_cast<int>(K), static_cast<int>(F), flags, eps, want_x ? x_pool : nullptr,
                   want_a ? adj_raw : nullptr, want_a ? adj_pool : nullptr, static_cast<int>(npad), graph_sizes};
      // eight waves per graph (r5, late) when there are at least eight strips to deal out and the S tile (> 80 KB) leaves a
      // single workgroup per CU (four waves would be one wave per SIMD): N = 300 / K = 64 / F = 128 151.8 -> 139.8 us.
      // Where two four-wave workgroups fit, they balance the strips better (N = 200 / K = 50: 132 us against 193 us
      // with eight waves).  TGP_MEDIUM_WAVES = 4 / 8 forces one form
      static const int force_waves = getenv("TGP_MEDIUM_WAVES") ? atoi(getenv("TGP_MEDIUM_WAVES")) : 0;
      const int64_t strips = (want_a ? (N + 31) / 32 : 0) + (want_x ? (F + 31) / 32 : 0);
      const bool eight = force_waves ? force_waves == 8 : (strips >= 8 && lds > 80 * 1024);
      const dim3 g(static_cast<unsigned>(B));
      if (K <= 32) {
        if (eight) hipLaunchKernelGGL((dense_pool_medium_kernel<1, 4, 8>), g, dim3(512), lds, stream, q);
        else hipLaunchKernelGGL((dense_pool_medium_kernel<1>), g, dim3(256), lds, stream, q);
      } else {  // (r6: the three-waves-per-SIMD instantiation <2, 3> -- a tuning knob that spilled 12 VGPRs -- is gone)
        if (eight) hipLaunchKernelGGL((dense_pool_medium_kernel<2, 2, 8>), g, dim3(512), lds, stream, q);
        else hipLaunchKernelGGL((dense_pool_medium_kernel<2, 2>), g, dim3(256), lds, stream, q);
      }
      return check_launch("tgp_dense_pool_f32(medium)");
    }
  }
  TGP_REQUIRE(ws && ws_bytes >= tgp_dense_pool_workspace_bytes(B, N, K, F), TGP_ERR_WORKSPACE,
              "tgp_dense_pool_f32: workspace too small");
  const DensePlan p = dense_plan(B, N, K, F);
  Carver cv(ws);
  float* U = cv.take<float>(p.u_floats);
  return dense_pool_tiled(S, A, X, B, N, K, F, flags, eps, x_pool, adj_raw, adj_pool, U, K, nullptr, p, cv, stream);
}

// The tiled path: U = A S (row stride ldu: the training step keeps U inside a wider buffer), S^T [U | X (| S)] split
// over N into slabs, and the slab combine + post-processing.  `gram` (optional, [B,K,K]): S^T S rides along as a third
// right-hand side of the second product (MinCut's orthogonality loss and DiffPool's link-loss gradient need it).
static int dense_pool_tiled(const float* S, const float* A, const float* X, int64_t B, int64_t N, int64_t K, int64_t F,
                            int flags, float eps, float* x_pool, float* adj_raw, float* adj_pool, float* U, int64_t ldu,
                            float* gram, const DensePlan& p, Carver& cv, hipStream_t stream) {
  const bool want_x = X && x_pool && F > 0;
  const bool want_a = A && (adj_raw || adj_pool);
  float* aslab = cv.take<float>(p.aslab_floats);
  float* xslab = cv.take<float>(p.xslab_floats);
  float* postws = cv.take<float>(p.post_floats);
  float* colpart = cv.take<float>(p.colsum_floats);
  float* gslab = gram ? cv.take<float>(p.aslab_floats) : nullptr;

  if (want_a) {
    // U[b] = A[b] S[b]     (M = N, Kd = N, Nc = K)
    GemmArgs g{};
    g.A = A; g.lda = N; g.sA = N * N;
    g.M = static_cast<int>(N); g.Kd = static_cast<int>(N);
    g.rhs[0] = GemmRhs{S, U, static_cast<int>(K), K, ldu, N * K, N * ldu, 0};
    g.splits = 1; g.k_per_split = static_cast<int>((N + BK - 1) / BK * BK);
    
    if (flags & TGP_ADJ_TRANSPOSED) launch_gemm<true>(g, static_cast<int>(B), stream);
    else launch_gemm<false>(g, static_cast<int>(B), stream);
  }
  if (want_a || want_x) {
    // aslab[b][s] = S[b]^T U[b], xslab[b][s] = S[b]^T X[b] over the s-th slice of N: one grid
    GemmArgs h{};
    h.A = S; h.lda = K; h.sA = N * K;
    h.M = static_cast<int>(K); h.Kd = static_cast<int>(N);
    h.splits = p.splits; h.k_per_split = p.k_per_split;
    h.force_bm = p.tile_m; h.force_bn = p.tile;
    const GemmRhs ra{U, aslab, static_cast<int>(K), ldu, K, N * ldu, static_cast<long>(p.splits) * K * K, K * K};
    const GemmRhs rx{X, xslab, static_cast<int>(F), F, F, N * F, static_cast<long>(p.splits) * K * F, K * F};
    const GemmRhs rg{S, gslab, static_cast<int>(K), K, K, N * K, static_cast<long>(p.splits) * K * K, K * K};
    int nr = 0;
    if (want_a) h.rhs[nr++] = ra;
    if (want_x) h.rhs[nr++] = rx;
    if (gram) h.rhs[nr++] = rg;
    // r5: K / 16 workgroups per graph post-process (post_rows_kernel) when the second product can leave the partial
    // column sums the degree vector needs (64 x 64 tiles, sum over dim -2, no edge_weight_norm)
    if (want_a && post_rows_ok(K, flags, aslab, adj_raw, adj_pool) && p.tile == 64 && p.tile_m == 64 &&
        p.splits <= PR_MAX_SPLITS && p.splits * ((K + 63) / 64) <= 2 * PR_MAX_SPLITS) {
      h.colsum = colpart;
      h.colsum_skip_diag = (flags & TGP_REMOVE_SELF_LOOPS) ? 1 : 0;
    }
    int tiles_m = 0;
    const bool have_colsum = launch_gemm<true>(h, static_cast<int>(B), stream, &tiles_m);
    if (have_colsum) {
      PostRowsArgs r{};
      r.slab = aslab; r.splits = p.splits; r.s_split = K * K; r.s_batch = static_cast<long>(p.splits) * K * K;
      r.colsum = (flags & TGP_DEGREE_NORM) ? colpart : nullptr; r.tiles_m = tiles_m;
      r.K = static_cast<int>(K); r.flags = flags; r.eps = eps; r.raw = adj_raw; r.dst = adj_pool;
      if (want_x) {
        r.xslab = xslab; r.xs_split = K * F; r.xs_batch = static_cast<long>(p.splits) * K * F;
        r.F = static_cast<int>(F); r.x_pool = x_pool;
      }
      unsigned grid = static_cast<unsigned>(B * ((K + PR_ROWS - 1) / PR_ROWS));
      r.main_blocks = static_cast<int>(grid);
      if (gram) {  // the same launch adds the S^T S slabs up (a second set of workgroups, no post-processing)
        r.gslab = gslab; r.gram = gram;
        grid *= 2;
      }
      hipLaunchKernelGGL(post_rows_kernel, dim3(grid), dim3(512), 0, stream, r);
      return check_launch("tgp_dense_pool_f32");
    }
  }
  if (gram) {
    const long total = K * K;
    int gx = static_cast<int>((total + 255) / 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(combine_slabs_kernel, dim3(gx, static_cast<unsigned>(B)), dim3(256), 0, stream, gslab,
                       p.splits, K * K, static_cast<long>(p.splits) * K * K, total, gram);
  }
  bool x_done = !want_x;
  if (want_a) {
    PostArgs q{};
    q.src = aslab; q.splits = p.splits; q.s_split = K * K; q.s_batch = static_cast<long>(p.splits) * K * K;
    q.ld_src = K; q.K = static_cast<int>(K); q.flags = flags; q.eps = eps; q.raw = adj_raw; q.dst = adj_pool;
    const XCombineArgs xc{xslab, p.splits, K * F, static_cast<long>(p.splits) * K * F, K * F, x_pool, 0};
    if (launch_post(q, B, postws, stream, want_x ? &xc : nullptr)) x_done = true;
  }
  if (!x_done) {
    const long total = K * F;
    int gx = static_cast<int>((total + 255) / 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(combine_slabs_kernel, dim3(gx, static_cast<unsigned>(B)), dim3(256), 0, stream, xslab,
                       p.splits, K * F, static_cast<long>(p.splits) * K * F, total, x_pool);
  }
  return check_launch("tgp_dense_pool_f32");
}

// ---- r6: forward of the dense poolers' TRAINING step for graphs beyond the one-wave / one-workgroup kernels ----------
// (C2-sized: B = 32, N = 1024, K = 128).  The same three launches as tgp_dense_pool_f32's tiled path, but U = A S is
// written where the caller says (row stride ldu >= K: column block 0 of the [B,N,3K+F] operand buffer the backward's
// single sum-of-products GEMM reads, see tgp_dense_pool_train_rhs_f32) and G = S^T S -- MinCut's orthogonality loss,
// DiffPool's link-loss gradient -- rides along as a third right-hand side of the second product.
// reference: base_reduce.py:158-161, dense_conn.py:111-122, utils/ops.py:282-335, utils/losses.py:59-70.
extern "C" size_t tgp_dense_pool_train_workspace_bytes(int64_t B, int64_t N, int64_t K, int64_t F) {
  if (B <= 0 || N <= 0 || K <= 0) return 256;
  size_t best = 0;
  for (int with_gram = 0; with_gram < 2; ++with_gram) {  // (the split count depends on the number of column tiles)
    const DensePlan p = dense_plan(B, N, K, F > 0 ? F : 0, with_gram != 0);
    const size_t bytes = 2 * align_up(p.aslab_floats * 4) + align_up(p.xslab_floats * 4) + align_up(p.post_floats * 4) +
                         align_up(p.colsum_floats * 4) + 256;
    if (bytes > best) best = bytes;
  }
  return best;
}

extern "C" int tgp_dense_pool_train_fwd_f32(const float* S, const float* A, const float* X, int64_t B, int64_t N,
                                            int64_t K, int64_t F, int flags, float eps, float* U, int64_t ldu,
                                            float* x_pool, float* adj_raw, float* adj_pool, float* gram, void* ws,
                                            size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(B > 0 && N > 0 && K > 0 && F >= 0, TGP_ERR_INVALID, "tgp_dense_pool_train_fwd_f32: empty or negative size");
  TGP_REQUIRE(S && A && U && adj_raw && (F == 0 || (X && x_pool)), TGP_ERR_INVALID,
              "tgp_dense_pool_train_fwd_f32: null pointer");
  TGP_REQUIRE(ldu >= K, TGP_ERR_INVALID, "tgp_dense_pool_train_fwd_f32: ldu < K");
  TGP_REQUIRE(N < (1ll << 31) && K <= 16000 && F < (1ll << 31) && B < 65536 && N * ldu * 4 < (1ll << 31) - 4096,
              TGP_ERR_RANGE, "tgp_dense_pool_train_fwd_f32: dimension too large");
  TGP_REQUIRE(ws && ws_bytes >= tgp_dense_pool_train_workspace_bytes(B, N, K, F), TGP_ERR_WORKSPACE,
              "tgp_dense_pool_train_fwd_f32: workspace too small");
  const DensePlan p = dense_plan(B, N, K, F, gram != nullptr);
  Carver cv(ws);
  return dense_pool_tiled(S, A, F > 0 ? X : nullptr, B, N, K, F, flags, eps, x_pool, adj_raw, adj_pool, U, ldu, gram,
                          p, cv, stream);
}

// dst[r, col_a : col_a + wa] = a[r, :], dst[r, col_b : col_b + wb] = b[r, :] for r < rows: the X and S column blocks
// of the backward's operand buffer in one pass (float4 when everything is 16-byte aligned); one_col >= 0: also
// dst[r, one_col : one_col + 4] = [1 0 0 0] (the column that turns dY^T [X | 1] into the weight AND bias gradient).
namespace tgp {
__global__ __launch_bounds__(256) void copy_cols2_kernel(const float* __restrict__ a, int wa, const float* __restrict__ b,
                                                         int wb, const float* __restrict__ c3, int wc, long rows,
                                                         float* __restrict__ dst, long ld, int col_a, int col_b, int col_c,
                                                         int one_col, int vec) {
  const int w_all = wa + wb + wc + (one_col >= 0 ? 4 : 0);
  const int per_row = vec ? w_all / 4 : w_all;
  const long total = rows * per_row;
  for (long e = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; e < total; e += static_cast<long>(gridDim.x) * 256) {
    const long r = e / per_row;
    int c = static_cast<int>(e - r * per_row) * (vec ? 4 : 1);
    const float* src = nullptr;
    float* out;
    if (c < wa) { src = a + r * wa + c; out = dst + r * ld + col_a + c; }
    else if (c < wa + wb) { c -= wa; src = b + r * wb + c; out = dst + r * ld + col_b + c; }
    else if (c < wa + wb + wc) { c -= wa + wb; src = c3 + r * wc + c; out = dst + r * ld + col_c + c; }
    else { c -= wa + wb + wc; out = dst + r * ld + one_col + c; }
    if (vec) *reinterpret_cast<float4*>(out) = src ? *reinterpret_cast<const float4*>(src) : make_float4(1.f, 0.f, 0.f, 0.f);
    else *out = src ? *src : (c == 0 ? 1.f : 0.f);
  }
}
}  // namespace tgp

extern "C" int tgp_copy_cols2_f32(const float* a, int64_t wa, const float* b, int64_t wb, int64_t rows, float* dst,
                                  int64_t ld, int64_t col_a, int64_t col_b, int64_t one_col, void* stream_) {
  return tgp_copy_cols3_f32(a, wa, b, wb, nullptr, 0, rows, dst, ld, col_a, col_b, 0, one_col, stream_);
}

extern "C" int tgp_copy_cols3_f32(const float* a, int64_t wa, const float* b, int64_t wb, const float* c, int64_t wc,
                                  int64_t rows, float* dst, int64_t ld, int64_t col_a, int64_t col_b, int64_t col_c,
                                  int64_t one_col, void* stream_) {
  TGP_REQUIRE(rows >= 0 && wa >= 0 && wb >= 0 && wc >= 0 && ld >= 0, TGP_ERR_INVALID, "tgp_copy_cols3_f32: negative size");
  if (rows == 0 || wa + wb + wc + (one_col >= 0 ? 4 : 0) == 0) return TGP_OK;
  TGP_REQUIRE(dst && (wa == 0 || a) && (wb == 0 || b) && (wc == 0 || c), TGP_ERR_INVALID, "tgp_copy_cols3_f32: null pointer");
  TGP_REQUIRE(col_a + wa <= ld && col_b + wb <= ld && col_c + wc <= ld && wa + wb + wc < (1ll << 30) &&
                  (one_col < 0 || one_col + 4 <= ld),
              TGP_ERR_INVALID, "tgp_copy_cols3_f32: column block outside the row");
  const int vec = (wa % 4 == 0 && wb % 4 == 0 && wc % 4 == 0 && ld % 4 == 0 && col_a % 4 == 0 && col_b % 4 == 0 &&
                   col_c % 4 == 0 && (one_col < 0 || one_col % 4 == 0) && reinterpret_cast<uintptr_t>(a) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(b) % 16 == 0 && reinterpret_cast<uintptr_t>(c) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(dst) % 16 == 0) ? 1 : 0;
  const int64_t total = rows * ((wa + wb + wc + (one_col >= 0 ? 4 : 0)) / (vec ? 4 : 1));
  int64_t grid = (total + 255) / 256;
  if (grid > 256 * 16) grid = 256 * 16;
  hipLaunchKernelGGL(copy_cols2_kernel, dim3(static_cast<unsigned>(grid)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     a, static_cast<int>(wa), b, static_cast<int>(wb), c, static_cast<int>(wc), static_cast<long>(rows), dst,
                     static_cast<long>(ld), static_cast<int>(col_a), static_cast<int>(col_b), static_cast<int>(col_c),
                     static_cast<int>(one_col >= 0 ? one_col : -1), vec);
  return check_launch("tgp_copy_cols3_f32");
}

// part [slabs][K][F + 4] (dY^T [X | 1 0 0 0] per row slab) -> gw [K][F] and gb [K], slabs added in order: the weight and
// bias gradient of the selector leave as two contiguous tensors (a strided view of one sum would be copied once more by
// autograd's gradient accumulation).
namespace tgp {
__global__ __launch_bounds__(256) void slab_sum_split_kernel(const float* __restrict__ part, int slabs, int K, int F, int W,
                                                             float* __restrict__ gw, float* __restrict__ gb) {
  // eight lanes per output element: lane q adds slabs q, q + 8, ... (their loads in flight together), the eight
  // partial sums are folded in a fixed order -- one thread per element walked 64 dependent loads (17.7 us at 64 x 128 x 68)
  const long total = static_cast<long>(K) * (F + 1);
  const int q = threadIdx.x & 7;
  for (long e0 = static_cast<long>(blockIdx.x) * 32; e0 < total; e0 += static_cast<long>(gridDim.x) * 32) {
    const long e = e0 + (threadIdx.x >> 3);
    const bool ok = e < total;
    const int k = ok ? static_cast<int>(e / (F + 1)) : 0, f = ok ? static_cast<int>(e - static_cast<long>(k) * (F + 1)) : 0;
    const float* src = part + static_cast<long>(k) * W + f;
    float v = 0.f;
    if (ok) {
#pragma unroll 4
      for (int sp = q; sp < slabs; sp += 8) v = __fadd_rn(v, src[static_cast<long>(sp) * K * W]);
    }
    v = __fadd_rn(v, __shfl_xor(v, 1, 64));
    v = __fadd_rn(v, __shfl_xor(v, 2, 64));
    v = __fadd_rn(v, __shfl_xor(v, 4, 64));
    if (ok && q == 0) {
      if (f < F) { if (gw) gw[static_cast<long>(k) * F + f] = v; }
      else if (gb) gb[k] = v;
    }
  }
}
}  // namespace tgp

extern "C" int tgp_slab_sum_split_f32(const float* part, int64_t slabs, int64_t K, int64_t F, int64_t W, float* gw,
                                      float* gb, void* stream_) {
  TGP_REQUIRE(slabs >= 1 && K >= 1 && F >= 0 && W >= F + 1 && slabs < (1ll << 20) && K < (1ll << 24) && W < (1ll << 24),
              TGP_ERR_INVALID, "tgp_slab_sum_split_f32: bad shape");
  TGP_REQUIRE(part && (gw || gb), TGP_ERR_INVALID, "tgp_slab_sum_split_f32: null pointer");
  const int64_t total = K * (F + 1);
  int64_t grid = (total + 31) / 32;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(slab_sum_split_kernel, dim3(static_cast<unsigned>(grid)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     part, static_cast<int>(slabs), static_cast<int>(K), static_cast<int>(F), static_cast<int>(W), gw, gb);
  return check_launch("tgp_slab_sum_split_f32");
}

extern "C" size_t tgp_postprocess_dense_workspace_bytes(int64_t B, int64_t K) {
  if (B <= 0 || K <= 0) return 256;
  return align_up(post_ws_floats(B, K) * 4) + 256;
}

// Backward of the post-processing (utils/ops.py:282-335 under autograd), one workgroup per graph, any K up to 4096:
//   R1 = R (1 - I);  c = sum_axis(R1);  d = sqrt(max(c, eps));  P_ij = (R1_ij / first) / second
//   gR_ij = G_ij / (d_i d_j) + gc_[axis index],  gc_t = -[c_t >= eps] (rowsum_t(G P) + colsum_t(G P)) / (2 d_t^2)
// and the diagonal of gR cleared when the self loops were.  Three sweeps over the graph's K x K pair (L2-resident), the
// per-index vectors in LDS.  As torch ops this was ~20 launches of a few KB each per training step.
namespace tgp {
// IN_LDS: the graph's R1 and G tiles are staged in LDS first ([K][K + 1] each: K <= 128) -- every sweep below is then
// LDS-local; from global memory (L2-resident, but ~0.6 us per dependent access) the serial column sweeps of one workgroup
// per graph took 98 us at B = 32, K = 128.
// r3 (late): the LDS form runs 1024 threads (the workgroup owns its CU anyway: 2 K^2 floats of LDS), stores the forward's
// P in place of R1 once (the two divisions per element were evaluated in both sweeps) and splits every column sweep over
// 8 row phases folded in a fixed order: 55 -> 15 us at B = 32, K = 128.
template <bool IN_LDS>
__global__ __launch_bounds__(IN_LDS ? 1024 : 256) void post_bwd_kernel(const float* __restrict__ R,
                                                                       const float* __restrict__ G, int K, int flags,
                                                                       float eps, float* __restrict__ out) {
  constexpr int T = IN_LDS ? 1024 : 256, NW = T / 64, PARTS = 8;
  extern __shared__ float sm[];
  float* s_d = sm;
  float* s_gs = sm + K;
  float* s_rowq = sm + 2 * K;
  float* s_colq = sm + 3 * K;
  float* sR = sm + 4 * K;
  float* sG = sR + (IN_LDS ? K * (K + 1) : 0);
  float* s_part = sG + (IN_LDS ? K * (K + 1) : 0);  // [PARTS][K] (IN_LDS only)
  const int ld = IN_LDS ? K + 1 : K;
  const long off = static_cast<long>(blockIdx.x) * K * K;
  const float* Rb = R + off;
  // bit 16 of flags (internal: tgp_postprocess_dense_bwd_f32 documents it): G is ONE value that stands for every element
  // (the gradient of a plain sum arrives as an expanded scalar: no [B,K,K] copy of it is made)
  const bool g_one = flags & (1 << 16);
  const float g_val = g_one ? G[0] : 0.f;
  const float* Gb = g_one ? G : G + off;
  float* ob = out + off;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool rsl = flags & TGP_REMOVE_SELF_LOOPS, dn = flags & TGP_DEGREE_NORM, cols = flags & TGP_SUM_AXIS_ROWS;
  if (!dn) {
    for (long e = tid; e < static_cast<long>(K) * K; e += T) {
      const int i = static_cast<int>(e / K), j = static_cast<int>(e - static_cast<long>(i) * K);
      ob[e] = (rsl && i == j) ? 0.f : (g_one ? g_val : Gb[e]);
    }
    return;
  }
  if constexpr (IN_LDS) {
    // r6 (late): five phases instead of nine.  A wave owns rows wave, wave + 16, ...; its lanes own columns lane and
    // lane + 64 (K <= 128).  Row sums are finished by the wave that loads the row; column sums are per-wave partials
    // over the wave's rows, folded in wave order; P = R1 / (d d) is formed on the fly inside the one sweep that needs
    // it instead of being stored first.  22 -> 18 us at B = 32, K = 128 (the kernel is a chain of barriers on B CUs).
    float* s_wpart = s_part;  // [NW][K]
    float cacc[2] = {0.f, 0.f};
    for (int i = wave; i < K; i += NW) {
      float racc = 0.f;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = lane + 64 * jj;
        if (j < K) {
          const float r = (rsl && i == j) ? 0.f : Rb[static_cast<long>(i) * K + j];
          sR[i * ld + j] = r;
          sG[i * ld + j] = g_one ? g_val : Gb[static_cast<long>(i) * K + j];
          racc += r;
          cacc[jj] += r;
        }
      }
      if (!cols) {
#pragma unroll
        for (int dd = 32; dd > 0; dd >>= 1) racc += __shfl_xor(racc, dd, WAVE);
        if (lane == 0) s_d[i] = racc;
      }
    }
    if (cols) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
        if (lane + 64 * jj < K) s_wpart[wave * K + lane + 64 * jj] = cacc[jj];
    }
    __syncthreads();
    for (int t = tid; t < K; t += T) {
      float c;
      if (cols) {
        c = 0.f;
#pragma unroll
        for (int q = 0; q < NW; ++q) c += s_wpart[q * K + t];
      } else {
        c = s_d[t];
      }
      s_gs[t] = c >= eps ? 1.f : 0.f;  // clamp(min = eps) passes the gradient where c >= eps
      s_d[t] = sqrtf(fmaxf(c, eps));
    }
    __syncthreads();
    cacc[0] = cacc[1] = 0.f;
    for (int i = wave; i < K; i += NW) {  // rowsum_i(G P) by the wave, colsum_j(G P) as per-wave partials
      const float di = s_d[i];
      float racc = 0.f;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = lane + 64 * jj;
        if (j < K) {
          const float dj = s_d[j], r = sR[i * ld + j];
          const float pv = cols ? (r / dj) / di : (r / di) / dj;  // the forward's P_ij, same arithmetic
          const float gp = sG[i * ld + j] * pv;
          racc += gp;
          cacc[jj] += gp;
        }
      }
#pragma unroll
      for (int dd = 32; dd > 0; dd >>= 1) racc += __shfl_xor(racc, dd, WAVE);
      if (lane == 0) s_rowq[i] = racc;
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
      if (lane + 64 * jj < K) s_wpart[wave * K + lane + 64 * jj] = cacc[jj];
    __syncthreads();
    for (int t = tid; t < K; t += T) {
      float cq = 0.f;
#pragma unroll
      for (int q = 0; q < NW; ++q) cq += s_wpart[q * K + t];
      const float d = s_d[t];
      s_gs[t] = s_gs[t] != 0.f ? -(s_rowq[t] + cq) / (2.0f * d * d) : 0.f;
    }
    __syncthreads();
    for (int i = wave; i < K; i += NW)
      for (int j = lane; j < K; j += 64) {
        const float v = sG[i * ld + j] * (1.0f / (s_d[i] * s_d[j])) + (cols ? s_gs[j] : s_gs[i]);
        ob[static_cast<long>(i) * K + j] = (rsl && i == j) ? 0.f : v;
      }
    return;
  }
  auto r1 = [&](int i, int j) -> float {
    if constexpr (IN_LDS) return sR[i * ld + j];
    return (rsl && i == j) ? 0.f : Rb[static_cast<long>(i) * K + j];
  };
  auto gv = [&](int i, int j) -> float {
    if constexpr (IN_LDS) return sG[i * ld + j];
    return g_one ? g_val : Gb[static_cast<long>(i) * K + j];
  };
  // column sums of f(i, j) into dst[j]: a thread per column, or (IN_LDS) 8 row phases per column folded in phase order
  auto col_sweep = [&](auto f, float* dst) {
    if constexpr (IN_LDS) {
      for (int idx = tid; idx < K * PARTS; idx += T) {
        const int j = idx % K, part = idx / K;
        float acc = 0.f;
        for (int i = part; i < K; i += PARTS) acc += f(i, j);
        s_part[part * K + j] = acc;
      }
      __syncthreads();
      for (int j = tid; j < K; j += T) {
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < PARTS; ++q) acc += s_part[q * K + j];
        dst[j] = acc;
      }
    } else {
      for (int j = tid; j < K; j += T) {
        float acc = 0.f;
#pragma unroll 4
        for (int i = 0; i < K; ++i) acc += f(i, j);
        dst[j] = acc;
      }
    }
  };
  if (!cols) {  // c_i = sum_j R1_ij: a wave per row
    for (int i = wave; i < K; i += NW) {
      float acc = 0.f;
      for (int j = lane; j < K; j += 64) acc += r1(i, j);
#pragma unroll
      for (int dd = 32; dd > 0; dd >>= 1) acc += __shfl_xor(acc, dd, WAVE);
      if (lane == 0) s_d[i] = acc;
    }
  } else {      // c_j = sum_i R1_ij
    col_sweep(r1, s_d);
  }
  __syncthreads();
  for (int t = tid; t < K; t += T) {
    const float c = s_d[t];
    s_gs[t] = c >= eps ? 1.f : 0.f;  // clamp(min = eps) passes the gradient where c >= eps
    s_d[t] = sqrtf(fmaxf(c, eps));
  }
  __syncthreads();
  auto pdiv = [&](int i, int j) -> float {  // the forward's P_ij, same arithmetic
    const float di = s_d[i], dj = s_d[j];
    return cols ? (r1(i, j) / dj) / di : (r1(i, j) / di) / dj;
  };
  if constexpr (IN_LDS) {  // P in place of R1 (each element read and written by one thread)
    for (int i = wave; i < K; i += NW)
      for (int j = lane; j < K; j += 64) sR[i * ld + j] = pdiv(i, j);
    __syncthreads();
  }
  auto pval = [&](int i, int j) -> float {
    if constexpr (IN_LDS) return sR[i * ld + j];
    return pdiv(i, j);
  };
  for (int i = wave; i < K; i += NW) {  // rowsum_i(G P)
    float acc = 0.f;
    for (int j = lane; j < K; j += 64) acc += gv(i, j) * pval(i, j);
#pragma unroll
    for (int dd = 32; dd > 0; dd >>= 1) acc += __shfl_xor(acc, dd, WAVE);
    if (lane == 0) s_rowq[i] = acc;
  }
  col_sweep([&](int i, int j) { return gv(i, j) * pval(i, j); }, s_colq);  // colsum_j(G P)
  __syncthreads();
  for (int t = tid; t < K; t += T) {
    const float d = s_d[t];
    s_gs[t] = s_gs[t] != 0.f ? -(s_rowq[t] + s_colq[t]) / (2.0f * d * d) : 0.f;
  }
  __syncthreads();
  for (int i = wave; i < K; i += NW)
    for (int j = lane; j < K; j += 64) {
      const float v = gv(i, j) * (1.0f / (s_d[i] * s_d[j])) + (cols ? s_gs[j] : s_gs[i]);
      ob[static_cast<long>(i) * K + j] = (rsl && i == j) ? 0.f : v;
    }
}
}  // namespace tgp

extern "C" int tgp_postprocess_dense_bwd_f32(const float* raw, const float* g_post, int64_t B, int64_t K, int flags,
                                             float eps, float* g_raw, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(B >= 0 && K >= 0, TGP_ERR_INVALID, "tgp_postprocess_dense_bwd_f32: negative size");
  if (B == 0 || K == 0) return TGP_OK;
  TGP_REQUIRE(raw && g_post && g_raw, TGP_ERR_INVALID, "tgp_postprocess_dense_bwd_f32: null pointer");
  TGP_REQUIRE(!(flags & TGP_EDGE_WEIGHT_NORM), TGP_ERR_INVALID,
              "tgp_postprocess_dense_bwd_f32: edge_weight_norm is not differentiated by this entry");
  TGP_REQUIRE(K <= 4096 && B < (1ll << 31), TGP_ERR_RANGE, "tgp_postprocess_dense_bwd_f32: K > 4096");
  if (K <= 128) {
    const size_t lds = (static_cast<size_t>(4 * K + 16 * K) + 2 * static_cast<size_t>(K) * (K + 1)) * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(post_bwd_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    hipLaunchKernelGGL(post_bwd_kernel<true>, dim3(static_cast<unsigned>(B)), dim3(1024), lds, stream, raw, g_post,
                       static_cast<int>(K), flags, eps, g_raw);
  } else {
    const size_t lds = static_cast<size_t>(4 * K) * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(post_bwd_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    hipLaunchKernelGGL(post_bwd_kernel<false>, dim3(static_cast<unsigned>(B)), dim3(256), lds, stream, raw, g_post,
                       static_cast<int>(K), flags, eps, g_raw);
  }
  return check_launch("tgp_postprocess_dense_bwd_f32");
}

extern "C" int tgp_postprocess_dense_f32(const float* src, float* dst, int64_t B, int64_t K, int flags, float eps, void* ws,
                                         size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(B >= 0 && K >= 0, TGP_ERR_INVALID, "tgp_postprocess_dense_f32: negative size");
  if (B == 0 || K == 0) return TGP_OK;
  TGP_REQUIRE(src && dst, TGP_ERR_INVALID, "tgp_postprocess_dense_f32: null pointer");
  TGP_REQUIRE(K <= 16000 && B < 65536, TGP_ERR_RANGE, "tgp_postprocess_dense_f32: K > 16000 or B >= 65536");
  TGP_REQUIRE(ws && ws_bytes >= tgp_postprocess_dense_workspace_bytes(B, K), TGP_ERR_WORKSPACE,
              "tgp_postprocess_dense_f32: workspace too small");
  PostArgs q{};
  q.src = src; q.splits = 1; q.s_split = 0; q.s_batch = K * K; q.ld_src = K;
  q.K = static_cast<int>(K); q.flags = flags; q.eps = eps; q.raw = nullptr; q.dst = dst;
  launch_post(q, B, static_cast<float*>(ws), stream);
  return check_launch("tgp_postprocess_dense_f32");
}


// Generic batched fp32 GEMM on the matrix cores (used by Lift and by the unbatched dense paths).
static int bmm_impl(const float* A, const float* Bm, float* C, int64_t batch, int64_t M, int64_t Nc, int64_t Kd,
                    int trans_a, int64_t lda, int64_t ldb, int64_t ldc, int64_t sA, int64_t sB, int64_t sC,
                    int accumulate, void* stream_);

extern "C" int tgp_bmm_f32(const float* A, const float* Bm, float* C, int64_t batch, int64_t M, int64_t Nc,
                           int64_t Kd, int trans_a, int64_t lda, int64_t ldb, int64_t ldc, int64_t sA,
                           int64_t sB, int64_t sC, void* stream_) {
  return bmm_impl(A, Bm, C, batch, M, Nc, Kd, trans_a, lda, ldb, ldc, sA, sB, sC, 0, stream_);
}

// C += op(A) Bm: the second term of a two-term gradient (dS = U dR^T + V dR, connect/dense_conn.py:111-122 under
// autograd) accumulates in the GEMM epilogue into the buffer the first term wrote, instead of a separate elementwise add
extern "C" int tgp_bmm_accumulate_f32(const float* A, const float* Bm, float* C, int64_t batch, int64_t M, int64_t Nc,
                                      int64_t Kd, int trans_a, int64_t lda, int64_t ldb, int64_t ldc, int64_t sA,
                                      int64_t sB, int64_t sC, void* stream_) {
  return bmm_impl(A, Bm, C, batch, M, Nc, Kd, trans_a, lda, ldb, ldc, sA, sB, sC, 1, stream_);
}

static int bmm_impl(const float* A, const float* Bm, float* C, int64_t batch, int64_t M, int64_t Nc, int64_t Kd,
                    int trans_a, int64_t lda, int64_t ldb, int64_t ldc, int64_t sA, int64_t sB, int64_t sC,
                    int accumulate, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(batch >= 0 && M >= 0 && Nc >= 0 && Kd >= 0, TGP_ERR_INVALID, "tgp_bmm_f32: negative size");
  if (batch == 0 || M == 0 || Nc == 0) return TGP_OK;
  TGP_REQUIRE(C && (Kd == 0 || (A && Bm)), TGP_ERR_INVALID, "tgp_bmm_f32: null pointer");
  TGP_REQUIRE(M < (1ll << 31) && Nc < (1ll << 31) && Kd < (1ll << 31), TGP_ERR_RANGE, "tgp_bmm_f32: too large");
  GemmArgs g{};
  g.A = A; g.lda = lda; g.sA = sA;
  g.M = static_cast<int>(M); g.Kd = static_cast<int>(Kd);
  g.rhs[0] = GemmRhs{Bm, C, static_cast<int>(Nc), ldb, ldc, sB, sC, 0};
  g.splits = 1; g.k_per_split = static_cast<int>((Kd + BK - 1) / BK * BK);
  g.accumulate = accumulate;
  TGP_REQUIRE(batch * ((M + 63) / 64) * ((Nc + 63) / 64) < (1ll << 31), TGP_ERR_RANGE, "tgp_bmm_f32: grid too large");
  // many small matrices: one wave per 32-row strip, operands straight from memory (see small_bmm_kernel)
  static const int no_small_bmm = getenv("TGP_NO_SMALL_BMM") ? 1 : 0;
  const int64_t strips = (M + 31) / 32;
  if (!no_small_bmm && !accumulate && Nc <= 64 && M <= 512 && Kd <= 512 && batch * strips >= 512) {
    SmallBmmArgs q{A, Bm, C, static_cast<int>(M), static_cast<int>(Nc), static_cast<int>(Kd), trans_a ? 1 : 0,
                   lda, ldb, ldc, sA, sB, sC, static_cast<int>(strips)};
    const long total = batch * strips;
    if (Nc <= 32)
      hipLaunchKernelGGL(small_bmm_kernel<1>, dim3(static_cast<unsigned>((total + 3) / 4)), dim3(256), 0, stream, q, total);
    else
      hipLaunchKernelGGL(small_bmm_kernel<2>, dim3(static_cast<unsigned>((total + 3) / 4)), dim3(256), 0, stream, q, total);
    return check_launch("tgp_bmm_f32(small)");
  }
  if (trans_a) launch_gemm<true>(g, static_cast<int>(batch), stream);
  else launch_gemm<false>(g, static_cast<int>(batch), stream);
  return check_launch("tgp_bmm_f32");
}

// A3' / A7' (reduce/base_reduce.py:170-182, connect/dense_conn.py:195-206): per-graph S_b^T Y_b for an
// un-padded batch.  Graph b owns node rows ptr[b]..ptr[b+1] of S [Ntot,K] and Y [Ntot,F]; C is [B,K,F].
// Replaces the reference's Python loop over graphs with one launch (no padding, no densification).
// Graphs are few and long in the unbatched modes (e.g. 2 graphs of 8192 nodes), so the node range of a graph
// is split across workgroups like the batched path does; the partial products go to slabs [B][splits][K][F]
// and are added in split order (deterministic).
static int segment_splits(int64_t B, int64_t K, int64_t F, int64_t span) {
  const int64_t tiles = ((K + 63) / 64) * ((F + 63) / 64);
  const int64_t wgs = (B > 0 ? B : 1) * tiles;
  int64_t splits = (3 * 256 + wgs - 1) / wgs;                       // aim at ~3 workgroups per CU
  const int64_t max_splits = (span + 4 * BK - 1) / (4 * BK);        // keep >= 4 k-steps per workgroup
  if (splits > max_splits) splits = max_splits;
  if (splits > 64) splits = 64;
  return splits < 1 ? 1 : static_cast<int>(splits);
}

extern "C" size_t tgp_segment_gemm_tn_workspace_bytes(int64_t B, int64_t K, int64_t F, int64_t max_nodes) {
  if (B <= 0 || K <= 0 || F <= 0) return 256;
  const int splits = segment_splits(B, K, F, max_nodes);
  return (splits > 1 ? align_up(static_cast<size_t>(B) * splits * K * F * sizeof(float)) : 0) + 256;
}

extern "C" int tgp_segment_gemm_tn_f32(const float* S, const float* Y, const int64_t* ptr, float* C, int64_t B,
                                       int64_t Ntot, int64_t K, int64_t F, int64_t max_nodes, void* ws,
                                       size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(B >= 0 && Ntot >= 0 && K >= 0 && F >= 0, TGP_ERR_INVALID, "tgp_segment_gemm_tn_f32: negative size");
  if (B == 0 || K == 0 || F == 0) return TGP_OK;
  TGP_REQUIRE(C && ptr && (Ntot == 0 || (S && Y)), TGP_ERR_INVALID, "tgp_segment_gemm_tn_f32: null pointer");
  TGP_REQUIRE(Ntot < (1ll << 31) && K < (1ll << 31) && F < (1ll << 31), TGP_ERR_RANGE,
              "tgp_segment_gemm_tn_f32: too large");
  const int64_t span = max_nodes > 0 ? max_nodes : Ntot;
  const int splits = segment_splits(B, K, F, span);
  TGP_REQUIRE(splits == 1 || (ws && ws_bytes >= tgp_segment_gemm_tn_workspace_bytes(B, K, F, max_nodes)),
              TGP_ERR_WORKSPACE, "tgp_segment_gemm_tn_f32: workspace too small");
  TGP_REQUIRE(B * splits * ((K + 63) / 64) * ((F + 63) / 64) < (1ll << 31), TGP_ERR_RANGE,
              "tgp_segment_gemm_tn_f32: grid too large");
  float* slab = splits > 1 ? static_cast<float*>(ws) : C;
  GemmArgs g{};
  g.A = S; g.lda = K; g.sA = 0;
  g.M = static_cast<int>(K); g.Kd = static_cast<int>(Ntot);
  g.rhs[0] = GemmRhs{Y, slab, static_cast<int>(F), F, F, 0, static_cast<long>(splits) * K * F, K * F};
  g.splits = splits;
  int64_t kps = ((span + splits - 1) / splits + BK - 1) / BK * BK;
  if (kps < BK) kps = BK;
  g.k_per_split = static_cast<int>(kps);
  g.k_ptr = ptr;
  launch_gemm<true>(g, static_cast<int>(B), stream);
  if (splits > 1) {
    const long total = K * F;
    int gx = static_cast<int>((total + 255) / 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(combine_slabs_kernel, dim3(gx, static_cast<unsigned>(B)), dim3(256), 0, stream, slab, splits,
                       K * F, static_cast<long>(splits) * K * F, total, C);
  }
  return check_launch("tgp_segment_gemm_tn_f32");
}

// r6: the two segment products with explicit row strides, for operands that are column blocks of a wider buffer (the
// unbatched training step's operand buffer [T | X | 1000 | S | T'], functions._PoolUnbatchedFn):
//   nn: C[rows of b] = A[rows of b, 0:Kd] Bm[b]        (A row stride lda, Bm [B][Kd][Nc] with row stride ldb / batch sB)
//   tn: C[b] = A[rows of b]^T Y[rows of b]             (no node-range split: the caller passes short row ranges)
extern "C" int tgp_segment_gemm_nn_ld_f32(const float* A, int64_t lda, const float* Bm, int64_t ldb, int64_t sB,
                                          const int64_t* ptr, float* C, int64_t ldc, int64_t B, int64_t Ntot, int64_t Kd,
                                          int64_t Nc, int64_t max_nodes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(B >= 0 && Ntot >= 0 && Kd >= 0 && Nc >= 0 && lda >= Kd && ldb >= Nc && ldc >= Nc, TGP_ERR_INVALID,
              "tgp_segment_gemm_nn_ld_f32: bad sizes");
  if (B == 0 || Ntot == 0 || Nc == 0) return TGP_OK;
  TGP_REQUIRE(C && ptr && (Kd == 0 || (A && Bm)), TGP_ERR_INVALID, "tgp_segment_gemm_nn_ld_f32: null pointer");
  TGP_REQUIRE(Ntot < (1ll << 31) && Kd < (1ll << 31) && Nc < (1ll << 31), TGP_ERR_RANGE,
              "tgp_segment_gemm_nn_ld_f32: too large");
  const int64_t span = max_nodes > 0 ? max_nodes : Ntot;
  TGP_REQUIRE(B * ((span + 63) / 64) * ((Nc + 63) / 64) < (1ll << 31), TGP_ERR_RANGE,
              "tgp_segment_gemm_nn_ld_f32: grid too large");
  GemmArgs g{};
  g.A = A; g.lda = lda; g.sA = 0;
  g.M = static_cast<int>(span); g.Kd = static_cast<int>(Kd);
  g.rhs[0] = GemmRhs{Bm, C, static_cast<int>(Nc), ldb, ldc, sB, 0, 0};
  g.splits = 1; g.k_per_split = static_cast<int>((Kd + BK - 1) / BK * BK);
  if (g.k_per_split < BK) g.k_per_split = BK;
  g.m_ptr = ptr;
  launch_gemm<false>(g, static_cast<int>(B), stream);
  return check_launch("tgp_segment_gemm_nn_ld_f32");
}

extern "C" int tgp_segment_gemm_tn_ld_f32(const float* A, int64_t lda, const float* Y, int64_t ldy, const int64_t* ptr,
                                          float* C, int64_t B, int64_t Ntot, int64_t M, int64_t Nc, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(B >= 0 && Ntot >= 0 && M >= 0 && Nc >= 0 && lda >= M && ldy >= Nc, TGP_ERR_INVALID,
              "tgp_segment_gemm_tn_ld_f32: bad sizes");
  if (B == 0 || M == 0 || Nc == 0) return TGP_OK;
  TGP_REQUIRE(C && ptr && (Ntot == 0 || (A && Y)), TGP_ERR_INVALID, "tgp_segment_gemm_tn_ld_f32: null pointer");
  TGP_REQUIRE(Ntot < (1ll << 31) && M < (1ll << 31) && Nc < (1ll << 31) &&
                  B * ((M + 63) / 64) * ((Nc + 63) / 64) < (1ll << 31), TGP_ERR_RANGE,
              "tgp_segment_gemm_tn_ld_f32: too large");
  GemmArgs g{};
  g.A = A; g.lda = lda; g.sA = 0;
  g.M = static_cast<int>(M); g.Kd = static_cast<int>(Ntot);
  g.rhs[0] = GemmRhs{Y, C, static_cast<int>(Nc), ldy, Nc, 0, M * Nc, 0};
  g.splits = 1;
  g.k_per_split = static_cast<int>((Ntot + BK - 1) / BK * BK);
  if (g.k_per_split < BK) g.k_per_split = BK;
  g.k_ptr = ptr;
  launch_gemm<true>(g, static_cast<int>(B), stream);
  return check_launch("tgp_segment_gemm_tn_ld_f32");
}

// r6: the same product with up to THREE right-hand sides in one grid -- C_j[b] = S_b^T Y_j,b -- for the unbatched dense
// poolers' forward: S^T [T | X | S] gives the raw pooled adjacency (T = A S from the CSR SpMM), the pooled features and
// the per-graph Gram matrices (orthogonality / link losses) from one pass over S (dense_conn.py:195-206,
// base_reduce.py:170-182, utils/losses.py:204-240).  Slabs [B][splits][K][F_j] per right-hand side, combined in split
// order by one launch.
namespace tgp {
struct Combine3Args {
  const float* src[3]; float* dst[3]; long total[3];
  int splits;
  int transpose0, K;  // output 0 is K x K and is written transposed (raw = S^T A^T S from the slabs of S^T (A S))
};
__global__ __launch_bounds__(256) void combine_slabs3_kernel(Combine3Args a) {
  const int b = blockIdx.y, j = blockIdx.z;
  const long total = a.total[j];
  if (total == 0) return;
  const float* sb = a.src[j] + static_cast<long>(b) * a.splits * total;
  float* db = a.dst[j] + static_cast<long>(b) * total;
  const bool tr = j == 0 && a.transpose0;
  for (long e = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; e < total; e += static_cast<long>(gridDim.x) * 256) {
    float v = sb[e];
    for (int sp = 1; sp < a.splits; ++sp) v = __fadd_rn(v, sb[sp * total + e]);
    const long r = e / a.K;
    db[tr ? (e - r * a.K) * a.K + r : e] = v;
  }
}
}  // namespace tgp

static int segment3_splits(int64_t B, int64_t K, int64_t F0, int64_t F1, int64_t F2, int64_t span) {
  const int64_t tiles = ((K + 63) / 64) * (((F0 + 63) / 64) + ((F1 + 63) / 64) + ((F2 + 63) / 64));
  const int64_t wgs = (B > 0 ? B : 1) * (tiles > 0 ? tiles : 1);
  int64_t splits = (3 * 256 + wgs - 1) / wgs;
  const int64_t max_splits = (span + 4 * BK - 1) / (4 * BK);
  if (splits > max_splits) splits = max_splits;
  if (splits > 64) splits = 64;
  return splits < 1 ? 1 : static_cast<int>(splits);
}

extern "C" size_t tgp_segment_gemm_tn3_workspace_bytes(int64_t B, int64_t K, int64_t F0, int64_t F1, int64_t F2,
                                                       int64_t max_nodes) {
  if (B <= 0 || K <= 0) return 256;
  const int splits = segment3_splits(B, K, F0, F1, F2, max_nodes);
  size_t bytes = 256;
  const int64_t fs[3] = {F0, F1, F2};
  for (int j = 0; j < 3; ++j)
    if (fs[j] > 0) bytes += align_up(static_cast<size_t>(B) * splits * K * fs[j] * sizeof(float));
  return bytes;
}

extern "C" int tgp_segment_gemm_tn3_f32(const float* S, const float* Y0, int64_t F0, const float* Y1, int64_t F1,
                                        const float* Y2, int64_t F2, const int64_t* ptr, float* C0, float* C1, float* C2,
                                        int64_t B, int64_t Ntot, int64_t K, int64_t max_nodes, int transpose0, void* ws,
                                        size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(B >= 0 && Ntot >= 0 && K >= 0 && F0 > 0 && F1 >= 0 && F2 >= 0 && (F1 > 0 || F2 == 0), TGP_ERR_INVALID,
              "tgp_segment_gemm_tn3_f32: bad sizes (right-hand sides are filled front to back)");
  if (B == 0 || K == 0) return TGP_OK;
  TGP_REQUIRE(ptr && C0 && (F1 == 0 || C1) && (F2 == 0 || C2) && (Ntot == 0 || (S && Y0 && (F1 == 0 || Y1) && (F2 == 0 || Y2))),
              TGP_ERR_INVALID, "tgp_segment_gemm_tn3_f32: null pointer");
  TGP_REQUIRE(Ntot < (1ll << 31) && K < (1ll << 31) && F0 < (1ll << 31) && F1 < (1ll << 31) && F2 < (1ll << 31),
              TGP_ERR_RANGE, "tgp_segment_gemm_tn3_f32: too large");
  const int64_t span = max_nodes > 0 ? max_nodes : Ntot;
  const int splits = segment3_splits(B, K, F0, F1, F2, span);
  TGP_REQUIRE(ws && ws_bytes >= tgp_segment_gemm_tn3_workspace_bytes(B, K, F0, F1, F2, max_nodes), TGP_ERR_WORKSPACE,
              "tgp_segment_gemm_tn3_f32: workspace too small");
  const float* Y[3] = {Y0, Y1, Y2};
  float* C[3] = {C0, C1, C2};
  const int64_t fs[3] = {F0, F1, F2};
  Carver cv(ws);
  GemmArgs g{};
  g.A = S; g.lda = K; g.sA = 0;
  g.M = static_cast<int>(K); g.Kd = static_cast<int>(Ntot);
  TGP_REQUIRE(!transpose0 || F0 == K, TGP_ERR_INVALID, "tgp_segment_gemm_tn3_f32: transpose0 needs a K x K first output");
  Combine3Args ca{};
  ca.splits = splits;
  ca.transpose0 = transpose0 ? 1 : 0;
  ca.K = static_cast<int>(K);
  long max_total = 0;
  for (int j = 0; j < 3; ++j) {
    if (fs[j] <= 0) continue;
    float* slab = cv.take<float>(static_cast<size_t>(B) * splits * K * fs[j]);
    g.rhs[j] = GemmRhs{Y[j], slab, static_cast<int>(fs[j]), fs[j], fs[j], 0, static_cast<long>(splits) * K * fs[j], K * fs[j]};
    ca.src[j] = slab; ca.dst[j] = C[j]; ca.total[j] = K * fs[j];
    if (ca.total[j] > max_total) max_total = ca.total[j];
  }
  g.splits = splits;
  int64_t kps = ((span + splits - 1) / splits + BK - 1) / BK * BK;
  if (kps < BK) kps = BK;
  g.k_per_split = static_cast<int>(kps);
  g.k_ptr = ptr;
  g.force_bm = 64; g.force_bn = 64;
  launch_gemm<true>(g, static_cast<int>(B), stream);
  int gx = static_cast<int>((max_total + 255) / 256);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(combine_slabs3_kernel, dim3(gx, static_cast<unsigned>(B), 3), dim3(256), 0, stream, ca);
  return check_launch("tgp_segment_gemm_tn3_f32");
}

// r6: the same three products with the post-processing of the first one (raw = S^T (A S) -> adj_pool, utils/ops.py:282-335)
// behind them in ONE more launch: for 64 < K <= 176 the workgroup that post-processes a graph sums that graph's raw
// slabs itself (post_lds_kernel) and the X' / Gram slabs are summed by extra workgroups of the same launch -- the
// combine launch and the re-read of raw are gone (mincut_u / diff_u forward at C2: 3 -> 2 launches behind the product,
// -6 us).  Other K: the combine launch + the post-processing kernels of that size.  Same adds in the same (slab) order.
extern "C" size_t tgp_segment_gemm_tn3_post_workspace_bytes(int64_t B, int64_t K, int64_t F1, int64_t F2, int64_t max_nodes) {
  if (B <= 0 || K <= 0) return 256;
  return tgp_segment_gemm_tn3_workspace_bytes(B, K, K, F1, F2, max_nodes) + align_up(post_ws_floats(B, K) * 4) + 256;
}

extern "C" int tgp_segment_gemm_tn3_post_f32(const float* S, const float* Y0, const float* Y1, int64_t F1, const float* Y2,
                                             int64_t F2, const int64_t* ptr, float* raw, float* C1, float* C2,
                                             float* adj_pool, int64_t B, int64_t Ntot, int64_t K, int64_t max_nodes,
                                             int transpose0, int flags, float eps, void* ws, size_t ws_bytes,
                                             void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(B >= 0 && Ntot >= 0 && K >= 0 && F1 >= 0 && F2 >= 0 && (F1 > 0 || F2 == 0), TGP_ERR_INVALID,
              "tgp_segment_gemm_tn3_post_f32: bad sizes (right-hand sides are filled front to back)");
  if (B == 0 || K == 0) return TGP_OK;
  TGP_REQUIRE(ptr && raw && adj_pool && (F1 == 0 || C1) && (F2 == 0 || C2) &&
                  (Ntot == 0 || (S && Y0 && (F1 == 0 || Y1) && (F2 == 0 || Y2))),
              TGP_ERR_INVALID, "tgp_segment_gemm_tn3_post_f32: null pointer");
  TGP_REQUIRE(Ntot < (1ll << 31) && K <= 16000 && B < 65536 && F1 < (1ll << 31) && F2 < (1ll << 31), TGP_ERR_RANGE,
              "tgp_segment_gemm_tn3_post_f32: too large");
  TGP_REQUIRE(ws && ws_bytes >= tgp_segment_gemm_tn3_post_workspace_bytes(B, K, F1, F2, max_nodes), TGP_ERR_WORKSPACE,
              "tgp_segment_gemm_tn3_post_f32: workspace too small");
  const int64_t span = max_nodes > 0 ? max_nodes : Ntot;
  const int splits = segment3_splits(B, K, K, F1, F2, span);
  const float* Y[3] = {Y0, Y1, Y2};
  float* C[3] = {raw, C1, C2};
  const int64_t fs[3] = {K, F1, F2};
  Carver cv(ws);
  GemmArgs g{};
  g.A = S; g.lda = K; g.sA = 0;
  g.M = static_cast<int>(K); g.Kd = static_cast<int>(Ntot);
  Combine3Args ca{};
  ca.splits = splits;
  ca.transpose0 = transpose0 ? 1 : 0;
  ca.K = static_cast<int>(K);
  float* slab[3] = {nullptr, nullptr, nullptr};
  for (int j = 0; j < 3; ++j) {
    if (fs[j] <= 0) continue;
    slab[j] = cv.take<float>(static_cast<size_t>(B) * splits * K * fs[j]);
    g.rhs[j] = GemmRhs{Y[j], slab[j], static_cast<int>(fs[j]), fs[j], fs[j], 0, static_cast<long>(splits) * K * fs[j], K * fs[j]};
    ca.src[j] = slab[j]; ca.dst[j] = C[j]; ca.total[j] = K * fs[j];
  }
  float* postws = cv.take<float>(post_ws_floats(B, K));
  g.splits = splits;
  int64_t kps = ((span + splits - 1) / splits + BK - 1) / BK * BK;
  if (kps < BK) kps = BK;
  g.k_per_split = static_cast<int>(kps);
  g.k_ptr = ptr;
  g.force_bm = 64; g.force_bn = 64;
  launch_gemm<true>(g, static_cast<int>(B), stream);
  PostArgs q{};
  q.K = static_cast<int>(K); q.flags = flags; q.eps = eps; q.dst = adj_pool; q.ld_src = K;
  bool folded = false;
  // (r6, late) K % 16 == 0 in post_lds_kernel's range: that kernel also transposes while it sums the slabs
  static const int no_tr_fold = getenv("TGP_NO_TRANSPOSED_POST") ? 1 : 0;
  PostArgs qs = q;  // the form that reads the slabs of the first product itself
  qs.src = slab[0]; qs.splits = splits; qs.s_split = K * K; qs.s_batch = static_cast<long>(splits) * K * K; qs.raw = raw;
  const bool fold_tr = transpose0 && !no_tr_fold && K % 16 == 0 && post_lds_takes(qs);
  if (transpose0 && !fold_tr) {  // raw leaves the combine transposed; the post-processing then reads it as a finished matrix
    q.src = raw; q.splits = 1; q.s_split = 0; q.s_batch = K * K; q.raw = nullptr;
  } else {
    q = qs;
    q.transpose_src = transpose0 ? 1 : 0;
    ca.total[0] = 0;  // (summed by the post-processing itself)
  }
  long max_total = 0;
  for (int j = 0; j < 3; ++j)
    if (ca.total[j] > max_total) max_total = ca.total[j];
  auto combine = [&]() {
    if (max_total == 0) return;
    int gx = static_cast<int>((max_total + 255) / 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(combine_slabs3_kernel, dim3(gx, static_cast<unsigned>(B), 3), dim3(256), 0, stream, ca);
  };
  if (transpose0 && !fold_tr) {
    combine();
    launch_post(q, B, postws, stream);
  } else {
    XCombineArgs x1{slab[1], splits, K * F1, static_cast<long>(splits) * K * F1, K * F1, C1, 0};
    XCombineArgs x2{slab[2], splits, K * F2, static_cast<long>(splits) * K * F2, K * F2, C2, 0};
    folded = launch_post(q, B, postws, stream, F1 > 0 ? &x1 : nullptr, F2 > 0 ? &x2 : nullptr);
    if (!folded) combine();
  }
  return check_launch("tgp_segment_gemm_tn3_post_f32");
}

// Row-side counterpart (lift/base_lift.py:138-247 on an un-padded batch; backward of the products above):
// C[rows of graph b] = A[rows of graph b] Bm[b], A [Ntot,Kd], Bm [B,Kd,Nc], C [Ntot,Nc]; one launch.
extern "C" int tgp_segment_gemm_nn_f32(const float* A, const float* Bm, const int64_t* ptr, float* C, int64_t B,
                                       int64_t Ntot, int64_t Kd, int64_t Nc, int64_t max_nodes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(B >= 0 && Ntot >= 0 && Kd >= 0 && Nc >= 0, TGP_ERR_INVALID, "tgp_segment_gemm_nn_f32: negative size");
  if (B == 0 || Ntot == 0 || Nc == 0) return TGP_OK;
  TGP_REQUIRE(C && ptr && (Kd == 0 || (A && Bm)), TGP_ERR_INVALID, "tgp_segment_gemm_nn_f32: null pointer");
  TGP_REQUIRE(Ntot < (1ll << 31) && Kd < (1ll << 31) && Nc < (1ll << 31), TGP_ERR_RANGE,
              "tgp_segment_gemm_nn_f32: too large");
  const int64_t span = max_nodes > 0 ? max_nodes : Ntot;
  TGP_REQUIRE(B * ((span + 63) / 64) * ((Nc + 63) / 64) < (1ll << 31), TGP_ERR_RANGE,
              "tgp_segment_gemm_nn_f32: grid too large");
  GemmArgs g{};
  g.A = A; g.lda = Kd; g.sA = 0;
  g.M = static_cast<int>(span); g.Kd = static_cast<int>(Kd);
  g.rhs[0] = GemmRhs{Bm, C, static_cast<int>(Nc), Nc, Nc, Kd * Nc, 0, 0};
  g.splits = 1; g.k_per_split = static_cast<int>((Kd + BK - 1) / BK * BK);
  if (g.k_per_split < BK) g.k_per_split = BK;
  g.m_ptr = ptr;
  launch_gemm<false>(g, static_cast<int>(B), stream);
  return check_launch("tgp_segment_gemm_nn_f32");
}

// N3: DiffPool's link-prediction residual ||A - S S^T||_F^2 per graph (utils/losses.py:644-708 computes
// torch.norm(adj - S S^T) after materialising S S^T [B,N,N]).  Here S S^T tiles live only in the MFMA
// accumulators; the epilogue subtracts them from the A tile and reduces the squares.  sq[b] is summed in a
// fixed order (tile partials in tile order), so the result is reproducible run to run.
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, int T,
                                                           float* __restrict__ out) {
  __shared__ float sh[256];
  const float* p = partial + static_cast<long>(blockIdx.x) * T;
  float s = 0.f;
  for (int i = threadIdx.x; i < T; i += 256) s += p[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (static_cast<int>(threadIdx.x) < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = sh[0];
}

static tgp::GemmArgs link_loss_args(const float* S, const float* A, int64_t N, int64_t K) {
  tgp::GemmArgs g{};
  g.A = S; g.lda = K; g.sA = N * K;
  g.M = static_cast<int>(N); g.Kd = static_cast<int>(K);
  g.rhs[0] = tgp::GemmRhs{S, nullptr, static_cast<int>(N), K, 0, N * K, 0, 0};
  g.splits = 1; g.k_per_split = static_cast<int>((K + tgp::BK - 1) / tgp::BK * tgp::BK);
  g.resid = A; g.ldr = N; g.sR = N * N;
  g.symmetric = 1;  // S S^T: tiles below the diagonal are their mirror images' job (launch_gemm_residual checks the rest)
  return g;
}

extern "C" size_t tgp_link_loss_workspace_bytes(int64_t B, int64_t N, int64_t K) {
  if (B <= 0 || N <= 0) return 256;
  tgp::GemmArgs g = link_loss_args(nullptr, nullptr, N, K);
  const int tiles = tgp::launch_gemm_residual(g, static_cast<int>(B), nullptr, true);
  return tgp::align_up(static_cast<size_t>(B) * tiles * sizeof(float)) + 256;
}

extern "C" int tgp_link_loss_f32(const float* S, const float* A, int64_t B, int64_t N, int64_t K,
                                 const int64_t* graph_sizes, float* sq, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(B >= 0 && N >= 0 && K >= 0, TGP_ERR_INVALID, "tgp_link_loss_f32: negative size");
  if (B == 0) return TGP_OK;
  TGP_REQUIRE(sq, TGP_ERR_INVALID, "tgp_link_loss_f32: null output");
  if (N == 0) {
    (void)hipMemsetAsync(sq, 0, B * sizeof(float), stream);
    return tgp::check_launch("tgp_link_loss_f32");
  }
  TGP_REQUIRE(S && A && ws, TGP_ERR_INVALID, "tgp_link_loss_f32: null pointer");
  TGP_REQUIRE(N < (1ll << 31) && K < (1ll << 31), TGP_ERR_RANGE, "tgp_link_loss_f32: too large");
  TGP_REQUIRE(ws_bytes >= tgp_link_loss_workspace_bytes(B, N, K), TGP_ERR_WORKSPACE,
              "tgp_link_loss_f32: workspace too small");
  TGP_REQUIRE(B * ((N + 63) / 64) * ((N + 63) / 64) < (1ll << 31), TGP_ERR_RANGE, "tgp_link_loss_f32: grid too large");
  tgp::GemmArgs g = link_loss_args(S, A, N, K);
  g.partial = static_cast<float*>(ws);
  g.sizes = graph_sizes;
  const int tiles = tgp::launch_gemm_residual(g, static_cast<int>(B), stream);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(static_cast<unsigned>(B)), dim3(256), 0, stream, g.partial, tiles, sq);
  return tgp::check_launch("tgp_link_loss_f32");
}

#ifdef TGP_GEMM_STAMPS
extern "C" int tgp_debug_set_gemm_reverse(int on) {
  return hipMemcpyToSymbol(HIP_SYMBOL(tgp::g_gemm_reverse), &on, sizeof(on)) == hipSuccess ? 0 : -3;
}
extern "C" int tgp_debug_set_gemm_stamps(unsigned long long* device_buffer) {
  return hipMemcpyToSymbol(HIP_SYMBOL(tgp::g_gemm_stamps), &device_buffer, sizeof(device_buffer)) == hipSuccess ? 0 : -3;
}
#endif
