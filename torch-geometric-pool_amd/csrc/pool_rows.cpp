// r6 (late): the forward of the dense poolers' un-padded rows route as ONE native call -- the launches are those of the
// entries it strings together (include/tgp_hip.h: tgp_mlp_select_f32, tgp_spmm_csr_{,stats_,entropy_}f32,
// tgp_segment_gemm_tn3_post_f32, tgp_mincut_terms_fused_f32 / tgp_diffpool_unbatched_tail_f32); what it removes is the host
// side between them: four Python wrappers with their argument checks and ctypes marshalling were ~85 us of a forward
// whose kernels take ~100 us (profiles/r06_host_time_train.txt).  Reference path: poolers/mincut.py:220-289,
// diffpool.py:208-284 (unbatched branches), connect/dense_conn.py:140-208, reduce/base_reduce.py:170-182,
// utils/losses.py:73-127,204-240,661-708.
#include "common.h"
#include "../../include/tgp_hip.h"

using namespace tgp;

extern "C" size_t tgp_pool_rows_fwd_workspace_bytes(int64_t B, int64_t K, int64_t F, int64_t max_nodes, int64_t Ntot) {
  return tgp_segment_gemm_tn3_post_workspace_bytes(B, K, F, K, max_nodes) +
         align_up(tgp_entropy_sum_workspace_bytes(Ntot > 0 && K > 0 ? Ntot * K : 1)) + 256;
}

extern "C" int tgp_pool_rows_fwd_f32(const float* x, int64_t Ntot, int64_t F, const float* W, const float* bias, float* S,
                                     const int32_t* row_ptr, const int64_t* col, const float* w, int64_t nnz,
                                     const int64_t* ptr, int64_t B, int64_t K, int64_t max_nodes, int transposed,
                                     int post_flags, float eps, float loss_eps, int mode, const float* sw2_dev,
                                     float sw2_host, float link_scale, float ent_scale, float* T, float* raw,
                                     float* x_pool, float* gram, float* adj_pool, float* rowstat, float* den, float* terms,
                                     float* stats, float* means, uint32_t* ticket, float* dstats, float* out2, void* ws,
                                     size_t ws_bytes, void* stream) {
  TGP_REQUIRE(Ntot >= 1 && F >= 1 && K >= 1 && B >= 1 && nnz >= 0 && mode >= 0 && mode <= 2, TGP_ERR_INVALID,
              "tgp_pool_rows_fwd_f32: bad shape or mode");
  TGP_REQUIRE(x && S && row_ptr && ptr && T && raw && x_pool && adj_pool && ws && (nnz == 0 || col), TGP_ERR_INVALID,
              "tgp_pool_rows_fwd_f32: null pointer");
  TGP_REQUIRE(mode == 0 || gram, TGP_ERR_INVALID, "tgp_pool_rows_fwd_f32: the losses need the Gram matrix");
  TGP_REQUIRE(mode != 1 || (rowstat && den && terms && stats), TGP_ERR_INVALID, "tgp_pool_rows_fwd_f32: MinCut outputs");
  TGP_REQUIRE(mode != 2 || (rowstat && dstats && out2), TGP_ERR_INVALID, "tgp_pool_rows_fwd_f32: DiffPool outputs");
  TGP_REQUIRE(ws_bytes >= tgp_pool_rows_fwd_workspace_bytes(B, K, F, max_nodes, Ntot), TGP_ERR_WORKSPACE,
              "tgp_pool_rows_fwd_f32: workspace too small");
  int rc = TGP_OK;
  if (W) {
    TGP_REQUIRE(K <= tgp_mlp_select_max_fused_k(), TGP_ERR_INVALID, "tgp_pool_rows_fwd_f32: K beyond the fused selector");
    rc = tgp_mlp_select_f32(x, W, bias, nullptr, Ntot, F, K, S, stream);
    if (rc != TGP_OK) return rc;
  }
  const size_t tn3_bytes = tgp_segment_gemm_tn3_post_workspace_bytes(B, K, F, K, max_nodes);
  char* ws_ent = static_cast<char*>(ws) + align_up(tn3_bytes);
  const float* ent_partial = rowstat;
  int n_partial = 0;
  if (mode == 1) {
    rc = tgp_spmm_csr_stats_f32(row_ptr, col, w, Ntot, nnz, S, K, T, rowstat, rowstat + Ntot, stream);
  } else if (mode == 2) {
    rc = tgp_spmm_csr_entropy_f32(row_ptr, col, w, Ntot, nnz, S, K, T, loss_eps, rowstat, &n_partial, stream);
    if (rc == TGP_OK && n_partial < 0) {  // (shapes the row kernel does not take: the loss' own pass over S)
      rc = tgp_entropy_partials_f32(S, Ntot * K, loss_eps, ws_ent, ws_bytes - align_up(tn3_bytes), &n_partial, stream);
      ent_partial = reinterpret_cast<const float*>(ws_ent);
    }
  } else {
    rc = tgp_spmm_csr_f32(row_ptr, col, w, Ntot, nnz, S, K, T, stream);
  }
  if (rc != TGP_OK) return rc;
  rc = tgp_segment_gemm_tn3_post_f32(S, T, x, F, mode ? S : nullptr, mode ? K : 0, ptr, raw, x_pool, gram, adj_pool, B,
                                     Ntot, K, max_nodes, transposed, post_flags, eps, ws, tn3_bytes, stream);
  if (rc != TGP_OK) return rc;
  if (mode == 1)
    return tgp_mincut_terms_fused_f32(raw, gram, transposed ? nullptr : rowstat, rowstat + Ntot, B, Ntot, K, loss_eps, den,
                                      terms, stats, ptr, ticket, means, transposed ? row_ptr : nullptr,
                                      transposed ? col : nullptr, transposed ? w : nullptr, stream);
  if (mode == 2)
    return tgp_diffpool_unbatched_tail_f32(raw, gram, B, K, sw2_dev, sw2_host, ent_partial, n_partial, link_scale,
                                           ent_scale, dstats, out2, stream);
  return TGP_OK;
}

// The backward of the same step as ONE native call, for the common case: the selector folded in (single Linear) and A = A^T
// (so T = A S serves as both U and V).  Strings together tgp_postprocess_dense_bwd_f32, tgp_dense_pool_train_rhs_f32,
// tgp_copy_cols3_f32, tgp_segment_gemm_nn_ld_f32 (gS, then gX), tgp_softmax_bwd_ex_f32, tgp_segment_gemm_tn_ld_f32 and
// tgp_slab_sum_split_f32 -- the launches functions._PoolUnbatchedFn.backward makes one by one (eight wrappers, ~110 us of
// host time in front of ~150 us of kernels).  Layout of the operand buffer acat [Ntot, 3K+F+4] = [T | X | 1 0 0 0 | S | dY]
// and of rcat [B, 3K+F+4, K]: DESIGN.md section 4.5.  Optional outputs are NULL when not wanted.
extern "C" int tgp_pool_rows_bwd_f32(const float* S, const float* T, const float* X, const float* W, const float* raw,
                                     const float* gram, const float* stats, const float* den, const float* deg,
                                     const float* link_loss, const int64_t* ptr, const int64_t* batch,
                                     const int64_t* slab_ptr, int64_t slabs, int64_t Ntot, int64_t B, int64_t K, int64_t F,
                                     int64_t max_nodes, int post_flags, float eps, float loss_eps, int mode, int transposed,
                                     float inv_b, float link_scale, float ent_scale, const float* g_adj, int g_adj_bcast,
                                     const float* g_raw, const float* g_x, int g_x_bcast, const float* g_s,
                                     const float* g_la, const float* g_lb, float* ga, float* rcat, float* c1, float* gwcat,
                                     float* acat, float* gS, float* gX, float* part, float* gW, float* gbias,
                                     void* stream) {
  TGP_REQUIRE(Ntot >= 1 && B >= 1 && K >= 1 && F >= 1 && slabs >= 1 && mode >= 0 && mode <= 2, TGP_ERR_INVALID,
              "tgp_pool_rows_bwd_f32: bad shape or mode");
  TGP_REQUIRE(S && T && X && raw && ptr && slab_ptr && rcat && acat && gS && (!gX || (W && gwcat)), TGP_ERR_INVALID,
              "tgp_pool_rows_bwd_f32: null pointer");
  TGP_REQUIRE(!g_adj || ga, TGP_ERR_INVALID, "tgp_pool_rows_bwd_f32: g_adj needs the ga buffer");
  TGP_REQUIRE(mode != 1 || (stats && den && gram && c1 && deg), TGP_ERR_INVALID, "tgp_pool_rows_bwd_f32: MinCut operands");
  TGP_REQUIRE(mode != 2 || (gram && link_loss), TGP_ERR_INVALID, "tgp_pool_rows_bwd_f32: DiffPool operands");
  TGP_REQUIRE((!gW && !gbias) || part, TGP_ERR_INVALID, "tgp_pool_rows_bwd_f32: the weight gradient needs the slab buffer");
  const int64_t pad = 4, ld = 3 * K + F + pad;
  const int64_t c_x = K, c_one = K + F, c_s = K + F + pad, c_v = 2 * K + F + pad;
  int rc = TGP_OK;
  if (g_adj) {
    rc = tgp_postprocess_dense_bwd_f32(raw, g_adj, B, K, post_flags | (g_adj_bcast ? (1 << 16) : 0), eps, ga, stream);
    if (rc != TGP_OK) return rc;
  }
  // A = A^T: bit 0 (the first block's right-hand side becomes gR + gR^T, the V rows are not written)
  rc = tgp_dense_pool_train_rhs_f32(g_adj ? ga : nullptr, g_raw, mode, mode == 1 ? stats : nullptr, mode == 1 ? den : nullptr,
                                    mode ? gram : nullptr, mode ? g_la : nullptr, mode == 1 ? g_lb : nullptr, inv_b,
                                    mode == 2 ? link_loss : nullptr, mode == 2 ? link_scale : 0.f, loss_eps, g_x, g_x_bcast,
                                    1, gX ? W : nullptr, B, K, F, rcat, mode == 1 ? c1 : nullptr, gX ? gwcat : nullptr,
                                    stream);
  if (rc != TGP_OK) return rc;
  rc = tgp_copy_cols3_f32(T, K, X, F, S, K, Ntot, acat, ld, 0, c_x, c_s, c_one, stream);
  if (rc != TGP_OK) return rc;
  rc = tgp_segment_gemm_nn_ld_f32(acat, ld, rcat, K, ld * K, ptr, gS, K, B, Ntot, c_v, K, max_nodes, stream);
  if (rc != TGP_OK) return rc;
  rc = tgp_softmax_bwd_ex_f32(S, gS, g_s, mode == 1 ? c1 : nullptr, mode == 1 ? deg : nullptr, Ntot,
                              mode == 2 ? g_lb : nullptr, mode == 2 ? ent_scale : 0.f, loss_eps, acat + c_v, ld, Ntot, K,
                              batch, stream);
  if (rc != TGP_OK) return rc;
  if (gX) {
    rc = tgp_segment_gemm_nn_ld_f32(acat + c_s, ld, gwcat, F, 2 * K * F, ptr, gX, F, B, Ntot, 2 * K, F, max_nodes, stream);
    if (rc != TGP_OK) return rc;
  }
  if (gW || gbias) {
    rc = tgp_segment_gemm_tn_ld_f32(acat + c_v, ld, acat + c_x, ld, slab_ptr, part, slabs, Ntot, K, F + pad, stream);
    if (rc != TGP_OK) return rc;
    rc = tgp_slab_sum_split_f32(part, slabs, K, F, F + pad, gW, gbias, stream);
  }
  return rc;
}
