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
