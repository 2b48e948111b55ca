// A14: GraclusSelect's matching (reference select/graclus_select.py:62-81 calls torch_cluster 1.6.3
// `graclus_cluster`, a third-party randomised greedy matching: every node is paired with its heaviest
// unmatched neighbour).  The exact pairs are not a contract (torch_cluster's own result depends on a random
// node permutation); what is: a maximal matching that prefers heavy edges, labelled with the smaller id of
// the pair.  This file is the build's own data-parallel "handshake" version of that algorithm:
//
//   every round, each free node proposes to the free neighbour with the largest edge key
//        key(i,j) = (w_ij, hash(min(i,j), max(i,j)), min, max)        -- a strict total order on edges
//   and mutual proposals are matched.  The largest live edge is always a mutual proposal, so every round
//   makes progress; the symmetric hash breaks weight ties pseudo-randomly, which is what keeps the number of
//   rounds logarithmic (smallest-id tie-breaking degenerates to one match per round on a path).
//
// Layout: the edge list is grouped by source node through the inverted index that tgp_assign_index_build makes
// of `row` (any edge order), gathered once into CSR arrays (int32 neighbour, fp32 weight); per-node state is one
// byte (free) + one int32 (proposal) + the int64 label.  A round streams the CSR once: HBM-bound, E * 8 bytes.
#include "common.h"

namespace tgp {

__device__ __forceinline__ uint32_t pair_hash(uint32_t a, uint32_t b) {  // symmetric by construction (a <= b)
  uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}

__global__ __launch_bounds__(256) void gm_csr_gather_kernel(const int64_t* __restrict__ col,
                                                            const float* __restrict__ w,
                                                            const int32_t* __restrict__ perm, int64_t E,
                                                            int32_t* __restrict__ nbr, float* __restrict__ wt) {
  const int64_t p = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (p >= E) return;
  const int64_t e = perm ? perm[p] : p;  // perm == NULL: the list is already grouped by source
  nbr[p] = static_cast<int32_t>(col[e]);
  wt[p] = w ? w[e] : 1.0f;
}

__global__ __launch_bounds__(256) void gm_init_kernel(int64_t n, int64_t* __restrict__ label,
                                                      uint8_t* __restrict__ is_free) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) {
    label[i] = i;
    is_free[i] = 1;
  }
}

// cand[i] = best free neighbour of free node i, or -1.
__global__ __launch_bounds__(256) void gm_propose_kernel(const int32_t* __restrict__ row_ptr,
                                                         const int32_t* __restrict__ nbr,
                                                         const float* __restrict__ wt, int64_t n,
                                                         uint8_t* __restrict__ is_free,
                                                         int32_t* __restrict__ cand) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  int32_t best = -1;
  if (is_free[i]) {
    float bw = 0.f;
    uint32_t bh = 0;
    const int32_t lo = row_ptr[i], hi = row_ptr[i + 1];
    // A wave is as slow as its slowest lane and a lane's scan is a chain of dependent loads (neighbour id ->
    // its free flag), so the neighbours are taken eight at a time: ids first, then flags and weights together.
    constexpr int U = 8;
    for (int32_t p0 = lo; p0 < hi; p0 += U) {
      int32_t js[U];
      float ws[U];
      uint8_t fs[U];
#pragma unroll
      for (int u = 0; u < U; ++u) js[u] = p0 + u < hi ? nbr[p0 + u] : -1;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        fs[u] = js[u] >= 0 ? is_free[js[u]] : 0;
        ws[u] = js[u] >= 0 ? wt[p0 + u] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int32_t j = js[u];
        if (j < 0 || j == i || !fs[u]) continue;
        const float wj = ws[u];
        if (wj != wj) continue;  // NaN weights never match
        const uint32_t a = static_cast<uint32_t>(i < j ? i : j), b = static_cast<uint32_t>(i < j ? j : i);
        const uint32_t h = pair_hash(a, b);
        // lexicographic (w, hash, min, max); min/max only matter for hash collisions between different pairs
        bool better = best < 0 || wj > bw || (wj == bw && h > bh);
        if (!better && best >= 0 && wj == bw && h == bh && j != best) {
          const uint32_t ca = static_cast<uint32_t>(i < best ? i : best), cb = static_cast<uint32_t>(i < best ? best : i);
          better = a > ca || (a == ca && b > cb);
        }
        if (better) { best = j; bw = wj; bh = h; }
      }
    }
    // No free neighbour left: the free set only shrinks, so this node stays single -- retire it, later rounds
    // skip its scan.  (No free node is adjacent to it, so nobody's proposal depends on this flag.)
    if (best < 0) is_free[i] = 0;
  }
  cand[i] = best;
}

// Mutual proposals become pairs.  Each endpoint writes only its own slots; *matched is set when any pair formed.
__global__ __launch_bounds__(256) void gm_match_kernel(const int32_t* __restrict__ cand, int64_t n,
                                                       int64_t* __restrict__ label, uint8_t* __restrict__ is_free,
                                                       unsigned int* __restrict__ matched) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  bool hit = false;
  if (i < n) {
    const int32_t j = cand[i];
    if (j >= 0 && cand[j] == static_cast<int32_t>(i)) {
      label[i] = i < j ? i : j;
      is_free[i] = 0;
      hit = true;
    }
  }
  // "did this round match anything": a plain store of 1 (one atomic per wave onto a single counter serialised
  // into 150 us at 1M nodes)
  if (__ballot(hit) && lane_id() == 0) *matched = 1u;
}

}  // namespace tgp

using namespace tgp;

extern "C" size_t tgp_graclus_match_workspace_bytes(int64_t num_nodes, int64_t num_edges) {
  const size_t n = static_cast<size_t>(num_nodes > 0 ? num_nodes : 1), e = static_cast<size_t>(num_edges > 0 ? num_edges : 1);
  return align_up(e * sizeof(int32_t)) + align_up(e * sizeof(float)) + align_up(n) + align_up(n * sizeof(int32_t)) + 256;
}

// Start: gathers the CSR and resets the state.  Rounds: runs `rounds` propose/match rounds; matched[r] becomes 1 if
// round r matched anything (device memory, uint32[rounds], zeroed here).  The host reads the last
// entries to decide whether to run more rounds (0 = the matching is maximal).
extern "C" int tgp_graclus_match_start(const int64_t* col, const float* weight, const int32_t* row_ptr,
                                       const int32_t* perm, int64_t num_nodes, int64_t num_edges, void* ws,
                                       size_t ws_bytes, int64_t* label, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(num_nodes >= 0 && num_edges >= 0, TGP_ERR_INVALID, "tgp_graclus_match_start: negative size");
  if (num_nodes == 0) return TGP_OK;
  TGP_REQUIRE(num_nodes < (1ll << 31) && num_edges < (1ll << 31), TGP_ERR_RANGE, "tgp_graclus_match_start: too large");
  TGP_REQUIRE(label && row_ptr && (num_edges == 0 || col), TGP_ERR_INVALID,
              "tgp_graclus_match_start: null pointer");
  TGP_REQUIRE(ws && ws_bytes >= tgp_graclus_match_workspace_bytes(num_nodes, num_edges), TGP_ERR_WORKSPACE,
              "tgp_graclus_match_start: workspace too small");
  Carver cv(ws);
  int32_t* nbr = cv.take<int32_t>(num_edges > 0 ? num_edges : 1);
  float* wt = cv.take<float>(num_edges > 0 ? num_edges : 1);
  uint8_t* is_free = cv.take<uint8_t>(num_nodes);
  if (num_edges > 0)
    hipLaunchKernelGGL(gm_csr_gather_kernel, dim3(cdiv(num_edges, 256)), dim3(256), 0, stream, col, weight, perm,
                       num_edges, nbr, wt);
  hipLaunchKernelGGL(gm_init_kernel, dim3(cdiv(num_nodes, 256)), dim3(256), 0, stream, num_nodes, label, is_free);
  return check_launch("tgp_graclus_match_start");
}

extern "C" int tgp_graclus_match_rounds(const int32_t* row_ptr, int64_t num_nodes, int64_t num_edges, void* ws,
                                        int rounds, unsigned int* matched, int64_t* label, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(num_nodes >= 0 && num_edges >= 0 && rounds >= 0, TGP_ERR_INVALID, "tgp_graclus_match_rounds: bad size");
  if (num_nodes == 0 || rounds == 0) return TGP_OK;
  TGP_REQUIRE(row_ptr && ws && matched && label, TGP_ERR_INVALID, "tgp_graclus_match_rounds: null pointer");
  Carver cv(ws);
  int32_t* nbr = cv.take<int32_t>(num_edges > 0 ? num_edges : 1);
  float* wt = cv.take<float>(num_edges > 0 ? num_edges : 1);
  uint8_t* is_free = cv.take<uint8_t>(num_nodes);
  int32_t* cand = cv.take<int32_t>(num_nodes);
  (void)hipMemsetAsync(matched, 0, static_cast<size_t>(rounds) * sizeof(unsigned int), stream);
  const int nb = cdiv(num_nodes, 256);
  for (int r = 0; r < rounds; ++r) {
    hipLaunchKernelGGL(gm_propose_kernel, dim3(nb), dim3(256), 0, stream, row_ptr, nbr, wt, num_nodes, is_free, cand);
    hipLaunchKernelGGL(gm_match_kernel, dim3(nb), dim3(256), 0, stream, cand, num_nodes, label, is_free, matched + r);
  }
  return check_launch("tgp_graclus_match_rounds");
}
