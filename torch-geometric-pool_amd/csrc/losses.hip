// N3: single-pass reductions behind the dense poolers' auxiliary losses.
//
//   entropy   sum(-S log(S + eps))                       utils/losses.py:476-483 (DiffPool)
//   cut terms deg_i = sum_j A[b,i,j],  q_i = sum_k S[b,i,k]^2,  den[b] = sum_i deg_i q_i = trace(S^T D S)
//                                                        utils/losses.py:39-81 (MinCut)
//
// All HBM-bound: every input byte is read once, reductions run in a fixed order (no float atomics).
#include "common.h"

namespace tgp {

constexpr int RED_BLOCKS = 1024;
typedef float nt_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

template <int T>
__device__ __forceinline__ float block_sum_t(float v, float* sh) {  // sh: T / 64 floats; fixed order
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < T / 64; ++w) t += sh[w];
  return t;
}

// four sums behind ONE pair of barriers (sh: 4 * T / 64 floats); each sum in the order block_sum_t adds it
template <int T>
__device__ __forceinline__ void block_sum4_t(float (&v)[4], float* sh) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[q] += __shfl_xor(v[q], o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) sh[q * (T / 64) + (threadIdx.x >> 6)] = v[q];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < T / 64; ++w) t += sh[q * (T / 64) + w];
    v[q] = t;
  }
}

__device__ __forceinline__ float ent_term(float s, float eps) { return -s * logf(s + eps); }

__global__ __launch_bounds__(256) void entropy_partial_kernel(const float* __restrict__ S, int64_t n, float eps,
                                                              float* __restrict__ partial) {
  __shared__ float sh[4];
  float acc = 0.f;
  const int64_t n4 = (reinterpret_cast<uintptr_t>(S) % 16 == 0) ? n / 4 : 0;
  const nt_f32x4* S4 = reinterpret_cast<const nt_f32x4*>(S);
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n4; i += 256ll * gridDim.x) {
    const nt_f32x4 v = __builtin_nontemporal_load(S4 + i);
    acc += (ent_term(v.x, eps) + ent_term(v.y, eps)) + (ent_term(v.z, eps) + ent_term(v.w, eps));
  }
  for (int64_t i = n4 * 4 + static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += 256ll * gridDim.x)
    acc += ent_term(S[i], eps);
  const float t = block_sum_256(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void final_sum_kernel(const float* __restrict__ partial, int n,
                                                        float* __restrict__ out) {
  __shared__ float sh[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
  const float t = block_sum_256(acc, sh);
  if (threadIdx.x == 0) out[0] = t;
}

// G lanes per node row: deg (row sum of A), q (squared norm of the S row); 256 / G rows per workgroup.  G = 64 for
// long rows; batches of small graphs (N <= 64: a row is at most 16 float4) use G = 16, i.e. four rows per wave.
template <int G>
__global__ __launch_bounds__(256) void cut_rows_kernel(const float* __restrict__ A, const float* __restrict__ S,
                                                       int64_t rows, int N, int K,
                                                       const int64_t* __restrict__ sizes, float* __restrict__ deg,
                                                       float* __restrict__ q) {
  const int sub = threadIdx.x % G;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * (256 / G) + threadIdx.x / G;
  float d = 0.f, qq = 0.f;
  // padded rows (beyond the graph's real size) are zero in A and S: nothing to read
  const bool real = row < rows && (!sizes || row % N < sizes[row / N]);
  if (real) {
    const float* a = A + row * N;
    const float* s = S + row * K;
    if ((N & 3) == 0 && reinterpret_cast<uintptr_t>(A) % 16 == 0) {
      const nt_f32x4* a4 = reinterpret_cast<const nt_f32x4*>(a);
      for (int j = sub; j < N / 4; j += G) {
        const nt_f32x4 v = __builtin_nontemporal_load(a4 + j);
        d += (v.x + v.y) + (v.z + v.w);
      }
    } else {
      for (int j = sub; j < N; j += G) d += a[j];
    }
    for (int k = sub; k < K; k += G) qq = fmaf(s[k], s[k], qq);
  }
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) {
    d += __shfl_xor(d, o, 64);
    qq += __shfl_xor(qq, o, 64);
  }
  if (row < rows && sub == 0) {
    deg[row] = d;
    q[row] = qq;
  }
}

__global__ __launch_bounds__(256) void cut_den_kernel(const float* __restrict__ deg, const float* __restrict__ q,
                                                      int N, float* __restrict__ den) {
  __shared__ float sh[4];
  const float* d = deg + static_cast<int64_t>(blockIdx.x) * N;
  const float* qq = q + static_cast<int64_t>(blockIdx.x) * N;
  float acc = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) acc = fmaf(d[i], qq[i], acc);
  const float t = block_sum_256(acc, sh);
  if (threadIdx.x == 0) den[blockIdx.x] = t;
}


// Per-graph tails of MinCut's two losses in ONE launch (utils/losses.py:39-56 and :59-70):
//   cut[b]   = -trace(raw[b]) / (den[b] + eps)                    raw = S^T A S, den = trace(S^T D S)
//   ortho[b] = || G_b / ||G_b||_F - I / sqrt(K) ||_F              G = S^T S
// (as torch ops these are ~14 launches of a few hundred bytes each).  One workgroup per graph; out[0,b], out[1,b].
template <int T>  // threads: 1024 for K >= 64 (few graphs, K^2 elements each: the 256-thread form took 28 us at B = 32, K = 128)
__global__ __launch_bounds__(T) void mincut_tail_kernel(const float* __restrict__ raw, const float* __restrict__ den,
                                                          const float* __restrict__ gram, int K, float eps,
                                                          int B, float* __restrict__ out) {
  __shared__ float sh[T / 64];
  const int b = blockIdx.x;
  const float* R = raw + static_cast<int64_t>(b) * K * K;
  const float* G = gram + static_cast<int64_t>(b) * K * K;
  float tr = 0.f, sq = 0.f;
  for (int i = threadIdx.x; i < K; i += T) tr += R[static_cast<int64_t>(i) * K + i];
  for (int i = threadIdx.x; i < K * K; i += T) sq = fmaf(G[i], G[i], sq);
  tr = block_sum_t<T>(tr, sh);
  sq = block_sum_t<T>(sq, sh);
  const float n = sqrtf(sq);
  const float t = 1.0f / sqrtf(static_cast<float>(K));
  float acc = 0.f;
  for (int i = threadIdx.x; i < K * K; i += T) {
    const float y = G[i] / n - ((i / K == i % K) ? t : 0.f);
    acc = fmaf(y, y, acc);
  }
  acc = block_sum_t<T>(acc, sh);
  if (threadIdx.x == 0) {
    out[b] = -(tr / (den[b] + eps));
    out[B + b] = sqrtf(acc);
  }
}

// Backward of those two tails, one launch (what autograd derives from utils/losses.py:39-70 behind S^T A S, trace(S^T D S)
// and S^T S: ~25 launches of a few hundred bytes each).  Given the upstream gradients g[0,b], g[1,b] of the two terms:
//   g_raw[b] = -(g_cut / (den + eps)) I                    (gradient with respect to raw = S^T A S)
//   c1[b]    = g_cut * trace(raw) / (den + eps)^2          (gradient with respect to den; dden/dS = 2 D S)
//   W[b]     = d ortho / d G = g_ortho / (|Y| |G|) (Y - G <G,Y> / |G|^2),  Y = G / |G| - I / sqrt(K)   (dS = S (W + W^T))
template <int T>
__global__ __launch_bounds__(T) void mincut_tail_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ den,
                                                              const float* __restrict__ gram, const float* __restrict__ g,
                                                              int K, float eps, int B, float* __restrict__ g_raw,
                                                              float* __restrict__ c1, float* __restrict__ W) {
  __shared__ float sh[T / 64];
  const int b = blockIdx.x;
  const int64_t off = static_cast<int64_t>(b) * K * K;
  const float* R = raw + off;
  const float* G = gram + off;
  float tr = 0.f, sq = 0.f;
  for (int i = threadIdx.x; i < K; i += T) tr += R[static_cast<int64_t>(i) * K + i];
  for (int i = threadIdx.x; i < K * K; i += T) sq = fmaf(G[i], G[i], sq);
  tr = block_sum_t<T>(tr, sh);
  sq = block_sum_t<T>(sq, sh);
  const float n = sqrtf(sq);
  const float t = 1.0f / sqrtf(static_cast<float>(K));
  float ny2 = 0.f, gy = 0.f;
  for (int i = threadIdx.x; i < K * K; i += T) {
    const float y = G[i] / n - ((i / K == i % K) ? t : 0.f);
    ny2 = fmaf(y, y, ny2);
    gy = fmaf(G[i], y, gy);
  }
  ny2 = block_sum_t<T>(ny2, sh);
  gy = block_sum_t<T>(gy, sh);
  const float ny = sqrtf(ny2);
  const float g_cut = g[b], g_ortho = g[B + b];
  const float dd = den[b] + eps;
  const float cdiag = -g_cut / dd;
  const float coef = ny > 0.f ? g_ortho / (ny * n) : 0.f;
  for (int i = threadIdx.x; i < K * K; i += T) {
    const bool diag = i / K == i % K;
    const float y = G[i] / n - (diag ? t : 0.f);
    W[off + i] = coef * (y - G[i] * (gy / sq));
    g_raw[off + i] = diag ? cdiag : 0.f;
  }
  if (threadIdx.x == 0) c1[b] = g_cut * tr / (dd * dd);
}

// DiffPool's two losses from their native partial results in ONE launch (poolers/diffpool.py:262-284):
//   out[0] = sqrt(sum_b sq[b]) * link_scale     (link_scale = link_loss_coeff, / adj.numel() when normalize_loss)
//   out[1] = (sum of the entropy partial sums) * ent_scale   (ent_scale = ent_loss_coeff / num_nodes)
// (as torch ops behind the kernels: sum, sqrt, two multiplications, a final-sum kernel and a division)
__global__ __launch_bounds__(256) void diffpool_tail_kernel(const float* __restrict__ sq, int B,
                                                            const float* __restrict__ ent_partial, int n_partial,
                                                            float link_scale, float ent_scale,
                                                            float* __restrict__ out) {
  __shared__ float sh[4];
  float a = 0.f, e = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) a += sq[i];
  for (int i = threadIdx.x; i < n_partial; i += 256) e += ent_partial[i];
  a = block_sum_256(a, sh);
  e = block_sum_256(e, sh);
  if (threadIdx.x == 0) {
    out[0] = sqrtf(a) * link_scale;
    out[1] = e * ent_scale;
  }
}

// ss[e] = <S[row_e,:], S[col_e,:]>: the per-edge entries of S S^T that the sparse (unbatched) losses need
// (utils/losses.py:73-127 sparse_mincut_loss, :661-708 sparse_link_pred_loss compute (S[src] * S[dst]).sum(-1),
// which materialises two [E,K] gathers and their product).  G lanes share an edge (float4 each when VEC),
// so a wave covers 64/G edges per step; S rows are re-read through L2 (S is N*K*4 bytes, far smaller than
// the 2*E*K*4 bytes of the gathers).
template <int G, bool VEC>
__global__ __launch_bounds__(256) void edge_dot_kernel(const int64_t* __restrict__ row,
                                                       const int64_t* __restrict__ col, int64_t E,
                                                       const float* __restrict__ S,
                                                       const float* __restrict__ S2, int K,
                                                       float* __restrict__ out) {
  constexpr int PER_WAVE = 64 / G;
  const int lane = threadIdx.x & 63, sub = lane % G, slot = lane / G;
  const int64_t wave = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) >> 6;
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * 4;
  for (int64_t base = wave * PER_WAVE; base < E; base += nwaves * PER_WAVE) {
    const int64_t e = base + slot;
    float acc = 0.f;
    if (e < E) {
      const float* a = S + row[e] * K;
      const float* b = S2 + col[e] * K;
      if constexpr (VEC) {
        for (int k = sub * 4; k < K; k += G * 4) {
          const float4 x = *reinterpret_cast<const float4*>(a + k);
          const float4 y = *reinterpret_cast<const float4*>(b + k);
          acc = fmaf(x.x, y.x, acc); acc = fmaf(x.y, y.y, acc);
          acc = fmaf(x.z, y.z, acc); acc = fmaf(x.w, y.w, acc);
        }
      } else {
        for (int k = sub; k < K; k += G) acc = fmaf(a[k], b[k], acc);
      }
    }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (e < E && sub == 0) out[e] = acc;
  }
}

template <int G, bool VEC>
static void launch_edge_dot(const int64_t* row, const int64_t* col, int64_t E, const float* S, const float* S2, int K,
                            float* out, hipStream_t stream) {
  int64_t blocks = cdiv(E, static_cast<int64_t>(4) * (64 / G));
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL((edge_dot_kernel<G, VEC>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, row, col, E,
                     S, S2, K, out);
}

}  // namespace tgp

using namespace tgp;

extern "C" size_t tgp_entropy_sum_workspace_bytes(int64_t n) {
  (void)n;
  return RED_BLOCKS * sizeof(float);
}

// d/dS of sum(-S log(S + eps)), times the upstream gradient (a device scalar) and a host scale: one elementwise pass
// (utils/losses.py:476-483 under autograd: log, add, div, add, neg, mul as six launches)
namespace tgp {
__global__ __launch_bounds__(256) void entropy_bwd_kernel(const float* __restrict__ S, int64_t n, float eps,
                                                          const float* __restrict__ g, float scale,
                                                          float* __restrict__ out) {
  const float gs = g[0] * scale;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * 256) {
    const float s = S[i];
    out[i] = -(logf(s + eps) + s / (s + eps)) * gs;
  }
}
}  // namespace tgp

extern "C" int tgp_entropy_bwd_f32(const float* S, int64_t n, float eps, const float* g, float scale, float* out,
                                   void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(n >= 0, TGP_ERR_INVALID, "tgp_entropy_bwd_f32: negative size");
  if (n == 0) return TGP_OK;
  TGP_REQUIRE(S && g && out, TGP_ERR_INVALID, "tgp_entropy_bwd_f32: null pointer");
  int64_t blocks = (n + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(entropy_bwd_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, S, n, eps, g, scale,
                     out);
  return check_launch("tgp_entropy_bwd_f32");
}

extern "C" int tgp_entropy_sum_f32(const float* S, int64_t n, float eps, float* out, void* ws, size_t ws_bytes,
                                   void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(n >= 0, TGP_ERR_INVALID, "tgp_entropy_sum_f32: negative size");
  TGP_REQUIRE(out, TGP_ERR_INVALID, "tgp_entropy_sum_f32: null output");
  if (n == 0) {
    (void)hipMemsetAsync(out, 0, sizeof(float), stream);
    return check_launch("tgp_entropy_sum_f32");
  }
  TGP_REQUIRE(S && ws, TGP_ERR_INVALID, "tgp_entropy_sum_f32: null pointer");
  TGP_REQUIRE(ws_bytes >= tgp_entropy_sum_workspace_bytes(n), TGP_ERR_WORKSPACE,
              "tgp_entropy_sum_f32: workspace too small");
  int blocks = cdiv(n, 256 * 4 * 4);
  if (blocks > RED_BLOCKS) blocks = RED_BLOCKS;
  if (blocks < 1) blocks = 1;
  float* partial = static_cast<float*>(ws);
  hipLaunchKernelGGL(entropy_partial_kernel, dim3(blocks), dim3(256), 0, stream, S, n, eps, partial);
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, stream, partial, blocks, out);
  return check_launch("tgp_entropy_sum_f32");
}

// entropy partial sums only (ws: RED_BLOCKS floats; *n_partial_out slots are written), for tgp_diffpool_loss_tail_f32
extern "C" int tgp_entropy_partials_f32(const float* S, int64_t n, float eps, void* ws, size_t ws_bytes,
                                        int* n_partial_out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(n >= 0 && ws && n_partial_out && (n == 0 || S), TGP_ERR_INVALID, "tgp_entropy_partials_f32: bad argument");
  TGP_REQUIRE(ws_bytes >= tgp_entropy_sum_workspace_bytes(n), TGP_ERR_WORKSPACE,
              "tgp_entropy_partials_f32: workspace too small");
  int blocks = cdiv(n, 256 * 4 * 4);
  if (blocks > RED_BLOCKS) blocks = RED_BLOCKS;
  if (blocks < 1) blocks = 1;
  *n_partial_out = blocks;
  hipLaunchKernelGGL(entropy_partial_kernel, dim3(blocks), dim3(256), 0, stream, S, n, eps, static_cast<float*>(ws));
  return check_launch("tgp_entropy_partials_f32");
}

extern "C" int tgp_diffpool_loss_tail_f32(const float* sq, int64_t B, const float* ent_partial, int n_partial,
                                          float link_scale, float ent_scale, float* out2, void* stream_) {
  TGP_REQUIRE(B >= 0 && B < (1ll << 31) && n_partial >= 0 && out2 && (B == 0 || sq) && (n_partial == 0 || ent_partial),
              TGP_ERR_INVALID, "tgp_diffpool_loss_tail_f32: bad argument");
  hipLaunchKernelGGL(diffpool_tail_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream_), sq,
                     static_cast<int>(B), ent_partial, n_partial, link_scale, ent_scale, out2);
  return check_launch("tgp_diffpool_loss_tail_f32");
}

extern "C" int tgp_cut_terms_f32(const float* A, const float* S, int64_t B, int64_t N, int64_t K,
                                 const int64_t* graph_sizes, float* deg, float* q, float* den, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(B >= 0 && N >= 0 && K >= 0, TGP_ERR_INVALID, "tgp_cut_terms_f32: negative size");
  if (B == 0) return TGP_OK;
  TGP_REQUIRE(den || (deg && q), TGP_ERR_INVALID, "tgp_cut_terms_f32: null output");
  if (N == 0) {
    if (den) (void)hipMemsetAsync(den, 0, B * sizeof(float), stream);
    return check_launch("tgp_cut_terms_f32");
  }
  TGP_REQUIRE(A && deg && q && (K == 0 || S), TGP_ERR_INVALID, "tgp_cut_terms_f32: null pointer");
  TGP_REQUIRE(N < (1ll << 31) && K < (1ll << 31) && B * N < (1ll << 33), TGP_ERR_RANGE, "tgp_cut_terms_f32: too large");
  const int64_t rows = B * N;
  if (N <= 64)
    hipLaunchKernelGGL(cut_rows_kernel<16>, dim3(cdiv(rows, 16)), dim3(256), 0, stream, A, S, rows, static_cast<int>(N),
                       static_cast<int>(K), graph_sizes, deg, q);
  else
    hipLaunchKernelGGL(cut_rows_kernel<64>, dim3(cdiv(rows, 4)), dim3(256), 0, stream, A, S, rows, static_cast<int>(N),
                       static_cast<int>(K), graph_sizes, deg, q);
  if (den)  // (r6: NULL = the caller's fused loss tail forms the dot product itself)
    hipLaunchKernelGGL(cut_den_kernel, dim3(static_cast<unsigned>(B)), dim3(256), 0, stream, deg, q, static_cast<int>(N), den);
  return check_launch("tgp_cut_terms_f32");
}

extern "C" int tgp_mincut_loss_terms_f32(const float* raw, const float* den, const float* gram, int64_t B, int64_t K,
                                         float eps, float* out, void* stream_) {
  TGP_REQUIRE(B >= 0 && K >= 1 && K < 32768, TGP_ERR_INVALID, "tgp_mincut_loss_terms_f32: bad shape");
  if (B == 0) return TGP_OK;
  TGP_REQUIRE(raw && den && gram && out, TGP_ERR_INVALID, "tgp_mincut_loss_terms_f32: null pointer");
  TGP_REQUIRE(B < (1ll << 31), TGP_ERR_RANGE, "tgp_mincut_loss_terms_f32: too many graphs");
  if (K >= 64)
    hipLaunchKernelGGL(mincut_tail_kernel<1024>, dim3(static_cast<unsigned>(B)), dim3(1024), 0,
                       static_cast<hipStream_t>(stream_), raw, den, gram, static_cast<int>(K), eps, static_cast<int>(B), out);
  else
    hipLaunchKernelGGL(mincut_tail_kernel<256>, dim3(static_cast<unsigned>(B)), dim3(256), 0,
                       static_cast<hipStream_t>(stream_), raw, den, gram, static_cast<int>(K), eps, static_cast<int>(B), out);
  return check_launch("tgp_mincut_loss_terms_f32");
}

extern "C" int tgp_mincut_loss_terms_bwd_f32(const float* raw, const float* den, const float* gram, const float* g_terms,
                                            int64_t B, int64_t K, float eps, float* g_raw, float* c1, float* W,
                                            void* stream_) {
  TGP_REQUIRE(B >= 0 && K >= 1 && K < 32768, TGP_ERR_INVALID, "tgp_mincut_loss_terms_bwd_f32: bad shape");
  if (B == 0) return TGP_OK;
  TGP_REQUIRE(raw && den && gram && g_terms && g_raw && c1 && W, TGP_ERR_INVALID,
              "tgp_mincut_loss_terms_bwd_f32: null pointer");
  TGP_REQUIRE(B < (1ll << 31), TGP_ERR_RANGE, "tgp_mincut_loss_terms_bwd_f32: too many graphs");
  if (K >= 64)
    hipLaunchKernelGGL(mincut_tail_bwd_kernel<1024>, dim3(static_cast<unsigned>(B)), dim3(1024), 0,
                       static_cast<hipStream_t>(stream_), raw, den, gram, g_terms, static_cast<int>(K), eps,
                       static_cast<int>(B), g_raw, c1, W);
  else
    hipLaunchKernelGGL(mincut_tail_bwd_kernel<256>, dim3(static_cast<unsigned>(B)), dim3(256), 0,
                       static_cast<hipStream_t>(stream_), raw, den, gram, g_terms, static_cast<int>(K), eps,
                       static_cast<int>(B), g_raw, c1, W);
  return check_launch("tgp_mincut_loss_terms_bwd_f32");
}

// ---- r6: the loss tails of the dense poolers' training step at C2 scale (one autograd node, functions._PoolLargeFn) ---
// MinCut's two per-graph terms with den = trace(S^T D S) = sum_i deg_i q_i formed in the same launch (the forward had a
// kernel of its own for that dot product); den is kept for the backward.
namespace tgp {
template <int T>
__global__ __launch_bounds__(T) void mincut_tail2_kernel(const float* __restrict__ raw, const float* __restrict__ gram,
                                                           const float* __restrict__ deg, const float* __restrict__ q,
                                                           int N, int K, float eps, int B, float* __restrict__ den,
                                                           float* __restrict__ out, float* __restrict__ stats,
                                                           const int64_t* __restrict__ ptr,
                                                           unsigned int* __restrict__ ticket, float* __restrict__ means,
                                                           const int* __restrict__ erp, const int64_t* __restrict__ ecol,
                                                           const float* __restrict__ ew) {
  __shared__ float sh[4 * (T / 64)];
  __shared__ bool s_last;
  const int b = blockIdx.x;
  const float* R = raw + static_cast<int64_t>(b) * K * K;
  const float* G = gram + static_cast<int64_t>(b) * K * K;
  // padded batch: graph b owns entries b N .. b N + N of deg / q; un-padded (ptr): entries ptr[b] .. ptr[b + 1]
  const int64_t lo = ptr ? ptr[b] : static_cast<int64_t>(b) * N;
  if (ptr) N = static_cast<int>(ptr[b + 1] - lo);
  const float* d = deg + lo;
  const float* qq = q + lo;
  float dn = 0.f, tr = 0.f, sq = 0.f, trg = 0.f;
  // r6 (late): the Gram matrix stays in registers between its two sweeps when it fits (K <= 128 at 1024 threads), all
  // its loads requested before the first is used, and the four sums share one pair of barriers (18 -> 15 us at B = 32,
  // K = 128: the kernel is a chain of dependent round trips and barriers on B workgroups); same adds in the same order
  constexpr int GV = 16;
  const int KK = K * K;
  const bool inreg = KK <= GV * T;
  float gv[GV];
  if (inreg) {
#pragma unroll
    for (int j = 0; j < GV; ++j) {
      const int i = threadIdx.x + j * T;
      gv[j] = i < KK ? G[i] : 0.f;
    }
  }
  if (erp) {
    // in-degree form (the batched poolers' S^T A^T S on the rows route): den = sum_j indeg_j q_j = sum over the graph's
    // entries of w_e q[col_e], straight from the CSR list (the graph's entries are rows lo .. lo + N of it) -- the SpMV
    // A q this replaces was a launch of its own
    const int e_lo = erp[lo], e_hi = erp[lo + N];
    for (int e = e_lo + threadIdx.x; e < e_hi; e += T) dn = fmaf(ew ? ew[e] : 1.0f, q[ecol[e]], dn);
  } else if (q) {
    for (int i = threadIdx.x; i < N; i += T) dn = fmaf(d[i], qq[i], dn);
  } else {
    for (int i = threadIdx.x; i < N; i += T) dn += d[i];  // (deg already carries the factor: (A q)_i)
  }
  for (int i = threadIdx.x; i < K; i += T) {
    tr += R[static_cast<int64_t>(i) * K + i];
    trg += G[static_cast<int64_t>(i) * K + i];
  }
  if (inreg) {
#pragma unroll
    for (int j = 0; j < GV; ++j)
      if (threadIdx.x + j * T < KK) sq = fmaf(gv[j], gv[j], sq);
  } else {
    for (int i = threadIdx.x; i < KK; i += T) sq = fmaf(G[i], G[i], sq);
  }
  {
    float four[4] = {dn, tr, trg, sq};
    block_sum4_t<T>(four, sh);
    dn = four[0]; tr = four[1]; trg = four[2]; sq = four[3];
  }
  const float n = sqrtf(sq);
  const float t = 1.0f / sqrtf(static_cast<float>(K));
  float acc = 0.f;
  if (inreg) {
#pragma unroll
    for (int j = 0; j < GV; ++j) {
      const int i = threadIdx.x + j * T;
      if (i < KK) {
        const float y = gv[j] / n - ((i / K == i % K) ? t : 0.f);
        acc = fmaf(y, y, acc);
      }
    }
  } else {
    for (int i = threadIdx.x; i < KK; i += T) {
      const float y = G[i] / n - ((i / K == i % K) ? t : 0.f);
      acc = fmaf(y, y, acc);
    }
  }
  acc = block_sum_t<T>(acc, sh);
  if (threadIdx.x == 0) {
    den[b] = dn;
    out[b] = -(tr / (dn + eps));
    out[B + b] = sqrtf(acc);
    if (stats) {  // what the backward's right-hand sides need of this graph, so that they need no reduction of their own
      stats[4 * b] = tr; stats[4 * b + 1] = sq; stats[4 * b + 2] = trg; stats[4 * b + 3] = sqrtf(acc);
    }
  }
  if (!means) return;
  // the batch means of the two terms (what MinCutPooling hands out) by the workgroup that arrives last: every term
  // added in graph order by one workgroup, so the value does not depend on the arrival order (a separate reduction
  // launch + two selects before)
  if (threadIdx.x == 0) {
    __threadfence();
    s_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  float a0 = 0.f, a1 = 0.f;
  for (int i = threadIdx.x; i < B; i += T) {
    a0 += __hip_atomic_load(out + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a1 += __hip_atomic_load(out + B + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  a0 = block_sum_t<T>(a0, sh);
  a1 = block_sum_t<T>(a1, sh);
  if (threadIdx.x == 0) {
    means[0] = a0 / static_cast<float>(B);
    means[1] = a1 / static_cast<float>(B);
    *ticket = 0;  // ready for the next call on this stream
  }
}

// The K-sized right-hand sides of the backward's ONE sum-of-products GEMM
//     gS = [U | X | 1000 | S | V] [RU ; RX ; 0 ; RS ; RV],     U = A S, V = A^T S
// in 32 x 32 tiles.  With gR = the total gradient of raw = S^T A S (post-processing backward `ga` + an upstream
// gradient of raw `gb` + the loss' own diagonal term):
//     RV = gR,  RU = gR^T                      (dense_conn.py:111-122 under autograd: dS = U dR^T + V dR)
//     RX = g_x^T                               (base_reduce.py:158-161: dS = X dX'^T)
//     RS = W + W^T (MinCut, W = d ortho / d G, losses.py:59-70)   or   2 c G (DiffPool, losses.py:644-658)   or 0
// `symmetric` (A = A^T, so V = U): RU = gR + gR^T and the RV rows are not written -- the caller multiplies the first
// 2K + F + 4 rows only and never forms V.
// mode 1 (MinCut): gR += -(g_cut / (den + eps)) I, c1[b] = g_cut trace(raw) / (den + eps)^2 (the gradient of den: the
// caller folds 2 c1 D S into the softmax backward); g_cut / g_ortho are the upstream gradients of the batch MEANS
// (0-dim tensors, or NULL) and `scale` = 1 / B.  mode 2 (DiffPool): c = link_scale^2 g_link / link_loss (0 when the
// loss is 0: torch.norm's subgradient), gR += -c I, RS = 2 c G.  rcat: [B][3K + F + 4][K]; the four zero rows face the
// operand buffer's [1 0 0 0] column block, which gives the selector's bias gradient from the same product as its weight
// gradient.  gw (optional, [B][2K][F]) = [g_x ; W]: the right-hand side of gX = [S | dY] [g_x ; W] (dY overwrites the
// V block, which lies behind S).
constexpr int TGP_TRAIN_PAD = 4;  // the [1 0 0 0] column block of the operand buffer (tgp_copy_cols2_f32 writes it)

struct TrainRhsArgs {
  const float* ga; const float* gb;
  const float* stats; const float* den; const float* gram;
  const float* g_la; const float* g_lb; float scale;
  const float* link_loss; float link_scale, eps;
  const float* g_x; int gx_bcast;
  int K, F, mode, tiles_k, tiles_f, symmetric;  // symmetric bit 1: swap the RU / RV slots (the first block holds V)
  float* rcat; float* c1;
  const float* W; float* gw;  // selector weight [K][F] and [B][2K][F] = [g_x ; W], or NULL
};

// 32 x 32 tiles, 256 threads; blockIdx.y = graph.  Tiles 0 .. tiles_k^2 - 1: tile (ti, tj) of RU / RV / RS (the
// transposed operands come through LDS patches of the mirror tile); the tiles behind them: RX = g_x^T.
// MinCut's scalars come from the forward's stats [B,4] = (trace(raw), |G|^2, trace(G), |Y|):
//   <G, Y> = |G|^2 / |G| - trace(G) / sqrt(K)   (Y = G / |G| - I / sqrt(K)): no reduction here.
__global__ __launch_bounds__(256) void train_rhs_kernel(TrainRhsArgs p) {
  __shared__ float t_a[32][33], t_g[32][33];
  const int b = blockIdx.y, K = p.K, F = p.F, tid = threadIdx.x;
  const int tx = tid & 31, ty = tid >> 5;  // ty: 0..7, four rows each
  const int64_t off = static_cast<int64_t>(b) * K * K;
  float* out = p.rcat + static_cast<int64_t>(b) * (3 * K + F + TGP_TRAIN_PAD) * K;
  const int kt = p.tiles_k;
  const int tile = blockIdx.x;
  if (tile >= kt * kt + p.tiles_f * kt) {  // gw[b] = [g_x[b] ; W], 1024 elements per workgroup
    const int64_t kf = static_cast<int64_t>(K) * F;
    const float* gx = p.g_x ? (p.gx_bcast ? p.g_x : p.g_x + static_cast<int64_t>(b) * kf) : nullptr;
    float* dst = p.gw + static_cast<int64_t>(b) * 2 * kf;
    const int64_t e0 = static_cast<int64_t>(tile - kt * kt - p.tiles_f * kt) * 1024;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t e = e0 + q * 256 + tid;
      if (e < kf) dst[e] = gx ? (p.gx_bcast ? gx[0] : gx[e]) : 0.f;
      else if (e < 2 * kf) dst[e] = p.W[e - kf];
    }
    return;
  }
  if (tile >= kt * kt) {  // RX[f][k] = g_x[k][f]
    const int id = tile - kt * kt;
    const int tf = id / kt, tk = id - tf * kt;
    float* RX = out + static_cast<int64_t>(K) * K;
    const float* gx = p.g_x ? (p.gx_bcast ? p.g_x : p.g_x + static_cast<int64_t>(b) * K * F) : nullptr;
    if (gx && !p.gx_bcast) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // patch[k][f], read along f
        const int k = tk * 32 + ty * 4 + q, f = tf * 32 + tx;
        t_a[ty * 4 + q][tx] = (k < K && f < F) ? gx[static_cast<int64_t>(k) * F + f] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f = tf * 32 + ty * 4 + q, k = tk * 32 + tx;
      if (f < F && k < K) RX[static_cast<int64_t>(f) * K + k] = gx ? (p.gx_bcast ? gx[0] : t_a[tx][ty * 4 + q]) : 0.f;
    }
    return;
  }
  const int ti = tile / kt, tj = tile - ti * kt;
  const float* ga = p.ga ? p.ga + off : nullptr;
  const float* gb = p.gb ? p.gb + off : nullptr;
  const float* G = p.gram ? p.gram + off : nullptr;
  float diag = 0.f, coef = 0.f, gq = 0.f, n = 1.f, two_c = 0.f;
  const float t = 1.0f / sqrtf(static_cast<float>(K));
  if (p.mode == 1) {
    const float tr = p.stats[4 * b], sq = p.stats[4 * b + 1], trg = p.stats[4 * b + 2], ny = p.stats[4 * b + 3];
    n = sqrtf(sq);
    const float g_cut = p.g_la ? p.g_la[0] * p.scale : 0.f, g_ortho = p.g_lb ? p.g_lb[0] * p.scale : 0.f;
    const float dd = p.den[b] + p.eps;
    diag = -g_cut / dd;
    coef = ny > 0.f ? g_ortho / (ny * n) : 0.f;
    gq = (sq / n - t * trg) / sq;  // <G, Y> / |G|^2
    if (tile == 0 && tid == 0 && p.c1) p.c1[b] = g_cut * tr / (dd * dd);
  } else if (p.mode == 2) {
    const float loss = p.link_loss ? p.link_loss[0] : 0.f;
    const float c = (p.g_la && loss > 0.f) ? p.link_scale * p.link_scale * p.g_la[0] / loss : 0.f;
    diag = -c;
    two_c = 2.0f * c;
  }
  // mirror tile (tj, ti) of gR (and of G for MinCut's W^T) into LDS, read along its rows
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = tj * 32 + ty * 4 + q, c = ti * 32 + tx;
    const bool ok = r < K && c < K;
    const int64_t e = static_cast<int64_t>(r) * K + c;
    float v = 0.f;
    if (ok && ga) v += ga[e];
    if (ok && gb) v += gb[e];
    t_a[ty * 4 + q][tx] = v;
    if (p.mode == 1) t_g[ty * 4 + q][tx] = ok ? G[e] : 0.f;
  }
  __syncthreads();
  float* RU = out;
  float* RS = out + static_cast<int64_t>(K + F + TGP_TRAIN_PAD) * K;
  float* RV = out + static_cast<int64_t>(2 * K + F + TGP_TRAIN_PAD) * K;
  if (tile == 0)
    for (int i = tid; i < TGP_TRAIN_PAD * K; i += 256) out[static_cast<int64_t>(K + F) * K + i] = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = ti * 32 + ty * 4 + q, c = tj * 32 + tx;
    if (r >= K || c >= K) continue;
    const int64_t e = static_cast<int64_t>(r) * K + c;
    float v = 0.f;
    if (ga) v += ga[e];
    if (gb) v += gb[e];
    const float dg = (r == c) ? diag : 0.f;
    const float vt = t_a[tx][ty * 4 + q];  // gR[c][r]
    if (p.symmetric & 1) {
      RU[e] = (v + dg) + (vt + dg);
    } else if (p.symmetric & 2) {  // the operand buffer's first block is A S where raw = S^T A^T S: the slots trade places
      RU[e] = v + dg;
      RV[e] = vt + dg;
    } else {
      RV[e] = v + dg;
      RU[e] = vt + dg;
    }
    if (p.mode == 0) {
      RS[e] = 0.f;
    } else if (p.mode == 1) {
      const float gij = G[e], gji = t_g[tx][ty * 4 + q];
      const float dt = (r == c) ? t : 0.f;
      RS[e] = coef * ((gij / n - dt) - gij * gq) + coef * ((gji / n - dt) - gji * gq);
    } else if (p.mode == 2) {
      RS[e] = two_c * G[e];
    }
  }
}
}  // namespace tgp

// ---- r6: the unbatched dense poolers' losses from the products their Connect forms anyway ---------------------------
// sparse_mincut_loss (utils/losses.py:73-127): num_g = sum_{e in g} w_e <S_row, S_col> = sum_{i in g} <S_i, (A S)_i>
//   = trace(S_g^T (A S)_g) = trace(raw_g);  den_g = sum_{i in g} deg_i |S_i|^2 with deg_i = sum of the weights of row i.
// sparse_link_pred_loss (losses.py:661-708): |A - S S^T|^2 = sum_e (w_e - ss_e)^2 + sum_g |G_g|^2 - sum_e ss_e^2
//   = sum_e w_e^2 - 2 sum_g trace(raw_g) + sum_g |G_g|^2   (ss_e = <S_row, S_col>; sum_e w_e ss_e = sum_g trace(raw_g)).
// So the per-edge dot products, the index_add_ scatters and the masks behind them (40-45 launches per forward, r5) are
// replaced by: one pass over the CSR rows (deg, q), the S^T [T | X | S] product, and a per-graph tail.
namespace tgp {
// deg[i] = sum of w over row i's entries (their count when w is NULL), q[i] = |S_i|^2; G lanes per row.
template <int G>
__global__ __launch_bounds__(256) void edge_row_stats_kernel(const int* __restrict__ row_ptr, const float* __restrict__ w,
                                                             const float* __restrict__ S, int64_t N, int K,
                                                             float* __restrict__ deg, float* __restrict__ q) {
  const int sub = threadIdx.x % G;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * (256 / G) + threadIdx.x / G;
  float d = 0.f, qq = 0.f;
  if (i < N) {
    const int e0 = row_ptr[i], e1 = row_ptr[i + 1];
    if (w) for (int e = e0 + sub; e < e1; e += G) d += w[e];
    else d = sub == 0 ? static_cast<float>(e1 - e0) : 0.f;
    const float* s = S + i * K;
    for (int k = sub; k < K; k += G) qq = fmaf(s[k], s[k], qq);
  }
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) {
    d += __shfl_xor(d, o, 64);
    qq += __shfl_xor(qq, o, 64);
  }
  if (i < N && sub == 0) {
    deg[i] = d;
    q[i] = qq;
  }
}

// per graph: stats[b] = (trace(raw_b), |G_b|_F^2)
template <int T>
__global__ __launch_bounds__(T) void graph_trace_gsq_kernel(const float* __restrict__ raw, const float* __restrict__ gram,
                                                            int K, float* __restrict__ stats) {
  __shared__ float sh[T / 64];
  const int b = blockIdx.x;
  const float* R = raw + static_cast<int64_t>(b) * K * K;
  const float* G = gram + static_cast<int64_t>(b) * K * K;
  float tr = 0.f, sq = 0.f;
  for (int i = threadIdx.x; i < K; i += T) tr += R[static_cast<int64_t>(i) * K + i];
  for (int i = threadIdx.x; i < K * K; i += T) sq = fmaf(G[i], G[i], sq);
  tr = block_sum_t<T>(tr, sh);
  sq = block_sum_t<T>(sq, sh);
  if (threadIdx.x == 0) { stats[2 * b] = tr; stats[2 * b + 1] = sq; }
}

// out[0] = sqrt(max(sw2 - 2 sum_b trace_b + sum_b gsq_b, 0)) * link_scale, out[1] = (sum of the entropy partials) * ent_scale
__global__ __launch_bounds__(256) void diffpool_u_final_kernel(const float* __restrict__ stats, int B,
                                                               const float* __restrict__ sw2_dev, float sw2_host,
                                                               const float* __restrict__ ent_partial, int n_partial,
                                                               float link_scale, float ent_scale, float* __restrict__ out) {
  __shared__ float sh[4];
  float tr = 0.f, gs = 0.f, e = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) { tr += stats[2 * i]; gs += stats[2 * i + 1]; }
  for (int i = threadIdx.x; i < n_partial; i += 256) e += ent_partial[i];
  tr = block_sum_256(tr, sh);
  gs = block_sum_256(gs, sh);
  e = block_sum_256(e, sh);
  if (threadIdx.x == 0) {
    const float sw2 = sw2_dev ? sw2_dev[0] : sw2_host;
    out[0] = sqrtf(fmaxf((sw2 - 2.0f * tr) + gs, 0.f)) * link_scale;
    out[1] = e * ent_scale;
  }
}
}  // namespace tgp

namespace tgp {
// out[0] = sqrt(max(sum_b (a2_b - 2 tr_b + fro_b), 0)) * link_scale, out[1] = (sum_b ent_b) * ent_scale from the [B,4]
// records of dense_pool_small_kernel (SmallArgs.diff_stats); one workgroup, sums in graph order per thread, fixed tree
__global__ __launch_bounds__(256) void diffpool_stats_tail_kernel(const float* __restrict__ stats, int B, float link_scale,
                                                                  float ent_scale, float* __restrict__ out) {
  __shared__ float sh[4];
  float a2 = 0.f, tr = 0.f, fro = 0.f, ent = 0.f;
  const float4* s4 = reinterpret_cast<const float4*>(stats);
  for (int i = threadIdx.x; i < B; i += 256) {
    const float4 v = s4[i];
    a2 += v.x; tr += v.y; fro += v.z; ent += v.w;
  }
  a2 = block_sum_256(a2, sh);
  tr = block_sum_256(tr, sh);
  fro = block_sum_256(fro, sh);
  ent = block_sum_256(ent, sh);
  if (threadIdx.x == 0) {
    out[0] = sqrtf(fmaxf((a2 - 2.0f * tr) + fro, 0.f)) * link_scale;
    out[1] = ent * ent_scale;
  }
}
}  // namespace tgp

extern "C" int tgp_diffpool_stats_tail_f32(const float* stats, int64_t B, float link_scale, float ent_scale, float* out2,
                                           void* stream_) {
  TGP_REQUIRE(B >= 1 && B < (1ll << 31), TGP_ERR_INVALID, "tgp_diffpool_stats_tail_f32: bad shape");
  TGP_REQUIRE(stats && out2 && reinterpret_cast<uintptr_t>(stats) % 16 == 0, TGP_ERR_INVALID,
              "tgp_diffpool_stats_tail_f32: null or misaligned pointer");
  hipLaunchKernelGGL(tgp::diffpool_stats_tail_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream_), stats,
                     static_cast<int>(B), link_scale, ent_scale, out2);
  return check_launch("tgp_diffpool_stats_tail_f32");
}

extern "C" int tgp_edge_row_stats_f32(const int32_t* row_ptr, const float* w, const float* S, int64_t N, int64_t K,
                                      float* deg, float* q, void* stream_) {
  TGP_REQUIRE(N >= 0 && K >= 0 && K < (1ll << 31), TGP_ERR_INVALID, "tgp_edge_row_stats_f32: bad shape");
  if (N == 0) return TGP_OK;
  TGP_REQUIRE(row_ptr && deg && q && (K == 0 || S), TGP_ERR_INVALID, "tgp_edge_row_stats_f32: null pointer");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (K <= 32)
    hipLaunchKernelGGL(edge_row_stats_kernel<8>, dim3(cdiv(N, 32)), dim3(256), 0, stream, row_ptr, w, S, N,
                       static_cast<int>(K), deg, q);
  else
    hipLaunchKernelGGL(edge_row_stats_kernel<32>, dim3(cdiv(N, 8)), dim3(256), 0, stream, row_ptr, w, S, N,
                       static_cast<int>(K), deg, q);
  return check_launch("tgp_edge_row_stats_f32");
}

extern "C" int tgp_diffpool_unbatched_tail_f32(const float* raw, const float* gram, int64_t B, int64_t K,
                                               const float* sw2_dev, float sw2_host, const float* ent_partial,
                                               int n_partial, float link_scale, float ent_scale, float* stats,
                                               float* out2, void* stream_) {
  TGP_REQUIRE(B >= 1 && B < (1ll << 31) && K >= 1 && K < 32768 && n_partial >= 0, TGP_ERR_INVALID,
              "tgp_diffpool_unbatched_tail_f32: bad shape");
  TGP_REQUIRE(raw && gram && stats && out2 && (n_partial == 0 || ent_partial), TGP_ERR_INVALID,
              "tgp_diffpool_unbatched_tail_f32: null pointer");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (K >= 64)
    hipLaunchKernelGGL(graph_trace_gsq_kernel<1024>, dim3(static_cast<unsigned>(B)), dim3(1024), 0, stream, raw, gram,
                       static_cast<int>(K), stats);
  else
    hipLaunchKernelGGL(graph_trace_gsq_kernel<256>, dim3(static_cast<unsigned>(B)), dim3(256), 0, stream, raw, gram,
                       static_cast<int>(K), stats);
  hipLaunchKernelGGL(diffpool_u_final_kernel, dim3(1), dim3(256), 0, stream, stats, static_cast<int>(B), sw2_dev, sw2_host,
                     ent_partial, n_partial, link_scale, ent_scale, out2);
  return check_launch("tgp_diffpool_unbatched_tail_f32");
}

extern "C" int tgp_mincut_terms_fused_f32(const float* raw, const float* gram, const float* deg, const float* q,
                                          int64_t B, int64_t N, int64_t K, float eps, float* den, float* out,
                                          float* stats, const int64_t* ptr, uint32_t* ticket, float* means,
                                          const int32_t* edge_row_ptr, const int64_t* edge_col, const float* edge_w,
                                          void* stream_) {
  TGP_REQUIRE(B >= 0 && N >= 0 && K >= 1 && K < 32768 && N < (1ll << 31), TGP_ERR_INVALID,
              "tgp_mincut_terms_fused_f32: bad shape");
  if (B == 0) return TGP_OK;
  TGP_REQUIRE(raw && gram && den && out && (N == 0 || deg || edge_row_ptr), TGP_ERR_INVALID,
              "tgp_mincut_terms_fused_f32: null pointer");
  TGP_REQUIRE(!edge_row_ptr || (ptr && q && edge_col), TGP_ERR_INVALID,
              "tgp_mincut_terms_fused_f32: the edge form needs ptr, q and the column array");
  TGP_REQUIRE(B < (1ll << 31), TGP_ERR_RANGE, "tgp_mincut_terms_fused_f32: too many graphs");
  TGP_REQUIRE(!means || ticket, TGP_ERR_INVALID, "tgp_mincut_terms_fused_f32: means need a zeroed ticket word");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (K >= 64)
    hipLaunchKernelGGL(mincut_tail2_kernel<1024>, dim3(static_cast<unsigned>(B)), dim3(1024), 0, stream, raw, gram, deg, q,
                       static_cast<int>(N), static_cast<int>(K), eps, static_cast<int>(B), den, out, stats, ptr, ticket, means,
                       edge_row_ptr, edge_col, edge_w);
  else
    hipLaunchKernelGGL(mincut_tail2_kernel<256>, dim3(static_cast<unsigned>(B)), dim3(256), 0, stream, raw, gram, deg, q,
                       static_cast<int>(N), static_cast<int>(K), eps, static_cast<int>(B), den, out, stats, ptr, ticket,
                       means, edge_row_ptr, edge_col, edge_w);
  return check_launch("tgp_mincut_terms_fused_f32");
}

extern "C" int tgp_dense_pool_train_rhs_f32(const float* g_raw_a, const float* g_raw_b, int mode, const float* stats,
                                            const float* den, const float* gram, const float* g_la, const float* g_lb,
                                            float scale, const float* link_loss, float link_scale, float eps,
                                            const float* g_x, int gx_bcast, int symmetric, const float* W, int64_t B,
                                            int64_t K, int64_t F, float* rcat, float* c1, float* gw, void* stream_) {
  TGP_REQUIRE(B >= 0 && K >= 1 && K < 32768 && F >= 0 && F < (1ll << 24) && mode >= 0 && mode <= 2, TGP_ERR_INVALID,
              "tgp_dense_pool_train_rhs_f32: bad shape or mode");
  if (B == 0) return TGP_OK;
  TGP_REQUIRE(rcat && B < 65536, TGP_ERR_INVALID, "tgp_dense_pool_train_rhs_f32: null output or B >= 65536");
  TGP_REQUIRE(mode != 1 || (stats && den && gram && c1), TGP_ERR_INVALID,
              "tgp_dense_pool_train_rhs_f32: mode 1 needs stats, den, gram and c1");
  TGP_REQUIRE(mode != 2 || gram, TGP_ERR_INVALID, "tgp_dense_pool_train_rhs_f32: mode 2 needs gram");
  TGP_REQUIRE(!gw || (W && F > 0), TGP_ERR_INVALID, "tgp_dense_pool_train_rhs_f32: gw needs W and F > 0");
  const int kt = static_cast<int>((K + 31) / 32), ft = static_cast<int>((F + 31) / 32);
  const int64_t gw_blocks = gw ? (2 * K * F + 1023) / 1024 : 0;
  tgp::TrainRhsArgs a{g_raw_a, g_raw_b, stats, den, gram, g_la, g_lb, scale, link_loss, link_scale, eps, g_x, gx_bcast,
                      static_cast<int>(K), static_cast<int>(F), mode, kt, ft, symmetric & 3, rcat, c1, W, gw};
  hipLaunchKernelGGL(train_rhs_kernel, dim3(static_cast<unsigned>(kt * kt + ft * kt + gw_blocks), static_cast<unsigned>(B)),
                     dim3(256), 0, static_cast<hipStream_t>(stream_), a);
  return check_launch("tgp_dense_pool_train_rhs_f32");
}

static int pair_dot(const int64_t* row, const int64_t* col, int64_t E, const float* S, const float* S2, int64_t K,
                    float* out, hipStream_t stream, const char* what) {
  const int k = static_cast<int>(K);
  const bool vec = (K % 4 == 0) && (reinterpret_cast<uintptr_t>(S) % 16 == 0) && (reinterpret_cast<uintptr_t>(S2) % 16 == 0);
  const int64_t units = vec ? K / 4 : K;  // work items per edge
  if (vec) {
    if (units <= 1) launch_edge_dot<1, true>(row, col, E, S, S2, k, out, stream);
    else if (units <= 2) launch_edge_dot<2, true>(row, col, E, S, S2, k, out, stream);
    else if (units <= 4) launch_edge_dot<4, true>(row, col, E, S, S2, k, out, stream);
    else if (units <= 8) launch_edge_dot<8, true>(row, col, E, S, S2, k, out, stream);
    else if (units <= 16) launch_edge_dot<16, true>(row, col, E, S, S2, k, out, stream);
    else if (units <= 32) launch_edge_dot<32, true>(row, col, E, S, S2, k, out, stream);
    else launch_edge_dot<64, true>(row, col, E, S, S2, k, out, stream);
  } else {
    if (units <= 4) launch_edge_dot<4, false>(row, col, E, S, S2, k, out, stream);
    else if (units <= 16) launch_edge_dot<16, false>(row, col, E, S, S2, k, out, stream);
    else launch_edge_dot<64, false>(row, col, E, S, S2, k, out, stream);
  }
  return check_launch(what);
}

extern "C" int tgp_edge_dot_f32(const int64_t* row, const int64_t* col, int64_t E, const float* S, int64_t N,
                                int64_t K, float* out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(E >= 0 && N >= 0 && K >= 0, TGP_ERR_INVALID, "tgp_edge_dot_f32: negative size");
  if (E == 0) return TGP_OK;
  TGP_REQUIRE(row && col && out && (K == 0 || S), TGP_ERR_INVALID, "tgp_edge_dot_f32: null pointer");
  TGP_REQUIRE(K < (1ll << 31), TGP_ERR_RANGE, "tgp_edge_dot_f32: K too large");
  return pair_dot(row, col, E, S, S, K, out, stream, "tgp_edge_dot_f32");
}

extern "C" int tgp_pair_dot_f32(const int64_t* ia, const int64_t* ib, int64_t E, const float* A, const float* Bm,
                                int64_t K, float* out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(E >= 0 && K >= 0, TGP_ERR_INVALID, "tgp_pair_dot_f32: negative size");
  if (E == 0) return TGP_OK;
  TGP_REQUIRE(ia && ib && out && (K == 0 || (A && Bm)), TGP_ERR_INVALID, "tgp_pair_dot_f32: null pointer");
  TGP_REQUIRE(K < (1ll << 31), TGP_ERR_RANGE, "tgp_pair_dot_f32: K too large");
  return pair_dot(ia, ib, E, A, Bm, K, out, stream, "tgp_pair_dot_f32");
}
