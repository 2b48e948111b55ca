// A3 + A7 + A8: dense pooling of a padded batch on the fp32 matrix cores.
//
//   U        = A  S            [B,N,K]   (2*B*N*N*K flop: the dominant product)
//   [A'|X']  = S^T [U | X]     [B,K,K+F] (split over N, partial slabs, fixed-order combine)
//   adj_pool = postprocess(A') fused into the combine (diag=0, D^-1/2 . D^-1/2, /max|.|)
//
// The reference computes (S^T A) S with two torch.matmul calls (connect/dense_conn.py:120-122)
// and S^T X with a third (reduce/base_reduce.py:159); fp32 in, fp32 out, rtol 1e-5.  That
// tolerance on length-N dot products rules out bf16 / split-bf16 inputs, so the products run on
// v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chains, 64 FLOP/clk/SIMD = 157 TFLOP/s peak).
// A S is associated the other way round from the reference (A(S) first): same flops, but the
// intermediate is [N,K] (1/8 .. 1/16 of A) and A is streamed from HBM exactly once in full
// 128-byte row segments.
//
// GEMM kernel: 128x128 output tile per 256-thread workgroup (2x2 waves, each 2x2 MFMA tiles of
// 32x32 -> 64 accumulator VGPRs), BK = 32, register-staged double-buffered LDS.  fp32 MFMA needs
// only 2 operand dwords per 64-cycle instruction, so LDS bandwidth is a non-issue; the layouts
// are chosen for conflict-free ds_write/ds_read and coalesced global loads:
//   * row-major operand tile  (A of A.S):     LDS [128][BK+1]  (odd stride => conflict free)
//   * k-major operand tile    (S, U, X, A^T): LDS [BK][128]    (lanes read consecutive floats)
#include "common.h"

namespace tgp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDA_ROWMAJOR = BK + 1;
constexpr int A_TILE_FLOATS = BM * LDA_ROWMAJOR;  // >= BK*BM, used for both A layouts
constexpr int B_TILE_FLOATS = BK * BN;
constexpr int STAGE_FLOATS = A_TILE_FLOATS + B_TILE_FLOATS;

struct GemmArgs {
  const float* A;
  const float* Bm;
  float* C;
  int M, Nc, Kd;           // C[M,Nc] = op(A)[M,Kd] * Bm[Kd,Nc]
  long lda, ldb, ldc;      // leading dimensions (elements)
  long sA, sB, sC;         // batch strides
  int splits;              // split of Kd across workgroups
  int k_per_split;         // multiple of BK
  long sCsplit;            // stride between partial slabs
  int tiles_m, tiles_n;
  const int64_t* k_ptr;    // optional [batches+1]: batch b reduces over rows k_ptr[b]..k_ptr[b+1] (segment GEMM)
};

// A_KMAJOR = false: A stored [M][Kd] (k contiguous).  true: stored [Kd][M] (m contiguous).
template <bool A_KMAJOR>
__global__ __launch_bounds__(256, 2) void gemm_f32_mfma_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // logical block id, XCD-aware: tiles of one batch element share S / U through one L2
  const int nwg = gridDim.x;
  int bid = xcd_remap(blockIdx.x, nwg);
  const int tn = bid % g.tiles_n; bid /= g.tiles_n;
  const int tm = bid % g.tiles_m; bid /= g.tiles_m;
  const int split = bid % g.splits;
  const int batch = bid / g.splits;

  const float* __restrict__ A = g.A + static_cast<long>(batch) * g.sA;
  const float* __restrict__ Bm = g.Bm + static_cast<long>(batch) * g.sB;
  float* __restrict__ C = g.C + static_cast<long>(batch) * g.sC + static_cast<long>(split) * g.sCsplit;

  const int m0 = tm * BM, n0 = tn * BN;
  int k_lo = 0, k_hi = g.Kd;
  if (g.k_ptr) {
    k_lo = static_cast<int>(g.k_ptr[batch]);
    k_hi = static_cast<int>(g.k_ptr[batch + 1]);
  }
  const int k_begin = k_lo + split * g.k_per_split;
  const int k_end = min(k_hi, k_begin + g.k_per_split);
  const int nk = (k_end - k_begin + BK - 1) / BK;

  const bool a_vec = (g.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
  const bool b_vec = (g.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(Bm) & 15) == 0);

  float4 ra[4], rb[4];

  // ---- global -> registers for stage starting at k0 ---------------------------------
  auto load_a = [&](int k0) {
    if constexpr (!A_KMAJOR) {
      // tile [128 m][32 k]; 8 lanes cover one 128-byte row segment
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + (tid >> 3) + i * 32;
        const int k = k0 + (tid & 7) * 4;
        const float* p = A + static_cast<long>(m) * g.lda + k;
        if (m < g.M && k + 3 < k_end && a_vec) {
          ra[i] = *reinterpret_cast<const float4*>(p);
        } else {
          float t[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) t[j] = (m < g.M && k + j < k_end) ? p[j] : 0.f;
          ra[i] = make_float4(t[0], t[1], t[2], t[3]);
        }
      }
    } else {
      // tile [32 k][128 m]; 32 lanes cover one 512-byte row
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = k0 + (tid >> 5) + i * 8;
        const int m = m0 + (tid & 31) * 4;
        const float* p = A + static_cast<long>(k) * g.lda + m;
        if (k < k_end && m + 3 < g.M && a_vec) {
          ra[i] = *reinterpret_cast<const float4*>(p);
        } else {
          float t[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) t[j] = (k < k_end && m + j < g.M) ? p[j] : 0.f;
          ra[i] = make_float4(t[0], t[1], t[2], t[3]);
        }
      }
    }
  };
  auto load_b = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + (tid >> 5) + i * 8;
      const int n = n0 + (tid & 31) * 4;
      const float* p = Bm + static_cast<long>(k) * g.ldb + n;
      if (k < k_end && n + 3 < g.Nc && b_vec) {
        rb[i] = *reinterpret_cast<const float4*>(p);
      } else {
        float t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = (k < k_end && n + j < g.Nc) ? p[j] : 0.f;
        rb[i] = make_float4(t[0], t[1], t[2], t[3]);
      }
    }
  };
  // ---- registers -> LDS ----------------------------------------------------------------
  auto store_stage = [&](float* As, float* Bs) {
    if constexpr (!A_KMAJOR) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float* d = As + ((tid >> 3) + i * 32) * LDA_ROWMAJOR + (tid & 7) * 4;
        d[0] = ra[i].x; d[1] = ra[i].y; d[2] = ra[i].z; d[3] = ra[i].w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<float4*>(As + ((tid >> 5) + i * 8) * BM + (tid & 31) * 4) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<float4*>(Bs + ((tid >> 5) + i * 8) * BN + (tid & 31) * 4) = rb[i];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nk > 0) {
    load_a(k_begin);
    load_b(k_begin);
    store_stage(smem, smem + A_TILE_FLOATS);
  }
  __syncthreads();

  const int lm = lane & 31, lk = lane >> 5;
  for (int t = 0; t < nk; ++t) {
    float* As = smem + (t & 1) * STAGE_FLOATS;
    float* Bs = As + A_TILE_FLOATS;
    const bool more = (t + 1) < nk;
    if (more) {
      load_a(k_begin + (t + 1) * BK);
      load_b(k_begin + (t + 1) * BK);
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int m = wm * 64 + i * 32 + lm;
        a[i] = A_KMAJOR ? As[(kk + lk) * BM + m] : As[m * LDA_ROWMAJOR + kk + lk];
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[(kk + lk) * BN + wn * 64 + j * 32 + lm];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      float* An = smem + ((t + 1) & 1) * STAGE_FLOATS;
      store_stage(An, An + A_TILE_FLOATS);
    }
    __syncthreads();
  }

  // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -----------
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + lm;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row < g.M && col < g.Nc) C[static_cast<long>(row) * g.ldc + col] = acc[i][j][r];
      }
    }
}

// ------------------------------------------------------------------------------------------
// Combine the split-K slabs of one graph in fixed order and apply utils/ops.py:282-335.
// One 1024-thread workgroup per graph; the K x K matrix is re-read from L2 between passes.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_reduce_max_1024(float v, float* s_red) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = fmaxf(v, __shfl_down(v, d, WAVE));
  if (lane_id() == 0) s_red[wave_id()] = v;
  __syncthreads();
  float r = s_red[0];
  for (int w = 1; w < 16; ++w) r = fmaxf(r, s_red[w]);
  __syncthreads();
  return r;
}

// src: [B][splits][K][ld_src] slabs (splits may be 1, ld_src >= K); raw (optional) and dst: [B][K][K].
__global__ __launch_bounds__(1024) void dense_post_kernel(const float* __restrict__ src, int splits,
                                                          long s_split, long s_batch, long ld_src, int K,
                                                          int flags, float* __restrict__ raw,
                                                          float* __restrict__ dst) {
  extern __shared__ __attribute__((aligned(16))) float dvec[];  // [K] degree vector
  __shared__ float s_red[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* sb = src + static_cast<long>(b) * s_batch;
  float* rawb = raw ? raw + static_cast<long>(b) * K * K : nullptr;
  float* dstb = dst ? dst + static_cast<long>(b) * K * K : nullptr;
  const long kk = static_cast<long>(K) * K;

  // pass 1: combine slabs (fixed order), emit raw, write diag-cleared matrix to dst
  for (long e = tid; e < kk; e += 1024) {
    const int i = static_cast<int>(e / K), j = static_cast<int>(e - static_cast<long>(i) * K);
    float v = sb[static_cast<long>(i) * ld_src + j];
    for (int s = 1; s < splits; ++s) v = __fadd_rn(v, sb[s * s_split + static_cast<long>(i) * ld_src + j]);
    if (rawb) rawb[e] = v;
    if (dstb) {
      if ((flags & TGP_REMOVE_SELF_LOOPS) && i == j) v = 0.f;
      dstb[e] = v;
    }
  }
  if (!dstb) return;
  __syncthreads();

  if (flags & TGP_DEGREE_NORM) {
    // d = sqrt(clamp(sum over axis, eps)); axis -2 (column sums) when TGP_SUM_AXIS_ROWS
    for (int j = tid; j < K; j += 1024) {
      float s = 0.f;
      if (flags & TGP_SUM_AXIS_ROWS) {
        for (int i = 0; i < K; ++i) s = __fadd_rn(s, dstb[static_cast<long>(i) * K + j]);
      } else {
        for (int i = 0; i < K; ++i) s = __fadd_rn(s, dstb[static_cast<long>(j) * K + i]);
      }
      dvec[j] = sqrtf(fmaxf(s, TGP_EPS));
    }
    __syncthreads();
    for (long e = tid; e < kk; e += 1024) {
      const int i = static_cast<int>(e / K), j = static_cast<int>(e - static_cast<long>(i) * K);
      // (adj / d) / d^T with d shaped [1,K] (axis -2) or [K,1] (axis -1): ops.py:318-319
      const float first = (flags & TGP_SUM_AXIS_ROWS) ? dvec[j] : dvec[i];
      const float second = (flags & TGP_SUM_AXIS_ROWS) ? dvec[i] : dvec[j];
      dstb[e] = (dstb[e] / first) / second;
    }
    __syncthreads();
  }
  if (flags & TGP_EDGE_WEIGHT_NORM) {
    float m = 0.f;
    for (long e = tid; e < kk; e += 1024) m = fmaxf(m, fabsf(dstb[e]));
    m = block_reduce_max_1024(m, s_red);
    if (m == 0.f) m = 1.f;
    for (long e = tid; e < kk; e += 1024) dstb[e] = dstb[e] / m;
  }
}

// x_pool slabs -> x_pool (fixed-order combine); src [B][splits][K][ld_src] (columns c0..c0+F)
__global__ __launch_bounds__(256) void combine_slabs_kernel(const float* __restrict__ src, int splits,
                                                            long s_split, long s_batch, long ld_src, int c0,
                                                            int K, int F, float* __restrict__ dst) {
  const long total = static_cast<long>(K) * F;
  const int b = blockIdx.y;
  const float* sb = src + static_cast<long>(b) * s_batch;
  for (long e = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; e < total;
       e += static_cast<long>(gridDim.x) * 256) {
    const int i = static_cast<int>(e / F), f = static_cast<int>(e - static_cast<long>(i) * F);
    float v = sb[static_cast<long>(i) * ld_src + c0 + f];
    for (int s = 1; s < splits; ++s) v = __fadd_rn(v, sb[s * s_split + static_cast<long>(i) * ld_src + c0 + f]);
    dst[static_cast<long>(b) * total + e] = v;
  }
}

template <bool A_KMAJOR>
static void launch_gemm(const GemmArgs& g, int batches, hipStream_t stream) {
  const int nwg = batches * g.splits * g.tiles_m * g.tiles_n;
  const size_t lds = 2 * STAGE_FLOATS * sizeof(float);
  hipLaunchKernelGGL((gemm_f32_mfma_kernel<A_KMAJOR>), dim3(nwg), dim3(256), lds, stream, g);
}

struct DensePlan {
  int splits;
  int k_per_split;
  size_t u_floats, slab_floats;
};

static DensePlan dense_plan(int64_t B, int64_t N, int64_t K, int64_t F) {
  DensePlan p;
  // second product: output is only K x (K+F) per graph -> split N so the chip is filled
  const int64_t tiles = ((K + BM - 1) / BM) * (((K + BN - 1) / BN) + ((F + BN - 1) / BN));
  const int64_t base = B * (tiles > 0 ? tiles : 1);
  int64_t splits = (2 * 256 + base - 1) / base;  // aim for ~2 workgroups per CU
  const int64_t max_splits = (N + BK - 1) / BK;
  if (splits > max_splits) splits = max_splits;
  if (splits > 32) splits = 32;
  if (splits < 1) splits = 1;
  int64_t kps = (N + splits - 1) / splits;
  kps = (kps + BK - 1) / BK * BK;
  splits = (N + kps - 1) / kps;
  if (splits < 1) splits = 1;
  p.splits = static_cast<int>(splits);
  p.k_per_split = static_cast<int>(kps);
  p.u_floats = static_cast<size_t>(B) * N * K;
  p.slab_floats = static_cast<size_t>(B) * splits * K * (K + F);
  return p;
}

}  // namespace tgp

using namespace tgp;

extern "C" size_t tgp_dense_pool_workspace_bytes(int64_t B, int64_t N, int64_t K, int64_t F) {
  if (B <= 0 || N <= 0 || K <= 0) return 256;
  const DensePlan p = dense_plan(B, N, K, F > 0 ? F : 0);
  return align_up(p.u_floats * 4) + align_up(p.slab_floats * 4) + 256;
}

extern "C" int tgp_dense_pool_f32(const float* S, const float* A, const float* X, int64_t B, int64_t N,
                                  int64_t K, int64_t F, int flags, float* x_pool, float* adj_raw,
                                  float* adj_pool, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(B >= 0 && N >= 0 && K >= 0 && F >= 0, TGP_ERR_INVALID, "tgp_dense_pool_f32: negative size");
  if (B == 0 || K == 0) return TGP_OK;
  const bool want_x = X && x_pool && F > 0;
  const bool want_a = A && (adj_raw || adj_pool);
  TGP_REQUIRE(S || N == 0, TGP_ERR_INVALID, "tgp_dense_pool_f32: S is null");
  TGP_REQUIRE(N < (1ll << 31) && K <= 16000 && F < (1ll << 31) && B < (1ll << 24), TGP_ERR_RANGE,
              "tgp_dense_pool_f32: dimension too large");
  if (N == 0) {
    if (want_x) (void)hipMemsetAsync(x_pool, 0, sizeof(float) * B * K * F, stream);
    if (adj_raw) (void)hipMemsetAsync(adj_raw, 0, sizeof(float) * B * K * K, stream);
    if (adj_pool) (void)hipMemsetAsync(adj_pool, 0, sizeof(float) * B * K * K, stream);
    return check_launch("tgp_dense_pool_f32");
  }
  TGP_REQUIRE(ws && ws_bytes >= tgp_dense_pool_workspace_bytes(B, N, K, F), TGP_ERR_WORKSPACE,
              "tgp_dense_pool_f32: workspace too small");
  const DensePlan p = dense_plan(B, N, K, F);
  Carver cv(ws);
  float* U = cv.take<float>(p.u_floats);
  float* slabs = cv.take<float>(p.slab_floats);
  const long ldslab = K + F;
  const long s_split = static_cast<long>(K) * ldslab;
  const long s_batch = s_split * p.splits;

  if (want_a) {
    // U[b] = A[b] S[b]     (M = N, Kd = N, Nc = K)
    GemmArgs g{};
    g.A = A; g.Bm = S; g.C = U;
    g.M = static_cast<int>(N); g.Nc = static_cast<int>(K); g.Kd = static_cast<int>(N);
    g.lda = N; g.ldb = K; g.ldc = K;
    g.sA = N * N; g.sB = N * K; g.sC = N * K;
    g.splits = 1; g.k_per_split = static_cast<int>((N + BK - 1) / BK * BK); g.sCsplit = 0;
    g.tiles_m = cdiv(N, BM); g.tiles_n = cdiv(K, BN);
    if (flags & TGP_ADJ_TRANSPOSED) launch_gemm<true>(g, static_cast<int>(B), stream);
    else launch_gemm<false>(g, static_cast<int>(B), stream);

    // slabs[b][s][:, 0:K] = S[b]^T U[b]  over the s-th slice of N
    GemmArgs h{};
    h.A = S; h.Bm = U; h.C = slabs;
    h.M = static_cast<int>(K); h.Nc = static_cast<int>(K); h.Kd = static_cast<int>(N);
    h.lda = K; h.ldb = K; h.ldc = ldslab;
    h.sA = N * K; h.sB = N * K; h.sC = s_batch;
    h.splits = p.splits; h.k_per_split = p.k_per_split; h.sCsplit = s_split;
    h.tiles_m = cdiv(K, BM); h.tiles_n = cdiv(K, BN);
    launch_gemm<true>(h, static_cast<int>(B), stream);
  }
  if (want_x) {
    // slabs[b][s][:, K:K+F] = S[b]^T X[b]
    GemmArgs h{};
    h.A = S; h.Bm = X; h.C = slabs + K;
    h.M = static_cast<int>(K); h.Nc = static_cast<int>(F); h.Kd = static_cast<int>(N);
    h.lda = K; h.ldb = F; h.ldc = ldslab;
    h.sA = N * K; h.sB = N * F; h.sC = s_batch;
    h.splits = p.splits; h.k_per_split = p.k_per_split; h.sCsplit = s_split;
    h.tiles_m = cdiv(K, BM); h.tiles_n = cdiv(F, BN);
    launch_gemm<true>(h, static_cast<int>(B), stream);
    const long total = K * F;
    int gx = static_cast<int>((total + 255) / 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(combine_slabs_kernel, dim3(gx, static_cast<unsigned>(B)), dim3(256), 0, stream, slabs,
                       p.splits, s_split, s_batch, ldslab, static_cast<int>(K), static_cast<int>(K),
                       static_cast<int>(F), x_pool);
  }
  if (want_a) {
    hipLaunchKernelGGL(dense_post_kernel, dim3(static_cast<unsigned>(B)), dim3(1024), K * sizeof(float), stream,
                       slabs, p.splits, s_split, s_batch, ldslab, static_cast<int>(K), flags, adj_raw, adj_pool);
  }
  return check_launch("tgp_dense_pool_f32");
}

extern "C" int tgp_postprocess_dense_f32(const float* src, float* dst, int64_t B, int64_t K, int flags,
                                         void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(B >= 0 && K >= 0, TGP_ERR_INVALID, "tgp_postprocess_dense_f32: negative size");
  if (B == 0 || K == 0) return TGP_OK;
  TGP_REQUIRE(src && dst, TGP_ERR_INVALID, "tgp_postprocess_dense_f32: null pointer");
  TGP_REQUIRE(K <= 16000, TGP_ERR_RANGE, "tgp_postprocess_dense_f32: K > 16000 not supported");
  hipLaunchKernelGGL(dense_post_kernel, dim3(static_cast<unsigned>(B)), dim3(1024), K * sizeof(float), stream, src,
                     1, 0L, static_cast<long>(K) * K, static_cast<long>(K), static_cast<int>(K), flags,
                     static_cast<float*>(nullptr), dst);
  return check_launch("tgp_postprocess_dense_f32");
}

// Generic batched fp32 GEMM on the matrix cores (used by Lift and by the unbatched dense paths).
extern "C" int tgp_bmm_f32(const float* A, const float* Bm, float* C, int64_t batch, int64_t M, int64_t Nc,
                           int64_t Kd, int trans_a, int64_t lda, int64_t ldb, int64_t ldc, int64_t sA,
                           int64_t sB, int64_t sC, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(batch >= 0 && M >= 0 && Nc >= 0 && Kd >= 0, TGP_ERR_INVALID, "tgp_bmm_f32: negative size");
  if (batch == 0 || M == 0 || Nc == 0) return TGP_OK;
  TGP_REQUIRE(C && (Kd == 0 || (A && Bm)), TGP_ERR_INVALID, "tgp_bmm_f32: null pointer");
  TGP_REQUIRE(M < (1ll << 31) && Nc < (1ll << 31) && Kd < (1ll << 31), TGP_ERR_RANGE, "tgp_bmm_f32: too large");
  GemmArgs g{};
  g.A = A; g.Bm = Bm; g.C = C;
  g.M = static_cast<int>(M); g.Nc = static_cast<int>(Nc); g.Kd = static_cast<int>(Kd);
  g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.sA = sA; g.sB = sB; g.sC = sC;
  g.splits = 1; g.k_per_split = static_cast<int>((Kd + BK - 1) / BK * BK); g.sCsplit = 0;
  g.tiles_m = cdiv(M, BM); g.tiles_n = cdiv(Nc, BN);
  const int64_t nwg = batch * g.tiles_m * g.tiles_n;
  TGP_REQUIRE(nwg < (1ll << 31), TGP_ERR_RANGE, "tgp_bmm_f32: grid too large");
  if (trans_a) launch_gemm<true>(g, static_cast<int>(batch), stream);
  else launch_gemm<false>(g, static_cast<int>(batch), stream);
  return check_launch("tgp_bmm_f32");
}

// A3' / A7' (reduce/base_reduce.py:170-182, connect/dense_conn.py:195-206): per-graph S_b^T Y_b for an
// un-padded batch.  Graph b owns node rows ptr[b]..ptr[b+1] of S [Ntot,K] and Y [Ntot,F]; C is [B,K,F].
// Replaces the reference's Python loop over graphs with one launch (no padding, no densification).
extern "C" int tgp_segment_gemm_tn_f32(const float* S, const float* Y, const int64_t* ptr, float* C, int64_t B,
                                       int64_t Ntot, int64_t K, int64_t F, int64_t max_nodes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(B >= 0 && Ntot >= 0 && K >= 0 && F >= 0, TGP_ERR_INVALID, "tgp_segment_gemm_tn_f32: negative size");
  if (B == 0 || K == 0 || F == 0) return TGP_OK;
  TGP_REQUIRE(C && ptr && (Ntot == 0 || (S && Y)), TGP_ERR_INVALID, "tgp_segment_gemm_tn_f32: null pointer");
  TGP_REQUIRE(Ntot < (1ll << 31) && K < (1ll << 31) && F < (1ll << 31), TGP_ERR_RANGE,
              "tgp_segment_gemm_tn_f32: too large");
  GemmArgs g{};
  g.A = S; g.Bm = Y; g.C = C;
  g.M = static_cast<int>(K); g.Nc = static_cast<int>(F); g.Kd = static_cast<int>(Ntot);
  g.lda = K; g.ldb = F; g.ldc = F; g.sA = 0; g.sB = 0; g.sC = K * F;
  g.splits = 1;
  const int64_t span = max_nodes > 0 ? max_nodes : Ntot;
  g.k_per_split = static_cast<int>((span + BK - 1) / BK * BK);
  if (g.k_per_split < BK) g.k_per_split = BK;
  g.sCsplit = 0;
  g.tiles_m = cdiv(K, BM); g.tiles_n = cdiv(F, BN);
  g.k_ptr = ptr;
  launch_gemm<true>(g, static_cast<int>(B), stream);
  return check_launch("tgp_segment_gemm_tn_f32");
}
