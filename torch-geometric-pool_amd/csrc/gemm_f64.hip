// fp64 dense path (r5): S^T X, S^T A S and their gradients for float64 callers on v_mfma_f64_16x16x4_f64.
//
// The reference runs model.double() inputs through torch.matmul in fp64 (reduce/base_reduce.py:158-161,
// connect/dense_conn.py:111-122); until r5 this build narrowed them to the fp32 matrix path.  This file is the fp64
// twin of gemm_mfma.h / dense.hip: one LDS-tiled batched GEMM kernel on the fp64 matrix instruction and the entry
// points the host mirror routes float64 tensors to (tgp_bmm_f64, tgp_dense_pool_f64, tgp_segment_gemm_{tn,nn}_f64,
// tgp_spmm_csr_f64).  Same association as the fp32 path (U = A S first, then S^T [U | X] split over the node range
// with a fixed-order slab combine), so a result is reproducible run to run.
//
// Kernel: 64 x 64 output tile per 256-thread workgroup, 4 waves as 2 x 2, each wave 32 x 32 = 2 x 2 MFMA tiles of
// 16 x 16 (4 accumulators x 4 doubles = 32 VGPRs).  BK = 16, register-staged double-buffered LDS, one barrier per
// k-tile.  Operand layout of v_mfma_f64_16x16x4_f64 (cdna_hip_programming.md section 3): lane l holds A[l & 15][l >> 4]
// and B[l >> 4][l & 15]; result register r of lane l is D[(l >> 4) + 4 r][l & 15].
// LDS images are padded so that the 64-bit operand reads of a 32-lane group touch 32 distinct bank pairs:
//   k contiguous in memory  -> [64][18]   (address m * 18 + k:  (18 m + k) mod 32 distinct for m < 16, k < 2)
//   m / n contiguous        -> [16][80]   (address k * 80 + n:  80 mod 32 = 16, so k = 0 / 1 take the two halves)
#include <stdlib.h>

#include "common.h"

namespace tgp {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int DBK = 16;           // k-tile
constexpr int DT = 64;            // output tile edge
constexpr int D_ROWMAJOR_LD = 18;
constexpr int D_KMAJOR_LD = 80;
constexpr int D_STAGE = DBK * D_KMAJOR_LD;  // 1280 doubles >= 64 * 18 = 1152: one operand tile of either layout

struct Gemm64Rhs {
  const double* Bm;
  double* C;
  int Nc;
  long ldb, ldc, sB, sC, sCsplit;
};

struct Gemm64Args {
  const double* A;
  long lda, sA;
  int M, Kd;               // C[M,Nc] = op(A)[M,Kd] * Bm[Kd,Nc]
  Gemm64Rhs rhs[2];        // column tiles >= tiles_n0 use rhs[1] (S^T [U | X] as one grid)
  int tiles_m, tiles_n0, tiles_n;
  int splits, k_per_split; // split of Kd across workgroups; k_per_split is a multiple of DBK
  const int64_t* k_ptr;    // optional [batches+1]: batch b reduces over rows k_ptr[b]..k_ptr[b+1]
  const int64_t* m_ptr;    // optional [batches+1] (row-major A only): batch b owns rows m_ptr[b]..m_ptr[b+1] of A and C
  int accumulate;          // splits == 1: C += op(A) Bm
};

// two consecutive doubles at p (elements e, e + 1 of a row whose valid length ends at `limit`)
__device__ __forceinline__ double2 ld2_guarded(const double* p, bool row_ok, long e, long limit) {
  double2 v = make_double2(0.0, 0.0);
  if (row_ok) {
    if (e + 1 < limit) {
      if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
        v = *reinterpret_cast<const double2*>(p);
      } else {
        v.x = p[0];
        v.y = p[1];
      }
    } else if (e < limit) {
      v.x = p[0];
    }
  }
  return v;
}

// A_KMAJOR = false: A stored [M][Kd] (k contiguous).  true: stored [Kd][M] (m contiguous; C = A^T Bm).
// BUF: buffer-descriptor loads, as in the fp32 kernel (gemm_mfma.h): base + valid bytes of this batch element in SGPRs,
// one per-lane byte offset computed once (rows / columns outside the problem point past the end, which the hardware's
// range check turns into zeros), a scalar offset that advances with the k-tile: the steady-state loop issues its four
// 16-byte loads without a branch or a vector-ALU instruction.  (The guarded form compiles to a dozen branches per
// k-tile, each a `s_waitcnt vmcnt(0)` boundary: 49 TFLOP/s at N = 8192 against rocBLAS' 77.)  A vector that runs past
// the end of its row picks up the head of the next row: masked in the k tail, otherwise it lands in output rows /
// columns that are never stored.  Needs every matrix of a batch element within 2^31 bytes; otherwise BUF = false.
template <bool A_KMAJOR, bool BUF>
__global__ __launch_bounds__(256) void gemm_f64_mfma_kernel(Gemm64Args g) {
  __shared__ __attribute__((aligned(16))) double smem[2 * 2 * D_STAGE];  // 40 KB: two stages of (A, B)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn_all = bid % g.tiles_n; bid /= g.tiles_n;
  const int tm = bid % g.tiles_m; bid /= g.tiles_m;
  const int split = bid % g.splits;
  const int batch = bid / g.splits;
  const int which = tn_all >= g.tiles_n0 ? 1 : 0;
  const int tn = which ? tn_all - g.tiles_n0 : tn_all;
  const Gemm64Rhs& R = g.rhs[which];

  const double* __restrict__ A = g.A + static_cast<long>(batch) * g.sA;
  const double* __restrict__ Bm = R.Bm + static_cast<long>(batch) * R.sB;
  double* __restrict__ C = R.C + static_cast<long>(batch) * R.sC + static_cast<long>(split) * R.sCsplit;
  const int Nc = R.Nc;
  const long lda = g.lda, ldb = R.ldb;
  const int m0 = tm * DT, n0 = tn * DT;
  int M = g.M;
  if (g.m_ptr) {
    const long m_lo = g.m_ptr[batch];
    M = static_cast<int>(g.m_ptr[batch + 1] - m_lo);
    if (m0 >= M) return;  // workgroup-uniform: this graph is shorter than the longest one
    A += m_lo * lda;
    C += m_lo * R.ldc;
  }
  int k_lo = 0, k_hi = g.Kd;
  if (g.k_ptr) {
    k_lo = static_cast<int>(g.k_ptr[batch]);
    k_hi = static_cast<int>(g.k_ptr[batch + 1]);
  }
  const int k_begin = k_lo + split * g.k_per_split;
  const int k_end = min(k_hi, k_begin + g.k_per_split);
  const int nk = k_end > k_begin ? (k_end - k_begin + DBK - 1) / DBK : 0;

  // global -> register staging: two double2 per operand per thread and k-tile
  // (two sets, r5 late: tile t + 2 is loaded into one during stage t while tile t + 1 is stored from the other -- a
  //  load has a whole stage and a half to land, where the single set gave it one stage's MFMAs)
  double2 ra2[2][2], rb2[2][2];
  constexpr int OOB = static_cast<int>(0x80000000u);
  [[maybe_unused]] __amdgpu_buffer_rsrc_t rsrc_a, rsrc_b;
  [[maybe_unused]] int voff_a[2], voff_b[2], kloc_a[2], kloc_b[2];
  if constexpr (BUF) {
    const int a_bytes = A_KMAJOR ? (static_cast<int>(g.Kd - 1) * static_cast<int>(lda) + M) * 8
                                 : (static_cast<int>(M - 1) * static_cast<int>(lda) + g.Kd) * 8;
    const int b_bytes = (static_cast<int>(g.Kd - 1) * static_cast<int>(ldb) + Nc) * 8;
    rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(A), 0, a_bytes, 0x00020000);
    rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Bm), 0, b_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + i * 256;
      if constexpr (!A_KMAJOR) {
        const int m = m0 + (idx >> 3);
        kloc_a[i] = (idx & 7) * 2;
        voff_a[i] = m < M ? (m * static_cast<int>(lda) + kloc_a[i]) * 8 : OOB;
      } else {
        const int m = m0 + (idx & 31) * 2;
        kloc_a[i] = idx >> 5;
        voff_a[i] = m < M ? (kloc_a[i] * static_cast<int>(lda) + m) * 8 : OOB;
      }
      const int n = n0 + (idx & 31) * 2;
      kloc_b[i] = idx >> 5;
      voff_b[i] = n < Nc ? (kloc_b[i] * static_cast<int>(ldb) + n) * 8 : OOB;
    }
  }
  auto buf_ld2 = [&](__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    double2 d;
    d.x = __hiloint2double(static_cast<int>(v.y), static_cast<int>(v.x));
    d.y = __hiloint2double(static_cast<int>(v.w), static_cast<int>(v.z));
    return d;
  };
  auto load_tiles = [&](auto set_c, int k0, auto steady_c) {
    double2 (&ra)[2] = ra2[decltype(set_c)::value];
    double2 (&rb)[2] = rb2[decltype(set_c)::value];
    if constexpr (BUF) {
      const bool tail = !decltype(steady_c)::value && k0 + DBK > k_end;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int soff_a = A_KMAJOR ? k0 * static_cast<int>(lda) * 8 : k0 * 8;
        const int soff_b = k0 * static_cast<int>(ldb) * 8;
        if (!tail) {
          ra[i] = buf_ld2(rsrc_a, voff_a[i], soff_a);
          rb[i] = buf_ld2(rsrc_b, voff_b[i], soff_b);
        } else {
          ra[i] = buf_ld2(rsrc_a, k0 + kloc_a[i] < k_end ? voff_a[i] : OOB, soff_a);
          if constexpr (!A_KMAJOR)  // k runs along the vector: an odd range ends inside one
            if (k0 + kloc_a[i] + 1 >= k_end) ra[i].y = 0.0;
          rb[i] = buf_ld2(rsrc_b, k0 + kloc_b[i] < k_end ? voff_b[i] : OOB, soff_b);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * 256;
        if constexpr (!A_KMAJOR) {   // [64 m][16 k]: 8 lanes cover one 128-byte row segment
          const int m = m0 + (idx >> 3), k = k0 + (idx & 7) * 2;
          ra[i] = ld2_guarded(A + static_cast<long>(m) * lda + k, m < M, k, k_end);
        } else {                     // [16 k][64 m]: 32 lanes cover one row
          const int k = k0 + (idx >> 5), m = m0 + (idx & 31) * 2;
          ra[i] = ld2_guarded(A + static_cast<long>(k) * lda + m, k < k_end, m, M);
        }
        const int k = k0 + (idx >> 5), n = n0 + (idx & 31) * 2;
        rb[i] = ld2_guarded(Bm + static_cast<long>(k) * ldb + n, k < k_end, n, Nc);
      }
    }
  };
  auto store_tiles = [&](auto set_c, double* As, double* Bs) {
    const double2 (&ra)[2] = ra2[decltype(set_c)::value];
    const double2 (&rb)[2] = rb2[decltype(set_c)::value];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + i * 256;
      if constexpr (!A_KMAJOR)
        *reinterpret_cast<double2*>(As + (idx >> 3) * D_ROWMAJOR_LD + (idx & 7) * 2) = ra[i];
      else
        *reinterpret_cast<double2*>(As + (idx >> 5) * D_KMAJOR_LD + (idx & 31) * 2) = ra[i];
      *reinterpret_cast<double2*>(Bs + (idx >> 5) * D_KMAJOR_LD + (idx & 31) * 2) = rb[i];
    }
  };

  f64x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;

  const int l15 = lane & 15, lk = lane >> 4;
  const int a_off = A_KMAJOR ? lk * D_KMAJOR_LD + wm * 32 + l15 : (wm * 32 + l15) * D_ROWMAJOR_LD + lk;
  constexpr int a_sub = A_KMAJOR ? 16 : 16 * D_ROWMAJOR_LD;   // next 16 rows of the wave's strip
  constexpr int a_kstep = A_KMAJOR ? 4 * D_KMAJOR_LD : 4;     // next MFMA k-group
  const int b_off = lk * D_KMAJOR_LD + wn * 32 + l15;

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  if (nk > 0) {
    load_tiles(S0{}, k_begin, std::false_type{});
    if (nk > 1) load_tiles(S1{}, k_begin + DBK, std::false_type{});
    store_tiles(S0{}, smem, smem + D_STAGE);
  }
  __syncthreads();
  auto multiply = [&](const double* As, const double* Bs) {
#pragma unroll
    for (int kk = 0; kk < DBK / 4; ++kk) {
      const double a0 = As[a_off + kk * a_kstep], a1 = As[a_off + kk * a_kstep + a_sub];
      const double b0 = Bs[b_off + kk * 4 * D_KMAJOR_LD], b1 = Bs[b_off + kk * 4 * D_KMAJOR_LD + 16];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
  };
  // stage t (parity PAR = t & 1): tile t is in LDS buffer PAR; tile t + 2 is loaded into register set PAR, tile t + 1
  // (in set 1 - PAR since the previous stage) is stored to buffer 1 - PAR behind the MFMAs.  STEADY: both exist and the
  // loaded one is full, known at compile time (no branch, no wait the compiler must place for a skipped block).
  auto stage = [&](auto par_c, auto steady_c, int t) {
    constexpr int PAR = decltype(par_c)::value;
    constexpr bool STEADY = decltype(steady_c)::value;
    const double* As = smem + PAR * 2 * D_STAGE;
    if (STEADY || t + 2 < nk) load_tiles(par_c, k_begin + (t + 2) * DBK, steady_c);
    multiply(As, As + D_STAGE);
    if (STEADY || t + 1 < nk) {
      double* An = smem + (1 - PAR) * 2 * D_STAGE;
      store_tiles(std::integral_constant<int, 1 - PAR>{}, An, An + D_STAGE);
    }
    __syncthreads();
  };
  int t = 0;
  if constexpr (BUF) {
    for (; t + 3 < nk && k_begin + (t + 4) * DBK <= k_end; t += 2) {
      stage(S0{}, std::true_type{}, t);
      stage(S1{}, std::true_type{}, t + 1);
    }
  }
  for (; t < nk; t += 2) {
    stage(S0{}, std::false_type{}, t);
    if (t + 1 < nk) stage(S1{}, std::false_type{}, t + 1);
  }

  // epilogue: D[(lane >> 4) + 4 r][lane & 15] of every 16 x 16 tile
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 32 + j * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 32 + i * 16 + lk + 4 * r;
        if (row < M && col < Nc) {
          double* dst = C + static_cast<long>(row) * R.ldc + col;
          *dst = g.accumulate ? *dst + acc[i][j][r] : acc[i][j][r];
        }
      }
    }
}

// the buffer-load form needs every matrix of a batch element inside the 32-bit byte offsets of a descriptor
static bool gemm64_buf_ok(const Gemm64Args& g, bool a_kmajor) {
  static const bool force_guarded = getenv("TGP_GEMM64_GUARDED") && atoi(getenv("TGP_GEMM64_GUARDED"));
  if (force_guarded) return false;
  const long lim = (1l << 31) - 4096;
  const long a_rows = a_kmajor ? g.Kd : (g.m_ptr ? g.M : g.M);
  bool ok = a_rows * g.lda * 8 < lim && (a_kmajor ? g.M : g.Kd) * 8l < lim;
  for (int w = 0; w < 2; ++w)
    if (w == 0 || g.rhs[1].Bm) ok = ok && static_cast<long>(g.Kd) * g.rhs[w].ldb * 8 < lim && g.rhs[w].Nc * 8l < lim;
  return ok;
}

template <bool A_KMAJOR>
static void launch_gemm64(Gemm64Args g, int batches, hipStream_t stream) {
  g.tiles_m = cdiv(g.M, DT);
  g.tiles_n0 = cdiv(g.rhs[0].Nc, DT);
  g.tiles_n = g.tiles_n0 + (g.rhs[1].Bm ? cdiv(g.rhs[1].Nc, DT) : 0);
  const long nwg = static_cast<long>(batches) * g.splits * g.tiles_m * g.tiles_n;
  if (nwg <= 0) return;
  if (gemm64_buf_ok(g, A_KMAJOR))
    hipLaunchKernelGGL((gemm_f64_mfma_kernel<A_KMAJOR, true>), dim3(static_cast<unsigned>(nwg)), dim3(256), 0, stream, g);
  else
    hipLaunchKernelGGL((gemm_f64_mfma_kernel<A_KMAJOR, false>), dim3(static_cast<unsigned>(nwg)), dim3(256), 0, stream,
                       g);
}

// dst[b][e] = sum over splits (in split order) of src[b][s][e]
__global__ __launch_bounds__(256) void combine_slabs_f64_kernel(const double* __restrict__ src, int splits, long s_split,
                                                                long s_batch, long total, double* __restrict__ dst) {
  const double* s = src + static_cast<long>(blockIdx.y) * s_batch;
  double* d = dst + static_cast<long>(blockIdx.y) * total;
  for (long e = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; e < total; e += static_cast<long>(gridDim.x) * 256) {
    double acc = s[e];
    for (int q = 1; q < splits; ++q) acc += s[q * s_split + e];
    d[e] = acc;
  }
}

// ---- tail of the fused float64 call, spread over K / 16 workgroups per graph (r5) ----------------------------------------
// r5 first had: slab combine (A), slab combine (X), f64_post_dense_kernel (one workgroup per graph): 5 + 5 + 29 us at
// C2 behind a 35 us second product.  Now two launches of B * ceil(K / 16) workgroups:
//   (1) dense64_combine_kernel: the block's 16 rows of S^T A S (and of S^T X) = slabs added in split order, written to
//       `raw`; and the block's partial COLUMN sums of those rows (diagonal left out when the self loops are removed);
//   (2) dense64_post_rows_kernel: d_j = sqrt(max(sum of the column's partials in block order, eps)) for every column,
//       then the block's rows of (R1 / d[1,K]) / d[K,1] (utils/ops.py:311-320 with the sum over dim -2).
// Only that axis (the connectors' default adj_transpose=True) without edge_weight_norm; other flags keep the composed form.
constexpr int D64_ROWS = 16;

__global__ __launch_bounds__(256) void dense64_combine_kernel(const double* __restrict__ aslab, int splits, long as_split,
                                                              long as_batch, const double* __restrict__ xslab,
                                                              long xs_split, long xs_batch, int K, int F, int skip_diag,
                                                              double* __restrict__ raw, double* __restrict__ x_pool,
                                                              double* __restrict__ colpart) {
  const int nb = (K + D64_ROWS - 1) / D64_ROWS;
  const int b = blockIdx.x / nb, rb = blockIdx.x - b * nb;
  const int i0 = rb * D64_ROWS, rows = K - i0 < D64_ROWS ? K - i0 : D64_ROWS;
  if (aslab) {
    const double* sb = aslab + static_cast<long>(b) * as_batch;
    for (int j = threadIdx.x; j < K; j += 256) {  // a thread per column: coalesced across the threads
      // the 16 rows of a slab are requested together (a loop with a run-time trip count would wait per load)
      double acc[D64_ROWS];
#pragma unroll
      for (int r = 0; r < D64_ROWS; ++r) acc[r] = sb[static_cast<long>(i0 + (r < rows ? r : 0)) * K + j];
      for (int q = 1; q < splits; ++q) {
        double v[D64_ROWS];
#pragma unroll
        for (int r = 0; r < D64_ROWS; ++r) v[r] = sb[q * as_split + static_cast<long>(i0 + (r < rows ? r : 0)) * K + j];
#pragma unroll
        for (int r = 0; r < D64_ROWS; ++r) acc[r] += v[r];
      }
      double cs = 0.0;
#pragma unroll
      for (int r = 0; r < D64_ROWS; ++r) {
        if (r < rows) {
          if (raw) raw[static_cast<long>(b) * K * K + static_cast<long>(i0 + r) * K + j] = acc[r];
          if (!(skip_diag && i0 + r == j)) cs += acc[r];
        }
      }
      if (colpart) colpart[(static_cast<long>(b) * nb + rb) * K + j] = cs;
    }
  }
  if (xslab) {
    const double* xb = xslab + static_cast<long>(b) * xs_batch + static_cast<long>(i0) * F;
    double* xo = x_pool + static_cast<long>(b) * K * F + static_cast<long>(i0) * F;
    for (int e0 = threadIdx.x; e0 < rows * F; e0 += 256 * 4) {  // four elements per thread in flight
      double v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = xb[e0 + u * 256 < rows * F ? e0 + u * 256 : 0];
      for (int q = 1; q < splits; ++q) {
        double w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = xb[q * xs_split + (e0 + u * 256 < rows * F ? e0 + u * 256 : 0)];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] += w[u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (e0 + u * 256 < rows * F) xo[e0 + u * 256] = v[u];
    }
  }
}

__global__ __launch_bounds__(256) void dense64_post_rows_kernel(const double* __restrict__ raw,
                                                                const double* __restrict__ colpart, int K, int flags,
                                                                double eps, double* __restrict__ dst) {
  extern __shared__ double s_d64[];
  const int nb = (K + D64_ROWS - 1) / D64_ROWS;
  const int b = blockIdx.x / nb, rb = blockIdx.x - b * nb;
  const int i0 = rb * D64_ROWS, rows = K - i0 < D64_ROWS ? K - i0 : D64_ROWS;
  const bool rsl = flags & TGP_REMOVE_SELF_LOOPS, dn = (flags & TGP_DEGREE_NORM) && colpart;
  if (dn) {
    for (int j = threadIdx.x; j < K; j += 256) {
      const double* cp = colpart + static_cast<long>(b) * nb * K + j;
      double c = 0.0;
      for (int q = 0; q < nb; ++q) c += cp[static_cast<long>(q) * K];
      s_d64[j] = sqrt(fmax(c, eps));
    }
  }
  __syncthreads();
  const double* a = raw + static_cast<long>(b) * K * K + static_cast<long>(i0) * K;
  double* o = dst + static_cast<long>(b) * K * K + static_cast<long>(i0) * K;
  for (int j = threadIdx.x; j < K; j += 256) {  // a thread per column, the block's rows requested together (raw may
    double v[D64_ROWS];                           // BE dst: loads could not pass the stores of an element loop)
#pragma unroll
    for (int r = 0; r < D64_ROWS; ++r) v[r] = a[static_cast<long>(r < rows ? r : 0) * K + j];
    const double dj = dn ? s_d64[j] : 1.0;
#pragma unroll
    for (int r = 0; r < D64_ROWS; ++r) {
      if (r < rows) {
        double x = v[r];
        if (rsl && i0 + r == j) x = 0.0;
        if (dn) x = (x / dj) / s_d64[i0 + r];
        o[static_cast<long>(r) * K + j] = x;
      }
    }
  }
}

// node-range split of S^T [U | X]: aim at ~3 workgroups per CU, keep >= 4 k-tiles per workgroup
static int splits_for(int64_t batches, int64_t tiles, int64_t span) {
  const int64_t wgs = (batches > 0 ? batches : 1) * (tiles > 0 ? tiles : 1);
  int64_t splits = (3 * 256 + wgs - 1) / wgs;
  const int64_t max_splits = (span + 4 * DBK - 1) / (4 * DBK);
  if (splits > max_splits) splits = max_splits;
  if (splits > 64) splits = 64;
  return splits < 1 ? 1 : static_cast<int>(splits);
}

struct Dense64Plan {
  int splits, k_per_split;
  size_t u, aslab, xslab, deg, colpart;
};

static Dense64Plan dense64_plan(int64_t B, int64_t N, int64_t K, int64_t F) {
  Dense64Plan p;
  const int64_t tiles = ((K + DT - 1) / DT) * (((K + DT - 1) / DT) + ((F + DT - 1) / DT));
  int splits = splits_for(B, tiles, N);
  int64_t kps = ((N + splits - 1) / splits + DBK - 1) / DBK * DBK;
  if (kps < DBK) kps = DBK;
  splits = static_cast<int>((N + kps - 1) / kps);
  if (splits < 1) splits = 1;
  p.splits = splits;
  p.k_per_split = static_cast<int>(kps);
  p.u = static_cast<size_t>(B) * N * K;
  p.aslab = static_cast<size_t>(B) * splits * K * K;
  p.xslab = static_cast<size_t>(B) * splits * K * F;
  p.deg = static_cast<size_t>(B) * K;
  p.colpart = static_cast<size_t>(B) * ((K + 15) / 16) * K;  // dense64_combine_kernel's partial column sums
  return p;
}

// T[i,:] = sum over the entries e of row i (ascending e) of w[e] * S[col[e],:]; one wave per row
__global__ __launch_bounds__(256) void spmm_csr_f64_kernel(const int32_t* __restrict__ row_ptr,
                                                           const int64_t* __restrict__ col,
                                                           const double* __restrict__ w, int64_t num_rows,
                                                           const double* __restrict__ S, int64_t K,
                                                           double* __restrict__ T) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * 4;
  for (int64_t i = wave; i < num_rows; i += nwaves) {
    const int lo = row_ptr[i], hi = row_ptr[i + 1];
    for (int64_t c0 = 0; c0 < K; c0 += 64) {
      const int64_t c = c0 + lane;
      double acc = 0.0;
      for (int e = lo; e < hi; ++e) {
        const double we = w ? w[e] : 1.0;
        const int64_t j = col[e];
        if (c < K) acc += we * S[j * K + c];
      }
      if (c < K) T[i * K + c] = acc;
    }
  }
}

}  // namespace tgp

using namespace tgp;

extern "C" int tgp_postprocess_dense_f64(const double* src, double* dst, int64_t B, int64_t K, int flags, double eps,
                                         void* ws, size_t ws_bytes, void* stream_);
extern "C" size_t tgp_postprocess_dense_workspace_bytes_f64(int64_t B, int64_t K);

extern "C" int tgp_bmm_f64(const double* A, const double* Bm, double* C, int64_t batch, int64_t M, int64_t Nc,
                           int64_t Kd, int trans_a, int64_t lda, int64_t ldb, int64_t ldc, int64_t sA, int64_t sB,
                           int64_t sC, int accumulate, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(batch >= 0 && M >= 0 && Nc >= 0 && Kd >= 0, TGP_ERR_INVALID, "tgp_bmm_f64: negative size");
  if (batch == 0 || M == 0 || Nc == 0) return TGP_OK;
  TGP_REQUIRE(C && (Kd == 0 || (A && Bm)), TGP_ERR_INVALID, "tgp_bmm_f64: null pointer");
  TGP_REQUIRE(M < (1ll << 31) && Nc < (1ll << 31) && Kd < (1ll << 31), TGP_ERR_RANGE, "tgp_bmm_f64: too large");
  TGP_REQUIRE(batch * ((M + DT - 1) / DT) * ((Nc + DT - 1) / DT) < (1ll << 31), TGP_ERR_RANGE,
              "tgp_bmm_f64: grid too large");
  Gemm64Args g{};
  g.A = A; g.lda = lda; g.sA = sA;
  g.M = static_cast<int>(M); g.Kd = static_cast<int>(Kd);
  g.rhs[0] = Gemm64Rhs{Bm, C, static_cast<int>(Nc), ldb, ldc, sB, sC, 0};
  g.splits = 1; g.k_per_split = static_cast<int>((Kd + DBK - 1) / DBK * DBK);
  if (g.k_per_split < DBK) g.k_per_split = DBK;
  g.accumulate = accumulate ? 1 : 0;
  if (trans_a) launch_gemm64<true>(g, static_cast<int>(batch), stream);
  else launch_gemm64<false>(g, static_cast<int>(batch), stream);
  return check_launch("tgp_bmm_f64");
}

extern "C" size_t tgp_dense_pool_workspace_bytes_f64(int64_t B, int64_t N, int64_t K, int64_t F) {
  if (B <= 0 || N <= 0 || K <= 0) return 256;
  const Dense64Plan p = dense64_plan(B, N, K, F > 0 ? F : 0);
  return align_up(p.u * 8) + align_up(p.aslab * 8) + align_up(p.xslab * 8) + align_up(p.deg * 8) +
         align_up(p.colpart * 8) + tgp_postprocess_dense_workspace_bytes_f64(B, K) + 256;
}

extern "C" int tgp_dense_pool_f64(const double* S, const double* A, const double* X, int64_t B, int64_t N, int64_t K,
                                  int64_t F, int flags, double eps, double* x_pool, double* adj_raw, double* adj_pool,
                                  void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(B >= 0 && N >= 0 && K >= 0 && F >= 0, TGP_ERR_INVALID, "tgp_dense_pool_f64: negative size");
  if (B == 0 || K == 0) return TGP_OK;
  const bool want_x = X && x_pool && F > 0;
  const bool want_a = A && (adj_raw || adj_pool);
  TGP_REQUIRE(S || N == 0, TGP_ERR_INVALID, "tgp_dense_pool_f64: S is null");
  TGP_REQUIRE(N < (1ll << 31) && K <= 16000 && F < (1ll << 31) && B < 65536, TGP_ERR_RANGE,
              "tgp_dense_pool_f64: dimension too large");
  if (N == 0) {
    if (want_x) (void)hipMemsetAsync(x_pool, 0, sizeof(double) * B * K * F, stream);
    if (adj_raw) (void)hipMemsetAsync(adj_raw, 0, sizeof(double) * B * K * K, stream);
    if (adj_pool) (void)hipMemsetAsync(adj_pool, 0, sizeof(double) * B * K * K, stream);
    return check_launch("tgp_dense_pool_f64");
  }
  TGP_REQUIRE(ws && ws_bytes >= tgp_dense_pool_workspace_bytes_f64(B, N, K, F), TGP_ERR_WORKSPACE,
              "tgp_dense_pool_f64: workspace too small");
  const Dense64Plan p = dense64_plan(B, N, K, F);
  Carver cv(ws);
  double* U = cv.take<double>(p.u);
  double* aslab = cv.take<double>(p.aslab);
  double* xslab = cv.take<double>(p.xslab);
  double* postws = cv.take<double>(p.deg);
  if (want_a) {  // U[b] = A[b] S[b]
    Gemm64Args g{};
    g.A = A; g.lda = N; g.sA = N * N;
    g.M = static_cast<int>(N); g.Kd = static_cast<int>(N);
    g.rhs[0] = Gemm64Rhs{S, U, static_cast<int>(K), K, K, N * K, N * K, 0};
    g.splits = 1; g.k_per_split = static_cast<int>((N + DBK - 1) / DBK * DBK);
    if (flags & TGP_ADJ_TRANSPOSED) launch_gemm64<true>(g, static_cast<int>(B), stream);
    else launch_gemm64<false>(g, static_cast<int>(B), stream);
  }
  if (want_a || want_x) {  // slabs of S^T [U | X] over slices of the node range: one grid
    const bool direct = p.splits == 1;  // a single slice writes the outputs themselves
    double* a_dst = direct ? (adj_raw ? adj_raw : adj_pool) : aslab;
    double* x_dst = direct ? x_pool : xslab;
    Gemm64Args h{};
    h.A = S; h.lda = K; h.sA = N * K;
    h.M = static_cast<int>(K); h.Kd = static_cast<int>(N);
    h.splits = p.splits; h.k_per_split = p.k_per_split;
    const Gemm64Rhs ra{U, a_dst, static_cast<int>(K), K, K, N * K,
                       direct ? K * K : static_cast<long>(p.splits) * K * K, K * K};
    const Gemm64Rhs rx{X, x_dst, static_cast<int>(F), F, F, N * F,
                       direct ? K * F : static_cast<long>(p.splits) * K * F, K * F};
    if (want_a && want_x) { h.rhs[0] = ra; h.rhs[1] = rx; }
    else h.rhs[0] = want_a ? ra : rx;
    launch_gemm64<true>(h, static_cast<int>(B), stream);
    static const int no_rows = getenv("TGP_NO_POST_ROWS") ? atoi(getenv("TGP_NO_POST_ROWS")) : 0;
    if (!no_rows && want_a && adj_pool && !(flags & TGP_EDGE_WEIGHT_NORM) && K <= 2048 &&
        (!(flags & TGP_DEGREE_NORM) || (flags & TGP_SUM_AXIS_ROWS))) {
      // (1) + (2) above: `raw` is the caller's adj_raw, or adj_pool itself post-processed in place (elementwise per
      // block: every element is read and written by one thread)
      double* raw = adj_raw ? adj_raw : adj_pool;
      const unsigned grid = static_cast<unsigned>(B * ((K + D64_ROWS - 1) / D64_ROWS));
      double* colpart = (flags & TGP_DEGREE_NORM) ? cv.take<double>(p.colpart) : nullptr;
      hipLaunchKernelGGL(dense64_combine_kernel, dim3(grid), dim3(256), 0, stream, direct ? a_dst : aslab,
                         direct ? 1 : p.splits, static_cast<long>(K * K),
                         direct ? static_cast<long>(K * K) : static_cast<long>(p.splits) * K * K,
                         (want_x && !direct) ? xslab : static_cast<const double*>(nullptr), static_cast<long>(K * F),
                         static_cast<long>(p.splits) * K * F, static_cast<int>(K), static_cast<int>(F),
                         (flags & TGP_REMOVE_SELF_LOOPS) ? 1 : 0, direct ? static_cast<double*>(nullptr) : raw, x_pool,
                         colpart);
      hipLaunchKernelGGL(dense64_post_rows_kernel, dim3(grid), dim3(256), static_cast<size_t>(K) * sizeof(double), stream,
                         raw, colpart, static_cast<int>(K), flags, eps, adj_pool);
      return check_launch("tgp_dense_pool_f64");
    }
    if (!direct) {
      if (want_a) {
        const long total = K * K;
        int gx = static_cast<int>((total + 255) / 256);
        if (gx > 256) gx = 256;
        hipLaunchKernelGGL(combine_slabs_f64_kernel, dim3(gx, static_cast<unsigned>(B)), dim3(256), 0, stream, aslab,
                           p.splits, K * K, static_cast<long>(p.splits) * K * K, total, adj_raw ? adj_raw : adj_pool);
      }
      if (want_x) {
        const long total = K * F;
        int gx = static_cast<int>((total + 255) / 256);
        if (gx > 256) gx = 256;
        hipLaunchKernelGGL(combine_slabs_f64_kernel, dim3(gx, static_cast<unsigned>(B)), dim3(256), 0, stream, xslab,
                           p.splits, K * F, static_cast<long>(p.splits) * K * F, total, x_pool);
      }
    }
  }
  if (want_a && adj_pool) {  // A8 on the combined raw product (in place when the raw tensor was not asked for)
    const int rc = tgp_postprocess_dense_f64(adj_raw ? adj_raw : adj_pool, adj_pool, B, K, flags, eps, postws,
                                             tgp_postprocess_dense_workspace_bytes_f64(B, K), stream_);
    if (rc != TGP_OK) return rc;
  }
  return check_launch("tgp_dense_pool_f64");
}

extern "C" size_t tgp_segment_gemm_tn_workspace_bytes_f64(int64_t B, int64_t K, int64_t F, int64_t max_nodes) {
  if (B <= 0 || K <= 0 || F <= 0) return 256;
  const int splits = splits_for(B, ((K + DT - 1) / DT) * ((F + DT - 1) / DT), max_nodes);
  return (splits > 1 ? align_up(static_cast<size_t>(B) * splits * K * F * sizeof(double)) : 0) + 256;
}

extern "C" int tgp_segment_gemm_tn_f64(const double* S, const double* Y, const int64_t* ptr, double* C, int64_t B,
                                       int64_t Ntot, int64_t K, int64_t F, int64_t max_nodes, void* ws,
                                       size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(B >= 0 && Ntot >= 0 && K >= 0 && F >= 0, TGP_ERR_INVALID, "tgp_segment_gemm_tn_f64: negative size");
  if (B == 0 || K == 0 || F == 0) return TGP_OK;
  TGP_REQUIRE(C && ptr && (Ntot == 0 || (S && Y)), TGP_ERR_INVALID, "tgp_segment_gemm_tn_f64: null pointer");
  TGP_REQUIRE(Ntot < (1ll << 31) && K < (1ll << 31) && F < (1ll << 31), TGP_ERR_RANGE,
              "tgp_segment_gemm_tn_f64: too large");
  const int64_t span = max_nodes > 0 ? max_nodes : Ntot;
  const int splits = splits_for(B, ((K + DT - 1) / DT) * ((F + DT - 1) / DT), span);
  TGP_REQUIRE(splits == 1 || (ws && ws_bytes >= tgp_segment_gemm_tn_workspace_bytes_f64(B, K, F, max_nodes)),
              TGP_ERR_WORKSPACE, "tgp_segment_gemm_tn_f64: workspace too small");
  TGP_REQUIRE(B * splits * ((K + DT - 1) / DT) * ((F + DT - 1) / DT) < (1ll << 31), TGP_ERR_RANGE,
              "tgp_segment_gemm_tn_f64: grid too large");
  double* slab = splits > 1 ? static_cast<double*>(ws) : C;
  Gemm64Args g{};
  g.A = S; g.lda = K; g.sA = 0;
  g.M = static_cast<int>(K); g.Kd = static_cast<int>(Ntot);
  g.rhs[0] = Gemm64Rhs{Y, slab, static_cast<int>(F), F, F, 0,
                       splits > 1 ? static_cast<long>(splits) * K * F : K * F, K * F};
  g.splits = splits;
  int64_t kps = ((span + splits - 1) / splits + DBK - 1) / DBK * DBK;
  if (kps < DBK) kps = DBK;
  g.k_per_split = static_cast<int>(kps);
  g.k_ptr = ptr;
  launch_gemm64<true>(g, static_cast<int>(B), stream);
  if (splits > 1) {
    const long total = K * F;
    int gx = static_cast<int>((total + 255) / 256);
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(combine_slabs_f64_kernel, dim3(gx, static_cast<unsigned>(B)), dim3(256), 0, stream, slab, splits,
                       K * F, static_cast<long>(splits) * K * F, total, C);
  }
  return check_launch("tgp_segment_gemm_tn_f64");
}

extern "C" int tgp_segment_gemm_nn_f64(const double* A, const double* Bm, const int64_t* ptr, double* C, int64_t B,
                                       int64_t Ntot, int64_t Kd, int64_t Nc, int64_t max_nodes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(B >= 0 && Ntot >= 0 && Kd >= 0 && Nc >= 0, TGP_ERR_INVALID, "tgp_segment_gemm_nn_f64: negative size");
  if (B == 0 || Ntot == 0 || Nc == 0) return TGP_OK;
  TGP_REQUIRE(C && ptr && (Kd == 0 || (A && Bm)), TGP_ERR_INVALID, "tgp_segment_gemm_nn_f64: null pointer");
  TGP_REQUIRE(Ntot < (1ll << 31) && Kd < (1ll << 31) && Nc < (1ll << 31), TGP_ERR_RANGE,
              "tgp_segment_gemm_nn_f64: too large");
  const int64_t span = max_nodes > 0 ? max_nodes : Ntot;
  TGP_REQUIRE(B * ((span + DT - 1) / DT) * ((Nc + DT - 1) / DT) < (1ll << 31), TGP_ERR_RANGE,
              "tgp_segment_gemm_nn_f64: grid too large");
  Gemm64Args g{};
  g.A = A; g.lda = Kd; g.sA = 0;
  g.M = static_cast<int>(span); g.Kd = static_cast<int>(Kd);
  g.rhs[0] = Gemm64Rhs{Bm, C, static_cast<int>(Nc), Nc, Nc, Kd * Nc, 0, 0};
  g.splits = 1; g.k_per_split = static_cast<int>((Kd + DBK - 1) / DBK * DBK);
  if (g.k_per_split < DBK) g.k_per_split = DBK;
  g.m_ptr = ptr;
  launch_gemm64<false>(g, static_cast<int>(B), stream);
  return check_launch("tgp_segment_gemm_nn_f64");
}

extern "C" int tgp_spmm_csr_f64(const int32_t* row_ptr, const int64_t* col, const double* w, int64_t num_rows,
                                int64_t nnz, const double* S, int64_t K, double* T, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(num_rows >= 0 && nnz >= 0 && K >= 0, TGP_ERR_INVALID, "tgp_spmm_csr_f64: negative size");
  if (num_rows == 0 || K == 0) return TGP_OK;
  TGP_REQUIRE(row_ptr && T && (nnz == 0 || (col && S)), TGP_ERR_INVALID, "tgp_spmm_csr_f64: null pointer");
  int64_t blocks = (num_rows + 3) / 4;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(spmm_csr_f64_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, row_ptr, col, w,
                     num_rows, S, K, T);
  return check_launch("tgp_spmm_csr_f64");
}
