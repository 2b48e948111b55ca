// fp32-MFMA batched GEMM of the dense path (LDS-tiled, buffer-descriptor loads) and its launchers.
//
// GEMM kernel: 128x128 output tile per 512-thread workgroup: 8 waves as 4(M) x 2(N), each wave owns
// 32 x 64 = two 32x32 MFMA tiles (32 accumulator VGPRs); two waves per SIMD so one wave's LDS
// operand fetch hides behind the other's MFMAs.  BK = 32, register-staged double-buffered LDS,
// operands for k-step t+1 are fetched from LDS before the MFMAs of step t are issued.
// fp32 MFMA needs only one operand dword per lane per 64-cycle instruction, so LDS bandwidth is a
// non-issue; layouts are chosen for conflict-free ds_write/ds_read and coalesced global loads:
//   * row-major operand tile  (A of A.S):     LDS [128][BK+1]  (odd stride => conflict free)
//   * k-major operand tile    (S, U, X, A^T): LDS [BK][128]    (lanes read consecutive floats)
#pragma once
#include <stdlib.h>
#include <type_traits>

#include "common.h"

namespace tgp {


typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
#ifndef TGP_OPERAND_PREFETCH
#define TGP_OPERAND_PREFETCH 1
#endif
#ifndef TGP_LOAD_AT
#define TGP_LOAD_AT 7
#endif
#ifndef TGP_STORE_AT
#define TGP_STORE_AT 3
#endif
#ifndef TGP_STEADY_LOOP
#define TGP_STEADY_LOOP 1
#endif
#ifndef TGP_SPREAD
#define TGP_SPREAD 1
#endif
#ifndef TGP_MIRROR_ROWS
#define TGP_MIRROR_ROWS 1
#endif
#ifndef TGP_VEC_EPILOGUE
#define TGP_VEC_EPILOGUE 1
#endif

constexpr int LDA_ROWMAJOR = BK + 1;

// One right-hand side / output pair.  A launch may carry up to three (column tiles >= tiles_n0 use the
// second, those >= tiles_n1 the third), which lets S^T [U | X] -- and, for the training step, S^T [U | X | S] --
// run as a single grid.
struct GemmRhs {
  const float* Bm;
  float* C;
  int Nc;
  long ldb, ldc, sB, sC, sCsplit;
};

struct GemmArgs {
  const float* A;
  long lda, sA;
  int M, Kd;               // C[M,Nc] = op(A)[M,Kd] * Bm[Kd,Nc]
  GemmRhs rhs[3];
  int tiles_m, tiles_n0, tiles_n;  // tiles_n = tiles_n0 + tiles of rhs[1] (+ tiles of rhs[2])
  int tiles_n1;            // first column tile of rhs[2]; 0 = there is no third right-hand side
  int splits;              // split of Kd across workgroups
  int k_per_split;         // multiple of BK
  const int64_t* k_ptr;    // optional [batches+1]: batch b reduces over rows k_ptr[b]..k_ptr[b+1]
  const int64_t* m_ptr;    // optional [batches+1] (row-major A only): batch b owns rows m_ptr[b]..m_ptr[b+1] of
                           // A and C (M = the longest range; sA = sC = 0) -- per-graph products on an un-padded batch
  // MODE 1 only (residual epilogue): nothing is stored; each workgroup writes sum((resid - C)^2) of its tile
  const float* resid;
  long ldr, sR;
  float* partial;          // [batches][tiles_m * tiles_n]
  int force_bm, force_bn;  // 0 = pick_tile decides
  // MODE 1 only, optional [batches]: rows / columns >= sizes[batch] of resid and op(A) Bm^T are zero (padded batch),
  // so tiles that lie entirely beyond them contribute nothing and are skipped
  const int64_t* sizes;
  // MODE 0, splits == 1: C += op(A) Bm (the second term of a two-term gradient lands in the first term's buffer: r4)
  int accumulate;
  // MODE 2 (= MODE 0 + this, r5): [batches][splits][tiles_m][rhs[0].Nc] partial COLUMN sums of the rhs[0] tiles, one
  // value per (row tile, column): the degree vector of the post-processing (utils/ops.py:311-320 with the sum over
  // dim -2) then needs 8 numbers per column instead of a pass over the K x K slabs
  float* colsum;
  int colsum_skip_diag;    // leave C[i][i] out of the sums (the post-processing clears the diagonal before it sums)
  // MODE 1, square tiles, Bm == A (the product is S S^T: symmetric) and resid square (r5): tile (I, J) with I < J also
  // takes the residual of its mirror image -- sum((resid[J-rows][I-cols] - C^T)^2) -- and tiles with I > J do nothing:
  // 36 instead of 64 products per graph at N = 1024
  int symmetric;
};

// Row of a [rows][BK+1] LDS tile served by slot t = 8*g + r (8 lanes per slot, slot = one 128-byte row segment).
// The 8 rows one wave instruction touches are {4g..4g+3} u {4g+32..4g+35}: with the odd row stride their
// ds_write banks (33*row + 4*q + c) mod 64 are all distinct, where 8 consecutive rows would collide 2-way.
__device__ __forceinline__ int tile_row(int t) {
  const int g = t >> 3, r = t & 7;
  return 4 * (g & 7) + (r & 3) + 32 * (r >> 2) + 64 * (g >> 3);
}

#ifdef TGP_GEMM_STAMPS
// Diagnostic build only (make stamps): per-workgroup wall-clock stamps (100 MHz) at kernel entry, after the
// prologue, after the k-loop and after the epilogue, plus the hardware id (XCC / SE / CU) the workgroup ran on.
__device__ unsigned long long* g_gemm_stamps = nullptr;
__device__ int g_gemm_reverse = 0;  // experiment: hand the tiles out in reverse dispatch order
#define TGP_STAMP(slot)                                                                        \
  do {                                                                                          \
    if (g_gemm_stamps && threadIdx.x == 0)                                                      \
      g_gemm_stamps[static_cast<long>(blockIdx.x) * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define TGP_STAMP(slot) do {} while (0)
#endif

__device__ __forceinline__ float4 ld4_guarded(const float* p, bool ok) {
  return ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// Output tile BM x BN per workgroup of (BM/32) x WN waves; each wave owns a 32 x BN/WN strip = NT 32x32 MFMA
// tiles.  Shapes in use: 128 x 128 with 16 waves (4 x 4, NT = 1: 51 VGPRs, two workgroups per CU, A and B tiles
// loaded once per 128 x 128 of output - measured 143 TFLOP/s at C5 against 137 for the 8-wave NT = 2 form),
// 128 x 64 and 64 x 128 with 8 waves, 64 x 64 with 4 waves (when the bigger tiles would leave CUs idle).
// A_KMAJOR = false: A stored [M][Kd] (k contiguous).  true: stored [Kd][M] (m contiguous).
// ALIGNED: buffer-descriptor path, all traffic is 16-byte vectors with one predicate per vector; bases and
// leading dimensions need dword alignment only (see gemm_aligned).  Otherwise: scalar guarded path (matrices too
// large for 32-bit descriptor offsets).
// MODE 0: C = op(A) Bm (MODE 2: the same + GemmArgs.colsum).  MODE 1 (link-prediction residual, utils/losses.py:644-708): Bm is stored
// [Nc][Kd] (n-major, i.e. the product is A Bm^T), and instead of storing C the epilogue accumulates
// sum((resid - C)^2) over the tile, so S S^T never exists in memory.
template <bool A_KMAJOR, bool ALIGNED, int BM, int BN, int MODE, int WN = 2>
__global__ __launch_bounds__(BM * 2 * WN) void gemm_f32_mfma_kernel(GemmArgs g) {
  constexpr int THREADS = BM * 2 * WN;              // (BM/32) x WN waves
  constexpr int NT = BN / (32 * WN);                // 32x32 MFMA tiles per wave (wave strip = 32 x BN/WN)
  constexpr int B_TILE_FLOATS = MODE == 1 ? BN * LDA_ROWMAJOR : BK * BN;
  constexpr int BN_LANES = BN / 4;                  // lanes per k-row of the B tile
  constexpr int A_TILE_FLOATS = BM * LDA_ROWMAJOR;  // >= BK*BM, used for both A layouts
  constexpr int STAGE_FLOATS = A_TILE_FLOATS + B_TILE_FLOATS;
  constexpr int A_VECS = BM * BK / 4 / THREADS;     // float4 per thread per stage (= 2)
  constexpr int B_VECS = BN * BK / 4 / THREADS;     // 2 (512 threads) or 4 (256 threads)
  constexpr int AK_LANES = BM / 4;                  // lanes per k-row of a k-major A tile
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  TGP_STAMP(0);
#ifdef TGP_GEMM_STAMPS
  if (g_gemm_stamps && threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_gemm_stamps[static_cast<long>(blockIdx.x) * 16 + 4] = hw;
    g_gemm_stamps[static_cast<long>(blockIdx.x) * 16 + 5] = xcc;
  }
#endif

  // logical block id, XCD-aware: tiles of one batch element share S / U through one L2
  int bid = xcd_remap(blockIdx.x, gridDim.x);
#ifdef TGP_GEMM_STAMPS
  if (g_gemm_reverse) bid = gridDim.x - 1 - bid;
#endif
  const int tn_all = bid % g.tiles_n; bid /= g.tiles_n;
  const int tm = bid % g.tiles_m; bid /= g.tiles_m;
  const int split = bid % g.splits;
  const int batch = bid / g.splits;
  const int which = tn_all >= g.tiles_n0 ? ((g.tiles_n1 && tn_all >= g.tiles_n1) ? 2 : 1) : 0;
  const int tn = which == 2 ? tn_all - g.tiles_n1 : (which ? tn_all - g.tiles_n0 : tn_all);
  const GemmRhs& R = g.rhs[which];

  const float* __restrict__ A = g.A + static_cast<long>(batch) * g.sA;
  const float* __restrict__ Bm = R.Bm + static_cast<long>(batch) * R.sB;
  float* __restrict__ C = R.C + static_cast<long>(batch) * R.sC + static_cast<long>(split) * R.sCsplit;
  const int Nc = R.Nc;
  const long lda = g.lda, ldb = R.ldb;

  const int m0 = tm * BM, n0 = tn * BN;
  if constexpr (MODE == 1) {
    if (g.sizes) {
      const int64_t nb = g.sizes[batch];
      if (m0 >= nb || n0 >= nb) {  // workgroup-uniform
        if (threadIdx.x == 0)
          g.partial[static_cast<long>(batch) * (g.tiles_m * g.tiles_n) + tm * g.tiles_n + tn_all] = 0.f;
        return;
      }
    }
    if (g.symmetric && m0 > n0) {  // the mirror tile (n0, m0) accounts for this one
      if (threadIdx.x == 0)
        g.partial[static_cast<long>(batch) * (g.tiles_m * g.tiles_n) + tm * g.tiles_n + tn_all] = 0.f;
      return;
    }
  }
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
  const int nk = k_end > k_begin ? (k_end - k_begin + BK - 1) / BK : 0;

  // register staging, two sets: tile t+2 is being loaded into one while tile t+1 is written to LDS from the other
  float4 ra[2][A_VECS], rb[2][B_VECS];

  // ALIGNED path: buffer loads.  Each operand gets a 128-bit resource descriptor (base of this batch element,
  // valid bytes) held in SGPRs, a per-lane byte offset computed ONCE (rows / columns outside the problem get an
  // offset past the end, which the hardware range check turns into zeros), and a scalar offset that advances
  // with the k-step.  The steady-state loop then issues its global loads without a single vector-ALU
  // instruction: measured per-wave time stamps showed ~0.5 us per k-step going to the address / predicate
  // arithmetic of plain global loads, which has to squeeze in between other waves' MFMAs.
  constexpr int OOB = static_cast<int>(0x80000000u);  // >= any valid size (matrices are < 2^31 bytes here)
  [[maybe_unused]] __amdgpu_buffer_rsrc_t rsrc_a, rsrc_b;
  [[maybe_unused]] int voff_a[A_VECS], voff_b[B_VECS];
  [[maybe_unused]] int kloc_a[A_VECS], kloc_b[B_VECS];  // this lane's k offset inside a stage (for the tail)
  if constexpr (ALIGNED) {
    const int a_bytes = A_KMAJOR ? (static_cast<int>(g.Kd - 1) * static_cast<int>(lda) + M) * 4
                                 : (static_cast<int>(M - 1) * static_cast<int>(lda) + g.Kd) * 4;
    const int b_bytes = MODE == 1 ? (static_cast<int>(Nc - 1) * static_cast<int>(ldb) + g.Kd) * 4
                                  : (static_cast<int>(g.Kd - 1) * static_cast<int>(ldb) + Nc) * 4;
    rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, a_bytes, 0x00020000);
    rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bm), 0, b_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < A_VECS; ++i) {
      if constexpr (!A_KMAJOR) {
        const int m = m0 + tile_row((tid >> 3) + i * (THREADS / 8));
        kloc_a[i] = (tid & 7) * 4;
        voff_a[i] = m < M ? (m * static_cast<int>(lda) + kloc_a[i]) * 4 : OOB;
      } else {
        const int m = m0 + (tid % AK_LANES) * 4;
        kloc_a[i] = tid / AK_LANES + i * (THREADS / AK_LANES);
        voff_a[i] = m < M ? (kloc_a[i] * static_cast<int>(lda) + m) * 4 : OOB;
      }
    }
#pragma unroll
    for (int i = 0; i < B_VECS; ++i) {
      if constexpr (MODE == 1) {
        const int n = n0 + tile_row((tid >> 3) + i * (THREADS / 8));
        kloc_b[i] = (tid & 7) * 4;
        voff_b[i] = n < Nc ? (n * static_cast<int>(ldb) + kloc_b[i]) * 4 : OOB;
      } else {
        const int n = n0 + (tid % BN_LANES) * 4;
        kloc_b[i] = tid / BN_LANES + i * (THREADS / BN_LANES);
        voff_b[i] = n < Nc ? (kloc_b[i] * static_cast<int>(ldb) + n) * 4 : OOB;
      }
    }
  }
  auto buf_ld4 = [&](__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
  };

  // one float4 of the A / B stage starting at k0 (tail = this stage is cut short by k_end)
  auto load_a = [&](int i, int k0, bool tail) -> float4 {
    if constexpr (ALIGNED) {
      const int soff = A_KMAJOR ? k0 * static_cast<int>(lda) * 4 : k0 * 4;
      if (!tail) return buf_ld4(rsrc_a, voff_a[i], soff);  // steady state: no per-lane arithmetic at all
      float4 v = buf_ld4(rsrc_a, k0 + kloc_a[i] < k_end ? voff_a[i] : OOB, soff);
      if constexpr (!A_KMAJOR) {  // k runs along the vector: a range that is not a multiple of 4 ends inside one
        const int rem = k_end - (k0 + kloc_a[i]);
        if (rem < 4) { v.w = 0.f; if (rem < 3) v.z = 0.f; if (rem < 2) v.y = 0.f; }
      }
      return v;
    } else {
      int m, k;
      if constexpr (!A_KMAJOR) {  // [BM m][32 k]: 8 lanes cover one 128-byte row segment
        m = m0 + tile_row((tid >> 3) + i * (THREADS / 8));
        k = k0 + (tid & 7) * 4;
      } else {                    // [32 k][BM m]: AK_LANES lanes cover one row
        k = k0 + tid / AK_LANES + i * (THREADS / AK_LANES);
        m = m0 + (tid % AK_LANES) * 4;
      }
      const float* p = A_KMAJOR ? A + static_cast<long>(k) * lda + m : A + static_cast<long>(m) * lda + k;
      float t[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = A_KMAJOR ? (k < k_end && m + j < M) : (m < M && k + j < k_end);
        t[j] = ok ? p[j] : 0.f;
      }
      return make_float4(t[0], t[1], t[2], t[3]);
    }
  };
  auto load_b = [&](int i, int k0, bool tail) -> float4 {
    if constexpr (ALIGNED) {
      const int soff = MODE == 1 ? k0 * 4 : k0 * static_cast<int>(ldb) * 4;
      if (!tail) return buf_ld4(rsrc_b, voff_b[i], soff);
      float4 v = buf_ld4(rsrc_b, k0 + kloc_b[i] < k_end ? voff_b[i] : OOB, soff);
      if constexpr (MODE == 1) {
        const int rem = k_end - (k0 + kloc_b[i]);
        if (rem < 4) { v.w = 0.f; if (rem < 3) v.z = 0.f; if (rem < 2) v.y = 0.f; }
      }
      return v;
    } else {
      float t[4];
      if constexpr (MODE == 1) {        // [BN n][32 k]: 8 lanes cover one 128-byte row segment
        const int n = n0 + tile_row((tid >> 3) + i * (THREADS / 8)), k = k0 + (tid & 7) * 4;
        const float* p = Bm + static_cast<long>(n) * ldb + k;
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = (n < Nc && k + j < k_end) ? p[j] : 0.f;
      } else {                          // [32 k][BN n]: BN_LANES lanes cover one row
        const int k = k0 + tid / BN_LANES + i * (THREADS / BN_LANES), n = n0 + (tid % BN_LANES) * 4;
        const float* p = Bm + static_cast<long>(k) * ldb + n;
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = (k < k_end && n + j < Nc) ? p[j] : 0.f;
      }
      return make_float4(t[0], t[1], t[2], t[3]);
    }
  };
  auto store_a = [&](int i, const float4& v, float* As) {
    if constexpr (!A_KMAJOR) {
      float* d = As + tile_row((tid >> 3) + i * (THREADS / 8)) * LDA_ROWMAJOR + (tid & 7) * 4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    } else {
      *reinterpret_cast<float4*>(As + (tid / AK_LANES + i * (THREADS / AK_LANES)) * BM + (tid % AK_LANES) * 4) = v;
    }
  };
  auto store_b = [&](int i, const float4& v, float* Bs) {
    if constexpr (MODE == 1) {
      float* d = Bs + tile_row((tid >> 3) + i * (THREADS / 8)) * LDA_ROWMAJOR + (tid & 7) * 4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    } else {
      *reinterpret_cast<float4*>(Bs + (tid / BN_LANES + i * (THREADS / BN_LANES)) * BN + (tid % BN_LANES) * 4) = v;
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int lm = lane & 31, lk = lane >> 5;
  // MODE 1: the residual tile is requested before the first k-step and consumed in the epilogue, so its
  // HBM latency hides behind the whole product (with Kd = K small the loop is only a few steps long).
  float rres[MODE == 1 ? NT : 1][16];
  [[maybe_unused]] float rmir[MODE == 1 ? NT : 1][16];  // the mirror tile's residual, transposed into the C/D layout
  [[maybe_unused]] const bool mirror = MODE == 1 && g.symmetric && m0 < n0;
  // the mirror tile read row-wise + an LDS transposition (interior, aligned tiles; one [32][33] patch per wave)
  constexpr bool MIRROR_PATCH_FITS = (THREADS / 64) * (32 * 33) <= 2 * STAGE_FLOATS;
  [[maybe_unused]] const bool mirror_rows = mirror && MIRROR_PATCH_FITS && TGP_MIRROR_ROWS && (g.ldr & 3) == 0 &&
                                            (reinterpret_cast<uintptr_t>(g.resid + static_cast<long>(batch) * g.sR) & 15) == 0 &&
                                            m0 + BM <= M && n0 + BN <= Nc && n0 + BN <= M && m0 + BM <= Nc;
  if constexpr (MODE == 1) {
    const float* __restrict__ Rm = g.resid + static_cast<long>(batch) * g.sR;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = n0 + wn * (BN / WN) + j * 32 + lm;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        rres[j][r] = (row < M && col < Nc) ? Rm[static_cast<long>(row) * g.ldr + col] : 0.f;
      }
    }
    if (mirror && mirror_rows) {
      // r5, late: the mirror tile is read the way it lies in memory -- 8 lanes per 128-byte row segment, a wave's 32 x 32
      // block in four passes -- and turned into the C/D layout through the wave's LDS patch in the epilogue.  Read
      // directly in that layout (below: lane = row of the residual, 16 bytes each) every lane touches another row:
      // 64 separate sectors per instruction for the half of the residual that goes this way.  (Measured at C2, same
      // box: 0.0757 -> 0.0745 ms per link-loss call -- the four passes of a lane cover its sector after all, the L2
      // absorbed the rest; the kernel is bound by its 4352 short-lived workgroups, not by these loads.)
      const float* __restrict__ base = Rm + static_cast<long>(n0 + wn * (BN / WN) + (lane >> 3)) * g.ldr + m0 + wm * 32 +
                                       4 * (lane & 7);
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = *reinterpret_cast<const float4*>(base + static_cast<long>(j * 32 + 8 * q) * g.ldr);
          rmir[j][4 * q] = v.x; rmir[j][4 * q + 1] = v.y; rmir[j][4 * q + 2] = v.z; rmir[j][4 * q + 3] = v.w;
        }
    } else if (mirror) {  // (workgroup-uniform)  resid[col][row]: a lane's four consecutive rows are 16 contiguous bytes
      const bool v4 = (g.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(Rm) & 15) == 0;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = n0 + wn * (BN / WN) + j * 32 + lm;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = m0 + wm * 32 + 8 * q + 4 * lk;
          const float* src = Rm + static_cast<long>(col) * g.ldr + row;
          if (v4 && col < Nc && row + 3 < M) {
            const float4 v = *reinterpret_cast<const float4*>(src);
            rmir[j][4 * q] = v.x; rmir[j][4 * q + 1] = v.y; rmir[j][4 * q + 2] = v.z; rmir[j][4 * q + 3] = v.w;
          } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) rmir[j][4 * q + u] = (col < Nc && row + u < M) ? src[u] : 0.f;
          }
        }
      }
    }
  }
  const int a_off = A_KMAJOR ? lk * BM + wm * 32 + lm : (wm * 32 + lm) * LDA_ROWMAJOR + lk;
  const int a_step = A_KMAJOR ? 2 * BM : 2;
  const int b_off = MODE == 1 ? (wn * (BN / WN) + lm) * LDA_ROWMAJOR + lk : lk * BN + wn * (BN / WN) + lm;
  constexpr int b_step = MODE == 1 ? 2 : 2 * BN;               // one k-pair
  constexpr int b_tile = MODE == 1 ? 32 * LDA_ROWMAJOR : 32;   // next 32 output columns
  // MFMAs of k-pairs [p0, p1) of one LDS stage.
  // LDS operand ring: the operands of k-pair p + PD are requested before the MFMAs of pair p are issued
  // (sched_group_barrier pins that order), so PD MFMA slots of latency are covered.
  constexpr int PD = TGP_OPERAND_PREFETCH;
  float a_r[PD + 1], b_r[PD + 1][NT];
  auto fetch_pair = [&](const float* As, const float* Bs, int p) {
    a_r[p % (PD + 1)] = As[a_off + p * a_step];
#pragma unroll
    for (int j = 0; j < NT; ++j) b_r[p % (PD + 1)][j] = Bs[b_off + p * b_step + b_tile * j];
  };
  auto mfma_pair = [&](int p) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_r[p % (PD + 1)], b_r[p % (PD + 1)][j], acc[j], 0, 0, 0);
  };

  // Stage schedule.  Per-wave time stamps (tools/gemm_stamps.py) showed that a wave loses most of a k-step not
  // in the MFMAs but queueing behind the other 15 waves of the CU whenever all of them issue their global loads
  // or their LDS stores at the same point of the step (the texture-address path takes 16 clk per 1 KB load and a
  // wave cannot issue its next MFMA while it is stuck in that queue).  So the memory work of a stage is dealt
  // out ONE instruction at a time between MFMA pairs:
  //   pair 1: load A0(t+2) | 2: store A0(t+1) | 3: load A1 | 4: store A1 | 5: load B0 | 6: store B0 | ...
  // Tile t+2 is loaded into register set t&1 during stage t, written to LDS from there during stage t+1 (the
  // other LDS buffer than the one being read) and multiplied in stage t+2, so a load has a whole stage to land.
  float* L0 = smem;
  float* L1 = smem + STAGE_FLOATS;
  auto kof = [&](int t) { return k_begin + t * BK; };
  // STEADY (r5, late): the stage as the middle of the k-loop runs it -- tile t + 1 is stored, tile t + 2 is loaded, and
  // that tile is a full one -- with all three facts known at compile time.  The general form decides them per stage,
  // and the waits for the staged registers then sit inside conditional blocks: on the path that skips a block the
  // compiler must assume the loads into a register set are still pending when the next stage overwrites it, and it
  // put an `s_waitcnt vmcnt(0)` in front of the B tile's load -- right behind the two A loads of the same stage, so
  // every wave sat out a full memory round trip per k-step -- plus ~25 branches per step.
  auto stage = [&](auto par_c, auto steady_c, int t) {
    constexpr int PAR = decltype(par_c)::value;
    constexpr bool STEADY = decltype(steady_c)::value;
    const float* As = PAR ? L1 : L0;
    const float* Bs = As + A_TILE_FLOATS;
    float* An = PAR ? L0 : L1;
    float* Bn = An + A_TILE_FLOATS;
    const bool do_store = STEADY || t + 1 < nk, do_load = STEADY || t + 2 < nk;
    const int k2 = kof(t + 2);
    const bool tail = !STEADY && k2 + BK > k_end;
#ifdef TGP_GEMM_STAMPS
    // phase clock of one k-step in the middle of the loop (wave 0 of every workgroup): slots 8.. hold the time at
    // step start, before / after the LDS stores, before / after the global loads, before / after the barrier
    const bool probe = g_gemm_stamps && t == (nk / 2) && threadIdx.x == 0;
#define TGP_PHASE(i) do { if (probe) g_gemm_stamps[static_cast<long>(blockIdx.x) * 16 + 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TGP_PHASE(i) do {} while (0)
#endif
    TGP_PHASE(0);
#pragma unroll
    for (int p = 0; p < PD; ++p) fetch_pair(As, Bs, p);
#pragma unroll
    for (int p = 0; p < BK / 2; ++p) {
      if (p + PD < BK / 2) fetch_pair(As, Bs, p + PD);
      mfma_pair(p);
      if (p + PD < BK / 2) __builtin_amdgcn_sched_group_barrier(0x100, 1 + NT, 0);  // ds_reads of pair p+PD
      __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);                            // NT x MFMA
      // memory work of the stage: loads of tile t+2 behind pair LOAD_AT, LDS stores of tile t+1 behind STORE_AT
      // (SPREAD: one vector per pair starting there, instead of all at once)
      constexpr int LOAD_AT = TGP_LOAD_AT, STORE_AT = TGP_STORE_AT, SPREAD = TGP_SPREAD;
      if (p == STORE_AT) TGP_PHASE(1);
      if (p == LOAD_AT) TGP_PHASE(3);
#pragma unroll
      for (int v = 0; v < A_VECS + B_VECS; ++v) {
        if (p == LOAD_AT + (SPREAD ? v : 0) && do_load) {
          if (v < A_VECS) ra[PAR][v] = load_a(v, k2, tail);
          else rb[PAR][v - A_VECS] = load_b(v - A_VECS, k2, tail);
        }
        if (p == STORE_AT + (SPREAD ? v : 0) && do_store) {
          if (v < A_VECS) store_a(v, ra[1 - PAR][v], An);
          else store_b(v - A_VECS, rb[1 - PAR][v - A_VECS], Bn);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (p == STORE_AT) TGP_PHASE(2);
      if (p == LOAD_AT) TGP_PHASE(4);
    }
    TGP_PHASE(5);
    __syncthreads();
    TGP_PHASE(6);
  };
  if (nk > 0) {
#pragma unroll
    for (int i = 0; i < A_VECS; ++i) ra[0][i] = load_a(i, kof(0), kof(0) + BK > k_end);
#pragma unroll
    for (int i = 0; i < B_VECS; ++i) rb[0][i] = load_b(i, kof(0), kof(0) + BK > k_end);
  }
  if (nk > 1) {
#pragma unroll
    for (int i = 0; i < A_VECS; ++i) ra[1][i] = load_a(i, kof(1), kof(1) + BK > k_end);
#pragma unroll
    for (int i = 0; i < B_VECS; ++i) rb[1][i] = load_b(i, kof(1), kof(1) + BK > k_end);
  }
  if (nk > 0) {
#pragma unroll
    for (int i = 0; i < A_VECS; ++i) store_a(i, ra[0][i], L0);
#pragma unroll
    for (int i = 0; i < B_VECS; ++i) store_b(i, rb[0][i], L0 + A_TILE_FLOATS);
  }
  __syncthreads();
  TGP_STAMP(1);
  int t = 0;
  using std::integral_constant;
  if constexpr (ALIGNED && TGP_STEADY_LOOP) {
    // pairs of stages whose loaded tiles (t + 2 and t + 3) exist and are full
    for (; t + 3 < nk && kof(t + 3) + BK <= k_end; t += 2) {
#ifdef TGP_GEMM_STAMPS
      if (t == (nk / 2 & ~1)) TGP_STAMP(6);
      if (t == ((3 * nk) / 4 & ~1)) TGP_STAMP(7);
#endif
      stage(integral_constant<int, 0>{}, integral_constant<bool, true>{}, t);
      stage(integral_constant<int, 1>{}, integral_constant<bool, true>{}, t + 1);
    }
  }
  for (; t < nk; t += 2) {
#ifdef TGP_GEMM_STAMPS
    if (t == (nk / 2 & ~1)) TGP_STAMP(6);
    if (t == ((3 * nk) / 4 & ~1)) TGP_STAMP(7);
#endif
    stage(integral_constant<int, 0>{}, integral_constant<bool, false>{}, t);
    if (t + 1 < nk) stage(integral_constant<int, 1>{}, integral_constant<bool, false>{}, t + 1);
  }
  TGP_STAMP(2);

  // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -----------
  if constexpr (MODE == 1) {
    // rows / columns past the edge: operands were zero-filled, so acc = 0 = rres there
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = rres[j][r] - acc[j][r];
        sq = fmaf(d, d, sq);
      }
    if (mirror && mirror_rows) {  // (workgroup-uniform; the k-loop's last barrier has passed: the LDS is free)
      float* patch = smem + wave * (32 * 33);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // patch[a][b] = resid[n0' + a][m0' + b]
          float* d = patch + ((lane >> 3) + 8 * q) * 33 + 4 * (lane & 7);
          d[0] = rmir[j][4 * q]; d[1] = rmir[j][4 * q + 1]; d[2] = rmir[j][4 * q + 2]; d[3] = rmir[j][4 * q + 3];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = patch[lm * 33 + (r & 3) + 8 * (r >> 2) + 4 * lk] - acc[j][r];
          sq = fmaf(d, d, sq);
        }
        __builtin_amdgcn_wave_barrier();
      }
    } else if (mirror) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = rmir[j][r] - acc[j][r];
          sq = fmaf(d, d, sq);
        }
    }
    // fixed-order reduction: lanes (xor butterfly) -> waves (LDS, summed in wave order)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    __syncthreads();
    if (lane == 0) smem[wave] = sq;
    __syncthreads();
    if (tid == 0) {
      float t = 0.f;
      for (int w = 0; w < THREADS / 64; ++w) t += smem[w];
      g.partial[static_cast<long>(batch) * (g.tiles_m * g.tiles_n) + tm * g.tiles_n + tn_all] = t;
    }
    return;
  }
  // Interior tiles of an aligned C leave as 16-byte stores (r5): a wave's 32 x 32 accumulator tile goes through its own
  // LDS patch [32][36] (the k-loop's last barrier has passed) and 8 lanes write one 128-byte row segment: 4 store
  // instructions per lane and tile where the scalar form issues 16 -- with 16 waves per CU the stores' address path
  // was ~1.5 us of the launch.  Edge tiles, C += and unaligned C keep the scalar stores.
  constexpr bool PATCH_FITS = (THREADS / 64) * (32 * 36) <= 2 * STAGE_FLOATS;  // (not the 16-wave 128 x 128 tile)
  const bool vec_store = ALIGNED && PATCH_FITS && TGP_VEC_EPILOGUE && !g.accumulate && (R.ldc & 3) == 0 &&
                         (reinterpret_cast<uintptr_t>(C) & 15) == 0 && m0 + BM <= M && n0 + BN <= Nc;
  if (vec_store) {  // (workgroup-uniform)
    float* patch = smem + wave * (32 * 36);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lk) * 36 + lm] = acc[j][r];
      __builtin_amdgcn_wave_barrier();
      const int c4 = (lane & 7) * 4, r8 = lane >> 3;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rr = r8 + 8 * q;
        const float4 v = *reinterpret_cast<const float4*>(patch + rr * 36 + c4);
        *reinterpret_cast<float4*>(C + static_cast<long>(m0 + wm * 32 + rr) * R.ldc + n0 + wn * (BN / WN) + j * 32 + c4) = v;
      }
      __builtin_amdgcn_wave_barrier();
    }
  } else {
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = n0 + wn * (BN / WN) + j * 32 + lm;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      if (row < M && col < Nc) {
        float* dst = C + static_cast<long>(row) * R.ldc + col;
        *dst = g.accumulate ? __fadd_rn(*dst, acc[j][r]) : acc[j][r];
      }
    }
  }
  }
  if constexpr (MODE == 2) {
    if (vec_store) __syncthreads();  // (the patches are read; the column sums below reuse the same LDS)
    // column sums of this tile (rows beyond M are zero: the operands were zero-filled): registers in order, the two
    // half-waves, then the row waves in order through LDS (the k-loop's last barrier has passed: smem is free)
    if (g.colsum && which == 0) {  // (workgroup-uniform)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        float cs = 0.f;
        const int dcol = n0 + wn * (BN / WN) + j * 32 + lm - (m0 + wm * 32 + 4 * lk);  // row offset of the diagonal
#pragma unroll
        for (int r = 0; r < 16; ++r)
          cs = __fadd_rn(cs, (g.colsum_skip_diag && (r & 3) + 8 * (r >> 2) == dcol) ? 0.f : acc[j][r]);
        cs = __fadd_rn(cs, __shfl_xor(cs, 32, 64));
        if (lk == 0) smem[wm * BN + wn * (BN / WN) + j * 32 + lm] = cs;
      }
      __syncthreads();
      if (tid < BN && n0 + tid < Nc) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < BM / 32; ++q) tot = __fadd_rn(tot, smem[q * BN + tid]);
        g.colsum[((static_cast<long>(batch) * g.splits + split) * g.tiles_m + tm) * Nc + n0 + tid] = tot;
      }
    }
  }
  TGP_STAMP(3);
}

// The buffer-load path needs dword alignment only (compute queues run in unaligned-access mode, so a 16-byte
// load may start on any dword): a vector that runs past the end of its row picks up the head of the next row,
// which is either masked (k tail, see load_a / load_b) or lands in output rows / columns that are never stored;
// past the end of the matrix the descriptor's per-dword range check returns zeros.  What remains a requirement
// is that every matrix fits the 32-bit byte offsets of a descriptor.  TGP_GEMM_SCALAR=1 forces the guarded
// scalar path (diagnostic).
static bool gemm_aligned(const GemmArgs& g, bool /*k_rows*/) {
  static const bool force_scalar = getenv("TGP_GEMM_SCALAR") && atoi(getenv("TGP_GEMM_SCALAR"));
  if (force_scalar) return false;
  auto ok = [](const void* p, long, long) { return reinterpret_cast<uintptr_t>(p) % 4 == 0; };
  bool a = ok(g.A, g.lda, g.sA);
  const long lim = (1l << 31) - 4096;
  a = a && (static_cast<long>(g.M) * g.lda * 4 < lim) && (static_cast<long>(g.Kd) * g.lda * 4 < lim);
  for (int w = 0; w < 3; ++w)
    if (w == 0 || (w == 1 && g.tiles_n > g.tiles_n0) || (w == 2 && g.tiles_n1))
      a = a && ok(g.rhs[w].Bm, g.rhs[w].ldb, g.rhs[w].sB) &&
          (static_cast<long>(g.Kd) * g.rhs[w].ldb * 4 < lim) && (static_cast<long>(g.rhs[w].Nc) * g.rhs[w].ldb * 4 < lim);
  return a;
}

// Tile shape: 128 x 128 unless that leaves the chip under two 512-thread workgroups per CU; then the
// tile is halved (128 x 64 first: twice the waves per SIMD at the same A traffic per CU-pair sharing
// an L2; 64 x 128 for short M).
struct TileCfg { int bm, bn; };
static TileCfg pick_tile(int64_t M, int64_t max_nc, int64_t batches_x_splits, const GemmArgs& g) {
  static const int fbm = getenv("TGP_GEMM_BM") ? atoi(getenv("TGP_GEMM_BM")) : 0;
  static const int fbn = getenv("TGP_GEMM_BN") ? atoi(getenv("TGP_GEMM_BN")) : 0;
  TileCfg t{128, 128};
  auto count = [&](int bm, int bn) {
    int64_t tn = 0;
    tn += (g.rhs[0].Nc + bn - 1) / bn;
    if (g.rhs[1].Bm) tn += (g.rhs[1].Nc + bn - 1) / bn;
    if (g.rhs[1].Bm && g.rhs[2].Bm) tn += (g.rhs[2].Nc + bn - 1) / bn;
    return ((M + bm - 1) / bm) * tn * batches_x_splits;
  };
  if (M <= 64) t.bm = 64;
  if (count(t.bm, 128) < 2 * 256 && max_nc >= 64) t.bn = 64;
  if (count(t.bm, t.bn) < 2 * 256 && t.bm == 128 && M > 64) t.bm = 64;
  if (g.force_bm) t.bm = g.force_bm;
  if (g.force_bn) t.bn = g.force_bn;
  if (fbm == 64 || fbm == 128) t.bm = fbm;
  if (fbn == 64 || fbn == 128) t.bn = fbn;
  return t;
}

template <bool A_KMAJOR, int BM, int BN, int MODE = 0, int WN = 2>
static void launch_gemm_cfg(const GemmArgs& g_in, int batches, hipStream_t stream) {
  const GemmArgs& g = g_in;
  const int nwg = batches * g.splits * g.tiles_m * g.tiles_n;
  const size_t lds = 2 * (BM * LDA_ROWMAJOR + (MODE == 1 ? BN * LDA_ROWMAJOR : BK * BN)) * sizeof(float);
  if (gemm_aligned(g, A_KMAJOR && MODE == 0))
    hipLaunchKernelGGL((gemm_f32_mfma_kernel<A_KMAJOR, true, BM, BN, MODE, WN>), dim3(nwg), dim3(BM * 2 * WN), lds, stream, g);
  else
    hipLaunchKernelGGL((gemm_f32_mfma_kernel<A_KMAJOR, false, BM, BN, MODE, WN>), dim3(nwg), dim3(BM * 2 * WN), lds, stream, g);
}

// g.tiles_* are filled in here: they follow from the tile shape chosen for this problem.  Returns whether g.colsum was
// written (only the 64 x 64 tile has the MODE 2 epilogue; *tiles_m_out = row tiles per batch element of that layout).
template <bool A_KMAJOR>
static bool launch_gemm(GemmArgs g, int batches, hipStream_t stream, int* tiles_m_out = nullptr) {
  int64_t max_nc = g.rhs[1].Bm && g.rhs[1].Nc > g.rhs[0].Nc ? g.rhs[1].Nc : g.rhs[0].Nc;
  const bool third = g.rhs[1].Bm && g.rhs[2].Bm;  // (a third right-hand side only behind a second one)
  if (third && g.rhs[2].Nc > max_nc) max_nc = g.rhs[2].Nc;
  const TileCfg t = pick_tile(g.M, max_nc, static_cast<int64_t>(batches) * g.splits, g);
  g.tiles_m = cdiv(g.M, t.bm);
  g.tiles_n0 = cdiv(g.rhs[0].Nc, t.bn);
  g.tiles_n = g.tiles_n0 + (g.rhs[1].Bm ? cdiv(g.rhs[1].Nc, t.bn) : 0);
  g.tiles_n1 = 0;
  if (third) {
    g.tiles_n1 = g.tiles_n;
    g.tiles_n += cdiv(g.rhs[2].Nc, t.bn);
  }
  if (tiles_m_out) *tiles_m_out = g.tiles_m;
  if (t.bm == 64 && t.bn == 64 && g.colsum) {
    launch_gemm_cfg<A_KMAJOR, 64, 64, 2>(g, batches, stream);
    return true;
  }
  if (t.bm == 64 && t.bn == 64) launch_gemm_cfg<A_KMAJOR, 64, 64>(g, batches, stream);
  else if (t.bm == 64) launch_gemm_cfg<A_KMAJOR, 64, 128>(g, batches, stream);
  else if (t.bn == 64) launch_gemm_cfg<A_KMAJOR, 128, 64>(g, batches, stream);
  else launch_gemm_cfg<A_KMAJOR, 128, 128, 0, 4>(g, batches, stream);  // 16 waves (4 x 4), one 32x32 tile each
  return false;
}

// MODE 1 launch: sum((resid - A Bm^T)^2) per tile into g.partial; returns tiles per batch element.
static int launch_gemm_residual(GemmArgs g, int batches, hipStream_t stream, bool dry_run = false) {
  TileCfg t = pick_tile(g.M, g.rhs[0].Nc, batches, g);
  // a short k-loop (Kd <= 8 stages: the link loss at K <= 256) is all prologue, residual loads and epilogue: 64 x 64
  // tiles keep four workgroups per CU in different phases (C2, symmetric form: 0.0903 -> 0.0764 ms for the call)
  if (g.symmetric && g.Kd <= 8 * BK && !g.force_bm && !g.force_bn && t.bm == 128 && t.bn == 128) t = TileCfg{64, 64};
  g.tiles_m = cdiv(g.M, t.bm);
  g.tiles_n0 = g.tiles_n = cdiv(g.rhs[0].Nc, t.bn);
  if (dry_run) return g.tiles_m * g.tiles_n;
  static const int no_sym = getenv("TGP_LINK_LOSS_FULL") ? atoi(getenv("TGP_LINK_LOSS_FULL")) : 0;
  if (t.bm != t.bn || g.M != g.rhs[0].Nc || g.rhs[0].Bm != g.A || g.rhs[0].sB != g.sA || g.rhs[0].ldb != g.lda || no_sym)
    g.symmetric = 0;  // (only S S^T against a square residual on square tiles)
  if (t.bm == 64 && t.bn == 64) launch_gemm_cfg<false, 64, 64, 1>(g, batches, stream);
  else if (t.bm == 64) launch_gemm_cfg<false, 64, 128, 1>(g, batches, stream);
  else if (t.bn == 64) launch_gemm_cfg<false, 128, 64, 1>(g, batches, stream);
  else launch_gemm_cfg<false, 128, 128, 1, 4>(g, batches, stream);
  return g.tiles_m * g.tiles_n;
}

}  // namespace tgp
