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
// GEMM kernel: 128x128 output tile per 512-thread workgroup: 8 waves as 4(M) x 2(N), each wave owns
// 32 x 64 = two 32x32 MFMA tiles (32 accumulator VGPRs); two waves per SIMD so one wave's LDS
// operand fetch hides behind the other's MFMAs.  BK = 32, register-staged double-buffered LDS,
// operands for k-step t+1 are fetched from LDS before the MFMAs of step t are issued.
// fp32 MFMA needs only one operand dword per lane per 64-cycle instruction, so LDS bandwidth is a
// non-issue; layouts are chosen for conflict-free ds_write/ds_read and coalesced global loads:
//   * row-major operand tile  (A of A.S):     LDS [128][BK+1]  (odd stride => conflict free)
//   * k-major operand tile    (S, U, X, A^T): LDS [BK][128]    (lanes read consecutive floats)
#include "common.h"
#include <stdlib.h>
#include <type_traits>

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
#ifndef TGP_SPREAD
#define TGP_SPREAD 0
#endif

constexpr int LDA_ROWMAJOR = BK + 1;

// One right-hand side / output pair.  A launch may carry two (column tiles >= tiles_n0 use the
// second), which lets S^T [U | X] run as a single grid.
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
  GemmRhs rhs[2];
  int tiles_m, tiles_n0, tiles_n;  // tiles_n = tiles_n0 + tiles of rhs[1]
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
// MODE 0: C = op(A) Bm.  MODE 1 (link-prediction residual, utils/losses.py:644-708): Bm is stored
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
  const int which = tn_all >= g.tiles_n0 ? 1 : 0;
  const int tn = which ? tn_all - g.tiles_n0 : tn_all;
  const GemmRhs& R = g.rhs[which];

  const float* __restrict__ A = g.A + static_cast<long>(batch) * g.sA;
  const float* __restrict__ Bm = R.Bm + static_cast<long>(batch) * R.sB;
  float* __restrict__ C = R.C + static_cast<long>(batch) * R.sC + static_cast<long>(split) * R.sCsplit;
  const int Nc = R.Nc;
  const long lda = g.lda, ldb = R.ldb;

  const int m0 = tm * BM, n0 = tn * BN;
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
  auto stage = [&](auto par_c, int t) {
    constexpr int PAR = decltype(par_c)::value;
    const float* As = PAR ? L1 : L0;
    const float* Bs = As + A_TILE_FLOATS;
    float* An = PAR ? L0 : L1;
    float* Bn = An + A_TILE_FLOATS;
    const bool do_store = t + 1 < nk, do_load = t + 2 < nk;
    const int k2 = kof(t + 2);
    const bool tail = k2 + BK > k_end;
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
  for (int t = 0; t < nk; t += 2) {
#ifdef TGP_GEMM_STAMPS
    if (t == (nk / 2 & ~1)) TGP_STAMP(6);
    if (t == ((3 * nk) / 4 & ~1)) TGP_STAMP(7);
#endif
    stage(std::integral_constant<int, 0>{}, t);
    if (t + 1 < nk) stage(std::integral_constant<int, 1>{}, t + 1);
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
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = n0 + wn * (BN / WN) + j * 32 + lm;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      if (row < M && col < Nc) C[static_cast<long>(row) * R.ldc + col] = acc[j][r];
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
  for (int w = 0; w < 2; ++w)
    if (w == 0 || g.tiles_n > g.tiles_n0)
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

// g.tiles_* are filled in here: they follow from the tile shape chosen for this problem.
template <bool A_KMAJOR>
static void launch_gemm(GemmArgs g, int batches, hipStream_t stream) {
  const int64_t max_nc = g.rhs[1].Bm && g.rhs[1].Nc > g.rhs[0].Nc ? g.rhs[1].Nc : g.rhs[0].Nc;
  const TileCfg t = pick_tile(g.M, max_nc, static_cast<int64_t>(batches) * g.splits, g);
  g.tiles_m = cdiv(g.M, t.bm);
  g.tiles_n0 = cdiv(g.rhs[0].Nc, t.bn);
  g.tiles_n = g.tiles_n0 + (g.rhs[1].Bm ? cdiv(g.rhs[1].Nc, t.bn) : 0);
  if (t.bm == 64 && t.bn == 64) launch_gemm_cfg<A_KMAJOR, 64, 64>(g, batches, stream);
  else if (t.bm == 64) launch_gemm_cfg<A_KMAJOR, 64, 128>(g, batches, stream);
  else if (t.bn == 64) launch_gemm_cfg<A_KMAJOR, 128, 64>(g, batches, stream);
  else launch_gemm_cfg<A_KMAJOR, 128, 128, 0, 4>(g, batches, stream);  // 16 waves (4 x 4), one 32x32 tile each
}

// MODE 1 launch: sum((resid - A Bm^T)^2) per tile into g.partial; returns tiles per batch element.
static int launch_gemm_residual(GemmArgs g, int batches, hipStream_t stream, bool dry_run = false) {
  const TileCfg t = pick_tile(g.M, g.rhs[0].Nc, batches, g);
  g.tiles_m = cdiv(g.M, t.bm);
  g.tiles_n0 = g.tiles_n = cdiv(g.rhs[0].Nc, t.bn);
  if (dry_run) return g.tiles_m * g.tiles_n;
  if (t.bm == 64 && t.bn == 64) launch_gemm_cfg<false, 64, 64, 1>(g, batches, stream);
  else if (t.bm == 64) launch_gemm_cfg<false, 64, 128, 1>(g, batches, stream);
  else if (t.bn == 64) launch_gemm_cfg<false, 128, 64, 1>(g, batches, stream);
  else launch_gemm_cfg<false, 128, 128, 1, 4>(g, batches, stream);
  return g.tiles_m * g.tiles_n;
}

// ------------------------------------------------------------------------------------------
// Slab combine + utils/ops.py:282-335 (diag <- 0, D^-1/2 A D^-1/2, / max|A| per graph).
// [B,K,K] is tiny next to A, so these are throughput-shaped elementwise / small-reduction
// kernels with many workgroups; every reduction has a fixed order (no float atomics).
//   pass 1  post_combine_kernel : sum the split-K slabs -> raw, diag-cleared dst
//   pass 2  post_degree_kernel  : d = sqrt(clamp(sum over axis, eps))           (degree_norm)
//   pass 3  post_scale_kernel   : (a / d) / d^T, per-block max|.|
//   pass 4  post_maxnorm_kernel : divide by the per-graph max                   (edge_weight_norm)
// ------------------------------------------------------------------------------------------
constexpr int POST_BLOCKS = 64; // workgroups per graph in the elementwise passes

struct PostArgs {
  const float* src;  // [B][splits][K][ld_src]
  int splits;
  long s_split, s_batch, ld_src;
  int K, flags;
  float* raw;        // optional [B][K][K]
  float* dst;        // optional [B][K][K]
  float* dvec;       // [B][K]
  float* maxpart;    // [B][POST_BLOCKS]
};

template <int VEC>
__global__ __launch_bounds__(256) void post_combine_kernel(PostArgs p) {
  const int b = blockIdx.y, K = p.K;
  const float* sb = p.src + static_cast<long>(b) * p.s_batch;
  float* rawb = p.raw ? p.raw + static_cast<long>(b) * K * K : nullptr;
  float* dstb = p.dst ? p.dst + static_cast<long>(b) * K * K : nullptr;
  const long groups = static_cast<long>(K) * K / VEC;
  for (long gidx = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; gidx < groups;
       gidx += static_cast<long>(gridDim.x) * 256) {
    const long e = gidx * VEC;
    const int i = static_cast<int>(e / K), j = static_cast<int>(e - static_cast<long>(i) * K);
    const long o = static_cast<long>(i) * p.ld_src + j;
    float v[VEC];
    if constexpr (VEC == 4) {
      float4 t = *reinterpret_cast<const float4*>(sb + o);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      for (int s = 1; s < p.splits; ++s) {
        t = *reinterpret_cast<const float4*>(sb + s * p.s_split + o);
        v[0] = __fadd_rn(v[0], t.x); v[1] = __fadd_rn(v[1], t.y);
        v[2] = __fadd_rn(v[2], t.z); v[3] = __fadd_rn(v[3], t.w);
      }
      if (rawb) *reinterpret_cast<float4*>(rawb + e) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      v[0] = sb[o];
      for (int s = 1; s < p.splits; ++s) v[0] = __fadd_rn(v[0], sb[s * p.s_split + o]);
      if (rawb) rawb[e] = v[0];
    }
    if (dstb) {
      if (p.flags & TGP_REMOVE_SELF_LOOPS) {
#pragma unroll
        for (int q = 0; q < VEC; ++q)
          if (i == j + q) v[q] = 0.f;
      }
      if constexpr (VEC == 4) *reinterpret_cast<float4*>(dstb + e) = make_float4(v[0], v[1], v[2], v[3]);
      else dstb[e] = v[0];
    }
  }
}

// grid (ceil(K/64), B), 1024 threads = 16 waves.  Column sums (axis -2): lane = column, wave w adds rows
// w, w+16, ... in order, then the 16 wave partials are added in order.  Row sums (axis -1): the block owns
// 64 rows, wave w rows w, w+16, ...: lanes stride over the columns, fixed shuffle tree.
__global__ __launch_bounds__(1024) void post_degree_kernel(PostArgs p) {
  __shared__ float s_part[16][64];
  const int b = blockIdx.y, K = p.K, lane = lane_id(), w = wave_id();
  const float* a = p.dst + static_cast<long>(b) * K * K;
  const int base = blockIdx.x * 64;
  if (p.flags & TGP_SUM_AXIS_ROWS) {
    const int j = base + lane;
    float s = 0.f;
    if (j < K)
      for (int i = w; i < K; i += 16) s = __fadd_rn(s, a[static_cast<long>(i) * K + j]);
    s_part[w][lane] = s;
    __syncthreads();
    if (w == 0 && j < K) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) t = __fadd_rn(t, s_part[q][lane]);
      p.dvec[static_cast<long>(b) * K + j] = sqrtf(fmaxf(t, TGP_EPS));  // sqrt(clamp(d, eps)): ops.py:318
    }
  } else {
    for (int r = w; r < 64; r += 16) {
      const int i = base + r;
      if (i >= K) break;
      float s = 0.f;
      for (int j = lane; j < K; j += 64) s = __fadd_rn(s, a[static_cast<long>(i) * K + j]);
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) s = __fadd_rn(s, __shfl_down(s, d, WAVE));
      if (lane == 0) p.dvec[static_cast<long>(b) * K + i] = sqrtf(fmaxf(s, TGP_EPS));
    }
  }
}

// grid (POST_BLOCKS, B): contiguous element range per workgroup
__global__ __launch_bounds__(256) void post_scale_kernel(PostArgs p) {
  __shared__ float s_max[4];
  const int b = blockIdx.y, K = p.K, tid = threadIdx.x;
  float* a = p.dst + static_cast<long>(b) * K * K;
  const float* dv = p.dvec + static_cast<long>(b) * K;
  const int rows_per = (K + POST_BLOCKS - 1) / POST_BLOCKS;  // a contiguous band of rows per workgroup
  const int r_lo = blockIdx.x * rows_per, r_hi = min(K, r_lo + rows_per);
  const bool by_cols = p.flags & TGP_SUM_AXIS_ROWS;
  float mx = 0.f;
  for (int i = r_lo; i < r_hi; ++i) {
    float* row = a + static_cast<long>(i) * K;
    const float di = dv[i];
    for (int j = tid; j < K; j += 256) {
      float v = row[j];
      if (p.flags & TGP_DEGREE_NORM) {
        // (adj / d) / d^T with d shaped [1,K] (axis -2) or [K,1] (axis -1): ops.py:319
        const float dj = dv[j];
        v = by_cols ? (v / dj) / di : (v / di) / dj;
        row[j] = v;
      }
      mx = fmaxf(mx, fabsf(v));
    }
  }
  if (p.flags & TGP_EDGE_WEIGHT_NORM) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_down(mx, d, WAVE));
    if (lane_id() == 0) s_max[wave_id()] = mx;
    __syncthreads();
    if (tid == 0)
      p.maxpart[static_cast<long>(b) * POST_BLOCKS + blockIdx.x] =
          fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
  }
}

__global__ __launch_bounds__(256) void post_maxnorm_kernel(PostArgs p) {
  const int b = blockIdx.y, K = p.K, tid = threadIdx.x;
  float m = 0.f;
#pragma unroll
  for (int q = 0; q < POST_BLOCKS; ++q) m = fmaxf(m, p.maxpart[static_cast<long>(b) * POST_BLOCKS + q]);
  if (m == 0.f) m = 1.f;
  float* a = p.dst + static_cast<long>(b) * K * K;
  const long kk = static_cast<long>(K) * K;
  const long per = (kk + POST_BLOCKS - 1) / POST_BLOCKS;
  const long lo = blockIdx.x * per, hi = min(kk, lo + per);
  for (long e = lo + tid; e < hi; e += 256) a[e] = a[e] / m;
}

// K <= 64: one wave per graph (lane = column); the K x K matrix stays L1/L2 resident across the passes.
__global__ __launch_bounds__(256) void post_small_kernel(PostArgs p, int B) {
  const int lane = lane_id(), K = p.K;
  const int b = blockIdx.x * 4 + wave_id();
  if (b >= B) return;
  const float* sb = p.src + static_cast<long>(b) * p.s_batch;
  float* rawb = p.raw ? p.raw + static_cast<long>(b) * K * K : nullptr;
  float* dstb = p.dst ? p.dst + static_cast<long>(b) * K * K : nullptr;
  const bool col_ok = lane < K;
  const bool rsl = p.flags & TGP_REMOVE_SELF_LOOPS;
  auto combined = [&](int i, int j) {  // fixed-order slab sum of element (i, j)
    const long o = static_cast<long>(i) * p.ld_src + j;
    float t = sb[o];
    for (int s = 1; s < p.splits; ++s) t = __fadd_rn(t, sb[s * p.s_split + o]);
    return t;
  };
  // pass 1: raw output, diag-cleared copy, column sums
  float colsum = 0.f;
  if (col_ok) {
    for (int i = 0; i < K; ++i) {
      float t = combined(i, lane);
      if (rawb) rawb[i * K + lane] = t;
      if (rsl && i == lane) t = 0.f;
      if (dstb) dstb[i * K + lane] = t;
      colsum = __fadd_rn(colsum, t);
    }
  }
  if (!dstb || !(p.flags & (TGP_DEGREE_NORM | TGP_EDGE_WEIGHT_NORM))) return;
  float d = 1.f;
  if (p.flags & TGP_DEGREE_NORM) {
    float mine = colsum;
    if (!(p.flags & TGP_SUM_AXIS_ROWS)) {  // degree over axis -1: the lane sums ITS ROW
      mine = 0.f;
      if (col_ok)
        for (int j = 0; j < K; ++j) {
          float t = combined(lane, j);
          if (rsl && j == lane) t = 0.f;
          mine = __fadd_rn(mine, t);
        }
    }
    d = sqrtf(fmaxf(mine, TGP_EPS));  // d[lane]
  }
  const bool by_cols = p.flags & TGP_SUM_AXIS_ROWS;
  float m = 0.f;
  for (int i = 0; i < K; ++i) {
    const float di = __shfl(d, i, WAVE);
    if (col_ok) {
      float t = dstb[i * K + lane];
      if (p.flags & TGP_DEGREE_NORM) {
        t = by_cols ? (t / d) / di : (t / di) / d;
        dstb[i * K + lane] = t;
      }
      m = fmaxf(m, fabsf(t));
    }
  }
  if (p.flags & TGP_EDGE_WEIGHT_NORM) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, WAVE));
    if (m == 0.f) m = 1.f;
    if (col_ok)
      for (int i = 0; i < K; ++i) dstb[i * K + lane] = dstb[i * K + lane] / m;
  }
}

// K <= 32: the whole matrix lives in one wave's registers.  Lanes 0..31 hold column `lane`, lanes 32..63
// hold row `lane - 32` (a second, transposed look at the same 4 KB), so both the axis -2 and the axis -1
// degree are plain per-lane sums in index order - the same fixed order as post_small_kernel.
__global__ __launch_bounds__(256) void post_tiny_kernel(PostArgs p, int B) {
  const int lane = lane_id(), K = p.K;
  const int b = blockIdx.x * 4 + wave_id();
  if (b >= B) return;
  const float* sb = p.src + static_cast<long>(b) * p.s_batch;
  float* rawb = p.raw ? p.raw + static_cast<long>(b) * K * K : nullptr;
  float* dstb = p.dst ? p.dst + static_cast<long>(b) * K * K : nullptr;
  const int idx = lane & 31;
  const bool hi = lane >= 32, ok = idx < K;
  const bool rsl = p.flags & TGP_REMOVE_SELF_LOOPS;
  float t[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    float v = 0.f;
    if (ok && j < K) {
      const long o = hi ? static_cast<long>(idx) * p.ld_src + j : static_cast<long>(j) * p.ld_src + idx;
      v = sb[o];
      for (int sp = 1; sp < p.splits; ++sp) v = __fadd_rn(v, sb[sp * p.s_split + o]);
    }
    t[j] = v;
  }
  if (rawb && !hi && ok) {
#pragma unroll
    for (int j = 0; j < 32; ++j)
      if (j < K) rawb[j * K + idx] = t[j];
  }
  if (!dstb) return;
  if (rsl) {
#pragma unroll
    for (int j = 0; j < 32; ++j)
      if (j == idx) t[j] = 0.f;
  }
  if (p.flags & TGP_DEGREE_NORM) {
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) sum = __fadd_rn(sum, t[j]);  // lanes < 32: column sums; >= 32: row sums
    const bool by_cols = p.flags & TGP_SUM_AXIS_ROWS;
    const float d = sqrtf(fmaxf(__shfl(sum, by_cols ? idx : 32 + idx, WAVE), TGP_EPS));  // d[idx] on every lane
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const float dj = __shfl(d, j, WAVE);
      t[j] = by_cols ? (t[j] / d) / dj : (t[j] / dj) / d;  // lanes < 32 hold element (row j, col idx)
    }
  }
  if (p.flags & TGP_EDGE_WEIGHT_NORM) {
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j)
      if (!hi && ok && j < K) m = fmaxf(m, fabsf(t[j]));
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) m = fmaxf(m, __shfl_xor(m, sft, WAVE));
    if (m == 0.f) m = 1.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) t[j] = t[j] / m;
  }
  if (!hi && ok) {
#pragma unroll
    for (int j = 0; j < 32; ++j)
      if (j < K) dstb[j * K + idx] = t[j];
  }
}

// 64 < K <= POST_LDS_MAX_K: one 1024-thread workgroup per graph keeps the K x K matrix in LDS, so the slab
// combine, the degree vector, the scaling and the max-norm are ONE launch (they are latency-, not
// bandwidth-shaped: [B,K,K] is tiny next to A).  Workgroups past the first B combine the X' slabs, which
// would otherwise be a launch of their own.  Every sum keeps the order of the multi-kernel path.
constexpr int POST_LDS_MAX_K = 176;  // K*K*4 + partials <= 160 KB

struct XCombineArgs {
  const float* src; int splits; long s_split, s_batch, total; float* dst; int blocks_per_graph;
};

__global__ __launch_bounds__(1024) void post_lds_kernel(PostArgs p, int B, XCombineArgs xc) {
  extern __shared__ __attribute__((aligned(16))) float m[];
  const int tid = threadIdx.x;
  if (static_cast<int>(blockIdx.x) >= B) {  // ---- X' slab combine ------------------------------------------
    const int xb = blockIdx.x - B;
    const int b = xb / xc.blocks_per_graph, part = xb - b * xc.blocks_per_graph;
    const float* sb = xc.src + static_cast<long>(b) * xc.s_batch;
    for (long e = static_cast<long>(part) * 1024 + tid; e < xc.total; e += static_cast<long>(xc.blocks_per_graph) * 1024) {
      float v = sb[e];
      int sp = 1;
      for (; sp + 3 < xc.splits; sp += 4) {  // four slab loads in flight, added in slab order
        const float u0 = sb[sp * xc.s_split + e], u1 = sb[(sp + 1) * xc.s_split + e];
        const float u2 = sb[(sp + 2) * xc.s_split + e], u3 = sb[(sp + 3) * xc.s_split + e];
        v = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(v, u0), u1), u2), u3);
      }
      for (; sp < xc.splits; ++sp) v = __fadd_rn(v, sb[sp * xc.s_split + e]);
      xc.dst[static_cast<long>(b) * xc.total + e] = v;
    }
    return;
  }
  const int b = blockIdx.x, K = p.K, lane = tid & 63, w = tid >> 6;
  const int kk = K * K;
  float* dv = m + kk;            // [K]
  float* s_part = dv + K;        // [16][64]
  float* s_max = s_part + 1024;  // [16]
  const float* sb = p.src + static_cast<long>(b) * p.s_batch;
  float* rawb = p.raw ? p.raw + static_cast<long>(b) * kk : nullptr;
  float* dstb = p.dst ? p.dst + static_cast<long>(b) * kk : nullptr;
  const bool rsl = p.flags & TGP_REMOVE_SELF_LOOPS;
  // pass 1: fixed-order slab sum -> raw output, diag-cleared copy in LDS  (K % 4 == 0, ld_src == K here)
  for (int e = tid * 4; e < kk; e += 4096) {
    float4 t = *reinterpret_cast<const float4*>(sb + e);
    auto add4 = [](float4& a, const float4& u) {
      a.x = __fadd_rn(a.x, u.x); a.y = __fadd_rn(a.y, u.y); a.z = __fadd_rn(a.z, u.z); a.w = __fadd_rn(a.w, u.w);
    };
    int sp = 1;
    for (; sp + 2 < p.splits; sp += 3) {  // three slab loads in flight, added in slab order
      const float4 u0 = *reinterpret_cast<const float4*>(sb + sp * p.s_split + e);
      const float4 u1 = *reinterpret_cast<const float4*>(sb + (sp + 1) * p.s_split + e);
      const float4 u2 = *reinterpret_cast<const float4*>(sb + (sp + 2) * p.s_split + e);
      add4(t, u0); add4(t, u1); add4(t, u2);
    }
    for (; sp < p.splits; ++sp) add4(t, *reinterpret_cast<const float4*>(sb + sp * p.s_split + e));
    if (rawb) *reinterpret_cast<float4*>(rawb + e) = t;
    if (rsl) {
      const int i = e / K, j = e - i * K;
      if (i == j) t.x = 0.f;
      if (i == j + 1) t.y = 0.f;
      if (i == j + 2) t.z = 0.f;
      if (i == j + 3) t.w = 0.f;
    }
    *reinterpret_cast<float4*>(m + e) = t;
  }
  if (!dstb) return;
  __syncthreads();
  if (p.flags & TGP_DEGREE_NORM) {
    if (p.flags & TGP_SUM_AXIS_ROWS) {  // column sums: wave w adds rows w, w+16, ...; partials added in order
      for (int base = 0; base < K; base += 64) {
        const int j = base + lane;
        float sacc = 0.f;
        if (j < K)
          for (int i = w; i < K; i += 16) sacc = __fadd_rn(sacc, m[i * K + j]);
        s_part[w * 64 + lane] = sacc;
        __syncthreads();
        if (w == 0 && j < K) {
          float t = 0.f;
#pragma unroll
          for (int q = 0; q < 16; ++q) t = __fadd_rn(t, s_part[q * 64 + lane]);
          dv[j] = sqrtf(fmaxf(t, TGP_EPS));
        }
        __syncthreads();
      }
    } else {                            // row sums: lanes stride over the columns, fixed shuffle tree
      for (int i = w; i < K; i += 16) {
        float sacc = 0.f;
        for (int j = lane; j < K; j += 64) sacc = __fadd_rn(sacc, m[i * K + j]);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) sacc = __fadd_rn(sacc, __shfl_down(sacc, d, WAVE));
        if (lane == 0) dv[i] = sqrtf(fmaxf(sacc, TGP_EPS));
      }
      __syncthreads();
    }
  }
  const bool by_cols = p.flags & TGP_SUM_AXIS_ROWS, ewn = p.flags & TGP_EDGE_WEIGHT_NORM;
  float mx = 0.f;
  for (int e = tid * 4; e < kk; e += 4096) {
    float4 t = *reinterpret_cast<const float4*>(m + e);
    if (p.flags & TGP_DEGREE_NORM) {
      const int i = e / K, j = e - i * K;
      const float di = dv[i];
      const float4 dj = *reinterpret_cast<const float4*>(dv + j);
      if (by_cols) {  // (adj / d[1,K]) / d[K,1]
        t.x = (t.x / dj.x) / di; t.y = (t.y / dj.y) / di; t.z = (t.z / dj.z) / di; t.w = (t.w / dj.w) / di;
      } else {
        t.x = (t.x / di) / dj.x; t.y = (t.y / di) / dj.y; t.z = (t.z / di) / dj.z; t.w = (t.w / di) / dj.w;
      }
    }
    if (ewn) {
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(t.x), fabsf(t.y))), fmaxf(fabsf(t.z), fabsf(t.w)));
      *reinterpret_cast<float4*>(m + e) = t;
    } else {
      *reinterpret_cast<float4*>(dstb + e) = t;
    }
  }
  if (!ewn) return;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, WAVE));
  if (lane == 0) s_max[w] = mx;
  __syncthreads();
  float gm = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) gm = fmaxf(gm, s_max[q]);
  if (gm == 0.f) gm = 1.f;
  for (int e = tid * 4; e < kk; e += 4096) {
    float4 t = *reinterpret_cast<const float4*>(m + e);
    t.x = t.x / gm; t.y = t.y / gm; t.z = t.z / gm; t.w = t.w / gm;
    *reinterpret_cast<float4*>(dstb + e) = t;
  }
}

static size_t post_ws_floats(int64_t B, int64_t K) { return static_cast<size_t>(B * K + B * POST_BLOCKS); }

// Returns true when the X' slab combine described by xc (if any) was folded into the launch.
static bool launch_post(PostArgs p, int64_t B, float* ws, hipStream_t stream, const XCombineArgs* xc = nullptr) {
  const int K = p.K;
  if (K <= 64) {  // one wave per graph: one launch instead of three or four
    const dim3 grid(static_cast<unsigned>((B + 3) / 4));
    if (K <= 32) hipLaunchKernelGGL(post_tiny_kernel, grid, dim3(256), 0, stream, p, static_cast<int>(B));
    else hipLaunchKernelGGL(post_small_kernel, grid, dim3(256), 0, stream, p, static_cast<int>(B));
    return false;
  }
  static const int no_lds = getenv("TGP_NO_POST_LDS") ? 1 : 0;
  if (!no_lds && K <= POST_LDS_MAX_K && K % 4 == 0 && p.ld_src == K && p.s_split % 4 == 0 && p.s_batch % 4 == 0 &&
      reinterpret_cast<uintptr_t>(p.src) % 16 == 0 && (!p.raw || reinterpret_cast<uintptr_t>(p.raw) % 16 == 0) &&
      (!p.dst || reinterpret_cast<uintptr_t>(p.dst) % 16 == 0)) {
    XCombineArgs x{};
    if (xc) {
      x = *xc;
      x.blocks_per_graph = static_cast<int>((x.total + 4095) / 4096);
      if (x.blocks_per_graph < 1) x.blocks_per_graph = 1;
      if (x.blocks_per_graph > 8) x.blocks_per_graph = 8;
    }
    const size_t lds = (static_cast<size_t>(K) * K + K + 1024 + 16) * sizeof(float);
    const unsigned grid = static_cast<unsigned>(B + (xc ? B * x.blocks_per_graph : 0));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(post_lds_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(post_lds_kernel, dim3(grid), dim3(1024), lds, stream, p, static_cast<int>(B), x);
    return xc != nullptr;
  }
  p.dvec = ws;
  p.maxpart = ws + static_cast<size_t>(B) * K;
  const bool vec = (K % 4 == 0) && (p.ld_src % 4 == 0) && (p.s_split % 4 == 0) && (p.s_batch % 4 == 0) &&
                   (reinterpret_cast<uintptr_t>(p.src) % 16 == 0) &&
                   (!p.raw || reinterpret_cast<uintptr_t>(p.raw) % 16 == 0) &&
                   (!p.dst || reinterpret_cast<uintptr_t>(p.dst) % 16 == 0);
  const long groups = static_cast<long>(K) * K / (vec ? 4 : 1);
  int gx = static_cast<int>((groups + 255) / 256);
  if (gx > 64) gx = 64;
  const dim3 gridc(gx, static_cast<unsigned>(B));
  if (vec) hipLaunchKernelGGL(post_combine_kernel<4>, gridc, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(post_combine_kernel<1>, gridc, dim3(256), 0, stream, p);
  if (!p.dst) return false;
  const dim3 gride(POST_BLOCKS, static_cast<unsigned>(B));
  if (p.flags & TGP_DEGREE_NORM)
    hipLaunchKernelGGL(post_degree_kernel, dim3((K + 63) / 64, static_cast<unsigned>(B)), dim3(1024), 0, stream, p);
  if (p.flags & (TGP_DEGREE_NORM | TGP_EDGE_WEIGHT_NORM))
    hipLaunchKernelGGL(post_scale_kernel, gride, dim3(256), 0, stream, p);
  if (p.flags & TGP_EDGE_WEIGHT_NORM) hipLaunchKernelGGL(post_maxnorm_kernel, gride, dim3(256), 0, stream, p);
  return false;
}

// x_pool slabs -> x_pool (fixed-order combine); src [B][splits][K][F]
__global__ __launch_bounds__(256) void combine_slabs_kernel(const float* __restrict__ src, int splits,
                                                            long s_split, long s_batch, long total,
                                                            float* __restrict__ dst) {
  const int b = blockIdx.y;
  const float* sb = src + static_cast<long>(b) * s_batch;
  for (long e = static_cast<long>(blockIdx.x) * 256 + threadIdx.x; e < total;
       e += static_cast<long>(gridDim.x) * 256) {
    float v = sb[e];
    for (int s = 1; s < splits; ++s) v = __fadd_rn(v, sb[s * s_split + e]);
    dst[static_cast<long>(b) * total + e] = v;
  }
}

// ------------------------------------------------------------------------------------------
// Small graphs (N <= 64, K <= 32, F <= 32; e.g. the PROTEINS-shaped batch of BASELINE configs[2]):
// one WAVE owns one graph.  S and X go straight from HBM into the MFMA operand registers (row-coalesced:
// lane = column, one node row per half-wave); only A, whose operand layout is the transpose of its
// memory layout, is staged through LDS (zero padded 64 x 65 per wave, so two workgroups fit a CU).
// Then X' = S^T X (32 MFMAs), U = A S (64) and A' = S^T U (32) run back to back.  U never leaves the
// accumulators: register r of the 32x32 C/D layout holds rows (rho(r), rho(r)+4) for the two half-waves,
// which is exactly a k-pair of the next MFMA's B operand, so every product walks the node dimension in
// that order (node(q) below) and all three share one register copy of S.  The post-processing
// (utils/ops.py:282-335) happens in registers + wave shuffles.  Each graph crosses HBM once: HBM-bound.
// ------------------------------------------------------------------------------------------
constexpr int SG_N = 64, SG_K = 32, SG_LDA = 65;
constexpr int SG_WAVE_FLOATS = SG_N * SG_LDA;  // A only

struct SmallArgs {
  const float* S; const float* A; const float* X;
  int B, N, K, F, flags;
  float* x_pool; float* adj_raw; float* adj_pool;
};

__device__ __forceinline__ int rho(int r) { return (r & 3) + 8 * (r >> 2); }
// uniform base + 32-bit per-lane byte offset: lets the load use the SGPR-base addressing form
__device__ __forceinline__ const float* byte_off(const float* base, int bytes) {
  return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + static_cast<unsigned>(bytes));
}

#ifdef TGP_GEMM_STAMPS
// per-WAVE stamps of the small-graph kernel (slot s of graph b at stamps[b*16 + s])
#define TGP_WSTAMP(slot)                                                                                  \
  do {                                                                                                    \
    if (g_gemm_stamps && lane_id() == 0)                                                                  \
      g_gemm_stamps[static_cast<long>(blockIdx.x * 4 + wave_id()) * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define TGP_WSTAMP(slot) do {} while (0)
#endif

__global__ __launch_bounds__(256, 2) void dense_pool_small_kernel(SmallArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = lane_id();
  const int w = __builtin_amdgcn_readfirstlane(wave_id());  // wave-uniform => graph bases stay in SGPRs
  const int lm = lane & 31, lk = lane >> 5;
  float* As = smem + w * SG_WAVE_FLOATS;
  const int N = p.N, K = p.K, F = p.F;
  const bool at = p.flags & TGP_ADJ_TRANSPOSED;
  const int b = blockIdx.x * 4 + w;  // one graph per wave, no loop (keeps the 64 + 64 load offsets transient)
  if (b >= p.B) return;
  TGP_WSTAMP(0);
  {
    // ---- request everything up front: A (float4 rows), then S and X in operand order ------------
    // out-of-range elements read element 0 of the graph (always valid) and are replaced by 0 afterwards, so
    // the loads stay unconditional and are issued back to back
    float4 v[16];
    if (p.A) {  // 16 lanes per row (64 floats), 4 rows per wave-instruction, 16 instructions
      const float* Ab = p.A + static_cast<long>(b) * N * N;
      const int q = lane & 15;
      const bool qfull = 4 * q + 3 < N;   // whole vector inside the row (rows need dword alignment only)
      const int qrem = N - 4 * q;         // 1..3 on the lane that holds a row's ragged tail (N % 4 != 0)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int i = (lane >> 4) + 4 * t;
        const bool ok = qfull && i < N;
        const float4 r = *reinterpret_cast<const float4*>(byte_off(Ab, ok ? (i * N + 4 * q) * 4 : 0));
        v[t] = ok ? r : make_float4(0.f, 0.f, 0.f, 0.f);
        if (!qfull && qrem > 0 && i < N) {  // never read past the end of the row (= of the tensor for the last one)
          const float* tail = byte_off(Ab, (i * N + 4 * q) * 4);
          v[t].x = tail[0];
          if (qrem > 1) v[t].y = tail[1];
          if (qrem > 2) v[t].z = tail[2];
        }
      }
    }
    // step q of every product contracts node rows node(q) = 32*(q>>4) + rho(q&15) + 4*lk
    float sr[32], xr[32];
    {
      const float* Sb = p.S + static_cast<long>(b) * N * K;
      const bool cok = lm < K;
#pragma unroll
      for (int q = 0; q < 32; ++q) {
        const int node = 32 * (q >> 4) + rho(q & 15) + 4 * lk;
        const bool ok = cok && node < N;
        const float r = *byte_off(Sb, ok ? (node * K + lm) * 4 : 0);
        sr[q] = ok ? r : 0.f;
      }
    }
    if (p.X) {
      const float* Xb = p.X + static_cast<long>(b) * N * F;
      const bool cok = lm < F;
#pragma unroll
      for (int q = 0; q < 32; ++q) {
        const int node = 32 * (q >> 4) + rho(q & 15) + 4 * lk;
        const bool ok = cok && node < N;
        const float r = *byte_off(Xb, ok ? (node * F + lm) * 4 : 0);
        xr[q] = ok ? r : 0.f;
      }
    }
    if (p.A) {
      const int q = lane & 15;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int i = (lane >> 4) + 4 * t;
        if (!at) {
          float* d = As + i * SG_LDA + 4 * q;
          d[0] = v[t].x; d[1] = v[t].y; d[2] = v[t].z; d[3] = v[t].w;
        } else {  // memory holds A^T: element (row i, cols 4q..4q+3) of memory is A[4q+j][i]
          As[(4 * q + 0) * SG_LDA + i] = v[t].x; As[(4 * q + 1) * SG_LDA + i] = v[t].y;
          As[(4 * q + 2) * SG_LDA + i] = v[t].z; As[(4 * q + 3) * SG_LDA + i] = v[t].w;
        }
      }
    }
    // the tile belongs to this wave alone and LDS serves a wave's requests in order: no workgroup barrier
    __builtin_amdgcn_wave_barrier();
    TGP_WSTAMP(1);

    // ---- X' = S^T X ---------------------------------------------------------------------
    if (p.X && p.x_pool) {
      f32x16 ax;
#pragma unroll
      for (int r = 0; r < 16; ++r) ax[r] = 0.f;
#pragma unroll
      for (int q = 0; q < 32; ++q) ax = __builtin_amdgcn_mfma_f32_32x32x2f32(sr[q], xr[q], ax, 0, 0, 0);
      if (lm < F) {
        float* o = p.x_pool + static_cast<long>(b) * K * F;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = rho(r) + 4 * lk;
          if (c < K) o[c * F + lm] = ax[r];
        }
      }
    }

    TGP_WSTAMP(2);
    // ---- U = A S (kept in accumulators), A' = S^T U -----------------------------------------
    if (p.A && (p.adj_raw || p.adj_pool)) {
      f32x16 u[2], aa;
#pragma unroll
      for (int r = 0; r < 16; ++r) { u[0][r] = 0.f; u[1][r] = 0.f; aa[r] = 0.f; }
#pragma unroll
      for (int q = 0; q < 32; ++q) {
        const int node = 32 * (q >> 4) + rho(q & 15) + 4 * lk;
        u[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(As[lm * SG_LDA + node], sr[q], u[0], 0, 0, 0);
        u[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(As[(32 + lm) * SG_LDA + node], sr[q], u[1], 0, 0, 0);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          aa = __builtin_amdgcn_mfma_f32_32x32x2f32(sr[mt * 16 + r], u[mt][r], aa, 0, 0, 0);

      TGP_WSTAMP(3);
      // aa[r] = A'[row = rho(r) + 4*lk][col = lm]
      if (p.adj_raw && lm < K) {
        float* o = p.adj_raw + static_cast<long>(b) * K * K;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = rho(r) + 4 * lk;
          if (i < K) o[i * K + lm] = aa[r];
        }
      }
      if (p.adj_pool) {
        if (p.flags & TGP_REMOVE_SELF_LOOPS) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (rho(r) + 4 * lk == lm) aa[r] = 0.f;
        }
        if (p.flags & TGP_DEGREE_NORM) {
          float dcol;  // degree of index `lm`, identical on both half-waves
          if (p.flags & TGP_SUM_AXIS_ROWS) {  // sum over rows (axis -2): per-lane column sum
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += aa[r];
            s += __shfl_xor(s, 32, WAVE);
            dcol = s;
          } else {                            // sum over columns (axis -1): reduce each row over lanes
            float mine = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float s = aa[r];
#pragma unroll
              for (int d = 16; d > 0; d >>= 1) s += __shfl_xor(s, d, WAVE);
              // row (rho(r) + 4*lk) total now on every lane of this half-wave; hand it to lane = row
              const int row = rho(r) + 4 * lk;
              if (lm == row) mine = s;
            }
            // lanes of the other half-wave own the other 16 rows: merge
            const float other = __shfl_xor(mine, 32, WAVE);
            bool own = false;
#pragma unroll
            for (int r = 0; r < 16; ++r) own |= (rho(r) + 4 * lk == lm);
            dcol = own ? mine : other;
          }
          const float d = sqrtf(fmaxf(dcol, TGP_EPS));  // d[lm]
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = rho(r) + 4 * lk;
            const float drow = __shfl(d, row, WAVE);      // d[row]
            const float first = (p.flags & TGP_SUM_AXIS_ROWS) ? d : drow;
            const float second = (p.flags & TGP_SUM_AXIS_ROWS) ? drow : d;
            aa[r] = (aa[r] / first) / second;
          }
        }
        if (p.flags & TGP_EDGE_WEIGHT_NORM) {
          float m = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (lm < K && rho(r) + 4 * lk < K) m = fmaxf(m, fabsf(aa[r]));
#pragma unroll
          for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, WAVE));
          if (m == 0.f) m = 1.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) aa[r] = aa[r] / m;
        }
        TGP_WSTAMP(4);
        if (lm < K) {
          float* o = p.adj_pool + static_cast<long>(b) * K * K;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int i = rho(r) + 4 * lk;
            if (i < K) o[i * K + lm] = aa[r];
          }
        }
      }
    }
  }
  TGP_WSTAMP(5);
}

// ------------------------------------------------------------------------------------------
// Medium graphs (TU-dataset-sized batches: N up to a few hundred, K <= 64): one WORKGROUP (4 waves) owns one
// graph, every byte of A / X crosses HBM once and nothing intermediate leaves the CU.
//   * S [N,K] is copied to LDS once (zero padded to 32-row / 32-column multiples); every MFMA reads one of its
//     operands from there (lane = cluster: consecutive words, conflict-free).
//   * A and X are read straight from HBM into the MFMA B-operand registers: lane = column, so a half-wave
//     reads 128 contiguous bytes of one row per k-step -- the operand layout IS the memory layout, no staging
//     (that is why the product is associated as (S^T A) S here; (A S) would need A transposed through LDS).
//     Buffer-descriptor loads: rows / columns outside the graph come back as zeros from the range check.
//   * work items = 32-column strips of A (then of X), dealt round-robin to the waves.  An A strip gives
//     P = (S^T A)^T restricted to the strip, [32 nodes x K], with the strip as the MFMA A operand; register r
//     of the C/D layout holds strip rows (rho(r), rho(r)+4) on the two half-waves = a k-pair of a B operand,
//     so P goes straight from the accumulators into A'[c1][c2] += sum_n P[n][c1] S[n][c2] (no LDS round trip).
//     The four waves' partial A' are added in wave order (deterministic), then the workgroup post-processes
//     the K x K result in LDS (utils/ops.py:282-335) and stores it.
// If memory holds A^T (TGP_ADJ_TRANSPOSED) the same program yields (A')^T, which is transposed on the way
// into the post-processing buffer.
// ------------------------------------------------------------------------------------------
struct MediumArgs {
  const float* S; const float* A; const float* X;
  int B, N, K, F, flags;
  float* x_pool; float* adj_raw; float* adj_pool;
  int npad;  // N rounded up to 32
};

template <int MT>
static size_t medium_lds_bytes(int64_t npad) {
  constexpr int KP = 32 * MT;
  return (static_cast<size_t>(npad) * KP + KP * (KP + 1) + KP) * sizeof(float);
}

template <int MT>
__global__ __launch_bounds__(256, MT == 1 ? 4 : 3) void dense_pool_medium_kernel(MediumArgs p) {
  constexpr int KP = 32 * MT;          // padded K
  constexpr int UNROLL = 8;            // k-pairs whose operands are requested together (two such sets in flight;
                                       // 16 measured no faster for K <= 32 and spills for K <= 64)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, lm = lane & 31, lk = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = p.N, K = p.K, F = p.F, NP = p.npad;
  const int b = blockIdx.x;
  float* Ss = smem;                       // [NP][KP]
  float* Rs = Ss + NP * KP;               // [KP][KP+1]
  float* ds = Rs + KP * (KP + 1);         // [KP] degrees
  const bool want_a = p.A && (p.adj_raw || p.adj_pool);
  const bool want_x = p.X && p.x_pool;

  TGP_WSTAMP(0);
  // ---- S -> LDS (zero padded) ------------------------------------------------------------
  {
    // eight independent loads in flight per thread: a one-element-per-iteration loop exposes the full load latency
    // NP * KP / 256 times (measured: 22 us of a 92 us workgroup at N = 200, K = 50)
    const float* Sb = p.S + static_cast<long>(b) * N * K;
    constexpr int UB = 8;
    for (int base = 0; base < NP * KP; base += 256 * UB) {
      float v[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int e = base + u * 256 + tid;
        const int r = e / KP, c = e - r * KP;
        v[u] = (r < N && c < K) ? Sb[r * K + c] : 0.f;  // r < N also covers e beyond the tile (NP >= N)
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int e = base + u * 256 + tid;
        if (e < NP * KP) Ss[e] = v[u];
      }
    }
  }
  __syncthreads();

  TGP_WSTAMP(1);
  f32x16 racc[MT][MT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) racc[i][j][r] = 0.f;

  const int nt_a = want_a ? (N + 31) / 32 : 0;
  const int nt_x = want_x ? (F + 31) / 32 : 0;
  constexpr int OOB = static_cast<int>(0x80000000u);
  // Strips are dealt round-robin, starting at a wave that rotates with the graph index: wave w always runs on SIMD
  // w, so a fixed start would pile every graph's extra strip onto the same SIMD of the CU.
  for (int job = (w + 4 - (b & 3)) & 3; job < nt_a + nt_x; job += 4) {
    const bool is_a = job < nt_a;
    const int n0 = (is_a ? job : job - nt_a) * 32;
    const int ld = is_a ? N : F;
    const float* src = is_a ? p.A + static_cast<long>(b) * N * N : p.X + static_cast<long>(b) * N * F;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, N * ld * 4, 0x00020000);
    const int voff = (n0 + lm < ld) ? (lk * ld + n0 + lm) * 4 : OOB;
    // strip element (node row k + lk, column n0 + lm) of A or X; two register sets: the next batch of k-pairs
    // is requested before the MFMAs of the current one
    float gv[2][UNROLL];
    auto request = [&](int set, int k0) {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
        gv[set][u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (k0 + 2 * u) * ld * 4, 0));
    };
    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // X strip:  X'[c][f]  = sum_k S[k][c] X[k][f]    (S = A operand from LDS, the strip = B operand)
    // A strip:  P [n][c]  = sum_k A[k][n] S[k][c]    (the strip = A operand, S = B operand from LDS) = T^T
    auto consume = [&](auto is_a_c, int set, int k0) {
      constexpr bool IS_A = decltype(is_a_c)::value;
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int k = k0 + 2 * u + lk;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const float sv = Ss[k * KP + i * 32 + lm];
          if constexpr (IS_A) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv[set][u], sv, acc[i], 0, 0, 0);
          else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv, gv[set][u], acc[i], 0, 0, 0);
        }
      }
    };
    auto k_loop = [&](auto is_a_c) {
      request(0, 0);
      for (int k0 = 0; k0 < NP; k0 += 4 * UNROLL) {  // NP is a multiple of 32 = 2 * UNROLL node rows
        if (k0 + 2 * UNROLL < NP) request(1, k0 + 2 * UNROLL);
        consume(is_a_c, 0, k0);
        if (k0 + 2 * UNROLL >= NP) break;
        if (k0 + 4 * UNROLL < NP) request(0, k0 + 4 * UNROLL);
        consume(is_a_c, 1, k0 + 2 * UNROLL);
      }
    };
    if (is_a) k_loop(std::true_type{});   // wave-uniform branch: one operand order per loop body
    else k_loop(std::false_type{});
    if (!is_a) {  // X' strip: rows = clusters, cols = features n0 .. n0+31
      if (n0 + lm < F) {
        float* o = p.x_pool + static_cast<long>(b) * K * F;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c = i * 32 + rho(r) + 4 * lk;
            if (c < K) o[c * F + n0 + lm] = acc[i][r];
          }
      }
      continue;
    }
    // R[c1][c2] = sum_n P[n][c1] S[n][c2]: register r of the C/D layout holds strip rows (rho(r), rho(r)+4) on the
    // two half-waves, which is exactly a k-pair of a B operand, so P never leaves the accumulators:
    //   D[row = c2][col = c1] += S[n0 + rho(r) + 4 lk][c2]  x  P_r[c1]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float sv[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) sv[i] = Ss[(n0 + rho(r) + 4 * lk) * KP + i * 32 + lm];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
          racc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[i], acc[j][r], racc[i][j], 0, 0, 0);
    }
  }
  TGP_WSTAMP(2);
  if (!want_a) return;

  // ---- A' = sum of the four partial products, in wave order ---------------------------------
  const bool at = p.flags & TGP_ADJ_TRANSPOSED;
  for (int turn = 0; turn < 4; ++turn) {
    if (w == turn) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c2 = i * 32 + rho(r) + 4 * lk, c1 = j * 32 + lm;  // the accumulators hold R[c1][c2]
            float* d = at ? &Rs[c2 * (KP + 1) + c1] : &Rs[c1 * (KP + 1) + c2];
            *d = turn == 0 ? racc[i][j][r] : __fadd_rn(*d, racc[i][j][r]);
          }
    }
    __syncthreads();
  }

  TGP_WSTAMP(3);
  // ---- post-processing on the K x K result (utils/ops.py:282-335) ---------------------------
  // element loops run over the padded [K][KP] index space: row / column come from shifts, not divisions
  const long obase = static_cast<long>(b) * K * K;
  if (p.adj_raw) {
    for (int e = tid; e < K * KP; e += 256) {
      const int i = e / KP, j = e % KP;
      if (j < K) p.adj_raw[obase + i * K + j] = Rs[i * (KP + 1) + j];
    }
    __syncthreads();  // the diagonal is cleared next
  }
  if (!p.adj_pool) return;
  if (p.flags & TGP_REMOVE_SELF_LOOPS) {
    if (tid < K) Rs[tid * (KP + 1) + tid] = 0.f;
    __syncthreads();
  }
  if (p.flags & TGP_DEGREE_NORM) {
    const bool rows = p.flags & TGP_SUM_AXIS_ROWS;
    if (tid < K) {
      float t = 0.f;
      for (int q = 0; q < K; ++q) t = __fadd_rn(t, rows ? Rs[q * (KP + 1) + tid] : Rs[tid * (KP + 1) + q]);
      ds[tid] = sqrtf(fmaxf(t, TGP_EPS));
    }
    __syncthreads();
    for (int e = tid; e < K * KP; e += 256) {
      const int i = e / KP, j = e % KP;
      if (j < K) {
        const float first = rows ? ds[j] : ds[i], second = rows ? ds[i] : ds[j];
        Rs[i * (KP + 1) + j] = (Rs[i * (KP + 1) + j] / first) / second;
      }
    }
    __syncthreads();
  }
  float scale = 1.f;
  if (p.flags & TGP_EDGE_WEIGHT_NORM) {
    float m = 0.f;
    for (int e = tid; e < K * KP; e += 256) {
      const int i = e / KP, j = e % KP;
      if (j < K) m = fmaxf(m, fabsf(Rs[i * (KP + 1) + j]));
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, WAVE));
    __syncthreads();
    if (lane == 0) ds[w] = m;
    __syncthreads();
    scale = fmaxf(fmaxf(ds[0], ds[1]), fmaxf(ds[2], ds[3]));
    if (scale == 0.f) scale = 1.f;
  }
  for (int e = tid; e < K * KP; e += 256) {
    const int i = e / KP, j = e % KP;
    if (j < K) {
      const float v = Rs[i * (KP + 1) + j];
      p.adj_pool[obase + i * K + j] = (p.flags & TGP_EDGE_WEIGHT_NORM) ? v / scale : v;
    }
  }
  TGP_WSTAMP(4);
}

static const int kStage2Tile = getenv("TGP_STAGE2_TILE") ? atoi(getenv("TGP_STAGE2_TILE")) : 0;
static const int kStage2TileM = getenv("TGP_STAGE2_TILE_M") ? atoi(getenv("TGP_STAGE2_TILE_M")) : 0;
struct DensePlan {
  int splits, tile, tile_m;
  int k_per_split;
  size_t u_floats, aslab_floats, xslab_floats, post_floats;
};

static DensePlan dense_plan(int64_t B, int64_t N, int64_t K, int64_t F) {
  DensePlan p;
  // second product: output is only K x (K+F) per graph -> split N so the chip is filled.  Small K: 64 x 64
  // tiles (256-thread workgroups) need fewer splits for the same number of workgroups, i.e. longer k-loops
  // per workgroup and less slab traffic.
  p.tile = (kStage2Tile == 64 || kStage2Tile == 128) ? kStage2Tile : 64;
  p.tile_m = (kStage2TileM == 64 || kStage2TileM == 128) ? kStage2TileM : p.tile;
  const int64_t T = p.tile, TM = p.tile_m;
  const int64_t tiles = ((K + TM - 1) / TM) * (((K + T - 1) / T) + ((F + T - 1) / T));
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
  return p;
}

}  // namespace tgp

using namespace tgp;

extern "C" size_t tgp_dense_pool_workspace_bytes(int64_t B, int64_t N, int64_t K, int64_t F) {
  if (B <= 0 || N <= 0 || K <= 0) return 256;
  const DensePlan p = dense_plan(B, N, K, F > 0 ? F : 0);
  return align_up(p.u_floats * 4) + align_up(p.aslab_floats * 4) + align_up(p.xslab_floats * 4) +
         align_up(p.post_floats * 4) + 256;
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
  TGP_REQUIRE(N < (1ll << 31) && K <= 16000 && F < (1ll << 31) && B < 65536, TGP_ERR_RANGE,
              "tgp_dense_pool_f32: dimension too large");
  if (N == 0) {
    if (want_x) (void)hipMemsetAsync(x_pool, 0, sizeof(float) * B * K * F, stream);
    if (adj_raw) (void)hipMemsetAsync(adj_raw, 0, sizeof(float) * B * K * K, stream);
    if (adj_pool) (void)hipMemsetAsync(adj_pool, 0, sizeof(float) * B * K * K, stream);
    return check_launch("tgp_dense_pool_f32");
  }
  static const int no_small = getenv("TGP_NO_SMALL_GRAPH_KERNEL") ? 1 : 0;
  const bool small_ok = N <= SG_N && K <= SG_K && F <= SG_K && B >= 64;  // any N, K, F: padded batches are ragged
  if (!no_small && small_ok) {
    SmallArgs q{S, want_a ? A : nullptr, want_x ? X : nullptr, static_cast<int>(B), static_cast<int>(N),
                static_cast<int>(K), static_cast<int>(F), flags, want_x ? x_pool : nullptr,
                want_a ? adj_raw : nullptr, want_a ? adj_pool : nullptr};
    const int grid = static_cast<int>((B + 3) / 4);
    hipLaunchKernelGGL(dense_pool_small_kernel, dim3(grid), dim3(256), 4 * SG_WAVE_FLOATS * sizeof(float), stream, q);
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
                   static_cast<int>(K), static_cast<int>(F), flags, want_x ? x_pool : nullptr,
                   want_a ? adj_raw : nullptr, want_a ? adj_pool : nullptr, static_cast<int>(npad)};
      if (K <= 32)
        hipLaunchKernelGGL(dense_pool_medium_kernel<1>, dim3(static_cast<unsigned>(B)), dim3(256), lds, stream, q);
      else
        hipLaunchKernelGGL(dense_pool_medium_kernel<2>, dim3(static_cast<unsigned>(B)), dim3(256), lds, stream, q);
      return check_launch("tgp_dense_pool_f32(medium)");
    }
  }
  TGP_REQUIRE(ws && ws_bytes >= tgp_dense_pool_workspace_bytes(B, N, K, F), TGP_ERR_WORKSPACE,
              "tgp_dense_pool_f32: workspace too small");
  const DensePlan p = dense_plan(B, N, K, F);
  Carver cv(ws);
  float* U = cv.take<float>(p.u_floats);
  float* aslab = cv.take<float>(p.aslab_floats);
  float* xslab = cv.take<float>(p.xslab_floats);
  float* postws = cv.take<float>(p.post_floats);

  if (want_a) {
    // U[b] = A[b] S[b]     (M = N, Kd = N, Nc = K)
    GemmArgs g{};
    g.A = A; g.lda = N; g.sA = N * N;
    g.M = static_cast<int>(N); g.Kd = static_cast<int>(N);
    g.rhs[0] = GemmRhs{S, U, static_cast<int>(K), K, K, N * K, N * K, 0};
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
    const GemmRhs ra{U, aslab, static_cast<int>(K), K, K, N * K, static_cast<long>(p.splits) * K * K, K * K};
    const GemmRhs rx{X, xslab, static_cast<int>(F), F, F, N * F, static_cast<long>(p.splits) * K * F, K * F};
    if (want_a && want_x) {
      h.rhs[0] = ra; h.rhs[1] = rx;

    } else {
      h.rhs[0] = want_a ? ra : rx;

    }
    launch_gemm<true>(h, static_cast<int>(B), stream);
  }
  bool x_done = !want_x;
  if (want_a) {
    PostArgs q{};
    q.src = aslab; q.splits = p.splits; q.s_split = K * K; q.s_batch = static_cast<long>(p.splits) * K * K;
    q.ld_src = K; q.K = static_cast<int>(K); q.flags = flags; q.raw = adj_raw; q.dst = adj_pool;
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

extern "C" size_t tgp_postprocess_dense_workspace_bytes(int64_t B, int64_t K) {
  if (B <= 0 || K <= 0) return 256;
  return align_up(post_ws_floats(B, K) * 4) + 256;
}

extern "C" int tgp_postprocess_dense_f32(const float* src, float* dst, int64_t B, int64_t K, int flags, void* ws,
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
  q.K = static_cast<int>(K); q.flags = flags; q.raw = nullptr; q.dst = dst;
  launch_post(q, B, static_cast<float*>(ws), stream);
  return check_launch("tgp_postprocess_dense_f32");
}

namespace tgp {
// Batched products of SMALL matrices (the backward of the dense poolers on TU-dataset-sized graphs: [N x F][F x K],
// [N x N][N x K], ... with N up to a few hundred and at most 64 output columns): the LDS-tiled kernel above spends
// such a launch on 64 / 128-wide tiles that are mostly padding.  Here one WAVE owns a 32-row strip of one batch
// element and keeps 32 x Nc of C in its accumulators; both operands go straight from memory into the MFMA operand
// registers (B: lane = column, coalesced; A: lane = row, each lane walks its own row, so every 64-byte line is
// fetched once and serves 16 k-steps from L1), eight k-pairs requested at a time.  No LDS, no barriers.
struct SmallBmmArgs {
  const float* A; const float* Bm; float* C;
  int M, Nc, Kd, trans_a;
  long lda, ldb, ldc, sA, sB, sC;
  int strips;  // 32-row strips per batch element
};

template <int NT>
__global__ __launch_bounds__(256) void small_bmm_kernel(SmallBmmArgs p, long total_strips) {
  const long strip = static_cast<long>(blockIdx.x) * 4 + wave_id();
  if (strip >= total_strips) return;
  const int lane = lane_id(), lm = lane & 31, lk = lane >> 5;
  const long b = strip / p.strips;
  const int m0 = static_cast<int>(strip - b * p.strips) * 32;
  const float* A = p.A + b * p.sA;
  const float* Bm = p.Bm + b * p.sB;
  float* C = p.C + b * p.sC;
  const bool row_ok = m0 + lm < p.M;
  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  constexpr int U = 8;
  for (int k0 = 0; k0 < p.Kd; k0 += 2 * U) {
    float av[U], bv[U][NT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + 2 * u + lk;
      const bool k_ok = k < p.Kd;
      const long a_off = p.trans_a ? static_cast<long>(k) * p.lda + m0 + lm : static_cast<long>(m0 + lm) * p.lda + k;
      av[u] = (k_ok && row_ok) ? A[a_off] : 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = j * 32 + lm;
        bv[u][j] = (k_ok && n < p.Nc) ? Bm[static_cast<long>(k) * p.ldb + n] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u][j], acc[j], 0, 0, 0);
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = j * 32 + lm;
    if (n >= p.Nc) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + rho(r) + 4 * lk;
      if (m < p.M) C[static_cast<long>(m) * p.ldc + n] = acc[j][r];
    }
  }
}
}  // namespace tgp

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
  g.A = A; g.lda = lda; g.sA = sA;
  g.M = static_cast<int>(M); g.Kd = static_cast<int>(Kd);
  g.rhs[0] = GemmRhs{Bm, C, static_cast<int>(Nc), ldb, ldc, sB, sC, 0};
  g.splits = 1; g.k_per_split = static_cast<int>((Kd + BK - 1) / BK * BK);
  TGP_REQUIRE(batch * ((M + 63) / 64) * ((Nc + 63) / 64) < (1ll << 31), TGP_ERR_RANGE, "tgp_bmm_f32: grid too large");
  // many small matrices: one wave per 32-row strip, operands straight from memory (see small_bmm_kernel)
  static const int no_small_bmm = getenv("TGP_NO_SMALL_BMM") ? 1 : 0;
  const int64_t strips = (M + 31) / 32;
  if (!no_small_bmm && Nc <= 64 && M <= 512 && Kd <= 512 && batch * strips >= 512) {
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
  return g;
}

extern "C" size_t tgp_link_loss_workspace_bytes(int64_t B, int64_t N, int64_t K) {
  if (B <= 0 || N <= 0) return 256;
  tgp::GemmArgs g = link_loss_args(nullptr, nullptr, N, K);
  const int tiles = tgp::launch_gemm_residual(g, static_cast<int>(B), nullptr, true);
  return tgp::align_up(static_cast<size_t>(B) * tiles * sizeof(float)) + 256;
}

extern "C" int tgp_link_loss_f32(const float* S, const float* A, int64_t B, int64_t N, int64_t K, float* sq,
                                 void* ws, size_t ws_bytes, void* stream_) {
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
