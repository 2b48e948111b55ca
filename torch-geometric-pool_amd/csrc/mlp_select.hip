// A13: MLPSelect's last layer as ONE pass over the node features.
//
//   S[m, :] = softmax(X[m, :] W^T + b) * mask[m]          reference select/mlp_select.py:139-145
//
// (single Linear(F, K) with bias when in_channels is an int, mlp_select.py:67).  The reference runs F.linear, a softmax
// and a mask multiply: the [M, K] logits cross HBM three times and S twice.  Here every node row is read once and S is
// written once: HBM-bound, algorithmic bytes M * 4 * (F + K) (+ M mask bytes, + K * (F + 1) * 4 of parameters).
//
// mlp_select_mfma_kernel: a wave owns 32 node rows and all K columns.  The product is taken TRANSPOSED on the fp32
// matrix cores (v_mfma_f32_32x32x2_f32): A operand = a 32-column tile of W (LDS), B operand = X^T, so that in the C/D
// layout (col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) a lane holds ONE node row's logits in its
// registers -- 16 per 32-column tile, the other half of the row on lane ^ 32 -- and the softmax is register arithmetic
// plus two cross-half exchanges per row (the natural orientation would need 16 rows x 5-step lane reductions).  The
// order of k inside the dot product is free as long as both operands agree: half-wave h takes k = 32 c + 16 h + j
// (chunk c, j = 0..15), so a lane's X operand is 64 contiguous bytes per chunk (four 16-byte loads, a node row's
// 128-byte chunk shared by lanes l and l ^ 32) and its W operand four ds_read_b128 of a padded LDS row.
// W stays in LDS for the life of the workgroup when it fits 64 KB; otherwise the workgroup's four waves walk F in
// slices between barriers (W then comes from L2 once per 128 node rows).
//
// K > 256 (more than 8 accumulator tiles): the caller runs tgp_bmm_f32 and softmax_rows_kernel (in place).
// tgp_softmax_bwd_f32: dY = S * (dS - <dS, S>) per row -- with S = softmax * mask this is the gradient w.r.t. the
// logits for kept rows and 0 for masked rows (their S is 0).
#include <stdlib.h>

#include "common.h"

namespace tgp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MS_WAVES = 4;
constexpr int MS_LDS_BYTES = 64 * 1024;

struct MlpSelArgs {
  const float* x; const float* w; const float* bias; const unsigned char* mask;
  float* s;
  long M;
  int F, K;
  int fc;      // k's per LDS slice (multiple of 32); >= Fpad: W resident
  int ldw;     // LDS row stride in floats (fc + 4)
  long tiles;  // ceil(M / 32)
};

__device__ __forceinline__ float half_swap(float v) { return __shfl_xor(v, 32, 64); }

// exp(x) for x <= 0 on v_exp_f32 with the rounding error of x * log2(e) carried along (the library expf keeps two
// compare masks per call alive: 128 calls per lane spilled > 200 SGPRs).  Relative error ~2e-7; results below
// 2^-126 come out as 0.
__device__ __forceinline__ float exp_neg(float x) {
  const float L = 1.44269504088896341f, Llo = 1.92596299112661746e-8f;
  x = fmaxf(x, -104.f);  // 2^-150 = 0 in fp32; keeps -inf (padded columns) out of the error term (inf - inf)
  const float y = x * L;
  float r = fmaf(x, L, -y);
  r = fmaf(x, Llo, r);
  const float e = __builtin_amdgcn_exp2f(y);
  return fmaf(e, r * 0.693147180559945309f, e);
}

template <int KT, bool VEC>
__global__ __launch_bounds__(64 * MS_WAVES, KT > 6 ? 1 : 2) void mlp_select_mfma_kernel(MlpSelArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wl = smem;                       // [KT * 32][ldw]
  float* bl = smem + KT * 32 * p.ldw;     // [KT * 32] bias; -inf beyond K: padded columns drop out of the softmax
  const int lane = lane_id(), w = wave_id();
  const int lm = lane & 31, h = lane >> 5;
  const int F = p.F, K = p.K, fc = p.fc, ldw = p.ldw;
  const int Fpad = (F + 31) & ~31;
  const bool resident = fc >= Fpad;
  const int nslices = resident ? 1 : (Fpad + fc - 1) / fc;

  auto stage_w = [&](int k0) {  // W[:, k0 : k0 + fc] -> LDS, zero beyond K rows / F columns
    const int width = fc < Fpad - k0 ? fc : Fpad - k0;
    const int q4 = width >> 2;
    for (int i = threadIdx.x; i < KT * 32 * q4; i += 64 * MS_WAVES) {
      const int n = i / q4, q = i - n * q4;
      const int k = k0 + 4 * q;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < K) {
        const float* src = p.w + static_cast<long>(n) * F + k;
        if (VEC && k + 3 < F) {
          v = *reinterpret_cast<const float4*>(src);
        } else {
          if (k < F) v.x = src[0];
          if (k + 1 < F) v.y = src[1];
          if (k + 2 < F) v.z = src[2];
          if (k + 3 < F) v.w = src[3];
        }
      }
      *reinterpret_cast<float4*>(Wl + n * ldw + 4 * q) = v;
    }
  };

  for (int i = threadIdx.x; i < KT * 32; i += 64 * MS_WAVES) bl[i] = i < K ? (p.bias ? p.bias[i] : 0.f) : -INFINITY;
  if (resident) stage_w(0);
  __syncthreads();

  // rounds: MS_WAVES consecutive row tiles per workgroup per round (all waves take part in the barriers of the
  // sliced mode, so the loop bound is workgroup-uniform)
  const long rounds = (p.tiles + MS_WAVES - 1) / MS_WAVES;
  for (long r = blockIdx.x; r < rounds; r += gridDim.x) {
    const long tile = r * MS_WAVES + w;
    const long m = tile * 32 + lm;
    const bool row_ok = m < p.M;
    const float* xrow = p.x + (row_ok ? m : 0) * F;

    f32x16 acc[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = bl[32 * t + (i & 3) + 8 * (i >> 2) + 4 * h];  // bias (-inf: padding)

    for (int sl = 0; sl < nslices; ++sl) {
      const int k0 = sl * fc;
      if (!resident) {
        __syncthreads();  // every wave is done with the previous slice
        stage_w(k0);
        __syncthreads();
      }
      const int width = fc < Fpad - k0 ? fc : Fpad - k0;
      // X operand of chunk c: k = k0 + c + 16 h + j.  Two register sets: the next chunk's four 16-byte loads are requested
      // before this chunk's MFMAs, unconditionally in all rounds but the last, and pinned in front of them (r5, late:
      // left to itself the scheduler sank every load to the MFMAs that use it -- load, `s_waitcnt vmcnt(0)`, 16 MFMAs,
      // four times per chunk: eight dependent round trips per tile at F = 64).
      // unconditional loads from clamped addresses, zeroed afterwards (guards would put every load in a branch of
      // its own): with F % 4 == 0 a 16-byte vector is wholly inside or wholly outside the row
      auto request = [&](float4 (&dst)[4], int c) {
        const int kb = k0 + c + 16 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k = kb + 4 * q;
          if (VEC) {
            dst[q] = *reinterpret_cast<const float4*>(xrow + (k < F ? k : F - 4));
          } else {
            dst[q].x = xrow[k < F ? k : F - 1];
            dst[q].y = xrow[k + 1 < F ? k + 1 : F - 1];
            dst[q].z = xrow[k + 2 < F ? k + 2 : F - 1];
            dst[q].w = xrow[k + 3 < F ? k + 3 : F - 1];
          }
        }
      };
      auto consume = [&](const float4 (&src)[4], int c) {
        float xb[16];
        const int kb = k0 + c + 16 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k = kb + 4 * q;
          const bool in = row_ok && k < F;
          xb[4 * q] = in ? src[q].x : 0.f;
          xb[4 * q + 1] = (VEC ? in : row_ok && k + 1 < F) ? src[q].y : 0.f;
          xb[4 * q + 2] = (VEC ? in : row_ok && k + 2 < F) ? src[q].z : 0.f;
          xb[4 * q + 3] = (VEC ? in : row_ok && k + 3 < F) ? src[q].w : 0.f;
        }
#pragma unroll
        for (int t = 0; t < KT; ++t) {
          const float* wr = Wl + (32 * t + lm) * ldw + c + 16 * h;
          float wa[16];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(wr + 4 * q);
            wa[4 * q] = v.x; wa[4 * q + 1] = v.y; wa[4 * q + 2] = v.z; wa[4 * q + 3] = v.w;
          }
#pragma unroll
          for (int j = 0; j < 16; ++j) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[j], xb[j], acc[t], 0, 0, 0);
        }
      };
      float4 xa[4];
      [[maybe_unused]] float4 xn[4];
      if constexpr (KT > 4) {  // (the second register set does not fit next to more than four accumulator tiles)
        for (int c = 0; c < width; c += 32) {
          request(xa, c);
          consume(xa, c);
        }
      } else if (width > 0) {
        request(xa, 0);
        int c = 0;
        for (; c + 64 < width; c += 64) {
          request(xn, c + 32);
          __builtin_amdgcn_sched_barrier(0);
          consume(xa, c);
          request(xa, c + 64);
          __builtin_amdgcn_sched_barrier(0);
          consume(xn, c + 32);
        }
        if (c + 32 < width) {
          request(xn, c + 32);
          __builtin_amdgcn_sched_barrier(0);
          consume(xa, c);
          consume(xn, c + 32);
        } else {
          consume(xa, c);
        }
      }
    }

    // ---- softmax over the row held by lanes (lm, h = 0/1): column n = 32 t + (i & 3) + 8 (i >> 2) + 4 h
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        mx = fmaxf(mx, acc[t][i]);
      }
    mx = fmaxf(mx, half_swap(mx));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float e = exp_neg(acc[t][i] - mx);  // exp(-inf) = 0 on the padded columns
        acc[t][i] = e;
        sum += e;
      }
    sum += half_swap(sum);
    const float keep = (p.mask && row_ok) ? static_cast<float>(p.mask[m] != 0) : 1.f;
    if (row_ok) {
      float* srow = p.s + m * K;
      const float scale = keep / sum;  // one division per row (64 of them per lane spilled ~100 SGPRs of masks)
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        const bool full = VEC && 32 * t + 32 <= K;  // wave-uniform: whole tile inside the row
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = 32 * t + 8 * g + 4 * h;
          float4 v;
          v.x = acc[t][4 * g] * scale;
          v.y = acc[t][4 * g + 1] * scale;
          v.z = acc[t][4 * g + 2] * scale;
          v.w = acc[t][4 * g + 3] * scale;
          if (full) {
            *reinterpret_cast<float4*>(srow + n) = v;
          } else {
            int rem = K - n;  // opaque to the optimiser: the 16 KT loop-invariant store masks would otherwise be
            asm volatile("" : "+v"(rem));  // hoisted out of the row loop and spilled (92 SGPRs at KT = 4)
            if (rem > 0) srow[n] = v.x;
            if (rem > 1) srow[n + 1] = v.y;
            if (rem > 2) srow[n + 2] = v.z;
            if (rem > 3) srow[n + 3] = v.w;
          }
        }
        __builtin_amdgcn_sched_barrier(0);  // scale-and-store tile by tile (else all 16 KT products are formed first)
      }
    }
  }
}

// One wave per row, any K: softmax(y + b) * mask in place (fallback behind tgp_bmm_f32 for K > 256).
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                           const unsigned char* __restrict__ mask, long M, int K) {
  const long m = static_cast<long>(blockIdx.x) * 4 + wave_id();
  if (m >= M) return;
  const int lane = lane_id();
  float* row = y + m * K;
  float mx = -INFINITY;
  for (int k = lane; k < K; k += 64) mx = fmaxf(mx, row[k] + (bias ? bias[k] : 0.f));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sum = 0.f;
  for (int k = lane; k < K; k += 64) sum += expf(row[k] + (bias ? bias[k] : 0.f) - mx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float keep = mask ? static_cast<float>(mask[m] != 0) : 1.f;
  for (int k = lane; k < K; k += 64) row[k] = expf(row[k] + (bias ? bias[k] : 0.f) - mx) / sum * keep;
}

// dY[m, :] = S[m, :] * (dS[m, :] - sum_k dS[m, k] S[m, k]); G lanes per row.
template <int G>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ s, const float* __restrict__ ds,
                                                          float* __restrict__ dy, long M, int K) {
  const int sub = threadIdx.x % G;
  const long m = static_cast<long>(blockIdx.x) * (256 / G) + threadIdx.x / G;
  const bool ok = m < M;
  const float* sr = s + (ok ? m : 0) * K;
  const float* dr = ds + (ok ? m : 0) * K;
  float dot = 0.f;
  if (ok)
    for (int k = sub; k < K; k += G) dot = fmaf(dr[k], sr[k], dot);
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
  if (ok) {
    float* out = dy + m * K;
    for (int k = sub; k < K; k += G) out[k] = sr[k] * (dr[k] - dot);
  }
}

// r6 (training step at C2 scale): the same softmax backward with the elementwise parts of the pooling step's gradient
// folded in, so that they cost no pass of their own over [B,N,K]:
//   dS_eff = dS + extra + 2 c1[graph] deg[row] S   (MinCut: gradient of den = trace(S^T D S), utils/losses.py:39-56)
//                       - g_ent ent_scale (log(S + eps) + S / (S + eps))   (DiffPool's entropy loss, losses.py:476-483)
// `extra` (optional): an upstream gradient of S itself (S used outside the pooler).  rows_per_graph = N of the padded batch.
template <int G>
__global__ __launch_bounds__(256) void softmax_bwd_ex_kernel(const float* __restrict__ s, const float* __restrict__ ds,
                                                             const float* __restrict__ extra,
                                                             const float* __restrict__ c1, const float* __restrict__ deg,
                                                             long rows_per_graph, const float* __restrict__ ent_g,
                                                             float ent_scale, float ent_eps, float* __restrict__ dy,
                                                             long ld_dy, long M, int K,
                                                             const int64_t* __restrict__ batch) {
  const int sub = threadIdx.x % G;
  const long m = static_cast<long>(blockIdx.x) * (256 / G) + threadIdx.x / G;
  const bool ok = m < M;
  const long base = (ok ? m : 0) * K;
  const float* sr = s + base;
  const float* dr = ds + base;
  const float* er = extra ? extra + base : nullptr;
  // graph of the row: m / rows_per_graph (padded batch) or batch[m] (un-padded batch)
  const float rowc = (ok && c1) ? 2.0f * c1[batch ? batch[m] : m / rows_per_graph] * deg[m] : 0.f;
  const float ge = ent_g ? ent_g[0] * ent_scale : 0.f;
  auto eff = [&](int k) -> float {
    const float sv = sr[k];
    float v = dr[k];
    if (er) v += er[k];
    if (c1) v = fmaf(rowc, sv, v);
    if (ent_g) v -= ge * (logf(sv + ent_eps) + sv / (sv + ent_eps));
    return v;
  };
  float dot = 0.f;
  if (ok)
    for (int k = sub; k < K; k += G) dot = fmaf(eff(k), sr[k], dot);
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
  if (ok) {
    float* out = dy + m * ld_dy;
    for (int k = sub; k < K; k += G) out[k] = sr[k] * (eff(k) - dot);
  }
}

// ---- backward of the selector's last layer: one launch (+ a one-round combine of gW / gb) ------------------------------
// Given S = softmax(X W^T + b) * mask and the upstream dS:   dY = S (dS - <dS, S>)   (rows of masked nodes: S = 0 -> 0),
//   gX (+)= dY W      [M,F]        gW = dY^T X   [K,F]        gb = column sums of dY   [K]
// (what autograd derives from select/mlp_select.py:139-145: softmax backward, two matmul backwards, a bias reduction:
// eight launches here before -- softmax_bwd, a GEMM + its split sum, a small bmm, two reductions with their memsets and
// autograd's accumulation into gX).  K <= 32, F <= 64 (four feature tiles per wave spill: wider layers keep the staged
// form).  A wave owns 64 node rows at a time: S and dS tiles are staged flat through LDS (coalesced), lane = row forms dY in registers and drops it back into LDS as [64][33]; both products
// run on the fp32 matrix cores from that tile: gX = dY W with W held in registers as the B operand (lane = feature),
// gW += dY^T X with X straight from memory as the B operand (lane = feature, k = row: a half-wave reads 128 contiguous
// bytes of one row) and dY^T read down the LDS columns (lane = cluster: conflict-free); gb rides on the A operands.
// The per-wave gW / gb accumulators are added in wave order through LDS, every workgroup stores one partial, and
// mlp_select_bwd_combine_kernel adds the partials in workgroup order (deterministic; no float atomics).
struct MlpSelBwdArgs {
  const float* s; const float* gs; const float* x; const float* w;
  float* gx; float* gw; float* gb;
  float* part;            // [gridDim.x][K * F + K]: one partial of gW | gb per workgroup
  long M, tiles;
  int K, F, accumulate;
};

constexpr int MB_WAVES = 8;
constexpr int MB_MAX_GRID = 256;  // workgroups (= partials of gW / gb) per launch
constexpr int MB_ZP = 33;
constexpr int MB_WAVE_FLOATS = 4096 + 64;  // flat S | dS tiles (2 x 2048), then dY [64][33]; at the end gW [32][32 FT] + gb [2][32]

__device__ __forceinline__ int mb_rho(int r) { return (r & 3) + 8 * (r >> 2); }

template <int FT>
__global__ __launch_bounds__(64 * MB_WAVES) void mlp_select_bwd_kernel(MlpSelBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = lane_id(), w = __builtin_amdgcn_readfirstlane(wave_id());
  const int lm = lane & 31, lk = lane >> 5;
  const int K = p.K, F = p.F;
  float* buf = smem + w * MB_WAVE_FLOATS;
  constexpr bool WREG = FT <= 2;
  float wreg[WREG ? 16 : 1][FT];
  auto w_at = [&](int kk, int ft) {
    const int c = 2 * kk + lk, f = ft * 32 + lm;
    return (c < K && f < F) ? p.w[c * F + f] : 0.f;
  };
  if constexpr (WREG) {
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
#pragma unroll
      for (int ft = 0; ft < FT; ++ft) wreg[kk][ft] = w_at(kk, ft);
  }
  f32x16 dw[FT];
#pragma unroll
  for (int ft = 0; ft < FT; ++ft)
#pragma unroll
    for (int r = 0; r < 16; ++r) dw[ft][r] = 0.f;
  float db = 0.f;
  for (long tile = static_cast<long>(blockIdx.x) * MB_WAVES + w; tile < p.tiles;
       tile += static_cast<long>(gridDim.x) * MB_WAVES) {
    const long row0 = tile * 64;
    const int nrows = static_cast<int>(p.M - row0 < 64 ? p.M - row0 : 64);
    const int cnt = nrows * K;
    // Buffer-descriptor loads: one lane offset per stream + scalar / immediate offsets, out-of-range dwords come back as
    // zeros (rows past M in the last tile), dword alignment suffices for the 16-byte loads.  Every global load of the
    // tile is requested before the first one is consumed: S and dS (staged flat through LDS), the X operand of gW for
    // the first 32 features, and -- one feature tile, accumulating -- the gX values to add to.
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.s + row0 * K), 0, cnt * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gs + row0 * K), 0, cnt * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + row0 * F), 0, nrows * F * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(p.gx ? p.gx + row0 * F : const_cast<float*>(p.x), 0,
                                                                         p.gx ? nrows * F * 4 : 0, 0x00020000);
    u32x4 sv[8], gv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      sv[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, q * 1024, 0);
      gv[q] = __builtin_amdgcn_raw_buffer_load_b128(rg, lane * 16, q * 1024, 0);
    }
    // (gfx950 checks voffset + soffset against the record size: tools/micro/buffer_range_soffset.hip)
    constexpr int OOB = 0x40000000;  // byte offset past any tile: a lane past the last feature reads zero / stores nothing
    const int vx = lm < F ? (lk * F + lm) * 4 : OOB, vo = lm < F ? (4 * lk * F + lm) * 4 : OOB;
    float xr[32];
#pragma unroll
    for (int kk = 0; kk < 32; ++kk)
      xr[kk] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vx, kk * 2 * F * 4, 0));
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      *reinterpret_cast<u32x4*>(buf + (q * 64 + lane) * 4) = sv[q];
      *reinterpret_cast<u32x4*>(buf + 2048 + (q * 64 + lane) * 4) = gv[q];
    }
    float old[(FT == 1) ? 32 : 1];
    if constexpr (FT == 1) {
      if (p.gx && p.accumulate) {
#pragma unroll
        for (int q = 0; q < 32; ++q)
          old[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
              ro, vo, ((q >> 4) * 32 + mb_rho(q & 15)) * F * 4, 0));
      }
    }
    __builtin_amdgcn_wave_barrier();
    float gz[32];
    {
      const bool live = lane < nrows;
      const float* S = buf + lane * K;
      const float* G = buf + 2048 + lane * K;
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < 32; ++c)
        if (c < K && live) dot = fmaf(G[c], S[c], dot);
#pragma unroll
      for (int c = 0; c < 32; ++c) gz[c] = (c < K && live) ? S[c] * (G[c] - dot) : 0.f;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < 32; ++c) buf[lane * MB_ZP + c] = gz[c];
    __builtin_amdgcn_wave_barrier();
    if (p.gx) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) {
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
          for (int kk = 0; kk < 16; ++kk)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(buf[(rt * 32 + lm) * MB_ZP + 2 * kk + lk],
                                                       WREG ? wreg[kk][ft] : w_at(kk, ft), acc, 0, 0, 0);
          const int f = ft * 32 + lm;
          const int voff = f < F ? (4 * lk * F + f) * 4 : OOB;  // (stores out of range are dropped)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int soff = (rt * 32 + mb_rho(r)) * F * 4;
            float v = acc[r];
            if (p.accumulate) {
              if constexpr (FT == 1) v = __fadd_rn(old[rt * 16 + r], v);
              else v = __fadd_rn(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ro, voff, soff, 0)), v);
            }
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ro, voff, soff, 0);
          }
        }
    }
    if (p.gw) {
#pragma unroll
      for (int ft = 0; ft < FT; ++ft) {
        if (ft > 0) {
          const int f = ft * 32 + lm, vx2 = f < F ? (lk * F + f) * 4 : OOB;
#pragma unroll
          for (int kk = 0; kk < 32; ++kk)
            xr[kk] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vx2, kk * 2 * F * 4, 0));
        }
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
          const float a = buf[(2 * kk + lk) * MB_ZP + lm];
          dw[ft] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xr[kk], dw[ft], 0, 0, 0);
          if (ft == 0) db += a;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (!p.gw) return;  // (uniform)
  // ---- workgroup partial: waves added in wave order ---------------------------------------------------------------------
  constexpr int FW = 32 * FT;
#pragma unroll
  for (int ft = 0; ft < FT; ++ft)
#pragma unroll
    for (int r = 0; r < 16; ++r) buf[(mb_rho(r) + 4 * lk) * FW + ft * 32 + lm] = dw[ft][r];
  buf[4096 + lk * 32 + lm] = db;
  __syncthreads();
  const int n_out = K * F + K;
  float* mine = p.part + static_cast<long>(blockIdx.x) * n_out;
  for (int o = threadIdx.x; o < n_out; o += 64 * MB_WAVES) {
    float v = 0.f;
    if (o < K * F) {
      const int c = o / F, f = o - c * F;
#pragma unroll
      for (int q = 0; q < MB_WAVES; ++q) v = __fadd_rn(v, smem[q * MB_WAVE_FLOATS + c * FW + f]);
    } else {
      const int c = o - K * F;
#pragma unroll
      for (int q = 0; q < MB_WAVES; ++q)
        v = __fadd_rn(v, __fadd_rn(smem[q * MB_WAVE_FLOATS + 4096 + c], smem[q * MB_WAVE_FLOATS + 4096 + 32 + c]));
    }
    mine[o] = v;
  }
}

// Second launch of the selector's backward: gW / gb = the workgroups' partials added in workgroup order.  One workgroup
// per 64 outputs; wave q takes partials q, q + 8, ... (all its loads in flight at once: the grid has at most 256
// partials), the eight waves' sums are added in wave order.  (Adding them inside the first launch -- the workgroup with
// the last arrival ticket, also as two ticket levels with write-through partials -- was measured: 35 us against
// 17.6 + 4 here; a hand-over through memory is five dependent round trips at 2-3 us each, profiles/r05_train_step.md.)
__global__ __launch_bounds__(64 * MB_WAVES) void mlp_select_bwd_combine_kernel(const float* __restrict__ part, int P,
                                                                                int n_out, int kf, float* __restrict__ gw,
                                                                                float* __restrict__ gb) {
  __shared__ float s_sum[MB_WAVES][64];
  const int lane = lane_id(), w = wave_id();
  const int o = blockIdx.x * 64 + lane;
  float v[MB_MAX_GRID / MB_WAVES];
#pragma unroll
  for (int ji = 0; ji < MB_MAX_GRID / MB_WAVES; ++ji) {
    const int j = w + ji * MB_WAVES;
    const bool ok = o < n_out && j < P;
    const float got = part[ok ? static_cast<long>(j) * n_out + o : 0];
    v[ji] = ok ? got : 0.f;
  }
  float a = 0.f;
#pragma unroll
  for (int ji = 0; ji < MB_MAX_GRID / MB_WAVES; ++ji) a = __fadd_rn(a, v[ji]);
  s_sum[w][lane] = a;
  __syncthreads();
  if (w == 0 && o < n_out) {
    float r = 0.f;
#pragma unroll
    for (int q = 0; q < MB_WAVES; ++q) r = __fadd_rn(r, s_sum[q][lane]);
    if (o < kf) gw[o] = r;
    else if (gb) gb[o - kf] = r;
  }
}

template <int KT>
static int launch_mlp_select(const MlpSelArgs& a, bool vec, int grid, size_t lds, hipStream_t st) {
  if (vec)
    hipLaunchKernelGGL((mlp_select_mfma_kernel<KT, true>), dim3(grid), dim3(64 * MS_WAVES), lds, st, a);
  else
    hipLaunchKernelGGL((mlp_select_mfma_kernel<KT, false>), dim3(grid), dim3(64 * MS_WAVES), lds, st, a);
  return check_launch("mlp_select_mfma_kernel");
}

}  // namespace tgp

using namespace tgp;

extern "C" int tgp_mlp_select_max_fused_k(void) { return 256; }

extern "C" int tgp_mlp_select_f32(const float* x, const float* weight, const float* bias, const unsigned char* mask,
                                  int64_t M, int64_t F, int64_t K, float* s_out, void* stream) {
  TGP_REQUIRE(M >= 0 && F >= 1 && K >= 1, TGP_ERR_INVALID, "tgp_mlp_select_f32: bad shape M=%lld F=%lld K=%lld",
              (long long)M, (long long)F, (long long)K);
  TGP_REQUIRE(K <= 256, TGP_ERR_INVALID, "tgp_mlp_select_f32: K=%lld > 256: use tgp_bmm_f32 + tgp_softmax_rows_f32",
              (long long)K);
  TGP_REQUIRE(F < (1 << 24), TGP_ERR_RANGE, "tgp_mlp_select_f32: F too large");
  if (M == 0) return TGP_OK;
  TGP_REQUIRE(x && weight && s_out, TGP_ERR_INVALID, "tgp_mlp_select_f32: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int KT = static_cast<int>((K + 31) / 32);
  const int Fpad = static_cast<int>((F + 31) & ~31ll);
  // slice width: the largest multiple of 32 whose [KT*32][fc+4] image (+ bias) fits MS_LDS_BYTES
  const int budget = (MS_LDS_BYTES - KT * 32 * 4) / (KT * 32 * 4) - 4;
  int fc = budget / 32 * 32;
  if (fc >= Fpad) fc = Fpad;
  TGP_REQUIRE(fc >= 32, TGP_ERR_INVALID, "tgp_mlp_select_f32: LDS budget");
  MlpSelArgs a;
  a.x = x; a.w = weight; a.bias = bias; a.mask = mask; a.s = s_out;
  a.M = M; a.F = static_cast<int>(F); a.K = static_cast<int>(K);
  a.fc = fc; a.ldw = fc + 4;
  a.tiles = (M + 31) / 32;
  const size_t lds = static_cast<size_t>(KT) * 32 * (a.ldw + 1) * sizeof(float);
  const long rounds = (a.tiles + MS_WAVES - 1) / MS_WAVES;
  const int cus = tgp_device_cu_count();
  long grid = static_cast<long>(cus > 0 ? cus : 256) * (KT <= 2 ? 4 : KT <= 6 ? 2 : 1);  // workgroups resident per CU (registers)
  if (grid > rounds) grid = rounds;
  const bool vec = (F % 4 == 0) && (K % 4 == 0) && reinterpret_cast<uintptr_t>(x) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(weight) % 16 == 0 && reinterpret_cast<uintptr_t>(s_out) % 16 == 0;
  switch (KT) {
    case 1: return launch_mlp_select<1>(a, vec, (int)grid, lds, st);
    case 2: return launch_mlp_select<2>(a, vec, (int)grid, lds, st);
    case 3: return launch_mlp_select<3>(a, vec, (int)grid, lds, st);
    case 4: return launch_mlp_select<4>(a, vec, (int)grid, lds, st);
    case 5: return launch_mlp_select<5>(a, vec, (int)grid, lds, st);
    case 6: return launch_mlp_select<6>(a, vec, (int)grid, lds, st);
    case 7: return launch_mlp_select<7>(a, vec, (int)grid, lds, st);
    default: return launch_mlp_select<8>(a, vec, (int)grid, lds, st);
  }
}

extern "C" int tgp_softmax_rows_f32(float* y, const float* bias, const unsigned char* mask, int64_t M, int64_t K,
                                    void* stream) {
  TGP_REQUIRE(M >= 0 && K >= 1 && K < (1ll << 31), TGP_ERR_INVALID, "tgp_softmax_rows_f32: bad shape");
  if (M == 0) return TGP_OK;
  TGP_REQUIRE(y, TGP_ERR_INVALID, "tgp_softmax_rows_f32: null pointer");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv(M, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), y, bias,
                     mask, static_cast<long>(M), static_cast<int>(K));
  return check_launch("softmax_rows_kernel");
}

extern "C" int tgp_softmax_bwd_f32(const float* s, const float* ds, float* dy, int64_t M, int64_t K, void* stream) {
  TGP_REQUIRE(M >= 0 && K >= 1 && K < (1ll << 31), TGP_ERR_INVALID, "tgp_softmax_bwd_f32: bad shape");
  if (M == 0) return TGP_OK;
  TGP_REQUIRE(s && ds && dy, TGP_ERR_INVALID, "tgp_softmax_bwd_f32: null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (K <= 32)
    hipLaunchKernelGGL((softmax_bwd_kernel<16>), dim3(cdiv(M, 16)), dim3(256), 0, st, s, ds, dy, (long)M, (int)K);
  else
    hipLaunchKernelGGL((softmax_bwd_kernel<64>), dim3(cdiv(M, 4)), dim3(256), 0, st, s, ds, dy, (long)M, (int)K);
  return check_launch("softmax_bwd_kernel");
}

extern "C" int tgp_softmax_bwd_ex_f32(const float* s, const float* ds, const float* extra, const float* c1,
                                      const float* deg, int64_t rows_per_graph, const float* ent_g, float ent_scale,
                                      float ent_eps, float* dy, int64_t ld_dy, int64_t M, int64_t K,
                                      const int64_t* batch, void* stream_) {
  TGP_REQUIRE(M >= 0 && K >= 1 && K < (1ll << 31) && ld_dy >= K, TGP_ERR_INVALID, "tgp_softmax_bwd_ex_f32: bad shape");
  if (M == 0) return TGP_OK;
  TGP_REQUIRE(s && ds && dy && (!c1 || (deg && (rows_per_graph > 0 || batch))), TGP_ERR_INVALID,
              "tgp_softmax_bwd_ex_f32: null pointer");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int k = static_cast<int>(K);
  if (K <= 16)
    hipLaunchKernelGGL(softmax_bwd_ex_kernel<16>, dim3(cdiv(M, 16)), dim3(256), 0, stream, s, ds, extra, c1, deg,
                       static_cast<long>(rows_per_graph), ent_g, ent_scale, ent_eps, dy, static_cast<long>(ld_dy),
                       static_cast<long>(M), k, batch);
  else
    hipLaunchKernelGGL(softmax_bwd_ex_kernel<64>, dim3(cdiv(M, 4)), dim3(256), 0, stream, s, ds, extra, c1, deg,
                       static_cast<long>(rows_per_graph), ent_g, ent_scale, ent_eps, dy, static_cast<long>(ld_dy),
                       static_cast<long>(M), k, batch);
  return check_launch("tgp_softmax_bwd_ex_f32");
}

extern "C" int tgp_mlp_select_bwd_fits(int64_t F, int64_t K) { return K >= 1 && K <= 32 && F >= 1 && F <= 64; }

static int mlp_select_bwd_grid(int64_t M) {
  // few, fat workgroups: every workgroup leaves one partial of gW / gb that the last one adds up
  const int64_t tiles = (M + 63) / 64;
  int64_t g = (tiles + MB_WAVES - 1) / MB_WAVES;
  if (g > MB_MAX_GRID) g = MB_MAX_GRID;
  return g < 1 ? 1 : static_cast<int>(g);
}

extern "C" size_t tgp_mlp_select_bwd_workspace_bytes(int64_t M, int64_t F, int64_t K) {
  return align_up(static_cast<size_t>(mlp_select_bwd_grid(M)) * static_cast<size_t>(K * F + K) * sizeof(float));
}

extern "C" int tgp_mlp_select_bwd_f32(const float* s, const float* ds, const float* x, const float* weight, int64_t M,
                                      int64_t F, int64_t K, float* gx, int accumulate_gx, float* gw, float* gb,
                                      void* ws, size_t ws_bytes, void* stream) {
  TGP_REQUIRE(M >= 0 && tgp_mlp_select_bwd_fits(F, K), TGP_ERR_INVALID,
              "tgp_mlp_select_bwd_f32: M=%lld F=%lld K=%lld outside K <= 32, F <= 64", (long long)M, (long long)F,
              (long long)K);
  TGP_REQUIRE(M * (F > K ? F : K) < (1ll << 40), TGP_ERR_RANGE, "tgp_mlp_select_bwd_f32: too many rows");
  if (M == 0) {
    hipStream_t st0 = static_cast<hipStream_t>(stream);
    if (gw) (void)hipMemsetAsync(gw, 0, sizeof(float) * K * F, st0);
    if (gb) (void)hipMemsetAsync(gb, 0, sizeof(float) * K, st0);
    return check_launch("tgp_mlp_select_bwd_f32");
  }
  TGP_REQUIRE(s && ds && x && weight && (gx || gw), TGP_ERR_INVALID, "tgp_mlp_select_bwd_f32: null pointer");
  TGP_REQUIRE(!gb || gw, TGP_ERR_INVALID, "tgp_mlp_select_bwd_f32: gb comes with gw");
  TGP_REQUIRE(!gw || (ws && ws_bytes >= tgp_mlp_select_bwd_workspace_bytes(M, F, K)), TGP_ERR_INVALID,
              "tgp_mlp_select_bwd_f32: workspace missing or too small");
  MlpSelBwdArgs a{s, ds, x, weight, gx, gw, gb, static_cast<float*>(ws), static_cast<long>(M),
                  static_cast<long>((M + 63) / 64), static_cast<int>(K), static_cast<int>(F), accumulate_gx ? 1 : 0};
  const int grid = mlp_select_bwd_grid(M);
  const size_t lds = static_cast<size_t>(MB_WAVES) * MB_WAVE_FLOATS * sizeof(float);
  hipStream_t st = static_cast<hipStream_t>(stream);
  auto go = [&](auto kern) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              static_cast<int>(lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * MB_WAVES), lds, st, a);
  };
  if (F <= 32) go(mlp_select_bwd_kernel<1>);
  else go(mlp_select_bwd_kernel<2>);
  if (gw) {
    const int n_out = static_cast<int>(K * F + K);
    hipLaunchKernelGGL(mlp_select_bwd_combine_kernel, dim3(cdiv(n_out, 64)), dim3(64 * MB_WAVES), 0, st,
                       static_cast<const float*>(ws), grid, n_out, static_cast<int>(K * F), gw, gb);
  }
  return check_launch("mlp_select_bwd_kernel");
}
