// A8: split-K slab combine + dense post-processing kernels (utils/ops.py:282-335) and their launcher.
#pragma once
#include <stdlib.h>

#include "common.h"

namespace tgp {

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
  float eps;         // clamp of the degree vector: the caller's eps at call time (ops.py:318)
  float* raw;        // optional [B][K][K]
  float* dst;        // optional [B][K][K]
  float* dvec;       // [B][K]
  float* maxpart;    // [B][POST_BLOCKS]
  // r6 (post_lds_kernel only, K % 16 == 0): the slabs hold the TRANSPOSE of the matrix to post-process (the rows route's
  // S^T (A S) when the batched poolers want S^T A^T S): summed slab element (a, b) becomes element (b, a) of raw / dst
  int transpose_src;
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
      p.dvec[static_cast<long>(b) * K + j] = sqrtf(fmaxf(t, p.eps));  // sqrt(clamp(d, eps)): ops.py:318
    }
  } else {
    for (int r = w; r < 64; r += 16) {
      const int i = base + r;
      if (i >= K) break;
      float s = 0.f;
      for (int j = lane; j < K; j += 64) s = __fadd_rn(s, a[static_cast<long>(i) * K + j]);
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) s = __fadd_rn(s, __shfl_down(s, d, WAVE));
      if (lane == 0) p.dvec[static_cast<long>(b) * K + i] = sqrtf(fmaxf(s, p.eps));
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
    d = sqrtf(fmaxf(mine, p.eps));  // d[lane]
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
    const float d = sqrtf(fmaxf(__shfl(sum, by_cols ? idx : 32 + idx, WAVE), p.eps));  // d[idx] on every lane
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

__global__ __launch_bounds__(1024) void post_lds_kernel(PostArgs p, int B, XCombineArgs xc, XCombineArgs xc2) {
  extern __shared__ __attribute__((aligned(16))) float m[];
  const int tid = threadIdx.x;
  if (static_cast<int>(blockIdx.x) >= B) {  // ---- X' slab combine (r6: and a second one, the Gram slabs) -----------
    int xb = blockIdx.x - B;
    if (xb >= B * xc.blocks_per_graph) {
      xb -= B * xc.blocks_per_graph;
      xc = xc2;
    }
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
  if (p.transpose_src) {
    // 16 x 16 patches: 4 lanes cover 64 contiguous bytes of a source row, 16 lanes the patch's 16 source rows; the
    // transposed stores then run along the 16 source rows: 64-byte runs in memory, 16 distinct LDS banks (4-way conflicts)
    const int kt = K >> 4;
    for (int v = tid; v < kk / 4; v += 1024) {
      const int lb = v & 3, la = (v >> 2) & 15, tile = v >> 6;
      const int a = 16 * (tile / kt) + la, bq = 16 * (tile % kt) + 4 * lb;
      const int e = a * K + bq;
      float4 t = *reinterpret_cast<const float4*>(sb + e);
      for (int sp = 1; sp < p.splits; ++sp) {
        const float4 u = *reinterpret_cast<const float4*>(sb + sp * p.s_split + e);
        t.x = __fadd_rn(t.x, u.x); t.y = __fadd_rn(t.y, u.y); t.z = __fadd_rn(t.z, u.z); t.w = __fadd_rn(t.w, u.w);
      }
      const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int i = bq + c;  // destination row = source column
        if (rawb) rawb[i * K + a] = tv[c];
        m[i * K + a] = (rsl && i == a) ? 0.f : tv[c];
      }
    }
  } else
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
          dv[j] = sqrtf(fmaxf(t, p.eps));
        }
        __syncthreads();
      }
    } else {                            // row sums: lanes stride over the columns, fixed shuffle tree
      for (int i = w; i < K; i += 16) {
        float sacc = 0.f;
        for (int j = lane; j < K; j += 64) sacc = __fadd_rn(sacc, m[i * K + j]);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) sacc = __fadd_rn(sacc, __shfl_down(sacc, d, WAVE));
        if (lane == 0) dv[i] = sqrtf(fmaxf(sacc, p.eps));
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

// r5: the same post-processing SPREAD over K / 16 workgroups per graph (C2: 9.5 us for one 1024-thread workgroup per
// graph -- 32 CUs walking four dependent phases over 256 KB each -- against ~5 us here).  A workgroup owns 16 rows of
// A'.  It does not read the other rows to learn their degrees: the second product's epilogue left partial COLUMN sums
// per (split, row tile, column) (GemmArgs.colsum, the diagonal left out when the self loops are removed), so the degree
// vector d_j = sqrt(max(sum_i R1_ij, eps)) of utils/ops.py:311-320 (sum over dim -2, R1 = R with its diagonal cleared)
// costs splits * tiles_m numbers per column.  Only for TGP_SUM_AXIS_ROWS (column sums: the connectors' default adj_transpose=True) without
// edge_weight_norm; K % 4 == 0.  The summation order of d differs from post_lds_kernel's (partials per 64-row tile
// first): both orders are fixed, results agree to rounding.
struct PostRowsArgs {
  const float* slab; int splits; long s_split, s_batch;  // [B][splits][K][K]
  const float* colsum; int tiles_m;                       // [B][splits][tiles_m][K] or NULL (no degree norm)
  int K, flags; float eps;
  float* raw; float* dst;                                 // [B][K][K], raw optional
  const float* xslab; long xs_split, xs_batch; int F; float* x_pool;  // [B][splits][K][F] -> [B][K][F], or NULL
  // r6 (training step): workgroups >= main_blocks add up a second slab set of the same shape (S^T S) into `gram`,
  // no post-processing
  int main_blocks; const float* gslab; float* gram;
};

constexpr int PR_ROWS = 16;
constexpr int PR_MAX_SPLITS = 8;  // node-range splits of the second product this kernel takes (x 2 row tiles for K <= 128...)

__global__ __launch_bounds__(512) void post_rows_kernel(PostRowsArgs p) {
  __shared__ float s_d[1024];
  const int K = p.K, tid = threadIdx.x;
  const int blocks_per_graph = (K + PR_ROWS - 1) / PR_ROWS;
  if (p.gram && static_cast<int>(blockIdx.x) >= p.main_blocks) {  // (workgroup-uniform) fixed-order slab sum of S^T S
    const int id = blockIdx.x - p.main_blocks;
    const int gb_ = id / blocks_per_graph, grb = id - gb_ * blocks_per_graph;
    const int rows = K - grb * PR_ROWS < PR_ROWS ? K - grb * PR_ROWS : PR_ROWS;
    const long base = static_cast<long>(grb) * PR_ROWS * K;
    const float* src = p.gslab + static_cast<long>(gb_) * p.s_batch + base;
    float* dst = p.gram + static_cast<long>(gb_) * K * K + base;
    for (int e = tid * 4; e < rows * K; e += 2048) {
      float4 t = *reinterpret_cast<const float4*>(src + e);
      for (int sp = 1; sp < p.splits; ++sp) {
        const float4 v = *reinterpret_cast<const float4*>(src + sp * p.s_split + e);
        t.x = __fadd_rn(t.x, v.x); t.y = __fadd_rn(t.y, v.y); t.z = __fadd_rn(t.z, v.z); t.w = __fadd_rn(t.w, v.w);
      }
      *reinterpret_cast<float4*>(dst + e) = t;
    }
    return;
  }
  const int b = blockIdx.x / blocks_per_graph, rb = blockIdx.x - b * blocks_per_graph;
  const int i0 = rb * PR_ROWS;
  const float* sb = p.slab + static_cast<long>(b) * p.s_batch;
  const bool rsl = p.flags & TGP_REMOVE_SELF_LOOPS, dn = (p.flags & TGP_DEGREE_NORM) && p.colsum;
  const int kq = K >> 2;  // float4 per row
  // this workgroup's elements: requested first, consumed behind the degree vector
  float4 acc[2];
  bool mine[2];
  int ei[2], ej[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = tid + u * 512;  // float4 index inside the 16-row block
    const int il = e / kq, jq = e - il * kq;
    ei[u] = i0 + il; ej[u] = jq * 4;
    mine[u] = il < PR_ROWS && ei[u] < K;
    // every slab's value is requested before the first add (a loop with a run-time trip count waits per load)
    float4 v[PR_MAX_SPLITS];
    const float* q = sb + (mine[u] ? static_cast<long>(ei[u]) * K + ej[u] : 0);
#pragma unroll
    for (int sp = 0; sp < PR_MAX_SPLITS; ++sp)
      v[sp] = *reinterpret_cast<const float4*>(q + (sp < p.splits ? sp * p.s_split : 0));
    float4 t = v[0];
#pragma unroll
    for (int sp = 1; sp < PR_MAX_SPLITS; ++sp)
      if (sp < p.splits) {
        t.x = __fadd_rn(t.x, v[sp].x); t.y = __fadd_rn(t.y, v[sp].y);
        t.z = __fadd_rn(t.z, v[sp].z); t.w = __fadd_rn(t.w, v[sp].w);
      }
    acc[u] = t;
  }
  // x_pool rows of the same block (fixed-order slab sum): the first two elements per thread are requested here as well,
  // so that everything this workgroup reads is one round trip
  const int xrows = K - i0 < PR_ROWS ? K - i0 : PR_ROWS;
  const int xcount = p.x_pool ? xrows * p.F : 0;
  const float* xb = p.x_pool ? p.xslab + static_cast<long>(b) * p.xs_batch + static_cast<long>(i0) * p.F : sb;
  float xv[2][PR_MAX_SPLITS];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = tid + u * 512;
#pragma unroll
    for (int sp = 0; sp < PR_MAX_SPLITS; ++sp)
      xv[u][sp] = xb[(e < xcount && sp < p.splits) ? sp * p.xs_split + e : 0];
  }
  if (dn) {
    const int parts = p.splits * p.tiles_m;  // <= 2 * PR_MAX_SPLITS: all in flight at once
    for (int j = tid; j < K; j += 512) {
      const float* cs = p.colsum + static_cast<long>(b) * parts * K + j;
      float pv[2 * PR_MAX_SPLITS];
#pragma unroll
      for (int qq = 0; qq < 2 * PR_MAX_SPLITS; ++qq) pv[qq] = cs[qq < parts ? static_cast<long>(qq) * K : 0];
      float c = 0.f;  // (the partial sums already leave the diagonal out when the self loops are removed)
#pragma unroll
      for (int qq = 0; qq < 2 * PR_MAX_SPLITS; ++qq)
        if (qq < parts) c = __fadd_rn(c, pv[qq]);
      s_d[j] = sqrtf(fmaxf(c, p.eps));
    }
  }
  __syncthreads();
  float* rawb = p.raw ? p.raw + static_cast<long>(b) * K * K : nullptr;
  float* dstb = p.dst ? p.dst + static_cast<long>(b) * K * K : nullptr;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    if (!mine[u]) continue;
    float4 t = acc[u];
    const long off = static_cast<long>(ei[u]) * K + ej[u];
    if (rawb) *reinterpret_cast<float4*>(rawb + off) = t;
    if (!dstb) continue;
    if (rsl) {
      if (ei[u] == ej[u]) t.x = 0.f;
      if (ei[u] == ej[u] + 1) t.y = 0.f;
      if (ei[u] == ej[u] + 2) t.z = 0.f;
      if (ei[u] == ej[u] + 3) t.w = 0.f;
    }
    if (dn) {  // (adj / d[1,K]) / d[K,1]
      const float di = s_d[ei[u]];
      const float4 dj = *reinterpret_cast<const float4*>(s_d + ej[u]);
      t.x = (t.x / dj.x) / di; t.y = (t.y / dj.y) / di; t.z = (t.z / dj.z) / di; t.w = (t.w / dj.w) / di;
    }
    *reinterpret_cast<float4*>(dstb + off) = t;
  }
  if (p.x_pool) {
    float* xo = p.x_pool + static_cast<long>(b) * K * p.F + static_cast<long>(i0) * p.F;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + u * 512;
      if (e < xcount) {
        float v = xv[u][0];
#pragma unroll
        for (int sp = 1; sp < PR_MAX_SPLITS; ++sp)
          if (sp < p.splits) v = __fadd_rn(v, xv[u][sp]);
        xo[e] = v;
      }
    }
    for (int e = tid + 1024; e < xcount; e += 512) {  // F > 64
      float v = xb[e];
      for (int sp = 1; sp < p.splits; ++sp) v = __fadd_rn(v, xb[sp * p.xs_split + e]);
      xo[e] = v;
    }
  }
}

// K <= 256 so that a 16-row block is at most two float4 per thread and d fits s_d
static bool post_rows_ok(int64_t K, int flags, const void* slab, const void* raw, const void* dst) {
  static const int off = getenv("TGP_NO_POST_ROWS") ? atoi(getenv("TGP_NO_POST_ROWS")) : 0;
  return !off && K % 4 == 0 && K > 64 && K <= 256 && !(flags & TGP_EDGE_WEIGHT_NORM) &&
         (!(flags & TGP_DEGREE_NORM) || (flags & TGP_SUM_AXIS_ROWS)) && reinterpret_cast<uintptr_t>(slab) % 16 == 0 &&
         (!raw || reinterpret_cast<uintptr_t>(raw) % 16 == 0) && (!dst || reinterpret_cast<uintptr_t>(dst) % 16 == 0);
}

static size_t post_ws_floats(int64_t B, int64_t K) { return static_cast<size_t>(B * K + B * POST_BLOCKS); }

// does launch_post hand this problem to post_lds_kernel (the only kernel that honours PostArgs.transpose_src)?
static bool post_lds_takes(const PostArgs& p) {
  static const int no_lds = getenv("TGP_NO_POST_LDS") ? 1 : 0;
  const int K = p.K;
  return !no_lds && K > 64 && K <= POST_LDS_MAX_K && K % 4 == 0 && p.ld_src == K && p.s_split % 4 == 0 &&
         p.s_batch % 4 == 0 && reinterpret_cast<uintptr_t>(p.src) % 16 == 0 &&
         (!p.raw || reinterpret_cast<uintptr_t>(p.raw) % 16 == 0) && (!p.dst || reinterpret_cast<uintptr_t>(p.dst) % 16 == 0);
}

// Returns whether the slab combines described by xc / xc2 (if any) were folded into the launch (both or none).
static bool launch_post(PostArgs p, int64_t B, float* ws, hipStream_t stream, const XCombineArgs* xc = nullptr,
                        const XCombineArgs* xc2 = nullptr) {
  const int K = p.K;
  if (K <= 64) {  // one wave per graph: one launch instead of three or four
    const dim3 grid(static_cast<unsigned>((B + 3) / 4));
    if (K <= 32) hipLaunchKernelGGL(post_tiny_kernel, grid, dim3(256), 0, stream, p, static_cast<int>(B));
    else hipLaunchKernelGGL(post_small_kernel, grid, dim3(256), 0, stream, p, static_cast<int>(B));
    return false;
  }
  if (post_lds_takes(p)) {
    XCombineArgs x{}, x2{};
    auto blocks_of = [](XCombineArgs& a) {
      a.blocks_per_graph = static_cast<int>((a.total + 4095) / 4096);
      if (a.blocks_per_graph < 1) a.blocks_per_graph = 1;
      if (a.blocks_per_graph > 8) a.blocks_per_graph = 8;
    };
    if (xc) {
      x = *xc;
      blocks_of(x);
    }
    if (xc && xc2) {
      x2 = *xc2;
      blocks_of(x2);
    }
    const size_t lds = (static_cast<size_t>(K) * K + K + 1024 + 16) * sizeof(float);
    const unsigned grid = static_cast<unsigned>(B + (xc ? B * x.blocks_per_graph : 0) + (xc && xc2 ? B * x2.blocks_per_graph : 0));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(post_lds_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(post_lds_kernel, dim3(grid), dim3(1024), lds, stream, p, static_cast<int>(B), x, x2);
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

}  // namespace tgp
