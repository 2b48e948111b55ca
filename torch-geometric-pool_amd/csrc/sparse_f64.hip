// r4: float64 value types of the HBM-bound operators.
//
// The reference's ATen calls compute `model.double()` inputs in fp64 (reduce/base_reduce.py:141-155,
// utils/ops.py:282-419); r1-r3 narrowed them to fp32 with a warning.  These operators move bytes, they do not need the
// matrix cores, so the value type costs nothing but bandwidth: here are their fp64 forms (the sparse Reduce, the sparse
// and dense post-processing); the edge-list kernels of sparse_connect.hip (subgraph, coalesce, block-diagonal export) are
// templates over the weight type.  Same summation orders as the fp32 kernels: members of a supernode in ascending
// assignment order, the edges of a row in list order, products rounded before the add -- so the sparse results are the
// bits of the reference's sequential CPU scatter in fp64.  The dense GEMM path (S^T X, S^T A S) stays fp32.
#include "primitives.h"

namespace tgp {

// ------------------------------------------------------------------------------------------------ A1, fp64
// one group of G lanes per pooled row, double2 per lane when the rows allow it
template <int G, bool VEC>
__global__ __launch_bounds__(256) void reduce_sparse_f64_kernel(const double* __restrict__ x, int64_t F, int64_t x_stride,
                                                                const int64_t* __restrict__ node_index,
                                                                const double* __restrict__ weight,
                                                                const int32_t* __restrict__ row_ptr,
                                                                const int32_t* __restrict__ perm, int64_t K,
                                                                double* __restrict__ x_pool) {
  constexpr int GROUPS = 256 / G;
  const int g = threadIdx.x % G;
  const int64_t group = static_cast<int64_t>(blockIdx.x) * GROUPS + threadIdx.x / G;
  const int64_t ngroups = static_cast<int64_t>(gridDim.x) * GROUPS;
  constexpr int W = VEC ? 2 : 1;
  for (int64_t c = group; c < K; c += ngroups) {
    const int32_t p0 = row_ptr ? row_ptr[c] : static_cast<int32_t>(c);
    const int32_t p1 = row_ptr ? row_ptr[c + 1] : p0 + 1;
    for (int64_t f = static_cast<int64_t>(W) * g; f < F; f += static_cast<int64_t>(W) * G) {
      double a0 = 0.0, a1 = 0.0;
      for (int32_t p = p0; p < p1; ++p) {
        const int32_t a = perm ? perm[p] : p;
        const double w = weight ? weight[a] : 1.0;
        const double* src = x + node_index[a] * x_stride + f;
        if constexpr (VEC) {
          const double2 v = *reinterpret_cast<const double2*>(src);
          a0 = __dadd_rn(a0, __dmul_rn(v.x, w));
          a1 = __dadd_rn(a1, __dmul_rn(v.y, w));
        } else {
          a0 = __dadd_rn(a0, __dmul_rn(*src, w));
        }
      }
      double* dst = x_pool + c * F + f;
      if constexpr (VEC) *reinterpret_cast<double2*>(dst) = make_double2(a0, a1);
      else *dst = a0;
    }
  }
}

// ------------------------------------------------------------------------------------------------ A6 norms, fp64
__global__ __launch_bounds__(256) void f64_rows_check_kernel(const int64_t* __restrict__ row, int64_t E,
                                                             int* __restrict__ unsorted) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (e + 1 < E && row[e] > row[e + 1]) *unsorted = 1;
}

// sorted rows: the FIRST edge of every run sums its run in list order (the order of the CPU scatter_add_); a hub row is
// one thread's loop -- acceptable for the fp64 route (model.double() is a debugging precision, not the fast path)
__global__ __launch_bounds__(256) void f64_degree_sorted_kernel(const int64_t* __restrict__ row,
                                                                const double* __restrict__ w, int64_t E,
                                                                const int* __restrict__ unsorted,
                                                                double* __restrict__ deg) {
  if (*unsorted) return;
  const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (e >= E) return;
  const int64_t r = row[e];
  if (e > 0 && row[e - 1] == r) return;
  double acc = 0.0;
  for (int64_t j = e; j < E && row[j] == r; ++j) acc = __dadd_rn(acc, w[j]);
  deg[r] = acc;
}

__global__ __launch_bounds__(256) void f64_degree_atomic_kernel(const int64_t* __restrict__ row,
                                                                const double* __restrict__ w, int64_t E,
                                                                const int* __restrict__ unsorted,
                                                                double* __restrict__ deg) {
  if (!*unsorted) return;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; e < E;
       e += static_cast<int64_t>(gridDim.x) * 256)
    atomicAdd(&deg[row[e]], w[e]);
}

__global__ __launch_bounds__(256) void f64_degree_scale_kernel(const int64_t* __restrict__ row,
                                                               const int64_t* __restrict__ col, double* __restrict__ w,
                                                               int64_t E, const double* __restrict__ deg, double eps) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (e < E) {  // deg.clamp(min=eps).pow(-0.5); w * dis[row] * dis[col]  (ops.py:395-401)
    const double dr = 1.0 / sqrt(fmax(deg[row[e]], eps)), dc = 1.0 / sqrt(fmax(deg[col[e]], eps));
    w[e] = __dmul_rn(__dmul_rn(w[e], dr), dc);
  }
}

// |w| >= 0: its IEEE bit pattern orders like an unsigned integer (exact, order-free max)
__global__ __launch_bounds__(256) void f64_graph_max_kernel(const int64_t* __restrict__ row,
                                                            const double* __restrict__ w, int64_t E,
                                                            const int64_t* __restrict__ batch_pooled,
                                                            unsigned long long* __restrict__ gmax) {
  int64_t g_cur = -1;
  unsigned long long m_cur = 0;
  const int64_t per = 16;
  const int64_t base = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) * per;
  for (int64_t e = base; e < base + per && e < E; ++e) {
    const int64_t g = batch_pooled[row[e]];
    const unsigned long long v = static_cast<unsigned long long>(__double_as_longlong(fabs(w[e])));
    if (g != g_cur) {
      if (g_cur >= 0 && m_cur > gmax[g_cur]) atomicMax(&gmax[g_cur], m_cur);
      g_cur = g;
      m_cur = v;
    } else {
      m_cur = m_cur > v ? m_cur : v;
    }
  }
  if (g_cur >= 0 && m_cur > gmax[g_cur]) atomicMax(&gmax[g_cur], m_cur);
}

__global__ __launch_bounds__(256) void f64_graph_max_scale_kernel(const int64_t* __restrict__ row,
                                                                  double* __restrict__ w, int64_t E,
                                                                  const int64_t* __restrict__ batch_pooled,
                                                                  const unsigned long long* __restrict__ gmax) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (e < E) {
    double m = __longlong_as_double(static_cast<long long>(gmax[batch_pooled[row[e]]]));
    if (m == 0.0) m = 1.0;
    w[e] = w[e] / m;
  }
}

// ------------------------------------------------------------------------------------------------ A8, fp64
// one workgroup per graph: zero diagonal, degree vector (fixed-order sums), D^-1/2 A D^-1/2, max-abs normalisation
// (utils/ops.py:282-335, in that order)
__global__ __launch_bounds__(256) void f64_post_dense_kernel(const double* __restrict__ src, double* __restrict__ dst,
                                                             int64_t K, int flags, double eps,
                                                             double* __restrict__ deg_ws,
                                                             unsigned long long* __restrict__ max_ws) {
  __shared__ unsigned long long s_max;
  const int64_t b = blockIdx.x;
  const double* a = src + b * K * K;
  double* o = dst + b * K * K;
  double* d = deg_ws + b * K;
  const bool rsl = (flags & TGP_REMOVE_SELF_LOOPS) != 0;
  // (the load is unconditional and the diagonal is masked afterwards: a conditional load is waited for before the next
  //  one is issued -- r5: this kernel took 55 us at B = 32, K = 128 with one exposed round trip per element)
  auto at = [&](int64_t r, int64_t c) -> double {
    const double v = a[r * K + c];
    return (rsl && r == c) ? 0.0 : v;
  };
  if (flags & TGP_DEGREE_NORM) {
    // sum over axis -2 (rows) when adj_transpose, else over axis -1 (ops.py:312-314), in index order.  Column sums: a
    // thread per column (coalesced across the threads).  Row sums: a WAVE per row, lanes stride over the columns and
    // keep eight loads in flight, partial sums folded by a fixed shuffle tree.
    if (flags & TGP_SUM_AXIS_ROWS) {
      for (int64_t i = threadIdx.x; i < K; i += 256) {
        double acc = 0.0;
#pragma unroll 8
        for (int64_t r = 0; r < K; ++r) acc += at(r, i);
        d[i] = sqrt(fmax(acc, eps));
      }
    } else {
      const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
      for (int64_t i = wv; i < K; i += 4) {
        double acc = 0.0;
#pragma unroll 4
        for (int64_t c = lane; c < K; c += 64) acc += at(i, c);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if (lane == 0) d[i] = sqrt(fmax(acc, eps));
      }
    }
  }
  if (threadIdx.x == 0) s_max = 0ull;
  __syncthreads();
  unsigned long long m = 0;
  {  // a wave per row, lanes over the columns: no 64-bit division per element
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int64_t r = wv; r < K; r += 4) {
      const double dr = (flags & TGP_DEGREE_NORM) ? d[r] : 1.0;
      for (int64_t c = lane; c < K; c += 64) {
        double v = at(r, c);
        if (flags & TGP_DEGREE_NORM) {
          // (adj / d) / d^T with d broadcast along the summed axis: d[c] then d[r] for axis -2, d[r] then d[c] for axis -1
          if (flags & TGP_SUM_AXIS_ROWS) v = (v / d[c]) / dr;
          else v = (v / dr) / d[c];
        }
        o[r * K + c] = v;
        const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(fabs(v)));
        m = m > bits ? m : bits;
      }
    }
  }
  if (flags & TGP_EDGE_WEIGHT_NORM) {
    atomicMax(&s_max, m);
    __syncthreads();
    double mx = __longlong_as_double(static_cast<long long>(s_max));
    if (mx == 0.0) mx = 1.0;
    for (int64_t i = threadIdx.x; i < K * K; i += 256) o[i] = o[i] / mx;
  }
  (void)max_ws;
}

}  // namespace tgp

using namespace tgp;

extern "C" int tgp_reduce_sparse_f64(const double* x, int64_t num_nodes, int64_t F, int64_t x_stride,
                                     const int64_t* node_index, const double* weight, const int32_t* row_ptr,
                                     const int32_t* perm, int64_t nnz, int64_t K, double* x_pool, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(num_nodes >= 0 && F >= 0 && K >= 0 && nnz >= 0 && (row_ptr || nnz == K), TGP_ERR_INVALID,
              "tgp_reduce_sparse_f64: bad argument");
  if (K == 0 || F == 0) return TGP_OK;
  TGP_REQUIRE(x_pool && (nnz == 0 || (x && node_index)), TGP_ERR_INVALID, "tgp_reduce_sparse_f64: null pointer");
  const bool vec = (F % 2 == 0) && (x_stride % 2 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0) &&
                   (reinterpret_cast<uintptr_t>(x_pool) % 16 == 0);
  const int64_t lanes = vec ? F / 2 : F;
  int G = 1;
  while (G < lanes && G < 64) G <<= 1;
  int64_t blocks = (K + 256 / G - 1) / (256 / G);
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  const dim3 grid(static_cast<unsigned>(blocks)), block(256);
#define TGP_F64_R(GG)                                                                                               \
  do {                                                                                                               \
    if (vec)                                                                                                         \
      hipLaunchKernelGGL((reduce_sparse_f64_kernel<GG, true>), grid, block, 0, stream, x, F, x_stride, node_index,   \
                         weight, row_ptr, perm, K, x_pool);                                                          \
    else                                                                                                             \
      hipLaunchKernelGGL((reduce_sparse_f64_kernel<GG, false>), grid, block, 0, stream, x, F, x_stride, node_index,  \
                         weight, row_ptr, perm, K, x_pool);                                                          \
  } while (0)
  switch (G) {
    case 1: TGP_F64_R(1); break;
    case 2: TGP_F64_R(2); break;
    case 4: TGP_F64_R(4); break;
    case 8: TGP_F64_R(8); break;
    case 16: TGP_F64_R(16); break;
    case 32: TGP_F64_R(32); break;
    default: TGP_F64_R(64); break;
  }
#undef TGP_F64_R
  return check_launch("tgp_reduce_sparse_f64");
}

extern "C" size_t tgp_postprocess_sparse_workspace_bytes_f64(int64_t /*E*/, int64_t num_nodes, int64_t num_graphs) {
  return align_up(static_cast<size_t>(num_nodes > 0 ? num_nodes : 1) * sizeof(double)) +
         align_up(static_cast<size_t>(num_graphs > 0 ? num_graphs : 1) * sizeof(unsigned long long)) + align_up(16) + 256;
}

extern "C" int tgp_postprocess_sparse_norm_f64(const int64_t* row, const int64_t* col, double* w, int64_t E,
                                               int64_t num_nodes, int flags, double eps, const int64_t* batch_pooled,
                                               int64_t num_graphs, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(E >= 0 && num_nodes >= 0 && num_graphs >= 0, TGP_ERR_INVALID, "tgp_postprocess_sparse_norm_f64: bad size");
  if (E == 0 || !(flags & (TGP_DEGREE_NORM | TGP_EDGE_WEIGHT_NORM))) return TGP_OK;
  TGP_REQUIRE(row && col && w, TGP_ERR_INVALID, "tgp_postprocess_sparse_norm_f64: null pointer");
  TGP_REQUIRE(ws && ws_bytes >= tgp_postprocess_sparse_workspace_bytes_f64(E, num_nodes, num_graphs), TGP_ERR_WORKSPACE,
              "tgp_postprocess_sparse_norm_f64: workspace too small");
  Carver cv(ws);
  double* deg = cv.take<double>(static_cast<size_t>(num_nodes > 0 ? num_nodes : 1));
  unsigned long long* gmax = cv.take<unsigned long long>(static_cast<size_t>(num_graphs > 0 ? num_graphs : 1));
  int* unsorted = cv.take<int>(4);
  const int nb = cdiv(E, 256);
  if (flags & TGP_DEGREE_NORM) {
    (void)hipMemsetAsync(deg, 0, static_cast<size_t>(num_nodes) * sizeof(double), stream);
    (void)hipMemsetAsync(unsorted, 0, sizeof(int), stream);
    hipLaunchKernelGGL(f64_rows_check_kernel, dim3(nb), dim3(256), 0, stream, row, E, unsorted);
    hipLaunchKernelGGL(f64_degree_sorted_kernel, dim3(nb), dim3(256), 0, stream, row, w, E, unsorted, deg);
    hipLaunchKernelGGL(f64_degree_atomic_kernel, dim3(nb < 4096 ? nb : 4096), dim3(256), 0, stream, row, w, E, unsorted,
                       deg);
    hipLaunchKernelGGL(f64_degree_scale_kernel, dim3(nb), dim3(256), 0, stream, row, col, w, E, deg, eps);
  }
  if (flags & TGP_EDGE_WEIGHT_NORM) {
    TGP_REQUIRE(batch_pooled && num_graphs > 0, TGP_ERR_INVALID,
                "tgp_postprocess_sparse_norm_f64: edge_weight_norm needs batch_pooled");
    (void)hipMemsetAsync(gmax, 0, static_cast<size_t>(num_graphs) * sizeof(unsigned long long), stream);
    hipLaunchKernelGGL(f64_graph_max_kernel, dim3(cdiv(cdiv(E, 16), 256)), dim3(256), 0, stream, row, w, E, batch_pooled,
                       gmax);
    hipLaunchKernelGGL(f64_graph_max_scale_kernel, dim3(nb), dim3(256), 0, stream, row, w, E, batch_pooled, gmax);
  }
  return check_launch("tgp_postprocess_sparse_norm_f64");
}

extern "C" size_t tgp_postprocess_dense_workspace_bytes_f64(int64_t B, int64_t K) {
  return align_up(static_cast<size_t>(B > 0 ? B : 1) * static_cast<size_t>(K > 0 ? K : 1) * sizeof(double)) + 256;
}

extern "C" int tgp_postprocess_dense_f64(const double* src, double* dst, int64_t B, int64_t K, int flags, double eps,
                                         void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(B >= 0 && K >= 0, TGP_ERR_INVALID, "tgp_postprocess_dense_f64: bad size");
  if (B == 0 || K == 0) return TGP_OK;
  TGP_REQUIRE(src && dst && ws && ws_bytes >= tgp_postprocess_dense_workspace_bytes_f64(B, K), TGP_ERR_WORKSPACE,
              "tgp_postprocess_dense_f64: null pointer or workspace too small");
  TGP_REQUIRE(B < (1ll << 31), TGP_ERR_RANGE, "tgp_postprocess_dense_f64: B >= 2^31");
  Carver cv(ws);
  double* deg = cv.take<double>(static_cast<size_t>(B * K));
  hipLaunchKernelGGL(f64_post_dense_kernel, dim3(static_cast<unsigned>(B)), dim3(256), 0, stream, src, dst, K, flags,
                     eps, deg, static_cast<unsigned long long*>(nullptr));
  return check_launch("tgp_postprocess_dense_f64");
}
