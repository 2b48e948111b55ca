// A14: NDPSelect's spectral partition (select/ndp_select.py:187-256) on the device, one workgroup per graph.
//
// The reference loops over the graphs of a batch on the host: symmetric normalised Laplacian Ls = I - D^-1/2 A D^-1/2,
// its largest eigenvector by torch.lobpcg, sign partition z, cut size z^T L z / (2 vol), random +-1 partition when the
// cut is below 0.5 (or the eigen-solve fails).  The partition is only defined up to the eigenvector's sign and solver
// tolerance, so the contract is: z is the sign pattern of a (converged) largest eigenvector of Ls, or the random
// fallback under the reference's rule.  Here every graph runs the method the reference names -- LOBPCG with one vector:
// every step maximises the Rayleigh quotient over span{x, residual, previous direction} (a 3 x 3 symmetric eigenproblem
// solved redundantly by every thread) -- in fp64 with the six work vectors and the graph's matrix in LDS; one sparse
// mat-vec and four workgroup reductions per step, a few dozen steps per graph (a plain power iteration needed a
// median of 400 and up to thousands when the two largest eigenvalues are close).
#include "primitives.h"

namespace tgp {

constexpr int NDP_MAX_N = 2048;

__device__ __forceinline__ uint32_t ndp_hash(uint64_t seed, uint64_t v) {
  uint64_t x = seed ^ (v * 0x9E3779B97F4A7C15ull);
  x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull;
  x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ull;
  x ^= x >> 33;
  return static_cast<uint32_t>(x);
}

// Sum over the 64 lanes of a wave, the same value in every lane: four rotate-and-add steps inside the 16-lane rows
// (DPP row_ror: register moves, no trip through the LDS pipe that a shuffle takes), then the four row sums are read
// through scalar registers.  The iteration below does a dozen of these per step, back to back.
template <int CTRL>
__device__ __forceinline__ double ndp_dpp_f64(double v) {
  const unsigned long long u = __double_as_longlong(v);
  const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(u), CTRL, 0xF, 0xF, false));
  const unsigned hi =
      static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(u >> 32), CTRL, 0xF, 0xF, false));
  return __longlong_as_double((static_cast<unsigned long long>(hi) << 32) | lo);
}
__device__ __forceinline__ double ndp_readlane_f64(double v, int lane) {
  const unsigned long long u = __double_as_longlong(v);
  const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(u), lane));
  const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(u >> 32), lane));
  return __longlong_as_double((static_cast<unsigned long long>(hi) << 32) | lo);
}
__device__ __forceinline__ double ndp_wave_sum(double v) {
  v += ndp_dpp_f64<0x121>(v);  // row_ror:1
  v += ndp_dpp_f64<0x122>(v);  // row_ror:2
  v += ndp_dpp_f64<0x124>(v);  // row_ror:4
  v += ndp_dpp_f64<0x128>(v);  // row_ror:8 -> every lane holds its row's sum
  return (ndp_readlane_f64(v, 0) + ndp_readlane_f64(v, 16)) + (ndp_readlane_f64(v, 32) + ndp_readlane_f64(v, 48));
}

template <int THREADS>
__device__ __forceinline__ double ndp_block_sum(double v, double* s_red) {
  v = ndp_wave_sum(v);
  if constexpr (THREADS == 64) return v;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < THREADS / 64; ++w) t += s_red[w];
  return t;
}
// K sums at once: the K wave reductions are independent instruction chains (they overlap), and a 256-thread
// workgroup pays its two barriers once for all of them.  s_red: K * (THREADS / 64) doubles.
template <int THREADS, int K>
__device__ __forceinline__ void ndp_block_sums(double (&v)[K], double* s_red) {
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = ndp_wave_sum(v[k]);
  if constexpr (THREADS == 64) return;
  constexpr int NW = THREADS / 64;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) s_red[k * NW + (threadIdx.x >> 6)] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += s_red[k * NW + w];
    v[k] = t;
  }
}

template <int THREADS>
__device__ __forceinline__ double ndp_block_max(double v, double* s_red) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = fmax(v, __shfl_xor(v, d, WAVE));
  if constexpr (THREADS == 64) return v;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < THREADS / 64; ++w) t = fmax(t, s_red[w]);
  return t;
}

// largest eigenpair of a symmetric 3 x 3 matrix (cyclic Jacobi, fixed sweeps; every thread runs it on the same numbers)
__device__ __forceinline__ void ndp_eig3_largest(double a[3][3], int dim, double& theta, double c[3]) {
  double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 8; ++sweep) {
    // every lane holds the same numbers, so the exits below are wave-uniform; near convergence the matrix is
    // already almost diagonal (a[0][1] is the residual norm) and one or two sweeps finish it
    const double off = fabs(a[0][1]) + (dim == 3 ? fabs(a[0][2]) + fabs(a[1][2]) : 0.0);
    const double dia = fabs(a[0][0]) + fabs(a[1][1]) + (dim == 3 ? fabs(a[2][2]) : 0.0);
    if (!(off > 1e-17 * dia)) break;
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
      const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
      if (q >= dim) continue;
      const double apq = a[p][q];
      if (!(fabs(apq) > 1e-18 * (fabs(a[p][p]) + fabs(a[q][q])))) continue;
      const double tau = (a[q][q] - a[p][p]) / (2.0 * apq);
      const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
      const double cs = 1.0 / sqrt(1.0 + t * t), sn = t * cs;
#pragma unroll
      for (int k = 0; k < 3; ++k) {  // A <- A J
        const double akp = a[k][p], akq = a[k][q];
        a[k][p] = cs * akp - sn * akq;
        a[k][q] = sn * akp + cs * akq;
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {  // A <- J^T A
        const double apk = a[p][k], aqk = a[q][k];
        a[p][k] = cs * apk - sn * aqk;
        a[q][k] = sn * apk + cs * aqk;
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double vkp = v[k][p], vkq = v[k][q];
        v[k][p] = cs * vkp - sn * vkq;
        v[k][q] = sn * vkp + cs * vkq;
      }
    }
  }
  int best = 0;
  for (int k = 1; k < dim; ++k)
    if (a[k][k] > a[best][best]) best = k;
  theta = a[best][best];
  for (int k = 0; k < 3; ++k) c[k] = k < dim ? v[k][best] : 0.0;
}

// The same in fp32, for the one-workgroup kernel below.  The 3 x 3 Rayleigh-Ritz only picks the next search direction:
// the iterate is renormalised and its Rayleigh quotient and residual are re-evaluated in fp64 every step, so the
// coefficients need a few digits, not sixteen.  The fp64 sweeps (two divisions and two square roots per rotation, each a
// sequence of ~30 dependent instructions) were 55 % of a step: 7.1 k of 13 k cycles per step on a 49-node graph (r3
// stamps), 1.03 ms for the 2048 graphs of a PROTEINS-shaped batch.
__device__ __forceinline__ void ndp_eig3_largest_f32(const double ad[3][3], int dim, double c[3]) {
  float a[3][3], v[3][3] = {{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) a[i][j] = static_cast<float>(ad[i][j]);
  for (int sweep = 0; sweep < 4; ++sweep) {
    const float off = fabsf(a[0][1]) + (dim == 3 ? fabsf(a[0][2]) + fabsf(a[1][2]) : 0.f);
    const float dia = fabsf(a[0][0]) + fabsf(a[1][1]) + (dim == 3 ? fabsf(a[2][2]) : 0.f);
    if (!(off > 1e-6f * dia)) break;  // (wave-uniform: every lane holds the same numbers; fp32 cannot go much lower)
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
      const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
      if (q >= dim) continue;
      const float apq = a[p][q];
      if (!(fabsf(apq) > 1e-9f * (fabsf(a[p][p]) + fabsf(a[q][q])))) continue;
      const float tau = (a[q][q] - a[p][p]) / (2.0f * apq);
      const float t = (tau >= 0.f ? 1.0f : -1.0f) / (fabsf(tau) + sqrtf(1.0f + tau * tau));
      const float cs = 1.0f / sqrtf(1.0f + t * t), sn = t * cs;
#pragma unroll
      for (int k = 0; k < 3; ++k) {  // A <- A J
        const float akp = a[k][p], akq = a[k][q];
        a[k][p] = cs * akp - sn * akq;
        a[k][q] = sn * akp + cs * akq;
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {  // A <- J^T A
        const float apk = a[p][k], aqk = a[q][k];
        a[p][k] = cs * apk - sn * aqk;
        a[q][k] = sn * apk + cs * aqk;
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float vkp = v[k][p], vkq = v[k][q];
        v[k][p] = cs * vkp - sn * vkq;
        v[k][q] = sn * vkp + cs * vkq;
      }
    }
  }
  int best = 0;
  for (int k = 1; k < dim; ++k)
    if (a[k][k] > a[best][best]) best = k;
  for (int k = 0; k < 3; ++k) c[k] = k < dim ? static_cast<double>(v[k][best]) : 0.0;
}

// Largest eigenpair of the 3 x 3 (or 2 x 2) Rayleigh-Ritz matrix in closed form, fp32: the eigenvalue from the
// trigonometric solution of the characteristic cubic, the eigenvector as the largest of the three cross products of rows
// of H - theta I.  ~100 instructions where four Jacobi sweeps are ~1000 (5.6 k cycles per step of the one-wave kernel,
// more than half of it).  When the largest eigenvalue is (nearly) double the cross products vanish: the Jacobi sweeps
// take over (rare: two coinciding Ritz values).  c comes back with unit length.  Square roots and reciprocals are the
// 1-ulp hardware ones (r6): the coefficients only pick the next search direction.
__device__ __forceinline__ void ndp_rr_largest_f32(const double ad[3][3], int dim, double c[3]) {
  const float a00 = static_cast<float>(ad[0][0]), a11 = static_cast<float>(ad[1][1]), a01 = static_cast<float>(ad[0][1]);
  if (dim == 2) {
    const float t = 0.5f * (a00 - a11), sr = __builtin_amdgcn_sqrtf(t * t + a01 * a01);
    const float theta = 0.5f * (a00 + a11) + sr;
    // rows of H - theta I: (a00 - theta, a01) and (a01, a11 - theta); eigenvector orthogonal to the larger one
    const float u0 = theta - a11, u1 = a01;      // from the second row
    const float v0 = a01, v1 = theta - a00;      // from the first row
    const float nu = u0 * u0 + u1 * u1, nv = v0 * v0 + v1 * v1;
    const float x0 = nu >= nv ? u0 : v0, x1 = nu >= nv ? u1 : v1, nn = nu >= nv ? nu : nv;
    if (nn > 0.f) {
      const float inv = __builtin_amdgcn_rsqf(nn);
      c[0] = static_cast<double>(x0 * inv); c[1] = static_cast<double>(x1 * inv); c[2] = 0.0;
    } else {  // a multiple of the identity
      c[0] = 1.0; c[1] = 0.0; c[2] = 0.0;
    }
    return;
  }
  const float a22 = static_cast<float>(ad[2][2]), a02 = static_cast<float>(ad[0][2]), a12 = static_cast<float>(ad[1][2]);
  const float p1 = a01 * a01 + a02 * a02 + a12 * a12;
  const float q = (a00 + a11 + a22) * (1.0f / 3.0f);
  const float b00 = a00 - q, b11 = a11 - q, b22 = a22 - q;
  const float p2 = b00 * b00 + b11 * b11 + b22 * b22 + 2.0f * p1;
  bool ok = p2 > 0.f;
  if (ok) {
    const float pp = __builtin_amdgcn_sqrtf(p2 * (1.0f / 6.0f)), ip = __builtin_amdgcn_rcpf(pp);
    const float c00 = b00 * ip, c11 = b11 * ip, c22 = b22 * ip, c01 = a01 * ip, c02 = a02 * ip, c12 = a12 * ip;
    float r = 0.5f * (c00 * (c11 * c22 - c12 * c12) - c01 * (c01 * c22 - c12 * c02) + c02 * (c01 * c12 - c11 * c02));
    r = fminf(1.0f, fmaxf(-1.0f, r));
    const float theta = q + 2.0f * pp * cosf(acosf(r) * (1.0f / 3.0f));
    const float m00 = a00 - theta, m11 = a11 - theta, m22 = a22 - theta;
    // cross products of the rows (m00, a01, a02), (a01, m11, a12), (a02, a12, m22)
    const float x12[3] = {m11 * m22 - a12 * a12, a12 * a02 - a01 * m22, a01 * a12 - m11 * a02};
    const float x02[3] = {a01 * m22 - a02 * a12, a02 * a02 - m00 * m22, m00 * a12 - a01 * a02};
    const float x01[3] = {a01 * a12 - a02 * m11, a02 * a01 - m00 * a12, m00 * m11 - a01 * a01};
    const float n12 = x12[0] * x12[0] + x12[1] * x12[1] + x12[2] * x12[2];
    const float n02 = x02[0] * x02[0] + x02[1] * x02[1] + x02[2] * x02[2];
    const float n01 = x01[0] * x01[0] + x01[1] * x01[1] + x01[2] * x01[2];
    float best = n12, y0 = x12[0], y1 = x12[1], y2 = x12[2];
    if (n02 > best) { best = n02; y0 = x02[0]; y1 = x02[1]; y2 = x02[2]; }
    if (n01 > best) { best = n01; y0 = x01[0]; y1 = x01[1]; y2 = x01[2]; }
    ok = best > 1e-6f * p2 * p2;  // (gap to the other eigenvalues) x (spread) well above fp32 noise
    if (ok) {
      const float inv = __builtin_amdgcn_rsqf(best);
      c[0] = static_cast<double>(y0 * inv); c[1] = static_cast<double>(y1 * inv); c[2] = static_cast<double>(y2 * inv);
    }
  }
  if (!ok) ndp_eig3_largest_f32(ad, dim, c);  // (wave-uniform)
}

// 1 / sqrt(v) for the per-graph kernels (r6): v_rsq_f64 (~26 bits) and two Newton steps -- a dozen dependent instructions
// where `1.0 / sqrt(v)` is a correctly rounded square root AND a division (~40); four of them sit in the chain of every
// step.  Full double precision to a few ulp: the iteration re-evaluates |x|, the Rayleigh quotient and the residual from
// the vectors every step, so nothing accumulates.
__device__ __forceinline__ double ndp_rcp(double v) {  // 1 / v: v_rcp_f64 and two Newton steps (a few ulp)
  double y = __builtin_amdgcn_rcp(v);
  y = fma(y, fma(-v, y, 1.0), y);
  y = fma(y, fma(-v, y, 1.0), y);
  return y;
}
__device__ __forceinline__ double ndp_rsqrt(double v) {
  double y = __builtin_amdgcn_rsq(v);
  const double h = 0.5 * v;
  y = fma(y, fma(-h * y, y, 0.5), y);
  y = fma(y, fma(-h * y, y, 0.5), y);
  return y;
}

// indptr / col / w: CSR over all nodes of the batch of a SYMMETRIC adjacency without self loops (the caller
// symmetrises with max, as to_undirected(reduce="max") does, ndp_select.py:198-202).
template <int THREADS>
__global__ __launch_bounds__(THREADS) void ndp_partition_kernel(const int32_t* __restrict__ indptr,
                                                                const int64_t* __restrict__ col,
                                                                const float* __restrict__ w,
                                                                const int64_t* __restrict__ graph_ptr,
                                                                unsigned long long seed, int max_iter, double tol,
                                                                int ncap, int ecap,
                                                                uint8_t* __restrict__ keep, int32_t* __restrict__ info,
                                                                int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  __shared__ double s_red[6 * (THREADS / 64) + 1];
  const int g = blockIdx.x, tid = threadIdx.x;
  const int64_t p0 = graph_ptr[g], p1 = graph_ptr[g + 1];
  const int64_t n64 = p1 - p0;
  if (n64 <= 0) return;
  if (n64 > NDP_MAX_N || n64 > ncap) {
    // beyond the kernel's limit, or beyond the max_graph_nodes the caller declared (which sized the LDS arrays): left
    // unpartitioned (keep stays 0, info = -2): the caller handles those graphs itself (tgp_ndp_large_*)
    if (tid == 0) info[g] = -2;
    return;
  }
  const int n = static_cast<int>(n64);
  if (n == 1) {  // trivial case (ndp_select.py:223-225): the node is kept
    if (tid == 0) { keep[p0] = 1; info[g] = 0; }
    return;
  }
  // LDS: six fp64 work vectors [ncap], then the graph's matrix M = D^-1/2 A D^-1/2 as a local CSR when it fits
  // (values [ecap] fp64, row offsets [ncap + 1], columns [ecap] uint16), dis [ncap] fp32: the iteration then never
  // leaves the CU (reading indptr / col / w from L2 every iteration cost ~2 us per iteration)
  double* x = s_dyn;              // iterate (unit length)
  double* ax = s_dyn + ncap;      // Ls x
  double* wv = s_dyn + 2 * ncap;  // normalised residual
  double* aw = s_dyn + 3 * ncap;
  double* pv = s_dyn + 4 * ncap;  // previous search direction, orthonormalised against x and wv
  double* ap = s_dyn + 5 * ncap;
  double* mval = s_dyn + 6 * ncap;
  int* mptr = reinterpret_cast<int*>(mval + ecap);
  float* dis = reinterpret_cast<float*>(mptr + ncap + 1);
  unsigned short* mcol = reinterpret_cast<unsigned short*>(dis + ncap);
  const int e_lo = indptr[p0], nnz_g = indptr[p1] - e_lo;
  const bool cached = nnz_g <= ecap && n <= 65535;
  // degrees, volume
  double vol_part = 0.0;
  for (int i = tid; i < n; i += THREADS) {
    double d = 0.0;
    for (int e = indptr[p0 + i]; e < indptr[p0 + i + 1]; ++e) {
      const int64_t c = col[e];
      if (c < p0 || c >= p1) { atomicOr(status, 2); continue; }  // an edge between two graphs of the batch
      d += w ? static_cast<double>(w[e]) : 1.0;
    }
    dis[i] = d > 0.0 ? static_cast<float>(1.0 / sqrt(d)) : 0.f;
    vol_part += d;
    x[i] = (static_cast<double>(ndp_hash(0x5EEDull, static_cast<uint64_t>(i) + 977ull * n) >> 8) / 8388608.0) - 1.0;
  }
  const double vol = ndp_block_sum<THREADS>(vol_part, s_red);
  __syncthreads();
  if (cached) {
    for (int i = tid; i <= n; i += THREADS) mptr[i] = indptr[p0 + i] - e_lo;
    for (int i = tid; i < n; i += THREADS) {
      for (int e = indptr[p0 + i]; e < indptr[p0 + i + 1]; ++e) {
        const int64_t c = col[e];
        const bool in = c >= p0 && c < p1;
        const int j = in ? static_cast<int>(c - p0) : 0;
        mcol[e - e_lo] = static_cast<unsigned short>(j);
        mval[e - e_lo] = in ? (w ? static_cast<double>(w[e]) : 1.0) * static_cast<double>(dis[i]) *
                                  static_cast<double>(dis[j]) : 0.0;
      }
    }
    __syncthreads();
  }
  int it = 0;
  bool random_part = !(vol > 0.0);
  auto matvec = [&](const double* src, double* dst) {  // dst = Ls src on this thread's rows
    for (int i = tid; i < n; i += THREADS) {
      double v;
      if (cached) {
        double acc = 0.0;
        for (int e = mptr[i]; e < mptr[i + 1]; ++e) acc += mval[e] * src[mcol[e]];
        v = src[i] - acc;
      } else {
        double acc = 0.0;
        for (int e = indptr[p0 + i]; e < indptr[p0 + i + 1]; ++e) {
          const int64_t c = col[e];
          if (c < p0 || c >= p1) continue;
          const int j = static_cast<int>(c - p0);
          acc += (w ? static_cast<double>(w[e]) : 1.0) * static_cast<double>(dis[j]) * src[j];
        }
        v = src[i] - static_cast<double>(dis[i]) * acc;
      }
      dst[i] = v;
    }
  };
  if (!random_part) {
    // x <- x / |x|, ax = Ls x, lambda = x . ax
    double sq = 0.0;
    for (int i = tid; i < n; i += THREADS) sq += x[i] * x[i];
    const double x2 = ndp_block_sum<THREADS>(sq, s_red);
    __syncthreads();
    {
      const double ix0 = ndp_rsqrt(x2);
      for (int i = tid; i < n; i += THREADS) { x[i] *= ix0; pv[i] = 0.0; ap[i] = 0.0; }
    }
    __syncthreads();
    matvec(x, ax);
    double dt = 0.0;
    for (int i = tid; i < n; i += THREADS) dt += x[i] * ax[i];
    double lam = ndp_block_sum<THREADS>(dt, s_red);
    // One step = three rounds of reductions (the basis {x, w, p} is orthonormalised with coefficients that ride
    // along with sums needed anyway, and |x| = 1 is restored together with the new Rayleigh quotient):
    //   A: |r|^2, x.p, r.p            r = Ls x - lambda x, w = r / |r|, p' = p - (x.p) x - (w.p) w
    //   B: |p'|^2, x.Aw, w.Aw, x.Ap', w.Ap', p'.Ap'   -> 3 x 3 Rayleigh-Ritz on {x, w, p' / |p'|}
    //   C: |x_new|^2, x_new.A x_new
    bool has_p = false;
    for (; it < max_iter; ++it) {
      double sa[3] = {0.0, 0.0, 0.0};
      for (int i = tid; i < n; i += THREADS) {
        const double r = ax[i] - lam * x[i];
        wv[i] = r;
        sa[0] += r * r;
        sa[1] += x[i] * pv[i];
        sa[2] += r * pv[i];
      }
      ndp_block_sums<THREADS, 3>(sa, s_red);
      const double rn2 = sa[0];
      if (!(rn2 > tol * tol * lam * lam)) break;  // |Ls x - lambda x| <= tol * lambda: converged
      const double inv_r = ndp_rsqrt(rn2);
      const double cxp = sa[1], cwp = sa[2] * inv_r;
      __syncthreads();
      for (int i = tid; i < n; i += THREADS) wv[i] *= inv_r;
      __syncthreads();
      matvec(wv, aw);
      double sb[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      for (int i = tid; i < n; i += THREADS) {
        double pi = 0.0, api = 0.0;
        if (has_p) {
          pi = pv[i] - cxp * x[i] - cwp * wv[i];
          api = ap[i] - cxp * ax[i] - cwp * aw[i];
          pv[i] = pi;
          ap[i] = api;
        }
        sb[0] += pi * pi;
        sb[1] += x[i] * aw[i];
        sb[2] += wv[i] * aw[i];
        sb[3] += x[i] * api;
        sb[4] += wv[i] * api;
        sb[5] += pi * api;
      }
      ndp_block_sums<THREADS, 6>(sb, s_red);
      const int dim = (has_p && sb[0] > 1e-24) ? 3 : 2;  // p was unit length: what is left of it outside span{x, w}
      const double ip = dim == 3 ? ndp_rsqrt(sb[0]) : 0.0;
      double h[3][3] = {{lam, sb[1], sb[3] * ip}, {sb[1], sb[2], sb[4] * ip}, {sb[3] * ip, sb[4] * ip, sb[5] * ip * ip}};
      double c[3];
      ndp_rr_largest_f32(h, dim, c);
      if (c[0] < 0.0) { c[0] = -c[0]; c[1] = -c[1]; c[2] = -c[2]; }
      // x <- c0 x + c1 w + c2 p^, p <- (c1 w + c2 p^) / |.| (the same combinations of Ls x, Ls w, Ls p^)
      const double c2p = c[2] * ip;
      const double pn2 = c[1] * c[1] + c[2] * c[2];
      const double sp = pn2 > 1e-300 ? ndp_rsqrt(pn2) : 0.0;
      double sc[2] = {0.0, 0.0};
      for (int i = tid; i < n; i += THREADS) {
        const double pn = c[1] * wv[i] + c2p * pv[i], apn = c[1] * aw[i] + c2p * ap[i];
        const double xn = c[0] * x[i] + pn, axn = c[0] * ax[i] + apn;
        pv[i] = pn * sp;
        ap[i] = apn * sp;
        x[i] = xn;
        ax[i] = axn;
        sc[0] += xn * xn;
        sc[1] += xn * axn;
      }
      ndp_block_sums<THREADS, 2>(sc, s_red);
      const double ix = ndp_rsqrt(sc[0]);
      for (int i = tid; i < n; i += THREADS) {
        x[i] *= ix;
        ax[i] *= ix;
      }
      lam = sc[1] * ix * ix;
      has_p = sp > 0.0;
      __syncthreads();
    }
    if (!(lam > 0.0)) random_part = true;  // cannot happen for vol > 0 (Ls has trace n > 0)
  }
  // sign partition and its cut: z^T L z / (2 vol) = (weight of the directed entries that cross the cut) / vol
  if (!random_part) {
    double cross = 0.0;
    for (int i = tid; i < n; i += THREADS) {
      const bool zi = x[i] >= 0.0;
      for (int e = indptr[p0 + i]; e < indptr[p0 + i + 1]; ++e) {
        const int64_t c = col[e];
        if (c < p0 || c >= p1) continue;
        if ((x[c - p0] >= 0.0) != zi) cross += w ? static_cast<double>(w[e]) : 1.0;
      }
    }
    const double cut = ndp_block_sum<THREADS>(cross, s_red) / vol;
    if (cut < 0.5) random_part = true;  // ndp_select.py:250-252
  }
  for (int i = tid; i < n; i += THREADS) {
    bool pos;
    if (random_part) {  // sign_partition(n): node 0 positive, node 1 negative, the rest random (ndp_select.py:171-185)
      pos = i == 0 ? true : (i == 1 ? false : (ndp_hash(seed, static_cast<uint64_t>(p0 + i)) & 1u) != 0);
    } else {
      pos = x[i] >= 0.0;
    }
    keep[p0 + i] = pos ? 1 : 0;
  }
  if (tid == 0) info[g] = random_part ? -1 : it;
}


// The same iteration for graphs of at most 64 nodes, ONE WAVE per graph with the six work vectors in REGISTERS (lane =
// node): the loops over "this thread's rows" and their LDS round trips are gone, a step is ~40 vector instructions, one
// LDS exchange for the sparse mat-vec, eleven DPP reductions and the 3 x 3 Rayleigh-Ritz.  r3, 2048 graphs of 20..60
// nodes: ndp_partition_kernel<64> took 1.03 ms (13 k cycles per step, 7 k of them fp64 Jacobi sweeps); same arithmetic,
// same stopping rule, same outputs up to the solver tolerance.
__global__ __launch_bounds__(64) void ndp_partition_wave_kernel(const int32_t* __restrict__ indptr,
                                                                const int64_t* __restrict__ col,
                                                                const float* __restrict__ w,
                                                                const int64_t* __restrict__ graph_ptr,
                                                                unsigned long long seed, int max_iter, double tol,
                                                                int ncap, int ecap, uint8_t* __restrict__ keep,
                                                                int32_t* __restrict__ info, int* __restrict__ status,
                                                                int warm) {
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  const int g = blockIdx.x, lane = threadIdx.x;
  const int64_t p0 = graph_ptr[g], p1 = graph_ptr[g + 1];
  const int64_t n64 = p1 - p0;
  if (n64 <= 0) return;
  if (n64 > 64 || n64 > ncap) {
    if (lane == 0) info[g] = -2;
    return;
  }
  const int n = static_cast<int>(n64);
  if (n == 1) {
    if (lane == 0) { keep[p0] = 1; info[g] = 0; }
    return;
  }
  double* xs = s_dyn;  // [ncap] the vector a mat-vec reads through the column indices
  double* mval = s_dyn + 6 * ncap;
  int* mptr = reinterpret_cast<int*>(mval + ecap);
  float* dis = reinterpret_cast<float*>(mptr + ncap + 1);
  unsigned short* mcol = reinterpret_cast<unsigned short*>(dis + ncap);
  const bool on = lane < n;
  const int e_beg = on ? indptr[p0 + lane] : 0, e_end = on ? indptr[p0 + lane + 1] : 0;
  const int e_lo = indptr[p0], nnz_g = indptr[p1] - e_lo;
  const bool cached = nnz_g <= ecap;
  double deg = 0.0;
  for (int e = e_beg; e < e_end; ++e) {
    const int64_t c = col[e];
    if (c < p0 || c >= p1) { atomicOr(status, 2); continue; }
    deg += w ? static_cast<double>(w[e]) : 1.0;
  }
  const float my_dis = deg > 0.0 ? static_cast<float>(1.0 / sqrt(deg)) : 0.f;
  if (on) dis[lane] = my_dis;
  const double vol = ndp_wave_sum(deg);
  __builtin_amdgcn_wave_barrier();
  if (cached && on) {
    for (int e = e_beg; e < e_end; ++e) {
      const int64_t c = col[e];
      const bool in = c >= p0 && c < p1;
      const int j = in ? static_cast<int>(c - p0) : 0;
      mcol[e - e_lo] = static_cast<unsigned short>(j);
      mval[e - e_lo] = in ? (w ? static_cast<double>(w[e]) : 1.0) * static_cast<double>(my_dis) *
                                static_cast<double>(dis[j]) : 0.0;
    }
  }
  __builtin_amdgcn_wave_barrier();
  const int r_beg = e_beg - e_lo, r_end = e_end - e_lo;
  // the lane's first NDP_RC matrix entries stay in registers for the whole iteration (r5, late): they never change, and
  // read from LDS every step the mat-vec was a chain of two dependent LDS reads per entry (value + column, then the
  // vector element); now the vector elements of those entries are requested together.  Same products added in the same
  // order (missing entries add 0 x xs[lane]).
  constexpr int NDP_RC = 8;
  double rc_val[NDP_RC];
  int rc_col[NDP_RC];
#pragma unroll
  for (int q = 0; q < NDP_RC; ++q) {
    const int e = r_beg + q;
    const bool ok = cached && on && e < r_end;
    rc_val[q] = ok ? mval[ok ? e : 0] : 0.0;
    rc_col[q] = ok ? static_cast<int>(mcol[ok ? e : 0]) : lane;
  }
  auto matvec = [&](double src) -> double {  // (Ls src)[lane]
    xs[lane] = src;
    __builtin_amdgcn_wave_barrier();
    double acc = 0.0;
    if (cached) {
      double t[NDP_RC];
#pragma unroll
      for (int q = 0; q < NDP_RC; ++q) t[q] = xs[rc_col[q]];
#pragma unroll
      for (int q = 0; q < NDP_RC; ++q) acc += rc_val[q] * t[q];
      for (int e = r_beg + NDP_RC; e < r_end; ++e) acc += mval[e] * xs[mcol[e]];
    } else {
      for (int e = e_beg; e < e_end; ++e) {
        const int64_t c = col[e];
        if (c < p0 || c >= p1) continue;
        const int j = static_cast<int>(c - p0);
        acc += (w ? static_cast<double>(w[e]) : 1.0) * static_cast<double>(dis[j]) * xs[j];
      }
      acc *= static_cast<double>(my_dis);
    }
    __builtin_amdgcn_wave_barrier();
    return on ? src - acc : 0.0;
  };
  int it = 0;
  bool random_part = !(vol > 0.0);
  double x = on ? (static_cast<double>(ndp_hash(0x5EEDull, static_cast<uint64_t>(lane) + 977ull * n) >> 8) / 8388608.0) - 1.0
                : 0.0;
  if (!random_part) {
    const double x2 = ndp_wave_sum(x * x);
    x *= ndp_rsqrt(x2);
    if (warm) {
      // ---- r6: Lanczos warm start.  A LOBPCG step is ~5 k cycles of reductions and Rayleigh-Ritz around a 150-cycle
      // mat-vec; a Lanczos step is the mat-vec and two reductions.  n steps from the same start vector span (in exact
      // arithmetic) the whole Krylov space of a graph of n <= 64 nodes: T = tridiag(alpha, beta) is kept one entry per
      // lane, its largest eigenvalue comes from 64-section on the sign of the leading minors of sigma I - T (lane =
      // shift; no divisions), its eigenvector from two inverse iterations with sigma just above it (sigma I - T is
      // positive definite: LDL^T without pivoting), and x = V y from a second run of the recurrence (the vectors are not
      // stored).  The LOBPCG loop below then starts from x: it keeps the stopping rule, so what leaves the kernel
      // satisfies the same residual bound as before -- from a start that usually satisfies it already.
      const double x0 = x;
      double ta = 0.0, tb = 0.0, tib = 0.0, tb2 = 0.0;  // lane j: alpha_j, beta_j (= |u_j|, links j -> j + 1), 1 / beta_j, beta_j^2
      int m = 0;
      {
        double v = x0, vp = 0.0, bprev = 0.0;
        const int mmax = n < warm ? n : warm;
        for (int j = 0; j < mmax; ++j) {
          double u = matvec(v);
          const double a = ndp_wave_sum(u * v);
          u = (u - a * v) - bprev * vp;
          const double b2 = ndp_wave_sum(u * u);
          if (lane == j) ta = a;
          m = j + 1;
          if (!(b2 > 1e-26)) break;  // an invariant subspace: T is exact
          const double ib = ndp_rsqrt(b2), b = b2 * ib;
          if (lane == j) { tb = b; tib = ib; tb2 = b2; }
          vp = v;
          v = u * ib;
          bprev = b;
        }
      }
      if (m >= 2) {
        double lo = -1e-6, hi = 2.0 + 1e-6;  // the spectrum of Ls lies in [0, 2], the Ritz values inside it
        for (int pass = 0; pass < 6; ++pass) {  // 2 / 65^6 = 3e-11: with the two inverse iterations below, enough
          const double step = (hi - lo) * (1.0 / 65.0);
          const double sig = lo + step * (lane + 1);
          double d0 = 1.0, d1 = sig - ndp_readlane_f64(ta, 0);
          bool ok = d1 > 0.0;
          for (int j = 1; j < m; ++j) {
            const double aj = ndp_readlane_f64(ta, j), bb = ndp_readlane_f64(tb2, j - 1);
            const double d2 = (sig - aj) * d1 - bb * d0;
            d0 = d1;
            d1 = d2;
            ok = ok && d2 > 0.0;
          }
          const unsigned long long okm = __ballot(ok);
          const int f = okm ? __ffsll(static_cast<long long>(okm)) - 1 : 64;  // first shift above the largest eigenvalue
          const double nlo = lo + step * f, nhi = f == 64 ? hi : lo + step * (f + 1);
          lo = nlo;
          hi = nhi;
        }
        const double sig = hi + 4e-16 * (fabs(hi) + 1.0);
        double yl = lane < m ? 1.0 : 0.0;  // lane j: y_j
        double dinv = 0.0, zl = 0.0;
        for (int rep = 0; rep < 2; ++rep) {
          // (sigma I - T) y = b:  delta_0 = sigma - a_0, delta_j = (sigma - a_j) - b_{j-1}^2 / delta_{j-1},
          // z_j = b_j + b_{j-1} z_{j-1} / delta_{j-1};  y_{m-1} = z_{m-1} / delta_{m-1}, y_j = (z_j + b_j y_{j+1}) / delta_j
          double dprev_inv = 0.0, zprev = 0.0;
          for (int j = 0; j < m; ++j) {
            const double aj = ndp_readlane_f64(ta, j), bjm = j ? ndp_readlane_f64(tb, j - 1) : 0.0;
            double dj = (sig - aj) - bjm * bjm * dprev_inv;
            if (!(dj > 1e-20)) dj = 1e-20;
            const double zj = ndp_readlane_f64(yl, j) + bjm * zprev * dprev_inv;
            dprev_inv = ndp_rcp(dj);
            zprev = zj;
            if (lane == j) { dinv = dprev_inv; zl = zj; }
          }
          double ynext = 0.0;
          for (int j = m - 1; j >= 0; --j) {
            const double bj = j + 1 < m ? ndp_readlane_f64(tb, j) : 0.0;
            const double yj = (ndp_readlane_f64(zl, j) + bj * ynext) * ndp_readlane_f64(dinv, j);
            ynext = yj;
            if (lane == j) yl = yj;
          }
          const double y2 = ndp_wave_sum(lane < m ? yl * yl : 0.0);
          yl = lane < m ? yl * ndp_rsqrt(y2) : 0.0;
        }
        double v = x0, vp = 0.0, bprev = 0.0, xn = 0.0;
        for (int j = 0; j < m; ++j) {
          xn += ndp_readlane_f64(yl, j) * v;
          if (j + 1 == m) break;
          double u = matvec(v);
          u = (u - ndp_readlane_f64(ta, j) * v) - bprev * vp;
          vp = v;
          v = u * ndp_readlane_f64(tib, j);
          bprev = ndp_readlane_f64(tb, j);
        }
        double sn[2] = {xn * xn, xn * x0};
        ndp_block_sums<64, 2>(sn, nullptr);
        if (sn[0] > 1e-200 && sn[0] < 1e200) {  // (else: keep the start vector; the loop below does the work as before)
          const double sc = ndp_rsqrt(sn[0]);
          x = xn * (sn[1] < 0.0 ? -sc : sc);  // the start vector's side, as the LOBPCG iterates keep it (c0 >= 0)
        }
      }
    }
    double ax = matvec(x), pv = 0.0, ap = 0.0;
    double lam = ndp_wave_sum(x * ax);
    bool has_p = false;
    for (; it < max_iter; ++it) {
      const double r = ax - lam * x;
      double sa[3] = {r * r, x * pv, r * pv};
      ndp_block_sums<64, 3>(sa, nullptr);
      const double rn2 = sa[0];
      if (!(rn2 > tol * tol * lam * lam)) break;  // |Ls x - lambda x| <= tol * lambda: converged
      const double inv_r = ndp_rsqrt(rn2);
      const double cxp = sa[1], cwp = sa[2] * inv_r;
      const double wv = r * inv_r;
      const double aw = matvec(wv);
      if (has_p) {
        pv = pv - cxp * x - cwp * wv;
        ap = ap - cxp * ax - cwp * aw;
      } else {
        pv = 0.0;
        ap = 0.0;
      }
      double sb[6] = {pv * pv, x * aw, wv * aw, x * ap, wv * ap, pv * ap};
      ndp_block_sums<64, 6>(sb, nullptr);
      const int dim = (has_p && sb[0] > 1e-24) ? 3 : 2;
      const double ip = dim == 3 ? ndp_rsqrt(sb[0]) : 0.0;
      const double h[3][3] = {{lam, sb[1], sb[3] * ip}, {sb[1], sb[2], sb[4] * ip}, {sb[3] * ip, sb[4] * ip, sb[5] * ip * ip}};
      double c[3];
      ndp_rr_largest_f32(h, dim, c);
      if (c[0] < 0.0) { c[0] = -c[0]; c[1] = -c[1]; c[2] = -c[2]; }
      const double c2p = c[2] * ip;
      const double pn2 = c[1] * c[1] + c[2] * c[2];
      const double sp = pn2 > 1e-300 ? ndp_rsqrt(pn2) : 0.0;
      const double pn = c[1] * wv + c2p * pv, apn = c[1] * aw + c2p * ap;
      const double xn = c[0] * x + pn, axn = c[0] * ax + apn;
      pv = pn * sp;
      ap = apn * sp;
      double sc[2] = {xn * xn, xn * axn};
      ndp_block_sums<64, 2>(sc, nullptr);
      const double ix = ndp_rsqrt(sc[0]);
      x = xn * ix;
      ax = axn * ix;
      lam = sc[1] * ix * ix;  // (= sc[1] / sc[0] to rounding, without the division's ~15 dependent instructions)
      has_p = sp > 0.0;
    }
    if (!(lam > 0.0)) random_part = true;
  }
  if (!random_part) {
    xs[lane] = x;
    __builtin_amdgcn_wave_barrier();
    double cross = 0.0;
    const bool zi = x >= 0.0;
    for (int e = e_beg; e < e_end; ++e) {
      const int64_t c = col[e];
      if (c < p0 || c >= p1) continue;
      if ((xs[c - p0] >= 0.0) != zi) cross += w ? static_cast<double>(w[e]) : 1.0;
    }
    const double cut = ndp_wave_sum(cross) / vol;
    if (cut < 0.5) random_part = true;  // ndp_select.py:250-252
  }
  if (on) {
    bool pos;
    if (random_part) {
      pos = lane == 0 ? true : (lane == 1 ? false : (ndp_hash(seed, static_cast<uint64_t>(p0 + lane)) & 1u) != 0);
    } else {
      pos = x >= 0.0;
    }
    keep[p0 + lane] = pos ? 1 : 0;
  }
  if (lane == 0) info[g] = random_part ? -1 : it;
}

// ====================================================================================================================
// r3: ONE LARGE GRAPH on the whole chip (graphs beyond NDP_MAX_N nodes; the N = 1M, E = 10M graph of BASELINE configs[3]).
// The same iteration as ndp_partition_kernel above -- LOBPCG with one vector on Ls = I - D^-1/2 A D^-1/2, orthonormalised
// basis {x, w = r / |r|, p' = p - (x.p) x - (w.p) w}, 3 x 3 Rayleigh-Ritz, sign partition, cut test, the reference's
// random fallback -- with every vector pass as a grid-wide kernel: fp64 vectors in the workspace, block partial sums in a
// fixed layout reduced in a fixed order by one small workgroup (deterministic), scalars handed from kernel to kernel
// through a device-side state record, so that the host launches steps in batches and reads one flag per batch.
// Per step: round A (residual, 3 sums), sparse mat-vec of the residual, round B (6 sums), update (2 sums); 32 MB-48 MB
// of vector traffic per round at N = 1M plus one pass over the CSR.
constexpr int NL_BLOCKS = 512;      // partial-sum slots (grid of every reducing kernel)
constexpr int NL_THREADS = 256;
// r4: HUB ROWS.  Every kernel that walks the CSR gives a row to one thread (start, cut) or one 16-lane group (mat-vec):
// a 100 000-entry row -- the hub of a power-law graph -- was then 6 000 dependent trips per LOBPCG step for one group
// while the chip idled (ten such hubs in a 1M-node graph: 7.4 ms per step against 0.15; 200 ms per partition against 5).
// Rows beyond NL_HUB entries are listed once per call (sorted: every sum keeps a fixed order) and each is reduced by a
// WHOLE WORKGROUP, four entries per thread in flight, in a second phase of the same kernels; workgroup h % grid takes
// listed row h, so neighbouring hubs (the first nodes of a preferential-attachment graph) spread over the chip.  More
// than NL_HUB_CAP such rows: the surplus is not listed and keeps the in-line path (a row is a hub iff it is LISTED).
constexpr int NL_HUB = 1024;
constexpr int NL_HUB_CAP = 4096;

struct NlState {        // device-resident scalars of the iteration
  double lam, rn2, inv_r, cxp, cwp, c0, c1, c2p, sp, scale, vol, x2, cut;
  int has_p, done, it, random_part, max_iter;
  int done_seen;  // `done` as of the end of the previous step (fused steps: written by the update / start kernels only)
};

struct NlVecs {
  double *x, *ax, *pv, *ap, *wv, *aw, *raw;
  double* ts;          // dis * raw: the vector the sparse mat-vec gathers (ONE random 8-byte read per entry, not two)
  float* dis;
  double* partial;     // [NL_BLOCKS][8]
  double* part_a;      // [NL_BLOCKS][4]   sums of the residual round  (fused steps: nl_update_a_kernel -> nl_matvec2_kernel)
  double* part_b;      // [NL_BLOCKS][16]  Gram sums of the basis      (nl_round_b2_kernel -> nl_update_a_kernel)
  NlState* st;
  int32_t* hub;        // [NL_HUB_CAP] listed hub rows, ascending (nl_hub_setup_kernel)
  int32_t* hub_raw;    // [NL_HUB_CAP] the same in arrival order (nl_init_kernel)
  int* hub_cnt;        // [0] rows that asked for a place in the list, [1] rows listed
  double* hub_part;    // [NL_HUB_CAP][2] degree and start-vector square of every listed row (start only)
};

static size_t nl_layout(void* ws, int64_t n, NlVecs* out) {
  Carver c(ws);
  NlVecs v;
  const size_t m = static_cast<size_t>(n > 0 ? n : 1);
  v.st = c.take<NlState>(1);
  v.partial = c.take<double>(NL_BLOCKS * 8);
  v.part_a = c.take<double>(NL_BLOCKS * 4);
  v.part_b = c.take<double>(NL_BLOCKS * 16);
  v.x = c.take<double>(m); v.ax = c.take<double>(m); v.pv = c.take<double>(m); v.ap = c.take<double>(m);
  v.wv = c.take<double>(m); v.aw = c.take<double>(m); v.raw = c.take<double>(m);
  v.ts = c.take<double>(m);
  v.dis = c.take<float>(m);
  v.hub = c.take<int32_t>(NL_HUB_CAP);
  v.hub_raw = c.take<int32_t>(NL_HUB_CAP);
  v.hub_cnt = c.take<int>(4);
  v.hub_part = c.take<double>(NL_HUB_CAP * 2);
  if (out) *out = v;
  return c.off;
}

// is row i (more than NL_HUB entries) one of the nh listed rows?  (binary search; only rows that long ever ask)
__device__ __forceinline__ bool nl_listed(const int32_t* __restrict__ hub, int nh, int64_t i) {
  int lo = 0, hi = nh;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (hub[mid] < i) lo = mid + 1; else hi = mid;
  }
  return lo < nh && hub[lo] == i;
}

// sum over the entries e of a hub row [a, b) of  w[e] * g(col[e] - p0)  by the WHOLE workgroup: four entries per thread
// in flight, every load unconditional (clamped), level by level; fixed order (thread-strided, then the workgroup tree).
// Entries that leave [p0, p0 + n) are dropped (and flagged when `status` is given).  Every thread gets the sum.
template <typename G>
__device__ __forceinline__ double nl_hub_row(const int64_t* __restrict__ col, const float* __restrict__ w, int a, int b,
                                             int64_t p0, int64_t n, int* __restrict__ status, G g, double* s_red) {
  constexpr int U = 4;
  double acc = 0.0;
  for (int e0 = a + static_cast<int>(threadIdx.x); e0 < b; e0 += U * NL_THREADS) {
    int64_t c[U];
    float ww[U];
    bool ok[U];
    double t[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = e0 + u * NL_THREADS;
      ok[u] = idx < b;
      const int ci = ok[u] ? idx : a;
      c[u] = col[ci];
      ww[u] = w ? w[ci] : 1.0f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = c[u] - p0;
      const bool inr = static_cast<uint64_t>(j) < static_cast<uint64_t>(n);
      if (ok[u] && !inr && status) atomicOr(status, 2);
      ok[u] = ok[u] && inr;
      t[u] = g(inr ? j : 0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += ok[u] ? static_cast<double>(ww[u]) * t[u] : 0.0;
  }
  double sv[1] = {acc};
  __syncthreads();  // s_red may still be read by the previous row's tree
  ndp_block_sums<NL_THREADS, 1>(sv, s_red);
  return sv[0];
}

template <int K>
__device__ __forceinline__ void nl_store_partials(double (&v)[K], double* __restrict__ partial) {
  __shared__ double s_red[K * (NL_THREADS / 64)];
  ndp_block_sums<NL_THREADS, K>(v, s_red);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) partial[blockIdx.x * 8 + k] = v[k];
  }
}

// sums of the first K columns of partial[NL_BLOCKS][8], in slot order (one workgroup; every thread gets them)
template <int K>
__device__ __forceinline__ void nl_reduce_partials(const double* __restrict__ partial, double (&out)[K]) {
  __shared__ double s_red[K * (NL_THREADS / 64)];
  double v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = 0.0;
  for (int b = threadIdx.x; b < NL_BLOCKS; b += NL_THREADS) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += partial[b * 8 + k];
  }
  ndp_block_sums<NL_THREADS, K>(v, s_red);
#pragma unroll
  for (int k = 0; k < K; ++k) out[k] = v[k];
}

// the same with a slot stride / slot count of the caller's choice (fused steps below)
template <int K, int STRIDE>
__device__ __forceinline__ void nl_store_slots(double (&v)[K], double* __restrict__ partial) {
  __shared__ double s_red[K * (NL_THREADS / 64)];
  ndp_block_sums<NL_THREADS, K>(v, s_red);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) partial[blockIdx.x * STRIDE + k] = v[k];
  }
}

template <int K, int STRIDE>
__device__ __forceinline__ void nl_reduce_slots(const double* __restrict__ partial, int nslots, double (&out)[K]) {
  __shared__ double s_red[K * (NL_THREADS / 64)];
  double v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = 0.0;
  for (int b = threadIdx.x; b < nslots; b += NL_THREADS) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += partial[b * STRIDE + k];
  }
  ndp_block_sums<NL_THREADS, K>(v, s_red);
#pragma unroll
  for (int k = 0; k < K; ++k) out[k] = v[k];
}

// row i of  M src  with M = D^-1/2 A D^-1/2 restricted to the graph's nodes [p0, p1)
__device__ __forceinline__ double nl_row_matvec(const int32_t* __restrict__ indptr, const int64_t* __restrict__ col,
                                                const float* __restrict__ w, const float* __restrict__ dis,
                                                const double* __restrict__ src, int64_t p0, int64_t p1, int64_t i,
                                                int* __restrict__ status) {
  double acc = 0.0;
  for (int e = indptr[p0 + i]; e < indptr[p0 + i + 1]; ++e) {
    const int64_t c = col[e];
    if (c < p0 || c >= p1) { atomicOr(status, 2); continue; }
    const int64_t j = c - p0;
    acc += (w ? static_cast<double>(w[e]) : 1.0) * static_cast<double>(dis[j]) * src[j];
  }
  return static_cast<double>(dis[i]) * acc;
}

// degrees -> dis, volume; pseudo-random start vector (raw) and its squared norm
__global__ __launch_bounds__(NL_THREADS) void nl_init_kernel(const int32_t* __restrict__ indptr,
                                                             const float* __restrict__ w, int64_t p0, int64_t n,
                                                             NlVecs v) {
  double s[2] = {0.0, 0.0};
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * NL_THREADS + threadIdx.x; i < n;
       i += static_cast<int64_t>(NL_BLOCKS) * NL_THREADS) {
    double d = 0.0;
    const int ra = indptr[p0 + i], rb = indptr[p0 + i + 1];
    if (rb - ra > NL_HUB) {  // a hub row: listed, its degree / start entry / sums come from nl_hub_setup_kernel
      const int pos = atomicAdd(v.hub_cnt, 1);
      if (pos < NL_HUB_CAP) {
        v.hub_raw[pos] = static_cast<int32_t>(i);
        continue;
      }
    }
    if (!w) d = static_cast<double>(rb - ra);
    else
      for (int e = ra; e < rb; ++e) d += static_cast<double>(w[e]);
    v.dis[i] = d > 0.0 ? static_cast<float>(1.0 / sqrt(d)) : 0.f;
    const double x0 =
        (static_cast<double>(ndp_hash(0x5EEDull, static_cast<uint64_t>(i) + 977ull * static_cast<uint64_t>(n)) >> 8) /
         8388608.0) - 1.0;
    v.raw[i] = x0;
    s[0] += d;
    s[1] += x0 * x0;
  }
  nl_store_partials<2>(s, v.partial);
}

// The listed hub rows in ascending order (every workgroup sorts the few of them in LDS; workgroup 0 writes the list),
// then what nl_init_kernel does for a row, one workgroup per hub row.
__global__ __launch_bounds__(NL_THREADS) void nl_hub_setup_kernel(const int32_t* __restrict__ indptr,
                                                                  const float* __restrict__ w, int64_t p0, int64_t n,
                                                                  NlVecs v) {
  __shared__ int32_t s_list[NL_HUB_CAP];
  __shared__ double s_red[NL_THREADS / 64];
  const int asked = v.hub_cnt[0];
  const int nh = asked < NL_HUB_CAP ? asked : NL_HUB_CAP;
  if (nh == 0) {
    if (blockIdx.x == 0 && threadIdx.x == 0) v.hub_cnt[1] = 0;
    return;
  }
  int P = 1;
  while (P < nh) P <<= 1;
  for (int t = threadIdx.x; t < P; t += NL_THREADS) s_list[t] = t < nh ? v.hub_raw[t] : 0x7FFFFFFF;
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < P; t += NL_THREADS) {
        const int x = t ^ j;
        if (x > t) {
          const int32_t lo = s_list[t], hi = s_list[x];
          if ((lo > hi) == ((t & k) == 0)) {
            s_list[t] = hi;
            s_list[x] = lo;
          }
        }
      }
      __syncthreads();
    }
  }
  if (blockIdx.x == 0) {
    for (int t = threadIdx.x; t < nh; t += NL_THREADS) v.hub[t] = s_list[t];
    if (threadIdx.x == 0) v.hub_cnt[1] = nh;
  }
  for (int h = blockIdx.x; h < nh; h += gridDim.x) {
    const int64_t i = s_list[h];
    const int a = indptr[p0 + i], b = indptr[p0 + i + 1];
    double d;
    if (!w) {
      d = static_cast<double>(b - a);
    } else {
      double sv[1] = {0.0};
      for (int e = a + static_cast<int>(threadIdx.x); e < b; e += NL_THREADS) sv[0] += static_cast<double>(w[e]);
      __syncthreads();
      ndp_block_sums<NL_THREADS, 1>(sv, s_red);
      d = sv[0];
    }
    if (threadIdx.x == 0) {
      v.dis[i] = d > 0.0 ? static_cast<float>(1.0 / sqrt(d)) : 0.f;
      const double x0 =
          (static_cast<double>(ndp_hash(0x5EEDull, static_cast<uint64_t>(i) + 977ull * static_cast<uint64_t>(n)) >> 8) /
           8388608.0) - 1.0;
      v.raw[i] = x0;
      v.hub_part[2 * h] = d;
      v.hub_part[2 * h + 1] = x0 * x0;
    }
  }
}

__global__ __launch_bounds__(NL_THREADS) void nl_init_reduce_kernel(NlVecs v, int max_iter) {
  double s[2];
  nl_reduce_partials<2>(v.partial, s);
  {  // the hub rows' share, in list (= row) order
    __shared__ double s_hub[2 * (NL_THREADS / 64)];
    const int nh = v.hub_cnt[1];
    double hv[2] = {0.0, 0.0};
    for (int h = threadIdx.x; h < nh; h += NL_THREADS) {
      hv[0] += v.hub_part[2 * h];
      hv[1] += v.hub_part[2 * h + 1];
    }
    __syncthreads();
    ndp_block_sums<NL_THREADS, 2>(hv, s_hub);
    s[0] += hv[0];
    s[1] += hv[1];
  }
  if (threadIdx.x == 0) {
    NlState* st = v.st;
    st->vol = s[0];
    st->x2 = s[1];
    st->random_part = !(s[0] > 0.0);
    st->done = st->random_part;
    st->has_p = 0;
    st->it = 0;
    st->scale = 1.0;
    st->max_iter = max_iter;
    st->cut = 0.0;
    st->done_seen = st->random_part;
  }
}

// x = raw / |raw|, ax = Ls x, p = ap = 0; partial of x . ax
__global__ __launch_bounds__(NL_THREADS) void nl_first_matvec_kernel(const int32_t* __restrict__ indptr,
                                                                     const int64_t* __restrict__ col,
                                                                     const float* __restrict__ w, int64_t p0,
                                                                     int64_t p1, NlVecs v, int* __restrict__ status) {
  const int64_t n = p1 - p0;
  const double inv = v.st->x2 > 0.0 ? 1.0 / sqrt(v.st->x2) : 0.0;
  const int nh = v.hub_cnt[1];
  double s[1] = {0.0};
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * NL_THREADS + threadIdx.x; i < n;
       i += static_cast<int64_t>(NL_BLOCKS) * NL_THREADS) {
    if (indptr[p0 + i + 1] - indptr[p0 + i] > NL_HUB && nl_listed(v.hub, nh, i)) continue;  // second phase
    const double xi = v.raw[i] * inv;
    const double axi = xi - inv * nl_row_matvec(indptr, col, w, v.dis, v.raw, p0, p1, i, status);
    v.x[i] = xi;
    v.ax[i] = axi;
    v.pv[i] = 0.0;
    v.ap[i] = 0.0;
    s[0] += xi * axi;
  }
  {  // hub rows: one workgroup each
    __shared__ double s_hub[NL_THREADS / 64];
    const float* dis = v.dis;
    const double* raw = v.raw;
    for (int h = blockIdx.x; h < nh; h += gridDim.x) {
      const int64_t i = v.hub[h];
      const double acc = nl_hub_row(col, w, indptr[p0 + i], indptr[p0 + i + 1], p0, n, status,
                                    [=](int64_t j) { return static_cast<double>(dis[j]) * raw[j]; }, s_hub);
      if (threadIdx.x == 0) {
        const double xi = v.raw[i] * inv;
        const double axi = xi - inv * (static_cast<double>(v.dis[i]) * acc);
        v.x[i] = xi;
        v.ax[i] = axi;
        v.pv[i] = 0.0;
        v.ap[i] = 0.0;
        s[0] += xi * axi;
      }
    }
  }
  nl_store_partials<1>(s, v.partial);
}

__global__ __launch_bounds__(NL_THREADS) void nl_first_reduce_kernel(NlVecs v) {
  double s[1];
  nl_reduce_partials<1>(v.partial, s);
  if (threadIdx.x == 0) v.st->lam = s[0];
}

// round A: x, ax rescaled by the previous step's 1 / |x_new|; raw = ax - lam x; sums |r|^2, x.p, r.p
__global__ __launch_bounds__(NL_THREADS) void nl_round_a_kernel(int64_t n, NlVecs v) {
  const NlState* st = v.st;
  if (st->done) return;
  const double sc = st->scale, lam = st->lam;
  double s[3] = {0.0, 0.0, 0.0};
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * NL_THREADS + threadIdx.x; i < n;
       i += static_cast<int64_t>(NL_BLOCKS) * NL_THREADS) {
    const double xi = v.x[i] * sc, axi = v.ax[i] * sc, pi = v.pv[i];
    v.x[i] = xi;
    v.ax[i] = axi;
    const double r = axi - lam * xi;
    v.raw[i] = r;
    v.ts[i] = static_cast<double>(v.dis[i]) * r;
    s[0] += r * r;
    s[1] += xi * pi;
    s[2] += r * pi;
  }
  nl_store_partials<3>(s, v.partial);
}

__global__ __launch_bounds__(NL_THREADS) void nl_reduce_a_kernel(NlVecs v, double tol) {
  NlState* st = v.st;
  if (st->done) return;
  double s[3];
  nl_reduce_partials<3>(v.partial, s);
  if (threadIdx.x == 0) {
    st->scale = 1.0;
    st->rn2 = s[0];
    if (!(s[0] > tol * tol * st->lam * st->lam) || st->it >= st->max_iter) {
      st->done = 1;  // |Ls x - lambda x| <= tol * lambda (or the step budget is spent: the caller decides what then)
    } else {
      st->inv_r = 1.0 / sqrt(s[0]);
      st->cxp = s[1];
      st->cwp = s[2] * st->inv_r;
    }
  }
}

// w = r / |r|, aw = Ls w.  NL_G lanes share a row (rows of ~10 entries: one lane per row leaves the 64 lanes of a
// load instruction on 64 different cache lines of col / w; with 8 lanes per row consecutive lanes read consecutive
// entries; 16 since r3), partial sums folded inside the lane group in a fixed order.  r3: a group works on NL_U rows at once with
// every load unconditional (clamped) and issued level by level -- row offsets, then columns, then ONE gather per entry
// from ts = dis * r (round A writes it) instead of dis[j] and r[j]: 242 us per call at N = 1M, E = 10M before (72 % of a
// LOBPCG step; a dependent chain of three loads per row with one row in flight per group).  Now ~120 us: 10 M random
// 8-byte reads are 10 M L2 sectors whatever the group shape (8 x 4, 8 x 8, 16 x 4, 16 x 2 lanes x rows measured within 8 %).
constexpr int NL_G = 16;
constexpr int NL_U = 2;
// FUSED: the residual round's sums are reduced here (what nl_reduce_a_kernel did in front).  FOLD_B (graphs of at most
// NL_FOLD_MAX_N nodes, where the step is launch latency): the lane that finishes a row also does round B for it, so the
// step is two launches; the grid is capped at NL_BLOCKS workgroups (= partial-sum slots).
constexpr int64_t NL_FOLD_MAX_N = 131072;
template <bool FUSED, bool FOLD_B>
__global__ __launch_bounds__(NL_THREADS) void nl_matvec_kernel(const int32_t* __restrict__ indptr,
                                                               const int64_t* __restrict__ col,
                                                               const float* __restrict__ w, int64_t p0, int64_t p1,
                                                               NlVecs v, int nslots, double tol,
                                                               int* __restrict__ status) {
  NlState* st = v.st;
  const int64_t n = p1 - p0;
  double inv_r, cxp = 0.0, cwp = 0.0;
  bool has_p = false;
  if constexpr (FUSED) {
    // every field read here was written by an EARLIER launch (done_seen by the update kernel), so the loads go out
    // together with the partial sums' and the exit below is uniform over the grid
    const int seen = st->done_seen, it = st->it, max_iter = st->max_iter;
    const double lam = st->lam, sp_prev = st->sp;
    double sa[3];
    nl_reduce_slots<3, 4>(v.part_a, nslots, sa);
    if (seen) return;
    const bool stop = !(sa[0] > tol * tol * lam * lam) || it >= max_iter;
    inv_r = stop ? 0.0 : 1.0 / sqrt(sa[0]);
    has_p = sp_prev > 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      st->rn2 = sa[0];
      if (stop) {
        st->done = 1;  // |Ls x - lambda x| <= tol * lambda (or the step budget is spent: the caller decides what then)
      } else {
        st->inv_r = inv_r;
        st->cxp = sa[1];
        st->cwp = sa[2] * inv_r;
        st->has_p = has_p;
      }
    }
    if (stop) return;
    if constexpr (FOLD_B) {
      cxp = sa[1];
      cwp = sa[2] * inv_r;
    }
  } else {
    if (st->done) return;
    inv_r = st->inv_r;
  }
  double sb[FOLD_B ? 12 : 1];
#pragma unroll
  for (int k = 0; k < (FOLD_B ? 12 : 1); ++k) sb[k] = 0.0;
  const int sub = threadIdx.x % NL_G;
  const int64_t ngroups = static_cast<int64_t>(gridDim.x) * (NL_THREADS / NL_G);
  const int e_first = indptr[p0];  // a valid entry index whenever any loop below runs
  const int nh = v.hub_cnt[1];
  for (int64_t i0 = static_cast<int64_t>(blockIdx.x) * (NL_THREADS / NL_G) + threadIdx.x / NL_G; i0 < n;
       i0 += ngroups * NL_U) {
    int e[NL_U], e1[NL_U];
    double acc[NL_U];
    double r_i[NL_U], x_i[NL_U], ax_i[NL_U], p_i[NL_U], ap_i[NL_U];  // the row's own entries, requested with its offsets
    float d_i[NL_U];
    bool hub_i[NL_U];
#pragma unroll
    for (int u = 0; u < NL_U; ++u) {
      const int64_t i = i0 + u * ngroups;
      const int64_t ic = i < n ? i : n - 1;
      const int a = indptr[p0 + ic], b = indptr[p0 + ic + 1];
      hub_i[u] = b - a > NL_HUB && nl_listed(v.hub, nh, ic);  // a listed hub row: the second phase below takes it
      r_i[u] = v.raw[ic];
      d_i[u] = v.dis[ic];
      if constexpr (FOLD_B) {
        x_i[u] = v.x[ic];
        ax_i[u] = v.ax[ic];
        p_i[u] = v.pv[ic];
        ap_i[u] = v.ap[ic];
      }
      e[u] = a + sub;
      e1[u] = (i < n && !hub_i[u]) ? b : a;
      acc[u] = 0.0;
    }
    bool more = false;
#pragma unroll
    for (int u = 0; u < NL_U; ++u) more = more || e[u] < e1[u];
    while (__any(more)) {
      int64_t c[NL_U];
      float ww[NL_U];
      double g[NL_U];
      bool ok[NL_U];
#pragma unroll
      for (int u = 0; u < NL_U; ++u) {
        ok[u] = e[u] < e1[u];
        const int idx = ok[u] ? e[u] : e_first;
        c[u] = col[idx];
        ww[u] = w ? w[idx] : 1.0f;
      }
#pragma unroll
      for (int u = 0; u < NL_U; ++u) {
        const int64_t j = c[u] - p0;
        const bool inr = static_cast<uint64_t>(j) < static_cast<uint64_t>(n);
        if (ok[u] && !inr) atomicOr(status, 2);
        ok[u] = ok[u] && inr;
        g[u] = v.ts[inr ? j : 0];
      }
      more = false;
#pragma unroll
      for (int u = 0; u < NL_U; ++u) {
        acc[u] += ok[u] ? static_cast<double>(ww[u]) * g[u] : 0.0;
        e[u] += NL_G;
        more = more || e[u] < e1[u];
      }
    }
#pragma unroll
    for (int u = 0; u < NL_U; ++u) {
      double a = acc[u];
#pragma unroll
      for (int o = NL_G / 2; o > 0; o >>= 1) a += __shfl_xor(a, o, WAVE);
      const int64_t i = i0 + u * ngroups;
      if (sub == 0 && i < n && !hub_i[u]) {
        const double ri = r_i[u];
        const double wi = ri * inv_r, awi = inv_r * (ri - static_cast<double>(d_i[u]) * a);
        v.wv[i] = wi;
        v.aw[i] = awi;
        if constexpr (FOLD_B) {  // round B of this row (nl_round_b2_kernel)
          const double xi = x_i[u], axi = ax_i[u];
          double pi = 0.0, api = 0.0;
          if (has_p) {
            pi = p_i[u] - cxp * xi - cwp * wi;
            api = ap_i[u] - cxp * axi - cwp * awi;
            v.pv[i] = pi;
            v.ap[i] = api;
          }
          sb[0] += pi * pi;
          sb[1] += xi * awi;
          sb[2] += wi * awi;
          sb[3] += xi * api;
          sb[4] += wi * api;
          sb[5] += pi * api;
          sb[6] += xi * xi;
          sb[7] += xi * wi;
          sb[8] += xi * pi;
          sb[9] += wi * wi;
          sb[10] += wi * pi;
          sb[11] += xi * axi;
        }
      }
    }
  }
  {  // hub rows: one workgroup each, what the loop above does for a row done by thread 0 behind the workgroup's sum
    __shared__ double s_hub[NL_THREADS / 64];
    const double* ts = v.ts;
    for (int h = blockIdx.x; h < nh; h += gridDim.x) {
      const int64_t i = v.hub[h];
      const double a = nl_hub_row(col, w, indptr[p0 + i], indptr[p0 + i + 1], p0, n, status,
                                  [=](int64_t j) { return ts[j]; }, s_hub);
      if (threadIdx.x == 0) {
        const double ri = v.raw[i];
        const double wi = ri * inv_r, awi = inv_r * (ri - static_cast<double>(v.dis[i]) * a);
        v.wv[i] = wi;
        v.aw[i] = awi;
        if constexpr (FOLD_B) {
          const double xi = v.x[i], axi = v.ax[i];
          double pi = 0.0, api = 0.0;
          if (has_p) {
            pi = v.pv[i] - cxp * xi - cwp * wi;
            api = v.ap[i] - cxp * axi - cwp * awi;
            v.pv[i] = pi;
            v.ap[i] = api;
          }
          sb[0] += pi * pi;
          sb[1] += xi * awi;
          sb[2] += wi * awi;
          sb[3] += xi * api;
          sb[4] += wi * api;
          sb[5] += pi * api;
          sb[6] += xi * xi;
          sb[7] += xi * wi;
          sb[8] += xi * pi;
          sb[9] += wi * wi;
          sb[10] += wi * pi;
          sb[11] += xi * axi;
        }
      }
    }
  }
  if constexpr (FOLD_B) nl_store_slots<12, 16>(sb, v.part_b);
}

// round B: p' = p - (x.p) x - (w.p) w (and the same combination of the products); six sums
__global__ __launch_bounds__(NL_THREADS) void nl_round_b_kernel(int64_t n, NlVecs v) {
  const NlState* st = v.st;
  if (st->done) return;
  const bool has_p = st->has_p != 0;
  const double cxp = st->cxp, cwp = st->cwp;
  double s[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * NL_THREADS + threadIdx.x; i < n;
       i += static_cast<int64_t>(NL_BLOCKS) * NL_THREADS) {
    const double xi = v.x[i], wi = v.wv[i], axi = v.ax[i], awi = v.aw[i];
    double pi = 0.0, api = 0.0;
    if (has_p) {
      pi = v.pv[i] - cxp * xi - cwp * wi;
      api = v.ap[i] - cxp * axi - cwp * awi;
      v.pv[i] = pi;
      v.ap[i] = api;
    }
    s[0] += pi * pi;
    s[1] += xi * awi;
    s[2] += wi * awi;
    s[3] += xi * api;
    s[4] += wi * api;
    s[5] += pi * api;
  }
  nl_store_partials<6>(s, v.partial);
}

__global__ __launch_bounds__(NL_THREADS) void nl_reduce_b_kernel(NlVecs v) {
  NlState* st = v.st;
  if (st->done) return;
  double sb[6];
  nl_reduce_partials<6>(v.partial, sb);
  if (threadIdx.x == 0) {
    const bool has_p = st->has_p != 0;
    const int dim = (has_p && sb[0] > 1e-24) ? 3 : 2;
    const double ip = dim == 3 ? 1.0 / sqrt(sb[0]) : 0.0;
    const double lam = st->lam;
    double h[3][3] = {{lam, sb[1], sb[3] * ip}, {sb[1], sb[2], sb[4] * ip}, {sb[3] * ip, sb[4] * ip, sb[5] * ip * ip}};
    double theta, c[3];
    ndp_eig3_largest(h, dim, theta, c);
    if (c[0] < 0.0) { c[0] = -c[0]; c[1] = -c[1]; c[2] = -c[2]; }
    const double pn2 = c[1] * c[1] + c[2] * c[2];
    st->c0 = c[0];
    st->c1 = c[1];
    st->c2p = c[2] * ip;
    st->sp = pn2 > 1e-300 ? 1.0 / sqrt(pn2) : 0.0;
  }
}

// x <- c0 x + c1 w + c2 p^, p <- (c1 w + c2 p^) / |.| (the same combinations of the products); sums |x|^2, x.Ax
__global__ __launch_bounds__(NL_THREADS) void nl_update_kernel(int64_t n, NlVecs v) {
  const NlState* st = v.st;
  if (st->done) return;
  const double c0 = st->c0, c1 = st->c1, c2p = st->c2p, sp = st->sp;
  double s[2] = {0.0, 0.0};
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * NL_THREADS + threadIdx.x; i < n;
       i += static_cast<int64_t>(NL_BLOCKS) * NL_THREADS) {
    const double pn = c1 * v.wv[i] + c2p * v.pv[i], apn = c1 * v.aw[i] + c2p * v.ap[i];
    const double xn = c0 * v.x[i] + pn, axn = c0 * v.ax[i] + apn;
    v.pv[i] = pn * sp;
    v.ap[i] = apn * sp;
    v.x[i] = xn;
    v.ax[i] = axn;
    s[0] += xn * xn;
    s[1] += xn * axn;
  }
  nl_store_partials<2>(s, v.partial);
}

__global__ __launch_bounds__(NL_THREADS) void nl_reduce_update_kernel(NlVecs v, int32_t* __restrict__ d_progress) {
  NlState* st = v.st;
  if (!st->done) {
    double s[2];
    nl_reduce_partials<2>(v.partial, s);
    if (threadIdx.x == 0) {
      st->scale = 1.0 / sqrt(s[0]);  // applied to x, ax by the next round A (or by the finish kernel)
      st->lam = s[1] / s[0];
      st->has_p = st->sp > 0.0;
      st->it += 1;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && d_progress) {
    d_progress[0] = st->done;
    d_progress[1] = st->it;
  }
}

// ---- r3 (late): the step as THREE launches instead of seven.  A mid-size graph (2 k .. 50 k nodes) spends its step in
// launch latency (~5 us per dependent kernel, seven of them), so (1) every one-workgroup "reduce" kernel is folded into
// the grid kernel that consumes its result: each workgroup sums the producer's partial slots itself (same slots, same
// order in every workgroup: bitwise the same scalars everywhere) and workgroup 0 records what later kernels need; a
// kernel never reads a state field it writes (`done` excepted: reading the value written by this very launch leads to
// the decision the reader would take anyway).  (2) The update's own sums go: with the twelve Gram sums of the basis
// {x, w, p'} from round B, |x_new|^2 = c^T G c and x_new . A x_new = c^T H c are known BEFORE x_new is formed, so update,
// normalisation, the new residual, ts = dis * r and the residual's three sums are one pass (17 vector streams -> 12).
// first residual round of the fused steps (start): raw = ax - lam x, ts, the three sums -> part_a
__global__ __launch_bounds__(NL_THREADS) void nl_round_a2_kernel(int64_t n, NlVecs v) {
  const NlState* st = v.st;
  const double lam = st->lam;
  double s[3] = {0.0, 0.0, 0.0};
  if (!st->done) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * NL_THREADS + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * NL_THREADS) {
      const double xi = v.x[i], pi = v.pv[i];
      const double r = v.ax[i] - lam * xi;
      v.raw[i] = r;
      v.ts[i] = static_cast<double>(v.dis[i]) * r;
      s[0] += r * r;
      s[1] += xi * pi;
      s[2] += r * pi;
    }
  }
  nl_store_slots<3, 4>(s, v.part_a);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    v.st->sp = 0.0;
    v.st->done_seen = st->done;
  }
}

// round B with the full Gram matrices: G = {xx, xw, xp, ww, wp, pp}, H = {x.ax, x.aw, w.aw, x.ap, w.ap, p.ap}
__global__ __launch_bounds__(NL_THREADS) void nl_round_b2_kernel(int64_t n, NlVecs v) {
  const NlState* st = v.st;
  if (st->done) return;  // (written by the mat-vec launch before this one)
  const bool has_p = st->has_p != 0;
  const double cxp = st->cxp, cwp = st->cwp;
  double s[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) s[k] = 0.0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * NL_THREADS + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * NL_THREADS) {
    const double xi = v.x[i], wi = v.wv[i], axi = v.ax[i], awi = v.aw[i];
    double pi = 0.0, api = 0.0;
    if (has_p) {
      pi = v.pv[i] - cxp * xi - cwp * wi;
      api = v.ap[i] - cxp * axi - cwp * awi;
      v.pv[i] = pi;
      v.ap[i] = api;
    }
    s[0] += pi * pi;
    s[1] += xi * awi;
    s[2] += wi * awi;
    s[3] += xi * api;
    s[4] += wi * api;
    s[5] += pi * api;
    s[6] += xi * xi;
    s[7] += xi * wi;
    s[8] += xi * pi;
    s[9] += wi * wi;
    s[10] += wi * pi;
    s[11] += xi * axi;
  }
  nl_store_slots<12, 16>(s, v.part_b);
}

// Rayleigh-Ritz from round B's sums, then in ONE pass: x <- (c0 x + c1 w + c2 p^) / |.|, p <- (c1 w + c2 p^) / |.| (and
// the same combinations of the products), raw = ax - lam x with the new lam = c^T H c / c^T G c, ts = dis * raw, and the
// residual round's three sums.
__global__ __launch_bounds__(NL_THREADS) void nl_update_a_kernel(int64_t n, NlVecs v, int nslots,
                                                                 int32_t* __restrict__ d_progress) {
  NlState* st = v.st;
  // the first element's operands are requested before the sums arrive (a mid-size graph has one element per thread)
  const int64_t stride = static_cast<int64_t>(gridDim.x) * NL_THREADS;
  int64_t i = static_cast<int64_t>(blockIdx.x) * NL_THREADS + threadIdx.x;
  int64_t ic = i < n ? i : 0;
  double wi = v.wv[ic], pi = v.pv[ic], awi = v.aw[ic], api = v.ap[ic], xi = v.x[ic], axi = v.ax[ic];
  float di = v.dis[ic];
  const int done = st->done, has_p_i = st->has_p;  // (written by the mat-vec launch of this step or an earlier one)
  double sb[12];
  nl_reduce_slots<12, 16>(v.part_b, nslots, sb);
  if (done) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      st->done_seen = 1;
      if (d_progress) {
        d_progress[0] = 1;
        d_progress[1] = st->it;
      }
    }
    return;
  }
  __shared__ double s_c[6];
  if (threadIdx.x == 0) {
    const bool has_p = has_p_i != 0;
    const int dim = (has_p && sb[0] > 1e-24) ? 3 : 2;
    const double ip = dim == 3 ? 1.0 / sqrt(sb[0]) : 0.0;
    const double hxx = sb[11], hxw = sb[1], hww = sb[2], hxp = sb[3] * ip, hwp = sb[4] * ip, hpp = sb[5] * ip * ip;
    double h[3][3] = {{hxx, hxw, hxp}, {hxw, hww, hwp}, {hxp, hwp, hpp}};
    double c[3];
    ndp_rr_largest_f32(h, dim, c);  // (as the one-workgroup kernels: the Gram sums below make the normalisation exact)
    if (c[0] < 0.0) { c[0] = -c[0]; c[1] = -c[1]; c[2] = -c[2]; }
    const double pn2 = c[1] * c[1] + c[2] * c[2];
    const double gxp = sb[8] * ip, gwp = sb[10] * ip, gpp = dim == 3 ? 1.0 : 0.0;
    const double x2 = c[0] * c[0] * sb[6] + c[1] * c[1] * sb[9] + c[2] * c[2] * gpp +
                      2.0 * (c[0] * c[1] * sb[7] + c[0] * c[2] * gxp + c[1] * c[2] * gwp);
    const double xax = c[0] * c[0] * hxx + c[1] * c[1] * hww + c[2] * c[2] * hpp +
                       2.0 * (c[0] * c[1] * hxw + c[0] * c[2] * hxp + c[1] * c[2] * hwp);
    s_c[0] = c[0];
    s_c[1] = c[1];
    s_c[2] = c[2] * ip;
    s_c[3] = pn2 > 1e-300 ? 1.0 / sqrt(pn2) : 0.0;
    s_c[4] = 1.0 / sqrt(x2);
    s_c[5] = xax / x2;
  }
  __syncthreads();
  const double c0 = s_c[0], c1 = s_c[1], c2p = s_c[2], sp = s_c[3], sc = s_c[4], lam = s_c[5];
  double s[3] = {0.0, 0.0, 0.0};
  while (i < n) {
    const double pn = c1 * wi + c2p * pi, apn = c1 * awi + c2p * api;
    const double xn = (c0 * xi + pn) * sc, axn = (c0 * axi + apn) * sc;
    const double pnew = pn * sp;
    const double r = axn - lam * xn;
    const int64_t at = i;
    const double ts = static_cast<double>(di) * r;
    i += stride;
    if (i < n) {  // next element's operands before this one's stores
      wi = v.wv[i]; pi = v.pv[i]; awi = v.aw[i]; api = v.ap[i]; xi = v.x[i]; axi = v.ax[i];
      di = v.dis[i];
    }
    v.pv[at] = pnew;
    v.ap[at] = apn * sp;
    v.x[at] = xn;
    v.ax[at] = axn;
    v.raw[at] = r;
    v.ts[at] = ts;
    s[0] += r * r;
    s[1] += xn * pnew;
    s[2] += r * pnew;
  }
  nl_store_slots<3, 4>(s, v.part_a);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->lam = lam;
    st->sp = sp;
    const int it = st->it + 1;
    st->it = it;
    if (d_progress) {
      d_progress[0] = 0;
      d_progress[1] = it;
    }
  }
}

// weight of the directed entries that cross the sign partition
__global__ __launch_bounds__(NL_THREADS) void nl_cut_kernel(const int32_t* __restrict__ indptr,
                                                            const int64_t* __restrict__ col,
                                                            const float* __restrict__ w, int64_t p0, int64_t p1,
                                                            NlVecs v) {
  const int64_t n = p1 - p0;
  double s[1] = {0.0};
  if (!v.st->random_part) {
    const int nh = v.hub_cnt[1];
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * NL_THREADS + threadIdx.x; i < n;
         i += static_cast<int64_t>(NL_BLOCKS) * NL_THREADS) {
      if (indptr[p0 + i + 1] - indptr[p0 + i] > NL_HUB && nl_listed(v.hub, nh, i)) continue;  // second phase
      const bool zi = v.x[i] >= 0.0;
      for (int e = indptr[p0 + i]; e < indptr[p0 + i + 1]; ++e) {
        const int64_t c = col[e];
        if (c < p0 || c >= p1) continue;
        if ((v.x[c - p0] >= 0.0) != zi) s[0] += w ? static_cast<double>(w[e]) : 1.0;
      }
    }
    __shared__ double s_hub[NL_THREADS / 64];
    const double* x = v.x;
    for (int h = blockIdx.x; h < nh; h += gridDim.x) {  // hub rows: one workgroup each
      const int64_t i = v.hub[h];
      const bool zi = v.x[i] >= 0.0;
      const double acc = nl_hub_row(col, w, indptr[p0 + i], indptr[p0 + i + 1], p0, n, static_cast<int*>(nullptr),
                                    [=](int64_t j) { return (x[j] >= 0.0) != zi ? 1.0 : 0.0; }, s_hub);
      if (threadIdx.x == 0) s[0] += acc;
    }
  }
  nl_store_partials<1>(s, v.partial);
}

__global__ __launch_bounds__(NL_THREADS) void nl_cut_reduce_kernel(NlVecs v, int32_t* __restrict__ info) {
  double s[1];
  nl_reduce_partials<1>(v.partial, s);
  if (threadIdx.x == 0) {
    NlState* st = v.st;
    if (!(st->lam > 0.0)) st->random_part = 1;
    if (!st->random_part) {
      st->cut = s[0] / st->vol;
      if (st->cut < 0.5) st->random_part = 1;  // ndp_select.py:250-252
    }
    if (info) *info = st->random_part ? -1 : st->it;
  }
}

__global__ __launch_bounds__(NL_THREADS) void nl_keep_kernel(int64_t p0, int64_t n, unsigned long long seed, NlVecs v,
                                                             uint8_t* __restrict__ keep) {
  const bool random_part = v.st->random_part != 0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * NL_THREADS + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * NL_THREADS) {
    bool pos;
    if (random_part) pos = i == 0 ? true : (i == 1 ? false : (ndp_hash(seed, static_cast<uint64_t>(p0 + i)) & 1u) != 0);
    else pos = v.x[i] >= 0.0;  // (the pending 1 / |x| factor is positive: signs are final)
    keep[p0 + i] = pos ? 1 : 0;
  }
}

}  // namespace tgp

using namespace tgp;

extern "C" int tgp_ndp_max_graph_nodes(void) { return NDP_MAX_N; }

extern "C" int tgp_ndp_partition(const int32_t* indptr, const int64_t* col, const float* w, int64_t N, int64_t nnz,
                                 const int64_t* graph_ptr, int64_t B, int64_t max_graph_nodes, uint64_t seed,
                                 int max_iter, double tol, uint8_t* keep, int32_t* info, int* d_status,
                                 void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(N >= 0 && B >= 0 && nnz >= 0 && max_iter > 0 && tol >= 0.0, TGP_ERR_INVALID,
              "tgp_ndp_partition: bad argument");
  TGP_REQUIRE(d_status, TGP_ERR_INVALID, "tgp_ndp_partition: null status");
  if (N == 0 || B == 0) {
    (void)hipMemsetAsync(d_status, 0, sizeof(int), stream);
    return check_launch("tgp_ndp_partition");
  }
  TGP_REQUIRE(indptr && graph_ptr && keep && info && (col || nnz == 0), TGP_ERR_INVALID,
              "tgp_ndp_partition: null pointer");
  TGP_REQUIRE(N < (1ll << 31) && nnz < (1ll << 31) && B < (1ll << 31), TGP_ERR_RANGE,
              "tgp_ndp_partition: N / nnz / B >= 2^31");
  // status | info | keep are cleared; ONE memset when the caller laid them out back to back (status at the front of a
  // 16-byte slot, info padded to a multiple of four entries: the host mirror does), three otherwise
  const size_t info_bytes = static_cast<size_t>((B + 3) / 4 * 4) * sizeof(int32_t);
  if (reinterpret_cast<uint8_t*>(info) == reinterpret_cast<uint8_t*>(d_status) + 16 &&
      keep == reinterpret_cast<uint8_t*>(info) + info_bytes) {
    (void)hipMemsetAsync(d_status, 0, 16 + info_bytes + static_cast<size_t>(N), stream);
  } else {
    (void)hipMemsetAsync(d_status, 0, sizeof(int), stream);
    (void)hipMemsetAsync(keep, 0, static_cast<size_t>(N), stream);
    (void)hipMemsetAsync(info, 0, static_cast<size_t>(B) * sizeof(int32_t), stream);
  }
  int64_t cap = max_graph_nodes < NDP_MAX_N ? max_graph_nodes : NDP_MAX_N;
  if (cap < 2) cap = 2;
  // entries of a graph's matrix kept in LDS: four times the batch average (graph sizes are not known on the host),
  // at most what one workgroup may take; larger graphs iterate out of L2
  int64_t ecap = 4 * (nnz / B + 1);
  if (ecap < 256) ecap = 256;
  const int64_t fixed = cap * (6 * sizeof(double) + sizeof(float) + sizeof(int)) + 64;
  const int64_t budget = cap <= 64 ? 12 * 1024 : 150 * 1024;  // 64-thread workgroups: keep ~12 of them on a CU
  if (fixed + ecap * 10 > budget) ecap = (budget - fixed) / 10 > 0 ? (budget - fixed) / 10 : 0;
  ecap &= ~static_cast<int64_t>(3);  // keeps the arrays behind the fp64 values 4-byte aligned
  const size_t lds = static_cast<size_t>(fixed + ecap * 10);
  const int ncap = static_cast<int>(cap), ec = static_cast<int>(ecap);
  if (cap <= 64) {
    static const int generic = getenv("TGP_NDP_GENERIC_KERNEL") ? 1 : 0;  // (A/B switch: the LDS-vector kernel)
    // Lanczos steps of the warm start: at most the graph's size and 48 (measured on 2048 graphs of 20-60 nodes: 32 steps
    // leave up to 34 LOBPCG steps to the slowest graph, 48 none, 64 cost 19 us more; profiles/r06_ndp_small.txt).
    // TGP_NDP_LANCZOS=0: LOBPCG alone -- the A/B switch
    static const int warm_steps = getenv("TGP_NDP_LANCZOS") ? atoi(getenv("TGP_NDP_LANCZOS")) : 48;
    if (generic)
      hipLaunchKernelGGL(ndp_partition_kernel<64>, dim3(static_cast<unsigned>(B)), dim3(64), lds, stream, indptr, col, w,
                         graph_ptr, static_cast<unsigned long long>(seed), max_iter, tol, ncap, ec, keep, info, d_status);
    else
      hipLaunchKernelGGL(ndp_partition_wave_kernel, dim3(static_cast<unsigned>(B)), dim3(64), lds, stream, indptr, col,
                         w, graph_ptr, static_cast<unsigned long long>(seed), max_iter, tol, ncap, ec, keep, info,
                         d_status, warm_steps < 0 ? 0 : (warm_steps > 64 ? 64 : warm_steps));
  } else {
    // graphs of 513 .. 2048 nodes: 512 threads per graph (64 graphs of 600 .. 2000 nodes 10.7 -> 7.1 ms; 1024 threads
    // were measured slower than 256: profiles/r03_c3_small_kernel_experiments.md).  TGP_NDP_THREADS=256: A/B switch
    static const int wide = getenv("TGP_NDP_THREADS") ? atoi(getenv("TGP_NDP_THREADS")) : 512;
    if (wide == 512 && cap > 512) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ndp_partition_kernel<512>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
      hipLaunchKernelGGL(ndp_partition_kernel<512>, dim3(static_cast<unsigned>(B)), dim3(512), lds, stream, indptr, col,
                         w, graph_ptr, static_cast<unsigned long long>(seed), max_iter, tol, ncap, ec, keep, info,
                         d_status);
      return check_launch("tgp_ndp_partition");
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ndp_partition_kernel<256>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    hipLaunchKernelGGL(ndp_partition_kernel<256>, dim3(static_cast<unsigned>(B)), dim3(256), lds, stream, indptr, col,
                       w, graph_ptr, static_cast<unsigned long long>(seed), max_iter, tol, ncap, ec, keep, info,
                       d_status);
  }
  return check_launch("tgp_ndp_partition");
}

// ------------------------------------------------------------------ is the list already what NDPSelect needs?
// NDPSelect first sums duplicates, drops self loops and takes the max with the transpose (ndp_select.py:198-202): two
// coalesce calls (two dozen launches, two host read-backs) for a list that, in a PyG dataset of undirected graphs, is
// already row-major sorted, duplicate-free, loop-free and symmetric.  One thread per entry checks exactly that -- strictly
// ascending (row, col), row != col, ids in range, the reverse entry present (binary search in the column's row) -- and
// writes max(w, w_reverse); any violation raises the flag and the caller takes the general route.
__global__ __launch_bounds__(256) void ndp_symmetric_max_kernel(const int64_t* __restrict__ row,
                                                                const int64_t* __restrict__ col,
                                                                const float* __restrict__ w, int64_t E, int64_t n,
                                                                const int32_t* __restrict__ indptr,
                                                                float* __restrict__ w_out, int* __restrict__ flag) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (e >= E || *flag) return;  // (a violation found by an earlier workgroup: the caller takes the general route anyway)
  const int64_t i = row[e], j = col[e];
  bool ok = static_cast<uint64_t>(i) < static_cast<uint64_t>(n) && static_cast<uint64_t>(j) < static_cast<uint64_t>(n) &&
            i != j;
  if (ok && e > 0) {
    const int64_t pi = row[e - 1], pj = col[e - 1];
    ok = pi < i || (pi == i && pj < j);
  }
  float wm = w ? w[e] : 1.0f;
  if (ok) {
    int64_t lo = indptr[j], hi = static_cast<int64_t>(indptr[j + 1]) - 1;
    if (lo < 0) lo = 0;
    if (hi >= E) hi = E - 1;
    int64_t found = -1;
    while (lo <= hi) {
      const int64_t mid = (lo + hi) >> 1;
      const int64_t c = col[mid];
      if (c == i) { found = mid; break; }
      if (c < i) lo = mid + 1; else hi = mid - 1;
    }
    ok = found >= 0 && row[found >= 0 ? found : 0] == j;
    if (ok && w) wm = fmaxf(wm, w[found]);
  }
  if (!ok) *flag = 1;
  w_out[e] = wm;
}

extern "C" int tgp_ndp_symmetric_max_f32(const int64_t* row, const int64_t* col, const float* w, int64_t E, int64_t n,
                                         const int32_t* indptr, float* w_out, int* d_flag, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(E >= 0 && n >= 0 && d_flag, TGP_ERR_INVALID, "tgp_ndp_symmetric_max_f32: bad argument");
  (void)hipMemsetAsync(d_flag, 0, sizeof(int), stream);
  if (E == 0) return check_launch("tgp_ndp_symmetric_max_f32");
  TGP_REQUIRE(row && col && indptr && w_out, TGP_ERR_INVALID, "tgp_ndp_symmetric_max_f32: null pointer");
  TGP_REQUIRE(E < (1ll << 31) && n < (1ll << 31), TGP_ERR_RANGE, "tgp_ndp_symmetric_max_f32: E / n >= 2^31");
  hipLaunchKernelGGL(ndp_symmetric_max_kernel, dim3(static_cast<unsigned>(cdiv(E, 256))), dim3(256), 0, stream, row, col, w,
                     E, n, indptr, w_out, d_flag);
  return check_launch("tgp_ndp_symmetric_max_f32");
}

// ------------------------------------------------------------------ one large graph, chip-wide (see nl_* kernels)
extern "C" size_t tgp_ndp_large_workspace_bytes(int64_t n) { return nl_layout(nullptr, n, nullptr) + 256; }

// TGP_NDP_LARGE_CLASSIC=1: the seven-launch step of the first r3 version (A/B measurements; start and steps must agree,
// so the switch is read once per process)
static bool nl_classic_steps() {
  static const bool classic = [] {
    const char* e = getenv("TGP_NDP_LARGE_CLASSIC");
    return e && e[0] == '1';
  }();
  return classic;
}
static unsigned nl_grid(int64_t n) {  // workgroups (= partial-sum slots) of the fused steps' vector kernels
  const int64_t b = cdiv(n, static_cast<int64_t>(NL_THREADS));
  return static_cast<unsigned>(b < NL_BLOCKS ? (b > 0 ? b : 1) : NL_BLOCKS);
}

static int nl_check(const int32_t* indptr, int64_t p0, int64_t p1, const void* ws, size_t ws_bytes, const char* what) {
  TGP_REQUIRE(indptr && ws && p0 >= 0 && p1 > p0, TGP_ERR_INVALID, "%s: bad argument", what);
  TGP_REQUIRE(p1 < (1ll << 31), TGP_ERR_RANGE, "%s: node ids >= 2^31", what);
  TGP_REQUIRE(ws_bytes >= tgp_ndp_large_workspace_bytes(p1 - p0), TGP_ERR_WORKSPACE, "%s: workspace too small", what);
  return TGP_OK;
}

extern "C" int tgp_ndp_large_start(const int32_t* indptr, const int64_t* col, const float* w, int64_t p0, int64_t p1,
                                   int max_iter, void* ws, size_t ws_bytes, int* d_status, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int rc = nl_check(indptr, p0, p1, ws, ws_bytes, "tgp_ndp_large_start");
  if (rc != TGP_OK) return rc;
  TGP_REQUIRE(col && d_status && max_iter > 0, TGP_ERR_INVALID, "tgp_ndp_large_start: bad argument");
  NlVecs v;
  nl_layout(ws, p1 - p0, &v);
  const int64_t n = p1 - p0;
  (void)hipMemsetAsync(v.hub_cnt, 0, 4 * sizeof(int), stream);
  hipLaunchKernelGGL(nl_init_kernel, dim3(NL_BLOCKS), dim3(NL_THREADS), 0, stream, indptr, w, p0, n, v);
  hipLaunchKernelGGL(nl_hub_setup_kernel, dim3(NL_BLOCKS), dim3(NL_THREADS), 0, stream, indptr, w, p0, n, v);
  hipLaunchKernelGGL(nl_init_reduce_kernel, dim3(1), dim3(NL_THREADS), 0, stream, v, max_iter);
  hipLaunchKernelGGL(nl_first_matvec_kernel, dim3(NL_BLOCKS), dim3(NL_THREADS), 0, stream, indptr, col, w, p0, p1, v,
                     d_status);
  hipLaunchKernelGGL(nl_first_reduce_kernel, dim3(1), dim3(NL_THREADS), 0, stream, v);
  if (!nl_classic_steps())
    hipLaunchKernelGGL(nl_round_a2_kernel, dim3(nl_grid(n)), dim3(NL_THREADS), 0, stream, n, v);
  return check_launch("tgp_ndp_large_start");
}

extern "C" int tgp_ndp_large_steps(const int32_t* indptr, const int64_t* col, const float* w, int64_t p0, int64_t p1,
                                   int steps, double tol, void* ws, size_t ws_bytes, int32_t* d_progress,
                                   int* d_status, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int rc = nl_check(indptr, p0, p1, ws, ws_bytes, "tgp_ndp_large_steps");
  if (rc != TGP_OK) return rc;
  TGP_REQUIRE(col && d_status && steps > 0 && tol >= 0.0, TGP_ERR_INVALID, "tgp_ndp_large_steps: bad argument");
  NlVecs v;
  nl_layout(ws, p1 - p0, &v);
  const int64_t n = p1 - p0;
  const int64_t mv_rows = static_cast<int64_t>(NL_THREADS / NL_G) * NL_U;  // rows a workgroup takes per pass
  const unsigned mv_blocks = static_cast<unsigned>(cdiv(n, mv_rows) < 16384 ? cdiv(n, mv_rows) : 16384);
  if (!nl_classic_steps()) {
    const unsigned nb = nl_grid(n);
    const bool fold = n <= NL_FOLD_MAX_N;
    const unsigned fb = mv_blocks < static_cast<unsigned>(NL_BLOCKS) ? mv_blocks : static_cast<unsigned>(NL_BLOCKS);
    for (int s = 0; s < steps; ++s) {
      int32_t* progress = s + 1 == steps ? d_progress : static_cast<int32_t*>(nullptr);
      if (fold) {
        hipLaunchKernelGGL((nl_matvec_kernel<true, true>), dim3(fb), dim3(NL_THREADS), 0, stream, indptr, col, w, p0, p1,
                           v, static_cast<int>(nb), tol, d_status);
        hipLaunchKernelGGL(nl_update_a_kernel, dim3(nb), dim3(NL_THREADS), 0, stream, n, v, static_cast<int>(fb),
                           progress);
      } else {
        hipLaunchKernelGGL((nl_matvec_kernel<true, false>), dim3(mv_blocks), dim3(NL_THREADS), 0, stream, indptr, col, w,
                           p0, p1, v, static_cast<int>(nb), tol, d_status);
        hipLaunchKernelGGL(nl_round_b2_kernel, dim3(nb), dim3(NL_THREADS), 0, stream, n, v);
        hipLaunchKernelGGL(nl_update_a_kernel, dim3(nb), dim3(NL_THREADS), 0, stream, n, v, static_cast<int>(nb),
                           progress);
      }
    }
    return check_launch("tgp_ndp_large_steps");
  }
  for (int s = 0; s < steps; ++s) {
    hipLaunchKernelGGL(nl_round_a_kernel, dim3(NL_BLOCKS), dim3(NL_THREADS), 0, stream, n, v);
    hipLaunchKernelGGL(nl_reduce_a_kernel, dim3(1), dim3(NL_THREADS), 0, stream, v, tol);
    hipLaunchKernelGGL((nl_matvec_kernel<false, false>), dim3(mv_blocks), dim3(NL_THREADS), 0, stream, indptr, col, w, p0, p1, v,
                       0, tol, d_status);
    hipLaunchKernelGGL(nl_round_b_kernel, dim3(NL_BLOCKS), dim3(NL_THREADS), 0, stream, n, v);
    hipLaunchKernelGGL(nl_reduce_b_kernel, dim3(1), dim3(NL_THREADS), 0, stream, v);
    hipLaunchKernelGGL(nl_update_kernel, dim3(NL_BLOCKS), dim3(NL_THREADS), 0, stream, n, v);
    hipLaunchKernelGGL(nl_reduce_update_kernel, dim3(1), dim3(NL_THREADS), 0, stream, v,
                       s + 1 == steps ? d_progress : static_cast<int32_t*>(nullptr));
  }
  return check_launch("tgp_ndp_large_steps");
}

extern "C" int tgp_ndp_large_finish(const int32_t* indptr, const int64_t* col, const float* w, int64_t p0, int64_t p1,
                                    uint64_t seed, void* ws, size_t ws_bytes, uint8_t* keep, int32_t* info,
                                    void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int rc = nl_check(indptr, p0, p1, ws, ws_bytes, "tgp_ndp_large_finish");
  if (rc != TGP_OK) return rc;
  TGP_REQUIRE(col && keep, TGP_ERR_INVALID, "tgp_ndp_large_finish: null pointer");
  NlVecs v;
  nl_layout(ws, p1 - p0, &v);
  const int64_t n = p1 - p0;
  hipLaunchKernelGGL(nl_cut_kernel, dim3(NL_BLOCKS), dim3(NL_THREADS), 0, stream, indptr, col, w, p0, p1, v);
  hipLaunchKernelGGL(nl_cut_reduce_kernel, dim3(1), dim3(NL_THREADS), 0, stream, v, info);
  hipLaunchKernelGGL(nl_keep_kernel, dim3(static_cast<unsigned>(cdiv(n, NL_THREADS) < 4096 ? cdiv(n, NL_THREADS) : 4096)),
                     dim3(NL_THREADS), 0, stream, p0, n, static_cast<unsigned long long>(seed), v, keep);
  return check_launch("tgp_ndp_large_finish");
}

// lambda (Rayleigh quotient), |residual|^2, steps, random-fallback flag, cut: diagnostics / tests
extern "C" int tgp_ndp_large_state(const void* ws, int64_t n, double* out5, void* stream_) {
  TGP_REQUIRE(ws && out5, TGP_ERR_INVALID, "tgp_ndp_large_state: null pointer");
  NlVecs v;
  nl_layout(const_cast<void*>(ws), n, &v);
  NlState h;
  if (hipMemcpyAsync(&h, v.st, sizeof(NlState), hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream_)) != hipSuccess ||
      hipStreamSynchronize(static_cast<hipStream_t>(stream_)) != hipSuccess) {
    set_error("tgp_ndp_large_state: copy failed");
    return TGP_ERR_LAUNCH;
  }
  out5[0] = h.lam; out5[1] = h.rn2; out5[2] = h.it; out5[3] = h.random_part; out5[4] = h.cut;
  return TGP_OK;
}
