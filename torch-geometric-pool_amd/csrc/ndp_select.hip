// A14: NDPSelect's spectral partition (select/ndp_select.py:187-256) on the device, one workgroup per graph.
//
// The reference loops over the graphs of a batch on the host: symmetric normalised Laplacian Ls = I - D^-1/2 A D^-1/2,
// its largest eigenvector by torch.lobpcg, sign partition z, cut size z^T L z / (2 vol), random +-1 partition when the
// cut is below 0.5 (or the eigen-solve fails).  The partition is only defined up to the eigenvector's sign and solver
// tolerance, so the contract is: z is the sign pattern of a (converged) largest eigenvector of Ls, or the random
// fallback under the reference's rule.  Here every graph runs a power iteration on Ls (positive semi-definite with
// spectrum in [0, 2]: the dominant eigenvector IS the largest one) in fp64, vectors in LDS, the graph's CSR rows read
// from L2; per iteration one sparse mat-vec and one workgroup reduction.
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

template <int THREADS>
__device__ __forceinline__ double ndp_block_sum(double v, double* s_red) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, WAVE);
  if constexpr (THREADS == 64) return v;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < THREADS / 64; ++w) t += s_red[w];
  return t;
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

// indptr / col / w: CSR over all nodes of the batch of a SYMMETRIC adjacency without self loops (the caller
// symmetrises with max, as to_undirected(reduce="max") does, ndp_select.py:198-202).
template <int THREADS>
__global__ __launch_bounds__(THREADS) void ndp_partition_kernel(const int32_t* __restrict__ indptr,
                                                                const int64_t* __restrict__ col,
                                                                const float* __restrict__ w,
                                                                const int64_t* __restrict__ graph_ptr,
                                                                unsigned long long seed, int max_iter, double tol,
                                                                uint8_t* __restrict__ keep, int32_t* __restrict__ info,
                                                                int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  __shared__ double s_red[THREADS / 64 + 1];
  const int g = blockIdx.x, tid = threadIdx.x;
  const int64_t p0 = graph_ptr[g], p1 = graph_ptr[g + 1];
  const int64_t n64 = p1 - p0;
  if (n64 <= 0) return;
  if (n64 > NDP_MAX_N) {
    if (tid == 0) atomicOr(status, 1);
    return;
  }
  const int n = static_cast<int>(n64);
  if (n == 1) {  // trivial case (ndp_select.py:223-225): the node is kept
    if (tid == 0) { keep[p0] = 1; info[g] = 0; }
    return;
  }
  double* x = s_dyn;
  double* y = s_dyn + n;
  float* dis = reinterpret_cast<float*>(s_dyn + 2 * n);
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
  int it = 0;
  bool random_part = !(vol > 0.0);
  if (!random_part) {
    for (; it < max_iter; ++it) {
      double sq = 0.0;
      for (int i = tid; i < n; i += THREADS) {
        double acc = 0.0;
        for (int e = indptr[p0 + i]; e < indptr[p0 + i + 1]; ++e) {
          const int64_t c = col[e];
          if (c < p0 || c >= p1) continue;
          const int j = static_cast<int>(c - p0);
          acc += (w ? static_cast<double>(w[e]) : 1.0) * static_cast<double>(dis[j]) * x[j];
        }
        const double v = x[i] - static_cast<double>(dis[i]) * acc;  // (Ls x)_i
        y[i] = v;
        sq += v * v;
      }
      const double nrm2 = ndp_block_sum<THREADS>(sq, s_red);
      if (!(nrm2 > 0.0)) { random_part = true; break; }  // x fell into the null space (cannot happen for vol > 0)
      const double inv = 1.0 / sqrt(nrm2);
      double diff = 0.0;
      __syncthreads();
      for (int i = tid; i < n; i += THREADS) {
        const double v = y[i] * inv;
        diff = fmax(diff, fabs(v - x[i]));
        x[i] = v;
      }
      const double dmax = ndp_block_max<THREADS>(diff, s_red);
      __syncthreads();
      if (dmax < tol) { ++it; break; }
    }
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
  (void)hipMemsetAsync(d_status, 0, sizeof(int), stream);
  if (N == 0 || B == 0) return check_launch("tgp_ndp_partition");
  TGP_REQUIRE(indptr && graph_ptr && keep && info && (col || nnz == 0), TGP_ERR_INVALID,
              "tgp_ndp_partition: null pointer");
  TGP_REQUIRE(N < (1ll << 31) && nnz < (1ll << 31) && B < (1ll << 31), TGP_ERR_RANGE,
              "tgp_ndp_partition: N / nnz / B >= 2^31");
  (void)hipMemsetAsync(keep, 0, static_cast<size_t>(N), stream);
  (void)hipMemsetAsync(info, 0, static_cast<size_t>(B) * sizeof(int32_t), stream);
  int64_t cap = max_graph_nodes < NDP_MAX_N ? max_graph_nodes : NDP_MAX_N;
  if (cap < 2) cap = 2;
  const size_t lds = static_cast<size_t>(cap) * (2 * sizeof(double) + sizeof(float)) + 16;
  if (cap <= 64)
    hipLaunchKernelGGL(ndp_partition_kernel<64>, dim3(static_cast<unsigned>(B)), dim3(64), lds, stream, indptr, col, w,
                       graph_ptr, static_cast<unsigned long long>(seed), max_iter, tol, keep, info, d_status);
  else
    hipLaunchKernelGGL(ndp_partition_kernel<256>, dim3(static_cast<unsigned>(B)), dim3(256), lds, stream, indptr, col,
                       w, graph_ptr, static_cast<unsigned long long>(seed), max_iter, tol, keep, info, d_status);
  return check_launch("tgp_ndp_partition");
}
