// A9: KronConnect (connect/kron_conn.py:117-165) as a block-batched Schur complement.
//
// The reference builds ONE sparse Laplacian for the whole batch and calls scipy's sparse LU on L[-,-].  L is block
// diagonal (one block per graph), so the Kron reduction of every graph is independent:
//     L'_g = L_g[+,+] - L_g[+,-] L_g[-,-]^-1 L_g[-,+]
// One workgroup per graph holds the graph's dense n x n Laplacian in fp64, ordered (dropped nodes, kept nodes), and
// eliminates the dropped nodes by Gaussian elimination without pivoting (L[-,-] is a principal block of a Laplacian:
// a symmetric, weakly diagonally dominant M-matrix, for which elimination in any order is stable); what is left in the
// trailing k x k block IS the Schur complement.  Graphs up to KRON_LDS_MAX_N nodes keep the matrix in LDS; larger ones
// (up to KRON_MAX_N) use a slab of the workspace and MANY workgroups (one launch per panel: kron_big_panel_kernel / kron_big_trail_kernel, the same
// arithmetic); beyond that the call declines
// (*d_count = -1) and the host keeps its library / scipy route for that batch.
//
// Update rule M[i][j] -= (M[i][p] * M[p][j]) * (1 / M[p][p]): the product commutes, so a symmetric L gives a bitwise
// symmetric result and the reference's "symmetrise if nearly symmetric" step (kron_conn.py:137-139) is the identity.
// Exactly singular L[-,-] (a component made of dropped nodes only): the block is redone with the reference's
// Marquardt-Levenberg damping 1e-6 I (kron_conn.py:131-135).
#include "primitives.h"
#include <mutex>
#include <utility>

namespace tgp {

// THE update of every elimination loop in this file: m - (c * u) / pivot as fma(-(c * u), 1 / pivot, m) -- the product is
// rounded, the rest is one rounding.  c * u commutes, so a symmetric L gives a bitwise symmetric result and the
// reference's "symmetrise if nearly symmetric" step (kron_conn.py:137-139) is the identity.  (r3: was (c * u) * inv and a
// subtraction; fp64 multiplies and adds issue at 8 cycles per wave here, the fused form is a third less work.)
__device__ __forceinline__ double kron_upd(double m, double c, double u, double inv) { return __fma_rn(-(c * u), inv, m); }

constexpr int KRON_LDS_MAX_N = 128;  // 128 x 129 doubles = 132 KB of the 160 KB LDS
constexpr int KRON_MAX_N = 8192;       // r5: was 4096 (r3: 1024); the multi-workgroup panel kernels do not care, the slab of a
                                       // graph is n^2 doubles: 537 MB at 8192 nodes, taken from the caller's workspace
constexpr int KRON_REDO_MAX_N = 1024;  // what the one-workgroup damped redo of a singular graph can hold in LDS
constexpr int KRON_THREADS = 256;
constexpr int KRON_BIG_THREADS = 1024;

enum KronStatus { KRON_UNSORTED = 1, KRON_BAD_INDEX = 2, KRON_TOO_LARGE = 4, KRON_CROSS_GRAPH = 8 };

__global__ __launch_bounds__(256) void kron_flags_kernel(const int64_t* __restrict__ node_index, int64_t k, int64_t n,
                                                         uint32_t* __restrict__ flags, int* __restrict__ status) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (j >= k) return;
  const int64_t v = node_index[j];
  if (v < 0 || v >= n) {
    atomicOr(status, KRON_BAD_INDEX);
    return;
  }
  flags[v] = 1u;
  // pooled id of a kept node = its position in node_index = its rank among the kept nodes only when ascending
  if (j > 0 && node_index[j - 1] >= v) atomicOr(status, KRON_UNSORTED);
}

__device__ __forceinline__ int64_t wave_incl_scan64(int64_t v) {
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    const int64_t t = __shfl_up(v, d, WAVE);
    if (lane_id() >= d) v += t;
  }
  return v;
}

// what the multi-workgroup kernels need to know about one of their graphs: ONE 40-byte record read per workgroup
// (graph_ptr -> rank -> big_off was a chain of three dependent global reads in front of every panel launch)
struct BigDesc {
  int g, n, k, m;
  int64_t p0;   // first node of the graph
  int64_t off;  // of its n x (n | 1) matrix in the scratch buffer
  uint32_t r0;  // rank of its first node = first pooled row
  int pad;
};

// per graph: offset of its k x k result in the dense buffer, offset of its n x (n|1) scratch matrix (graphs that do
// not fit LDS), and the size checks.  One 1024-thread workgroup.
__global__ __launch_bounds__(1024) void kron_plan_kernel(const int64_t* __restrict__ graph_ptr, int B,
                                                         const uint32_t* __restrict__ rank,
                                                         int64_t* __restrict__ sq_off, int64_t* __restrict__ big_off,
                                                         int64_t cap_dense, int64_t cap_big, int skip_oversize,
                                                         int lds_cap, int64_t declared_max,
                                                         BigDesc* __restrict__ big_desc,
                                                         int* __restrict__ big_count, int big_cap,
                                                         int* __restrict__ status) {
  __shared__ int64_t s_w[2][16];
  __shared__ int64_t s_carry[2];
  __shared__ int s_nbig;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid < 2) s_carry[tid] = 0;
  if (tid == 0) s_nbig = 0;
  __syncthreads();
  for (int base = 0; base < B; base += 1024) {
    const int g = base + tid;
    int64_t sq = 0, big = 0;
    int slot = -1;
    BigDesc desc{};
    if (g < B) {
      const int64_t p0 = graph_ptr[g], p1 = graph_ptr[g + 1];
      const int64_t n = p1 - p0;
      const int64_t k = static_cast<int64_t>(rank[p1]) - static_cast<int64_t>(rank[p0]);
      desc.g = g; desc.n = static_cast<int>(n); desc.k = static_cast<int>(k); desc.m = static_cast<int>(n - k);
      desc.p0 = p0; desc.r0 = rank[p0];
      // (with skip_oversize the caller's bound -- at most KRON_MAX_N -- is the size limit: larger graphs are the caller's)
      const bool oversize = n > (skip_oversize ? static_cast<int64_t>(declared_max) : static_cast<int64_t>(KRON_MAX_N));
      if (n < 0 || (oversize && !skip_oversize)) atomicOr(status, KRON_TOO_LARGE);
      if (!oversize) {  // (skipped graphs take no scratch and emit nothing: the caller reduces them itself)
        sq = k * k;
        if (n > lds_cap && k > 0) big = n * (n | 1);
        if (n > declared_max) atomicOr(status, KRON_TOO_LARGE);  // the caller's max_graph_nodes sized the launches
        if (n > lds_cap && k > 0) {  // reduced by the multi-workgroup kernels: any order in the list will do
          slot = atomicAdd(&s_nbig, 1);
          if (slot >= big_cap) { slot = -1; atomicOr(status, KRON_TOO_LARGE); }
        }
      }
    }
    const int64_t isq = wave_incl_scan64(sq), ibig = wave_incl_scan64(big);
    if (lane == WAVE - 1) { s_w[0][w] = isq; s_w[1][w] = ibig; }
    __syncthreads();
    int64_t osq = s_carry[0], obig = s_carry[1], tsq = 0, tbig = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j < w) { osq += s_w[0][j]; obig += s_w[1][j]; }
      tsq += s_w[0][j];
      tbig += s_w[1][j];
    }
    if (g < B) {
      sq_off[g] = osq + isq - sq;
      big_off[g] = obig + ibig - big;
      if (slot >= 0) {
        desc.off = obig + ibig - big;
        big_desc[slot] = desc;
      }
    }
    __syncthreads();
    if (tid == 0) { s_carry[0] += tsq; s_carry[1] += tbig; }
    __syncthreads();
  }
  if (tid == 0) {
    sq_off[B] = s_carry[0];
    big_off[B] = s_carry[1];
    *big_count = s_nbig < big_cap ? s_nbig : big_cap;
    if (s_carry[0] > cap_dense || s_carry[1] > cap_big) atomicOr(status, KRON_TOO_LARGE);
  }
}

struct KronArgs {
  const int32_t* indptr;   // [N+1] CSR row offsets of the entry list
  const int64_t* col;      // [nnz]
  const float* val32;      // one of val32 / val64, or neither (= ones)
  const double* val64;
  const int32_t* perm;     // optional indirection: CSR slot -> entry
  int from_adj;            // entries are adjacency weights: L = D - A is formed here (self loops skipped)
  const int64_t* graph_ptr;
  const uint32_t* rank;    // [N+1] exclusive prefix sums of the keep flags
  const int64_t* sq_off;
  const int64_t* big_off;
  float* dense;
  double* big;
  uint32_t* counts;
  int* status;
  double threshold;
  int lds_cap;             // graphs up to this many nodes use LDS
  int lds_lo;              // this launch of the LDS kernel takes graphs of lds_lo < n <= lds_hi nodes
  int lds_hi;
  const BigDesc* big_desc; // graphs beyond it (and up to KRON_MAX_N), kept nodes > 0
  const int* big_count;
  int* sing;               // [B] exactly singular L[-,-] met: the graph is redone with damping
  double* big_inv;         // [num_big][KRON_SNB] reciprocal pivots of the step in flight
  uint32_t* rowcnt;        // [num_kept] survivors per pooled row (big graphs only; NULL: not wanted)
};

template <int THREADS>
__device__ __forceinline__ void kron_build(const KronArgs& a, double* M, int ld, int64_t p0, int n, int m, uint32_t r0,
                                           double damp) {
  for (int e = threadIdx.x; e < n * ld; e += THREADS) M[e] = 0.0;
  __syncthreads();
  for (int t = threadIdx.x; t < n; t += THREADS) {
    const int64_t v = p0 + t;
    const uint32_t rv = a.rank[v] - r0;
    const bool keep_v = a.rank[v + 1] - a.rank[v] != 0;
    const int lr = keep_v ? m + static_cast<int>(rv) : t - static_cast<int>(rv);
    if (damp != 0.0 && !keep_v) atomicAdd(&M[lr * ld + lr], damp);
    for (int e = a.indptr[v]; e < a.indptr[v + 1]; ++e) {
      const int idx = a.perm ? a.perm[e] : e;
      const int64_t c = a.col[idx];
      if (c < p0 || c >= p0 + n) {  // an entry that couples two graphs: L is not block diagonal
        atomicOr(a.status, KRON_CROSS_GRAPH);
        continue;
      }
      const double val = a.val64 ? a.val64[idx] : (a.val32 ? static_cast<double>(a.val32[idx]) : 1.0);
      const uint32_t rc = a.rank[c] - r0;
      const bool keep_c = a.rank[c + 1] - a.rank[c] != 0;
      const int lc = keep_c ? m + static_cast<int>(rc) : static_cast<int>(c - p0) - static_cast<int>(rc);
      if (a.from_adj) {
        if (c != v) {
          atomicAdd(&M[lr * ld + lc], -val);
          atomicAdd(&M[lr * ld + lr], val);
        }
      } else {
        atomicAdd(&M[lr * ld + lc], val);
      }
    }
  }
  __syncthreads();
}

// eliminate pivots 0..m-1; returns (uniformly) whether an exactly zero pivot was met.  A wave takes KRON_RU rows of the
// pivot step at once, every read issued before the first update (r3: one row at a time behind a "multiplier != 0" branch
// made a pivot step a chain of dependent LDS round trips -- 94 us for 2048 graphs of 20..60 nodes; a zero multiplier now
// simply subtracts a zero, which changes nothing but, at most, the sign of a zero).
constexpr int KRON_RU = 4;
template <int THREADS>
__device__ __forceinline__ bool kron_eliminate(double* M, int ld, int n, int m, int* s_flag) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  constexpr int NW = THREADS / 64;
  for (int p = 0; p < m; ++p) {
    const double piv = M[p * ld + p];
    if (piv == 0.0 || !(piv == piv)) {
      if (threadIdx.x == 0) *s_flag = 1;
      // a zero pivot with a zero column below it contributes nothing; anything else is the singular case
    }
    const double inv = piv != 0.0 ? 1.0 / piv : 0.0;
    for (int i0 = p + 1 + w; i0 < n; i0 += NW * KRON_RU) {
      double c[KRON_RU];
      int row[KRON_RU];
#pragma unroll
      for (int t = 0; t < KRON_RU; ++t) {
        const int i = i0 + NW * t;
        row[t] = i < n ? i : p;  // (a row past the end re-reads the pivot row with a zero multiplier and is not stored)
        c[t] = i < n ? M[i * ld + p] : 0.0;
      }
      for (int j = p + 1 + lane; j < n; j += 64) {
        const double up = M[p * ld + j];
        double v[KRON_RU];
#pragma unroll
        for (int t = 0; t < KRON_RU; ++t) v[t] = M[row[t] * ld + j];
#pragma unroll
        for (int t = 0; t < KRON_RU; ++t)
          if (i0 + NW * t < n) M[row[t] * ld + j] = kron_upd(v[t], c[t], up, inv);
      }
    }
    __syncthreads();
  }
  const bool bad = *s_flag != 0;
  __syncthreads();
  return bad;
}

// The same elimination for matrices in global memory (graphs beyond the LDS capacity), KRON_NB pivots at a time: the
// pivot rows U [NB][n] and pivot columns C [n][NB] of a block are staged in LDS, factored there, and the trailing
// matrix is then read and written ONCE per block instead of once per pivot; every element still receives the updates
// of pivots 0, 1, ... in that order with the same (a * b) * (1 / pivot) arithmetic, so the result is bit-identical to
// the pivot-at-a-time loop above (1000-node graph: 38 ms -> a few ms; tests compare both against the oracle).
constexpr int KRON_NB = 8;
template <int THREADS>
__device__ __forceinline__ bool kron_eliminate_blocked(double* M, int ld, int n, int m, int* s_flag, double* pu,
                                                       double* pc, double* s_inv) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  constexpr int NW = THREADS / 64;
  for (int p0 = 0; p0 < m; p0 += KRON_NB) {
    const int nb = m - p0 < KRON_NB ? m - p0 : KRON_NB;
    const int rem = n - p0;  // rows / columns p0 .. n-1 take part
    // stage: U[q][j - p0] = M[p0 + q][j], C[(i - p0)][q] = M[i][p0 + q]
    for (int q = w; q < nb; q += NW)
      for (int j = lane; j < rem; j += 64) pu[q * n + j] = M[(p0 + q) * ld + p0 + j];
    for (int e = threadIdx.x; e < rem * nb; e += THREADS) {
      const int i = e / nb, q = e - i * nb;
      pc[i * KRON_NB + q] = M[(p0 + i) * ld + p0 + q];
    }
    __syncthreads();
    for (int q = 0; q < nb; ++q) {  // factor the panel
      const double piv = pu[q * n + q];
      if (threadIdx.x == 0) {
        if (piv == 0.0 || !(piv == piv)) *s_flag = 1;
        s_inv[q] = piv != 0.0 ? 1.0 / piv : 0.0;
      }
      const double inv = piv != 0.0 ? 1.0 / piv : 0.0;
      // rows r > q of U (columns beyond the pivot) and columns c > q of C (rows beyond the pivot)
      for (int r = q + 1 + w; r < nb; r += NW) {
        const double crq = pc[r * KRON_NB + q];
        if (crq != 0.0)
          for (int j = q + 1 + lane; j < rem; j += 64) pu[r * n + j] = kron_upd(pu[r * n + j], crq, pu[q * n + j], inv);
      }
      for (int e = threadIdx.x; e < (rem - q - 1) * (nb - q - 1); e += THREADS) {
        const int i = q + 1 + e / (nb - q - 1), c = q + 1 + e % (nb - q - 1);
        const double ciq = pc[i * KRON_NB + q];
        if (ciq != 0.0) pc[i * KRON_NB + c] = kron_upd(pc[i * KRON_NB + c], ciq, pu[q * n + c], inv);
      }
      __syncthreads();
    }
    // trailing update, one pass over M[p0 + nb .., p0 + nb ..]
    for (int i = nb + w; i < rem; i += NW) {
      double c[KRON_NB];
      bool any = false;
#pragma unroll
      for (int q = 0; q < KRON_NB; ++q) {
        c[q] = q < nb ? pc[i * KRON_NB + q] : 0.0;
        any = any || c[q] != 0.0;
      }
      if (!any) continue;
      double* row = M + static_cast<long>(p0 + i) * ld + p0;
      for (int j = nb + lane; j < rem; j += 64) {
        double v = row[j];
#pragma unroll
        for (int q = 0; q < KRON_NB; ++q)
          if (q < nb && c[q] != 0.0) v -= (c[q] * pu[q * n + j]) * s_inv[q];
        row[j] = v;
      }
    }
    __syncthreads();
  }
  const bool bad = *s_flag != 0;
  __syncthreads();
  return bad;
}

template <int THREADS, bool BLOCKED>
__device__ __forceinline__ void kron_graph(const KronArgs& a, double* M, int g, int64_t p0, int n, int k, uint32_t r0,
                                           int* s_flag, uint32_t* s_cnt, double* pu = nullptr, double* pc = nullptr,
                                           double* s_inv = nullptr) {
  const int m = n - k, ld = n | 1;
  if (threadIdx.x == 0) { *s_flag = 0; *s_cnt = 0; }
  auto eliminate = [&]() -> bool {
    if constexpr (BLOCKED) return kron_eliminate_blocked<THREADS>(M, ld, n, m, s_flag, pu, pc, s_inv);
    else return kron_eliminate<THREADS>(M, ld, n, m, s_flag);
  };
  kron_build<THREADS>(a, M, ld, p0, n, m, r0, 0.0);
  if (eliminate()) {
    if (threadIdx.x == 0) *s_flag = 0;
    kron_build<THREADS>(a, M, ld, p0, n, m, r0, 1e-6);  // Marquardt-Levenberg damping (kron_conn.py:131-135)
    (void)eliminate();
  }
  // A = -L', |A| > threshold, zero diagonal, explicit zeros dropped (kron_conn.py:141-146); 0.f marks "no edge"
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  constexpr int NW = THREADS / 64;
  float* out = a.dense + a.sq_off[g];
  uint32_t mine = 0;
  for (int i = w; i < k; i += NW) {
    uint32_t rowc = 0;
    for (int j0 = 0; j0 < k; j0 += 64) {
      const int j = j0 + lane;
      bool keep = false;
      float f = 0.f;
      if (j < k) {
        const double v = -M[(m + i) * ld + m + j];
        keep = i != j && v != 0.0 && (a.threshold > 0.0 ? fabs(v) > a.threshold : v == v);
        f = keep ? static_cast<float>(v) : 0.f;
        keep = keep && f != 0.f;
        out[static_cast<int64_t>(i) * k + j] = f;
      }
      rowc += __popcll(__ballot(keep));
    }
    mine += rowc;
    if (a.rowcnt && lane == 0) a.rowcnt[r0 + i] = rowc;
  }
  if (lane == 0 && mine) atomicAdd(s_cnt, mine);
  __syncthreads();
  if (threadIdx.x == 0) a.counts[g] = *s_cnt;
}

// LDS-resident graphs: one 256-thread workgroup per graph
__global__ __launch_bounds__(KRON_THREADS) void kron_schur_lds_kernel(KronArgs a) {
  extern __shared__ __attribute__((aligned(16))) double s_M[];
  __shared__ int s_flag;
  __shared__ uint32_t s_cnt;
  const int g = blockIdx.x;
  const int64_t p0 = a.graph_ptr[g], p1 = a.graph_ptr[g + 1];
  const int64_t n64 = p1 - p0;
  if (n64 <= a.lds_lo || n64 > a.lds_hi) {
    if (n64 <= 0 && a.lds_lo == 0 && threadIdx.x == 0) a.counts[g] = 0;
    return;  // empty, another size class of this kernel, or handled by the big-graph kernels
  }
  const uint32_t r0 = a.rank[p0];
  const int k = static_cast<int>(a.rank[p1] - r0);
  if (k == 0) {
    if (threadIdx.x == 0) a.counts[g] = 0;
    return;
  }
  kron_graph<KRON_THREADS, false>(a, s_M, g, p0, static_cast<int>(n64), k, r0, &s_flag, &s_cnt);
}

// ------------------------------------------------------------------------------------------------------------
// r3: graphs of 129 .. 1024 nodes on MANY workgroups.  One workgroup walking a 1000-node matrix through 62 panels
// pulled ~1 GB through a single CU: 10 ms for a batch whose 2048 small graphs take 0.25 ms (profiles/r02_kron.txt).
// The same blocked elimination is now one launch per panel of KRON_SNB pivots over every big graph of the batch, a
// 64 x 64 tile of the trailing matrix per workgroup: the tile's workgroup factors the panel's diagonal block and its
// own slices of the pivot rows U and pivot columns C redundantly in LDS, then applies the panel to its tile, each
// element receiving the pivots' updates in order with the same (c * u) * (1 / pivot) arithmetic as the single-
// workgroup loops above (bit-identical; the launch boundary is the only synchronisation: no spin, no fence).
// A graph with an exactly singular L[-,-] is flagged and redone (build + damped elimination) by the one-workgroup
// kernel; it is the rare case (a connected component made of dropped nodes only).
constexpr int KRON_SNB = 32;   // pivots per launch
constexpr int KRON_TILE = 64;

struct BigGraph {
  int g, n, k, m, ld;
  int64_t p0;
  uint32_t r0;
  double* M;
};

__device__ __forceinline__ bool kron_big_graph(const KronArgs& a, int b, BigGraph* out) {
  if (b >= *a.big_count) return false;
  const BigDesc d = a.big_desc[b];
  BigGraph q;
  q.g = d.g; q.p0 = d.p0; q.n = d.n; q.r0 = d.r0; q.k = d.k; q.m = d.m;
  q.ld = q.n | 1;
  q.M = a.big + d.off;
  *out = q;
  return true;
}

// M (zeroed by a memset) += the graph's Laplacian entries, ordered (dropped nodes, kept nodes): 256 nodes per workgroup
__global__ __launch_bounds__(256) void kron_big_build_kernel(KronArgs a) {
  BigGraph q;
  if (!kron_big_graph(a, blockIdx.y, &q)) return;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= q.n) return;
  const int64_t v = q.p0 + t;
  const uint32_t rv = a.rank[v] - q.r0;
  const bool keep_v = a.rank[v + 1] - a.rank[v] != 0;
  const int lr = keep_v ? q.m + static_cast<int>(rv) : t - static_cast<int>(rv);
  for (int e = a.indptr[v]; e < a.indptr[v + 1]; ++e) {
    const int idx = a.perm ? a.perm[e] : e;
    const int64_t c = a.col[idx];
    if (c < q.p0 || c >= q.p0 + q.n) {
      atomicOr(a.status, KRON_CROSS_GRAPH);
      continue;
    }
    const double val = a.val64 ? a.val64[idx] : (a.val32 ? static_cast<double>(a.val32[idx]) : 1.0);
    const uint32_t rc = a.rank[c] - q.r0;
    const bool keep_c = a.rank[c + 1] - a.rank[c] != 0;
    const int lc = keep_c ? q.m + static_cast<int>(rc) : static_cast<int>(c - q.p0) - static_cast<int>(rc);
    if (a.from_adj) {
      if (c != v) {
        atomicAdd(&q.M[static_cast<long>(lr) * q.ld + lc], -val);
        atomicAdd(&q.M[static_cast<long>(lr) * q.ld + lr], val);
      }
    } else {
      atomicAdd(&q.M[static_cast<long>(lr) * q.ld + lc], val);
    }
  }
}

// Panel kernel: workgroup (slice, graph) factors the diagonal block D of the step's KRON_SNB pivots (redundantly: it is
// 32 x 32) and ITS slice of KRON_PW columns of the pivot rows U and KRON_PW rows of the pivot columns C, and writes the
// updated slices back in place; slice 0 also stores the reciprocal pivots.
//   1. the workgroup stages D and its C rows in LDS with coalesced reads (a C row is 256 contiguous bytes: one thread
//      per row is 64 cache lines per load instruction, the same in the write-back: 4.6 us of stores per launch); every
//      thread then holds ONE line in registers: thread t < 128 column S0 + t of U, thread 128 + t row S0 + t of C;
//   2. wave 0 factors D with no barrier and no LDS traffic: lane r owns row r (32 registers), the pivot row reaches the
//      other lanes through v_readlane, the next reciprocal is started as soon as its pivot is final; the factored block
//      is then published twice, sUp[p][r] = D[p][r] and sLo[p][r] = D[r][p] (as they stood at pivot p for r > p; the
//      other triangle of each copy is never read);
//   3. ONE barrier, then every thread runs the forward substitution of its own line out of registers (the U columns
//      against sLo, the C rows against sUp: the same code): x[r] -= (tab[p][r] * x[p]) * inv_p, p ascending.
// Every element receives the pivots' updates in order with the (c * u) * (1 / pivot) arithmetic of the one-workgroup
// loops.  (No "multiplier != 0" test: (0 * u) * inv is a zero unless a pivot was singular, and such a graph is redone;
// x - 0 differs from x at most in the sign of a zero, which nothing downstream looks at.  Rows at or above the pivot
// are frozen by zeroing their multiplier, the same argument; and a last panel of fewer than 32 pivots is padded with
// identity pivots and zero lines, for which every update subtracts a zero.)  The updates are written in groups of 8 independent
// products: left to itself the compiler chained mul -> mul -> sub through one temporary, 42 cycles per element.
// Measured r3 on the 1000-node graph of profiles/r03_kron.txt: one barrier per pivot with the slices spread over the
// workgroup's registers took ~37 us per launch (0.9 us per pivot).
constexpr int KRON_PW = 128;
constexpr int KRON_CH = 8;

__device__ __forceinline__ double kron_bcast(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

// one pivot of the D factorisation / of a line's forward substitution, the pivot a template parameter: with a run-time
// p the compiler declined to unroll the 32 steps, indexed the register arrays dynamically and put them in scratch
template <int P>
__device__ __forceinline__ void kron_dfac_step(double (&d)[KRON_SNB], int row, int lane, double& inv, bool& bad,
                                               double* s_inv) {
  if (lane == 0) s_inv[P] = inv;
  const double crp = row > P ? d[P] : 0.0;  // rows at or above the pivot are final: a zero multiplier freezes them
  double inv_next = 0.0;
  if constexpr (P + 1 < KRON_SNB) {  // the next pivot first: its reciprocal is taken while the other columns are updated
    const double up1 = kron_bcast(d[P + 1], P);
    d[P + 1] = kron_upd(d[P + 1], crp, up1, inv);
    const double pn = kron_bcast(d[P + 1], P + 1);
    inv_next = pn != 0.0 ? 1.0 / pn : 0.0;
    bad = bad || pn == 0.0 || !(pn == pn);
  }
#pragma unroll
  for (int c0 = P + 2; c0 < KRON_SNB; c0 += KRON_CH) {
    double t[KRON_CH];
#pragma unroll
    for (int i = 0; i < KRON_CH; ++i)
      if (c0 + i < KRON_SNB) t[i] = kron_bcast(d[c0 + i], P);
#pragma unroll
    for (int i = 0; i < KRON_CH; ++i)
      if (c0 + i < KRON_SNB) t[i] = crp * t[i];
#pragma unroll
    for (int i = 0; i < KRON_CH; ++i)
      if (c0 + i < KRON_SNB) d[c0 + i] = __fma_rn(-t[i], inv, d[c0 + i]);
  }
  inv = inv_next;
}

template <int P>
__device__ __forceinline__ void kron_subst_step(double (&x)[KRON_SNB], const double* tab, const double* s_inv) {
  const double t = x[P], ip = s_inv[P];
#pragma unroll
  for (int r0 = P + 1; r0 < KRON_SNB; r0 += KRON_CH) {
    double v[KRON_CH];
#pragma unroll
    for (int i = 0; i < KRON_CH; ++i)
      if (r0 + i < KRON_SNB) v[i] = tab[P * KRON_SNB + r0 + i];
#pragma unroll
    for (int i = 0; i < KRON_CH; ++i)
      if (r0 + i < KRON_SNB) v[i] = v[i] * t;
#pragma unroll
    for (int i = 0; i < KRON_CH; ++i)
      if (r0 + i < KRON_SNB) x[r0 + i] = __fma_rn(-v[i], ip, x[r0 + i]);
  }
}

template <int... P>
__device__ __forceinline__ void kron_dfac_all(double (&d)[KRON_SNB], int row, int lane, double& inv, bool& bad,
                                              double* s_inv, std::integer_sequence<int, P...>) {
  (kron_dfac_step<P>(d, row, lane, inv, bad, s_inv), ...);
}

template <int... P>
__device__ __forceinline__ void kron_subst_all(double (&x)[KRON_SNB], const double* tab, const double* s_inv,
                                               std::integer_sequence<int, P...>) {
  (kron_subst_step<P>(x, tab, s_inv), ...);
}

// diagnostic build (-DKRON_STAMPS, never shipped): cycle stamps of the panel kernel's phases, printed by one workgroup
#ifdef KRON_STAMPS
#define KRON_STAMP(i) st[i] = clock64()
#define KRON_STAMP_PRINT()                                                                                            \
  if (q.n == 1000 && slice == 0 && (tid == 0 || tid == 192) && p0 == 0)                                               \
  printf("panel tid %d cycles: issue-loads %lld wait+bar %lld read-lds %lld dfac %lld publish %lld bar %lld subst %lld " \
         "to-lds %lld bar %lld store %lld\n", tid, st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], \
         0ll, st[6] - st[5], st[7] - st[6], st[8] - st[7], st[9] - st[8])
#else
#define KRON_STAMP(i)
#define KRON_STAMP_PRINT()
#endif

__device__ __forceinline__ void kron_panel_body(const KronArgs& a, const BigGraph& q, int p0, int nb, int slice,
                                                int graph_slot, double (*sT)[KRON_SNB + 1],
                                                double (*sDin)[KRON_SNB + 1], double* sF, double* s_inv) {
  const int T0 = p0 + nb;
  const int S0 = T0 + slice * KRON_PW;  // this slice: columns S0.. of U, rows S0.. of C
  const int tid = threadIdx.x, lane = tid & 63;
  const long ld = q.ld;
  double* M = q.M;
  const bool is_u = tid < KRON_PW;
  const int li = tid & (KRON_PW - 1);
  const int line = S0 + li;
  const bool live = line < q.n;
  double x[KRON_SNB];
#ifdef KRON_STAMPS
  long long st[10] = {};
#endif
  KRON_STAMP(0);
  // every load is issued unconditionally from a clamped address and masked afterwards: a branch per guarded load made
  // the compiler wait for each one in turn (7 us of this kernel's 22)
  const int lc = line < q.n ? line : q.n - 1;
  if (is_u) {
#pragma unroll
    for (int r = 0; r < KRON_SNB; ++r) x[r] = M[(p0 + (r < nb ? r : nb - 1)) * ld + lc];
#pragma unroll
    for (int r = 0; r < KRON_SNB; ++r) x[r] = (live && r < nb) ? x[r] : 0.0;
  }
  {
    double tv[KRON_PW * KRON_SNB / 256], dv[KRON_SNB * KRON_SNB / 256];
    const int cc = tid & 31, cq = cc < nb ? cc : nb - 1;
#pragma unroll
    for (int i = 0; i < KRON_PW * KRON_SNB / 256; ++i) {
      const int rr = (tid >> 5) + 8 * i, gr = S0 + rr < q.n ? S0 + rr : q.n - 1;
      tv[i] = M[static_cast<long>(gr) * ld + p0 + cq];
    }
#pragma unroll
    for (int i = 0; i < KRON_SNB * KRON_SNB / 256; ++i) {
      const int rr = (tid >> 5) + 8 * i;
      dv[i] = M[(p0 + (rr < nb ? rr : nb - 1)) * ld + p0 + cq];
    }
#pragma unroll
    for (int i = 0; i < KRON_PW * KRON_SNB / 256; ++i) {
      const int rr = (tid >> 5) + 8 * i;
      sT[rr][cc] = (S0 + rr < q.n && cc < nb) ? tv[i] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < KRON_SNB * KRON_SNB / 256; ++i) {
      const int rr = (tid >> 5) + 8 * i;
      sDin[rr][cc] = (rr < nb && cc < nb) ? dv[i] : (rr == cc ? 1.0 : 0.0);
    }
  }
  KRON_STAMP(1);
  __syncthreads();
  KRON_STAMP(2);
  if (!is_u) {
#pragma unroll
    for (int r = 0; r < KRON_SNB; ++r) x[r] = sT[li][r];
  }
  if (tid < 64) {
    double d[KRON_SNB];
    const int row = lane & 31;  // (lanes 32..63 mirror lanes 0..31: their results are never read)
#pragma unroll
    for (int c = 0; c < KRON_SNB; ++c) d[c] = sDin[row][c];
    double piv = kron_bcast(d[0], 0);
    double inv = piv != 0.0 ? 1.0 / piv : 0.0;
    bool bad = piv == 0.0 || !(piv == piv);
    KRON_STAMP(3);
    kron_dfac_all(d, row, lane, inv, bad, s_inv, std::make_integer_sequence<int, KRON_SNB>{});
    KRON_STAMP(4);
    if (lane < KRON_SNB) {
#pragma unroll
      for (int c = 0; c < KRON_SNB; ++c) {
        sF[KRON_SNB * KRON_SNB + row * KRON_SNB + c] = d[c];  // sUp
        sF[c * KRON_SNB + row] = d[c];                        // sLo
      }
    }
    if (slice == 0 && lane == 0 && bad) a.sing[q.g] = 1;
  }
  __syncthreads();
  KRON_STAMP(5);
  {
    const double* tab = sF + __builtin_amdgcn_readfirstlane(is_u ? 0 : KRON_SNB * KRON_SNB);
    kron_subst_all(x, tab, s_inv, std::make_integer_sequence<int, KRON_SNB>{});
  }
  KRON_STAMP(6);
  // the factored slices go back in place: the trailing kernel (next launch) reads them
  if (is_u) {
    if (live) {
#pragma unroll
      for (int r = 0; r < KRON_SNB; ++r)
        if (r < nb) M[(p0 + r) * ld + line] = x[r];
    }
  } else {
#pragma unroll
    for (int r = 0; r < KRON_SNB; ++r) sT[li][r] = x[r];
  }
  KRON_STAMP(7);
  __syncthreads();
  KRON_STAMP(8);
#pragma unroll
  for (int i = 0; i < KRON_PW * KRON_SNB / 256; ++i) {
    const int e = tid + 256 * i, rr = e >> 5, cc = e & 31;
    if (S0 + rr < q.n && cc < nb) M[static_cast<long>(S0 + rr) * ld + p0 + cc] = sT[rr][cc];
  }
  KRON_STAMP(9);
  KRON_STAMP_PRINT();
  if (slice == 0 && tid < nb) a.big_inv[static_cast<long>(graph_slot) * KRON_SNB + tid] = s_inv[tid];
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void kron_big_panel_kernel(KronArgs a,
                                                                                                       int step) {
  __shared__ double sT[KRON_PW][KRON_SNB + 1];
  __shared__ double sDin[KRON_SNB][KRON_SNB + 1];
  __shared__ double sF[2 * KRON_SNB * KRON_SNB];  // sLo | sUp
  __shared__ double s_inv[KRON_SNB];
  BigGraph q;
  if (!kron_big_graph(a, blockIdx.y, &q)) return;
  const int p0 = step * KRON_SNB;
  if (p0 >= q.m) return;
  const int nb = q.m - p0 < KRON_SNB ? q.m - p0 : KRON_SNB;
  if (p0 + nb + static_cast<int>(blockIdx.x) * KRON_PW >= q.n) return;
  kron_panel_body(a, q, p0, nb, blockIdx.x, blockIdx.y, sT, sDin, sF, s_inv);
}

// Trailing kernel: one 64 x 64 tile per workgroup, M[i][j] = kron_upd(M[i][j], C[i][p], U[p][j], inv_p) over the step's
// pivots in order (the arithmetic of the single-workgroup loops).  16 x 16 threads with a 4 x 4 register tile each (rows
// ti + 16 a, columns tj + 16 b): a pivot costs 8 LDS reads for 16 updates (one column per thread: 17 reads for 16, and the
// LDS reads were a third of the loop).  Loads are issued from clamped addresses and masked afterwards (a branch per
// guarded load serialised them).  (64 x 128 tiles measured slower in r3: 16 vs 11.6 us per launch.)
__global__ __launch_bounds__(256) void kron_big_trail_kernel(KronArgs a, int step) {
  __shared__ double sU[KRON_SNB][KRON_TILE];
  __shared__ double sC[KRON_TILE][KRON_SNB + 1];
  __shared__ double s_inv[KRON_SNB];
  BigGraph q;
  if (!kron_big_graph(a, blockIdx.y, &q)) return;
  const int p0 = step * KRON_SNB;
  if (p0 >= q.m) return;
  const int nb = q.m - p0 < KRON_SNB ? q.m - p0 : KRON_SNB;
  const int T0 = p0 + nb, rem = q.n - T0;
  const int nt = (rem + KRON_TILE - 1) / KRON_TILE;
  const int tile = blockIdx.x;
  if (tile >= nt * nt) return;
  const int R0 = T0 + (tile / nt) * KRON_TILE, C0 = T0 + (tile % nt) * KRON_TILE;
  const int rows = q.n - R0 < KRON_TILE ? q.n - R0 : KRON_TILE, cols = q.n - C0 < KRON_TILE ? q.n - C0 : KRON_TILE;
  const int tid = threadIdx.x;
  const long ld = q.ld;
  double* M = q.M;
  const int tj = tid & 15, ti = tid >> 4;
#ifdef KRON_STAMPS
  long long tt[5] = {};
  tt[0] = clock64();
#endif
  double v[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    const int i = ti + 16 * x, gi = R0 + (i < rows ? i : rows - 1);
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      const int j = tj + 16 * y;
      v[x][y] = M[gi * ld + C0 + (j < cols ? j : cols - 1)];
    }
  }
  {
    double uv[KRON_SNB * KRON_TILE / 256], cv[KRON_TILE * KRON_SNB / 256];
    const int uj = tid & 63, cc = tid & 31;
#pragma unroll
    for (int x = 0; x < KRON_SNB * KRON_TILE / 256; ++x) {
      const int r = (tid >> 6) + 4 * x;
      uv[x] = M[(p0 + (r < nb ? r : nb - 1)) * ld + C0 + (uj < cols ? uj : cols - 1)];
    }
#pragma unroll
    for (int x = 0; x < KRON_TILE * KRON_SNB / 256; ++x) {  // 32 lanes read 256 contiguous bytes of a C row
      const int i = (tid >> 5) + 8 * x;
      cv[x] = M[(R0 + (i < rows ? i : rows - 1)) * ld + p0 + (cc < nb ? cc : nb - 1)];
    }
#pragma unroll
    for (int x = 0; x < KRON_SNB * KRON_TILE / 256; ++x) {
      const int r = (tid >> 6) + 4 * x;
      sU[r][uj] = (r < nb && uj < cols) ? uv[x] : 0.0;
    }
#pragma unroll
    for (int x = 0; x < KRON_TILE * KRON_SNB / 256; ++x) {
      const int i = (tid >> 5) + 8 * x;
      sC[i][cc] = (i < rows && cc < nb) ? cv[x] : 0.0;
    }
  }
  if (tid < KRON_SNB) s_inv[tid] = a.big_inv[static_cast<long>(blockIdx.y) * KRON_SNB + tid];
  __syncthreads();
#ifdef KRON_STAMPS
  tt[1] = clock64();
#endif
  // (no c != 0 test here: (0 * u) * inv is 0 unless a pivot was singular, and such a graph is redone anyway; v - 0
  //  differs from v at most in the sign of a zero, which no later comparison sees)
#pragma unroll 2
  for (int p = 0; p < nb; ++p) {
    const double inv = s_inv[p];
    double cu[4], uu[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) cu[x] = sC[ti + 16 * x][p];
#pragma unroll
    for (int y = 0; y < 4; ++y) uu[y] = sU[p][tj + 16 * y];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int y = 0; y < 4; ++y) v[x][y] = kron_upd(v[x][y], cu[x], uu[y], inv);
  }
#ifdef KRON_STAMPS
  tt[2] = clock64();
#endif
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    const int i = ti + 16 * x;
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      const int j = tj + 16 * y;
      if (i < rows && j < cols) M[(R0 + i) * ld + C0 + j] = v[x][y];
    }
  }
#ifdef KRON_STAMPS
  tt[3] = clock64();
  if (q.n == 1000 && (tile == 0 || tile == 200) && tid == 0 && p0 == 0)
    printf("trail tile %d cycles: load+lds %lld compute %lld store %lld\n", tile, tt[1] - tt[0], tt[2] - tt[1], tt[3] - tt[2]);
#endif
}

// flagged graphs only: build + elimination with the reference's damping, threshold and count, on one workgroup
__global__ __launch_bounds__(KRON_BIG_THREADS) void kron_big_redo_kernel(KronArgs a) {
  extern __shared__ __attribute__((aligned(16))) double s_panel[];  // U [NB][n] | C [n][NB]
  __shared__ int s_flag;
  __shared__ uint32_t s_cnt;
  __shared__ double s_inv[KRON_NB];
  BigGraph q;
  if (!kron_big_graph(a, blockIdx.x, &q)) return;
  if (a.sing[q.g] == 0) return;
  if (q.n > KRON_REDO_MAX_N) {  // a singular graph beyond this kernel's LDS panels: decline, the caller has its own route
    if (threadIdx.x == 0) atomicOr(a.status, KRON_TOO_LARGE);
    return;
  }
  kron_graph<KRON_BIG_THREADS, true>(a, q.M, q.g, q.p0, q.n, q.k, q.r0, &s_flag, &s_cnt, s_panel,
                                     s_panel + static_cast<size_t>(KRON_NB) * q.n, s_inv);
}

// A = -L', |A| > threshold, zero diagonal, fp32 (kron_conn.py:141-146) for the graphs the step kernels reduced:
// 16 rows of the k x k block per workgroup
__global__ __launch_bounds__(256) void kron_big_finish_kernel(KronArgs a) {
  BigGraph q;
  if (!kron_big_graph(a, blockIdx.y, &q)) return;
  if (a.sing[q.g] != 0) return;  // redone (and counted) by kron_big_redo_kernel
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* out = a.dense + a.sq_off[q.g];
  uint32_t mine = 0;
  const int i1 = (blockIdx.x + 1) * 16 < q.k ? (blockIdx.x + 1) * 16 : q.k;
  for (int i = blockIdx.x * 16 + w; i < i1; i += 4) {
    uint32_t rowc = 0;
    for (int j0 = 0; j0 < q.k; j0 += 64) {
      const int j = j0 + lane;
      bool keep = false;
      if (j < q.k) {
        const double v = -q.M[static_cast<long>(q.m + i) * q.ld + q.m + j];
        keep = i != j && v != 0.0 && (a.threshold > 0.0 ? fabs(v) > a.threshold : v == v);
        const float f = keep ? static_cast<float>(v) : 0.f;
        keep = keep && f != 0.f;
        out[static_cast<int64_t>(i) * q.k + j] = f;
      }
      rowc += __popcll(__ballot(keep));
    }
    mine += rowc;
    if (lane == 0) a.rowcnt[q.r0 + i] = rowc;
  }
  if (lane == 0 && mine) atomicAdd(&a.counts[q.g], mine);
}

// edge list of the surviving entries in (graph, row, col) order = row-major order of the whole pooled batch
__global__ __launch_bounds__(KRON_THREADS) void kron_fill_kernel(const int64_t* __restrict__ graph_ptr,
                                                                 const uint32_t* __restrict__ rank,
                                                                 const int64_t* __restrict__ sq_off,
                                                                 const float* __restrict__ dense,
                                                                 const uint32_t* __restrict__ out_off, int lds_cap,
                                                                 int64_t* __restrict__ out_row,
                                                                 int64_t* __restrict__ out_col,
                                                                 float* __restrict__ out_w) {
  __shared__ uint32_t s_row[KRON_LDS_MAX_N + 1];
  const int g = blockIdx.x;
  const int64_t p0 = graph_ptr[g], p1 = graph_ptr[g + 1];
  if (p1 <= p0 || p1 - p0 > lds_cap) return;  // (graphs beyond the LDS capacity: kron_big_fill_kernel; oversize: no block)
  const uint32_t r0 = rank[p0];
  const int k = static_cast<int>(rank[p1] - r0);
  if (k == 0) return;
  const float* d = dense + sq_off[g];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = w; i < k; i += KRON_THREADS / 64) {
    uint32_t c = 0;
    for (int j0 = 0; j0 < k; j0 += 64) {
      const int j = j0 + lane;
      c += __popcll(__ballot(j < k && d[static_cast<int64_t>(i) * k + j] != 0.f));
    }
    if (lane == 0) s_row[i] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // k <= 1024 row counts: a serial scan is a few hundred cycles
    uint32_t run = out_off[g];
    for (int i = 0; i < k; ++i) {
      const uint32_t c = s_row[i];
      s_row[i] = run;
      run += c;
    }
  }
  __syncthreads();
  for (int i = w; i < k; i += KRON_THREADS / 64) {
    uint32_t pos = s_row[i];
    for (int j0 = 0; j0 < k; j0 += 64) {
      const int j = j0 + lane;
      const float v = j < k ? d[static_cast<int64_t>(i) * k + j] : 0.f;
      const unsigned long long mk = __ballot(v != 0.f);
      if (v != 0.f) {
        const uint32_t q = pos + __popcll(mk & lanemask_lt());
        out_row[q] = static_cast<int64_t>(r0) + i;
        out_col[q] = static_cast<int64_t>(r0) + j;
        out_w[q] = v;
      }
      pos += __popcll(mk);
    }
  }
}

// The same for a graph of 129 .. 1024 nodes, 16 rows of its k x k block per workgroup (one workgroup walking 500 rows
// took 0.58 ms): the rows' survivor counts come from the finish pass, the workgroup adds up those in front of its own
// block, then its four waves emit four rows each.
__global__ __launch_bounds__(256) void kron_big_fill_kernel(KronArgs a, const uint32_t* __restrict__ out_off,
                                                            int64_t* __restrict__ out_row,
                                                            int64_t* __restrict__ out_col,
                                                            float* __restrict__ out_w) {
  __shared__ uint32_t s_w[4];
  __shared__ uint32_t s_rowcnt[16];
  BigGraph q;
  if (!kron_big_graph(a, blockIdx.y, &q)) return;
  const int i0 = blockIdx.x * 16;
  if (i0 >= q.k) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* d = a.dense + a.sq_off[q.g];
  uint32_t before = 0;  // survivors of the graph's rows in front of this block (per-row counts from the finish pass)
  for (int i = threadIdx.x; i < i0; i += 256) before += a.rowcnt[q.r0 + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
  if (lane == 0) s_w[w] = before;
  if (threadIdx.x < 16) s_rowcnt[threadIdx.x] = i0 + threadIdx.x < q.k ? a.rowcnt[q.r0 + i0 + threadIdx.x] : 0u;
  __syncthreads();
  uint32_t base = out_off[q.g] + s_w[0] + s_w[1] + s_w[2] + s_w[3];
  for (int rr = w; rr < 16; rr += 4) {
    const int i = i0 + rr;
    if (i >= q.k) break;
    uint32_t pos = base;
    for (int x = 0; x < rr; ++x) pos += s_rowcnt[x];
    for (int j0 = 0; j0 < q.k; j0 += 64) {
      const int j = j0 + lane;
      const float v = j < q.k ? d[static_cast<int64_t>(i) * q.k + j] : 0.f;
      const unsigned long long mk = __ballot(v != 0.f);
      if (v != 0.f) {
        const uint32_t o = pos + __popcll(mk & lanemask_lt());
        out_row[o] = static_cast<int64_t>(q.r0) + i;
        out_col[o] = static_cast<int64_t>(q.r0) + j;
        out_w[o] = v;
      }
      pos += __popcll(mk);
    }
  }
}

struct KronWs {
  uint32_t* flags;   // [N+2] keep flags
  uint32_t* rank;    // [N+2] exclusive prefix sums
  int64_t* scan_total;
  int* status;
  int64_t* sq_off;   // [B+1]
  int64_t* big_off;  // [B+1]
  uint32_t* counts;  // [B]
  uint32_t* out_off; // [B]
  BigDesc* big_desc; // [B] graphs beyond the LDS capacity
  int* big_count;
  int* sing;         // [B]
  double* big_inv;   // [B][32]
  uint32_t* scan_tiles;  // multi-block scan scratch
  float* dense;
  double* big;
  int64_t cap_dense, cap_big;
};

// Sizes of the two big buffers: the caller's exact figures when it has them (sum over graphs of n_g^2 bounds the
// k_g x k_g results; sum of n_g * (n_g | 1) over the graphs beyond the LDS capacity is the scratch), else the
// worst case from the longest graph alone (N * max_nodes: one 1000-node graph among 80 k nodes of small ones made that
// 1 GB where 12 MB are needed).  The plan kernel checks the real totals against them and declines when they do not fit.
static void kron_caps(int64_t N, int64_t max_nodes, int64_t want_dense, int64_t want_big, int64_t* cap_dense,
                      int64_t* cap_big) {
  *cap_dense = want_dense >= 0 ? want_dense : N * (max_nodes > 0 ? max_nodes : 1);
  *cap_big = want_big >= 0 ? want_big : (max_nodes > KRON_LDS_MAX_N ? N * ((max_nodes | 1) + 1) : 0);
}

static size_t kron_layout(void* ws, int64_t N, int64_t B, int64_t max_nodes, int64_t want_dense, int64_t want_big,
                          KronWs* out) {
  Carver c(ws);
  KronWs s;
  kron_caps(N, max_nodes, want_dense, want_big, &s.cap_dense, &s.cap_big);
  // flags | counts | status sit next to each other: ONE memset zeroes the three (r6: they were three launches)
  s.flags = c.take<uint32_t>(N + 2);
  s.counts = c.take<uint32_t>(B + 1);
  s.status = c.take<int>(1);
  s.rank = c.take<uint32_t>(N + 2);
  s.scan_total = c.take<int64_t>(1);
  s.sq_off = c.take<int64_t>(B + 1);
  s.big_off = c.take<int64_t>(B + 1);
  s.out_off = c.take<uint32_t>(B + 1);
  s.big_desc = c.take<BigDesc>(B + 1);
  s.big_count = c.take<int>(4);
  s.sing = c.take<int>(B + 1);
  s.big_inv = c.take<double>(static_cast<size_t>(B + 1) * 32);
  s.scan_tiles = c.take<uint32_t>(2 * static_cast<size_t>(cdiv(N + 2, SCAN_TILE)) + 16);
  s.dense = c.take<float>(s.cap_dense > 0 ? s.cap_dense : 1);
  s.big = c.take<double>(s.cap_big > 0 ? s.cap_big : 1);
  if (out) *out = s;
  return c.off;
}

// A second stream per device for the LDS-resident graphs of a batch that also holds mid-size graphs: the panel chain of
// the latter is ~50 dependent launches that keep a few dozen CUs busy, the 2048 small graphs of a PROTEINS-like batch
// are 100 us of work for the whole chip -- run one beside the other (fork after the plan kernel, join before the
// counts are scanned; a plain fork / join of events, so a captured caller stream captures the side stream with it).
struct KronSide {
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  bool ok = false, tried = false;
};
static KronSide g_kron_side[64];
static std::mutex g_kron_side_mutex;

static KronSide* kron_side() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  KronSide& k = g_kron_side[dev];
  if (!k.tried) {
    k.tried = true;
    k.ok = hipStreamCreateWithFlags(&k.stream, hipStreamNonBlocking) == hipSuccess &&
           hipEventCreateWithFlags(&k.fork, hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&k.join, hipEventDisableTiming) == hipSuccess;
  }
  return k.ok ? &k : nullptr;
}

}  // namespace tgp

using namespace tgp;

extern "C" size_t tgp_kron_batched_workspace_bytes(int64_t N, int64_t B, int64_t max_graph_nodes, int64_t cap_dense,
                                                   int64_t cap_big) {
  if (N < 0 || B < 0 || max_graph_nodes < 0) return 0;
  return kron_layout(nullptr, N, B, max_graph_nodes, cap_dense, cap_big, nullptr) + 256;
}

extern "C" int tgp_kron_batched_max_graph_nodes(void) { return KRON_MAX_N; }

extern "C" int tgp_kron_batched_count(const int32_t* indptr, const int64_t* col, const float* val32,
                                      const double* val64, const int32_t* perm, int from_adjacency, int64_t N,
                                      int64_t nnz, const int64_t* graph_ptr, int64_t B, int64_t max_graph_nodes,
                                      int64_t cap_dense, int64_t cap_big, int64_t num_big,
                                      const int64_t* node_index, int64_t num_kept, double threshold,
                                      const uint32_t* node_rank, void* ws, size_t ws_bytes, int64_t* d_count,
                                      void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(N >= 0 && B >= 0 && nnz >= 0 && num_kept >= 0 && d_count && graph_ptr && indptr && (col || nnz == 0) &&
                  (node_index || num_kept == 0),
              TGP_ERR_INVALID, "tgp_kron_batched_count: bad argument");
  TGP_REQUIRE(!(val32 && val64), TGP_ERR_INVALID, "tgp_kron_batched_count: give fp32 or fp64 values, not both");
  TGP_REQUIRE(N < (1ll << 31) - 2 && nnz < (1ll << 31) && B < (1ll << 31), TGP_ERR_RANGE,
              "tgp_kron_batched_count: N / nnz / B >= 2^31");
  TGP_REQUIRE(ws && ws_bytes >= tgp_kron_batched_workspace_bytes(N, B, max_graph_nodes, cap_dense, cap_big),
              TGP_ERR_WORKSPACE, "tgp_kron_batched_count: workspace too small");
  KronWs s;
  kron_layout(ws, N, B, max_graph_nodes, cap_dense, cap_big, &s);
  if (num_big < 0 || num_big > B) num_big = B;
  // r6, node_rank [N + 1] (optional): the exclusive prefix sums of the keep flags, when the caller already has them (a
  // selector that compacted the kept nodes itself: tgp_mask_index_fill) -- the flags scatter and the two scan launches
  // fall away, and only the counters behind the flags are cleared
  uint32_t* const clear_from = node_rank ? s.counts : s.flags;
  (void)hipMemsetAsync(clear_from, 0, reinterpret_cast<char*>(s.status + 1) - reinterpret_cast<char*>(clear_from), stream);
  if (B == 0 || N == 0) {  // (otherwise the count scan at the end of this call writes *d_count)
    (void)hipMemsetAsync(d_count, 0, sizeof(int64_t), stream);
    return check_launch("tgp_kron_batched_count");
  }
  if (node_rank) {
    s.rank = const_cast<uint32_t*>(node_rank);  // (read-only from here on; tgp_kron_batched_fill is handed the same table)
  } else {
    if (num_kept > 0)
      hipLaunchKernelGGL(kron_flags_kernel, dim3(cdiv(num_kept, 256)), dim3(256), 0, stream, node_index, num_kept, N,
                         s.flags, s.status);
    device_scan_u32(s.flags, N + 1, s.rank, s.scan_total, s.scan_tiles, stream);  // (one workgroup took 73 us at N = 82 k)
  }
  const int cap = static_cast<int>(max_graph_nodes < KRON_LDS_MAX_N ? max_graph_nodes : KRON_LDS_MAX_N);
  hipLaunchKernelGGL(kron_plan_kernel, dim3(1), dim3(1024), 0, stream, graph_ptr, static_cast<int>(B), s.rank, s.sq_off,
                     s.big_off, s.cap_dense, s.cap_big, (from_adjacency & 2) ? 1 : 0, cap, max_graph_nodes, s.big_desc, s.big_count,
                     static_cast<int>(num_big), s.status);
  KronArgs a{};
  a.indptr = indptr; a.col = col; a.val32 = val32; a.val64 = val64; a.perm = perm; a.from_adj = from_adjacency & 1;
  a.graph_ptr = graph_ptr; a.rank = s.rank; a.sq_off = s.sq_off; a.big_off = s.big_off; a.dense = s.dense;
  a.big = s.big; a.counts = s.counts; a.status = s.status; a.threshold = threshold;
  a.lds_cap = cap;
  a.big_desc = s.big_desc; a.big_count = s.big_count; a.sing = s.sing; a.big_inv = s.big_inv;
  a.rowcnt = nullptr;
  const bool has_big = max_graph_nodes > cap && num_big > 0;
  std::unique_lock<std::mutex> side_lock(g_kron_side_mutex, std::defer_lock);
  KronSide* side = nullptr;
  hipStream_t lds_stream = stream;
  // Off unless TGP_KRON_SIDE_STREAM=1: measured 0.89 -> 0.86 ms when it works, but in some processes the second hardware
  // queue makes the whole call 3-4 ms (queue-to-queue hand-offs of hundreds of microseconds): not worth the risk.
  static const int use_side = getenv("TGP_KRON_SIDE_STREAM") ? atoi(getenv("TGP_KRON_SIDE_STREAM")) : 0;
  if (has_big && use_side) {
    side_lock.lock();  // (the fork / join events are shared by the callers of one device)
    side = kron_side();
    if (side && hipEventRecord(side->fork, stream) == hipSuccess &&
        hipStreamWaitEvent(side->stream, side->fork, 0) == hipSuccess)
      lds_stream = side->stream;
    else
      side = nullptr;
  }
  // One workgroup per graph with the graph's matrix in LDS: the allocation decides how many graphs a CU works on at
  // once (128 nodes = 132 KB = one; 64 nodes = 33 KB = four), so a batch whose longest LDS graph has more than 64
  // nodes runs the kernel twice, small graphs first with the small allocation (2048 graphs of 20..60 nodes beside one
  // 100-node graph: 320 -> ~100 us).  The kernel also has a few static LDS words: ask for what a launch needs.
  const int cuts[3] = {0, cap > 64 ? 64 : cap, cap};
  for (int cls = 0; cls < 2; ++cls) {
    if (cuts[cls + 1] <= cuts[cls]) continue;
    a.lds_lo = cuts[cls];
    a.lds_hi = cuts[cls + 1];
    const size_t lds = static_cast<size_t>(a.lds_hi) * (a.lds_hi | 1) * sizeof(double) + 16;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kron_schur_lds_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    hipLaunchKernelGGL(kron_schur_lds_kernel, dim3(static_cast<unsigned>(B)), dim3(KRON_THREADS), lds, lds_stream, a);
  }
  bool joined = true;
  if (side) joined = hipEventRecord(side->join, side->stream) == hipSuccess;
  if (has_big) {
    // graphs of 129 .. 1024 nodes: one launch per panel of KRON_SNB pivots over all of them (see kron_big_panel_kernel / kron_big_trail_kernel)
    const int nmax = static_cast<int>(max_graph_nodes < KRON_MAX_N ? max_graph_nodes : KRON_MAX_N);
    const unsigned nbig = static_cast<unsigned>(num_big);
    a.rowcnt = s.flags;  // the keep flags are spent once the ranks exist: [N + 2] words, indexed by pooled row
    (void)hipMemsetAsync(s.sing, 0, (B + 1) * sizeof(int), stream);
    (void)hipMemsetAsync(s.big, 0, static_cast<size_t>(s.cap_big) * sizeof(double), stream);
    hipLaunchKernelGGL(kron_big_build_kernel, dim3(cdiv(nmax, 256), nbig), dim3(256), 0, stream, a);
    // Unused dynamic LDS on top of the kernels' static arrays: ONE workgroup per CU.  A launch has a few hundred busy
    // workgroups at most and the dispatcher packs them four to a CU (33 KB of LDS each) while other CUs get none: the
    // trailing update of the first panels took 28 us where its arithmetic is 5 us per workgroup.
    constexpr int KRON_LDS_PAD = 56 * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kron_big_panel_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, KRON_LDS_PAD);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kron_big_trail_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, KRON_LDS_PAD);
    for (int step = 0; step * KRON_SNB < nmax - 1; ++step) {  // at least one node is kept: m <= n - 1
      const int rem = nmax - step * KRON_SNB - 1;             // trailing rows at most (the panel holds >= 1 pivot)
      const int nt = cdiv(rem, KRON_TILE);
      hipLaunchKernelGGL(kron_big_panel_kernel, dim3(static_cast<unsigned>(cdiv(rem, KRON_PW)), nbig), dim3(256), KRON_LDS_PAD, stream, a, step);
      // (a launch with several workgroups per CU anyway -- the early panels of a graph of a few thousand nodes -- is better
      //  off packed: 4 workgroups per CU hide each other's load latency: the harness batch's 2669-node graph, whole ndp forward 10.6 -> 9.8 ms)
      static const int cus = [] { int v = tgp_device_cu_count(); return v > 0 ? v : 256; }();
      const size_t trail_pad = static_cast<long>(nt) * nt * nbig >= 3l * cus ? 0 : KRON_LDS_PAD;
      hipLaunchKernelGGL(kron_big_trail_kernel, dim3(static_cast<unsigned>(nt * nt), nbig), dim3(256), trail_pad, stream, a, step);
    }
    hipLaunchKernelGGL(kron_big_finish_kernel, dim3(cdiv(nmax, 16), nbig), dim3(256), 0, stream, a);
    const size_t plds = static_cast<size_t>(2 * KRON_NB) * (nmax < KRON_REDO_MAX_N ? nmax : KRON_REDO_MAX_N) * sizeof(double);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kron_big_redo_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(plds));
    hipLaunchKernelGGL(kron_big_redo_kernel, dim3(nbig), dim3(KRON_BIG_THREADS), plds, stream, a);
  }
  if (side) {
    if (!joined || hipStreamWaitEvent(stream, side->join, 0) != hipSuccess) (void)hipStreamSynchronize(side->stream);
    side_lock.unlock();
  }
  // (the status word rides along: a declined call -- see KronStatus -- leaves -1 instead of the count; r6: one launch less)
  hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, stream, s.counts, static_cast<int>(B), s.out_off,
                     d_count, static_cast<const int*>(s.status), -1ll);
  return check_launch("tgp_kron_batched_count");
}

extern "C" int tgp_kron_batched_fill(const void* ws, int64_t N, int64_t B, int64_t max_graph_nodes, int64_t cap_dense,
                                     int64_t cap_big, int64_t num_big, const int64_t* graph_ptr, int64_t num_out,
                                     int64_t* out_row, int64_t* out_col, float* out_weight, const uint32_t* node_rank,
                                     void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(ws && N >= 0 && B >= 0 && num_out >= 0 && graph_ptr, TGP_ERR_INVALID,
              "tgp_kron_batched_fill: bad argument");
  if (num_out == 0 || B == 0) return TGP_OK;
  TGP_REQUIRE(out_row && out_col && out_weight, TGP_ERR_INVALID, "tgp_kron_batched_fill: null output");
  KronWs s;
  kron_layout(const_cast<void*>(ws), N, B, max_graph_nodes, cap_dense, cap_big, &s);
  if (node_rank) s.rank = const_cast<uint32_t*>(node_rank);  // (the table the count call was given)
  const int cap = static_cast<int>(max_graph_nodes < KRON_LDS_MAX_N ? max_graph_nodes : KRON_LDS_MAX_N);
  hipLaunchKernelGGL(kron_fill_kernel, dim3(static_cast<unsigned>(B)), dim3(KRON_THREADS), 0, stream, graph_ptr, s.rank,
                     s.sq_off, s.dense, s.out_off, cap, out_row, out_col, out_weight);
  if (num_big < 0 || num_big > B) num_big = B;
  if (max_graph_nodes > cap && num_big > 0) {
    KronArgs a{};
    a.graph_ptr = graph_ptr; a.rank = s.rank; a.sq_off = s.sq_off; a.big_off = s.big_off; a.dense = s.dense;
    a.big = s.big; a.big_desc = s.big_desc; a.big_count = s.big_count; a.rowcnt = s.flags;
    const int nmax = static_cast<int>(max_graph_nodes < KRON_MAX_N ? max_graph_nodes : KRON_MAX_N);
    hipLaunchKernelGGL(kron_big_fill_kernel, dim3(cdiv(nmax, 16), static_cast<unsigned>(num_big)), dim3(256), 0, stream, a,
                       s.out_off, out_row, out_col, out_weight);
  }
  return check_launch("tgp_kron_batched_fill");
}
