// A1 / A2: sparse Reduce  x_pool = S^T X for a sparse assignment S (one HBM pass, no atomics).
//
// Reference: tgp/reduce/base_reduce.py:141-155 materialises src = x[node_index] * weight
// ([nnz,F]) and scatter_add_s it.  Here the assignment list is inverted once (CSR by
// supernode, stable => ascending assignment order inside a supernode) and every pooled row is
// produced by one sub-wave that streams its member rows: each x row and each x_pool row
// crosses HBM exactly once, fully coalesced, and the summation order equals the sequential
// CPU scatter (so the fp32 result is reproducible and bit-identical to it: the product is
// rounded before the add, exactly as the reference's two-step form does).
#include "primitives.h"
#include <stdlib.h>

namespace tgp {

__global__ __launch_bounds__(256) void assign_keys_kernel(const int64_t* __restrict__ cluster_index,
                                                          int64_t nnz, uint32_t* __restrict__ keys,
                                                          uint32_t* __restrict__ vals) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < nnz) {
    keys[i] = static_cast<uint32_t>(cluster_index[i]);
    vals[i] = static_cast<uint32_t>(i);
  }
}

// sorted keys -> row_ptr[K+1]; perm copied out as int32.
__global__ __launch_bounds__(256) void assign_rowptr_kernel(const uint32_t* __restrict__ keys,
                                                            const uint32_t* __restrict__ vals, int64_t nnz,
                                                            int64_t K, int32_t* __restrict__ row_ptr,
                                                            int32_t* __restrict__ perm) {
  const int64_t p = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (p >= nnz) {
    if (p == nnz) {  // one extra thread closes the tail (also covers nnz == 0)
      const int64_t first = nnz > 0 ? static_cast<int64_t>(keys[nnz - 1]) + 1 : 0;
      for (int64_t c = first; c <= K; ++c) row_ptr[c] = static_cast<int32_t>(nnz);
    }
    return;
  }
  perm[p] = static_cast<int32_t>(vals[p]);
  const int64_t cur = keys[p];
  const int64_t prev = p > 0 ? static_cast<int64_t>(keys[p - 1]) : -1;
  for (int64_t c = prev + 1; c <= cur; ++c) row_ptr[c] = static_cast<int32_t>(p);
}

// ------------------------------------------------------------------ counting index (few members per supernode)
// The radix route above is three launch-bound passes (nine kernels) for a 20-bit key.  When supernodes hold only a
// few assignments each (matchings, k-MIS, partitions: nnz <= 8 K) a counting sort does it in four kernels:
// count with returning atomics (slot of each assignment inside its supernode, in arrival order), scan, scatter,
// and then every supernode's few members are put in ascending order - which is exactly what the stable sort yields.
__global__ __launch_bounds__(256) void assign_count_kernel(const int64_t* __restrict__ cluster_index, int64_t nnz,
                                                           int64_t K, uint32_t* __restrict__ cnt,
                                                           uint32_t* __restrict__ slot) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= nnz) return;
  const int64_t c = cluster_index[i];
  // an id outside [0, K) has no row: the assignment is left out (callers validate ids; this only keeps the
  // counters in bounds)
  slot[i] = static_cast<uint64_t>(c) < static_cast<uint64_t>(K) ? atomicAdd(&cnt[c], 1u) : 0xFFFFFFFFu;
}

__global__ __launch_bounds__(256) void assign_scatter_kernel(const int64_t* __restrict__ cluster_index, int64_t nnz,
                                                             int64_t K, const uint32_t* __restrict__ slot,
                                                             const int64_t* __restrict__ total,
                                                             int32_t* __restrict__ row_ptr,
                                                             int32_t* __restrict__ perm) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i == 0) row_ptr[K] = static_cast<int32_t>(*total);
  if (i >= nnz) return;
  const uint32_t sl = slot[i];
  if (sl != 0xFFFFFFFFu) perm[row_ptr[cluster_index[i]] + static_cast<int32_t>(sl)] = static_cast<int32_t>(i);
}

constexpr int AO_SMALL = 8;     // members sorted by the supernode's own thread
constexpr int AO_MED = 8192;    // members sorted by the workgroup in LDS
__global__ __launch_bounds__(256) void assign_order_kernel(const int64_t* __restrict__ cluster_index, int64_t nnz,
                                                           const int32_t* __restrict__ row_ptr, int64_t K,
                                                           int32_t* __restrict__ perm) {
  __shared__ int32_t s_buf[AO_MED];
  __shared__ int s_list[256];
  __shared__ int s_nlist;
  __shared__ uint32_t s_cnt[8 * 4];
  const int tid = threadIdx.x;
  if (tid == 0) s_nlist = 0;
  __syncthreads();
  const int64_t c = static_cast<int64_t>(blockIdx.x) * 256 + tid;
  if (c < K) {
    const int32_t b = row_ptr[c], n = row_ptr[c + 1] - b;
    if (n == 2) {
      const int32_t a0 = perm[b], a1 = perm[b + 1];
      if (a0 > a1) {
        perm[b] = a1;
        perm[b + 1] = a0;
      }
    } else if (n > 2 && n <= AO_SMALL) {  // insertion sort in place (the segment is this thread's alone)
      for (int i = 1; i < n; ++i) {
        const int32_t key = perm[b + i];
        int j = i - 1;
        while (j >= 0 && perm[b + j] > key) {
          perm[b + j + 1] = perm[b + j];
          --j;
        }
        perm[b + j + 1] = key;
      }
    } else if (n > AO_SMALL) {
      s_list[atomicAdd(&s_nlist, 1)] = tid;
    }
  }
  __syncthreads();
  const int nlist = s_nlist;
  for (int li = 0; li < nlist; ++li) {
    const int64_t cc = static_cast<int64_t>(blockIdx.x) * 256 + s_list[li];
    const int32_t b = row_ptr[cc], n = row_ptr[cc + 1] - b;
    if (n <= AO_MED) {  // bitonic sort in LDS
      int P = 16;
      while (P < n) P <<= 1;
      for (int i = tid; i < P; i += 256) s_buf[i] = i < n ? perm[b + i] : 0x7FFFFFFF;
      __syncthreads();
      for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
          for (int i = tid; i < P; i += 256) {
            const int x = i ^ j;
            if (x > i) {
              const int32_t a0 = s_buf[i], a1 = s_buf[x];
              if ((a0 > a1) == ((i & k) == 0)) {
                s_buf[i] = a1;
                s_buf[x] = a0;
              }
            }
          }
          __syncthreads();
        }
      }
      for (int i = tid; i < n; i += 256) perm[b + i] = s_buf[i];
      __syncthreads();
    } else {  // a very large supernode: its members in ascending order = a stable compaction of all assignments
      uint32_t run = 0;
      for (int64_t base = 0; base < nnz; base += 2048) {
        bool flag[8];
        uint32_t rank[8], block_total;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int64_t i = base + it * 256 + tid;
          flag[it] = i < nnz && cluster_index[i] == cc;
        }
        block_compact_ranks<8>(flag, rank, block_total, s_cnt);
#pragma unroll
        for (int it = 0; it < 8; ++it)
          if (flag[it]) perm[b + run + rank[it]] = static_cast<int32_t>(base + it * 256 + tid);
        run += block_total;
        __syncthreads();
      }
    }
  }
}

// One group of G lanes per supernode; lane g owns features [4g,4g+4) (+ 4G strides).  Every group
// works on U supernodes at once: the (row_ptr -> perm -> node_index/weight -> x row) chains of the U
// supernodes are issued side by side, so U row gathers are in flight per group instead of one
// (the kernel is latency/occupancy-bound otherwise: supernodes own only 1-2 rows each).
typedef float nt_f32x4 __attribute__((ext_vector_type(4)));

template <int G, int U, bool NT>
__global__ __launch_bounds__(256) void reduce_sparse_vec4_kernel(
    const float* __restrict__ x, int64_t F, int64_t x_stride, const int64_t* __restrict__ node_index,
    const float* __restrict__ weight, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ perm,
    int64_t nnz, int64_t K, float* __restrict__ x_pool) {
  constexpr int GROUPS = 256 / G;
  const int g = threadIdx.x % G;
  const int64_t group = static_cast<int64_t>(blockIdx.x) * GROUPS + threadIdx.x / G;
  const int64_t ngroups = static_cast<int64_t>(gridDim.x) * GROUPS;
  const int32_t last = static_cast<int32_t>(nnz > 0 ? nnz - 1 : 0);
  // Every load is unconditional, from a clamped index, and the U chains are issued level by level (a guarded load in
  // its own branch is waited for before the next one is issued: r3 found four serial round trips in front of the row
  // requests); what a clamped load returned is dropped by the select in front of the add.
  for (int64_t c0 = group; c0 < K; c0 += ngroups * U) {
    int32_t beg[U], len[U];
    int32_t maxlen = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t c = c0 + u * ngroups;
      const int64_t cc = c < K ? c : K - 1;
      if (row_ptr) {
        beg[u] = row_ptr[cc];
        len[u] = row_ptr[cc + 1] - beg[u];
      } else {  // one assignment per supernode: the table is the identity (reduce_one_to_one_kernel is the fast path)
        beg[u] = static_cast<int32_t>(cc);
        len[u] = 1;
      }
      if (c >= K) len[u] = -1;
      maxlen = len[u] > maxlen ? len[u] : maxlen;
    }
    for (int64_t f = 4 * g; f < F; f += 4 * G) {
      float4 acc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      // r4: TWO members of every supernode per round, level by level (slots, then node ids + weights, then rows): a
      // Graclus supernode (one or two members) is row offsets -> slots -> [ids] -> rows, three or four dependent round
      // trips where the one-member-at-a-time loop paid seven; the adds keep the member order.  node_index == NULL: the
      // source row IS the assignment id (S's row index is 0..N-1: GraclusSelect, a cluster vector), one level less.
      constexpr int MP = 2;
      for (int32_t m0 = 0; m0 < maxlen; m0 += MP) {
        int32_t a[U][MP];
        float w[U][MP];
        int64_t n[U][MP];
        float4 v[U][MP];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int q = 0; q < MP; ++q) {
            int32_t slot = beg[u] + (m0 + q < len[u] ? m0 + q : 0);
            slot = slot < last ? slot : last;
            a[u][q] = perm ? perm[slot] : slot;
          }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int q = 0; q < MP; ++q) {
            w[u][q] = weight ? weight[a[u][q]] : 1.0f;
            n[u][q] = node_index ? node_index[a[u][q]] : static_cast<int64_t>(a[u][q]);
          }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int q = 0; q < MP; ++q) {
            const float* src = x + n[u][q] * x_stride + f;
            if constexpr (NT) {  // rows are read exactly once: keep them out of the way of the index tables
              const nt_f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const nt_f32x4*>(src));
              v[u][q] = make_float4(t.x, t.y, t.z, t.w);
            } else {
              v[u][q] = *reinterpret_cast<const float4*>(src);
            }
          }
#pragma unroll
        for (int q = 0; q < MP; ++q)
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const bool on = m0 + q < len[u];  // product rounded before the add, as the reference's two-step form
            const float wq = w[u][q];
            const float px = __fadd_rn(acc[u].x, __fmul_rn(v[u][q].x, wq)), py = __fadd_rn(acc[u].y, __fmul_rn(v[u][q].y, wq));
            const float pz = __fadd_rn(acc[u].z, __fmul_rn(v[u][q].z, wq)), pw = __fadd_rn(acc[u].w, __fmul_rn(v[u][q].w, wq));
            acc[u].x = on ? px : acc[u].x;
            acc[u].y = on ? py : acc[u].y;
            acc[u].z = on ? pz : acc[u].z;
            acc[u].w = on ? pw : acc[u].w;
          }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t c = c0 + u * ngroups;
        if (len[u] >= 0) {
          if constexpr (NT) {
            nt_f32x4 t = {acc[u].x, acc[u].y, acc[u].z, acc[u].w};
            __builtin_nontemporal_store(t, reinterpret_cast<nt_f32x4*>(x_pool + c * F + f));
          } else {
            *reinterpret_cast<float4*>(x_pool + c * F + f) = acc[u];
          }
        }
      }
    }
  }
}

// One assignment per supernode (row_ptr == NULL: TopK, NDP) with every load unconditional and the U chains issued level
// by level: U table entries, then U node ids + U weights, then the U rows.  In the general kernel above each guarded
// load sits in its own branch and the compiler waits for it before the next one is issued (four serial round trips in
// front of the two row requests, and the rows leave as FLAT loads); here a group has U rows in flight after three.
// Supernodes past the end re-read the last one and are not stored.  Same products as the general kernel: 0 + x * w.
template <int G, int U, bool HAS_W, bool HAS_PERM>
__global__ __launch_bounds__(256) void reduce_one_to_one_kernel(
    const float* __restrict__ x, int64_t F, int64_t x_stride, const int64_t* __restrict__ node_index,
    const float* __restrict__ weight, const int32_t* __restrict__ perm, int64_t K, float* __restrict__ x_pool) {
  constexpr int GROUPS = 256 / G;
  const int g = threadIdx.x % G;
  const int64_t group = static_cast<int64_t>(blockIdx.x) * GROUPS + threadIdx.x / G;
  const int64_t ngroups = static_cast<int64_t>(gridDim.x) * GROUPS;
  for (int64_t c0 = group; c0 < K; c0 += ngroups * U) {
    int64_t c[U];
    int32_t a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      c[u] = c0 + u * ngroups;
      const int64_t cc = c[u] < K ? c[u] : K - 1;
      a[u] = HAS_PERM ? perm[cc] : static_cast<int32_t>(cc);
    }
    int64_t n[U];
    float w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      n[u] = node_index[a[u]];
      w[u] = HAS_W ? weight[a[u]] : 1.0f;
    }
    for (int64_t f = 4 * g; f < F; f += 4 * G) {
      nt_f32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        v[u] = __builtin_nontemporal_load(reinterpret_cast<const nt_f32x4*>(x + n[u] * x_stride + f));
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (c[u] < K) {  // product rounded before the add, as the reference's two-step form
          nt_f32x4 t = {__fadd_rn(0.f, __fmul_rn(v[u].x, w[u])), __fadd_rn(0.f, __fmul_rn(v[u].y, w[u])),
                        __fadd_rn(0.f, __fmul_rn(v[u].z, w[u])), __fadd_rn(0.f, __fmul_rn(v[u].w, w[u]))};
          __builtin_nontemporal_store(t, reinterpret_cast<nt_f32x4*>(x_pool + c[u] * F + f));
        }
      }
    }
  }
}

// r4: the same with the PACKED one-to-one index: pack[c] = {int32 source row, fp32 weight} of supernode c's single
// assignment, written by the selector next to its own outputs (TopkSelect) or by one_to_one_index_kernel.  The kernel
// above reads perm[c] and then gathers an 8-byte node id and a 4-byte weight at perm[c]: two random 64-byte sectors
// per supernode (PMC, r3: 70 MB of the 590 MB the TopK Reduce moved at N = 1M) and one more dependent load level in
// front of the row requests.  Here the index is ONE streamed 8-byte load per supernode.
template <int G, int U, bool HAS_W>
__global__ __launch_bounds__(256) void reduce_one_to_one_packed_kernel(
    const float* __restrict__ x, int64_t F, int64_t x_stride, const uint2* __restrict__ pack, int64_t K,
    float* __restrict__ x_pool) {
  constexpr int GROUPS = 256 / G;
  const int g = threadIdx.x % G;
  const int64_t group = static_cast<int64_t>(blockIdx.x) * GROUPS + threadIdx.x / G;
  const int64_t ngroups = static_cast<int64_t>(gridDim.x) * GROUPS;
  for (int64_t c0 = group; c0 < K; c0 += ngroups * U) {
    int64_t c[U];
    uint2 pk[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      c[u] = c0 + u * ngroups;
      pk[u] = pack[c[u] < K ? c[u] : K - 1];
    }
    for (int64_t f = 4 * g; f < F; f += 4 * G) {
      nt_f32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        v[u] = __builtin_nontemporal_load(reinterpret_cast<const nt_f32x4*>(x + static_cast<int64_t>(pk[u].x) * x_stride + f));
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (c[u] < K) {  // product rounded before the add, as the reference's two-step form
          const float w = HAS_W ? __uint_as_float(pk[u].y) : 1.0f;
          nt_f32x4 t = {__fadd_rn(0.f, __fmul_rn(v[u].x, w)), __fadd_rn(0.f, __fmul_rn(v[u].y, w)),
                        __fadd_rn(0.f, __fmul_rn(v[u].z, w)), __fadd_rn(0.f, __fmul_rn(v[u].w, w))};
          __builtin_nontemporal_store(t, reinterpret_cast<nt_f32x4*>(x_pool + c[u] * F + f));
        }
      }
    }
  }
}

// perm[c] = a, pack[c] = {node_index[a], weight[a]} for the single assignment a of supernode c = cluster_index[a]
__global__ __launch_bounds__(256) void one_to_one_index_kernel(const int64_t* __restrict__ node_index,
                                                               const int64_t* __restrict__ cluster_index,
                                                               const float* __restrict__ weight, int64_t k,
                                                               int32_t* __restrict__ perm, uint2* __restrict__ pack) {
  const int64_t a = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (a >= k) return;
  const int64_t c = cluster_index[a];
  if (static_cast<uint64_t>(c) >= static_cast<uint64_t>(k)) return;  // (callers validate ids; this keeps the stores in bounds)
  perm[c] = static_cast<int32_t>(a);
  pack[c] = make_uint2(static_cast<uint32_t>(node_index[a]), weight ? __float_as_uint(weight[a]) : 0x3F800000u);
}

// Fallback for feature counts / strides that are not 16-byte friendly: one lane per feature.
__global__ __launch_bounds__(256) void reduce_sparse_scalar_kernel(
    const float* __restrict__ x, int64_t F, int64_t x_stride, const int64_t* __restrict__ node_index,
    const float* __restrict__ weight, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ perm,
    int64_t K, float* __restrict__ x_pool) {
  const int64_t total = K * F;
  for (int64_t o = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; o < total;
       o += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t c = o / F, f = o - c * F;
    float acc = 0.f;
    const int32_t p_beg = row_ptr ? row_ptr[c] : static_cast<int32_t>(c);
    const int32_t p_end = row_ptr ? row_ptr[c + 1] : p_beg + 1;
    for (int32_t p = p_beg; p < p_end; ++p) {
      const int32_t a = perm ? perm[p] : p;
      const float w = weight ? weight[a] : 1.0f;
      acc = __fadd_rn(acc, __fmul_rn(x[(node_index ? node_index[a] : static_cast<int64_t>(a)) * x_stride + f], w));
    }
    x_pool[o] = acc;
  }
}

__global__ __launch_bounds__(256) void arange_i64_kernel(int64_t* out, int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) out[i] = i;
}

__global__ __launch_bounds__(256) void reduce_batch_kernel(const int64_t* __restrict__ batch,
                                                           const int64_t* __restrict__ node_index,
                                                           const int64_t* __restrict__ cluster_index,
                                                           int64_t nnz, int64_t* __restrict__ out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < nnz) out[cluster_index[i]] = batch[node_index[i]];
}

}  // namespace tgp

using namespace tgp;

// counting route: cnt [K], slot [nnz], scan tiles, total
static size_t assign_counting_bytes(int64_t nnz, int64_t K) {
  const size_t n = nnz > 0 ? static_cast<size_t>(nnz) : 1, k = K > 0 ? static_cast<size_t>(K) : 1;
  return align_up(k * sizeof(uint32_t)) + align_up(n * sizeof(uint32_t)) +
         align_up((2 * static_cast<size_t>(cdiv(static_cast<int64_t>(k), SCAN_TILE)) + 16) * sizeof(uint32_t)) +
         align_up(2 * sizeof(int64_t));
}
static bool assign_use_counting(int64_t nnz, int64_t K) {
  return nnz >= 4096 && nnz <= 8 * K && K <= 4 * nnz;  // few members per supernode, and not mostly empty rows
}

extern "C" size_t tgp_assign_index_workspace_bytes(int64_t nnz, int64_t num_supernodes) {
  const size_t n = nnz > 0 ? static_cast<size_t>(nnz) : 1;
  const size_t radix = 4 * align_up(n * sizeof(uint32_t)) + align_up(sort_scratch_words() * sizeof(uint32_t));
  const size_t counting = assign_use_counting(nnz, num_supernodes) ? assign_counting_bytes(nnz, num_supernodes) : 0;
  return radix > counting ? radix : counting;
}

extern "C" int tgp_assign_index_build(const int64_t* cluster_index, int64_t nnz, int64_t K, int32_t* row_ptr,
                                      int32_t* perm, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(nnz >= 0 && K >= 0 && row_ptr && (nnz == 0 || (cluster_index && perm)), TGP_ERR_INVALID,
              "tgp_assign_index_build: bad argument");
  TGP_REQUIRE(nnz < (1ll << 31) && K < (1ll << 31), TGP_ERR_RANGE, "tgp_assign_index_build: nnz/K >= 2^31");
  TGP_REQUIRE(ws && ws_bytes >= tgp_assign_index_workspace_bytes(nnz, K), TGP_ERR_WORKSPACE,
              "tgp_assign_index_build: workspace too small");
  Carver cv(ws);
  const size_t n = nnz > 0 ? static_cast<size_t>(nnz) : 1;
  if (assign_use_counting(nnz, K)) {
    uint32_t* cnt = cv.take<uint32_t>(static_cast<size_t>(K));
    uint32_t* slot = cv.take<uint32_t>(n);
    uint32_t* tiles = cv.take<uint32_t>(2 * static_cast<size_t>(cdiv(K, SCAN_TILE)) + 16);
    int64_t* total = cv.take<int64_t>(2);
    (void)hipMemsetAsync(cnt, 0, static_cast<size_t>(K) * sizeof(uint32_t), stream);
    hipLaunchKernelGGL(assign_count_kernel, dim3(cdiv(nnz, 256)), dim3(256), 0, stream, cluster_index, nnz, K, cnt, slot);
    device_scan_u32(cnt, K, reinterpret_cast<uint32_t*>(row_ptr), total, tiles, stream);
    hipLaunchKernelGGL(assign_scatter_kernel, dim3(cdiv(nnz, 256)), dim3(256), 0, stream, cluster_index, nnz, K, slot,
                       total, row_ptr, perm);
    hipLaunchKernelGGL(assign_order_kernel, dim3(cdiv(K, 256)), dim3(256), 0, stream, cluster_index, nnz, row_ptr, K,
                       perm);
    return check_launch("tgp_assign_index_build");
  }
  uint32_t* k0 = cv.take<uint32_t>(n);
  uint32_t* v0 = cv.take<uint32_t>(n);
  uint32_t* k1 = cv.take<uint32_t>(n);
  uint32_t* v1 = cv.take<uint32_t>(n);
  uint32_t* scratch = cv.take<uint32_t>(sort_scratch_words());
  bool first = true;
  if (nnz > 0) {
    hipLaunchKernelGGL(assign_keys_kernel, dim3(cdiv(nnz, 256)), dim3(256), 0, stream, cluster_index, nnz, k0,
                       v0);
    const int rc = radix_sort_pairs<uint32_t, uint32_t>(k0, v0, k1, v1, nnz, bits_for(K > 0 ? K - 1 : 0),
                                                        scratch, stream, &first);
    if (rc != TGP_OK) return rc;
  }
  hipLaunchKernelGGL(assign_rowptr_kernel, dim3(cdiv(nnz + 1, 256)), dim3(256), 0, stream, first ? k0 : k1,
                     first ? v0 : v1, nnz, K, row_ptr, perm);
  return check_launch("tgp_assign_index_build");
}

extern "C" int tgp_reduce_sparse_f32(const float* x, int64_t num_nodes, int64_t F, int64_t x_stride,
                                     const int64_t* node_index, const float* weight, const int32_t* row_ptr,
                                     const int32_t* perm, int64_t nnz, int64_t K, float* x_pool,
                                     void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(num_nodes >= 0 && F >= 0 && K >= 0 && nnz >= 0 && (row_ptr || nnz == K), TGP_ERR_INVALID,
              "tgp_reduce_sparse_f32: bad argument");  // row_ptr == NULL: one assignment per supernode
  if (K == 0 || F == 0) return TGP_OK;
  TGP_REQUIRE(x_pool && (nnz == 0 || x), TGP_ERR_INVALID,  // perm == NULL: identity order
              "tgp_reduce_sparse_f32: null pointer");
  // node_index == NULL (r4): assignment i reads row i (S's row index is 0..nnz-1); only the vectorised general kernel
  TGP_REQUIRE(node_index || (row_ptr && nnz <= num_nodes), TGP_ERR_INVALID,
              "tgp_reduce_sparse_f32: node_index may only be omitted with a supernode index over nnz <= num_nodes rows");
  const bool vec = (F % 4 == 0) && (x_stride % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0) &&
                   (reinterpret_cast<uintptr_t>(x_pool) % 16 == 0);
  const int cus = 256;
  if (vec) {
    const int64_t lanes = F / 4;
    int G = 1;
    while (G < lanes && G < 64) G <<= 1;
    const int64_t groups_per_block = 256 / G;
    int64_t blocks = (K + groups_per_block * 4 - 1) / (groups_per_block * 4);
    if (blocks > cus * 8) blocks = cus * 8;
    if (blocks < 1) blocks = 1;
    dim3 grid(static_cast<unsigned>(blocks)), block(256);
    // supernodes in flight per lane group: with two MEMBERS of a supernode requested per round (r4) a Graclus-shaped
    // assignment (at most two members: nnz <= 2 K) has its rows in flight from ONE supernode per group and a second one
    // only costs registers (C4 Reduce, MI355X: U = 1 0.134 ms = 0.755 of HBM, U = 2 0.142, U = 4 0.155); longer member
    // lists keep two supernodes per group as before.  TGP_REDUCE_U overrides (A/B measurements).
    static const int kUenv = getenv("TGP_REDUCE_U") ? atoi(getenv("TGP_REDUCE_U")) : 0;
    const int kU = kUenv > 0 ? kUenv : (row_ptr && nnz <= 2 * K ? 1 : 2);
    static const int kNT = getenv("TGP_REDUCE_NT") ? atoi(getenv("TGP_REDUCE_NT")) : 1;
    static const int kBlocksPerCu = getenv("TGP_REDUCE_BPC") ? atoi(getenv("TGP_REDUCE_BPC")) : 8;
    blocks = (K + groups_per_block * kU - 1) / (groups_per_block * kU);
    if (blocks > cus * kBlocksPerCu) blocks = cus * kBlocksPerCu;
    if (blocks < 1) blocks = 1;
    grid = dim3(static_cast<unsigned>(blocks));
    static const int kO2O = getenv("TGP_REDUCE_O2O") ? atoi(getenv("TGP_REDUCE_O2O")) : 2;  // 0: general kernel
    if (!row_ptr && kO2O > 0 && G >= 8) {
      const int uu = kO2O >= 8 ? 8 : (kO2O >= 4 ? 4 : (kO2O >= 2 ? 2 : 1));
      blocks = (K + groups_per_block * uu - 1) / (groups_per_block * uu);
      if (blocks > cus * kBlocksPerCu) blocks = cus * kBlocksPerCu;
      if (blocks < 1) blocks = 1;
      grid = dim3(static_cast<unsigned>(blocks));
#define TGP_LAUNCH_O(GG, UU)                                                                                     \
  do {                                                                                                            \
    if (weight && perm)                                                                                           \
      hipLaunchKernelGGL((reduce_one_to_one_kernel<GG, UU, true, true>), grid, block, 0, stream, x, F, x_stride,  \
                         node_index, weight, perm, K, x_pool);                                                    \
    else if (weight)                                                                                              \
      hipLaunchKernelGGL((reduce_one_to_one_kernel<GG, UU, true, false>), grid, block, 0, stream, x, F, x_stride, \
                         node_index, weight, perm, K, x_pool);                                                    \
    else if (perm)                                                                                                \
      hipLaunchKernelGGL((reduce_one_to_one_kernel<GG, UU, false, true>), grid, block, 0, stream, x, F, x_stride, \
                         node_index, weight, perm, K, x_pool);                                                    \
    else                                                                                                          \
      hipLaunchKernelGGL((reduce_one_to_one_kernel<GG, UU, false, false>), grid, block, 0, stream, x, F,          \
                         x_stride, node_index, weight, perm, K, x_pool);                                          \
  } while (0)
#define TGP_LAUNCH_OG(GG)                   \
  do {                                      \
    if (uu == 1) TGP_LAUNCH_O(GG, 1);       \
    else if (uu == 2) TGP_LAUNCH_O(GG, 2);  \
    else if (uu == 4) TGP_LAUNCH_O(GG, 4);  \
    else TGP_LAUNCH_O(GG, 8);               \
  } while (0)
      switch (G) {
        case 8: TGP_LAUNCH_OG(8); break;
        case 16: TGP_LAUNCH_OG(16); break;
        case 32: TGP_LAUNCH_OG(32); break;
        default: TGP_LAUNCH_OG(64); break;
      }
#undef TGP_LAUNCH_OG
#undef TGP_LAUNCH_O
      return check_launch("tgp_reduce_sparse_f32");
    }
#define TGP_LAUNCH_GU(GG, UU)                                                                                  \
  do {                                                                                                          \
    if (kNT)                                                                                                    \
      hipLaunchKernelGGL((reduce_sparse_vec4_kernel<GG, UU, true>), grid, block, 0, stream, x, F, x_stride,     \
                         node_index, weight, row_ptr, perm, nnz, K, x_pool);                                       \
    else                                                                                                        \
      hipLaunchKernelGGL((reduce_sparse_vec4_kernel<GG, UU, false>), grid, block, 0, stream, x, F, x_stride,    \
                         node_index, weight, row_ptr, perm, nnz, K, x_pool);                                       \
  } while (0)
#define TGP_LAUNCH_G(GG)                  \
  do {                                    \
    if (kU == 1) TGP_LAUNCH_GU(GG, 1);    \
    else if (kU == 2) TGP_LAUNCH_GU(GG, 2); \
    else TGP_LAUNCH_GU(GG, 4);            \
  } while (0)
    switch (G) {
      case 1: TGP_LAUNCH_G(1); break;
      case 2: TGP_LAUNCH_G(2); break;
      case 4: TGP_LAUNCH_G(4); break;
      case 8: TGP_LAUNCH_G(8); break;
      case 16: TGP_LAUNCH_G(16); break;
      case 32: TGP_LAUNCH_G(32); break;
      default: TGP_LAUNCH_G(64); break;
    }
#undef TGP_LAUNCH_G
  } else {
    int64_t blocks = (K * F + 255) / 256;
    if (blocks > cus * 16) blocks = cus * 16;
    hipLaunchKernelGGL(reduce_sparse_scalar_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, x,
                       F, x_stride, node_index, weight, row_ptr, perm, K, x_pool);
  }
  return check_launch("tgp_reduce_sparse_f32");
}

extern "C" int tgp_one_to_one_index_build(const int64_t* node_index, const int64_t* cluster_index, const float* weight,
                                          int64_t k, int32_t* perm, uint64_t* pack, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(k >= 0, TGP_ERR_INVALID, "tgp_one_to_one_index_build: negative size");
  if (k == 0) return TGP_OK;
  TGP_REQUIRE(node_index && cluster_index && perm && pack, TGP_ERR_INVALID, "tgp_one_to_one_index_build: null pointer");
  TGP_REQUIRE(k < (1ll << 31), TGP_ERR_RANGE, "tgp_one_to_one_index_build: k >= 2^31");
  hipLaunchKernelGGL(one_to_one_index_kernel, dim3(cdiv(k, 256)), dim3(256), 0, stream, node_index, cluster_index, weight,
                     k, perm, reinterpret_cast<uint2*>(pack));
  return check_launch("tgp_one_to_one_index_build");
}

extern "C" int tgp_reduce_one_to_one_f32(const float* x, int64_t num_nodes, int64_t F, int64_t x_stride,
                                         const uint64_t* pack, int has_weight, int64_t K, float* x_pool, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(num_nodes >= 0 && F >= 0 && K >= 0, TGP_ERR_INVALID, "tgp_reduce_one_to_one_f32: bad argument");
  if (K == 0 || F == 0) return TGP_OK;
  TGP_REQUIRE(x && pack && x_pool, TGP_ERR_INVALID, "tgp_reduce_one_to_one_f32: null pointer");
  TGP_REQUIRE((F % 4 == 0) && (x_stride % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0) &&
                  (reinterpret_cast<uintptr_t>(x_pool) % 16 == 0) && F >= 32 && num_nodes < (1ll << 32),
              TGP_ERR_INVALID, "tgp_reduce_one_to_one_f32: needs 16-byte friendly rows of at least 32 features "
                               "(tgp_reduce_sparse_f32 takes every shape)");
  const uint2* pk = reinterpret_cast<const uint2*>(pack);
  const int64_t lanes = F / 4;
  int G = 8;
  while (G < lanes && G < 64) G <<= 1;
  static const int kBlocksPerCu = getenv("TGP_REDUCE_BPC") ? atoi(getenv("TGP_REDUCE_BPC")) : 8;
  // supernodes in flight per lane group: one (r4 sweep on MI355X, topk1m / c4_ndp: U = 1 0.760 / 0.787-0.793 of HBM,
  // U = 2 0.751 / 0.781, U = 4 0.68: the packed index made the chain short enough that occupancy wins)
  static const int kU = getenv("TGP_REDUCE_O2O") ? atoi(getenv("TGP_REDUCE_O2O")) : 1;
  const int uu = kU >= 4 ? 4 : (kU >= 2 ? 2 : 1);
  const int64_t groups_per_block = 256 / G;
  int64_t blocks = (K + groups_per_block * uu - 1) / (groups_per_block * uu);
  if (blocks > 256 * kBlocksPerCu) blocks = 256 * kBlocksPerCu;
  if (blocks < 1) blocks = 1;
  const dim3 grid(static_cast<unsigned>(blocks)), block(256);
#define TGP_LAUNCH_P(GG, UU)                                                                                        \
  do {                                                                                                               \
    if (has_weight)                                                                                                  \
      hipLaunchKernelGGL((reduce_one_to_one_packed_kernel<GG, UU, true>), grid, block, 0, stream, x, F, x_stride, pk, \
                         K, x_pool);                                                                                 \
    else                                                                                                             \
      hipLaunchKernelGGL((reduce_one_to_one_packed_kernel<GG, UU, false>), grid, block, 0, stream, x, F, x_stride,   \
                         pk, K, x_pool);                                                                             \
  } while (0)
#define TGP_LAUNCH_PG(GG)                 \
  do {                                    \
    if (uu == 1) TGP_LAUNCH_P(GG, 1);     \
    else if (uu == 2) TGP_LAUNCH_P(GG, 2); \
    else TGP_LAUNCH_P(GG, 4);             \
  } while (0)
  switch (G) {
    case 8: TGP_LAUNCH_PG(8); break;
    case 16: TGP_LAUNCH_PG(16); break;
    case 32: TGP_LAUNCH_PG(32); break;
    default: TGP_LAUNCH_PG(64); break;
  }
#undef TGP_LAUNCH_PG
#undef TGP_LAUNCH_P
  return check_launch("tgp_reduce_one_to_one_f32");
}

extern "C" int tgp_reduce_batch_i64(const int64_t* batch, const int64_t* node_index,
                                    const int64_t* cluster_index, int64_t nnz, int64_t K, int every_cluster_has_a_node,
                                    int64_t* batch_pool, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(nnz >= 0 && K >= 0, TGP_ERR_INVALID, "tgp_reduce_batch_i64: bad size");
  if (K == 0) return TGP_OK;
  TGP_REQUIRE(batch_pool && (nnz == 0 || (batch && node_index && cluster_index)), TGP_ERR_INVALID,
              "tgp_reduce_batch_i64: null pointer");
  // (the arange only shows through for supernodes without a node: a caller that knows there is none skips the launch)
  if (!every_cluster_has_a_node || nnz == 0)
    hipLaunchKernelGGL(arange_i64_kernel, dim3(cdiv(K, 256)), dim3(256), 0, stream, batch_pool, K);
  if (nnz > 0)
    hipLaunchKernelGGL(reduce_batch_kernel, dim3(cdiv(nnz, 256)), dim3(256), 0, stream, batch, node_index,
                       cluster_index, nnz, batch_pool);
  return check_launch("tgp_reduce_batch_i64");
}
