// A12: per-graph top-k node selection (reference select/topk_select.py:163-203 -> PyG
// nn/pool/select/topk.py `topk(score, ratio, batch)`), ratio mode.
//
// PyG sorts all scores descending, then stable-sorts by graph id and keeps the first k_g entries of every
// graph; the reference's SelectOutput then sorts the kept node ids once more to build a row-sorted COO
// assignment (select/base_select.py:58).  That is three device-wide comparison sorts of N (or k) elements.
// Here:  one stable LSD radix sort of the composite key (graph id : descending score bits) with the node id
// as payload  ->  a rank pass that hands every kept node its supernode id (= its position in PyG's
// graph-major, score-descending order)  ->  an order-preserving compaction over the node ids, which yields
// the row-sorted assignment directly, together with the supernode -> assignment inverted index the sparse
// Reduce kernel wants.  Ties keep the lower node id first (what a stable descending sort does).
#include "common.h"
#include "primitives.h"

namespace tgp {

constexpr int kTopkItems = 8;
constexpr int kTopkTile = 256 * kTopkItems;

// Bit pattern that orders like the float, descending, under an unsigned ascending sort.  -0 == +0 and every
// NaN sorts first (torch.sort's descending order treats NaN as the largest value).
__device__ __forceinline__ uint32_t descending_key(float f) {
  uint32_t b = __float_as_uint(f);
  if (f == 0.f) b = 0u;
  if (f != f) b = 0x7FC00000u;
  const uint32_t asc = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
  return ~asc;
}

__global__ __launch_bounds__(256) void topk_keys_kernel(const float* __restrict__ score,
                                                        const int64_t* __restrict__ batch, int64_t n,
                                                        uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  const uint64_t g = batch ? static_cast<uint64_t>(batch[i]) : 0ull;
  keys[i] = (g << 32) | descending_key(score[i]);
  vals[i] = static_cast<uint32_t>(i);
}

// Sorted position p belongs to graph g = key >> 32 at local rank q = p - ptr[g]; the first k[g] of a graph are
// kept and become supernode koff[g] + q.
__global__ __launch_bounds__(256) void topk_rank_kernel(const uint64_t* __restrict__ keys,
                                                        const uint32_t* __restrict__ vals, int64_t n,
                                                        const int64_t* __restrict__ ptr,
                                                        const int64_t* __restrict__ k,
                                                        const int64_t* __restrict__ koff,
                                                        int32_t* __restrict__ rank_of) {
  const int64_t p = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (p >= n) return;
  const int64_t g = static_cast<int64_t>(keys[p] >> 32);
  const int64_t q = p - ptr[g];
  if (q < k[g]) rank_of[vals[p]] = static_cast<int32_t>(koff[g] + q);
}

// Batches of many small graphs (sorted batch vector: graph g owns nodes ptr[g] .. ptr[g+1]): the composite-key
// radix sort needs 32 + log2(B) bits = 5-6 passes of three launches each, all of them launch-bound on a few
// hundred KB.  Instead every graph is sorted on its own: one WAVE per graph when it has at most 64 nodes
// (bitonic network over the lanes, 64-bit (descending score bits : local index) keys, so ties keep the lower
// node id), one WORKGROUP per graph up to kSegSortMax nodes (the same network through LDS).  The q-th element of
// a graph's order gets supernode koff[g] + q if q < k[g] -- exactly what topk_rank_kernel writes.
constexpr int kSegSortMax = 2048;

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int mask) {
  const unsigned lo = __shfl_xor(static_cast<unsigned>(v), mask, WAVE);
  const unsigned hi = __shfl_xor(static_cast<unsigned>(v >> 32), mask, WAVE);
  return (static_cast<unsigned long long>(hi) << 32) | lo;
}

__global__ __launch_bounds__(256) void topk_segsort_wave_kernel(const float* __restrict__ score,
                                                                const int64_t* __restrict__ ptr,
                                                                const int64_t* __restrict__ k,
                                                                const int64_t* __restrict__ koff, int64_t B,
                                                                int32_t* __restrict__ rank_of) {
  const int64_t g = static_cast<int64_t>(blockIdx.x) * 4 + wave_id();
  if (g >= B) return;
  const int lane = lane_id();
  const int64_t lo = ptr[g];
  const int n = static_cast<int>(ptr[g + 1] - lo);
  unsigned long long v = ~0ull;  // sentinel: sorts last
  if (lane < n) v = (static_cast<unsigned long long>(descending_key(score[lo + lane])) << 32) | static_cast<unsigned>(lane);
#pragma unroll
  for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const unsigned long long o = shfl_xor_u64(v, stride);
      const bool up = (lane & size) == 0;          // ascending block
      const bool lower = (lane & stride) == 0;     // this lane keeps the smaller of the pair in an ascending block
      const bool take_min = up == lower;
      v = take_min ? (v < o ? v : o) : (v > o ? v : o);
    }
  }
  if (lane < n && lane < k[g]) rank_of[lo + static_cast<int64_t>(v & 0xFFFFFFFFull)] = static_cast<int32_t>(koff[g] + lane);
}

// The same sort with the outputs written by the wave itself (r3, late): a sorted batch puts graph g's kept nodes at
// [koff[g], koff[g] + k_g) of the ascending node_index, so the position of a kept node is koff[g] + (kept nodes of ITS
// graph in front of it) -- a ballot -- and the device-wide rank table, its memset, the count, the scan and the fill
// launch are not needed: TopkSelect on a batch of small graphs is the score kernel + this one.
__global__ __launch_bounds__(256) void topk_segsort_wave_fill_kernel(
    const float* __restrict__ score, const int64_t* __restrict__ ptr, const int64_t* __restrict__ k,
    const int64_t* __restrict__ koff, int64_t B, int64_t N, int64_t* __restrict__ node_index,
    int64_t* __restrict__ cluster_index, int32_t* __restrict__ assign_perm, float* __restrict__ values,
    int32_t* __restrict__ lift_ptr, uint2* __restrict__ assign_pack) {
  __shared__ int s_inv[4][64];
  const int w = wave_id();
  const int64_t g = static_cast<int64_t>(blockIdx.x) * 4 + w;
  if (g >= B) return;
  const int lane = lane_id();
  const int64_t lo = ptr[g];
  const int n = static_cast<int>(ptr[g + 1] - lo);
  if (n > 64) return;  // beyond the promised segments_max_nodes <= 64: left out (see topk_segsort_block_kernel)
  const float sc = lane < n ? score[lo + lane] : 0.f;
  unsigned long long v = ~0ull;  // sentinel: sorts last
  if (lane < n) v = (static_cast<unsigned long long>(descending_key(sc)) << 32) | static_cast<unsigned>(lane);
#pragma unroll
  for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const unsigned long long o = shfl_xor_u64(v, stride);
      const bool up = (lane & size) == 0;
      const bool lower = (lane & stride) == 0;
      const bool take_min = up == lower;
      v = take_min ? (v < o ? v : o) : (v > o ? v : o);
    }
  }
  // lane q holds the node with rank q; every node looks its rank up (one LDS exchange inside the wave)
  if (lane < n) s_inv[w][static_cast<int>(v & 0xFFFFFFFFull)] = lane;
  __builtin_amdgcn_wave_barrier();
  const int q_of = lane < n ? s_inv[w][lane] : 64;
  const int64_t kg64 = k[g];
  const int kg = static_cast<int>(kg64 < n ? kg64 : n);
  const bool kept = lane < n && q_of < kg;
  const unsigned long long mask = __ballot(kept);
  const int64_t base = koff[g];
  const int64_t pos = base + __popcll(mask & lanemask_lt());
  if (lift_ptr) {
    if (lane < n) lift_ptr[lo + lane] = static_cast<int32_t>(pos);
    if (g == B - 1 && lane == 0) lift_ptr[N] = static_cast<int32_t>(koff[B]);
  }
  if (kept) {
    node_index[pos] = lo + lane;
    cluster_index[pos] = base + q_of;
    assign_perm[base + q_of] = static_cast<int32_t>(pos);
    if (assign_pack) assign_pack[base + q_of] = make_uint2(static_cast<uint32_t>(lo + lane), __float_as_uint(sc));
    if (values) values[pos] = sc;
  }
}

// T = 256 threads for graphs up to kSegSortMax nodes, 1024 threads (dynamic LDS) up to kSegSortLarge: a handful of
// graphs of a few thousand nodes (the reference harness's batches) took the device-wide radix sort before -- 15 launches
// for 4500 keys.
constexpr int kSegSortLarge = 8192;
// FILL: the workgroup also writes the final outputs (see topk_segsort_wave_fill_kernel): kept flags by local node id in
// LDS behind the keys ([m] ints), positions from a workgroup scan in node order.
template <int T, bool FILL>
__global__ __launch_bounds__(T) void topk_segsort_block_kernel(const float* __restrict__ score,
                                                               const int64_t* __restrict__ ptr,
                                                               const int64_t* __restrict__ k,
                                                               const int64_t* __restrict__ koff,
                                                               int32_t* __restrict__ rank_of, int64_t B, int64_t N,
                                                               int64_t* __restrict__ node_index,
                                                               int64_t* __restrict__ cluster_index,
                                                               int32_t* __restrict__ assign_perm,
                                                               float* __restrict__ values,
                                                               int32_t* __restrict__ lift_ptr,
                                                               uint2* __restrict__ assign_pack, int cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_v[];
  const int64_t g = blockIdx.x;
  const int64_t lo = ptr[g];
  const int n = static_cast<int>(ptr[g + 1] - lo);
  if (n > cap) return;  // a graph beyond the promised segments_max_nodes: the LDS tile is sized for `cap` keys -- left
                        // out rather than written past it (the caller broke its promise; the host mirror passes the
                        // exact maximum from the batch facts)
  if constexpr (FILL) {
    if (lift_ptr && g == B - 1 && threadIdx.x == 0) lift_ptr[N] = static_cast<int32_t>(koff[B]);
  }
  if (n == 0) return;
  int m = 64;
  while (m < n) m <<= 1;  // padded power of two (<= kSegSortMax by dispatch)
  for (int i = threadIdx.x; i < m; i += T)
    s_v[i] = i < n ? (static_cast<unsigned long long>(descending_key(score[lo + i])) << 32) | static_cast<unsigned>(i) : ~0ull;
  __syncthreads();
  for (int size = 2; size <= m; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < m / 2; t += T) {
        const int i = 2 * t - (t & (stride - 1));   // lower index of the pair
        const int j = i + stride;
        const bool up = (i & size) == 0;
        const unsigned long long a = s_v[i], b = s_v[j];
        if ((a > b) == up) { s_v[i] = b; s_v[j] = a; }
      }
      __syncthreads();
    }
  }
  const int kg = static_cast<int>(k[g] < n ? k[g] : n);
  if constexpr (!FILL) {
    for (int q = threadIdx.x; q < kg; q += T)
      rank_of[lo + static_cast<int64_t>(s_v[q] & 0xFFFFFFFFull)] = static_cast<int32_t>(koff[g] + q);
  } else {
    __shared__ uint32_t s_w[T / 64];
    int* s_q = reinterpret_cast<int*>(s_v + m);  // rank of every node of the graph, -1 = not kept
    for (int i = threadIdx.x; i < n; i += T) s_q[i] = -1;
    __syncthreads();
    for (int q = threadIdx.x; q < kg; q += T) s_q[static_cast<int>(s_v[q] & 0xFFFFFFFFull)] = q;
    __syncthreads();
    const int64_t base = koff[g];
    uint32_t carry = 0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i0 = 0; i0 < n; i0 += T) {
      const int i = i0 + static_cast<int>(threadIdx.x);
      const int q = i < n ? s_q[i] : -1;
      const unsigned long long mk = __ballot(q >= 0);
      if (lane == 0) s_w[w] = __popcll(mk);
      __syncthreads();
      uint32_t before = 0, total = 0;
#pragma unroll
      for (int ww = 0; ww < T / 64; ++ww) {
        const uint32_t c = s_w[ww];
        if (ww < w) before += c;
        total += c;
      }
      const int64_t pos = base + carry + before + __popcll(mk & lanemask_lt());
      if (i < n) {
        if (lift_ptr) lift_ptr[lo + i] = static_cast<int32_t>(pos);
        if (q >= 0) {
          node_index[pos] = lo + i;
          cluster_index[pos] = base + q;
          assign_perm[base + q] = static_cast<int32_t>(pos);
          if (assign_pack || values) {
            const float sc = score[lo + i];
            if (assign_pack) assign_pack[base + q] = make_uint2(static_cast<uint32_t>(lo + i), __float_as_uint(sc));
            if (values) values[pos] = sc;
          }
        }
      }
      carry += total;
      __syncthreads();
    }
  }
}

__global__ __launch_bounds__(256) void topk_count_kernel(const int32_t* __restrict__ rank_of, int64_t n,
                                                         uint32_t* __restrict__ counts) {
  __shared__ uint32_t s_w[4];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kTopkTile;
  uint32_t c = 0;
#pragma unroll
  for (int it = 0; it < kTopkItems; ++it) {
    const int64_t i = base + it * 256 + threadIdx.x;
    c += __popcll(__ballot(i < n && rank_of[i] >= 0));
  }
  if (lane_id() == 0) s_w[wave_id()] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ __launch_bounds__(256) void topk_fill_kernel(const int32_t* __restrict__ rank_of, int64_t n,
                                                        const uint32_t* __restrict__ offsets,
                                                        int64_t* __restrict__ node_index,
                                                        int64_t* __restrict__ cluster_index,
                                                        int32_t* __restrict__ assign_perm,
                                                        const float* __restrict__ score,
                                                        float* __restrict__ values,
                                                        int32_t* __restrict__ lift_ptr,
                                                        uint2* __restrict__ assign_pack,
                                                        uint32_t* __restrict__ member_bits,
                                                        uint32_t* __restrict__ rank128, int64_t dir_blocks) {
  __shared__ uint32_t s_cnt[kTopkItems * 4];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kTopkTile;
  bool flag[kTopkItems];
  int32_t r[kTopkItems];
  uint32_t rank[kTopkItems];
#pragma unroll
  for (int it = 0; it < kTopkItems; ++it) {
    const int64_t i = base + it * 256 + threadIdx.x;
    r[it] = i < n ? rank_of[i] : -1;
    flag[it] = r[it] >= 0;
  }
  uint32_t total;
  block_compact_ranks<kTopkItems>(flag, rank, total, s_cnt);
  const uint32_t off = offsets[blockIdx.x];
  if (member_bits) {
    // r5, by-products for the subgraph Connect (connect/base_conn.py:79-82): the kept-node BITMAP (a wave's ballot is
    // the two words of its 64 consecutive nodes: every word is written whole, no memset, no atomics) and the rank
    // directory rank128[b] = kept nodes with id < 128 b (the exclusive rank this pass has just computed).  With them
    // tgp_connect_subgraph_single needs neither its memset nor its scatter kernel nor the directory scan every
    // workgroup ran: the Connect call is ONE launch.  dir_blocks 128-node blocks are covered (the tiles end earlier or
    // later than that: the last workgroup pads).
#pragma unroll
    for (int it = 0; it < kTopkItems; ++it) {
      const unsigned long long m = __ballot(flag[it]);
      const int64_t i0 = base + it * 256 + (threadIdx.x & ~63);
      if ((threadIdx.x & 63) == 0 && (i0 >> 7) < dir_blocks) {
        member_bits[i0 >> 5] = static_cast<uint32_t>(m);
        member_bits[(i0 >> 5) + 1] = static_cast<uint32_t>(m >> 32);
      }
      const int64_t i = base + it * 256 + threadIdx.x;
      if ((i & 127) == 0 && (i >> 7) < dir_blocks) rank128[i >> 7] = off + rank[it];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
      const int64_t tile_end = base + kTopkTile;  // a multiple of 128
      for (int64_t b = tile_end >> 7; b < dir_blocks; ++b) {
        rank128[b] = off + total;
        for (int q = 0; q < 4; ++q) member_bits[4 * b + q] = 0u;
      }
    }
  }
  if (lift_ptr) {  // CSR offsets of the node -> assignment index: kept nodes are written in node order, so node i's
                   // (at most one) assignment is the number of kept nodes in front of it; perm is the identity
#pragma unroll
    for (int it = 0; it < kTopkItems; ++it) {
      const int64_t i = base + it * 256 + threadIdx.x;
      if (i < n) {
        lift_ptr[i] = static_cast<int32_t>(off + rank[it]);
        if (i == n - 1) lift_ptr[n] = static_cast<int32_t>(off + rank[it] + (flag[it] ? 1u : 0u));
      }
    }
  }
#pragma unroll
  for (int it = 0; it < kTopkItems; ++it) {
    if (!flag[it]) continue;
    const uint32_t j = off + rank[it];
    node_index[j] = base + it * 256 + threadIdx.x;
    cluster_index[j] = r[it];
    assign_perm[r[it]] = static_cast<int32_t>(j);
    if (assign_pack || values) {
      const float sc = score[base + it * 256 + threadIdx.x];
      if (assign_pack)
        assign_pack[r[it]] = make_uint2(static_cast<uint32_t>(base + it * 256 + threadIdx.x), __float_as_uint(sc));
      if (values) values[j] = sc;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------
// Scoring (select/topk_select.py:176: score = (x * w).sum(-1)): one pass over x instead of an [N,F] product
// written and re-read; and its weight gradient dw[f] = sum_i g[i] x[i,f].  Both are pure HBM streams over x.
// G lanes share a row (float4 each when VEC); w lives in registers.
template <int G, bool VEC>
__global__ __launch_bounds__(256) void row_dot_kernel(const float* __restrict__ x, int64_t n, int F, int64_t ldx,
                                                      const float* __restrict__ w, float* __restrict__ out,
                                                      int post) {
  constexpr int PER_WAVE = 64 / G;
  constexpr int MAXC = 8;  // column chunks per lane kept in registers (covers F <= G*4*8)
  const int lane = threadIdx.x & 63, sub = lane % G, slot = lane / G;
  const int64_t wave = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) >> 6;
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * 4;
  constexpr int STEP = VEC ? G * 4 : G;
  float wr[MAXC][VEC ? 4 : 1];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int k = (VEC ? sub * 4 : sub) + c * STEP;
#pragma unroll
    for (int j = 0; j < (VEC ? 4 : 1); ++j) wr[c][j] = (k + j < F) ? w[k + j] : 0.f;
  }
  float nrm = 1.f;
  if (post) {  // TopkSelect's score: act(x.w / ||w||_2), act = identity (1) or tanh (2); every lane group sums all of w
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
#pragma unroll
      for (int j = 0; j < (VEC ? 4 : 1); ++j) sq = fmaf(wr[c][j], wr[c][j], sq);
    for (int k = (VEC ? sub * 4 : sub) + MAXC * STEP; k < F; k += STEP)
#pragma unroll
      for (int j = 0; j < (VEC ? 4 : 1); ++j) sq = fmaf(w[k + j], w[k + j], sq);
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
    nrm = sqrtf(sq);
  }
  for (int64_t base = wave * PER_WAVE; base < n; base += nwaves * PER_WAVE) {
    const int64_t i = base + slot;
    float acc = 0.f;
    if (i < n) {
      const float* a = x + i * ldx;
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        const int k = (VEC ? sub * 4 : sub) + c * STEP;
        if (k >= F) break;
        if constexpr (VEC) {
          const float4 v = *reinterpret_cast<const float4*>(a + k);
          acc = fmaf(v.x, wr[c][0], acc); acc = fmaf(v.y, wr[c][1], acc);
          acc = fmaf(v.z, wr[c][2], acc); acc = fmaf(v.w, wr[c][3], acc);
        } else {
          acc = fmaf(a[k], wr[c][0], acc);
        }
      }
      for (int k = (VEC ? sub * 4 : sub) + MAXC * STEP; k < F; k += STEP) {  // very wide rows
#pragma unroll
        for (int j = 0; j < (VEC ? 4 : 1); ++j) acc = fmaf(a[k + j], w[k + j], acc);
      }
    }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (post) {
      acc = acc / nrm;
      if (post == 2) acc = tanhf(acc);
    }
    if (i < n && sub == 0) out[i] = acc;
  }
}

constexpr int kColsumBlocks = 1024;
// partial[b][f] = sum over the rows of block b of g[i] * x[i][f].  A thread owns one column chunk (a float4 when
// VEC) and every `phases`-th row of the block's slab, four rows in flight; LDS folds the row phases in a fixed
// order.  Columns beyond 256 chunks are covered by an outer sweep.
template <bool VEC>
__global__ __launch_bounds__(256) void weighted_colsum_partial_kernel(const float* __restrict__ x, int64_t n, int F,
                                                                      int64_t ldx, const float* __restrict__ g,
                                                                      float* __restrict__ partial) {
  constexpr int W = VEC ? 4 : 1;
  extern __shared__ float s_acc[];  // [phases][cols * W]
  const int64_t rows_per_block = (n + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
  const int chunks = F / W;
  const int cols = chunks < 256 ? chunks : 256;  // chunks handled per sweep
  const int phases = 256 / cols;                 // rows in flight per sweep
  const int col = threadIdx.x % cols, phase = threadIdx.x / cols;
  for (int c0 = 0; c0 < chunks; c0 += cols) {
    const int c = c0 + col;
    float acc[W];
#pragma unroll
    for (int j = 0; j < W; ++j) acc[j] = 0.f;
    if (phase < phases && c < chunks) {
      const float* xc = x + static_cast<int64_t>(c) * W;
      int64_t i = r0 + phase;
      for (; i + 3 * phases < r1; i += 4 * phases) {
        float gv[4], xv[4][W];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int64_t r = i + static_cast<int64_t>(u) * phases;
          gv[u] = g[r];
          if constexpr (VEC) {
            const float4 v = *reinterpret_cast<const float4*>(xc + r * ldx);
            xv[u][0] = v.x; xv[u][1] = v.y; xv[u][2] = v.z; xv[u][3] = v.w;
          } else {
            xv[u][0] = xc[r * ldx];
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < W; ++j) acc[j] = fmaf(gv[u], xv[u][j], acc[j]);
      }
      for (; i < r1; i += phases) {
        const float gv = g[i];
#pragma unroll
        for (int j = 0; j < W; ++j) acc[j] = fmaf(gv, xc[i * ldx + j], acc[j]);
      }
    }
    if (phase < phases) {
#pragma unroll
      for (int j = 0; j < W; ++j) s_acc[(phase * cols + col) * W + j] = acc[j];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < cols * W; e += 256) {
      if (c0 * W + e < F) {
        float t = 0.f;
        for (int p = 0; p < phases; ++p) t += s_acc[p * cols * W + e];
        partial[static_cast<int64_t>(blockIdx.x) * F + c0 * W + e] = t;
      }
    }
    __syncthreads();
  }
}

// out[f] = sum_b partial[b][f]: one workgroup per column, 256 partial sums in flight, fixed-order tree.
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int blocks, int F,
                                                           float* __restrict__ out) {
  __shared__ float s_t[256];
  const int f = blockIdx.x;
  float t = 0.f;
  for (int b = threadIdx.x; b < blocks; b += 256) t += partial[static_cast<int64_t>(b) * F + f];
  s_t[threadIdx.x] = t;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) s_t[threadIdx.x] += s_t[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[f] = s_t[0];
}

// ---------------------------------------------------------------------------------------------------------
// Backward of TopK pooling's trained path (r5): score t = x w / ||w||, s = act(t) (select/topk_select.py:176-184), the
// kept nodes' scores are the values of S, and x'[c_a] = s_a x[i_a] (reduce/base_reduce.py:141-155 with one assignment
// per supernode).  From g' = dL/dx' [K,F] and ge = dL/d(values of S) [K] (either may be absent), per assignment a:
//   gv = ge_a + <g'[c_a], x[i_a]>,  g_t = gv (1 - s_a^2) (tanh) or gv,  g~ = g_t / ||w||,  t_a = <x[i_a], w> / ||w||
//   gx[i_a] = s_a g'[c_a] + g~ w          (rows of nodes that were not kept stay zero)
//   gw      = sum_a g~ x[i_a] - (sum_a g~ t_a) w / ||w||
// where the stock autograd graph runs ~18 launches over [N] and [N,F] temporaries.  Eight lanes share a row (16 bytes
// each, CPL chunks per lane: F <= 32 CPL), two rows per lane group in flight; the gw sums stay in registers per lane,
// are folded over the wave's eight groups by shuffles, over the four waves through LDS in wave order, and leave as
// one partial per workgroup; topk_pool_bwd_final_kernel adds the partials in workgroup order (deterministic).
struct TopkPoolBwdArgs {
  const float* x; int64_t ldx;
  const int64_t* node; const int64_t* cluster;  // [K] i_a, c_a (cluster NULL: c_a = a)
  const float* values;                          // [K] s_a
  const float* g_xpool;                         // [K,F] contiguous, or NULL
  const float* g_values;                        // [K], or NULL
  const float* w;                               // [F]
  float* gx;                                    // [N,F] contiguous, zeroed; or NULL
  float* part;                                  // [gridDim.x][F + 1]; or NULL
  int64_t K; int F; int use_tanh;
};
constexpr int TPB_MAX_GRID = 256;

template <int CPL>
__global__ __launch_bounds__(256) void topk_pool_bwd_kernel(TopkPoolBwdArgs p) {
  __shared__ float s_red[4][CPL * 32 + 1];
  const int tid = threadIdx.x, lane = tid & 63, wv_id = tid >> 6, sub = lane & 7, grp = tid >> 3;
  const int F = p.F, chunks = F >> 2;
  float4 wr[CPL];
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int c = sub + 8 * j;
    wr[j] = c < chunks ? *reinterpret_cast<const float4*>(p.w + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
    sq = fmaf(wr[j].x, wr[j].x, sq); sq = fmaf(wr[j].y, wr[j].y, sq);
    sq = fmaf(wr[j].z, wr[j].z, sq); sq = fmaf(wr[j].w, wr[j].w, sq);
  }
#pragma unroll
  for (int off = 4; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
  const float inv = 1.f / sqrtf(sq);
  float4 acc[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  float accc = 0.f;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 32;
  for (int64_t a0 = static_cast<int64_t>(blockIdx.x) * 32 + grp; a0 < p.K; a0 += 2 * stride) {
    // two assignments per lane group: every load of both is requested before the first sum
    int64_t node[2], clu[2];
    float sv[2], ge[2];
    bool ok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t a = a0 + u * stride;
      ok[u] = a < p.K;
      const int64_t aa = ok[u] ? a : a0;
      node[u] = p.node[aa];
      clu[u] = p.cluster ? p.cluster[aa] : aa;
      sv[u] = p.values[aa];
      ge[u] = p.g_values ? p.g_values[aa] : 0.f;
    }
    float4 xr[2][CPL], gp[2][CPL];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const int c = sub + 8 * j;
        const bool in = c < chunks;
        xr[u][j] = in ? *reinterpret_cast<const float4*>(p.x + node[u] * p.ldx + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
        gp[u][j] = (in && p.g_xpool) ? *reinterpret_cast<const float4*>(p.g_xpool + clu[u] * F + 4 * c)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float d1 = 0.f, d2 = 0.f;
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        d1 = fmaf(gp[u][j].x, xr[u][j].x, d1); d1 = fmaf(gp[u][j].y, xr[u][j].y, d1);
        d1 = fmaf(gp[u][j].z, xr[u][j].z, d1); d1 = fmaf(gp[u][j].w, xr[u][j].w, d1);
        d2 = fmaf(xr[u][j].x, wr[j].x, d2); d2 = fmaf(xr[u][j].y, wr[j].y, d2);
        d2 = fmaf(xr[u][j].z, wr[j].z, d2); d2 = fmaf(xr[u][j].w, wr[j].w, d2);
      }
#pragma unroll
      for (int off = 4; off > 0; off >>= 1) {
        d1 += __shfl_xor(d1, off);
        d2 += __shfl_xor(d2, off);
      }
      const float gv = ge[u] + d1;
      const float gt = p.use_tanh ? gv * (1.f - sv[u] * sv[u]) : gv;
      const float gtn = ok[u] ? gt * inv : 0.f;
      const float t = d2 * inv;
      accc = fmaf(gtn, t, accc);
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        acc[j].x = fmaf(gtn, xr[u][j].x, acc[j].x); acc[j].y = fmaf(gtn, xr[u][j].y, acc[j].y);
        acc[j].z = fmaf(gtn, xr[u][j].z, acc[j].z); acc[j].w = fmaf(gtn, xr[u][j].w, acc[j].w);
        const int c = sub + 8 * j;
        if (p.gx && ok[u] && c < chunks) {
          float4 o;
          o.x = fmaf(gp[u][j].x, sv[u], gtn * wr[j].x); o.y = fmaf(gp[u][j].y, sv[u], gtn * wr[j].y);
          o.z = fmaf(gp[u][j].z, sv[u], gtn * wr[j].z); o.w = fmaf(gp[u][j].w, sv[u], gtn * wr[j].w);
          *reinterpret_cast<float4*>(p.gx + node[u] * F + 4 * c) = o;
        }
      }
    }
  }
  if (!p.part) return;
  // the wave's eight lane groups (lane bits 3..5), then the four waves in wave order
#pragma unroll
  for (int off = 8; off < 64; off <<= 1) {
    accc += __shfl_xor(accc, off);
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      acc[j].x += __shfl_xor(acc[j].x, off); acc[j].y += __shfl_xor(acc[j].y, off);
      acc[j].z += __shfl_xor(acc[j].z, off); acc[j].w += __shfl_xor(acc[j].w, off);
    }
  }
  if (lane < 8) {
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      float* d = &s_red[wv_id][4 * (sub + 8 * j)];
      d[0] = acc[j].x; d[1] = acc[j].y; d[2] = acc[j].z; d[3] = acc[j].w;
    }
    if (lane == 0) s_red[wv_id][CPL * 32] = accc;
  }
  __syncthreads();
  float* out = p.part + static_cast<int64_t>(blockIdx.x) * (F + 1);
  for (int f = tid; f <= F; f += 256) {
    const int src = f < F ? f : CPL * 32;
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) t = __fadd_rn(t, s_red[q][src]);
    out[f] = t;
  }
}

// gw[f] = sum_b part[b][f] - (sum_b part[b][F]) w[f] / ||w||: wave q adds partials q, q + 8, ... (all of a wave's loads
// requested together), the eight sums are added in wave order.  One workgroup; F <= 256.
__global__ __launch_bounds__(512) void topk_pool_bwd_final_kernel(const float* __restrict__ part, int P, int F,
                                                                  const float* __restrict__ w, float* __restrict__ gw) {
  __shared__ float s_sum[8][320];
  __shared__ float s_tot[320];
  __shared__ float s_inv;
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6, n_out = F + 1;
  for (int o0 = 0; o0 < n_out; o0 += 64) {
    const int o = o0 + lane;
    float v[TPB_MAX_GRID / 8];
#pragma unroll
    for (int ji = 0; ji < TPB_MAX_GRID / 8; ++ji) {
      const int j = q + 8 * ji;
      const bool ok = o < n_out && j < P;
      const float got = part[ok ? static_cast<long>(j) * n_out + o : 0];
      v[ji] = ok ? got : 0.f;
    }
    float a = 0.f;
#pragma unroll
    for (int ji = 0; ji < TPB_MAX_GRID / 8; ++ji) a = __fadd_rn(a, v[ji]);
    if (o < n_out) s_sum[q][o] = a;
  }
  if (q == 0) {
    float sq = 0.f;
    for (int f = lane; f < F; f += 64) sq = fmaf(w[f], w[f], sq);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
    if (lane == 0) s_inv = 1.f / sqrtf(sq);
  }
  __syncthreads();
  for (int o = threadIdx.x; o < n_out; o += 512) {
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) r = __fadd_rn(r, s_sum[k][o]);
    s_tot[o] = r;
  }
  __syncthreads();
  for (int f = threadIdx.x; f < F; f += 512) gw[f] = s_tot[f] - s_tot[F] * s_inv * w[f];
}

template <int G, bool VEC>
static void launch_row_dot(const float* x, int64_t n, int F, int64_t ldx, const float* w, float* out,
                           hipStream_t stream, int post = 0) {
  int64_t blocks = cdiv(n, static_cast<int64_t>(4) * (64 / G));
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL((row_dot_kernel<G, VEC>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, x, n, F, ldx,
                     w, out, post);
}

// k[g] = ceil(ratio * n_g) in fp32 (ratio < 1; exactly PyG's `(ratio * num_nodes.to(float)).ceil()`) or
// min(ratio, n_g) (ratio >= 1), and koff = its exclusive prefix sums: one launch instead of eight tiny tensor ops.
__global__ __launch_bounds__(1024) void topk_plan_kernel(const int64_t* __restrict__ sizes, int64_t B, float ratio,
                                                         int64_t* __restrict__ k, int64_t* __restrict__ koff) {
  __shared__ long long s_w[16];
  __shared__ long long s_carry;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < B; base += 1024) {
    const int64_t g = base + tid;
    long long v = 0;
    if (g < B) {
      const long long n = sizes[g];
      if (ratio >= 1.0f) {
        const long long r = static_cast<long long>(ratio);
        v = r < n ? r : n;
      } else {
        v = static_cast<long long>(ceilf(__fmul_rn(ratio, static_cast<float>(n))));
      }
      k[g] = v;
    }
    long long inc = v;  // inclusive scan over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const long long o = __shfl_up(inc, off, WAVE);
      if (lane >= off) inc += o;
    }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    long long before = s_carry, tot = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const long long c = s_w[q];
      if (q < w) before += c;
      tot += c;
    }
    if (g < B) koff[g] = before + inc - v;
    __syncthreads();
    if (tid == 0) s_carry += tot;
    __syncthreads();
  }
  if (tid == 0) koff[B] = s_carry;
}

struct TopkLayout {
  uint64_t *k0, *k1;
  uint32_t *v0, *v1, *scratch, *counts, *offsets;
  int32_t* rank_of;
  int64_t* total;
  size_t bytes;
};

static TopkLayout topk_layout(void* ws, int64_t n) {
  Carver cv(ws);
  TopkLayout s;
  const size_t m = static_cast<size_t>(n > 0 ? n : 1);
  const size_t nb = static_cast<size_t>(cdiv(n > 0 ? n : 1, kTopkTile));
  s.k0 = cv.take<uint64_t>(m);
  s.k1 = cv.take<uint64_t>(m);
  s.v0 = cv.take<uint32_t>(m);
  s.v1 = cv.take<uint32_t>(m);
  s.scratch = cv.take<uint32_t>(sort_scratch_words());
  s.rank_of = cv.take<int32_t>(m);
  s.counts = cv.take<uint32_t>(nb);
  s.offsets = cv.take<uint32_t>(nb);
  s.total = cv.take<int64_t>(1);
  s.bytes = cv.off;
  return s;
}

// ------------------------------------------------------------------------------------------------------------
// min_score mode (select/topk_select.py:186-194): score = per-graph softmax of x.w (PyG utils.softmax: max
// subtraction, +1e-16 in the denominator), kept = score > min(max_g(score) - tol, min_score) (PyG topk, tol 1e-7),
// in ascending node order (nonzero()).  One workgroup per graph of the sorted batch walks its nodes three times (max,
// sum of exponentials, probabilities + keep flags; the scores stay in L2): batches of small graphs are one launch,
// a single large graph is one workgroup's stream.  Sums are taken in a fixed order (strided partial sums, then waves).
__device__ __forceinline__ float ms_block_max(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}
__device__ __forceinline__ float ms_block_sum(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void topk_minscore_kernel(const float* __restrict__ score,
                                                            const int64_t* __restrict__ ptr, float min_score,
                                                            float tol, float* __restrict__ prob,
                                                            uint8_t* __restrict__ keep,
                                                            uint32_t* __restrict__ counts) {
  __shared__ float sh[4];
  const int g = blockIdx.x;
  const int64_t p0 = ptr[g], p1 = ptr[g + 1];
  if (p1 <= p0) {
    if (threadIdx.x == 0) counts[g] = 0;
    return;
  }
  float mx = -INFINITY;
  for (int64_t i = p0 + threadIdx.x; i < p1; i += 256) mx = fmaxf(mx, score[i]);
  mx = ms_block_max(mx, sh);
  float sum = 0.f;
  for (int64_t i = p0 + threadIdx.x; i < p1; i += 256) sum += expf(score[i] - mx);
  sum = ms_block_sum(sum, sh);
  const float den = sum + 1e-16f;
  const float floor_ = fminf(1.0f / den - tol, min_score);  // max probability = exp(0) / den
  uint32_t mine = 0;
  for (int64_t i = p0 + threadIdx.x; i < p1; i += 256) {
    const float pr = expf(score[i] - mx) / den;
    prob[i] = pr;
    const bool k = pr > floor_;
    keep[i] = k ? 1 : 0;
    mine += k ? 1u : 0u;
  }
  const float total = ms_block_sum(static_cast<float>(mine), sh);  // (< 2^24 per graph: exact; larger graphs below)
  if (p1 - p0 < (1 << 24)) {
    if (threadIdx.x == 0) counts[g] = static_cast<uint32_t>(total);
  } else {
    __shared__ uint32_t s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) counts[g] = s_cnt;
  }
}

// kept nodes of graph g, ascending, at out[off[g] ..]
__global__ __launch_bounds__(256) void topk_minscore_fill_kernel(const uint8_t* __restrict__ keep,
                                                                 const int64_t* __restrict__ ptr,
                                                                 const uint32_t* __restrict__ off,
                                                                 int64_t* __restrict__ node_index) {
  __shared__ uint32_t s_w[4];
  const int g = blockIdx.x;
  const int64_t p0 = ptr[g], p1 = ptr[g + 1];
  uint32_t base = off[g];
  for (int64_t c0 = p0; c0 < p1; c0 += 256) {
    const int64_t i = c0 + threadIdx.x;
    const bool k = i < p1 && keep[i] != 0;
    const unsigned long long m = __ballot(k);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_w[w] = __popcll(m);
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q < w) before += s_w[q];
      all += s_w[q];
    }
    if (k) node_index[base + before + __popcll(m & lanemask_lt())] = i;
    base += all;
  }
}

}  // namespace tgp

using namespace tgp;

extern "C" size_t tgp_topk_select_workspace_bytes(int64_t N) { return topk_layout(nullptr, N).bytes + 256; }

extern "C" int64_t tgp_topk_select_directory_blocks(int64_t N) { return ((N > 0 ? N : 1) / 32 + 1 + 3) / 4; }

extern "C" int tgp_topk_select(const float* score, const int64_t* batch, int64_t N, int64_t B, const int64_t* ptr,
                               const int64_t* k, const int64_t* koff, int64_t segments_max_nodes, void* ws,
                               size_t ws_bytes,
                               int64_t* node_index, int64_t* cluster_index, int32_t* assign_perm, float* values,
                               int32_t* lift_row_ptr, uint64_t* assign_pack_, uint32_t* member_bits,
                               uint32_t* rank128, int* directory_written, void* stream_) {
  if (directory_written) *directory_written = 0;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  uint2* assign_pack = reinterpret_cast<uint2*>(assign_pack_);
  TGP_REQUIRE(N >= 0 && B >= 0, TGP_ERR_INVALID, "tgp_topk_select: negative size");
  if (N == 0 || B == 0) return TGP_OK;
  TGP_REQUIRE(N < (1ll << 31), TGP_ERR_RANGE, "tgp_topk_select: more than 2^31 nodes");
  TGP_REQUIRE(score && ptr && k && koff && (batch || B == 1), TGP_ERR_INVALID, "tgp_topk_select: null pointer");
  TGP_REQUIRE(ws && ws_bytes >= tgp_topk_select_workspace_bytes(N), TGP_ERR_WORKSPACE,
              "tgp_topk_select: workspace too small");
  const TopkLayout s = topk_layout(ws, N);
  const int nb256 = cdiv(N, 256), nbt = cdiv(N, kTopkTile);
  if (segments_max_nodes > 0 && segments_max_nodes <= 64 && node_index) {
    hipLaunchKernelGGL(topk_segsort_wave_fill_kernel, dim3(cdiv(B, 4)), dim3(256), 0, stream, score, ptr, k, koff, B, N,
                       node_index, cluster_index, assign_perm, values, lift_row_ptr, assign_pack);
    return check_launch("tgp_topk_select");
  }
  if (segments_max_nodes > 64 && segments_max_nodes <= kSegSortLarge && node_index) {  // the same, one workgroup per graph
    int m = 64;
    while (m < segments_max_nodes) m <<= 1;
    const size_t lds = static_cast<size_t>(m) * (sizeof(unsigned long long) + sizeof(int));
    if (segments_max_nodes <= kSegSortMax) {
      hipLaunchKernelGGL((topk_segsort_block_kernel<256, true>), dim3(static_cast<unsigned>(B)), dim3(256), lds, stream,
                         score, ptr, k, koff, s.rank_of, B, N, node_index, cluster_index, assign_perm, values,
                         lift_row_ptr, assign_pack, m);
      return check_launch("tgp_topk_select");
    }
    // 96 KB of dynamic LDS at 8192 nodes: above the default limit -- if the device does not grant it, the rank-table
    // route below takes the call (its sort tile is 64 KB)
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(topk_segsort_block_kernel<1024, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) == hipSuccess) {
      hipLaunchKernelGGL((topk_segsort_block_kernel<1024, true>), dim3(static_cast<unsigned>(B)), dim3(1024), lds, stream,
                         score, ptr, k, koff, s.rank_of, B, N, node_index, cluster_index, assign_perm, values,
                         lift_row_ptr, assign_pack, m);
      return check_launch("tgp_topk_select");
    }
    (void)hipGetLastError();
  }
  (void)hipMemsetAsync(s.rank_of, 0xFF, static_cast<size_t>(N) * sizeof(int32_t), stream);
  if (segments_max_nodes > 0 && segments_max_nodes <= 64) {
    hipLaunchKernelGGL(topk_segsort_wave_kernel, dim3(cdiv(B, 4)), dim3(256), 0, stream, score, ptr, k, koff, B,
                       s.rank_of);
  } else if (segments_max_nodes > 0 && segments_max_nodes <= kSegSortMax) {
    hipLaunchKernelGGL((topk_segsort_block_kernel<256, false>), dim3(static_cast<unsigned>(B)), dim3(256),
                       kSegSortMax * sizeof(unsigned long long), stream, score, ptr, k, koff, s.rank_of, B, N,
                       node_index, cluster_index, assign_perm, values, lift_row_ptr, assign_pack, kSegSortMax);
  } else if (segments_max_nodes > 0 && segments_max_nodes <= kSegSortLarge) {
    int m = 64;
    while (m < segments_max_nodes) m <<= 1;
    const size_t lds = static_cast<size_t>(m) * sizeof(unsigned long long);
    TGP_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(topk_segsort_block_kernel<1024, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) == hipSuccess,
                TGP_ERR_LAUNCH, "tgp_topk_select: the device does not grant %zu bytes of dynamic LDS", lds);
    hipLaunchKernelGGL((topk_segsort_block_kernel<1024, false>), dim3(static_cast<unsigned>(B)), dim3(1024), lds, stream,
                       score, ptr, k, koff, s.rank_of, B, N, node_index, cluster_index, assign_perm, values,
                       lift_row_ptr, assign_pack, m);
  } else {
    hipLaunchKernelGGL(topk_keys_kernel, dim3(nb256), dim3(256), 0, stream, score, batch, N, s.k0, s.v0);
    bool first = true;
    const int key_bits = 32 + (B > 1 ? bits_for(static_cast<uint64_t>(B - 1)) : 0);
    const int rc = radix_sort_pairs<uint64_t, uint32_t>(s.k0, s.v0, s.k1, s.v1, N, key_bits, s.scratch, stream, &first);
    if (rc != TGP_OK) return rc;
    hipLaunchKernelGGL(topk_rank_kernel, dim3(nb256), dim3(256), 0, stream, first ? s.k0 : s.k1, first ? s.v0 : s.v1,
                       N, ptr, k, koff, s.rank_of);
  }
  hipLaunchKernelGGL(topk_count_kernel, dim3(nbt), dim3(256), 0, stream, s.rank_of, N, s.counts);
  hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, stream, s.counts, nbt, s.offsets, s.total,
                     static_cast<const int*>(nullptr));
  if (node_index) {
    const bool dir = member_bits && rank128 && directory_written;
    hipLaunchKernelGGL(topk_fill_kernel, dim3(nbt), dim3(256), 0, stream, s.rank_of, N, s.offsets, node_index,
                       cluster_index, assign_perm, score, values, lift_row_ptr, assign_pack,
                       dir ? member_bits : nullptr, dir ? rank128 : nullptr, tgp_topk_select_directory_blocks(N));
    if (dir) *directory_written = 1;  // (the per-graph sort routes above do not write it: small graphs take the fused path)
  }
  return check_launch("tgp_topk_select");
}

static int row_dot_dispatch(const float* x, int64_t N, int64_t F, int64_t ldx, const float* w, float* out, int post,
                            hipStream_t stream, const char* who) {
  TGP_REQUIRE(N >= 0 && F >= 0 && ldx >= F, TGP_ERR_INVALID, "row dot: bad size");
  if (N == 0) return TGP_OK;
  TGP_REQUIRE(out && (F == 0 || (x && w)), TGP_ERR_INVALID, "row dot: null pointer");
  TGP_REQUIRE(F < (1ll << 31), TGP_ERR_RANGE, "row dot: F too large");
  const int f = static_cast<int>(F);
  const bool vec = (F % 4 == 0) && (ldx % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0);
  const int64_t units = vec ? F / 4 : F;
  if (vec) {
    if (units <= 1) launch_row_dot<1, true>(x, N, f, ldx, w, out, stream, post);
    else if (units <= 2) launch_row_dot<2, true>(x, N, f, ldx, w, out, stream, post);
    else if (units <= 4) launch_row_dot<4, true>(x, N, f, ldx, w, out, stream, post);
    else if (units <= 8) launch_row_dot<8, true>(x, N, f, ldx, w, out, stream, post);
    else if (units <= 16) launch_row_dot<16, true>(x, N, f, ldx, w, out, stream, post);
    else if (units <= 32) launch_row_dot<32, true>(x, N, f, ldx, w, out, stream, post);
    else launch_row_dot<64, true>(x, N, f, ldx, w, out, stream, post);
  } else {
    if (units <= 4) launch_row_dot<4, false>(x, N, f, ldx, w, out, stream, post);
    else if (units <= 16) launch_row_dot<16, false>(x, N, f, ldx, w, out, stream, post);
    else launch_row_dot<64, false>(x, N, f, ldx, w, out, stream, post);
  }
  return check_launch(who);
}

extern "C" int tgp_row_dot_f32(const float* x, int64_t N, int64_t F, int64_t ldx, const float* w, float* out,
                               void* stream_) {
  return row_dot_dispatch(x, N, F, ldx, w, out, 0, static_cast<hipStream_t>(stream_), "tgp_row_dot_f32");
}

// TopkSelect's whole score in the same pass (select/topk_select.py:176-184, ratio mode): out[i] = act(<x[i,:], w> /
// ||w||_2), act = identity (0) or tanh (1); the norm is summed by every wave from the copy of w it holds in registers.
extern "C" int tgp_topk_score_f32(const float* x, int64_t N, int64_t F, int64_t ldx, const float* w, int act,
                                  float* out, void* stream_) {
  TGP_REQUIRE(act == 0 || act == 1, TGP_ERR_INVALID, "tgp_topk_score_f32: act must be 0 (identity) or 1 (tanh)");
  return row_dot_dispatch(x, N, F, ldx, w, out, 1 + act, static_cast<hipStream_t>(stream_), "tgp_topk_score_f32");
}

extern "C" size_t tgp_weighted_colsum_workspace_bytes(int64_t F) {
  return align_up(static_cast<size_t>(kColsumBlocks) * (F > 0 ? F : 1) * sizeof(float)) + 256;
}

extern "C" int tgp_weighted_colsum_f32(const float* x, int64_t N, int64_t F, int64_t ldx, const float* g, float* out,
                                       void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(N >= 0 && F >= 0 && ldx >= F, TGP_ERR_INVALID, "tgp_weighted_colsum_f32: bad size");
  if (F == 0) return TGP_OK;
  TGP_REQUIRE(out, TGP_ERR_INVALID, "tgp_weighted_colsum_f32: null output");
  if (N == 0) {
    (void)hipMemsetAsync(out, 0, static_cast<size_t>(F) * sizeof(float), stream);
    return check_launch("tgp_weighted_colsum_f32");
  }
  TGP_REQUIRE(x && g, TGP_ERR_INVALID, "tgp_weighted_colsum_f32: null pointer");
  TGP_REQUIRE(F < (1ll << 24), TGP_ERR_RANGE, "tgp_weighted_colsum_f32: F too large");
  TGP_REQUIRE(ws && ws_bytes >= tgp_weighted_colsum_workspace_bytes(F), TGP_ERR_WORKSPACE,
              "tgp_weighted_colsum_f32: workspace too small");
  int blocks = cdiv(N, 256);
  if (blocks > kColsumBlocks) blocks = kColsumBlocks;
  float* partial = static_cast<float*>(ws);
  const bool vec = (F % 4 == 0) && (ldx % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0);
  const int chunks = static_cast<int>(vec ? F / 4 : F);
  const int cols = chunks < 256 ? chunks : 256;
  const size_t lds = static_cast<size_t>(256 / cols) * cols * (vec ? 4 : 1) * sizeof(float);
  if (vec)
    hipLaunchKernelGGL(weighted_colsum_partial_kernel<true>, dim3(blocks), dim3(256), lds, stream, x, N,
                       static_cast<int>(F), ldx, g, partial);
  else
    hipLaunchKernelGGL(weighted_colsum_partial_kernel<false>, dim3(blocks), dim3(256), lds, stream, x, N,
                       static_cast<int>(F), ldx, g, partial);
  hipLaunchKernelGGL(colsum_final_kernel, dim3(static_cast<unsigned>(F)), dim3(256), 0, stream, partial, blocks,
                     static_cast<int>(F), out);
  return check_launch("tgp_weighted_colsum_f32");
}

extern "C" int tgp_topk_pool_bwd_fits(int64_t F) { return (F > 0 && F % 4 == 0 && F <= 256) ? 1 : 0; }

extern "C" size_t tgp_topk_pool_bwd_workspace_bytes(int64_t F) {
  return static_cast<size_t>(TPB_MAX_GRID) * static_cast<size_t>(F + 1) * sizeof(float) + 256;
}

// x [N,F] (row stride ldx, rows 16-byte aligned), node / cluster / values [K] = the assignments of the one-to-one S
// (cluster NULL: 0..K-1), g_xpool [K,F] contiguous or NULL, g_values [K] or NULL, w [F].  gx [N,F] contiguous (NULL: not
// wanted) is written entirely -- zero rows for nodes that were not kept; gw [F] (NULL: not wanted).
extern "C" int tgp_topk_pool_bwd_f32(const float* x, int64_t N, int64_t F, int64_t ldx, const int64_t* node,
                                     const int64_t* cluster, const float* values, int64_t K, const float* g_xpool,
                                     const float* g_values, const float* w, int use_tanh, float* gx, float* gw,
                                     void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(N >= 0 && K >= 0 && tgp_topk_pool_bwd_fits(F) && ldx >= F && ldx % 4 == 0, TGP_ERR_INVALID,
              "tgp_topk_pool_bwd_f32: bad size (F must be a multiple of 4, at most 256)");
  TGP_REQUIRE(w && (K == 0 || (x && node && values)), TGP_ERR_INVALID, "tgp_topk_pool_bwd_f32: null pointer");
  TGP_REQUIRE((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(gx) |
               reinterpret_cast<uintptr_t>(g_xpool)) % 16 == 0,
              TGP_ERR_INVALID, "tgp_topk_pool_bwd_f32: x, w, gx and g_xpool must be 16-byte aligned");
  TGP_REQUIRE(!gw || (ws && ws_bytes >= tgp_topk_pool_bwd_workspace_bytes(F)), TGP_ERR_WORKSPACE,
              "tgp_topk_pool_bwd_f32: workspace too small");
  if (gx && N > 0) (void)hipMemsetAsync(gx, 0, sizeof(float) * static_cast<size_t>(N) * static_cast<size_t>(F), stream);
  if (K == 0) {
    if (gw) (void)hipMemsetAsync(gw, 0, sizeof(float) * static_cast<size_t>(F), stream);
    return check_launch("tgp_topk_pool_bwd_f32");
  }
  int64_t grid = cdiv(K, static_cast<int64_t>(64));  // two assignments per lane group and round
  if (grid > TPB_MAX_GRID) grid = TPB_MAX_GRID;
  TopkPoolBwdArgs a{x, ldx, node, cluster, values, g_xpool, g_values, w, gx, gw ? static_cast<float*>(ws) : nullptr,
                    K, static_cast<int>(F), use_tanh};
  const dim3 g(static_cast<unsigned>(grid)), b(256);
  if (F <= 32) hipLaunchKernelGGL(topk_pool_bwd_kernel<1>, g, b, 0, stream, a);
  else if (F <= 64) hipLaunchKernelGGL(topk_pool_bwd_kernel<2>, g, b, 0, stream, a);
  else if (F <= 128) hipLaunchKernelGGL(topk_pool_bwd_kernel<4>, g, b, 0, stream, a);
  else hipLaunchKernelGGL(topk_pool_bwd_kernel<8>, g, b, 0, stream, a);
  if (gw)
    hipLaunchKernelGGL(topk_pool_bwd_final_kernel, dim3(1), dim3(512), 0, stream, static_cast<const float*>(ws),
                       static_cast<int>(grid), static_cast<int>(F), w, gw);
  return check_launch("tgp_topk_pool_bwd_f32");
}

extern "C" int tgp_topk_plan(const int64_t* sizes, int64_t B, double ratio, int64_t* k, int64_t* koff, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(B >= 0 && ratio > 0.0, TGP_ERR_INVALID, "tgp_topk_plan: bad argument");
  TGP_REQUIRE(koff && (B == 0 || (sizes && k)), TGP_ERR_INVALID, "tgp_topk_plan: null pointer");
  hipLaunchKernelGGL(topk_plan_kernel, dim3(1), dim3(1024), 0, stream, sizes, B, static_cast<float>(ratio), k, koff);
  return check_launch("tgp_topk_plan");
}

// ---------------------------------------------------------------------------- min_score mode
extern "C" size_t tgp_topk_minscore_workspace_bytes(int64_t N, int64_t B) {
  return align_up(static_cast<size_t>(N > 0 ? N : 1)) + 2 * align_up(static_cast<size_t>(B + 1) * sizeof(uint32_t)) + 256;
}

extern "C" int tgp_topk_minscore_count(const float* score, const int64_t* ptr, int64_t N, int64_t B, float min_score,
                                       float tol, float* prob, void* ws, size_t ws_bytes, int64_t* d_count,
                                       void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(N >= 0 && B >= 0 && d_count, TGP_ERR_INVALID, "tgp_topk_minscore_count: bad argument");
  if (N == 0 || B == 0) {
    (void)hipMemsetAsync(d_count, 0, sizeof(int64_t), stream);
    return check_launch("tgp_topk_minscore_count");
  }
  TGP_REQUIRE(score && ptr && prob && ws, TGP_ERR_INVALID, "tgp_topk_minscore_count: null pointer");
  TGP_REQUIRE(N < (1ll << 32) && B < (1ll << 31), TGP_ERR_RANGE, "tgp_topk_minscore_count: too large");
  TGP_REQUIRE(ws_bytes >= tgp_topk_minscore_workspace_bytes(N, B), TGP_ERR_WORKSPACE,
              "tgp_topk_minscore_count: workspace too small");
  Carver cv(ws);
  uint8_t* keep = cv.take<uint8_t>(N);
  uint32_t* counts = cv.take<uint32_t>(B + 1);
  uint32_t* off = cv.take<uint32_t>(B + 1);
  hipLaunchKernelGGL(topk_minscore_kernel, dim3(static_cast<unsigned>(B)), dim3(256), 0, stream, score, ptr, min_score,
                     tol, prob, keep, counts);
  hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, stream, counts, static_cast<int>(B), off, d_count,
                     static_cast<const int*>(nullptr));
  return check_launch("tgp_topk_minscore_count");
}

extern "C" int tgp_topk_minscore_fill(const void* ws, const int64_t* ptr, int64_t N, int64_t B, int64_t num_out,
                                      int64_t* node_index, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(ws && ptr && N >= 0 && B >= 0 && num_out >= 0, TGP_ERR_INVALID, "tgp_topk_minscore_fill: bad argument");
  if (num_out == 0 || B == 0) return TGP_OK;
  TGP_REQUIRE(node_index, TGP_ERR_INVALID, "tgp_topk_minscore_fill: null output");
  Carver cv(const_cast<void*>(ws));
  const uint8_t* keep = cv.take<uint8_t>(N);
  (void)cv.take<uint32_t>(B + 1);
  const uint32_t* off = cv.take<uint32_t>(B + 1);
  hipLaunchKernelGGL(topk_minscore_fill_kernel, dim3(static_cast<unsigned>(B)), dim3(256), 0, stream, keep, ptr, off,
                     node_index);
  return check_launch("tgp_topk_minscore_fill");
}
