// A14: GraclusSelect's matching (reference select/graclus_select.py:62-81 calls torch_cluster 1.6.3
// `graclus_cluster`, a third-party randomised greedy matching: every node is paired with its heaviest
// unmatched neighbour).  The exact pairs are not a contract (torch_cluster's own result depends on a random
// node permutation); what is: a maximal matching that prefers heavy edges, labelled with the smaller id of
// the pair.  This file is the build's own data-parallel "handshake" version of that algorithm:
//
//   every round, each free node proposes to the free neighbour with the largest edge key
//        key(i,j) = (w_ij, hash(min(i,j), max(i,j)), min, max)        -- a strict total order on edges
//   and mutual proposals are matched.  The largest live edge is always a mutual proposal, so every round
//   makes progress; the symmetric hash breaks weight ties pseudo-randomly, which is what keeps the number of
//   rounds logarithmic (smallest-id tie-breaking degenerates to one match per round on a path).
//
// Layout: the edge list is grouped by source node through the inverted index that tgp_assign_index_build makes
// of `row` (any edge order), gathered once into CSR arrays (int32 neighbour, fp32 weight); per-node state is one
// byte (free) + one int32 (proposal) + the int64 label.  A round streams the CSR once: HBM-bound, E * 8 bytes.
#include "common.h"
#include "primitives.h"
#include "lookback.h"

namespace tgp {

__device__ __forceinline__ uint32_t pair_hash(uint32_t a, uint32_t b) {  // symmetric by construction (a <= b)
  uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}

__device__ __forceinline__ unsigned long long entry_hash64(uint32_t a, uint32_t b, uint32_t wbits) {
  unsigned long long h = (static_cast<unsigned long long>(a) << 32 | b) * 0x9E3779B97F4A7C15ull;
  h ^= h >> 29; h += wbits; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32; h *= 0x94D049BB133111EBull; h ^= h >> 29;
  return h;
}

// Gathers the CSR (source, neighbour, weight per entry) and, on the way, a fingerprint of the list's symmetry:
// sum over the entries of sign(i - j) * hash(min, max, weight bits) in 64-bit wrap-around arithmetic.  Every
// entry that has a reverse with the same weight cancels exactly, so the sum is 0 for an undirected, symmetrically
// weighted list (what PyG hands out) and non-zero otherwise except with probability 2^-64 -- which lets the
// common case skip the reverse-edge search of gm_symmetrise_kernel.
__global__ __launch_bounds__(256) void gm_csr_gather_kernel(const int64_t* __restrict__ row,
                                                            const int64_t* __restrict__ col,
                                                            const float* __restrict__ w,
                                                            const int32_t* __restrict__ perm, int64_t E,
                                                            int32_t* __restrict__ src, int32_t* __restrict__ nbr,
                                                            float* __restrict__ wt,
                                                            unsigned long long* __restrict__ partial) {
  __shared__ unsigned long long s_sum[4];
  const int64_t p = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  unsigned long long term = 0;
  if (p < E) {
    const int64_t e = perm ? perm[p] : p;  // perm == NULL: the list is already grouped by source
    const int32_t i = static_cast<int32_t>(row[e]), j = static_cast<int32_t>(col[e]);
    const float x = w ? w[e] : 1.0f;
    src[p] = i;
    nbr[p] = j;
    wt[p] = x;
    if (i != j) {
      const unsigned long long h = entry_hash64(static_cast<uint32_t>(i < j ? i : j), static_cast<uint32_t>(i < j ? j : i),
                                                __float_as_uint(x));
      term = i < j ? h : 0ull - h;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned lo = __shfl_xor(static_cast<unsigned>(term), off, WAVE);
    const unsigned hi = __shfl_xor(static_cast<unsigned>(term >> 32), off, WAVE);
    term += (static_cast<unsigned long long>(hi) << 32) | lo;
  }
  if (lane_id() == 0) s_sum[wave_id()] = term;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}

// *asymmetric = (sum of the per-block fingerprints != 0)
__global__ __launch_bounds__(1024) void gm_fingerprint_kernel(const unsigned long long* __restrict__ partial, int nb,
                                                              int* __restrict__ asymmetric) {
  __shared__ unsigned long long s_sum[16];
  unsigned long long t = 0;
  for (int b = threadIdx.x; b < nb; b += 1024) t += partial[b];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned lo = __shfl_xor(static_cast<unsigned>(t), off, WAVE);
    const unsigned hi = __shfl_xor(static_cast<unsigned>(t >> 32), off, WAVE);
    t += (static_cast<unsigned long long>(hi) << 32) | lo;
  }
  if (lane_id() == 0) s_sum[wave_id()] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long all = 0;
    for (int q = 0; q < 16; ++q) all += s_sum[q];
    *asymmetric = all != 0ull;
  }
}

__global__ __launch_bounds__(256) void gm_symmetrise_kernel(const int32_t* __restrict__ row_ptr,
                                                            const int32_t* __restrict__ src, int64_t E,
                                                            const int* __restrict__ asymmetric,
                                                            int32_t* __restrict__ nbr, float* __restrict__ wt) {
  if (!*asymmetric) return;  // fingerprint says every entry has an equal-weight reverse: nothing to do
  const int64_t p = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;  // one thread per CSR entry
  if (p >= E) return;
  const int32_t i = src[p], j = nbr[p];
  if (j < 0 || j == i) return;
  const int32_t lo = row_ptr[j], hi = row_ptr[j + 1];
  int32_t a = lo, b = hi, q = -1;
  while (a < b) {
    const int32_t mid = (a + b) >> 1;
    const int32_t v = nbr[mid];
    if (v == i) { q = mid; break; }
    if (v >= 0 && v < i) a = mid + 1; else b = mid;
  }
  if (q < 0)
    for (int32_t t = lo; t < hi; ++t)
      if (nbr[t] == i) { q = t; break; }
  if (q < 0) nbr[p] = -1;
  else wt[p] = fmaxf(wt[p], wt[q]);
}

constexpr int GM_LONG_ROW = 256;  // rows beyond this many entries are scanned by whole waves (see gm_best_neighbour_wave)
// r4: a long row is dealt out in CHUNKS of GM_CHUNK entries -- one ticket (an 8-word record {row, chunk, best j, w, hash,
// min, max, arrivals}) per chunk -- so that a 100 000-entry hub is scanned by 25 waves at once instead of one; the wave
// that brings a row's last chunk folds the chunk bests (the key is a total order: any dealing gives the sequential
// scan's answer) and writes the candidate.  Rows of at most GM_CHUNK entries have one ticket and no fold.
constexpr int GM_CHUNK = 4096;
constexpr int GM_REC = 8;  // int32 words per ticket record
__global__ __launch_bounds__(256) void gm_init_kernel(int64_t n, int64_t* __restrict__ label,
                                                      uint8_t* __restrict__ is_free,
                                                      const int32_t* __restrict__ row_ptr,
                                                      int32_t* __restrict__ long_list, unsigned int* __restrict__ long_ctr) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) {
    label[i] = i;
    is_free[i] = 1;
    const int32_t deg = row_ptr[i + 1] - row_ptr[i];
    if (deg > GM_LONG_ROW) {
      const int nch = (deg + GM_CHUNK - 1) / GM_CHUNK;
      const unsigned int base = atomicAdd(long_ctr, static_cast<unsigned int>(nch));
      for (int c = 0; c < nch; ++c) {
        int32_t* rec = long_list + static_cast<size_t>(base + c) * GM_REC;
        rec[0] = static_cast<int32_t>(i);
        rec[1] = c;
        rec[7] = 0;
      }
    }
  }
}

// Best free neighbour of free node i under the edge key (w, hash, min, max), or -1.  `is_free_of(j)` tests a
// neighbour's flag (global byte array or LDS bitmap).  A wave is as slow as its slowest lane and a lane's scan is
// a chain of dependent loads, so the neighbours are taken eight at a time: ids and weights first, flags second.
// a table addressed with GLOBAL indices whose storage starts at index `off` (an LDS copy of a graph's slice)
template <typename T>
struct GmShifted {
  const T* base;
  int64_t off;
  __device__ __forceinline__ T operator[](int64_t k) const { return base[k - off]; }
};

template <typename FreeFn, typename PtrT = const int32_t*, typename NbrT = const int32_t*, typename WtT = const float*>
__device__ __forceinline__ int32_t gm_best_neighbour(int64_t i, PtrT row_ptr, NbrT nbr, WtT wt, FreeFn is_free_of) {
  int32_t best = -1;
  float bw = 0.f;
  uint32_t bh = 0;
  const int32_t lo = row_ptr[i], hi = row_ptr[i + 1];
  constexpr int U = 8;
  for (int32_t p0 = lo; p0 < hi; p0 += U) {
    int32_t js[U];
    float ws[U];
    bool fs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      js[u] = p0 + u < hi ? nbr[p0 + u] : -1;
      ws[u] = p0 + u < hi ? wt[p0 + u] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) fs[u] = js[u] >= 0 && is_free_of(js[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int32_t j = js[u];
      if (!fs[u] || j == i) continue;
      const float wj = ws[u];
      if (wj != wj) continue;  // NaN weights never match
      const uint32_t a = static_cast<uint32_t>(i < j ? i : j), b = static_cast<uint32_t>(i < j ? j : i);
      const uint32_t h = pair_hash(a, b);
      // lexicographic (w, hash, min, max); min/max only matter for hash collisions between different pairs
      bool better = best < 0 || wj > bw || (wj == bw && h > bh);
      if (!better && wj == bw && h == bh && j != best) {
        const uint32_t ca = static_cast<uint32_t>(i < best ? i : best), cb = static_cast<uint32_t>(i < best ? best : i);
        better = a > ca || (a == ca && b > cb);
      }
      if (better) { best = j; bw = wj; bh = h; }
    }
  }
  return best;
}

// The same for ONE long row by the whole wave (every lane calls it with the same i): lane l takes entries lo + l,
// lo + l + 64, ... and the lanes' bests are folded under the same lexicographic key, so the result is the one the
// sequential scan gives.  Rows beyond GM_LONG_ROW entries take this: a hub of a power-law graph scanned by one lane
// held its whole wave (and the round) for degree / 8 dependent trips -- ten hubs of degree 100 000 in a 1M-node graph
// made the matching 27.6 ms instead of 0.32 ms.
__device__ __forceinline__ bool gm_key_better(bool have, float bw, uint32_t bh, uint32_t ca, uint32_t cb, bool ohave,
                                              float ow, uint32_t oh, uint32_t oa, uint32_t ob) {
  if (!ohave) return false;
  if (!have) return true;
  if (ow != bw) return ow > bw;
  if (oh != bh) return oh > bh;
  if (oa != ca) return oa > ca;
  return ob > cb;
}
struct GmBest {  // a candidate under the edge key (w, hash, min, max); j < 0: none
  int32_t j;
  float w;
  uint32_t h, a, b;
};
__device__ __forceinline__ void gm_fold(GmBest& x, const GmBest& o) {
  if (gm_key_better(x.j >= 0, x.w, x.h, x.a, x.b, o.j >= 0, o.w, o.h, o.a, o.b)) x = o;
}
__device__ __forceinline__ GmBest gm_wave_fold(GmBest x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    GmBest y;
    y.j = __shfl_xor(x.j, o, 64);
    y.w = __shfl_xor(x.w, o, 64);
    y.h = __shfl_xor(x.h, o, 64);
    y.a = __shfl_xor(x.a, o, 64);
    y.b = __shfl_xor(x.b, o, 64);
    gm_fold(x, y);
  }
  return x;
}
// entries [lo, hi) of row i by the whole wave; every lane returns the wave's best
template <typename FreeFn>
__device__ __forceinline__ GmBest gm_best_neighbour_wave(int64_t i, int32_t lo, int32_t hi,
                                                         const int32_t* __restrict__ nbr,
                                                         const float* __restrict__ wt, FreeFn is_free_of) {
  GmBest x{-1, 0.f, 0u, 0u, 0u};
  // r4: eight entries per lane in flight (512 per wave and trip), every load unconditional from a clamped position,
  // level by level -- entries, then their free flags: one entry per lane and trip made a 100 000-entry row 1 563
  // dependent round trips (1.5 ms; ten such hubs were 1.5 of the 1.8 ms the whole matching of a 1M-node graph took).
  // The key is a total order, so the fold does not depend on how the entries are dealt to the lanes.
  constexpr int U = 8;
  for (int32_t p0 = lo + lane_id(); p0 < hi; p0 += 64 * U) {
    int32_t js[U];
    float ws[U];
    bool fs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int32_t p = p0 + 64 * u;
      const bool in = p < hi;
      const int32_t pc = in ? p : lo;
      const int32_t j = nbr[pc];
      js[u] = in ? j : -1;
      ws[u] = wt[pc];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool f = is_free_of(js[u] >= 0 ? js[u] : static_cast<int32_t>(i));  // (a valid node id either way)
      fs[u] = f && js[u] >= 0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int32_t j = js[u];
      const float wj = ws[u];
      if (!fs[u] || j == i || wj != wj) continue;
      const uint32_t a = static_cast<uint32_t>(i < j ? i : j), b = static_cast<uint32_t>(i < j ? j : i);
      gm_fold(x, GmBest{j, wj, pair_hash(a, b), a, b});
    }
  }
  return gm_wave_fold(x);
}

// Proposal of a wave's nodes: short rows lane by lane; long rows are left to gm_steal_long_rows.  `mine`: this lane
// holds a free node i.  *is_long: the lane's row is long (its candidate is written by whoever takes it from the list).
template <typename FreeFn>
__device__ __forceinline__ int32_t gm_propose_short(bool mine, int64_t i, const int32_t* __restrict__ row_ptr,
                                                    const int32_t* __restrict__ nbr, const float* __restrict__ wt,
                                                    FreeFn is_free_of, bool* is_long) {
  *is_long = mine && row_ptr[i + 1] - row_ptr[i] > GM_LONG_ROW;
  return (mine && !*is_long) ? gm_best_neighbour(i, row_ptr, nbr, wt, is_free_of) : -1;
}

// Every wave of the launch takes long rows from the list until it is empty (a ticket counter: hubs sit next to each
// other -- the first nodes of a preferential-attachment graph -- and one wave owning 64 of them would scan them one
// after the other).  `owner_free(i)`: node i is free at the start of this round.
template <typename FreeFn, typename OwnFn>
__device__ __forceinline__ void gm_steal_long_rows(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ nbr,
                                                   const float* __restrict__ wt, const int32_t* __restrict__ long_list,
                                                   unsigned int* __restrict__ long_ctr, FreeFn is_free_of,
                                                   OwnFn owner_free, uint8_t* __restrict__ is_free,
                                                   int32_t* __restrict__ cand) {
  const unsigned int count = long_ctr[0];
  if (count == 0) return;
  for (;;) {
    // a plain look first: once the list is empty the ~15 k waves of a launch must not each queue an atomic on the one
    // ticket word (it takes ~88 atomics per us: 150 us per launch for nothing)
    unsigned int t = 0;
    if (lane_id() == 0) {
      t = __hip_atomic_load(long_ctr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t < count) t = atomicAdd(long_ctr + 1, 1u);
    }
    t = __shfl(t, 0, 64);
    if (t >= count) return;
    int32_t* rec = const_cast<int32_t*>(long_list) + static_cast<size_t>(t) * GM_REC;
    const int32_t i = rec[0], c = rec[1];
    if (!owner_free(i)) continue;  // (its candidate is already -1; every chunk of the row sees the same flag: it is only
                                   //  cleared by the wave that folds the row, after all chunks have arrived)
    const int32_t lo = row_ptr[i], hi = row_ptr[i + 1];
    const int nch = (hi - lo + GM_CHUNK - 1) / GM_CHUNK;
    const int32_t clo = lo + c * GM_CHUNK;
    GmBest r = gm_best_neighbour_wave(i, clo, clo + GM_CHUNK < hi ? clo + GM_CHUNK : hi, nbr, wt, is_free_of);
    if (nch > 1) {
      int32_t* rec0 = rec - static_cast<size_t>(c) * GM_REC;  // a row's tickets are consecutive, chunk 0 first
      int last = 0;
      if (lane_id() == 0) {
        __hip_atomic_store(rec + 2, r.j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(rec + 3, static_cast<int32_t>(__float_as_uint(r.w)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(rec + 4, static_cast<int32_t>(r.h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(rec + 5, static_cast<int32_t>(r.a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(rec + 6, static_cast<int32_t>(r.b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = __hip_atomic_fetch_add(rec0 + 7, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nch - 1;
      }
      last = __shfl(last, 0, 64);
      if (!last) continue;
      GmBest x{-1, 0.f, 0u, 0u, 0u};
      for (int k = lane_id(); k < nch; k += 64) {
        const int32_t* q = rec0 + static_cast<size_t>(k) * GM_REC;
        GmBest y;
        y.j = __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        y.w = __uint_as_float(static_cast<uint32_t>(__hip_atomic_load(q + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
        y.h = static_cast<uint32_t>(__hip_atomic_load(q + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        y.a = static_cast<uint32_t>(__hip_atomic_load(q + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        y.b = static_cast<uint32_t>(__hip_atomic_load(q + 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        gm_fold(x, y);
      }
      r = gm_wave_fold(x);
      if (lane_id() == 0) __hip_atomic_store(rec0 + 7, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next round
    }
    if (lane_id() == 0) {
      if (r.j < 0) is_free[i] = 0;  // retire (see gm_propose_kernel)
      cand[i] = r.j;
    }
  }
}

// cand[i] = best free neighbour of free node i, or -1 (free flags read from the global byte array).
__global__ __launch_bounds__(256) void gm_propose_kernel(const int32_t* __restrict__ row_ptr,
                                                         const int32_t* __restrict__ nbr,
                                                         const float* __restrict__ wt, int64_t n,
                                                         uint8_t* __restrict__ is_free,
                                                         int32_t* __restrict__ cand,
                                                         const int32_t* __restrict__ long_list,
                                                         unsigned int* __restrict__ long_ctr) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const bool mine = i < n && is_free[i] != 0;
  auto free_of = [&](int32_t j) { return is_free[j] != 0; };
  bool is_long;
  const int32_t best = gm_propose_short(mine, mine ? i : 0, row_ptr, nbr, wt, free_of, &is_long);
  if (i < n && !is_long) {
    // No free neighbour left: the free set only shrinks, so this node stays single -- retire it, later rounds
    // skip its scan.  (No free node is adjacent to it, so nobody's proposal depends on this flag.)
    if (mine && best < 0) is_free[i] = 0;
    cand[i] = best;
  }
  // (a long row's flag is only cleared by the wave that scans it, after its scan: nobody reads it as "retired" early;
  //  a neighbour reading it as free while it is being retired is the benign race described above)
  gm_steal_long_rows(row_ptr, nbr, wt, long_list, long_ctr, free_of, free_of, is_free, cand);
}

// Same proposal step with the free flags as a 1-bit-per-node map held in LDS (N <= kGmLdsNodes): the per-neighbour
// flag test is what bounds the byte-array version (10 M random one-byte reads = 10 M 64-byte L2 -> L1 sectors per
// round); from LDS it costs nothing next to the CSR stream.  Persistent 1024-thread workgroups, one per CU.
constexpr int kGmLdsWords = 36 * 1024;            // 144 KB of LDS (+ 4 KB of work list)
constexpr int64_t kGmLdsNodes = static_cast<int64_t>(kGmLdsWords) * 32;

__global__ __launch_bounds__(256) void gm_pack_free_kernel(const uint8_t* __restrict__ is_free, int64_t n,
                                                           uint32_t* __restrict__ bits) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const unsigned long long m = __ballot(i < n && is_free[i]);
  if (lane_id() == 0 && (i >> 5) < ((n + 31) >> 5)) {
    bits[i >> 5] = static_cast<uint32_t>(m);
    if (((i >> 5) + 1) < ((n + 31) >> 5)) bits[(i >> 5) + 1] = static_cast<uint32_t>(m >> 32);
  }
}

__global__ __launch_bounds__(1024) void gm_propose_lds_kernel(const int32_t* __restrict__ row_ptr,
                                                              const int32_t* __restrict__ nbr,
                                                              const float* __restrict__ wt, int64_t n,
                                                              const uint32_t* __restrict__ bits,
                                                              uint8_t* __restrict__ is_free,
                                                              int32_t* __restrict__ cand,
                                                              const int32_t* __restrict__ long_list,
                                                              unsigned int* __restrict__ long_ctr) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_bits[];
  __shared__ int32_t s_list[1024];
  __shared__ int s_cnt[16];
  lds_copy_words<1024>(s_bits, bits, static_cast<int>((n + 31) >> 5));  // 128 KB at 1 M nodes
  __syncthreads();
  auto free_bit = [&](int32_t j) { return ((s_bits[j >> 5] >> (j & 31)) & 1u) != 0; };
  const int w = threadIdx.x >> 6;
  // A wave pays the full scan latency as soon as ONE of its lanes holds a free node, and after a few rounds
  // nearly every wave still has one (5 % free nodes: 96 % of the waves).  So each 1024-node chunk first packs
  // its free nodes to the front (ballot ranks + a 16-entry scan) and only the packed prefix is scanned.
  for (int64_t c0 = static_cast<int64_t>(blockIdx.x) * 1024; c0 < n; c0 += static_cast<int64_t>(gridDim.x) * 1024) {
    const int64_t own = c0 + threadIdx.x;
    const bool is_f = own < n && free_bit(static_cast<int32_t>(own));
    if (own < n && !is_f) cand[own] = -1;
    const unsigned long long m = __ballot(is_f);
    if (lane_id() == 0) s_cnt[w] = __popcll(m);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int c = s_cnt[q];
      if (q < w) before += c;
      total += c;
    }
    if (is_f) s_list[before + __popcll(m & lanemask_lt())] = static_cast<int32_t>(own);
    __syncthreads();
    {
      const bool mine = static_cast<int>(threadIdx.x) < total;
      const int64_t i = mine ? s_list[threadIdx.x] : 0;
      bool is_long;
      const int32_t best = gm_propose_short(mine, i, row_ptr, nbr, wt, free_bit, &is_long);
      if (mine && !is_long) {
        if (best < 0) is_free[i] = 0;  // retire (see gm_propose_kernel)
        cand[i] = best;
      }
    }
    __syncthreads();  // s_list / s_cnt are reused by the next chunk
  }
  gm_steal_long_rows(row_ptr, nbr, wt, long_list, long_ctr, free_bit, free_bit, is_free, cand);
}

// Mutual proposals become pairs.  Each endpoint writes only its own slots; *matched is set when any pair formed.
__global__ __launch_bounds__(256) void gm_match_kernel(const int32_t* __restrict__ cand, int64_t n,
                                                       int64_t* __restrict__ label, uint8_t* __restrict__ is_free,
                                                       unsigned int* __restrict__ matched,
                                                       unsigned int* __restrict__ long_ctr) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  bool hit = false;
  if (i < n) {
    const int32_t j = cand[i];
    if (j >= 0 && cand[j] == static_cast<int32_t>(i)) {
      label[i] = i < j ? i : j;
      is_free[i] = 0;
      hit = true;
    }
  }
  // "did this round match anything": a plain store of 1 (one atomic per wave onto a single counter serialised
  // into 150 us at 1M nodes)
  if (__ballot(hit) && lane_id() == 0) *matched = 1u;
  if (i == 0 && long_ctr) long_ctr[1] = 0;  // the long rows' ticket counter of the next round's proposal kernel
}

// Batches of graphs of at most GM_GRAPH_MAX nodes with a sorted batch vector: ONE workgroup per graph runs ALL the
// propose / match rounds of its graph in one launch (graphs do not interact, so the matching is the one the device-wide
// rounds give; the workgroup barrier replaces the launch boundary).  A PROTEINS-shaped batch took 6 rounds x 3 launches
// and two host read-backs of the "anything matched" flags; an entry that leaves the graph raises *status (the caller
// takes the device-wide rounds then).
constexpr int GM_GRAPH_MAX = 1024;
constexpr int GM_GRAPH_LDS_E = 2048;  // entries of a graph's CSR kept in LDS (16 KB: six workgroups per CU)

// The rounds of one graph.  `ptr`, `nb`, `wv` are indexed with GLOBAL node ids / entry offsets (LDS copies come shifted),
// free flags and candidates live in LDS under local ids.
template <int T, typename P, typename NB, typename WV>
__device__ __forceinline__ void gm_graph_rounds(P ptr, NB nb, WV wv, int64_t p0, int n, uint8_t* s_free, int32_t* s_cand,
                                                int64_t* __restrict__ label) {
  for (int t = threadIdx.x; t < n; t += T) {
    s_free[t] = 1;
    label[p0 + t] = p0 + t;  // (pairs overwrite it below, after a barrier)
  }
  __syncthreads();
  for (int round = 0; round < 2 * GM_GRAPH_MAX; ++round) {  // (a round matches at least one pair while a free edge exists)
    for (int t = threadIdx.x; t < n; t += T) {
      int32_t c = -1;
      if (s_free[t]) {
        c = gm_best_neighbour(p0 + t, ptr, nb, wv, [&](int32_t j) { return s_free[j - p0] != 0; });
        if (c < 0) s_free[t] = 0;  // retire (see gm_propose_kernel); nobody adjacent is free, so nobody reads this flag
      }
      s_cand[t] = c;
    }
    __syncthreads();
    bool hit = false;
    for (int t = threadIdx.x; t < n; t += T) {
      const int32_t j = s_cand[t];
      const int32_t i = static_cast<int32_t>(p0) + t;
      if (j >= 0 && s_cand[j - p0] == i) {
        label[p0 + t] = i < j ? i : j;
        s_free[t] = 0;  // (own flag; this step reads candidates only, the barrier below orders it before the next scan)
        hit = true;
      }
    }
    if (!__syncthreads_or(hit ? 1 : 0)) break;
  }
}

// T threads per graph (64 when the batch's longest graph fits one wave: the barriers cost nothing and a CU holds 32
// graphs); LDS: offsets [cap_n + 1], candidates [cap_n], CSR entries 2 x [cap_e], flags [cap_n] bytes.
template <int T>
__global__ __launch_bounds__(T) void gm_graph_rounds_kernel(const int32_t* __restrict__ row_ptr,
                                                            const int32_t* __restrict__ nbr,
                                                            const float* __restrict__ wt,
                                                            const int64_t* __restrict__ graph_ptr, int cap_n, int cap_e,
                                                            int64_t* __restrict__ label, int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) int32_t gm_lds[];
  int32_t* s_ptr = gm_lds;
  int32_t* s_cand = s_ptr + cap_n + 1;
  int32_t* s_nbr = s_cand + cap_n;
  float* s_wt = reinterpret_cast<float*>(s_nbr + cap_e);
  uint8_t* s_free = reinterpret_cast<uint8_t*>(s_wt + cap_e);
  const int64_t p0 = graph_ptr[blockIdx.x], p1 = graph_ptr[blockIdx.x + 1];
  const int n = static_cast<int>(p1 - p0);
  if (n <= 0) return;
  if (n > cap_n) {
    if (threadIdx.x == 0) atomicOr(status, 1);
    return;
  }
  for (int t = threadIdx.x; t <= n; t += T) s_ptr[t] = row_ptr[p0 + t];
  const int32_t e0 = row_ptr[p0], e1 = row_ptr[p1];
  const bool in_lds = e1 - e0 <= cap_e;
  bool out = false;  // entries must stay inside the graph (the rounds read the neighbours' flags without a range test)
  // ... and the graph's offsets must BE a CSR slice: ascending, inside [e0, e1].  On the optimistic route (a list nobody
  // has checked for row order yet) an unsorted list leaves in-range but non-monotonic offsets behind; the rounds would
  // then index their LDS views outside the graph's slice.  Refused here, before anything is read through them.
  bool broken = e1 < e0;
  for (int t = threadIdx.x; t < n; t += T) {
    const int32_t a = row_ptr[p0 + t], b = row_ptr[p0 + t + 1];
    broken = broken || a > b || a < e0 || b > e1;
  }
  if (__syncthreads_or(broken ? 1 : 0)) {
    if (threadIdx.x == 0) atomicOr(status, 4);
    return;
  }
  for (int32_t e = e0 + threadIdx.x; e < e1; e += T) {
    const int32_t j = nbr[e];
    out = out || (j >= 0 && (j < p0 || j >= p1));  // (-1: an entry the symmetrisation dropped)
    if (in_lds) {
      s_nbr[e - e0] = j;
      s_wt[e - e0] = wt[e];
    }
  }
  if (__syncthreads_or(out ? 1 : 0)) {
    if (threadIdx.x == 0) atomicOr(status, 2);
    return;
  }
  const GmShifted<int32_t> ptr_g{s_ptr, p0};
  if (in_lds)
    gm_graph_rounds<T>(ptr_g, GmShifted<int32_t>{s_nbr, e0}, GmShifted<float>{s_wt, e0}, p0, n, s_free, s_cand, label);
  else
    gm_graph_rounds<T>(ptr_g, nbr, wt, p0, n, s_free, s_cand, label);
}

// ---- r4: matching + relabelling + members index of a sorted batch of SMALL graphs in ONE launch -------------------------
// What `graclus` on a PROTEINS-shaped batch spent around the rounds kernel above: offsets of the list (1 launch), CSR
// gather + symmetry fingerprint + symmetrise (3), a status memset, the rounds (1), a bitmap memset + three relabel
// launches, a device-to-host copy -- eleven launches for 80 us of GPU time.  Here ONE WAVE PER GRAPH (<= 64 nodes,
// <= GF_CAP entries; lane = node) does all of it out of LDS:
//   * its graph's entry range from the row array itself (128-ary lower bounds + one boundary pass per workgroup, as
//     sparse_pool_small.hip: no offsets kernel), validated -- rows ascending, ranges tiling [0, E), entries inside the graph;
//   * symmetrisation by looking the reverse entry up in LDS (pair weight = the larger direction, entries without a
//     reverse dropped: gm_symmetrise_kernel's rule, applied always -- max(w, w) = w on a symmetric list);
//   * the handshake rounds with the free set as ONE 64-bit ballot and the candidates exchanged by `__shfl` (same key on
//     GLOBAL ids, so the pairs are those of the device-wide rounds);
//   * cluster ids = rank of the representative among the representatives (popcounts of a ballot) + the clusters of all
//     earlier graphs from the epoch-tagged look-back of lookback.h; the members index needs no second prefix: a graph's
//     clusters cover exactly its nodes, so its slots start at its first node.
// {epoch, refused, K} lands in *result (pinned host memory).  Refuses (the caller takes the staged route): a graph beyond
// 64 nodes or GF_CAP entries, an entry that leaves its graph, rows not ascending.
constexpr int GF_WAVES = 4;
constexpr int GF_CAP = 512;
struct GfWaveLds {
  int32_t ptr[65];
  uint8_t row[GF_CAP];
  uint8_t nbr[GF_CAP];  // local column; 255: dropped by the symmetrisation
  float wt[GF_CAP];
};
struct GfArgs {
  const int64_t *row, *col;
  const float* w;
  int64_t N, E;
  const int64_t* gptr;
  int64_t B;
  const int64_t* edge_ptr;  // NULL, or [B + 1]: first entry of every graph (the caller's memo; re-checked)
  int64_t* label;  // NULL ok
  int64_t* index;  // [2, N]
  int32_t *a_row_ptr, *a_perm;
  float* ones;
  unsigned long long *status, *result;
  unsigned long long tag;
};

__global__ __launch_bounds__(GF_WAVES * 64) void gm_graph_fused_kernel(GfArgs p) {
  __shared__ uint32_t s_cnt[GF_WAVES];
  __shared__ uint32_t s_base;
  __shared__ int s_ok;
  __shared__ int64_t s_nb[GF_WAVES + 1];
  __shared__ int s_eb[GF_WAVES + 1];
  __shared__ int64_t s_rng[2];
  __shared__ GfWaveLds s_all[GF_WAVES];
  const int lane = lane_id(), wv = wave_id();
  const int64_t g0 = static_cast<int64_t>(blockIdx.x) * GF_WAVES;
  if (threadIdx.x <= GF_WAVES) {
    const int64_t gi = g0 + threadIdx.x < p.B ? g0 + threadIdx.x : p.B;
    s_nb[threadIdx.x] = p.gptr[gi];
    s_eb[threadIdx.x] = INT_MAX;
  }
  const bool given = p.edge_ptr != nullptr;  // (as sparse_pool_small_kernel: one round trip instead of the searches)
  if (given) {
    if (threadIdx.x <= GF_WAVES) {
      const int64_t gi = g0 + threadIdx.x < p.B ? g0 + threadIdx.x : p.B;
      const int64_t eg = p.edge_ptr[gi], e_first = p.edge_ptr[g0 < p.B ? g0 : p.B];
      if (threadIdx.x == 0) s_rng[0] = eg;
      if (threadIdx.x == GF_WAVES) s_rng[1] = eg;
      const int64_t de = eg - e_first;
      s_eb[threadIdx.x] = (de < 0 || de > INT_MAX - 1) ? -1 : static_cast<int>(de);
    }
  } else if (wv == 0) {
    const int64_t N0 = p.gptr[g0], N1 = p.gptr[g0 + GF_WAVES < p.B ? g0 + GF_WAVES : p.B];
    const int64_t* const arrs[2] = {p.row, p.row};
    const int64_t ns[2] = {p.E, p.E}, keys[2] = {N0, N1};
    int64_t res[2];
    wave_lower_bounds<2>(arrs, ns, keys, res);
    if (lane < 2) s_rng[lane] = lane == 0 ? res[0] : res[1];
  }
  __syncthreads();
  const int64_t E0 = s_rng[0], E1 = s_rng[1], LE = E1 - E0;
  bool bad = false;
  // the workgroups' ranges tile the list (see sparse_pool_small_kernel): the first starts at 0, the last ends at E
  if (LE < 0 || (blockIdx.x == 0 && (E0 != 0 || s_nb[0] != 0)) ||
      (blockIdx.x == gridDim.x - 1 && (E1 != p.E || s_nb[GF_WAVES] != p.N)))
    bad = true;
  if (!bad && !given) sps_boundaries<GF_WAVES>(p.row, E0, LE, s_nb, s_eb);
  if (!given) __syncthreads();
  if (given && (s_eb[wv] < 0 || s_eb[wv + 1] < 0)) bad = true;  // a corrupt table
  int64_t n0 = s_nb[wv], n1 = s_nb[wv + 1];
  if (n1 < n0 || n1 - n0 > 64 || n0 < 0 || n1 > p.N) {
    bad = true;
    n0 = n1 = 0;
  }
  int64_t e0 = 0;
  int m = 0;
  if (!bad) {
    const int64_t b0 = s_eb[wv] < LE ? s_eb[wv] : LE, b1 = s_eb[wv + 1] < LE ? s_eb[wv + 1] : LE;
    if (b1 < b0 || (wv == 0 && b0 != 0) || (wv == GF_WAVES - 1 && b1 != LE) || b1 - b0 > GF_CAP) bad = true;
    else {
      e0 = E0 + b0;
      m = static_cast<int>(b1 - b0);
    }
  }
  const int n = bad ? 0 : static_cast<int>(n1 - n0);
  GfWaveLds& L = s_all[wv];
  L.ptr[lane] = m;
  if (lane == 0) L.ptr[64] = m;
  __builtin_amdgcn_wave_barrier();
  // stage the entries; ptr[t] = first entry of local row t (entries are grouped by ascending row, or the wave refuses)
  for (int e = lane; e < m; e += 64) {
    const int64_t r = p.row[e0 + e] - n0, rp = e > 0 ? p.row[e0 + e - 1] - n0 : -1;
    const int64_t c = p.col[e0 + e] - n0;
    const bool okr = r >= 0 && r < n && rp <= r && rp >= -1, okc = c >= 0 && c < n;
    if (!okr || !okc) bad = true;
    L.row[e] = static_cast<uint8_t>(okr ? r : 0);
    L.nbr[e] = static_cast<uint8_t>(okc ? c : 0);
    L.wt[e] = p.w ? p.w[e0 + e] : 1.0f;
    if (okr && r != rp)
      for (int64_t t = rp + 1; t <= r; ++t) L.ptr[t] = e;
  }
  bad = __any(bad);
  __builtin_amdgcn_wave_barrier();
  // symmetrise (reverse entry looked up in LDS); two phases: all reads, then all writes
  constexpr int PER = GF_CAP / 64;
  float nw[PER];
  bool drop[PER];
  const int mm = bad ? 0 : m;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int e = lane + 64 * k;
    nw[k] = 0.f;
    drop[k] = false;
    if (e < mm) {
      const int i = L.row[e], j = L.nbr[e];
      const float w = L.wt[e];
      nw[k] = w;
      if (j != i) {
        // (the probe order of gm_symmetrise_kernel -- bisection over the row, then a linear scan -- so that a list with
        //  DUPLICATE entries of different weights picks the same reverse entry and the two routes give the same pairs)
        const int lo = L.ptr[j], hi = L.ptr[j + 1];
        int a = lo, b = hi, q = -1;
        while (a < b) {
          const int mid = (a + b) >> 1;
          const int v = L.nbr[mid];
          if (v == i) { q = mid; break; }
          if (v < i) a = mid + 1; else b = mid;
        }
        if (q < 0)
          for (int t = lo; t < hi; ++t)
            if (L.nbr[t] == i) { q = t; break; }
        drop[k] = q < 0;
        if (q >= 0) nw[k] = fmaxf(w, L.wt[q]);
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int e = lane + 64 * k;
    if (e < mm) {
      if (drop[k]) L.nbr[e] = 255;
      else L.wt[e] = nw[k];
    }
  }
  __builtin_amdgcn_wave_barrier();
  // the rounds: lane = node, free set = one ballot, candidates exchanged by shuffle
  const int nn = bad ? 0 : n;
  bool isfree = lane < nn;
  int partner = -1;
  const uint32_t gi = static_cast<uint32_t>(n0) + static_cast<uint32_t>(lane);
  for (int round = 0; round < 2 * 64 + 2; ++round) {
    const unsigned long long free_mask = __ballot(isfree);
    GmBest best{-1, 0.f, 0u, 0u, 0u};
    if (isfree) {
      for (int e = L.ptr[lane]; e < L.ptr[lane + 1]; ++e) {
        const int j = L.nbr[e];
        if (j == 255 || j == lane || !((free_mask >> j) & 1ull)) continue;
        const float wj = L.wt[e];
        if (wj != wj) continue;  // NaN weights never match
        const uint32_t gj = static_cast<uint32_t>(n0) + static_cast<uint32_t>(j);
        const uint32_t a = gi < gj ? gi : gj, b = gi < gj ? gj : gi;
        gm_fold(best, GmBest{j, wj, pair_hash(a, b), a, b});
      }
      if (best.j < 0) isfree = false;  // no free neighbour left: retire (see gm_propose_kernel)
    }
    const int c = best.j;
    const int cj = __shfl(c, c >= 0 ? c : 0, 64);
    const bool hit = c >= 0 && cj == lane;
    if (hit) {
      partner = c;
      isfree = false;
    }
    if (!__any(hit)) break;
  }
  const bool valid = lane < nn;
  const int lab = partner >= 0 && partner < lane ? partner : lane;  // local id of the pair's smaller node
  const bool rep = valid && lab == lane;
  const unsigned long long repmask = __ballot(rep), pairmask = __ballot(rep && partner >= 0);
  const uint32_t cnt = static_cast<uint32_t>(__popcll(repmask));
  if (lane == 0) s_cnt[wv] = cnt | (bad ? 0x80000000u : 0u);
  __syncthreads();
  if (wv == 0) {
    const int tile = blockIdx.x;
    uint32_t tile_tot = 0;
    bool refused = false;
#pragma unroll
    for (int w2 = 0; w2 < GF_WAVES; ++w2) {
      tile_tot += s_cnt[w2] & 0x7FFFFFFFu;
      refused = refused || (s_cnt[w2] >> 31) != 0u;
    }
    if (lane == 0)
      sps_store(p.status + 2 + tile, p.tag | (tile == 0 ? SPS_PRE : SPS_AGG) | (refused ? 0x80000000ull : 0ull) | tile_tot);
    uint32_t excl = 0;
    if (tile > 0) {
      bool before = false;
      sps_lookback<4>(p.status, tile, p.tag, &excl, &before);
      refused = refused || before;
      if (lane == 0)
        sps_store(p.status + 2 + tile, p.tag | SPS_PRE | (refused ? 0x80000000ull : 0ull) |
                                           static_cast<unsigned long long>((excl + tile_tot) & 0x7FFFFFFFu));
    }
    if (lane == 0) {
      s_base = excl;
      s_ok = refused ? 0 : 1;
      if (tile == static_cast<int>(gridDim.x) - 1) {
        if (!refused) p.a_row_ptr[excl + tile_tot] = static_cast<int32_t>(p.N);
        __hip_atomic_store(p.result, p.tag | (refused ? 0x80000000ull : 0ull) |
                                         static_cast<unsigned long long>((excl + tile_tot) & 0x7FFFFFFFu),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  __syncthreads();
  if (!s_ok || !valid) return;
  uint32_t base = s_base;
  for (int w2 = 0; w2 < wv; ++w2) base += s_cnt[w2] & 0x7FFFFFFFu;
  const unsigned long long below = (1ull << lab) - 1ull;
  const uint32_t lr = static_cast<uint32_t>(__popcll(repmask & below));
  const int64_t slot = n0 + lr + __popcll(pairmask & below);  // members of this graph's earlier clusters
  const int64_t i = n0 + lane;
  p.index[i] = i;
  p.index[p.N + i] = static_cast<int64_t>(base + lr);
  p.ones[i] = 1.0f;
  if (p.label) p.label[i] = n0 + lab;
  if (rep) {
    p.a_row_ptr[base + lr] = static_cast<int32_t>(slot);
    p.a_perm[slot] = static_cast<int32_t>(i);
  } else {
    p.a_perm[slot + 1] = static_cast<int32_t>(i);
  }
}

// ---- the tail of the device-wide rounds ---------------------------------------------------------------------------------
// After a few rounds a random-like graph has a few thousand free nodes left that still have a free neighbour, and every
// further round costs three launches over all N nodes (15 us at N = 1M) to match a handful of pairs.  The tail gathers
// those nodes into a list (at most GM_TAIL_CAP, else it declines) and ONE workgroup runs all remaining rounds over the
// list: same proposals, same pairs (a proposal depends only on the free set), the workgroup barrier is the round boundary.
constexpr int GM_TAIL_CAP = 16384;
__global__ __launch_bounds__(256) void gm_tail_list_kernel(const uint8_t* __restrict__ is_free, int64_t n,
                                                           int32_t* __restrict__ list, unsigned int* __restrict__ count) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const bool f = i < n && is_free[i];
  const unsigned long long m = __ballot(f);
  if (m == 0) return;
  unsigned int base = 0;
  if (lane_id() == 0) base = atomicAdd(count, static_cast<unsigned int>(__popcll(m)));
  base = __shfl(base, 0, WAVE);
  const unsigned int at = base + static_cast<unsigned int>(__popcll(m & lanemask_lt()));
  if (f && at < static_cast<unsigned int>(GM_TAIL_CAP)) list[at] = static_cast<int32_t>(i);
}

__global__ __launch_bounds__(1024) void gm_tail_rounds_kernel(const int32_t* __restrict__ row_ptr,
                                                              const int32_t* __restrict__ nbr,
                                                              const float* __restrict__ wt,
                                                              const int32_t* __restrict__ list,
                                                              const unsigned int* __restrict__ count,
                                                              int64_t* __restrict__ label, uint8_t* __restrict__ is_free,
                                                              int32_t* __restrict__ cand, int* __restrict__ status) {
  __shared__ int32_t s_list[GM_TAIL_CAP];
  __shared__ unsigned int s_cnt[16];
  unsigned int m = *count;
  if (m > static_cast<unsigned int>(GM_TAIL_CAP)) {
    if (threadIdx.x == 0) *status = 0;  // too many: the caller goes on with device-wide rounds
    return;
  }
  for (unsigned int t = threadIdx.x; t < m; t += 1024) s_list[t] = list[t];
  __syncthreads();
  for (int round = 0; round < (1 << 30); ++round) {
    for (unsigned int t0 = 0; t0 < m; t0 += 1024) {  // (same trip count for every lane: long rows are scanned by the wave)
      const unsigned int t = t0 + threadIdx.x;
      const int32_t i = t < m ? s_list[t] : 0;
      const bool mine = t < m && is_free[i] != 0;
      auto free_of = [&](int32_t j) { return is_free[j] != 0; };
      bool is_long;
      int32_t c = gm_propose_short(mine, i, row_ptr, nbr, wt, free_of, &is_long);
      unsigned long long lm = __ballot(is_long);  // (a hub that is still free this late: its wave scans it)
      while (lm) {
        const int l = __ffsll(static_cast<long long>(lm)) - 1;
        lm &= lm - 1;
        const int32_t il = __shfl(i, l, 64);
        const int32_t r = gm_best_neighbour_wave(il, row_ptr[il], row_ptr[il + 1], nbr, wt, free_of).j;
        if (lane_id() == l) c = r;
      }
      if (mine && c < 0) is_free[i] = 0;  // retire (see gm_propose_kernel)
      if (t < m) cand[i] = c;
    }
    __syncthreads();
    bool hit = false;
    for (unsigned int t = threadIdx.x; t < m; t += 1024) {
      const int32_t i = s_list[t];
      const int32_t j = cand[i];
      if (j >= 0 && cand[j] == i) {  // (j is free and has a free neighbour, so it is on the list and proposed this round)
        label[i] = i < j ? i : j;
        is_free[i] = 0;
        hit = true;
      }
    }
    if (!__syncthreads_or(hit ? 1 : 0)) break;
    // keep only the nodes that are still free (matched and retired ones drop out: later rounds are one node per thread)
    unsigned int kept = 0;
    for (unsigned int base = 0; base < m; base += 1024) {
      const unsigned int t = base + threadIdx.x;
      const int32_t i = t < m ? s_list[t] : 0;
      const bool f = t < m && is_free[i] != 0;
      const unsigned long long bm = __ballot(f);
      if (lane_id() == 0) s_cnt[threadIdx.x >> 6] = __popcll(bm);
      __syncthreads();  // (also: every entry of this chunk has been read)
      unsigned int before = 0, total = 0;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const unsigned int c = s_cnt[q];
        if (q < static_cast<int>(threadIdx.x >> 6)) before += c;
        total += c;
      }
      if (f) s_list[kept + before + __popcll(bm & lanemask_lt())] = i;
      kept += total;
      __syncthreads();
    }
    m = kept;
  }
  if (threadIdx.x == 0) *status = 1;
}

// ---- labels -> consecutive cluster ids (the torch.unique(return_inverse=True) of select/graclus_select.py:68) -------
// A representative is a node with label[i] == i; its id is the number of representatives before it, and every node takes
// the id of its label.  Two launches, no sort: 1024-node tiles write their flag words, the running popcount of the words
// inside the tile and the tile total; the second launch scans the (few) tile totals in LDS in every workgroup and looks
// the rank of label[i] up as tile prefix + word prefix + popcount of the lower bits.
constexpr int RL_TILE = 1024;
constexpr int RL_MAX_TILES = 4096;  // tile prefixes held in LDS (two tables of 16 KB): 4.2 M nodes
__global__ __launch_bounds__(256) void rl_flags_kernel(const int64_t* __restrict__ label, int64_t n,
                                                       uint32_t* __restrict__ bits, uint32_t* __restrict__ wprefix,
                                                       uint32_t* __restrict__ tile_sum,
                                                       uint32_t* __restrict__ paired) {
  __shared__ uint32_t pc[32];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * RL_TILE;
  const int w = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    const int64_t li = i < n ? label[i] : i;
    if (paired && i < n && li != i && li >= 0 && li < n)  // the partner marks its representative (zeroed bitmap)
      atomicOr(paired + (li >> 5), 1u << (li & 31));
    const unsigned long long m = __ballot(i < n && li == i);
    if (lane_id() == 0) {
      const int q = r * 8 + w * 2;
      const uint32_t lo = static_cast<uint32_t>(m), hi = static_cast<uint32_t>(m >> 32);
      bits[(base >> 5) + q] = lo;
      bits[(base >> 5) + q + 1] = hi;
      pc[q] = __popc(lo);
      pc[q + 1] = __popc(hi);
    }
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    uint32_t before = 0;
    for (int q = 0; q < static_cast<int>(threadIdx.x); ++q) before += pc[q];
    wprefix[(base >> 5) + threadIdx.x] = before;
    if (threadIdx.x == 31) tile_sum[blockIdx.x] = before + pc[31];
  }
}

// word prefixes and tile totals of the "has a partner" bitmap (complete only after rl_flags_kernel has finished)
__global__ __launch_bounds__(256) void rl_paired_prefix_kernel(const uint32_t* __restrict__ paired, int tiles,
                                                               uint32_t* __restrict__ wprefix2,
                                                               uint32_t* __restrict__ tile_sum2) {
  const int t = blockIdx.x * 8 + (threadIdx.x >> 5), q = threadIdx.x & 31;  // 32 lanes per tile, 8 tiles per workgroup
  if (t >= tiles) return;
  const uint32_t c = __popc(paired[static_cast<int64_t>(t) * 32 + q]);
  uint32_t inc = c;
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) {
    const uint32_t v = __shfl_up(inc, o, 32);
    if (q >= o) inc += v;
  }
  wprefix2[static_cast<int64_t>(t) * 32 + q] = inc - c;
  if (q == 31) tile_sum2[t] = inc;
}

// exclusive scan of `tiles` totals into s_tp (every workgroup does it for itself); returns the grand total
__device__ __forceinline__ uint32_t rl_scan_tiles(const uint32_t* __restrict__ tile_sum, int tiles, uint32_t* s_tp,
                                                  uint32_t* s_part) {
  const int per = (tiles + 255) / 256;  // thread t owns a contiguous run of `per` tiles
  uint32_t mine = 0;
  for (int q = 0; q < per; ++q) {
    const int t = threadIdx.x * per + q;
    mine += t < tiles ? tile_sum[t] : 0u;
  }
  __syncthreads();  // (s_part may still be read from a previous call)
  s_part[threadIdx.x] = mine;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const uint32_t v = static_cast<int>(threadIdx.x) >= d ? s_part[threadIdx.x - d] : 0u;
    __syncthreads();
    s_part[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = s_part[threadIdx.x] - mine;
  for (int q = 0; q < per; ++q) {
    const int t = threadIdx.x * per + q;
    if (t < tiles) {
      s_tp[t] = run;
      run += tile_sum[t];
    }
  }
  __syncthreads();
  return s_part[255];
}

// PAIRS: also the supernode -> members index of a MATCHING (every label shared by at most two nodes): cluster c of
// representative r starts at c + (paired representatives before r) and holds r, then its partner -- ascending node
// order, what tgp_assign_index_build derives from the cluster ids with five launches.
template <bool PAIRS>
__global__ __launch_bounds__(256) void rl_assign_kernel(const int64_t* __restrict__ label, int64_t n,
                                                        const uint32_t* __restrict__ bits,
                                                        const uint32_t* __restrict__ wprefix,
                                                        const uint32_t* __restrict__ tile_sum,
                                                        const uint32_t* __restrict__ paired,
                                                        const uint32_t* __restrict__ wprefix2,
                                                        const uint32_t* __restrict__ tile_sum2, int tiles,
                                                        int64_t* __restrict__ index_out, int64_t* __restrict__ d_k,
                                                        int32_t* __restrict__ a_row_ptr, int32_t* __restrict__ a_perm,
                                                        float* __restrict__ ones) {
  __shared__ uint32_t s_tp[RL_MAX_TILES];
  __shared__ uint32_t s_tp2[PAIRS ? RL_MAX_TILES : 1];
  __shared__ uint32_t s_part[256];
  const uint32_t total = rl_scan_tiles(tile_sum, tiles, s_tp, s_part);
  if constexpr (PAIRS) rl_scan_tiles(tile_sum2, tiles, s_tp2, s_part);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *d_k = static_cast<int64_t>(total);
    if constexpr (PAIRS) a_row_ptr[total] = static_cast<int32_t>(n);
  }
  const int64_t base = static_cast<int64_t>(blockIdx.x) * RL_TILE;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    if (i >= n) continue;
    int64_t j = label[i];
    j = j < 0 ? 0 : (j >= n ? n - 1 : j);  // (the matcher's labels are in range; a caller's might not be)
    const uint32_t below = (1u << (j & 31)) - 1u;
    const uint32_t rank = s_tp[j >> 10] + wprefix[j >> 5] + __popc(bits[j >> 5] & below);
    index_out[i] = i;
    index_out[n + i] = static_cast<int64_t>(rank);
    if (ones) ones[i] = 1.0f;  // (the values of the one-over-K assignment matrix)
    if constexpr (PAIRS) {
      const uint32_t slot = rank + s_tp2[j >> 10] + wprefix2[j >> 5] + __popc(paired[j >> 5] & below);
      if (i == j) {
        a_row_ptr[rank] = static_cast<int32_t>(slot);
        a_perm[slot] = static_cast<int32_t>(i);
      } else {
        a_perm[slot + 1] = static_cast<int32_t>(i);
      }
    }
  }
}

}  // namespace tgp

using namespace tgp;

extern "C" int64_t tgp_graclus_relabel_max_nodes(void) { return static_cast<int64_t>(RL_TILE) * RL_MAX_TILES; }

extern "C" size_t tgp_graclus_relabel_workspace_bytes(int64_t num_nodes) {
  const size_t tiles = static_cast<size_t>(cdiv(num_nodes > 0 ? num_nodes : 1, RL_TILE));
  return 4 * align_up(tiles * 32 * sizeof(uint32_t)) + 2 * align_up(tiles * sizeof(uint32_t)) + 256;
}

// index_out[0][i] = i, index_out[1][i] = consecutive id of label[i] (ids in the order of the representatives
// label[r] == r), *d_k = number of ids: the indices of the [N, K] assignment the selector returns.
extern "C" int tgp_graclus_relabel_i64(const int64_t* label, int64_t num_nodes, void* ws, size_t ws_bytes,
                                       int64_t* index_out, int64_t* d_k, int32_t* assign_row_ptr,
                                       int32_t* assign_perm, float* ones, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(num_nodes >= 0 && d_k, TGP_ERR_INVALID, "tgp_graclus_relabel_i64: bad argument");
  TGP_REQUIRE((assign_row_ptr == nullptr) == (assign_perm == nullptr), TGP_ERR_INVALID,
              "tgp_graclus_relabel_i64: assign_row_ptr and assign_perm come together");
  if (num_nodes == 0) {
    (void)hipMemsetAsync(d_k, 0, sizeof(int64_t), stream);
    if (assign_row_ptr) (void)hipMemsetAsync(assign_row_ptr, 0, sizeof(int32_t), stream);
    return check_launch("tgp_graclus_relabel_i64");
  }
  TGP_REQUIRE(num_nodes <= tgp_graclus_relabel_max_nodes(), TGP_ERR_RANGE, "tgp_graclus_relabel_i64: too many nodes");
  TGP_REQUIRE(label && index_out && ws && ws_bytes >= tgp_graclus_relabel_workspace_bytes(num_nodes),
              TGP_ERR_WORKSPACE, "tgp_graclus_relabel_i64: null pointer / workspace too small");
  const int tiles = static_cast<int>(cdiv(num_nodes, RL_TILE));
  Carver cv(ws);
  uint32_t* bits = cv.take<uint32_t>(static_cast<int64_t>(tiles) * 32);
  uint32_t* wprefix = cv.take<uint32_t>(static_cast<int64_t>(tiles) * 32);
  uint32_t* tile_sum = cv.take<uint32_t>(tiles);
  uint32_t* paired = cv.take<uint32_t>(static_cast<int64_t>(tiles) * 32);
  uint32_t* wprefix2 = cv.take<uint32_t>(static_cast<int64_t>(tiles) * 32);
  uint32_t* tile_sum2 = cv.take<uint32_t>(tiles);
  if (assign_row_ptr) {
    (void)hipMemsetAsync(paired, 0, static_cast<size_t>(tiles) * 32 * sizeof(uint32_t), stream);
    hipLaunchKernelGGL(rl_flags_kernel, dim3(tiles), dim3(256), 0, stream, label, num_nodes, bits, wprefix, tile_sum,
                       paired);
    hipLaunchKernelGGL(rl_paired_prefix_kernel, dim3(cdiv(tiles, 8)), dim3(256), 0, stream, paired, tiles, wprefix2,
                       tile_sum2);
    hipLaunchKernelGGL(rl_assign_kernel<true>, dim3(tiles), dim3(256), 0, stream, label, num_nodes, bits, wprefix,
                       tile_sum, paired, wprefix2, tile_sum2, tiles, index_out, d_k, assign_row_ptr, assign_perm, ones);
  } else {
    hipLaunchKernelGGL(rl_flags_kernel, dim3(tiles), dim3(256), 0, stream, label, num_nodes, bits, wprefix, tile_sum,
                       static_cast<uint32_t*>(nullptr));
    hipLaunchKernelGGL(rl_assign_kernel<false>, dim3(tiles), dim3(256), 0, stream, label, num_nodes, bits, wprefix,
                       tile_sum, paired, wprefix2, tile_sum2, tiles, index_out, d_k, assign_row_ptr, assign_perm, ones);
  }
  return check_launch("tgp_graclus_relabel_i64");
}

extern "C" int tgp_graclus_match_graphs_fused_max_graph_nodes(void) { return 64; }
extern "C" int64_t tgp_graclus_match_graphs_fused_status_words(int64_t B) { return 2 + cdiv(B > 0 ? B : 1, GF_WAVES); }

// r4: matching + consecutive cluster ids + supernode -> members index of a sorted batch of graphs of at most 64 nodes in
// ONE launch (gm_graph_fused_kernel).  row / col / w: the row-sorted edge list itself (no CSR needed).
extern "C" int tgp_graclus_match_graphs_fused(const int64_t* row, const int64_t* col, const float* w, int64_t N, int64_t E,
                                              const int64_t* graph_ptr, int64_t B, const int64_t* edge_ptr,
                                              int64_t* label, int64_t* index,
                                              int32_t* assign_row_ptr, int32_t* assign_perm, float* ones,
                                              uint64_t* status, int64_t status_words, uint64_t* result, uint32_t epoch,
                                              void* stream_) {
  TGP_REQUIRE(N > 0 && E >= 0 && B > 0 && graph_ptr && index && assign_row_ptr && assign_perm && ones && status &&
                  result && (E == 0 || (row && col)),
              TGP_ERR_INVALID, "tgp_graclus_match_graphs_fused: bad argument");
  TGP_REQUIRE(N < (1ll << 31) && E < (1ll << 31) && B < (1ll << 31), TGP_ERR_RANGE,
              "tgp_graclus_match_graphs_fused: N / E / B >= 2^31");
  TGP_REQUIRE(status_words >= tgp_graclus_match_graphs_fused_status_words(B), TGP_ERR_WORKSPACE,
              "tgp_graclus_match_graphs_fused: status buffer too small");
  TGP_REQUIRE(epoch != 0 && epoch < (1u << 29), TGP_ERR_RANGE, "tgp_graclus_match_graphs_fused: epoch out of range");
  const GfArgs a{row, col, w, N, E, graph_ptr, B, edge_ptr, label, index, assign_row_ptr, assign_perm, ones,
                 reinterpret_cast<unsigned long long*>(status), reinterpret_cast<unsigned long long*>(result),
                 static_cast<unsigned long long>(epoch) << SPS_EPOCH_SHIFT};
  hipLaunchKernelGGL(gm_graph_fused_kernel, dim3(static_cast<unsigned>(cdiv(B, GF_WAVES))), dim3(GF_WAVES * 64), 0,
                     static_cast<hipStream_t>(stream_), a);
  return check_launch("tgp_graclus_match_graphs_fused");
}

extern "C" int tgp_graclus_match_max_graph_nodes(void) { return GM_GRAPH_MAX; }

// All rounds of every graph in one launch (after tgp_graclus_match_start, instead of tgp_graclus_match_rounds), for a
// batch whose graphs own contiguous node ranges graph_ptr[b] .. graph_ptr[b+1] of at most
// tgp_graclus_match_max_graph_nodes() nodes (max_graph_nodes: the caller's bound on the longest graph, which picks
// the workgroup size).  *d_status != 0: not applicable (a graph longer than the bound: 1, an entry that leaves
// its graph: 2) -- the labels are then partial and the caller runs tgp_graclus_match_rounds from a fresh start.
extern "C" int tgp_graclus_match_graphs(const int32_t* row_ptr, int64_t num_nodes, int64_t num_edges, void* ws,
                                        const int64_t* graph_ptr, int64_t B, int64_t max_graph_nodes, int64_t* label,
                                        int* d_status, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(num_nodes >= 0 && num_edges >= 0 && B >= 0 && d_status, TGP_ERR_INVALID,
              "tgp_graclus_match_graphs: bad argument");
  TGP_REQUIRE(max_graph_nodes <= GM_GRAPH_MAX, TGP_ERR_RANGE, "tgp_graclus_match_graphs: graphs too long");
  (void)hipMemsetAsync(d_status, 0, sizeof(int), stream);
  if (num_nodes == 0 || B == 0) return check_launch("tgp_graclus_match_graphs");
  TGP_REQUIRE(row_ptr && ws && graph_ptr && label && B < (1ll << 31), TGP_ERR_INVALID,
              "tgp_graclus_match_graphs: null pointer");
  Carver cv(ws);
  int32_t* nbr = cv.take<int32_t>(num_edges > 0 ? num_edges : 1);
  float* wt = cv.take<float>(num_edges > 0 ? num_edges : 1);
  const bool one_wave = max_graph_nodes <= 64;
  const int cap_n = one_wave ? 64 : GM_GRAPH_MAX, cap_e = one_wave ? 512 : GM_GRAPH_LDS_E;
  const size_t lds = (static_cast<size_t>(cap_n) * 2 + 1 + 2 * cap_e) * 4 + cap_n;
  if (one_wave)
    hipLaunchKernelGGL(gm_graph_rounds_kernel<64>, dim3(static_cast<unsigned>(B)), dim3(64), lds, stream, row_ptr, nbr,
                       wt, graph_ptr, cap_n, cap_e, label, d_status);
  else
    hipLaunchKernelGGL(gm_graph_rounds_kernel<256>, dim3(static_cast<unsigned>(B)), dim3(256), lds, stream, row_ptr, nbr,
                       wt, graph_ptr, cap_n, cap_e, label, d_status);
  return check_launch("tgp_graclus_match_graphs");
}

extern "C" size_t tgp_graclus_match_workspace_bytes(int64_t num_nodes, int64_t num_edges) {
  const size_t n = static_cast<size_t>(num_nodes > 0 ? num_nodes : 1), e = static_cast<size_t>(num_edges > 0 ? num_edges : 1);
  return align_up(e * sizeof(int32_t)) + align_up(e * sizeof(float)) + align_up(n) + align_up(n * sizeof(int32_t)) +
         align_up((n / 32 + 8) * sizeof(uint32_t)) + align_up(e * sizeof(int32_t)) +
         align_up((e / 256 + 2) * sizeof(unsigned long long)) + align_up((GM_TAIL_CAP + 16) * sizeof(int32_t)) +
         align_up((4 + GM_REC * (e / GM_LONG_ROW + e / GM_CHUNK + 8)) * sizeof(int32_t)) + 512;
}

// rows of more than GM_LONG_ROW entries (listed by the init kernel) and {count, next ticket}: behind everything else
static void gm_long_ptrs(void* ws, int64_t num_nodes, int64_t num_edges, int32_t** list, unsigned int** ctr) {
  Carver cv(ws);
  const int64_t e = num_edges > 0 ? num_edges : 1;
  (void)cv.take<int32_t>(e);
  (void)cv.take<float>(e);
  (void)cv.take<uint8_t>(num_nodes);
  (void)cv.take<int32_t>(num_nodes);
  (void)cv.take<uint32_t>(num_nodes / 32 + 8);
  (void)cv.take<int32_t>(e);
  (void)cv.take<unsigned long long>(num_edges / 256 + 2);
  (void)cv.take<int>(4);
  (void)cv.take<int32_t>(GM_TAIL_CAP + 16);
  int32_t* l = cv.take<int32_t>(4 + GM_REC * (e / GM_LONG_ROW + e / GM_CHUNK + 8));
  *ctr = reinterpret_cast<unsigned int*>(l);  // [0] tickets, [1] next ticket; the ticket records start at l + 4
  *list = l + 4;
}

// Start: gathers the CSR and resets the state.  Rounds: runs `rounds` propose/match rounds; matched[r] becomes 1 if
// round r matched anything (device memory, uint32[rounds], zeroed here).  The host reads the last
// entries to decide whether to run more rounds (0 = the matching is maximal).
extern "C" int tgp_graclus_match_start(const int64_t* row, const int64_t* col, const float* weight, const int32_t* row_ptr,
                                       const int32_t* perm, int64_t num_nodes, int64_t num_edges, void* ws,
                                       size_t ws_bytes, int64_t* label, int init_state, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(num_nodes >= 0 && num_edges >= 0, TGP_ERR_INVALID, "tgp_graclus_match_start: negative size");
  if (num_nodes == 0) return TGP_OK;
  TGP_REQUIRE(num_nodes < (1ll << 31) && num_edges < (1ll << 31), TGP_ERR_RANGE, "tgp_graclus_match_start: too large");
  TGP_REQUIRE(label && row_ptr && (num_edges == 0 || (row && col)), TGP_ERR_INVALID,
              "tgp_graclus_match_start: null pointer");
  TGP_REQUIRE(ws && ws_bytes >= tgp_graclus_match_workspace_bytes(num_nodes, num_edges), TGP_ERR_WORKSPACE,
              "tgp_graclus_match_start: workspace too small");
  Carver cv(ws);
  int32_t* nbr = cv.take<int32_t>(num_edges > 0 ? num_edges : 1);
  float* wt = cv.take<float>(num_edges > 0 ? num_edges : 1);
  uint8_t* is_free = cv.take<uint8_t>(num_nodes);
  (void)cv.take<int32_t>(num_nodes);                 // cand   (used by the rounds)
  (void)cv.take<uint32_t>(num_nodes / 32 + 8);       // bitmap (used by the rounds)
  int32_t* src = cv.take<int32_t>(num_edges > 0 ? num_edges : 1);
  unsigned long long* partial = cv.take<unsigned long long>(num_edges / 256 + 2);
  int* asymmetric = cv.take<int>(4);
  if (num_edges > 0) {
    const int nbe = cdiv(num_edges, 256);
    hipLaunchKernelGGL(gm_csr_gather_kernel, dim3(nbe), dim3(256), 0, stream, row, col, weight, perm, num_edges, src,
                       nbr, wt, partial);
    hipLaunchKernelGGL(gm_fingerprint_kernel, dim3(1), dim3(1024), 0, stream, partial, nbe, asymmetric);
    hipLaunchKernelGGL(gm_symmetrise_kernel, dim3(nbe), dim3(256), 0, stream, row_ptr, src, num_edges, asymmetric, nbr,
                       wt);
  }
  // init_state == 0: the caller goes on with tgp_graclus_match_graphs, which sets every label itself and keeps its free
  // flags in LDS (the device-wide rounds need the state: 1)
  if (init_state) {
    int32_t* long_list;
    unsigned int* long_ctr;
    gm_long_ptrs(ws, num_nodes, num_edges, &long_list, &long_ctr);
    (void)hipMemsetAsync(long_ctr, 0, 2 * sizeof(unsigned int), stream);
    hipLaunchKernelGGL(gm_init_kernel, dim3(cdiv(num_nodes, 256)), dim3(256), 0, stream, num_nodes, label, is_free,
                       row_ptr, long_list, long_ctr);
  }
  return check_launch("tgp_graclus_match_start");
}

extern "C" int tgp_graclus_match_rounds(const int32_t* row_ptr, int64_t num_nodes, int64_t num_edges, void* ws,
                                        int rounds, unsigned int* matched, int64_t* label, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(num_nodes >= 0 && num_edges >= 0 && rounds >= 0, TGP_ERR_INVALID, "tgp_graclus_match_rounds: bad size");
  if (num_nodes == 0 || rounds == 0) return TGP_OK;
  TGP_REQUIRE(row_ptr && ws && matched && label, TGP_ERR_INVALID, "tgp_graclus_match_rounds: null pointer");
  Carver cv(ws);
  int32_t* nbr = cv.take<int32_t>(num_edges > 0 ? num_edges : 1);
  float* wt = cv.take<float>(num_edges > 0 ? num_edges : 1);
  uint8_t* is_free = cv.take<uint8_t>(num_nodes);
  int32_t* cand = cv.take<int32_t>(num_nodes);
  (void)hipMemsetAsync(matched, 0, static_cast<size_t>(rounds) * sizeof(unsigned int), stream);
  uint32_t* bits = cv.take<uint32_t>(num_nodes / 32 + 8);
  const int nb = cdiv(num_nodes, 256);
  int32_t* long_list;
  unsigned int* long_ctr;
  gm_long_ptrs(ws, num_nodes, num_edges, &long_list, &long_ctr);
  static const int cus = [] { int v = tgp_device_cu_count(); return v > 0 ? v : 256; }();
  const bool lds_map = num_nodes <= kGmLdsNodes && num_nodes >= 65536;
  const size_t lds_bytes = align_up(static_cast<size_t>((num_nodes + 31) / 32) * sizeof(uint32_t), 16);
  for (int r = 0; r < rounds; ++r) {
    if (lds_map) {
      hipLaunchKernelGGL(gm_pack_free_kernel, dim3(cdiv(num_nodes, 256)), dim3(256), 0, stream, is_free, num_nodes, bits);
      hipLaunchKernelGGL(gm_propose_lds_kernel, dim3(cus), dim3(1024), lds_bytes, stream, row_ptr, nbr, wt, num_nodes,
                         bits, is_free, cand, long_list, long_ctr);
    } else
    hipLaunchKernelGGL(gm_propose_kernel, dim3(nb), dim3(256), 0, stream, row_ptr, nbr, wt, num_nodes, is_free, cand,
                       long_list, long_ctr);
    hipLaunchKernelGGL(gm_match_kernel, dim3(nb), dim3(256), 0, stream, cand, num_nodes, label, is_free, matched + r,
                       long_ctr);
  }
  return check_launch("tgp_graclus_match_rounds");
}

// Every remaining round in two launches when at most 16384 free nodes are left (after tgp_graclus_match_rounds, same
// workspace).  *d_status = 1: the matching is now maximal; 0: too many free nodes, nothing was changed.
extern "C" int tgp_graclus_match_tail(const int32_t* row_ptr, int64_t num_nodes, int64_t num_edges, void* ws,
                                      int64_t* label, int* d_status, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(num_nodes >= 0 && num_edges >= 0 && d_status, TGP_ERR_INVALID, "tgp_graclus_match_tail: bad argument");
  if (num_nodes == 0) {
    (void)hipMemsetAsync(d_status, 0, sizeof(int), stream);
    return check_launch("tgp_graclus_match_tail");
  }
  TGP_REQUIRE(row_ptr && ws && label, TGP_ERR_INVALID, "tgp_graclus_match_tail: null pointer");
  Carver cv(ws);
  int32_t* nbr = cv.take<int32_t>(num_edges > 0 ? num_edges : 1);
  float* wt = cv.take<float>(num_edges > 0 ? num_edges : 1);
  uint8_t* is_free = cv.take<uint8_t>(num_nodes);
  int32_t* cand = cv.take<int32_t>(num_nodes);
  (void)cv.take<uint32_t>(num_nodes / 32 + 8);
  (void)cv.take<int32_t>(num_edges > 0 ? num_edges : 1);
  (void)cv.take<unsigned long long>(num_edges / 256 + 2);
  (void)cv.take<int>(4);
  int32_t* list = cv.take<int32_t>(GM_TAIL_CAP + 16);
  unsigned int* count = reinterpret_cast<unsigned int*>(list + GM_TAIL_CAP);
  (void)hipMemsetAsync(count, 0, sizeof(unsigned int), stream);
  hipLaunchKernelGGL(gm_tail_list_kernel, dim3(cdiv(num_nodes, 256)), dim3(256), 0, stream, is_free, num_nodes, list,
                     count);
  hipLaunchKernelGGL(gm_tail_rounds_kernel, dim3(1), dim3(1024), 0, stream, row_ptr, nbr, wt, list, count, label,
                     is_free, cand, d_status);
  return check_launch("tgp_graclus_match_tail");
}
