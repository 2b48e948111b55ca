// A11: sparse -> padded dense (the step right before the timed path).
// Reference: DenseSRCPooling.preprocessing (tgp/src.py:374-452) calls PyG to_dense_adj (scatter-add of the
// edge weights into zeros [B,Nmax,Nmax], duplicates summed) and to_dense_batch (row scatter + mask).  Here:
// one memset + one edge-parallel kernel, one memset + one row-parallel kernel.  The transposed adjacency
// the reference builds as a view (src.py:442-443) is written directly, so the GEMMs read a plain layout.
#include "common.h"

namespace tgp {

__global__ __launch_bounds__(256) void dense_adj_kernel(const int64_t* __restrict__ row,
                                                        const int64_t* __restrict__ col,
                                                        const float* __restrict__ w, int64_t E,
                                                        const int64_t* __restrict__ batch,
                                                        const int64_t* __restrict__ ptr, int64_t Nmax,
                                                        int transposed, float* __restrict__ adj) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (e >= E) return;
  const int64_t r = row[e], c = col[e];
  const int64_t b = batch[r];
  const int64_t lr = r - ptr[b], lc = c - ptr[batch[c]];
  if (lr >= Nmax || lc >= Nmax) return;  // PyG drops entries beyond max_num_nodes
  const int64_t o = transposed ? (b * Nmax + lc) * Nmax + lr : (b * Nmax + lr) * Nmax + lc;
  // no-return float atomic; duplicates (the only contended case) are summed as PyG's scatter-add does
  atomicAdd(adj + o, w ? w[e] : 1.0f);
}

// Multi-channel edge attributes [E, C] -> [B, Nmax, Nmax, C] (PyG to_dense_adj with a 2-D edge_attr, src.py:434): one
// thread per (edge, channel), channels contiguous, duplicates summed.  (r5: the last torch-form scatter of the path.)
__global__ __launch_bounds__(256) void dense_adj_channels_kernel(const int64_t* __restrict__ row,
                                                                 const int64_t* __restrict__ col,
                                                                 const float* __restrict__ attr, int64_t E, int64_t C,
                                                                 const int64_t* __restrict__ batch,
                                                                 const int64_t* __restrict__ ptr, int64_t Nmax,
                                                                 int transposed, float* __restrict__ adj) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (t >= E * C) return;
  const int64_t e = t / C, ch = t - e * C;
  const int64_t r = row[e], c = col[e];
  const int64_t b = batch[r];
  const int64_t lr = r - ptr[b], lc = c - ptr[batch[c]];
  if (lr >= Nmax || lc >= Nmax) return;
  const int64_t o = transposed ? (b * Nmax + lc) * Nmax + lr : (b * Nmax + lr) * Nmax + lc;
  atomicAdd(adj + o * C + ch, attr[t]);
}

// The inverse gather (backward of to_dense_adj w.r.t. the edge weights): dw[e] = g[b, r, c] at the slot edge e was
// added to; 0 for entries a caller-imposed max_num_nodes dropped.  Duplicates each receive the slot's gradient.
__global__ __launch_bounds__(256) void from_dense_adj_kernel(const float* __restrict__ g, const int64_t* __restrict__ row,
                                                             const int64_t* __restrict__ col, int64_t E,
                                                             const int64_t* __restrict__ batch,
                                                             const int64_t* __restrict__ ptr, int64_t Nmax,
                                                             int transposed, float* __restrict__ dw) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (e >= E) return;
  const int64_t r = row[e], c = col[e];
  const int64_t b = batch[r];
  const int64_t lr = r - ptr[b], lc = c - ptr[batch[c]];
  if (lr >= Nmax || lc >= Nmax) {
    dw[e] = 0.f;
    return;
  }
  dw[e] = g[transposed ? (b * Nmax + lc) * Nmax + lr : (b * Nmax + lr) * Nmax + lc];
}

__global__ __launch_bounds__(256) void dense_batch_kernel(const float* __restrict__ x, int64_t N, int64_t F,
                                                          const int64_t* __restrict__ batch,
                                                          const int64_t* __restrict__ ptr, int64_t Nmax,
                                                          float* __restrict__ out, uint8_t* __restrict__ mask) {
  const int64_t total = N * F;
  for (int64_t o = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; o < total;
       o += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t i = o / F, f = o - i * F;
    const int64_t b = batch[i];
    const int64_t li = i - ptr[b];
    if (li >= Nmax) continue;
    out[(b * Nmax + li) * F + f] = x[o];
    if (f == 0 && mask) mask[b * Nmax + li] = 1;
  }
}

// The same for a SORTED batch vector (graph b owns nodes ptr[b] .. ptr[b+1]): output-parallel, every padded row
// written once -- real rows copied, padding zeroed, the mask set -- so no memsets run in front (r3: two launches fewer
// per dense pooler call).  One lane per 4 bytes of an output row.
__global__ __launch_bounds__(256) void dense_batch_sorted_kernel(const float* __restrict__ x, int64_t F,
                                                                 const int64_t* __restrict__ ptr, int64_t B,
                                                                 int64_t Nmax, float* __restrict__ out,
                                                                 uint8_t* __restrict__ mask,
                                                                 float* __restrict__ zero_buf, int64_t zero_count) {
  // (r3) the same launch zero-fills a second buffer, the [B,Nmax,Nmax] adjacency the edge scatter adds into next: both
  // are pure streaming writes, and a dependent launch costs ~5 us on this part whatever it does
  {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t n4 = zero_count / 4;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n4; i += static_cast<int64_t>(gridDim.x) * 256)
      reinterpret_cast<float4*>(zero_buf)[i] = z;
    for (int64_t i = n4 * 4 + static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < zero_count;
         i += static_cast<int64_t>(gridDim.x) * 256)
      zero_buf[i] = 0.f;
  }
  const int64_t total = B * Nmax * F;
  for (int64_t o = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; o < total;
       o += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t row = o / F, f = o - row * F;
    const int64_t b = row / Nmax, li = row - b * Nmax;
    const int64_t p0 = ptr[b], n = ptr[b + 1] - p0;
    const bool real = li < n;
    out[o] = real ? x[(p0 + li) * F + f] : 0.f;
    if (f == 0 && mask) mask[row] = real ? 1 : 0;
  }
}

// The inverse gather (backward of to_dense_batch): x[i,:] = dense[batch[i], i - ptr[batch[i]], :], zero for nodes that
// a caller-imposed max_num_nodes dropped.
__global__ __launch_bounds__(256) void from_dense_batch_kernel(const float* __restrict__ dense, int64_t N, int64_t F,
                                                               const int64_t* __restrict__ batch,
                                                               const int64_t* __restrict__ ptr, int64_t Nmax,
                                                               float* __restrict__ x) {
  const int64_t total = N * F;
  for (int64_t o = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; o < total;
       o += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t i = o / F, f = o - i * F;
    const int64_t b = batch[i];
    const int64_t li = i - ptr[b];
    x[o] = li < Nmax ? dense[(b * Nmax + li) * F + f] : 0.f;
  }
}

// Facts about a batch vector in one pass + a small second launch, for ONE host read (utils/ops.py batch_info; as torch
// ops: bincount -- which synchronises for its output length --, a comparison, any, cat, a second synchronising copy).
// sizes [N + 1] must be zero; facts = {max id, unsorted flag | 2 * (an id outside [0, N]), longest graph, non-empty graphs}.
// Runs of equal ids inside a wave add their length with ONE atomic (a sorted batch of 40-node graphs: 2-3 per wave).
__global__ __launch_bounds__(256) void batch_facts_kernel(const int64_t* __restrict__ batch, int64_t n,
                                                          unsigned long long* __restrict__ sizes,
                                                          unsigned long long* __restrict__ facts) {
  const int lane = threadIdx.x & 63;
  for (int64_t base = (static_cast<int64_t>(blockIdx.x) * 256 + (threadIdx.x & ~63)); base < n;
       base += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t i = base + lane;
    const bool in = i < n;
    const int64_t b = in ? batch[i] : 0;
    const int64_t prev = (in && i > 0) ? batch[i - 1] : b;
    const bool bad = in && (b < 0 || b > n);
    const bool head = in && (lane == 0 || b != prev);
    const unsigned long long heads = __ballot(head), valid = __ballot(in);
    if (__ballot(in && b < prev) && lane == 0) atomicOr(facts + 1, 1ull);
    if (__ballot(bad)) {
      if (lane == 0) atomicOr(facts + 1, 2ull);
      continue;
    }
    if (head) {
      const unsigned long long above = heads & ~((2ull << lane) - 1ull);  // heads after this lane
      const int end = above ? __ffsll(static_cast<long long>(above)) - 1 : __popcll(valid);
      atomicAdd(sizes + b, static_cast<unsigned long long>(end - lane));
    }
    long long mx = in ? b : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const long long t = __shfl_xor(mx, o, 64);
      mx = t > mx ? t : mx;
    }
    if (lane == 0) atomicMax(facts, static_cast<unsigned long long>(mx));
  }
}

__global__ __launch_bounds__(256) void batch_facts_finish_kernel(const unsigned long long* __restrict__ sizes, int64_t n,
                                                                 unsigned long long* __restrict__ facts, float ratio) {
  if (facts[1] & 2ull) return;
  const int64_t B = static_cast<int64_t>(facts[0]) + 1;  // (written by the launch before this one)
  unsigned long long mx = 0, cnt = 0, keep = 0;
  for (int64_t g = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; g < B && g <= n;
       g += static_cast<int64_t>(gridDim.x) * 256) {
    const unsigned long long v = sizes[g];
    mx = v > mx ? v : mx;
    cnt += v > 0 ? 1 : 0;
    if (ratio > 0.f) {  // TopkSelect's k_g, the arithmetic of topk_plan_kernel (PyG: ceil(fp32(ratio) * n_g) / min)
      if (ratio >= 1.0f) {
        const unsigned long long r = static_cast<unsigned long long>(ratio);
        keep += r < v ? r : v;
      } else {
        keep += static_cast<unsigned long long>(ceilf(__fmul_rn(ratio, static_cast<float>(v))));
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long t = __shfl_xor(mx, o, 64);
    mx = t > mx ? t : mx;
    cnt += __shfl_xor(cnt, o, 64);
    keep += __shfl_xor(keep, o, 64);
  }
  if ((threadIdx.x & 63) == 0 && (mx | cnt)) {
    atomicMax(facts + 2, mx);
    atomicAdd(facts + 3, cnt);
    if (keep) atomicAdd(facts + 4, keep);
  }
}

// r5: ALL facts of a SORTED batch vector in ONE launch, no memset, no copy back (verdict r4 item 2: a forward on new
// tensor objects paid two memsets, two kernels, a device-to-host copy, a fill and a two-kernel cumsum for them: ~60 us of
// device time on 2048 small graphs).  A sorted vector needs no counting: node i with batch[i] != batch[i-1] is the first
// node of its graph, so ptr[g] = i for every graph id g in (batch[i-1], batch[i]] (ids without nodes get an empty range)
// -- every entry of ptr [B + 1] is written exactly once, by one thread.  The workgroup that takes the last ticket then
// turns ptr into sizes, the longest graph, the number of non-empty graphs and, for a TopK selector, k_g =
// ceil(fp32(ratio) n_g) with its prefix sums (the arithmetic of topk_plan_kernel), and stores
// {tag, B - 1, flags, longest, non-empty, sum k_g} into pinned host memory: the host polls word 0.
// flags: 1 = the vector is NOT sorted, 2 = an id outside [0, N], 4 = more than 64 consecutive graph ids without a node
// (any of them: nothing else is meaningful, the caller takes the general route).  `ticket` is a device word per (device, stream) that is zero between calls (the last workgroup resets it).
__global__ __launch_bounds__(256) void batch_facts_sorted_kernel(const int64_t* __restrict__ batch, int64_t n,
                                                                 int64_t* __restrict__ ptr, int64_t* __restrict__ sizes,
                                                                 float ratio, int64_t* __restrict__ k,
                                                                 int64_t* __restrict__ koff,
                                                                 unsigned int* __restrict__ ticket,
                                                                 unsigned int* __restrict__ bad,
                                                                 unsigned long long* __restrict__ result,
                                                                 unsigned long long tag) {
  constexpr int IT = 8;  // graphs per thread and round of the last workgroup: their offsets are requested together
  __shared__ long long s_w[4][3];
  __shared__ long long s_carry;
  __shared__ bool s_last;
  unsigned int flags = 0;
  // eight elements per thread and round, all sixteen loads (the element and its predecessor) requested before the first
  // comparison (r5, late: one element per round was load -> wait -> compare -> store, eight dependent round trips per
  // thread for the usual 2048 elements of a workgroup -- most of this kernel's 12-14 us)
  constexpr int UN = 8;
  for (int64_t i0 = static_cast<int64_t>(blockIdx.x) * 256 * UN + threadIdx.x; i0 < n;
       i0 += static_cast<int64_t>(gridDim.x) * 256 * UN) {
    int64_t bv[UN], pv[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t i = i0 + u * 256;
      const int64_t ic = i < n ? i : n - 1;      // (clamped: unconditional loads)
      bv[u] = batch[ic];
      pv[u] = batch[ic > 0 ? ic - 1 : 0];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t i = i0 + u * 256;
      if (i >= n) continue;
      const int64_t b = bv[u];
      const int64_t prev = i > 0 ? pv[u] : -1;
      if (b < 0 || b > n) { flags |= 2u; continue; }
      if (b < prev) { flags |= 1u; continue; }
      if (prev < 0 && i > 0) continue;              // (the predecessor is out of range: its own thread reports it)
      if (b - prev > 64) { flags |= 4u; continue; } // a long run of ids without nodes: left to the general route
      for (int64_t g = prev + 1; g <= b; ++g) ptr[g] = i;   // first node of graph b; empty ranges for skipped ids
      if (i == n - 1) ptr[b + 1] = n;
    }
  }
  if (flags) atomicOr(bad, flags);
  // hand-off to the workgroup that takes the last ticket (MI355X_MICROARCH.md, "Valid forms"): every storing wave drains
  // its stores, the workgroup meets, ONE lane releases at agent scope and takes the ticket
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    if (s_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  if (!s_last) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const unsigned int fl = __hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  long long mx = 0, cnt = 0, keep = 0;
  int64_t B = 0;
  if (!fl) {
    B = batch[n - 1] + 1;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < B; base += 256 * IT) {
      const int64_t g0 = base + static_cast<int64_t>(tid) * IT;
      long long pv[IT + 1];
#pragma unroll
      for (int q = 0; q <= IT; ++q) pv[q] = g0 + q <= B ? ptr[g0 + q] : 0;   // one round trip for the thread's graphs
      // (all of them have landed before the first store below: vmcnt also counts stores not yet acknowledged, and with
      //  the stores in conditional blocks every later wait for one of these loads would be a wait for the stores too)
      __builtin_amdgcn_s_waitcnt(0x0F70);
      long long kv[IT], mine = 0;
#pragma unroll
      for (int q = 0; q < IT; ++q) {
        kv[q] = 0;
        if (g0 + q < B) {
          const long long v = pv[q + 1] - pv[q];
          sizes[g0 + q] = v;
          mx = v > mx ? v : mx;
          cnt += v > 0 ? 1 : 0;
          if (ratio > 0.f) {  // PyG topk: ceil(fp32(ratio) * n_g), or min(ratio, n_g) for ratio >= 1
            if (ratio >= 1.0f) {
              const long long r = static_cast<long long>(ratio);
              kv[q] = r < v ? r : v;
            } else {
              kv[q] = static_cast<long long>(ceilf(__fmul_rn(ratio, static_cast<float>(v))));
            }
            if (k) k[g0 + q] = kv[q];
          }
        }
        mine += kv[q];
      }
      keep += mine;
      if (koff) {  // exclusive prefix of k over the graphs, in graph order: thread-local, then over the workgroup
        long long inc = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const long long o = __shfl_up(inc, off, WAVE);
          if (lane >= off) inc += o;
        }
        if (lane == 63) s_w[w][0] = inc;
        __syncthreads();
        long long before = s_carry, tot = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const long long c = s_w[q][0];
          if (q < w) before += c;
          tot += c;
        }
        long long run = before + inc - mine;
#pragma unroll
        for (int q = 0; q < IT; ++q) {
          if (g0 + q < B) koff[g0 + q] = run;
          run += kv[q];
        }
        __syncthreads();
        if (tid == 0) s_carry += tot;
        __syncthreads();
      }
    }
    if (koff && tid == 0) koff[B] = s_carry;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const long long t = __shfl_xor(mx, o, 64);
    mx = t > mx ? t : mx;
    cnt += __shfl_xor(cnt, o, 64);
    keep += __shfl_xor(keep, o, 64);
  }
  __syncthreads();
  if (lane == 0) { s_w[w][0] = mx; s_w[w][1] = cnt; s_w[w][2] = keep; }
  __syncthreads();
  if (tid == 0) {
    for (int q = 1; q < 4; ++q) {
      mx = s_w[q][0] > mx ? s_w[q][0] : mx;
      cnt += s_w[q][1];
      keep += s_w[q][2];
    }
    *ticket = 0;  // ready for the next call on this stream
    *bad = 0;
    result[1] = static_cast<unsigned long long>(B - 1);
    result[2] = fl;
    result[3] = static_cast<unsigned long long>(mx);
    result[4] = static_cast<unsigned long long>(cnt);
    result[5] = static_cast<unsigned long long>(keep);
    __threadfence_system();
    __hip_atomic_store(result, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// r5: what the sparse-input fused call (tgp_dense_pool_select_sparse_f32) needs to know about a NEW edge list, from one
// launch and without a host round trip in front of the consumer: edge_ptr[g] = first entry whose source node belongs
// to a graph >= g (entry e with graph(row[e]) != graph(row[e-1]) writes the offsets of its graph and of the empty ids in
// between, at most 64; the last entry writes the tail up to B = batch[n-1] + 1), and -- through the workgroup with the
// last ticket -- ONE flag word in pinned host memory: 1 = the rows are not grouped by ascending source node (or the
// batch vector is not sorted), 2 = a source id outside [0, n), 4 = a run of more than 64 graph ids without an entry.
// The consumer may be launched behind this kernel at once (it clamps what it reads through edge_ptr); the host looks
// at the flags afterwards and discards the consumer's outputs when they are set.
__global__ __launch_bounds__(256) void edge_facts_sorted_kernel(const int64_t* __restrict__ row, int64_t E,
                                                                const int64_t* __restrict__ batch, int64_t n,
                                                                int64_t* __restrict__ edge_ptr,
                                                                unsigned int* __restrict__ ticket,
                                                                unsigned int* __restrict__ bad,
                                                                unsigned long long* __restrict__ result,
                                                                unsigned long long tag) {
  __shared__ bool s_last;
  unsigned int flags = 0;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; e < E; e += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t r = row[e];
    if (r < 0 || r >= n) { flags |= 2u; continue; }
    const int64_t rp = e > 0 ? row[e - 1] : -1;
    if (e > 0 && (rp < 0 || rp >= n)) continue;  // (the predecessor's thread reports it)
    if (r < rp) { flags |= 1u; continue; }
    const int64_t g = batch[r], gp = e > 0 ? batch[rp] : -1;
    if (g < gp || g < 0 || g > n) { flags |= 1u; continue; }
    if (g - gp > 64) { flags |= 4u; continue; }
    for (int64_t gg = gp + 1; gg <= g; ++gg) edge_ptr[gg] = e;
    if (e == E - 1) {
      const int64_t B = batch[n - 1] + 1;
      // edge_ptr holds n + 2 entries: a last graph id beyond the node count is a malformed batch vector (as `g > n`
      // above), reported and nothing written
      if (B < 0 || B > n + 1) flags |= 1u;
      else if (B < g || B - g > 64) flags |= 4u;
      else
        for (int64_t gg = g + 1; gg <= B; ++gg) edge_ptr[gg] = E;
    }
  }
  if (flags) atomicOr(bad, flags);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    if (s_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned int fl = __hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *ticket = 0;  // ready for the next call on this stream
      *bad = 0;
      result[1] = 0ull;
      result[2] = fl;
      result[3] = result[4] = result[5] = 0ull;
      __threadfence_system();
      __hip_atomic_store(result, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// r6: is the dense adjacency the edge list was scattered into SYMMETRIC?  One thread per entry compares adj[b,r,c] with
// adj[b,c,r] (exact: a sum of duplicates that came out in another order counts as a mismatch); every other element of
// the buffer is zero on both sides.  The answer lets the dense poolers' training step skip V = A^T S, the second of its
// two N^2 K products (V = U for a symmetric A).  Verdict through the pinned host words of edge_facts_sorted_kernel's
// protocol (word 2: 1 = some entry differs from its mirror image): launched in the forward, read in the backward.
__global__ __launch_bounds__(256) void adj_symmetry_kernel(const int64_t* __restrict__ row, const int64_t* __restrict__ col,
                                                           int64_t E, const int64_t* __restrict__ batch,
                                                           const int64_t* __restrict__ ptr, int64_t Nmax,
                                                           const float* __restrict__ adj,
                                                           unsigned int* __restrict__ ticket,
                                                           unsigned int* __restrict__ bad,
                                                           unsigned long long* __restrict__ result,
                                                           unsigned long long tag) {
  __shared__ bool s_last;
  unsigned int flags = 0;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; e < E; e += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t r = row[e], c = col[e];
    const int64_t b = batch[r];
    const int64_t lr = r - ptr[b], lc = c - ptr[batch[c]];
    if (lr >= Nmax || lc >= Nmax || lr < 0 || lc < 0) continue;  // (dropped by the scatter as well)
    const float* g = adj + b * Nmax * Nmax;
    if (g[lr * Nmax + lc] != g[lc * Nmax + lr]) flags = 1u;
  }
  if (flags) atomicOr(bad, flags);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    if (s_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned int fl = __hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *ticket = 0;  // ready for the next call on this stream
      *bad = 0;
      result[1] = 0ull;
      result[2] = fl;
      result[3] = result[4] = result[5] = 0ull;
      __threadfence_system();
      __hip_atomic_store(result, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

}  // namespace tgp

using namespace tgp;

// One-launch form for a sorted batch vector (see batch_facts_sorted_kernel).  ptr [N + 2], sizes [N + 1]; k [N + 1] /
// koff [N + 2] optional (TopkSelect's plan for `topk_ratio` > 0); `ticket`: two zeroed uint32 words owned by the caller
// per (device, stream), left zero by the call; `result`: 6 words of pinned host memory, word 0 = `tag` stored last.
extern "C" int tgp_batch_facts_sorted_i64(const int64_t* batch, int64_t N, int64_t* ptr, int64_t* sizes,
                                          double topk_ratio, int64_t* k, int64_t* koff, uint32_t* ticket,
                                          uint64_t* result, uint64_t tag, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(N > 0 && batch && ptr && sizes && ticket && result, TGP_ERR_INVALID,
              "tgp_batch_facts_sorted_i64: bad argument");
  // few, fat workgroups: every workgroup ends with a release fence and a ticket on ONE word (~12 ns each, serialised)
  int64_t blocks = (N + 2047) / 2048;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(batch_facts_sorted_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, batch, N, ptr,
                     sizes, static_cast<float>(topk_ratio > 0.0 ? topk_ratio : 0.0), k, koff, ticket, ticket + 1,
                     reinterpret_cast<unsigned long long*>(result), static_cast<unsigned long long>(tag));
  return check_launch("tgp_batch_facts_sorted_i64");
}

// batch [N] int64 -> sizes [N + 1] int64 (graph g's node count at sizes[g]; the caller keeps sizes[:B]) and
// facts int64[5] = {B - 1, bit 0: not sorted / bit 1: an id outside [0, N] (sizes are then meaningless), longest graph,
// number of non-empty graphs, sum_g k_g of TopkSelect for `topk_ratio` (0: not asked)}: everything utils/ops.py batch_info
// and a TopK selector behind it read back, in one copy.
extern "C" int tgp_batch_facts_i64(const int64_t* batch, int64_t N, int64_t* sizes, int64_t* facts, double topk_ratio,
                                   void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(N >= 0 && sizes && facts, TGP_ERR_INVALID, "tgp_batch_facts_i64: bad argument");
  (void)hipMemsetAsync(sizes, 0, sizeof(int64_t) * static_cast<size_t>(N + 1), stream);
  (void)hipMemsetAsync(facts, 0, sizeof(int64_t) * 5, stream);
  if (N == 0) return check_launch("tgp_batch_facts_i64");
  TGP_REQUIRE(batch, TGP_ERR_INVALID, "tgp_batch_facts_i64: null pointer");
  int64_t blocks = (N + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(batch_facts_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, batch, N,
                     reinterpret_cast<unsigned long long*>(sizes), reinterpret_cast<unsigned long long*>(facts));
  hipLaunchKernelGGL(batch_facts_finish_kernel, dim3(static_cast<unsigned>(blocks < 256 ? blocks : 256)), dim3(256), 0,
                     stream, reinterpret_cast<const unsigned long long*>(sizes), N,
                     reinterpret_cast<unsigned long long*>(facts), static_cast<float>(topk_ratio > 0.0 ? topk_ratio : 0.0));
  return check_launch("tgp_batch_facts_i64");
}

extern "C" int tgp_to_dense_adj_f32(const int64_t* row, const int64_t* col, const float* w, int64_t E,
                                    const int64_t* batch, const int64_t* ptr, int64_t B, int64_t Nmax,
                                    int transposed, int adj_is_zeroed, float* adj, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(E >= 0 && B >= 0 && Nmax >= 0, TGP_ERR_INVALID, "tgp_to_dense_adj_f32: negative size");
  if (B == 0 || Nmax == 0) return TGP_OK;
  TGP_REQUIRE(adj && (E == 0 || (row && col && batch && ptr)), TGP_ERR_INVALID, "tgp_to_dense_adj_f32: null pointer");
  if (!adj_is_zeroed) (void)hipMemsetAsync(adj, 0, sizeof(float) * B * Nmax * Nmax, stream);
  if (E > 0)
    hipLaunchKernelGGL(dense_adj_kernel, dim3(cdiv(E, 256)), dim3(256), 0, stream, row, col, w, E, batch, ptr, Nmax,
                       transposed, adj);
  return check_launch("tgp_to_dense_adj_f32");
}

extern "C" int tgp_to_dense_adj_channels_f32(const int64_t* row, const int64_t* col, const float* attr, int64_t E,
                                             int64_t C, const int64_t* batch, const int64_t* ptr, int64_t B,
                                             int64_t Nmax, int transposed, float* adj, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(E >= 0 && C >= 0 && B >= 0 && Nmax >= 0, TGP_ERR_INVALID, "tgp_to_dense_adj_channels_f32: negative size");
  if (B == 0 || Nmax == 0 || C == 0) return TGP_OK;
  TGP_REQUIRE(adj && (E == 0 || (row && col && attr && batch && ptr)), TGP_ERR_INVALID,
              "tgp_to_dense_adj_channels_f32: null pointer");
  (void)hipMemsetAsync(adj, 0, sizeof(float) * B * Nmax * Nmax * C, stream);
  if (E > 0)
    hipLaunchKernelGGL(dense_adj_channels_kernel, dim3(cdiv(E * C, 256)), dim3(256), 0, stream, row, col, attr, E, C,
                       batch, ptr, Nmax, transposed, adj);
  return check_launch("tgp_to_dense_adj_channels_f32");
}

extern "C" int tgp_to_dense_batch_f32(const float* x, int64_t N, int64_t F, const int64_t* batch,
                                      const int64_t* ptr, int64_t B, int64_t Nmax, float* out, uint8_t* mask,
                                      void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(N >= 0 && F >= 0 && B >= 0 && Nmax >= 0, TGP_ERR_INVALID, "tgp_to_dense_batch_f32: negative size");
  if (B == 0 || Nmax == 0) return TGP_OK;
  TGP_REQUIRE(out && (N == 0 || (x && batch && ptr)), TGP_ERR_INVALID, "tgp_to_dense_batch_f32: null pointer");
  (void)hipMemsetAsync(out, 0, sizeof(float) * B * Nmax * F, stream);
  if (mask) (void)hipMemsetAsync(mask, 0, static_cast<size_t>(B) * Nmax, stream);
  if (N > 0 && F > 0) {
    int64_t blocks = (N * F + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(dense_batch_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, x, N, F, batch,
                       ptr, Nmax, out, mask);
  }
  return check_launch("tgp_to_dense_batch_f32");
}

extern "C" int tgp_to_dense_batch_sorted_f32(const float* x, int64_t N, int64_t F, const int64_t* ptr, int64_t B,
                                             int64_t Nmax, float* out, uint8_t* mask, float* zero_buf,
                                             int64_t zero_count, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(N >= 0 && F >= 0 && B >= 0 && Nmax >= 0, TGP_ERR_INVALID, "tgp_to_dense_batch_sorted_f32: negative size");
  if (B == 0 || Nmax == 0) return TGP_OK;
  TGP_REQUIRE(out && ptr && (N == 0 || x), TGP_ERR_INVALID, "tgp_to_dense_batch_sorted_f32: null pointer");
  TGP_REQUIRE(zero_count >= 0 && (zero_count == 0 || (zero_buf && reinterpret_cast<uintptr_t>(zero_buf) % 16 == 0)),
              TGP_ERR_INVALID, "tgp_to_dense_batch_sorted_f32: zero_buf must be 16-byte aligned");
  if (F == 0) {
    if (mask) (void)hipMemsetAsync(mask, 0, static_cast<size_t>(B) * Nmax, stream);  // (no feature column to ride on)
    if (zero_count) (void)hipMemsetAsync(zero_buf, 0, sizeof(float) * zero_count, stream);
    return check_launch("tgp_to_dense_batch_sorted_f32");
  }
  const int64_t work = B * Nmax * F > zero_count / 4 ? B * Nmax * F : zero_count / 4;
  int64_t blocks = (work + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(dense_batch_sorted_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, x, F, ptr, B,
                     Nmax, out, mask, zero_buf, zero_count);
  return check_launch("tgp_to_dense_batch_sorted_f32");
}

extern "C" int tgp_from_dense_adj_f32(const float* grad_adj, const int64_t* row, const int64_t* col, int64_t E,
                                      const int64_t* batch, const int64_t* ptr, int64_t B, int64_t Nmax,
                                      int transposed, float* grad_weight, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(E >= 0 && B >= 0 && Nmax >= 0, TGP_ERR_INVALID, "tgp_from_dense_adj_f32: negative size");
  if (E == 0) return TGP_OK;
  TGP_REQUIRE(grad_adj && row && col && batch && ptr && grad_weight, TGP_ERR_INVALID,
              "tgp_from_dense_adj_f32: null pointer");
  hipLaunchKernelGGL(from_dense_adj_kernel, dim3(cdiv(E, 256)), dim3(256), 0, stream, grad_adj, row, col, E, batch, ptr,
                     Nmax, transposed, grad_weight);
  return check_launch("tgp_from_dense_adj_f32");
}

extern "C" int tgp_from_dense_batch_f32(const float* dense, int64_t N, int64_t F, const int64_t* batch,
                                        const int64_t* ptr, int64_t B, int64_t Nmax, float* x, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(N >= 0 && F >= 0 && B >= 0 && Nmax >= 0, TGP_ERR_INVALID, "tgp_from_dense_batch_f32: negative size");
  if (N == 0 || F == 0) return TGP_OK;
  TGP_REQUIRE(x && batch && ptr && (dense || B == 0 || Nmax == 0), TGP_ERR_INVALID,
              "tgp_from_dense_batch_f32: null pointer");
  if (B == 0 || Nmax == 0) {
    (void)hipMemsetAsync(x, 0, sizeof(float) * N * F, stream);
    return check_launch("tgp_from_dense_batch_f32");
  }
  int64_t blocks = (N * F + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(from_dense_batch_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, dense, N, F,
                     batch, ptr, Nmax, x);
  return check_launch("tgp_from_dense_batch_f32");
}

// edge_ptr [N + 2] (entries [0, B] are written); ticket: two zeroed uint32 words owned by the caller per (device,
// stream) -- not the pair tgp_batch_facts_sorted_i64 uses --, left zero; result: 6 words of pinned host memory, word 0 =
// `tag` stored last, word 2 = the flags (see edge_facts_sorted_kernel).  E > 0, N > 0.
extern "C" int tgp_edge_facts_sorted_i64(const int64_t* row, int64_t E, const int64_t* batch, int64_t N, int64_t* edge_ptr,
                                         uint32_t* ticket, uint64_t* result, uint64_t tag, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(E > 0 && N > 0 && row && batch && edge_ptr && ticket && result, TGP_ERR_INVALID,
              "tgp_edge_facts_sorted_i64: bad argument");
  int64_t blocks = (E + 2047) / 2048;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(edge_facts_sorted_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, row, E, batch, N,
                     edge_ptr, ticket, ticket + 1, reinterpret_cast<unsigned long long*>(result),
                     static_cast<unsigned long long>(tag));
  return check_launch("tgp_edge_facts_sorted_i64");
}

// r6: and for a dense [B,N,N] adjacency the caller holds (a fixed dense graph pooled every epoch): 32 x 32 tiles on and
// above the diagonal are compared with their mirror tiles through an LDS patch; one pass over the matrix, once per tensor
// object (the answer is remembered).
namespace tgp {
__global__ __launch_bounds__(256) void dense_symmetry_kernel(const float* __restrict__ adj, int N, int tiles,
                                                             unsigned int* __restrict__ ticket,
                                                             unsigned int* __restrict__ bad,
                                                             unsigned long long* __restrict__ result,
                                                             unsigned long long tag) {
  __shared__ float t_m[32][33];
  __shared__ bool s_last;
  const int b = blockIdx.y;
  const float* A = adj + static_cast<long>(b) * N * N;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  unsigned int flags = 0;
  // upper-triangular tile pairs (ti <= tj), dealt out over the grid's x dimension
  const long pairs = static_cast<long>(tiles) * (tiles + 1) / 2;
  for (long pi = blockIdx.x; pi < pairs; pi += gridDim.x) {
    // (ti, tj) of the pi-th pair in row-major order of the upper triangle
    int ti = 0;
    long rem = pi;
    while (rem >= tiles - ti) { rem -= tiles - ti; ++ti; }
    const int tj = ti + static_cast<int>(rem);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // mirror tile (tj, ti), read along its rows
      const int r = tj * 32 + ty * 4 + q, c = ti * 32 + tx;
      t_m[ty * 4 + q][tx] = (r < N && c < N) ? A[static_cast<long>(r) * N + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = ti * 32 + ty * 4 + q, c = tj * 32 + tx;
      if (r < N && c < N && A[static_cast<long>(r) * N + c] != t_m[tx][ty * 4 + q]) flags = 1u;
    }
  }
  if (flags) atomicOr(bad, flags);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_last = atomicAdd(ticket, 1u) == gridDim.x * gridDim.y - 1;
    if (s_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned int fl = __hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *ticket = 0;
      *bad = 0;
      result[1] = 0ull;
      result[2] = fl;
      result[3] = result[4] = result[5] = 0ull;
      __threadfence_system();
      __hip_atomic_store(result, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
}  // namespace tgp

extern "C" int tgp_dense_symmetry_f32(const float* adj, int64_t B, int64_t N, uint32_t* ticket, uint64_t* result,
                                      uint64_t tag, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(B > 0 && N > 0 && B < 65536 && N < (1ll << 20) && adj && ticket && result, TGP_ERR_INVALID,
              "tgp_dense_symmetry_f32: bad argument");
  const int tiles = static_cast<int>((N + 31) / 32);
  int64_t gx = static_cast<int64_t>(tiles) * (tiles + 1) / 2;
  if (gx * B > 4096) gx = (4096 + B - 1) / B;  // a few workgroups per CU, every one walks several tile pairs
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(dense_symmetry_kernel, dim3(static_cast<unsigned>(gx), static_cast<unsigned>(B)), dim3(256), 0, stream, adj,
                     static_cast<int>(N), tiles, ticket, ticket + 1, reinterpret_cast<unsigned long long*>(result),
                     static_cast<unsigned long long>(tag));
  return check_launch("tgp_dense_symmetry_f32");
}

// r6: the same question for a COALESCED row-sorted edge list with its CSR offsets (the unbatched dense poolers never
// densify): entry (r, c, w) must have a mirror entry (c, r, w) -- found by binary search in row c.
namespace tgp {
__global__ __launch_bounds__(256) void edge_symmetry_kernel(const int64_t* __restrict__ row, const int64_t* __restrict__ col,
                                                            const float* __restrict__ w, int64_t E,
                                                            const int* __restrict__ row_ptr, int64_t N,
                                                            unsigned int* __restrict__ ticket,
                                                            unsigned int* __restrict__ bad,
                                                            unsigned long long* __restrict__ result,
                                                            unsigned long long tag) {
  __shared__ bool s_last;
  unsigned int flags = 0;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; e < E; e += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t r = row[e], c = col[e];
    if (r == c) continue;
    if (c < 0 || c >= N) { flags = 1u; continue; }
    int lo = row_ptr[c], hi = row_ptr[c + 1];
    while (lo < hi) {  // first entry of row c whose column is >= r
      const int mid = (lo + hi) >> 1;
      if (col[mid] < r) lo = mid + 1; else hi = mid;
    }
    if (lo >= row_ptr[c + 1] || col[lo] != r || (w && w[lo] != w[e])) flags = 1u;
  }
  if (flags) atomicOr(bad, flags);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    if (s_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned int fl = __hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *ticket = 0;
      *bad = 0;
      result[1] = 0ull;
      result[2] = fl;
      result[3] = result[4] = result[5] = 0ull;
      __threadfence_system();
      __hip_atomic_store(result, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
}  // namespace tgp

extern "C" int tgp_edge_symmetry_f32(const int64_t* row, const int64_t* col, const float* w, int64_t E,
                                     const int32_t* row_ptr, int64_t N, uint32_t* ticket, uint64_t* result, uint64_t tag,
                                     void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(E > 0 && N > 0 && row && col && row_ptr && ticket && result, TGP_ERR_INVALID,
              "tgp_edge_symmetry_f32: bad argument");
  int64_t blocks = (E + 1023) / 1024;
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(edge_symmetry_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, row, col, w, E, row_ptr,
                     N, ticket, ticket + 1, reinterpret_cast<unsigned long long*>(result),
                     static_cast<unsigned long long>(tag));
  return check_launch("tgp_edge_symmetry_f32");
}

// r6: symmetry of a densified adjacency, see adj_symmetry_kernel.  `ticket` / `result` / `tag` as
// tgp_edge_facts_sorted_i64 (two zeroed uint32 words per (device, stream); 6 pinned host words, word 0 = tag stored last,
// word 2 = 1 when some entry differs from its mirror image).
extern "C" int tgp_adj_symmetry_f32(const int64_t* row, const int64_t* col, int64_t E, const int64_t* batch,
                                    const int64_t* ptr, int64_t Nmax, const float* adj, uint32_t* ticket,
                                    uint64_t* result, uint64_t tag, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(E > 0 && Nmax > 0 && row && col && batch && ptr && adj && ticket && result, TGP_ERR_INVALID,
              "tgp_adj_symmetry_f32: bad argument");
  int64_t blocks = (E + 1023) / 1024;
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(adj_symmetry_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, row, col, E, batch, ptr,
                     Nmax, adj, ticket, ticket + 1, reinterpret_cast<unsigned long long*>(result),
                     static_cast<unsigned long long>(tag));
  return check_launch("tgp_adj_symmetry_f32");
}
