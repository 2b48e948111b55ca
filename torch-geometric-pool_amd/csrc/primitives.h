// Device-wide primitives shared by the sparse kernels: block scans, order-preserving
// compaction ranks, and a stable LSD radix sort (8-bit digits, wave match-any ranking).
// Everything is written for 64-wide wavefronts and 256-thread workgroups.
#pragma once
#include "common.h"

namespace tgp {

// ------------------------------------------------------------------ wave / block scans
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    uint32_t t = __shfl_up(v, d, WAVE);
    if (lane_id() >= d) v += t;
  }
  return v;
}

// Exclusive scan over the 256 threads of a workgroup.  s_w: 4 words of LDS.
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t* s_w, uint32_t* total) {
  const uint32_t inc = wave_incl_scan(v);
  if (lane_id() == WAVE - 1) s_w[wave_id()] = inc;
  __syncthreads();
  uint32_t off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const uint32_t c = s_w[w];
    if (w < wave_id()) off += c;
    tot += c;
  }
  if (total) *total = tot;
  __syncthreads();
  return off + inc - v;
}

// Order-preserving compaction ranks for a 256-thread workgroup that owns ITEMS*256
// consecutive elements laid out item-major (element = item*256 + tid).  s_cnt: ITEMS*4 words.
template <int ITEMS>
__device__ __forceinline__ void block_compact_ranks(const bool (&flag)[ITEMS], uint32_t (&rank)[ITEMS],
                                                    uint32_t& block_total, uint32_t* s_cnt) {
  const int w = wave_id();
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const unsigned long long m = __ballot(flag[it]);
    rank[it] = __popcll(m & lanemask_lt());
    if (lane_id() == 0) s_cnt[it * 4 + w] = __popcll(m);
  }
  __syncthreads();
  uint32_t run = 0;
#pragma unroll
  for (int i = 0; i < ITEMS * 4; ++i) {
    const uint32_t c = s_cnt[i];
    if ((i & 3) == w) rank[i >> 2] += run;
    run += c;
  }
  block_total = run;
  __syncthreads();
}

// counts[nb] -> exclusive offsets[nb], *total (int64).  One 1024-thread workgroup.  `refuse` (optional): a device
// flag; when set, *total becomes `refuse_code` (-2: the count -> fill protocol's "bad input" code; KronConnect passes its
// status word and -1, "declined") instead of the sum.
static __global__ __launch_bounds__(1024) void scan_counts_kernel(const uint32_t* __restrict__ counts, int nb,
                                                           uint32_t* __restrict__ offsets,
                                                           int64_t* __restrict__ total,
                                                           const int* __restrict__ refuse,
                                                           long long refuse_code = -2) {
  __shared__ uint32_t s_w[16];
  __shared__ uint32_t s_carry;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += 1024) {
    const int i = base + tid;
    const uint32_t v = i < nb ? counts[i] : 0u;
    const uint32_t inc = wave_incl_scan(v);
    if (lane == WAVE - 1) s_w[w] = inc;
    __syncthreads();
    uint32_t off = s_carry, tot = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint32_t c = s_w[j];
      if (j < w) off += c;
      tot += c;
    }
    if (i < nb) offsets[i] = off + inc - v;
    __syncthreads();
    if (tid == 0) s_carry += tot;
    __syncthreads();
  }
  if (tid == 0) *total = (refuse && *refuse) ? static_cast<int64_t>(refuse_code) : static_cast<int64_t>(s_carry);
}

constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = 256 * SCAN_ITEMS;
// total survivors -> d_count, or -1 when a precondition failed
static __global__ void cr_finish_count_kernel(const int* __restrict__ bad, const int64_t* __restrict__ total,
                                       int64_t* __restrict__ d_count) {
  const int b = *bad;  // 8 alone: the row-local coalesce met a supernode row beyond its LDS sort and was not asked to
  *d_count = b == 0 ? *total : (b == 8 ? -5 : -1);  // handle such rows (-5: call again with TGP_HUGE_ROWS)
}

// ------------------------------------------------------------------ multi-block exclusive scan (u32)
static __global__ __launch_bounds__(256) void scan_tile_sums_kernel(const uint32_t* __restrict__ in, int64_t n,
                                                             uint32_t* __restrict__ tile_sums) {
  __shared__ uint32_t s_w[4];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * SCAN_TILE + static_cast<int64_t>(threadIdx.x) * SCAN_ITEMS;
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i)
    if (base + i < n) s += in[base + i];
  uint32_t total;
  block_excl_scan_256(s, s_w, &total);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// Second (and last) pass: every workgroup adds up the sums of the tiles before it (<= SCAN_SELF_TILES of them, a
// few hundred words) instead of waiting for a separate scan-of-sums launch; the last one also publishes the
// total and, if asked, the caller-visible count (-1 when a precondition flag is set).
constexpr int SCAN_SELF_TILES = 4096;
static __global__ __launch_bounds__(256) void scan_apply_kernel(const uint32_t* __restrict__ in, int64_t n,
                                                         const uint32_t* __restrict__ tile_sums, int self_offsets,
                                                         uint32_t* __restrict__ out, int64_t* __restrict__ total,
                                                         const int* __restrict__ bad, int64_t* __restrict__ d_count) {
  __shared__ uint32_t s_w[4];
  __shared__ uint32_t s_off;
  uint32_t tile_off;
  if (self_offsets) {
    uint32_t acc = 0;
    for (int i = threadIdx.x; i < static_cast<int>(blockIdx.x); i += 256) acc += tile_sums[i];
    uint32_t tot;
    block_excl_scan_256(acc, s_w, &tot);
    if (threadIdx.x == 0) s_off = tot;
    __syncthreads();
    tile_off = s_off;
    __syncthreads();
  } else {
    tile_off = tile_sums[blockIdx.x];  // already exclusive offsets
  }
  const int64_t base = static_cast<int64_t>(blockIdx.x) * SCAN_TILE + static_cast<int64_t>(threadIdx.x) * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS], sacc = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    v[i] = base + i < n ? in[base + i] : 0u;
    sacc += v[i];
  }
  uint32_t tile_total;
  uint32_t run = tile_off + block_excl_scan_256(sacc, s_w, &tile_total);
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    if (base + i < n) out[base + i] = run;
    run += v[i];
  }
  if (self_offsets && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    const int64_t t = static_cast<int64_t>(tile_off) + tile_total;
    if (total) *total = t;
    if (d_count) *d_count = (bad && *bad) ? (*bad == 8 ? -5 : -1) : t;
  }
}

// out[0..n) = exclusive prefix of in, *total = sum.  tile scratch: 2 * ceil(n / SCAN_TILE) words.
// d_count (optional): also receives the total, or -1 if *bad is set.
static inline void device_scan_u32(const uint32_t* in, int64_t n, uint32_t* out, int64_t* total, uint32_t* tile_scratch,
                            hipStream_t stream, const int* bad = nullptr, int64_t* d_count = nullptr) {
  const int nt = cdiv(n > 0 ? n : 1, SCAN_TILE);
  uint32_t* sums = tile_scratch;
  uint32_t* offs = tile_scratch + nt;
  hipLaunchKernelGGL(scan_tile_sums_kernel, dim3(nt), dim3(256), 0, stream, in, n, sums);
  if (nt <= SCAN_SELF_TILES) {
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nt), dim3(256), 0, stream, in, n, sums, 1, out, total, bad, d_count);
    return;
  }
  hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, stream, sums, nt, offs, total, static_cast<const int*>(nullptr));
  hipLaunchKernelGGL(scan_apply_kernel, dim3(nt), dim3(256), 0, stream, in, n, offs, 0, out, nullptr, nullptr, nullptr);
  if (d_count) hipLaunchKernelGGL(cr_finish_count_kernel, dim3(1), dim3(1), 0, stream, bad, total, d_count);
}


// Workgroup-wide copy of `nwords` 32-bit words from global memory into LDS (both 16-byte aligned): 16-byte
// loads, eight in flight per thread.  A one-word-per-iteration loop exposes the full load latency on every
// iteration (~1 us each: 30 us for the 128 KB node bitmap of a 1 M-node graph, paid by every workgroup).
template <int THREADS>
__device__ __forceinline__ void lds_copy_words(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src,
                                               int nwords) {
  const int n4 = nwords >> 2;
  const uint4* src4 = reinterpret_cast<const uint4*>(src);
  uint4* dst4 = reinterpret_cast<uint4*>(dst);
  for (int base = 0; base < n4; base += 8 * THREADS) {
    uint4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = base + u * THREADS + static_cast<int>(threadIdx.x);
      v[u] = t < n4 ? src4[t] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = base + u * THREADS + static_cast<int>(threadIdx.x);
      if (t < n4) dst4[t] = v[u];
    }
  }
  for (int t = (n4 << 2) + static_cast<int>(threadIdx.x); t < nwords; t += THREADS) dst[t] = src[t];
}

// ------------------------------------------------------------------ LSD radix sort
// Pass structure (per DB-bit digit):  histogram per workgroup chunk -> per-digit scan over
// chunks -> stable scatter.  A workgroup owns one contiguous chunk of the input, so block
// order == input order and the sort is stable.  DB = 8 (small inputs) or 11 (large inputs:
// a 40-bit (row, col) key takes 4 passes instead of 5; 2048 bins still fit LDS comfortably).
constexpr int kSortThreads = 256;
constexpr int kSortItems = 8;
constexpr int kSortTile = kSortThreads * kSortItems;
constexpr int kSortMaxBlocks = 1024;
constexpr int kSortMaxBins = 2048;

template <typename KeyT, int DB>
__global__ __launch_bounds__(kSortThreads) void radix_hist_kernel(const KeyT* __restrict__ keys, int64_t n,
                                                                   int64_t chunk, int shift, int nblocks,
                                                                   uint32_t* __restrict__ hist,
                                                                   const uint32_t* __restrict__ n_dev = nullptr) {
  constexpr int BINS = 1 << DB;
  __shared__ uint32_t s_h[BINS];
  const int tid = threadIdx.x;
  if (n_dev) {  // element count known on the device only (<= the n the launch was sized for): the chunks are re-cut so
    n = *n_dev < n ? *n_dev : n;  // that every workgroup of the launch gets a share
    const int64_t tiles = (n + kSortTile - 1) / kSortTile;
    chunk = ((tiles + nblocks - 1) / nblocks) * kSortTile;
  }
  for (int d = tid; d < BINS; d += kSortThreads) s_h[d] = 0;
  __syncthreads();
  const int64_t begin = static_cast<int64_t>(blockIdx.x) * chunk;
  const int64_t end = begin + chunk < n ? begin + chunk : n;
  for (int64_t i = begin + tid; i < end; i += kSortThreads) {
    const uint32_t d = static_cast<uint32_t>(keys[i] >> shift) & (BINS - 1);
    atomicAdd(&s_h[d], 1u);
  }
  __syncthreads();
  for (int d = tid; d < BINS; d += kSortThreads) hist[static_cast<size_t>(d) * nblocks + blockIdx.x] = s_h[d];
}

// grid = BINS (one workgroup per digit); exclusive scan over the nblocks (<=1024) chunk counts.
static __global__ __launch_bounds__(1024) void radix_scan_kernel(uint32_t* __restrict__ hist, int nblocks,
                                                          uint32_t* __restrict__ digit_total) {
  __shared__ uint32_t s_w[16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  uint32_t* row = hist + static_cast<size_t>(blockIdx.x) * nblocks;
  const uint32_t v = tid < nblocks ? row[tid] : 0u;
  const uint32_t inc = wave_incl_scan(v);
  if (lane == WAVE - 1) s_w[w] = inc;
  __syncthreads();
  uint32_t off = 0, tot = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const uint32_t c = s_w[j];
    if (j < w) off += c;
    tot += c;
  }
  if (tid < nblocks) row[tid] = off + inc - v;
  if (tid == 0) digit_total[blockIdx.x] = tot;
}

template <typename KeyT, typename ValT, int DB>
__global__ __launch_bounds__(kSortThreads) void radix_scatter_kernel(
    const KeyT* __restrict__ keys_in, const ValT* __restrict__ vals_in, KeyT* __restrict__ keys_out,
    ValT* __restrict__ vals_out, const uint32_t* __restrict__ hist_scanned,
    const uint32_t* __restrict__ digit_total, int64_t n, int64_t chunk, int shift, int nblocks,
    const uint32_t* __restrict__ n_dev = nullptr) {
  constexpr int BINS = 1 << DB;
  constexpr int PER = BINS / kSortThreads;
  if (n_dev) {
    n = *n_dev < n ? *n_dev : n;
    const int64_t tiles = (n + kSortTile - 1) / kSortTile;
    chunk = ((tiles + nblocks - 1) / nblocks) * kSortTile;
  }
  __shared__ uint32_t s_base[BINS];
  __shared__ uint32_t s_whist[4][BINS];
  __shared__ uint32_t s_scan[4];
  __shared__ uint32_t s_toff[BINS];          // first tile-local slot of every digit
  __shared__ KeyT s_key[kSortTile];          // the tile, reordered by digit, before it goes out
  __shared__ ValT s_val[kSortTile];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;

  {  // first output slot of every digit = exclusive prefix of the digit totals (every workgroup redoes this tiny
     // scan instead of a launch of its own) + what earlier chunks hold of that digit
    uint32_t v[PER], sum = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      v[q] = digit_total[tid * PER + q];
      sum += v[q];
    }
    uint32_t run = block_excl_scan_256(sum, s_scan, nullptr);
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int d = tid * PER + q;
      s_base[d] = run + hist_scanned[static_cast<size_t>(d) * nblocks + blockIdx.x];
      run += v[q];
    }
  }
  __syncthreads();

  const int64_t begin = static_cast<int64_t>(blockIdx.x) * chunk;
  const int64_t end = begin + chunk < n ? begin + chunk : n;
  for (int64_t tile = begin; tile < end; tile += kSortTile) {
    for (int d = tid; d < BINS; d += kSortThreads) {
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) s_whist[ww][d] = 0;
    }
    __syncthreads();

    KeyT k[kSortItems];
    ValT v[kSortItems];
    uint32_t r[kSortItems];
    uint32_t dg[kSortItems];
#pragma unroll
    for (int it = 0; it < kSortItems; ++it) {
      const int64_t idx = tile + static_cast<int64_t>(w) * (WAVE * kSortItems) + it * WAVE + lane;
      const bool valid = idx < end;
      k[it] = valid ? keys_in[idx] : KeyT(0);
      v[it] = valid ? vals_in[idx] : ValT(0);
      const uint32_t d = static_cast<uint32_t>(k[it] >> shift) & (BINS - 1);
      dg[it] = valid ? d : BINS;  // BINS = not participating
      unsigned long long peers = __ballot(valid);
#pragma unroll
      for (int bit = 0; bit < DB; ++bit) {
        const bool set = (d >> bit) & 1u;
        const unsigned long long m = __ballot(set);
        peers &= set ? m : ~m;
      }
      const uint32_t cnt = __popcll(peers);
      const uint32_t lrank = __popcll(peers & lanemask_lt());
      uint32_t prev = 0;
      if (valid) prev = s_whist[w][d];
      __builtin_amdgcn_wave_barrier();
      if (valid && lrank == 0) s_whist[w][d] = prev + cnt;
      __builtin_amdgcn_wave_barrier();
      r[it] = prev + lrank;
    }
    __syncthreads();
    // per digit: exclusive prefix over the 4 waves (tile-local offsets) and the tile total
    uint32_t tot[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int d = tid + q * kSortThreads;
      uint32_t run = 0;
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) {
        const uint32_t c = s_whist[ww][d];
        s_whist[ww][d] = run;
        run += c;
      }
      tot[q] = run;
    }
    {  // tile-local exclusive offsets of the digits
      uint32_t sum = 0;
#pragma unroll
      for (int q = 0; q < PER; ++q) sum += tot[q];
      // digits are dealt out as d = tid + q * 256, so a plain scan over threads is only digit order for PER == 1
      static_assert(PER == 1, "LDS-staged scatter assumes 8-bit digits (one digit per thread)");
      s_toff[tid] = block_excl_scan_256(sum, s_scan, nullptr);
    }
    __syncthreads();
    // stage the tile in LDS in digit order (stable: wave order, then rank inside the wave) ...
#pragma unroll
    for (int it = 0; it < kSortItems; ++it) {
      if (dg[it] < BINS) {
        const uint32_t lpos = s_toff[dg[it]] + s_whist[w][dg[it]] + r[it];
        s_key[lpos] = k[it];
        s_val[lpos] = v[it];
      }
    }
    __syncthreads();
    // ... and write it out so that consecutive threads write consecutive addresses inside every digit's run
    const int ntile = static_cast<int>(end - tile < kSortTile ? end - tile : kSortTile);
#pragma unroll
    for (int it = 0; it < kSortItems; ++it) {
      const int i = it * kSortThreads + tid;
      if (i < ntile) {
        const KeyT kk = s_key[i];
        const uint32_t d = static_cast<uint32_t>(kk >> shift) & (BINS - 1);
        const uint32_t pos = s_base[d] + (static_cast<uint32_t>(i) - s_toff[d]);
        keys_out[pos] = kk;
        vals_out[pos] = s_val[i];
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PER; ++q) s_base[tid + q * kSortThreads] += tot[q];
  }
}

struct SortPlan {
  int nblocks;
  int64_t chunk;
};

inline SortPlan sort_plan(int64_t n) {
  SortPlan p;
  int64_t tiles = (n + kSortTile - 1) / kSortTile;
  if (tiles < 1) tiles = 1;
  p.nblocks = static_cast<int>(tiles < kSortMaxBlocks ? tiles : kSortMaxBlocks);
  const int64_t tiles_per_block = (tiles + p.nblocks - 1) / p.nblocks;
  p.chunk = tiles_per_block * kSortTile;
  p.nblocks = static_cast<int>((n + p.chunk - 1) / p.chunk);
  if (p.nblocks < 1) p.nblocks = 1;
  return p;
}

// scratch words needed besides the two ping-pong (key,val) buffers
inline size_t sort_scratch_words() { return static_cast<size_t>(kSortMaxBins) * kSortMaxBlocks + kSortMaxBins; }

// 11-bit digits (2048 bins) were measured SLOWER on MI355X (207 us vs 117 us per pass at 10M keys:
// the per-tile LDS histogram reset and the 11 ballots per key outweigh the saved pass), so 8 bits it is.
inline int sort_digit_bits(int64_t /*n*/) { return 8; }
inline int sort_passes(int64_t n, int key_bits) {
  if (n <= 1 || key_bits <= 0) return 0;
  const int db = sort_digit_bits(n);
  return (key_bits + db - 1) / db;
}

template <typename KeyT, typename ValT, int DB>
static void radix_pass(const KeyT* ki, const ValT* vi, KeyT* ko, ValT* vo, int64_t n, int shift, const SortPlan& p,
                       uint32_t* hist, uint32_t* digit_total, hipStream_t stream, const uint32_t* n_dev = nullptr) {
  constexpr int BINS = 1 << DB;
  hipLaunchKernelGGL((radix_hist_kernel<KeyT, DB>), dim3(p.nblocks), dim3(kSortThreads), 0, stream, ki, n,
                     p.chunk, shift, p.nblocks, hist, n_dev);
  hipLaunchKernelGGL(radix_scan_kernel, dim3(BINS), dim3(1024), 0, stream, hist, p.nblocks, digit_total);
  hipLaunchKernelGGL((radix_scatter_kernel<KeyT, ValT, DB>), dim3(p.nblocks), dim3(kSortThreads), 0, stream,
                     ki, vi, ko, vo, hist, digit_total, n, p.chunk, shift, p.nblocks, n_dev);
}

// Sorts (keys, vals) by the low `key_bits` bits of the key.  Buffers ping-pong between
// (k0,v0) and (k1,v1); returns in *result_in_first whether the sorted data ended in (k0,v0).
// `n_dev` (optional): the real element count lives on the device (*n_dev <= n); the launches are sized for n, chunks
// past *n_dev are empty.
template <typename KeyT, typename ValT>
int radix_sort_pairs(KeyT* k0, ValT* v0, KeyT* k1, ValT* v1, int64_t n, int key_bits, uint32_t* scratch,
                     hipStream_t stream, bool* result_in_first, const uint32_t* n_dev = nullptr) {
  *result_in_first = true;
  const int passes = sort_passes(n, key_bits);
  if (passes == 0) return TGP_OK;
  const SortPlan p = sort_plan(n);
  const int db = sort_digit_bits(n);
  uint32_t* hist = scratch;
  uint32_t* digit_total = scratch + static_cast<size_t>(kSortMaxBins) * kSortMaxBlocks;
  KeyT *ki = k0, *ko = k1;
  ValT *vi = v0, *vo = v1;
  for (int pass = 0; pass < passes; ++pass) {
    (void)db;  // 8-bit digits only (11-bit digits measured slower, see sort_digit_bits)
    radix_pass<KeyT, ValT, 8>(ki, vi, ko, vo, n, pass * 8, p, hist, digit_total, stream, n_dev);
    KeyT* tk = ki; ki = ko; ko = tk;
    ValT* tv = vi; vi = vo; vo = tv;
  }
  *result_in_first = (ki == k0);
  return check_launch("radix_sort_pairs");
}

inline int bits_for(uint64_t max_value) {
  int b = 0;
  while (max_value) { ++b; max_value >>= 1; }
  return b < 1 ? 1 : b;
}

}  // namespace tgp
