// A4 + A6 fast path: coalesce Connect WITHOUT a global sort, for row-sorted edge lists.
//
// PyG edge lists are row-sorted almost everywhere (coalesced inputs, to_undirected, dataset loaders).
// Then the edges of one supernode r are the union of the contiguous edge ranges of its member nodes, in the
// stable order (members ascending, input order inside a member) at slot
//     seg_dst[p] + (e - node_ptr[node])     (p = position of the member in the inverted index),
// so grouping the relabelled edges by supernode row needs no sort, and only the short per-row segments
// (~E/K entries) have to be ordered by column.  Pipeline:
//   K1 CSR of the input (node_ptr) + sortedness check, one pass     read E*8 B once
//   K2 member degrees: tile sums (+ the int32 cluster table)       N-sized
//   K3 scan -> seg_src / seg_dst per member; raw_off per row      N-, K-sized
//   K4 gather-sort-merge (cr_gather_sort_kernel): one workgroup per 32 supernode rows gathers the rows' edges
//      through the cluster table straight into LDS, sorts every row there (8 lanes x 4 keys per row of <= 32
//      entries, 16 x 4 for 33..64: cr_sort_rows; workgroup LDS bitonic for 65..1024), merges duplicates with
//      reduce_op in input order, fused self-loop / eps filters, survivors compacted per row
//                                                                  read E*12 B + gathers, write E'*8 B
//   K5 scan survivors -> output offsets, total                    K-sized            [host reads total]
//   K6 fill (output-parallel per 64-row block)                     read E'*8 B, write E'*20 B
// vs. five radix passes of 32 B/edge each in the general path (sparse_connect.hip).  The result is
// identical to the general path (row-major sorted, unique, duplicates reduced in input order).
// Preconditions are checked on the device (rows sorted; no supernode row longer than 1024 raw entries); if they fail *d_count is set to -1 and the caller falls back to the sort-based path.
#include "lookback.h"

namespace tgp {

constexpr int CR_LONG = 1024;    // longest supernode row (raw entries) the LDS kernel sorts
// ------------------------------------------------------------------ K1 / K2
// CSR offsets of the (row-sorted) input AND the sortedness check in one pass over the row array: thread p owns
// edge p and fills node_ptr for the nodes that start between its predecessor's row and its own.  An inversion (or a
// node id outside [0, num_rows)) sets *bad = 1; fill loops are only entered on an ascending step, re-read the flag
// now and then, and are bounded by num_rows, so unsorted input costs a few microseconds, not an unbounded walk.
__global__ __launch_bounds__(256) void cr_node_ptr_kernel(const int64_t* __restrict__ rows, int64_t n,
                                                          int64_t num_rows, int* __restrict__ bad,
                                                          uint32_t* __restrict__ node_ptr) {
  const int64_t p = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (p > n) return;
  volatile int* vbad = bad;
  if (p == n) {
    int64_t first = n > 0 ? rows[n - 1] + 1 : 0;
    if (first < 0) first = 0;
    for (int64_t c = first; c <= num_rows; ++c) node_ptr[c] = static_cast<uint32_t>(n);
    return;
  }
  const int64_t cur = rows[p], prev = p > 0 ? rows[p - 1] : -1;
  if (cur < prev || cur < 0 || cur >= num_rows) {
    if (*vbad == 0) atomicOr(bad, 1);  // the atomic drops the line from this XCD's L2: later plain reads see it
    return;
  }
  for (int64_t c = prev + 1; c <= cur; ++c) {
    node_ptr[c] = static_cast<uint32_t>(p);
    if (((c - prev) & 63) == 0 && *vbad) return;
  }
}

// The same pass with four consecutive edges per thread (two 16-byte loads): a quarter of the waves to launch for a
// kernel that only streams 8 bytes per edge.  `rows` must be 16-byte aligned (the host picks the variant).
__global__ __launch_bounds__(256) void cr_node_ptr_vec_kernel(const int64_t* __restrict__ rows, int64_t n,
                                                              int64_t num_rows, int* __restrict__ bad,
                                                              uint32_t* __restrict__ node_ptr) {
  const int64_t p0 = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) * 4;
  volatile int* vbad = bad;
  if (p0 == 0) {  // the tail [last row + 1, num_rows] (one thread, it also covers n == 0)
    int64_t first = n > 0 ? rows[n - 1] + 1 : 0;
    if (first < 0) first = 0;
    for (int64_t c = first; c <= num_rows; ++c) node_ptr[c] = static_cast<uint32_t>(n);
  }
  if (p0 >= n) return;
  int64_t v[4];
  if (p0 + 4 <= n) {
    const longlong2 a = *reinterpret_cast<const longlong2*>(rows + p0);
    const longlong2 b = *reinterpret_cast<const longlong2*>(rows + p0 + 2);
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = p0 + i < n ? rows[p0 + i] : num_rows - 1;  // padded with a valid, final id
  }
  int64_t prev = p0 > 0 ? rows[p0 - 1] : -1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (p0 + i >= n) break;
    const int64_t cur = v[i];
    if (cur < prev || cur < 0 || cur >= num_rows) {
      if (*vbad == 0) atomicOr(bad, 1);
      return;
    }
    for (int64_t c = prev + 1; c <= cur; ++c) {
      node_ptr[c] = static_cast<uint32_t>(p0 + i);
      if (((c - prev) & 63) == 0 && *vbad) return;
    }
    prev = cur;
  }
}

// ------------------------------------------------------------------ K2 / K3: member segments
// Every member p of the inverted index (members of supernode 0, then of supernode 1, ...) contributes one contiguous
// range of the row-sorted input: seg_src[p] = its first edge, and its first slot in the supernode-row-major raw
// layout is the exclusive prefix of the member degrees, seg_dst[p] (seg_dst[nnz] = E).  Two kernels: degree sums per
// tile of MS_TILE members (+ the int32 cluster table, an independent N-sized copy that rides along), then the scan.
constexpr int MS_ITEMS = 8;
constexpr int MS_TILE = 256 * MS_ITEMS;
__global__ __launch_bounds__(256) void cr_member_sums_kernel(const int32_t* __restrict__ a_perm, int64_t nnz,
                                                             const uint32_t* __restrict__ node_ptr,
                                                             const int64_t* __restrict__ cluster, int64_t n_nodes,
                                                             const int* __restrict__ bad, int32_t* __restrict__ table,
                                                             uint32_t* __restrict__ tile_sums) {
  __shared__ uint32_t s_w[4];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * MS_TILE + static_cast<int64_t>(threadIdx.x) * MS_ITEMS;
#pragma unroll
  for (int i = 0; i < MS_ITEMS; ++i)
    if (base + i < n_nodes) table[base + i] = static_cast<int32_t>(cluster[base + i]);
  if (static_cast<int64_t>(blockIdx.x) * MS_TILE >= nnz) return;
  uint32_t sum = 0;
  if (*bad == 0) {  // rows not sorted: node_ptr is not a CSR, the call declines
    int32_t node[MS_ITEMS];
#pragma unroll
    for (int i = 0; i < MS_ITEMS; ++i) node[i] = base + i < nnz ? a_perm[base + i] : -1;
#pragma unroll
    for (int i = 0; i < MS_ITEMS; ++i)
      if (node[i] >= 0) sum += node_ptr[node[i] + 1] - node_ptr[node[i]];
  }
  uint32_t total;
  block_excl_scan_256(sum, s_w, &total);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// self_offsets: tile_sums are raw sums (every workgroup adds up the ones before it); else exclusive offsets
__global__ __launch_bounds__(256) void cr_member_scan_kernel(const int32_t* __restrict__ a_perm, int64_t nnz,
                                                             const uint32_t* __restrict__ node_ptr,
                                                             const uint32_t* __restrict__ tile_sums, int self_offsets,
                                                             const int* __restrict__ bad,
                                                             uint32_t* __restrict__ seg_src,
                                                             uint32_t* __restrict__ seg_dst) {
  __shared__ uint32_t s_w[4];
  __shared__ uint32_t s_off;
  if (*bad) return;
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
    tile_off = tile_sums[blockIdx.x];
  }
  const int64_t base = static_cast<int64_t>(blockIdx.x) * MS_TILE + static_cast<int64_t>(threadIdx.x) * MS_ITEMS;
  int32_t node[MS_ITEMS];
  uint32_t first[MS_ITEMS], len[MS_ITEMS], sum = 0;
#pragma unroll
  for (int i = 0; i < MS_ITEMS; ++i) node[i] = base + i < nnz ? a_perm[base + i] : -1;
#pragma unroll
  for (int i = 0; i < MS_ITEMS; ++i) {
    first[i] = node[i] >= 0 ? node_ptr[node[i]] : 0u;
    len[i] = node[i] >= 0 ? node_ptr[node[i] + 1] - first[i] : 0u;
    sum += len[i];
  }
  uint32_t tile_total;
  uint32_t run = tile_off + block_excl_scan_256(sum, s_w, &tile_total);
#pragma unroll
  for (int i = 0; i < MS_ITEMS; ++i) {
    if (base + i < nnz) {
      seg_src[base + i] = first[i];
      seg_dst[base + i] = run;
    }
    run += len[i];
    if (base + i == nnz - 1) seg_dst[nnz] = run;
  }
}

// r4: the two kernels above as ONE launch (the published count route, which has an epoch-tagged status buffer): a tile's
// degree sum is published, the sums in front of it come from the decoupled look-back of lookback.h (a few hundred tiles,
// all resident), and the tile writes its members' segments at once -- the member ids and their CSR offsets are loaded
// once instead of twice.  csr_ptr != NULL: the handed-over CSR is checked here too (offsets start at 0 and end at E: what
// cr_check_csr_kernel did in a launch of its own) and workgroup 0 resets the status words for the kernels behind.
__global__ __launch_bounds__(256) void cr_member_single_kernel(const int32_t* __restrict__ a_perm, int64_t nnz,
                                                               const uint32_t* __restrict__ node_ptr,
                                                               const int64_t* __restrict__ cluster, int64_t n_nodes,
                                                               int64_t E, int csr_given, int* __restrict__ bad,
                                                               int32_t* __restrict__ table,
                                                               uint32_t* __restrict__ seg_src,
                                                               uint32_t* __restrict__ seg_dst,
                                                               unsigned long long* status, unsigned long long tag,
                                                               int ticket) {
  __shared__ uint32_t s_w[4];
  __shared__ uint32_t s_off;
  __shared__ int s_refused;
  __shared__ int s_tile;
  const int tile = sps_tile_id(status + 1, tag, ticket, &s_tile);
  const int64_t base = static_cast<int64_t>(tile) * MS_TILE + static_cast<int64_t>(threadIdx.x) * MS_ITEMS;
#pragma unroll
  for (int i = 0; i < MS_ITEMS; ++i)
    if (base + i < n_nodes) table[base + i] = static_cast<int32_t>(cluster[base + i]);
  int declined;
  if (csr_given) {  // (every workgroup looks for itself: nobody may rely on workgroup 0's reset inside this launch)
    declined = (node_ptr[0] != 0u || static_cast<int64_t>(node_ptr[n_nodes]) != E) ? 4 : 0;
    if (tile == 0 && threadIdx.x == 0) {
      bad[0] = declined;
      bad[1] = 0;
      bad[2] = 0;
      bad[3] = 0;
    }
  } else {
    declined = *bad;  // written by the CSR pass in front (rows not sorted: node_ptr is not a CSR)
  }
  if (declined || static_cast<int64_t>(tile) * MS_TILE >= nnz) return;  // (uniform over the tiles that take part)
  int32_t node[MS_ITEMS];
  uint32_t first[MS_ITEMS], len[MS_ITEMS], sum = 0;
#pragma unroll
  for (int i = 0; i < MS_ITEMS; ++i) node[i] = base + i < nnz ? a_perm[base + i] : -1;
#pragma unroll
  for (int i = 0; i < MS_ITEMS; ++i) {
    first[i] = node[i] >= 0 ? node_ptr[node[i]] : 0u;
    len[i] = node[i] >= 0 ? node_ptr[node[i] + 1] - first[i] : 0u;
    sum += len[i];
  }
  uint32_t tile_total;
  const uint32_t local = block_excl_scan_256(sum, s_w, &tile_total);
  if (threadIdx.x < WAVE) {
    if (threadIdx.x == 0)
      sps_store(status + 2 + tile, tag | (tile == 0 ? SPS_PRE : SPS_AGG) | static_cast<unsigned long long>(tile_total));
    uint32_t excl = 0;
    bool refused = false;
    if (tile > 0) {
      sps_lookback<4>(status, tile, tag, &excl, &refused);
      if (threadIdx.x == 0)
        sps_store(status + 2 + tile, tag | SPS_PRE | (refused ? 0x80000000ull : 0ull) |
                                         static_cast<unsigned long long>((excl + tile_total) & 0x7FFFFFFFu));
    }
    if (threadIdx.x == 0) {
      s_off = excl;
      s_refused = refused ? 1 : 0;
      // a look-back that ran into its spin bound declines the call.  NOT through bad[0]: workgroup 0 resets that word
      // with a plain store in this very launch, and a tile that times out does so exactly when workgroup 0 is late --
      // the reset could land on top of the refusal.  The refusal goes to word [0] of this look-back's
      // region of the status buffer, tagged with the call's epoch (never cleared, never reset); cr_raw_off_kernel,
      // the next launch, folds it into bad[0] for everything behind it.
      if (refused) sps_store(status, tag | 1ull);
    }
  }
  __syncthreads();
  if (s_refused) return;
  uint32_t run = s_off + local;
#pragma unroll
  for (int i = 0; i < MS_ITEMS; ++i) {
    if (base + i < nnz) {
      seg_src[base + i] = first[i];
      seg_dst[base + i] = run;
    }
    run += len[i];
    if (base + i == nnz - 1) seg_dst[nnz] = run;
  }
}

// raw_off[r] = first slot of supernode row r = first slot of its first member (an empty row shares its successor's);
// a row longer than the LDS sort takes declines the call before the heavy kernel runs.
// A row beyond CR_LONG entries (a hub): listed for the huge-row kernels below when the caller asked for them
// (huge_list != NULL: TGP_HUGE_ROWS), else the call declines with status 8 (d_count = -5).
constexpr int HUGE_MAX = 4096;  // huge rows per call (beyond: decline to the radix routes)
__global__ __launch_bounds__(256) void cr_raw_off_kernel(const int32_t* __restrict__ a_row_ptr, int64_t K,
                                                         const uint32_t* __restrict__ seg_dst, int* __restrict__ bad,
                                                         uint32_t* __restrict__ raw_off,
                                                         uint32_t* __restrict__ huge_list,
                                                         uint32_t* __restrict__ n_out,
                                                         const unsigned long long* __restrict__ refusal,
                                                         unsigned long long tag) {
  const int64_t r = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (refusal && sps_load(refusal) == (tag | 1ull)) {  // cr_member_single_kernel declined (see there): uniform
    if (r == 0) atomicOr(bad, 1);
    return;
  }
  if (r >= K || (*bad & 1)) return;
  const uint32_t lo = seg_dst[a_row_ptr[r]], hi = seg_dst[a_row_ptr[r + 1]];
  raw_off[r] = lo;
  if (hi - lo > static_cast<uint32_t>(CR_LONG)) {
    if (!huge_list) {
      atomicOr(bad, 8);
    } else {
      const int idx = atomicAdd(bad + 2, 1);
      if (idx < HUGE_MAX) huge_list[idx] = static_cast<uint32_t>(r);
      else atomicOr(bad, 2);
      n_out[r] = 0;
    }
  }
}

// ------------------------------------------------------------------ K5
// value of lane (lane ^ J) within the 32-lane half: one ds_swizzle (bit-mask mode: and 0x1F, xor J) - no address
// arithmetic and no LDS access, where __shfl_xor costs two VALU instructions plus a ds_bpermute
template <int J>
__device__ __forceinline__ uint32_t xor_lane(uint32_t v) {
  return static_cast<uint32_t>(__builtin_amdgcn_ds_swizzle(static_cast<int>(v), 0x1F | (J << 10)));
}

__device__ __forceinline__ float cr_reduce(float acc, float v, int op) {
  switch (op) {
    case TGP_MIN: return fminf(acc, v);
    case TGP_MAX: return fmaxf(acc, v);
    case TGP_MUL: return __fmul_rn(acc, v);
    default: return __fadd_rn(acc, v);
  }
}
// float64 edge weights (r5): the row-local routes merge in double, as the reference's coalesce does for a double tensor
__device__ __forceinline__ double cr_reduce(double acc, double v, int op) {
  switch (op) {
    case TGP_MIN: return fmin(acc, v);
    case TGP_MAX: return fmax(acc, v);
    case TGP_MUL: return __dmul_rn(acc, v);
    default: return __dadd_rn(acc, v);
  }
}
__device__ __forceinline__ float cr_abs(float v) { return fabsf(v); }
__device__ __forceinline__ double cr_abs(double v) { return fabs(v); }

// ------------------------------------------------------------------ K4 + K5 fused: gather, sort, merge
// One workgroup owns GS_ROWS consecutive supernode rows.  Their members are consecutive in the inverted index
// and every member's edges are one contiguous range of the (row-sorted) input, so the workgroup
//   (a) reads its members' (first edge, first slot) segments, 8 lanes per member,
//   (b) gathers member-parallel: the 8 lanes walk the member's edge range, map the columns through the cluster
//       table and drop (cluster, w) into the member's LDS slots - no scatter through HBM,
//   (c) sorts every row inside LDS / registers (8 or 16 lanes x 4 keys for <= 64 entries; longer rows go out raw
//       for the workgroup bitonic of cr_rows_long_kernel), merges duplicates in input order with the A6 filters
//       fused, and writes the survivors of row r compacted at tmp[raw_off[r] ...] with their count in n_out[r].
// Every dependent-load level (row offsets -> segments -> edges -> table) is one block-wide request with all of a
// thread's loads in flight together, so the latency chain is paid once per ~1000 edges instead of once per row.
#ifdef TGP_GEMM_STAMPS  // diagnostic build only (make stamps): time per phase of cr_gather_sort_kernel, per workgroup
__device__ unsigned long long* g_gs_stamps = nullptr;
__device__ int g_gs_ablate = 0;  // 1: no in-row sort, 2: no edge / table loads (synthetic keys), 3: neither
#define GS_STAMP(slot)                                                                              \
  do {                                                                                              \
    if (g_gs_stamps && threadIdx.x == 0) {                                                          \
      const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();                             \
      g_gs_stamps[static_cast<long>(blockIdx.x) * 8 + (slot)] += now_ - t_prev;                     \
      t_prev = now_;                                                                                \
    }                                                                                               \
  } while (0)
#else
#define GS_STAMP(slot) do {} while (0)
#endif
constexpr int GS_ROWS = 32;
constexpr int GS_CAP = 1024;   // raw entries staged per pass (>= CR_LONG)

// ------------------------------------------------------------------ in-row sort + merge, LPR lanes per row
// A row of up to 4 * LPR entries is sorted by LPR lanes holding 4 keys each (element e of the row lives in lane e / 4,
// register e % 4): the two lowest index bits are lane-local, so 9 of the 15 compare-exchange levels of a 32-element
// bitonic network are plain v_min / v_max on registers and only 6 cross lanes (ds_swizzle inside the 32-lane half).
// Flip formulation (first level of every merge pairs e with e ^ (k - 1), then half-cleaners e ^ j): every exchange
// is ascending, the lower element keeps the minimum - no per-level direction logic.  Only keys travel: key =
// column << PB | input position, which also makes the sort stable; weights are fetched from LDS by position after.
// A wave sorts 64 / LPR rows at once: 8 rows of <= 32 entries instead of the 2 of a lane-per-entry network.
template <int M>
__device__ __forceinline__ uint32_t cr_swz(uint32_t v) {
  return static_cast<uint32_t>(__builtin_amdgcn_ds_swizzle(static_cast<int>(v), 0x1F | (M << 10)));
}
__device__ __forceinline__ void cr_ce(uint32_t& a, uint32_t& b) {
  const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
  a = lo;
  b = hi;
}
// partner lane = lane ^ M, partner register = 3 - q (FLIP) or q; the lower lane keeps the minima
template <int M, bool FLIP>
__device__ __forceinline__ void cr_xlane(uint32_t (&k)[4], int l) {
  constexpr int HB = M & ~(M >> 1);  // highest set bit of M (M is 2^n or 2^n - 1)
  const bool lower = (l & HB) == 0;
  uint32_t o[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) o[q] = cr_swz<M>(k[FLIP ? 3 - q : q]);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t mn = k[q] < o[q] ? k[q] : o[q], mx = k[q] < o[q] ? o[q] : k[q];
    k[q] = lower ? mn : mx;
  }
}
__device__ __forceinline__ void cr_intra21(uint32_t (&k)[4]) {
  cr_ce(k[0], k[2]); cr_ce(k[1], k[3]);  // j = 2
  cr_ce(k[0], k[1]); cr_ce(k[2], k[3]);  // j = 1
}

// DPP move (row_shr:n = 0x110 + n, row_shl:n = 0x100 + n, inside a 16-lane row; lanes shifted in from outside read 0)
template <int CTRL>
__device__ __forceinline__ uint32_t cr_dpp(uint32_t v) {
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, 0xF, 0xF, true));
}

template <int LPR, typename WT>
__device__ __forceinline__ void cr_sort_rows(uint32_t* s_key, WT* s_val, uint32_t b, uint32_t T, bool mine,
                                             uint32_t row_id, bool has_w, int reduce_op, int flags, WT eps,
                                             uint32_t* __restrict__ n_out_row) {
  constexpr int PB = LPR == 8 ? 5 : 6;          // position bits: 32 or 64 entries per row
  constexpr uint32_t PM = (1u << PB) - 1u;
  const int l = threadIdx.x & (LPR - 1);
  uint32_t k[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t e = static_cast<uint32_t>(l) * 4 + q;
    k[q] = e < T ? (s_key[b + e] << PB) | e : 0xFFFFFFFFu;
  }
  cr_ce(k[0], k[1]); cr_ce(k[2], k[3]);                              // k = 2
  cr_ce(k[0], k[3]); cr_ce(k[1], k[2]); cr_ce(k[0], k[1]); cr_ce(k[2], k[3]);  // k = 4: flip, j = 1
  cr_xlane<1, true>(k, l); cr_intra21(k);                             // k = 8
  cr_xlane<3, true>(k, l); cr_xlane<1, false>(k, l); cr_intra21(k);   // k = 16
  cr_xlane<7, true>(k, l); cr_xlane<2, false>(k, l); cr_xlane<1, false>(k, l); cr_intra21(k);  // k = 32
  if constexpr (LPR == 16) {
    cr_xlane<15, true>(k, l); cr_xlane<4, false>(k, l); cr_xlane<2, false>(k, l); cr_xlane<1, false>(k, l);
    cr_intra21(k);                                                     // k = 64
  }
  // sorted order = (lane, register); invalid keys (0xFFFFFFFF) are last
  uint32_t c[4];
  bool valid[4], head[4];
  WT acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    valid[q] = k[q] != 0xFFFFFFFFu;
    c[q] = k[q] >> PB;
    acc[q] = (valid[q] && has_w) ? s_val[b + (k[q] & PM)] : WT(0);
  }
  const uint32_t pc = cr_dpp<0x111>(c[3]);  // row_shr:1; an earlier element of a valid one is valid
  head[0] = valid[0] && (l == 0 || pc != c[0]);
#pragma unroll
  for (int q = 1; q < 4; ++q) head[q] = valid[q] && c[q] != c[q - 1];
  // Duplicates of a column that the self-loop filter drops anyway need no folding (a Graclus row always holds its
  // own column twice: the matched pair's edge, seen from both ends).
  const bool drop_self = (flags & TGP_REMOVE_SELF_LOOPS) != 0;
  bool nonhead[4], fold_here = false;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    nonhead[q] = valid[q] && !head[q];
    fold_here = fold_here || (nonhead[q] && !(drop_self && c[q] == row_id));
  }
  uint32_t ncnt[4] = {1, 1, 1, 1};
  const bool any_fold = __any(fold_here);
  bool long_run = false;
  if (any_fold) {  // a run of three or more equal columns anywhere in the wave?
    // (DPP reads 0 from lanes that are switched off: move first, with every lane on, then mask)
    const uint32_t prev_nh = cr_dpp<0x111>(nonhead[3] ? 1u : 0u);
    const bool prev_nonhead = (l > 0) & (prev_nh != 0u);
    long_run = __any((nonhead[0] && prev_nonhead) || (nonhead[1] && nonhead[0]) || (nonhead[2] && nonhead[1]) ||
                     (nonhead[3] && nonhead[2]));
  }
  if (any_fold && !long_run) {  // runs of two: the head takes its successor's weight, in registers
    const uint32_t next_nh = cr_dpp<0x101>(nonhead[0] ? 1u : 0u);  // row_shl:1
    const bool next_nonhead = (l < LPR - 1) & (next_nh != 0u);
    WT next_acc;
    if constexpr (sizeof(WT) == 4) {
      next_acc = __uint_as_float(cr_dpp<0x101>(__float_as_uint(acc[0])));
    } else {  // a double travels as its two halves
      const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(acc[0]));
      const uint32_t lo = cr_dpp<0x101>(static_cast<uint32_t>(bits)), hi = cr_dpp<0x101>(static_cast<uint32_t>(bits >> 32));
      next_acc = __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(hi) << 32) | lo));
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      if (head[q] && nonhead[q + 1]) {
        acc[q] = cr_reduce(acc[q], acc[q + 1], reduce_op);
        ncnt[q] = 2;
      }
    }
    if (head[3] && next_nonhead) {
      acc[3] = cr_reduce(acc[3], next_acc, reduce_op);
      ncnt[3] = 2;
    }
  } else if (any_fold) {  // longer runs (rare): heads fold their run from LDS, in input order
    // the sorted keys go back to the row's own slots (every lane of the row has its keys in registers by now)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (valid[q]) s_key[b + static_cast<uint32_t>(l) * 4 + q] = k[q];
    __threadfence_block();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (head[q]) {
        for (uint32_t p = static_cast<uint32_t>(l) * 4 + q + 1; p < T; ++p) {
          const uint32_t kk = s_key[b + p];
          if ((kk >> PB) != c[q]) break;
          if (has_w) acc[q] = cr_reduce(acc[q], s_val[b + (kk & PM)], reduce_op);
          ++ncnt[q];
        }
      }
    }
  }
  uint32_t cnt_lane = 0;
  bool keep[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (has_w && reduce_op == TGP_MEAN) acc[q] = acc[q] / static_cast<WT>(ncnt[q]);
    keep[q] = head[q];
    if ((flags & TGP_REMOVE_SELF_LOOPS) && c[q] == row_id) keep[q] = false;
    if (has_w && (flags & TGP_EPS_FILTER) && !(cr_abs(acc[q]) > eps)) keep[q] = false;
    cnt_lane += keep[q] ? 1u : 0u;
  }
  uint32_t incl = cnt_lane;  // inclusive scan over the LPR lanes of the row
  {
    uint32_t t = cr_dpp<0x111>(incl);  // row_shr:d inside the 16-lane row; the l >= d guard keeps it inside the LPR group
    if (l >= 1) incl += t;
    t = cr_dpp<0x112>(incl);
    if (l >= 2) incl += t;
    t = cr_dpp<0x114>(incl);
    if (l >= 4) incl += t;
    if constexpr (LPR == 16) {
      t = cr_dpp<0x118>(incl);
      if (l >= 8) incl += t;
    }
  }
  uint32_t rank = incl - cnt_lane;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (keep[q]) {  // survivors stay in LDS, compacted at the head of the row's slots (every lane of the row holds
      s_key[b + rank] = c[q];  // its keys and weights in registers by now, and the row's lanes share a wave);
      if (has_w) s_val[b + rank] = acc[q];  // the workgroup copies its whole slot range out afterwards, coalesced
      ++rank;
    }
  }
  if (mine && l == LPR - 1) *n_out_row = incl;
}

// DIRECT = false: rows are assembled from the members' edge ranges (row-sorted input, above).
// DIRECT = true : `grouped` already holds the (cluster column | weight bits << 32) entries in supernode-row order
//                 (the output of the 3-pass radix sort by supernode row, below); slot t of a row is grouped[...+t].
// CT: int64 columns of the caller's list, or the int32 copy GraclusSelect's CSR already holds (half the column stream).
// WT: the weights' type (float, or double for float64 edge weights: r5; DIRECT is float only).
template <bool DIRECT, typename CT = int64_t, typename WT = float>
__global__ __launch_bounds__(256) void cr_gather_sort_kernel(
    const CT* __restrict__ col, const WT* __restrict__ w, int64_t E, const int32_t* __restrict__ table,
    const int32_t* __restrict__ a_row_ptr, const uint32_t* __restrict__ seg_src, const uint32_t* __restrict__ seg_dst,
    const unsigned long long* __restrict__ grouped, const uint32_t* __restrict__ raw_off, int64_t K, int reduce_op,
    int flags, WT eps,
    int* __restrict__ bad, uint32_t* __restrict__ tmp_c, WT* __restrict__ tmp_w, uint32_t* __restrict__ n_out,
    int64_t n_nodes, uint32_t* __restrict__ long_list) {
  __shared__ uint32_t s_key[GS_CAP];
  __shared__ WT s_val[GS_CAP];
  __shared__ uint32_t s_roff[GS_ROWS + 1];
  __shared__ int32_t s_rp[GS_ROWS + 1];
  __shared__ int s_mid[GS_ROWS];
  __shared__ int s_nmid;
  if (*bad) return;
#ifdef TGP_GEMM_STAMPS
  unsigned long long t_prev = __builtin_amdgcn_s_memrealtime();
  if (g_gs_stamps && threadIdx.x == 0) g_gs_stamps[static_cast<long>(blockIdx.x) * 8 + 7] = t_prev;
#endif
  const int tid = threadIdx.x;
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * GS_ROWS;
  const int nrows = static_cast<int>(K - r0 < GS_ROWS ? K - r0 : GS_ROWS);
  if (tid <= nrows) {
    s_roff[tid] = r0 + tid < K ? raw_off[r0 + tid] : static_cast<uint32_t>(E);
    s_rp[tid] = DIRECT ? 0 : a_row_ptr[r0 + tid];
  }
  __syncthreads();
  GS_STAMP(5);
  const bool has_w = tmp_w != nullptr;
  int rs = 0;
  while (rs < nrows) {
    if (s_roff[rs + 1] - s_roff[rs] > static_cast<uint32_t>(CR_LONG)) {  // a supernode row too long for the LDS sort
      if constexpr (DIRECT) {  // the grouped route has no huge-row kernels: decline to the general route
        if (tid == 0) *bad = 2;
        return;
      }
      ++rs;      // listed by cr_raw_off_kernel for the huge-row kernels (or the call has declined already: status 8);
      continue;  // its slots of tmp and its n_out entry are theirs
    }
    int re = rs + 1;
    if (s_roff[nrows] - s_roff[rs] <= static_cast<uint32_t>(GS_CAP)) {
      re = nrows;  // the usual case: all remaining rows fit one pass (no serial walk over the rows)
    } else {
      while (re < nrows && s_roff[re + 1] - s_roff[rs] <= static_cast<uint32_t>(GS_CAP)) ++re;
    }
    const uint32_t base = s_roff[rs];
    const int cnt = static_cast<int>(s_roff[re] - base);
    const int p_lo = s_rp[rs], M = s_rp[re] - p_lo;
    if (tid == 0) s_nmid = 0;
    if constexpr (DIRECT) {
      for (int t = tid; t < cnt; t += 256) {
        const unsigned long long v = grouped[base + t];
        s_key[t] = static_cast<uint32_t>(v);
        if constexpr (sizeof(WT) == 4) s_val[t] = __uint_as_float(static_cast<uint32_t>(v >> 32));
      }
    } else {
    // (a+b) member-parallel gather: 8 lanes walk one member's edge range (a contiguous run of the row-sorted
    // input), map the columns through the cluster table and drop (cluster, w) into the member's LDS slots.  Two
    // rounds of 32 members x two steps of 8 edges are requested before any is consumed: one round trip per level
    // (segments -> edges -> table) for the usual 64 members of <= 16 edges; longer members finish in the tail loop.
    {
      const int grp = tid >> 3, l = tid & 7;
      const uint32_t end_all = base + static_cast<uint32_t>(cnt);
      constexpr int GR = GS_CAP / 512;  // rounds of 32 members in flight
      for (int m0 = 0; m0 < M; m0 += 32 * GR) {
        uint32_t src[GR], dst[GR], len[GR];
#pragma unroll
        for (int r = 0; r < GR; ++r) {
          const int m = m0 + r * 32 + grp;
          const bool v = m < M;
          src[r] = v ? seg_src[p_lo + m] : 0u;
          const uint32_t d = v ? seg_dst[p_lo + m] : 0u;
          const uint32_t nx = m + 1 < M ? seg_dst[p_lo + m + 1] : end_all;  // zero-length members share a slot
          len[r] = v ? nx - d : 0u;
          dst[r] = d - base;
        }
        CT cc[GR][2];
        WT wv[GR][2];
#pragma unroll
        for (int r = 0; r < GR; ++r)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const uint32_t j = static_cast<uint32_t>(l + 8 * q);
            bool ok = j < len[r];
#ifdef TGP_GEMM_STAMPS
            if (g_gs_ablate & 2) ok = false;
#endif
#ifdef TGP_GEMM_STAMPS
            if (g_gs_ablate & 8) {  // no col / w loads: synthetic node ids, table lookups stay
              cc[r][q] = static_cast<CT>(((src[r] + j) * 2654435761u) % static_cast<uint32_t>(n_nodes));
              wv[r][q] = 1.f;
              continue;
            }
#endif
            // streamed once: non-temporal, so that the edge list does not push the cluster table out of the L2
            cc[r][q] = ok ? __builtin_nontemporal_load(col + src[r] + j) : 0;
            wv[r][q] = (ok && has_w) ? __builtin_nontemporal_load(w + src[r] + j) : WT(0);
          }
        // The table look-ups of ALL the lane's edges are requested before the first one is consumed (r5, late):
        // unconditional loads through a clamped index (entry 0 for a slot without an edge or a column out of range).
        // Written as `inr ? table[c] : 0` inside the `j < len` block below, as r2-r5 had it, every look-up compiled to
        // branch + load + `s_waitcnt vmcnt(0)` + LDS store: the 2 GR look-ups of a lane went out one at a time.
        // (Measured at C4, same box: 0.2336 ms per Connect call both ways -- with eight waves per SIMD the other
        //  waves' look-ups cover a lane's serial ones; the kernel is bound by what the misses move, not by their
        //  latency: profiles/r05_tcc_coalesce.md.  Kept: it is the form the comment above the loop describes.)
        uint32_t tv[GR][2];
#pragma unroll
        for (int r = 0; r < GR; ++r)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const uint32_t j = static_cast<uint32_t>(l + 8 * q);
            const bool inr = static_cast<uint64_t>(static_cast<int64_t>(cc[r][q])) < static_cast<uint64_t>(n_nodes);
            const int64_t idx = (j < len[r] && inr) ? static_cast<int64_t>(cc[r][q]) : 0;
            tv[r][q] = static_cast<uint32_t>(table[idx]);
          }
#pragma unroll
        for (int r = 0; r < GR; ++r)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const uint32_t j = static_cast<uint32_t>(l + 8 * q);
            if (j < len[r]) {
              // a column outside [0, n_nodes): decline (the general path reports it); never used as an index
              const bool inr = static_cast<uint64_t>(static_cast<int64_t>(cc[r][q])) < static_cast<uint64_t>(n_nodes);
              if (!inr) *bad = 4;
#ifdef TGP_GEMM_STAMPS
              if (g_gs_ablate & 2) {
                s_key[dst[r] + j] = ((src[r] + j) * 2654435761u) >> 13;
                s_val[dst[r] + j] = 1.f;
                continue;
              }
#endif
#ifdef TGP_GEMM_STAMPS
              if (g_gs_ablate & 4) {  // no table lookups
                s_key[dst[r] + j] = static_cast<uint32_t>(cc[r][q]) >> 1;
                s_val[dst[r] + j] = wv[r][q];
                continue;
              }
#endif
#ifdef TGP_GEMM_STAMPS
              if (g_gs_ablate & (256 | 512 | 1024)) {  // key = node id / 2 (as mode 4) + a look-up whose value is dropped:
                uint32_t v;                             // 4-, 2- or 1-byte table entries (footprint 4 / 2 / 1 MB)
                if (g_gs_ablate & 256) v = static_cast<uint32_t>(table[cc[r][q]]);
                else if (g_gs_ablate & 512) v = reinterpret_cast<const uint16_t*>(table)[cc[r][q]];
                else v = reinterpret_cast<const uint8_t*>(table)[cc[r][q]];
                asm volatile("" ::"v"(v));
                s_key[dst[r] + j] = static_cast<uint32_t>(cc[r][q]) >> 1;
                s_val[dst[r] + j] = wv[r][q];
                continue;
              }
              if (g_gs_ablate & 16) {  // nontemporal table lookups
                s_key[dst[r] + j] = inr ? static_cast<uint32_t>(__builtin_nontemporal_load(table + cc[r][q])) : 0u;
                s_val[dst[r] + j] = wv[r][q];
                continue;
              }
#endif
              s_key[dst[r] + j] = inr ? tv[r][q] : 0u;
              s_val[dst[r] + j] = wv[r][q];
            }
          }
#pragma unroll
        for (int r = 0; r < GR; ++r)
          for (uint32_t j = static_cast<uint32_t>(l) + 16; j < len[r]; j += 8) {
            const int64_t c = static_cast<int64_t>(col[src[r] + j]);
            const bool inr = static_cast<uint64_t>(c) < static_cast<uint64_t>(n_nodes);
            if (!inr) *bad = 4;
            s_key[dst[r] + j] = inr ? static_cast<uint32_t>(table[c]) : 0u;
            s_val[dst[r] + j] = has_w ? w[src[r] + j] : WT(0);
          }
      }
    }
    }
    __syncthreads();
    GS_STAMP(1);
    // (c1) rows of <= 32 entries: 8 lanes x 4 keys each, all 32 rows of the pass at once (cr_sort_rows)
#ifdef TGP_GEMM_STAMPS
    if (g_gs_ablate & 1) {
      if (tid < re - rs) n_out[r0 + rs + tid] = s_key[tid] & 1;
      rs = re;
      __syncthreads();
      continue;
    }
#endif
    {
      constexpr int LPR = 8;
      const int grp = tid / LPR;
      for (int i0 = rs; i0 < re; i0 += 256 / LPR) {
        const int i = i0 + grp;
        uint32_t b = 0, T = 0;
        if (i < re) {
          b = s_roff[i] - base;
          T = s_roff[i + 1] - s_roff[i];
        }
        // rows of 33..64 entries wait for (c1b); longer ones go out raw with the copy below, for cr_rows_long_kernel
        if (i < re && T > 32 && T <= 64 && (tid & (LPR - 1)) == 0) s_mid[atomicAdd(&s_nmid, 1)] = i;
        // (long rows are listed for cr_rows_long_kernel, which spreads the list over its workgroups: hub supernodes sit
        //  next to each other -- the first nodes of a preferential-attachment graph -- and a workgroup per 256 rows that
        //  sorted its own long rows one after the other took 150 us for a 4500-node batch with a dozen hubs)
        if (i < re && T > 64 && T <= static_cast<uint32_t>(CR_LONG) && (tid & (LPR - 1)) == 0)
          long_list[atomicAdd(bad + 1, 1)] = static_cast<uint32_t>(r0 + i);
        const bool mine = i < re && T <= 32;
        cr_sort_rows<LPR, WT>(s_key, s_val, b, mine ? T : 0, mine, static_cast<uint32_t>(r0 + i), has_w, reduce_op,
                              flags, eps, n_out + r0 + i);
      }
    }
    __syncthreads();
    GS_STAMP(2);
    // (c1b) rows of 33 .. 64 entries: 16 lanes x 4 keys each
    {
      constexpr int LPR = 16;
      const int nmid = s_nmid;
      for (int li0 = 0; li0 < nmid; li0 += 256 / LPR) {
        const int li = li0 + tid / LPR;
        const bool mine = li < nmid;
        const int i = mine ? s_mid[li] : rs;
        const uint32_t b = s_roff[i] - base, T = s_roff[i + 1] - s_roff[i];
        cr_sort_rows<LPR, WT>(s_key, s_val, b, mine ? T : 0, mine, static_cast<uint32_t>(r0 + i), has_w, reduce_op,
                              flags, eps, n_out + r0 + i);
      }
    }
    __syncthreads();
    // the pass's slot range goes out in one coalesced sweep: survivors at the head of every row's slots (what is
    // behind them is never read: the fill kernel takes n_out[r] entries from raw_off[r]), long rows still raw
    for (int t = tid; t < cnt; t += 256) {
      __builtin_nontemporal_store(s_key[t], tmp_c + base + t);
      if (has_w) __builtin_nontemporal_store(s_val[t], tmp_w + base + t);
    }
    rs = re;
    __syncthreads();
    GS_STAMP(3);
  }
}

// Rows of 65..1024 raw entries (rare: hub supernodes), left raw in tmp and listed by the gather kernel: a workgroup
// takes every gridDim.x-th row of the list; bitonic sort of (column << 32 | position) in LDS, sorted + merged in place.
template <typename WT>
__global__ __launch_bounds__(256) void cr_rows_long_kernel(uint32_t* __restrict__ tmp_c, WT* __restrict__ tmp_w,
                                                           const uint32_t* __restrict__ raw_off, int64_t K,
                                                           int64_t E, int reduce_op, int flags, WT eps,
                                                           const int* __restrict__ bad,
                                                           const uint32_t* __restrict__ long_list,
                                                           uint32_t* __restrict__ n_out) {
  __shared__ unsigned long long s_key[CR_LONG];
  __shared__ WT s_w[CR_LONG];
  __shared__ uint32_t s_cnt[4 * 4];
  if (*bad) return;
  const int tid = threadIdx.x;
  const int nlist = bad[1];
  for (int li = blockIdx.x; li < nlist; li += gridDim.x) {
    const int64_t r = long_list[li];
    const uint32_t b = raw_off[r];
    const uint32_t T = (r + 1 < K ? raw_off[r + 1] : static_cast<uint32_t>(E)) - b;
    uint32_t P = 64;
    while (P < T) P <<= 1;
    for (uint32_t i = tid; i < P; i += 256) {
      s_key[i] = i < T ? (static_cast<unsigned long long>(tmp_c[b + i]) << 32) | i : ~0ull;
      s_w[i] = (i < T && tmp_w) ? tmp_w[b + i] : WT(0);
    }
    __syncthreads();
    for (uint32_t k = 2; k <= P; k <<= 1) {
      for (uint32_t j = k >> 1; j > 0; j >>= 1) {
        for (uint32_t i = tid; i < P; i += 256) {
          const uint32_t x = i ^ j;
          if (x > i) {
            const bool up = (i & k) == 0;
            const unsigned long long a = s_key[i], c2 = s_key[x];
            if ((a > c2) == up) {
              s_key[i] = c2; s_key[x] = a;
              const WT t = s_w[i]; s_w[i] = s_w[x]; s_w[x] = t;
            }
          }
        }
        __syncthreads();
      }
    }
    bool keep[4];
    uint32_t rank[4], col[4];
    WT val[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const uint32_t i = it * 256 + tid;
      keep[it] = false; col[it] = 0; val[it] = WT(0);
      if (i < T) {
        const uint32_t c = static_cast<uint32_t>(s_key[i] >> 32);
        const bool head = i == 0 || static_cast<uint32_t>(s_key[i - 1] >> 32) != c;
        if (head) {
          WT acc = s_w[i];
          uint32_t cnt = 1;
          for (uint32_t q = i + 1; q < T && static_cast<uint32_t>(s_key[q] >> 32) == c; ++q, ++cnt)
            acc = cr_reduce(acc, s_w[q], reduce_op);
          if (tmp_w && reduce_op == TGP_MEAN) acc = acc / static_cast<WT>(cnt);
          bool k2 = true;
          if ((flags & TGP_REMOVE_SELF_LOOPS) && c == static_cast<uint32_t>(r)) k2 = false;
          if (tmp_w && (flags & TGP_EPS_FILTER) && !(cr_abs(acc) > eps)) k2 = false;
          keep[it] = k2; col[it] = c; val[it] = acc;
        }
      }
    }
    uint32_t total;
    block_compact_ranks<4>(keep, rank, total, s_cnt);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      if (keep[it]) {
        tmp_c[b + rank[it]] = col[it];
        if (tmp_w) tmp_w[b + rank[it]] = val[it];
      }
    }
    if (tid == 0) n_out[r] = total;
    __syncthreads();
  }
}

// ------------------------------------------------------------------ huge rows (r4)
// Supernode rows beyond CR_LONG raw entries -- the hubs of a power-law graph -- used to send the WHOLE list to the radix
// routes (r3: `graclus` forward on a 1M-node graph with ten 100 000-entry hubs 3.2 ms against 0.92 ms without).  Now
// only THEIR entries are sorted device-wide: the rows are listed, their raw entries gathered as (row rank : cluster
// column) keys with the input position as payload, sorted by the stable LSD radix sort of primitives.h (element count
// on the device), run heads fold their run in input order with the A6 filters fused, a scan ranks the survivors inside
// their row, and they land compacted at the head of the row's slots of tmp with n_out[r] set -- exactly what the LDS
// kernels leave for every other row, so the survivor scan and the fill take it from there.  Same output as the radix
// routes (tests compare both).
struct HugeWs {
  uint32_t* list;    // [HUGE_MAX] rows
  uint32_t* hoff;    // [HUGE_MAX + 1] first sub-list entry of every listed row; hoff[nh] = H
  uint64_t *k0, *k1;
  uint32_t *v0, *v1;
  float* sub_w;      // weights by sub-list position
  uint32_t* flags;   // keep flags of the sorted entries (zero behind H)
  uint32_t* ranks;   // their exclusive scan
  float* mval;       // merged weight of every run head
  uint32_t* sort_scratch;
  uint32_t* scan_scratch;
  int64_t* total;
};

__global__ __launch_bounds__(1024) void cr_huge_offsets_kernel(const uint32_t* __restrict__ list,
                                                               const uint32_t* __restrict__ raw_off, int64_t K,
                                                               int64_t E, int* __restrict__ bad,
                                                               uint32_t* __restrict__ hoff) {
  __shared__ uint32_t s_w[16];
  __shared__ uint32_t s_carry;
  if (bad[0] & ~8) return;
  const int nh = bad[2] < HUGE_MAX ? bad[2] : HUGE_MAX;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < nh; base += 1024) {
    const int li = base + static_cast<int>(threadIdx.x);
    uint32_t T = 0;
    if (li < nh) {
      const int64_t r = list[li];
      T = (r + 1 < K ? raw_off[r + 1] : static_cast<uint32_t>(E)) - raw_off[r];
    }
    const uint32_t inc = wave_incl_scan(T);
    if (lane_id() == WAVE - 1) s_w[wave_id()] = inc;
    __syncthreads();
    uint32_t off = s_carry, tot = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint32_t c = s_w[j];
      if (j < wave_id()) off += c;
      tot += c;
    }
    if (li < nh) hoff[li] = off + inc - T;
    __syncthreads();
    if (threadIdx.x == 0) s_carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    hoff[nh] = s_carry;
    bad[3] = static_cast<int>(s_carry);  // H: the sort's element count
  }
}

// (raw_off[r + 1] is not the end of row r when r + 1 is an EMPTY row's shared offset -- it is: an empty row shares its
//  successor's first slot, so the difference is still row r's length)
__global__ __launch_bounds__(256) void cr_huge_gather_kernel(
    const int64_t* __restrict__ col, const float* __restrict__ w, const int32_t* __restrict__ table,
    const int32_t* __restrict__ a_row_ptr, const uint32_t* __restrict__ seg_src, const uint32_t* __restrict__ seg_dst,
    const uint32_t* __restrict__ raw_off, const uint32_t* __restrict__ list, const uint32_t* __restrict__ hoff,
    int64_t n_nodes, int colbits, int* __restrict__ bad, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
    float* __restrict__ sub_w) {
  if (bad[0] & ~8) return;
  const int nh = bad[2] < HUGE_MAX ? bad[2] : HUGE_MAX;
  for (int li = 0; li < nh; ++li) {
    const int64_t r = list[li];
    const uint32_t b = raw_off[r], h0 = hoff[li], T = hoff[li + 1] - h0;
    const int p0 = a_row_ptr[r], p1 = a_row_ptr[r + 1];
    for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < T; t += gridDim.x * 256) {
      const uint32_t slot = b + t;
      int lo = p0, hi = p1;  // last member p with seg_dst[p] <= slot (members without edges share their successor's)
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (seg_dst[mid] <= slot) lo = mid; else hi = mid;
      }
      const uint32_t e = seg_src[lo] + (slot - seg_dst[lo]);
      const int64_t c = col[e];
      const bool inr = static_cast<uint64_t>(c) < static_cast<uint64_t>(n_nodes);
      if (!inr) *bad = 4;
      const uint64_t cl = inr ? static_cast<uint32_t>(table[c]) : 0u;
      keys[h0 + t] = (static_cast<uint64_t>(li) << colbits) | cl;
      vals[h0 + t] = h0 + t;
      sub_w[h0 + t] = w ? w[e] : 0.f;
    }
  }
}

// run heads of the sorted sub-list fold their run (input order: the sort is stable and the payload is the position)
__global__ __launch_bounds__(256) void cr_huge_merge_kernel(const uint64_t* __restrict__ keys,
                                                            const uint32_t* __restrict__ vals,
                                                            const float* __restrict__ sub_w, int has_w,
                                                            const uint32_t* __restrict__ list, int colbits,
                                                            int reduce_op, int flags, float eps,
                                                            const int* __restrict__ bad, int64_t n_max,
                                                            uint32_t* __restrict__ keep, float* __restrict__ mval) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n_max) return;
  const int64_t H = (bad[0] & ~8) ? 0 : bad[3];
  uint32_t k = 0;
  if (i < H) {
    const uint64_t key = keys[i];
    if (i == 0 || keys[i - 1] != key) {
      float acc = has_w ? sub_w[vals[i]] : 0.f;
      uint32_t cnt = 1;
      for (int64_t q = i + 1; q < H && keys[q] == key; ++q, ++cnt)
        if (has_w) acc = cr_reduce(acc, sub_w[vals[q]], reduce_op);
      if (has_w && reduce_op == TGP_MEAN) acc = acc / static_cast<float>(cnt);
      const uint32_t c = static_cast<uint32_t>(key & ((1ull << colbits) - 1ull));
      const uint32_t r = list[key >> colbits];
      k = 1;
      if ((flags & TGP_REMOVE_SELF_LOOPS) && c == r) k = 0;
      if (has_w && (flags & TGP_EPS_FILTER) && !(fabsf(acc) > eps)) k = 0;
      mval[i] = acc;
    }
  }
  keep[i] = k;
}

__global__ __launch_bounds__(256) void cr_huge_write_kernel(const uint64_t* __restrict__ keys,
                                                            const uint32_t* __restrict__ keep,
                                                            const uint32_t* __restrict__ ranks,
                                                            const float* __restrict__ mval,
                                                            const uint32_t* __restrict__ list,
                                                            const uint32_t* __restrict__ hoff,
                                                            const uint32_t* __restrict__ raw_off, int colbits,
                                                            const int* __restrict__ bad, const int64_t* __restrict__ total,
                                                            int64_t n_max, uint32_t* __restrict__ tmp_c,
                                                            float* __restrict__ tmp_w, uint32_t* __restrict__ n_out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if ((bad[0] & ~8) || i >= bad[3]) return;
  const uint64_t key = keys[i];
  const uint32_t li = static_cast<uint32_t>(key >> colbits);
  const uint32_t h0 = hoff[li], h1 = hoff[li + 1];
  const uint32_t first = ranks[h0];
  const int64_t r = list[li];
  if (i == h0) n_out[r] = (h1 < n_max ? ranks[h1] : static_cast<uint32_t>(*total)) - first;
  if (keep[i]) {
    const uint32_t dst = raw_off[r] + (ranks[i] - first);
    tmp_c[dst] = static_cast<uint32_t>(key & ((1ull << colbits) - 1ull));
    if (tmp_w) tmp_w[dst] = mval[i];
  }
}

// ------------------------------------------------------------------ K7: fill
// One workgroup per FILL_ROWS consecutive rows: their survivors are one contiguous run of the output, so the
// workgroup walks that run output-parallel (thread t -> output slot o0 + t: fully coalesced stores), finds each
// slot's row by binary search over the rows' output offsets in LDS, and reads the survivor from the row's
// compacted run in tmp (contiguous up to the gaps the merge left).
constexpr int FILL_ROWS = 64;
// rows with more survivors than this (only a hub row can have them: a row the LDS kernels sorted holds <= CR_LONG) are
// left to cr_fill_huge_kernel, which spreads each over the whole grid -- one workgroup walking a 200 000-entry row slot
// by slot took 3.5 ms (r4, ten hubs)
constexpr uint32_t FILL_LONG = CR_LONG;
template <typename WT>
__global__ __launch_bounds__(256) void cr_fill_kernel(const uint32_t* __restrict__ tmp_c,
                                                      const WT* __restrict__ tmp_w,
                                                      const uint32_t* __restrict__ raw_off,
                                                      const uint32_t* __restrict__ out_off,
                                                      const int64_t* __restrict__ total, int64_t K,
                                                      const int* __restrict__ bad /* NULL, or: declined -> no-op */,
                                                      int64_t* __restrict__ out_row, int64_t* __restrict__ out_col,
                                                      WT* __restrict__ out_w) {
  __shared__ uint32_t s_out[FILL_ROWS + 1], s_raw[FILL_ROWS];
  __shared__ unsigned long long s_hub;
  if (bad && *bad) return;
  const int tid = threadIdx.x;
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * FILL_ROWS;
  const int nr = static_cast<int>(K - r0 < FILL_ROWS ? K - r0 : FILL_ROWS);
  if (tid <= nr) s_out[tid] = r0 + tid < K ? out_off[r0 + tid] : static_cast<uint32_t>(*total);
  if (tid < nr) s_raw[tid] = raw_off[r0 + tid];
  __syncthreads();
  if (tid < 64) {
    const unsigned long long m = __ballot(tid < nr && s_out[tid + 1] - s_out[tid] > FILL_LONG);
    if (tid == 0) s_hub = m;
  }
  __syncthreads();
  unsigned long long hub = s_hub;
  int i0 = 0;
  while (i0 < nr) {  // maximal runs [i0, i1) of rows without a hub row (one run, the whole block, almost always)
    if ((hub >> i0) & 1ull) {
      ++i0;
      continue;
    }
    const unsigned long long rest = hub >> i0;
    const int i1 = rest ? i0 + __builtin_ctzll(rest) : nr;
    const uint32_t o0 = s_out[i0], cnt = s_out[i1] - o0;
    for (uint32_t t = tid; t < cnt; t += 256) {
      const uint32_t o = o0 + t;
      int lo = i0, hi = i1;  // last row i with s_out[i] <= o (rows without survivors share their successor's offset)
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_out[mid] <= o) lo = mid; else hi = mid;
      }
      const uint32_t src = s_raw[lo] + (o - s_out[lo]);
      out_row[o] = r0 + lo;
      out_col[o] = tmp_c[src];
      if (out_w) out_w[o] = tmp_w[src];
    }
    i0 = i1;
  }
}

// the listed hub rows with more than FILL_LONG survivors: every row a grid-wide strided copy
__global__ __launch_bounds__(256) void cr_fill_huge_kernel(const uint32_t* __restrict__ tmp_c,
                                                           const float* __restrict__ tmp_w,
                                                           const uint32_t* __restrict__ raw_off,
                                                           const uint32_t* __restrict__ out_off,
                                                           const uint32_t* __restrict__ n_out, int64_t K,
                                                           const uint32_t* __restrict__ list, const int* __restrict__ bad,
                                                           int64_t* __restrict__ out_row, int64_t* __restrict__ out_col,
                                                           float* __restrict__ out_w) {
  const int nh = bad[2] < HUGE_MAX ? bad[2] : HUGE_MAX;
  for (int li = 0; li < nh; ++li) {
    const int64_t r = list[li];
    const uint32_t o0 = out_off[r], n = n_out[r];
    if (n <= FILL_LONG) continue;  // cr_fill_kernel wrote it
    const uint32_t b = raw_off[r];
    for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) {
      out_row[o0 + t] = r;
      out_col[o0 + t] = tmp_c[b + t];
      if (out_w) out_w[o0 + t] = tmp_w[b + t];
    }
  }
}

// r4: the survivor scan as ONE launch that also hands the count to the host.  A workgroup owns SCAN_TILE rows; the
// survivors in front of them come from the epoch-tagged decoupled look-back of lookback.h (a few hundred tiles, all
// published within a microsecond of each other: 256 predecessor words per round), and the last workgroup leaves the
// total in *total, the caller-visible count in *d_count and {epoch << 34 | count as a 34-bit two's complement number} in
// *result -- pinned host memory the caller polls: no device-to-host copy, no stream synchronise.  Replaces
// scan_tile_sums + scan_apply + the count's copy kernel (r4, C4: 4.8 + 7.8 us + the `.item()` round trip).
// (The look-back does NOT pay inside the fill -- thousands of 64-row tiles, 107 us against 46 + 19 -- nor does running
// the fill inside the count call with the host wait at its end: profiles/r04_coalesce_tail_experiments.md.)
__global__ __launch_bounds__(256) void cr_scan_publish_kernel(const uint32_t* __restrict__ n_out, int64_t K,
                                                              uint32_t* __restrict__ out_off,
                                                              int64_t* __restrict__ total, const int* __restrict__ bad,
                                                              int64_t* __restrict__ d_count,
                                                              unsigned long long* status, unsigned long long* result,
                                                              unsigned long long tag, int ticket) {
  __shared__ uint32_t s_w[4];
  __shared__ uint32_t s_base;
  __shared__ int s_tile;
  const int tid = threadIdx.x;
  const int tile = sps_tile_id(status + 1, tag, ticket, &s_tile);
  const int64_t base = static_cast<int64_t>(tile) * SCAN_TILE + static_cast<int64_t>(tid) * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS], sacc = 0;
  if (base + SCAN_ITEMS <= K) {  // (the workspace carves 256-byte aligned arrays)
    const uint4* p = reinterpret_cast<const uint4*>(n_out + base);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS / 4; ++i) {
      const uint4 q = p[i];
      v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) v[i] = base + i < K ? n_out[base + i] : 0u;
  }
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) sacc += v[i];
  uint32_t tile_total;
  const uint32_t local = block_excl_scan_256(sacc, s_w, &tile_total);
  if (tid < WAVE) {  // wave 0: publish, look back, publish the prefix
    if (tid == 0)
      sps_store(status + 2 + tile, tag | (tile == 0 ? SPS_PRE : SPS_AGG) | static_cast<unsigned long long>(tile_total));
    uint32_t excl = 0;
    bool refused = false;
    if (tile > 0) {
      sps_lookback<4>(status, tile, tag, &excl, &refused);
      if (tid == 0)
        sps_store(status + 2 + tile, tag | SPS_PRE | (refused ? 0x80000000ull : 0ull) |
                                         static_cast<unsigned long long>((excl + tile_total) & 0x7FFFFFFFu));
    }
    if (tid == 0) {
      s_base = excl;
      if (tile == static_cast<int>(gridDim.x) - 1) {
        const int64_t t = static_cast<int64_t>(excl) + tile_total;
        const int b = *bad;  // a look-back that ran into its spin bound declines the call like an unmet precondition
        const int64_t count = (b || refused) ? (b == 8 ? -5 : -1) : t;
        *total = t;
        if (d_count) *d_count = count;
        if (result)
          __hip_atomic_store(result, tag | (static_cast<unsigned long long>(count) & ((1ull << SPS_EPOCH_SHIFT) - 1)),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  __syncthreads();
  uint32_t run = s_base + local;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    if (base + i < K) out_off[base + i] = run;
    run += v[i];
  }
}

// ======================================================================================================
// r3: the row-sorted path as ONE heavy kernel (cr_fused_kernel) + a widening fill.
//
// What the staged pipeline above spends outside its gather-sort kernel -- member degree sums + scan + row offsets
// (30 us at C4), the survivor scan (15-19 us), the long-row kernel's launch, and a fill that re-reads weights it only
// copies -- comes from handing positions between kernels through global arrays.  Here a workgroup derives all of it
// locally and the only cross-workgroup quantity, the number of survivors in front of it, comes from a decoupled
// look-back over 8-byte {state, value} granules (agent-scope relaxed atomics: the data IS the flag, R2 of the
// programming guide's Guideline 16).  A tile waits for the tiles in front of it, so the kernel is only launched with
// at most FZ_MAX_TILES tiles -- one launch wave, every workgroup resident -- and every spin is bounded (a timeout
// declines the call to the staged pipeline, so no result depends on dispatch order or placement).  Measured r3 at
// C4 (17 k tiles): the waiting costs 95 us of a 288 us kernel and a ticket counter for arrival-order tiles 600 us more
// (one word takes ~88 atomics per us), against 186 us for the staged kernels it replaces: large lists stay on the
// staged pipeline, batches of small graphs (where ten launches of a few us each dominate) take this one:
//   * rows r0 .. r0 + R of the tile -> their members (inverted index) -> each member's edge range (node_ptr) ->
//     block scan of the member degrees = LDS slots.  No seg_src / seg_dst / raw_off arrays.
//   * gather through the int32 cluster table into LDS, in-row sort + merge + filters exactly as above
//     (cr_sort_rows); rows of 65..1024 entries are sorted by the whole workgroup in LDS (solo pass) instead of a
//     second kernel.
//   * survivors leave at their FINAL offsets: weights straight into the caller's output buffer, columns as uint32
//     into tmp (the int64 [2, n] tensor can only be allocated once n is known), per-row output offsets for the fill.
// The fill then reads 4 bytes and writes 16 per survivor (28 + 4 before).  Declines: unsorted rows / a row longer
// than 1024 entries / K >= 2^22 with a long row (d_count = -1: radix routes), more than 256 members in a tile
// (d_count = -3: the staged row-local pipeline above takes any member count).
constexpr int FZ_MEMBERS = 256;
constexpr int FZ_MAX_TILES = 1024;  // all tiles of a call are resident together (256 CUs x 6 workgroups): see below
constexpr uint32_t FZ_POS_BITS = 10;  // long rows: key = column << 10 | position

__device__ __forceinline__ unsigned long long fz_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void fz_store(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
constexpr unsigned long long FZ_AGG = 1ull << 32, FZ_PRE = 2ull << 32;

// Exclusive prefix of tile `tile` (survivors of all earlier tiles); called by every lane of ONE wave.  Returns false
// when the call was declined meanwhile (a predecessor may then never publish).
__device__ __forceinline__ bool fz_lookback(const unsigned long long* __restrict__ status, int64_t tile,
                                            int* __restrict__ bad, uint32_t* excl_out) {
  const int lane = lane_id();
  uint32_t excl = 0;
  int64_t j = tile - 1;
  while (j >= 0) {
    const int64_t idx = j - lane;
    unsigned long long st = idx >= 0 ? fz_load(status + idx) : FZ_PRE;
    int spins = 0;
    while (__any((st >> 32) == 0)) {
      ++spins;
      if ((spins & 15) == 0) {
        int b = 0;
        if (lane == 0) b = __hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__shfl(b, 0, 64) != 0) return false;
        if (spins > (1 << 20)) {  // every spin is bounded: give the call back to the staged pipeline
          if (lane == 0) atomicOr(bad, 16);
          return false;
        }
      }
      if (spins > 4) __builtin_amdgcn_s_sleep(2);
      if ((st >> 32) == 0) st = fz_load(status + idx);
    }
    const unsigned long long pre = __ballot((st >> 32) == 2);
    const int first = pre ? __builtin_ctzll(pre) : 64;  // nearest predecessor that already knows its prefix
    uint32_t v = lane <= first ? static_cast<uint32_t>(st) : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    excl += v;
    if (pre) break;
    j -= 64;
  }
  *excl_out = excl;
  return true;
}

// One row of 65 .. 1024 entries, alone in LDS at slots [0, T): workgroup-wide bitonic on (column << 10 | position),
// duplicates folded by the run heads in input order, filters, survivors compacted back to slots [0, n).
__device__ __forceinline__ uint32_t fz_sort_long(uint32_t* s_key, float* s_val, uint32_t* s_cnt, uint32_t T,
                                                 uint32_t row_id, bool has_w, int reduce_op, int flags, float eps) {
  const int tid = threadIdx.x;
  uint32_t P = 128;
  while (P < T) P <<= 1;
  for (uint32_t i = tid; i < P; i += 256) s_key[i] = i < T ? (s_key[i] << FZ_POS_BITS) | i : 0xFFFFFFFFu;
  __syncthreads();
  for (uint32_t k = 2; k <= P; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = tid; i < P; i += 256) {
        const uint32_t x = i ^ j;
        if (x > i) {
          const bool up = (i & k) == 0;
          const uint32_t a = s_key[i], c2 = s_key[x];
          if ((a > c2) == up) { s_key[i] = c2; s_key[x] = a; }
        }
      }
      __syncthreads();
    }
  }
  bool keep[4];
  uint32_t rank[4], colv[4];
  float val[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const uint32_t i = it * 256 + tid;
    keep[it] = false; colv[it] = 0; val[it] = 0.f;
    if (i < T) {
      const uint32_t kk = s_key[i], c = kk >> FZ_POS_BITS;
      const bool head = i == 0 || (s_key[i - 1] >> FZ_POS_BITS) != c;
      if (head) {
        float acc = has_w ? s_val[kk & 1023u] : 0.f;
        uint32_t cnt = 1;
        for (uint32_t q = i + 1; q < T; ++q, ++cnt) {
          const uint32_t k2 = s_key[q];
          if ((k2 >> FZ_POS_BITS) != c) break;
          if (has_w) acc = cr_reduce(acc, s_val[k2 & 1023u], reduce_op);
        }
        if (has_w && reduce_op == TGP_MEAN) acc = acc / static_cast<float>(cnt);
        bool k3 = true;
        if ((flags & TGP_REMOVE_SELF_LOOPS) && c == row_id) k3 = false;
        if (has_w && (flags & TGP_EPS_FILTER) && !(fabsf(acc) > eps)) k3 = false;
        keep[it] = k3; colv[it] = c; val[it] = acc;
      }
    }
  }
  uint32_t total;
  block_compact_ranks<4>(keep, rank, total, s_cnt);  // (two barriers: every read of the sorted keys is done)
#pragma unroll
  for (int it = 0; it < 4; ++it)
    if (keep[it]) {
      s_key[rank[it]] = colv[it];
      if (has_w) s_val[rank[it]] = val[it];
    }
  __syncthreads();
  return total;
}

template <typename ColT>
__global__ __launch_bounds__(256, 6) void cr_fused_kernel(
    const ColT* __restrict__ col, const float* __restrict__ w, int64_t E, const int32_t* __restrict__ table,
    const int32_t* __restrict__ a_row_ptr, const int32_t* __restrict__ a_perm, const uint32_t* __restrict__ node_ptr,
    int64_t K, int rows_per_wg, int reduce_op, int flags, float eps, int* __restrict__ bad,
    unsigned long long* __restrict__ status, uint32_t* __restrict__ tmp_c,
    float* __restrict__ out_w, uint32_t* __restrict__ out_off, int64_t* __restrict__ total, int64_t n_nodes) {
  __shared__ uint32_t s_key[GS_CAP];
  __shared__ float s_val[GS_CAP];
  __shared__ uint32_t s_src[FZ_MEMBERS + 1], s_dst[FZ_MEMBERS + 1];
  __shared__ uint32_t s_roff[GS_ROWS + 1], s_nout[GS_ROWS], s_ooff[GS_ROWS + 1];
  __shared__ int32_t s_rp[GS_ROWS + 1];
  __shared__ int s_mid[GS_ROWS];
  __shared__ uint32_t s_w4[16];
  __shared__ int s_nmid, s_anylong;
  __shared__ uint32_t s_base;
  __shared__ int s_ok;
  const int tid = threadIdx.x;
  if (tid == 0) {
    s_anylong = 0;
    // one lane reads the decline flag for the workgroup (per-thread reads could disagree and split a barrier)
    s_ok = __hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
  }
  __syncthreads();
  const int64_t tile = blockIdx.x;
  const int64_t ntiles = gridDim.x;
  const int64_t r0 = tile * rows_per_wg;
  const int nrows = static_cast<int>(K - r0 < rows_per_wg ? K - r0 : rows_per_wg);
  const bool has_w = w != nullptr;
  if (!s_ok) return;  // declined already (its successors leave their look-back through the same flag)
  if (tid <= nrows) s_rp[tid] = a_row_ptr[r0 + tid];
  __syncthreads();
  const int p_lo = s_rp[0], M = s_rp[nrows] - p_lo;
  if (M > FZ_MEMBERS) {
    if (tid == 0) atomicOr(bad, 8);
    return;
  }
  {  // member degrees -> LDS slots
    uint32_t src = 0, len = 0;
    if (tid < M) {
      const int32_t node = a_perm[p_lo + tid];
      src = node_ptr[node];
      len = node_ptr[node + 1] - src;
    }
    uint32_t cnt_all;
    const uint32_t dst = block_excl_scan_256(len, s_w4, &cnt_all);
    s_src[tid] = src;
    s_dst[tid] = dst;
    if (tid == 0) s_dst[FZ_MEMBERS] = cnt_all;
  }
  __syncthreads();
  if (tid <= nrows) {
    const int m = s_rp[tid] - p_lo;
    s_roff[tid] = m < FZ_MEMBERS ? s_dst[m] : s_dst[FZ_MEMBERS];
    // (member M of a full tile: s_dst[256] holds the total; for M < 256 thread M's scan value is the total too)
  }
  __syncthreads();
  if (tid < nrows) {
    const uint32_t T = s_roff[tid + 1] - s_roff[tid];
    if (T > 64) s_anylong = 1;
    if (T > static_cast<uint32_t>(CR_LONG) || (T > 64 && K >= (1ll << 22))) {
      atomicOr(bad, 2);
      s_ok = 0;
    }
  }
  __syncthreads();
  if (!s_ok) return;

  const bool single = !s_anylong && s_roff[nrows] <= static_cast<uint32_t>(GS_CAP);
  bool have_base = false;
  uint32_t out_base = 0, wg_total = 0;
  int rs = 0;
  while (rs < nrows) {
    int re = rs + 1;
    if (single) {
      re = nrows;
    } else if (s_roff[rs + 1] - s_roff[rs] <= 64) {
      while (re < nrows && s_roff[re + 1] - s_roff[re] <= 64 &&
             s_roff[re + 1] - s_roff[rs] <= static_cast<uint32_t>(GS_CAP))
        ++re;
    }
    const uint32_t base = s_roff[rs];
    const int cnt = static_cast<int>(s_roff[re] - base);
    const int m_lo = s_rp[rs] - p_lo, Mp = s_rp[re] - s_rp[rs];
    if (tid == 0) s_nmid = 0;
    {  // gather: 8 lanes per member, two rounds of 32 members x two steps of 8 edges requested before any is used
      const int grp = tid >> 3, l = tid & 7;
      constexpr int GR = 2;
      for (int m0 = 0; m0 < Mp; m0 += 32 * GR) {
        uint32_t src[GR], dst[GR], len[GR];
#pragma unroll
        for (int r = 0; r < GR; ++r) {
          const int m = m0 + r * 32 + grp;
          const bool v = m < Mp;
          src[r] = v ? s_src[m_lo + m] : 0u;
          const uint32_t d = v ? s_dst[m_lo + m] : 0u;
          len[r] = v ? s_dst[m_lo + m + 1] - d : 0u;  // s_dst[M .. 256] all hold the tile's total
          dst[r] = d - base;
        }
        ColT cc[GR][2];
        float wv[GR][2];
#pragma unroll
        for (int r = 0; r < GR; ++r)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const uint32_t j = static_cast<uint32_t>(l + 8 * q);
            const bool ok = j < len[r];
            cc[r][q] = ok ? __builtin_nontemporal_load(col + src[r] + j) : 0;
            wv[r][q] = (ok && has_w) ? __builtin_nontemporal_load(w + src[r] + j) : 0.f;
          }
        uint32_t tv[GR][2];  // all table look-ups of the lane in flight together (see cr_gather_sort_kernel)
#pragma unroll
        for (int r = 0; r < GR; ++r)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const uint32_t j = static_cast<uint32_t>(l + 8 * q);
            const bool inr = static_cast<uint64_t>(static_cast<int64_t>(cc[r][q])) < static_cast<uint64_t>(n_nodes);
            const int64_t idx = (j < len[r] && inr) ? static_cast<int64_t>(cc[r][q]) : 0;
            tv[r][q] = static_cast<uint32_t>(table[idx]);
          }
#pragma unroll
        for (int r = 0; r < GR; ++r)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const uint32_t j = static_cast<uint32_t>(l + 8 * q);
            if (j < len[r]) {
              const bool inr = static_cast<uint64_t>(static_cast<int64_t>(cc[r][q])) < static_cast<uint64_t>(n_nodes);
              if (!inr) atomicOr(bad, 4);
              s_key[dst[r] + j] = inr ? tv[r][q] : 0u;
              s_val[dst[r] + j] = wv[r][q];
            }
          }
#pragma unroll
        for (int r = 0; r < GR; ++r)
          for (uint32_t j = static_cast<uint32_t>(l) + 16; j < len[r]; j += 8) {
            const int64_t c = static_cast<int64_t>(col[src[r] + j]);
            const bool inr = static_cast<uint64_t>(c) < static_cast<uint64_t>(n_nodes);
            if (!inr) atomicOr(bad, 4);
            s_key[dst[r] + j] = inr ? static_cast<uint32_t>(table[c]) : 0u;
            s_val[dst[r] + j] = has_w ? w[src[r] + j] : 0.f;
          }
      }
    }
    __syncthreads();
    if (cnt > 64 && re == rs + 1 && s_roff[rs + 1] - s_roff[rs] > 64) {
      const uint32_t n = fz_sort_long(s_key, s_val, s_w4, static_cast<uint32_t>(cnt), static_cast<uint32_t>(r0 + rs),
                                      has_w, reduce_op, flags, eps);
      if (tid == 0) s_nout[rs] = n;
    } else {
      {
        constexpr int LPR = 8;
        const int grp = tid / LPR;
        for (int i0 = rs; i0 < re; i0 += 256 / LPR) {
          const int i = i0 + grp;
          uint32_t b = 0, T = 0;
          if (i < re) {
            b = s_roff[i] - base;
            T = s_roff[i + 1] - s_roff[i];
          }
          if (i < re && T > 32 && (tid & (LPR - 1)) == 0) s_mid[atomicAdd(&s_nmid, 1)] = i;
          const bool mine = i < re && T <= 32;
          cr_sort_rows<LPR>(s_key, s_val, b, mine ? T : 0, mine, static_cast<uint32_t>(r0 + i), has_w, reduce_op, flags,
                            eps, s_nout + (mine ? i : 0));
        }
      }
      __syncthreads();
      {
        constexpr int LPR = 16;
        const int nmid = s_nmid;
        for (int li0 = 0; li0 < nmid; li0 += 256 / LPR) {
          const int li = li0 + tid / LPR;
          const bool mine = li < nmid;
          const int i = mine ? s_mid[li] : rs;
          const uint32_t b = s_roff[i] - base, T = s_roff[i + 1] - s_roff[i];
          cr_sort_rows<LPR>(s_key, s_val, b, mine ? T : 0, mine, static_cast<uint32_t>(r0 + i), has_w, reduce_op, flags,
                            eps, s_nout + i);
        }
      }
    }
    __syncthreads();
    // output offsets of the pass's rows (<= 32 rows: one wave scans them)
    if (tid < 64) {
      const int i = rs + tid;
      const uint32_t v = i < re ? s_nout[i] : 0u;
      const uint32_t inc = wave_incl_scan(v);
      if (i < re) s_ooff[i] = inc - v;
      if (tid == 63) s_ooff[re] = inc;  // rows beyond re contributed 0
    }
    __syncthreads();
    const uint32_t pass_total = s_ooff[re];
    if (!have_base) {
      if (tid < 64) {
        if (single && tid == 0) fz_store(status + tile, FZ_AGG | pass_total);
        uint32_t excl = 0;
        const bool ok = fz_lookback(status, tile, bad, &excl);
        if (tid == 0) {
          s_base = excl;
          s_ok = ok ? 1 : 0;
          if (single && ok) fz_store(status + tile, FZ_PRE | (excl + pass_total));
        }
      }
      __syncthreads();
      if (!s_ok) return;
      out_base = s_base;
      have_base = true;
    }
    const uint32_t o0 = out_base + wg_total;
    if (tid < re - rs) out_off[r0 + rs + tid] = o0 + s_ooff[rs + tid];
    for (uint32_t t = tid; t < pass_total; t += 256) {
      int lo = rs, hi = re;  // last row i of the pass with s_ooff[i] <= t
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_ooff[mid] <= t) lo = mid; else hi = mid;
      }
      const uint32_t slot = (s_roff[lo] - base) + (t - s_ooff[lo]);
      __builtin_nontemporal_store(s_key[slot], tmp_c + o0 + t);
      if (has_w) __builtin_nontemporal_store(s_val[slot], out_w + o0 + t);
    }
    wg_total += pass_total;
    rs = re;
    __syncthreads();
  }
  if (tid == 0) {
    if (!single) fz_store(status + tile, FZ_PRE | (out_base + wg_total));
    if (tile == ntiles - 1) {
      *total = static_cast<int64_t>(out_base) + wg_total;
      out_off[K] = out_base + wg_total;
    }
  }
}

static __global__ void fz_decline_kernel(int64_t* __restrict__ d_count) { *d_count = -3; }

// d_count = total, or the decline code: -2 an id out of range (the general path reports it), -3 only the fused
// kernel's member limit (the staged row-local pipeline can take the call), -1 unsorted rows / a row too long.
static __global__ void fz_finish_kernel(const int* __restrict__ bad, const int64_t* __restrict__ total,
                                        int64_t* __restrict__ d_count) {
  const int b = *bad;
  *d_count = b == 0 ? *total : ((b & 4) ? -1 : ((b & 3) ? -1 : -3));
}

// out_col[o] = tmp_c[o] widened, out_row[o] = the row whose output range holds o: one workgroup per 64 rows.
__global__ __launch_bounds__(256) void fz_fill_kernel(const uint32_t* __restrict__ tmp_c,
                                                      const uint32_t* __restrict__ out_off, int64_t K,
                                                      int64_t* __restrict__ out_row, int64_t* __restrict__ out_col) {
  __shared__ uint32_t s_out[FILL_ROWS + 1];
  const int tid = threadIdx.x;
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * FILL_ROWS;
  const int nr = static_cast<int>(K - r0 < FILL_ROWS ? K - r0 : FILL_ROWS);
  if (tid <= nr) s_out[tid] = out_off[r0 + tid];
  __syncthreads();
  const uint32_t o0 = s_out[0], cnt = s_out[nr] - o0;
  for (uint32_t t = tid; t < cnt; t += 256) {
    const uint32_t o = o0 + t;
    int lo = 0, hi = nr;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (s_out[mid] <= o) lo = mid; else hi = mid;
    }
    out_row[o] = r0 + lo;
    out_col[o] = tmp_c[o];
  }
}

struct FzWs {
  int32_t* table;            // [N]
  uint32_t* node_ptr;        // [N + 1]
  uint32_t* out_off;         // [K + 1]
  uint32_t* tmp_c;           // [E]
  unsigned long long* status;  // [tiles] + control words behind it (one memset)
  int* bad;
  int64_t* total;
  size_t zero_bytes;
};

static size_t fz_layout(void* ws, int64_t E, int64_t N, int64_t K, FzWs* out) {
  Carver cv(ws);
  const size_t n = static_cast<size_t>(N > 0 ? N : 1), k = static_cast<size_t>(K > 0 ? K : 1);
  const size_t e = static_cast<size_t>(E > 0 ? E : 1);
  FzWs s;
  // the zeroed block comes first (Guideline 16: a block of its own at the allocation's start, a multiple of 16 bytes)
  const size_t tiles = k + 2;  // rows_per_wg >= 1
  s.status = cv.take<unsigned long long>(tiles + 6);
  s.bad = reinterpret_cast<int*>(s.status + tiles + 1);
  s.total = reinterpret_cast<int64_t*>(s.status + tiles + 2);
  s.zero_bytes = (tiles + 6) * sizeof(unsigned long long) / 16 * 16;
  s.table = cv.take<int32_t>(n);
  s.node_ptr = cv.take<uint32_t>(n + 1);
  s.out_off = cv.take<uint32_t>(k + 1);
  s.tmp_c = cv.take<uint32_t>(e);
  if (out) *out = s;
  return cv.off;
}

struct CrWs {
  int32_t* table;        // [N]
  uint32_t* node_ptr;    // [N+1]
  uint32_t* seg_src;     // [N] first edge of every member (inverted-index order)
  uint32_t* seg_dst;     // [N + 1] first slot of every member, then E
  uint32_t* raw_off;     // [K]
  uint32_t* n_out;       // [K]
  uint32_t* out_off;     // [K]
  uint32_t* tmp_c;       // [E]
  float* tmp_w;          // [E]
  uint32_t* scan_scratch;
  int64_t* total;
  int* bad;              // [0] status, [1] number of long rows (65 .. CR_LONG raw entries) listed in long_list
  uint32_t* long_list;   // [E / 64 + 2] long rows in the order the gather kernel met them
};

static size_t cr_layout(void* ws, int64_t E, int64_t N, int64_t K, CrWs* out, size_t weight_bytes = sizeof(float)) {
  Carver cv(ws);
  const size_t n = static_cast<size_t>(N > 0 ? N : 1), k = static_cast<size_t>(K > 0 ? K : 1);
  const size_t e = static_cast<size_t>(E > 0 ? E : 1);
  CrWs s;
  s.table = cv.take<int32_t>(n);
  s.node_ptr = cv.take<uint32_t>(n + 1);
  s.seg_src = cv.take<uint32_t>(n);
  s.seg_dst = cv.take<uint32_t>(n + 1);
  s.raw_off = cv.take<uint32_t>(k);
  s.n_out = cv.take<uint32_t>(k);
  s.out_off = cv.take<uint32_t>(k);
  s.tmp_c = cv.take<uint32_t>(e);
  s.tmp_w = reinterpret_cast<float*>(cv.take<char>(e * weight_bytes));  // (doubles for the float64 entry points)
  s.scan_scratch = cv.take<uint32_t>(2 * static_cast<size_t>(cdiv(k, SCAN_TILE) + cdiv(n, MS_TILE)) + 16);
  s.total = cv.take<int64_t>(2);
  s.bad = cv.take<int>(4);
  s.long_list = cv.take<uint32_t>(e / 64 + 2);
  if (out) *out = s;
  return cv.off;
}

static size_t huge_layout(void* ws, size_t off0, int64_t E, HugeWs* out) {
  Carver cv(ws);
  cv.off = off0;
  const size_t e = static_cast<size_t>(E > 0 ? E : 1);
  HugeWs h;
  h.list = cv.take<uint32_t>(HUGE_MAX);
  h.hoff = cv.take<uint32_t>(HUGE_MAX + 1);
  h.k0 = cv.take<uint64_t>(e);
  h.k1 = cv.take<uint64_t>(e);
  h.v0 = cv.take<uint32_t>(e);
  h.v1 = cv.take<uint32_t>(e);
  h.sub_w = cv.take<float>(e);
  h.flags = cv.take<uint32_t>(e);
  h.ranks = cv.take<uint32_t>(e);
  h.mval = cv.take<float>(e);
  h.sort_scratch = cv.take<uint32_t>(sort_scratch_words());
  h.scan_scratch = cv.take<uint32_t>(2 * static_cast<size_t>(cdiv(e, SCAN_TILE)) + 16);
  h.total = cv.take<int64_t>(2);
  if (out) *out = h;
  return cv.off;
}

static unsigned cr_long_grid(int64_t K) {  // workgroups of cr_rows_long_kernel (each takes every grid-th listed row)
  return static_cast<unsigned>(K < 1024 ? (K > 0 ? K : 1) : 1024);
}

__global__ __launch_bounds__(256) void cr_table_kernel(const int64_t* __restrict__ cluster, int64_t n,
                                                       int32_t* __restrict__ table) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) table[i] = static_cast<int32_t>(cluster[i]);
}

}  // namespace tgp

using namespace tgp;

extern "C" size_t tgp_connect_coalesce_rows_workspace_bytes(int64_t E, int64_t N, int64_t K) {
  return cr_layout(nullptr, E, N, K, nullptr) + 256;
}
extern "C" size_t tgp_connect_coalesce_rows_workspace_bytes_f64(int64_t E, int64_t N, int64_t K) {
  return cr_layout(nullptr, E, N, K, nullptr, sizeof(double)) + 256;
}
// with TGP_HUGE_ROWS in the flags of tgp_connect_coalesce_rows_count: room for the device-wide sort of the hub rows
extern "C" size_t tgp_connect_coalesce_rows_huge_workspace_bytes(int64_t E, int64_t N, int64_t K) {
  return huge_layout(nullptr, align_up(cr_layout(nullptr, E, N, K, nullptr)), E, nullptr) + 256;
}

// ---------------------------------------------------------------------------------------------------------------------
// The host read of a count -> fill pair without a device-to-host copy and a stream synchronise: one thread stores
// {epoch << 34 | count (34-bit two's complement: decline codes are negative)} into a pinned host word the caller polls.
__global__ void count_publish_kernel(const int64_t* __restrict__ d_count, unsigned long long* __restrict__ result,
                                     unsigned long long tag) {
  __hip_atomic_store(result, tag | (static_cast<unsigned long long>(*d_count) & ((1ull << SPS_EPOCH_SHIFT) - 1)),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// a handed-over CSR covers the whole list: offsets start at 0 and end at E (a list with ids outside [0, N) does not)
// (reset: this one-thread launch also clears the status words, saving the memset in front of it)
static __global__ void cr_check_csr_kernel(const int32_t* __restrict__ csr_ptr, int64_t N, int64_t E,
                                           int* __restrict__ bad, int reset) {
  const bool broken = csr_ptr[0] != 0 || static_cast<int64_t>(csr_ptr[N]) != E;
  if (reset) {
    bad[0] = broken ? 4 : 0;
    bad[1] = 0;
    bad[2] = 0;
    bad[3] = 0;
  } else if (broken) {
    *bad = 4;
  }
}

struct CrPublish {  // the single-launch survivor scan with the count handed to the host through *result
  const int32_t* csr_col;  // NULL, or the int32 columns of this very list (only with csr_ptr)
  uint64_t* status;
  uint64_t* result;
  uint32_t epoch;
};

template <typename WT>
static int cr_rows_count_impl(const int64_t* row, const int64_t* col, const WT* w, int64_t E,
                              const int64_t* cluster_index, int64_t N, int64_t K, const int32_t* assign_row_ptr,
                              const int32_t* assign_perm, const int32_t* csr_ptr, int reduce_op, int flags, WT eps,
                              void* ws, size_t ws_bytes, int64_t* d_count, const CrPublish* pub, void* stream_) {
  constexpr bool F64 = sizeof(WT) == 8;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(E >= 0 && N >= 0 && K >= 0 && d_count, TGP_ERR_INVALID, "tgp_connect_coalesce_rows_count: bad argument");
  TGP_REQUIRE(E == 0 || (row && col && cluster_index && assign_row_ptr && assign_perm), TGP_ERR_INVALID,
              "tgp_connect_coalesce_rows_count: null pointer");
  TGP_REQUIRE(reduce_op >= TGP_SUM && reduce_op <= TGP_MUL, TGP_ERR_INVALID,
              "tgp_connect_coalesce_rows_count: unknown reduce_op %d", reduce_op);
  TGP_REQUIRE(E < (1ll << 31) && K < (1ll << 26) && N < (1ll << 31), TGP_ERR_RANGE,
              "tgp_connect_coalesce_rows_count: E/N >= 2^31 or K >= 2^26");
  const bool huge = (flags & TGP_HUGE_ROWS) != 0;
  TGP_REQUIRE(!(F64 && huge), TGP_ERR_INVALID,
              "tgp_connect_coalesce_rows_count_published_f64: TGP_HUGE_ROWS is float32 only (a list with hub rows answers "
              "-5 and the caller takes the general float64 route)");
  TGP_REQUIRE(ws && ws_bytes >= (F64 ? tgp_connect_coalesce_rows_workspace_bytes_f64(E, N, K)
                                 : huge ? tgp_connect_coalesce_rows_huge_workspace_bytes(E, N, K)
                                        : tgp_connect_coalesce_rows_workspace_bytes(E, N, K)),
              TGP_ERR_WORKSPACE, "tgp_connect_coalesce_rows_count: workspace too small");
  if (E == 0 || K == 0) {
    (void)hipMemsetAsync(d_count, 0, sizeof(int64_t), stream);
    if (pub)
      hipLaunchKernelGGL(count_publish_kernel, dim3(1), dim3(1), 0, stream, d_count,
                         reinterpret_cast<unsigned long long*>(pub->result),
                         static_cast<unsigned long long>(pub->epoch) << SPS_EPOCH_SHIFT);
    return check_launch("tgp_connect_coalesce_rows_count");
  }
  CrWs s;
  const size_t cr_end = cr_layout(ws, E, N, K, &s, sizeof(WT));
  HugeWs h{};
  if (huge) huge_layout(ws, align_up(cr_end), E, &h);
  WT* tmp_w = w ? reinterpret_cast<WT*>(s.tmp_w) : nullptr;
  if (!csr_ptr) (void)hipMemsetAsync(s.bad, 0, 4 * sizeof(int), stream);
  if (csr_ptr) {
    // CSR offsets of this very list from the caller (GraclusSelect builds them): no pass over the row array
    s.node_ptr = reinterpret_cast<uint32_t*>(const_cast<int32_t*>(csr_ptr));  // non-negative: same bits; read only
    if (!pub) hipLaunchKernelGGL(cr_check_csr_kernel, dim3(1), dim3(1), 0, stream, csr_ptr, N, E, s.bad, 1);
  } else if ((reinterpret_cast<uintptr_t>(row) & 15) == 0) {
    hipLaunchKernelGGL(cr_node_ptr_vec_kernel, dim3(cdiv(cdiv(E, 4) + 1, 256)), dim3(256), 0, stream, row, E, N, s.bad,
                       s.node_ptr);
  } else {
    hipLaunchKernelGGL(cr_node_ptr_kernel, dim3(cdiv(E + 1, 256)), dim3(256), 0, stream, row, E, N, s.bad, s.node_ptr);
  }
  const int nt = cdiv(N, MS_TILE);
  if (pub) {  // member segments in one launch: look-back over the second region of the status buffer
    unsigned long long* st2 = reinterpret_cast<unsigned long long*>(pub->status) + 2 + cdiv(K > 0 ? K : 1, SCAN_TILE);
    hipLaunchKernelGGL(cr_member_single_kernel, dim3(nt), dim3(256), 0, stream, assign_perm, N, s.node_ptr,
                       cluster_index, N, E, csr_ptr ? 1 : 0, s.bad, s.table, s.seg_src, s.seg_dst, st2,
                       static_cast<unsigned long long>(pub->epoch) << SPS_EPOCH_SHIFT, kLookbackTicket);
    hipLaunchKernelGGL(cr_raw_off_kernel, dim3(cdiv(K, 256)), dim3(256), 0, stream, assign_row_ptr, K, s.seg_dst, s.bad,
                       s.raw_off, huge ? h.list : static_cast<uint32_t*>(nullptr), s.n_out, st2,
                       static_cast<unsigned long long>(pub->epoch) << SPS_EPOCH_SHIFT);
  } else {  // member segments: degree sums per tile (+ cluster table), scan, row offsets
    uint32_t* sums = s.scan_scratch;
    uint32_t* offs = s.scan_scratch + nt;
    hipLaunchKernelGGL(cr_member_sums_kernel, dim3(nt), dim3(256), 0, stream, assign_perm, N, s.node_ptr, cluster_index,
                       N, s.bad, s.table, sums);
    if (nt <= SCAN_SELF_TILES) {
      hipLaunchKernelGGL(cr_member_scan_kernel, dim3(nt), dim3(256), 0, stream, assign_perm, N, s.node_ptr, sums, 1, s.bad,
                         s.seg_src, s.seg_dst);
    } else {
      hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, stream, sums, nt, offs, s.total,
                         static_cast<const int*>(nullptr));
      hipLaunchKernelGGL(cr_member_scan_kernel, dim3(nt), dim3(256), 0, stream, assign_perm, N, s.node_ptr, offs, 0, s.bad,
                         s.seg_src, s.seg_dst);
    }
    hipLaunchKernelGGL(cr_raw_off_kernel, dim3(cdiv(K, 256)), dim3(256), 0, stream, assign_row_ptr, K, s.seg_dst, s.bad,
                       s.raw_off, huge ? h.list : static_cast<uint32_t*>(nullptr), s.n_out,
                       static_cast<const unsigned long long*>(nullptr), 0ull);
  }
  if constexpr (!F64)
  if (huge) {  // rows beyond CR_LONG entries (hubs): their entries alone are sorted device-wide (see cr_huge_*_kernel)
    const int colbits = bits_for(static_cast<uint64_t>(K - 1));
    const uint32_t* n_dev = reinterpret_cast<const uint32_t*>(s.bad + 3);
    hipLaunchKernelGGL(cr_huge_offsets_kernel, dim3(1), dim3(1024), 0, stream, h.list, s.raw_off, K, E, s.bad, h.hoff);
    hipLaunchKernelGGL(cr_huge_gather_kernel, dim3(1024), dim3(256), 0, stream, col, w, s.table, assign_row_ptr, s.seg_src,
                       s.seg_dst, s.raw_off, h.list, h.hoff, N, colbits, s.bad, h.k0, h.v0, h.sub_w);
    bool first = true;
    const int rc = radix_sort_pairs<uint64_t, uint32_t>(h.k0, h.v0, h.k1, h.v1, E, colbits + 12, h.sort_scratch, stream,
                                                        &first, n_dev);
    if (rc != TGP_OK) return rc;
    const uint64_t* ks = first ? h.k0 : h.k1;
    const uint32_t* vs = first ? h.v0 : h.v1;
    hipLaunchKernelGGL(cr_huge_merge_kernel, dim3(cdiv(E, 256)), dim3(256), 0, stream, ks, vs, h.sub_w, w ? 1 : 0, h.list,
                       colbits, reduce_op, flags, eps, s.bad, E, h.flags, h.mval);
    device_scan_u32(h.flags, E, h.ranks, h.total, h.scan_scratch, stream);
    hipLaunchKernelGGL(cr_huge_write_kernel, dim3(cdiv(E, 256)), dim3(256), 0, stream, ks, h.flags, h.ranks, h.mval,
                       h.list, h.hoff, s.raw_off, colbits, s.bad, h.total, E, s.tmp_c, tmp_w, s.n_out);
  }
  // (measured r2: splitting the gather into an edge-parallel permute pass + the DIRECT sort kernel costs 114 + 103 us
  //  against 185 us for the fused gather: the 10 M random 4-byte table look-ups take ~50 us wherever they run)
  if (pub && pub->csr_col && csr_ptr)
    hipLaunchKernelGGL((cr_gather_sort_kernel<false, int32_t, WT>), dim3(cdiv(K, GS_ROWS)), dim3(256), 0, stream, pub->csr_col,
                       w, E, s.table, assign_row_ptr, s.seg_src, s.seg_dst,
                       static_cast<const unsigned long long*>(nullptr), s.raw_off, K, reduce_op, flags, eps, s.bad,
                       s.tmp_c, tmp_w, s.n_out, N, s.long_list);
  else
    hipLaunchKernelGGL((cr_gather_sort_kernel<false, int64_t, WT>), dim3(cdiv(K, GS_ROWS)), dim3(256), 0, stream, col, w, E,
                       s.table, assign_row_ptr, s.seg_src, s.seg_dst, static_cast<const unsigned long long*>(nullptr),
                       s.raw_off, K, reduce_op, flags, eps, s.bad, s.tmp_c, tmp_w, s.n_out, N, s.long_list);
  hipLaunchKernelGGL(cr_rows_long_kernel<WT>, dim3(cr_long_grid(K)), dim3(256), 0, stream, s.tmp_c, tmp_w, s.raw_off, K, E,
                     reduce_op, flags, eps, s.bad, s.long_list, s.n_out);
  if (pub) {
    hipLaunchKernelGGL(cr_scan_publish_kernel, dim3(cdiv(K, SCAN_TILE)), dim3(256), 0, stream, s.n_out, K, s.out_off,
                       s.total, s.bad, d_count, reinterpret_cast<unsigned long long*>(pub->status),
                       reinterpret_cast<unsigned long long*>(pub->result),
                       static_cast<unsigned long long>(pub->epoch) << SPS_EPOCH_SHIFT, kLookbackTicket);
    return check_launch("tgp_connect_coalesce_rows_count_published");
  }
  device_scan_u32(s.n_out, K, s.out_off, s.total, s.scan_scratch, stream, s.bad, d_count);
  return check_launch("tgp_connect_coalesce_rows_count");
}

extern "C" int tgp_connect_coalesce_rows_count(const int64_t* row, const int64_t* col, const float* w, int64_t E,
                                               const int64_t* cluster_index, int64_t N, int64_t K,
                                               const int32_t* assign_row_ptr, const int32_t* assign_perm,
                                               const int32_t* csr_ptr, int reduce_op, int flags, float eps, void* ws,
                                               size_t ws_bytes, int64_t* d_count, void* stream_) {
  return cr_rows_count_impl<float>(row, col, w, E, cluster_index, N, K, assign_row_ptr, assign_perm, csr_ptr, reduce_op,
                                   flags, eps, ws, ws_bytes, d_count, nullptr, stream_);
}

// The same pipeline with the survivor scan as one launch (decoupled look-back over `status`, epoch-tagged: caller-owned,
// never cleared) that also stores {epoch << 34 | count} into *result -- pinned host memory the caller polls instead of
// copying *d_count back; `csr_col`: the int32 columns GraclusSelect's CSR holds for this very list (with csr_ptr).
// [2 + survivor-scan tiles] then [2 + member tiles]: the two look-backs of a call share the buffer, not their words
extern "C" int64_t tgp_connect_coalesce_rows_count_status_words(int64_t K, int64_t N) {
  return 4 + cdiv(K > 0 ? K : 1, SCAN_TILE) + cdiv(N > 0 ? N : 1, MS_TILE);
}

extern "C" int tgp_connect_coalesce_rows_count_published(const int64_t* row, const int64_t* col, const int32_t* csr_col,
                                                         const float* w, int64_t E, const int64_t* cluster_index,
                                                         int64_t N, int64_t K, const int32_t* assign_row_ptr,
                                                         const int32_t* assign_perm, const int32_t* csr_ptr,
                                                         int reduce_op, int flags, float eps, void* ws, size_t ws_bytes,
                                                         int64_t* d_count, uint64_t* status, int64_t status_words,
                                                         uint64_t* result, uint32_t epoch, void* stream_) {
  TGP_REQUIRE(status && result && d_count, TGP_ERR_INVALID, "tgp_connect_coalesce_rows_count_published: bad argument");
  TGP_REQUIRE(status_words >= tgp_connect_coalesce_rows_count_status_words(K, N), TGP_ERR_WORKSPACE,
              "tgp_connect_coalesce_rows_count_published: status buffer too small");
  TGP_REQUIRE(epoch != 0 && epoch < (1u << 29), TGP_ERR_RANGE,
              "tgp_connect_coalesce_rows_count_published: epoch out of range");
  TGP_REQUIRE(!csr_col || csr_ptr, TGP_ERR_INVALID,
              "tgp_connect_coalesce_rows_count_published: int32 columns without the offsets of the same list");
  const CrPublish pub{csr_col, status, result, epoch};
  return cr_rows_count_impl<float>(row, col, w, E, cluster_index, N, K, assign_row_ptr, assign_perm, csr_ptr, reduce_op,
                                   flags, eps, ws, ws_bytes, d_count, &pub, stream_);
}

// float64 edge weights on the same row-local pipeline (r5: they took the general sort-based route, 0.6 ms against 0.24
// at C4): values staged, merged (sum / mean / min / max / mul) and eps-filtered in double, as the reference's coalesce does
// for a double tensor (connect/base_conn.py:86-89).  Workspace: tgp_connect_coalesce_rows_workspace_bytes_f64; hub rows
// (TGP_HUGE_ROWS) are float32 only: such a list answers -5 here and the caller takes tgp_connect_coalesce_count_f64.
extern "C" int tgp_connect_coalesce_rows_count_published_f64(
    const int64_t* row, const int64_t* col, const int32_t* csr_col, const double* w, int64_t E,
    const int64_t* cluster_index, int64_t N, int64_t K, const int32_t* assign_row_ptr, const int32_t* assign_perm,
    const int32_t* csr_ptr, int reduce_op, int flags, double eps, void* ws, size_t ws_bytes, int64_t* d_count,
    uint64_t* status, int64_t status_words, uint64_t* result, uint32_t epoch, void* stream_) {
  TGP_REQUIRE(status && result && d_count && w, TGP_ERR_INVALID,
              "tgp_connect_coalesce_rows_count_published_f64: bad argument");
  TGP_REQUIRE(status_words >= tgp_connect_coalesce_rows_count_status_words(K, N), TGP_ERR_WORKSPACE,
              "tgp_connect_coalesce_rows_count_published_f64: status buffer too small");
  TGP_REQUIRE(epoch != 0 && epoch < (1u << 29), TGP_ERR_RANGE,
              "tgp_connect_coalesce_rows_count_published_f64: epoch out of range");
  TGP_REQUIRE(!csr_col || csr_ptr, TGP_ERR_INVALID,
              "tgp_connect_coalesce_rows_count_published_f64: int32 columns without the offsets of the same list");
  const CrPublish pub{csr_col, status, result, epoch};
  return cr_rows_count_impl<double>(row, col, w, E, cluster_index, N, K, assign_row_ptr, assign_perm, csr_ptr, reduce_op,
                                    flags, eps, ws, ws_bytes, d_count, &pub, stream_);
}

// ------------------------------------------------------------------ fused row-sorted path (r3), see cr_fused_kernel
extern "C" size_t tgp_connect_coalesce_fused_workspace_bytes(int64_t E, int64_t N, int64_t K) {
  return fz_layout(nullptr, E, N, K, nullptr) + 256;
}

extern "C" int tgp_connect_coalesce_fused_count(const int64_t* row, const int64_t* col, const int32_t* csr_ptr,
                                                const int32_t* csr_col, const float* w, int64_t E,
                                                const int64_t* cluster_index, int64_t N, int64_t K,
                                                const int32_t* assign_row_ptr, const int32_t* assign_perm,
                                                int reduce_op, int flags, float eps, float* out_w, void* ws,
                                                size_t ws_bytes, int64_t* d_count, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const bool csr = csr_ptr && csr_col;  // int32 columns are only usable with the offsets of the same list
  TGP_REQUIRE(E >= 0 && N >= 0 && K >= 0 && d_count, TGP_ERR_INVALID, "tgp_connect_coalesce_fused_count: bad argument");
  TGP_REQUIRE(E == 0 || (((csr_ptr || row) && (col || csr)) && cluster_index && assign_row_ptr && assign_perm),
              TGP_ERR_INVALID, "tgp_connect_coalesce_fused_count: null pointer");
  TGP_REQUIRE(!w || out_w || E == 0, TGP_ERR_INVALID, "tgp_connect_coalesce_fused_count: weights without an output buffer");
  TGP_REQUIRE(reduce_op >= TGP_SUM && reduce_op <= TGP_MUL, TGP_ERR_INVALID,
              "tgp_connect_coalesce_fused_count: unknown reduce_op %d", reduce_op);
  TGP_REQUIRE(E < (1ll << 31) && K < (1ll << 26) && N < (1ll << 31), TGP_ERR_RANGE,
              "tgp_connect_coalesce_fused_count: E/N >= 2^31 or K >= 2^26");
  TGP_REQUIRE(ws && ws_bytes >= tgp_connect_coalesce_fused_workspace_bytes(E, N, K), TGP_ERR_WORKSPACE,
              "tgp_connect_coalesce_fused_count: workspace too small");
  if (E == 0 || K == 0) {
    (void)hipMemsetAsync(d_count, 0, sizeof(int64_t), stream);
    return check_launch("tgp_connect_coalesce_fused_count");
  }
  FzWs s;
  fz_layout(ws, E, N, K, &s);
  (void)hipMemsetAsync(s.status, 0, s.zero_bytes, stream);
  const uint32_t* node_ptr = s.node_ptr;
  if (csr_ptr) {
    node_ptr = reinterpret_cast<const uint32_t*>(csr_ptr);  // offsets are non-negative: same bits
    hipLaunchKernelGGL(cr_check_csr_kernel, dim3(1), dim3(1), 0, stream, csr_ptr, N, E, s.bad, 0);
  } else if ((reinterpret_cast<uintptr_t>(row) & 15) == 0) {
    hipLaunchKernelGGL(cr_node_ptr_vec_kernel, dim3(cdiv(cdiv(E, 4) + 1, 256)), dim3(256), 0, stream, row, E, N, s.bad,
                       s.node_ptr);
  } else {
    hipLaunchKernelGGL(cr_node_ptr_kernel, dim3(cdiv(E + 1, 256)), dim3(256), 0, stream, row, E, N, s.bad, s.node_ptr);
  }
  hipLaunchKernelGGL(cr_table_kernel, dim3(cdiv(N, 256)), dim3(256), 0, stream, cluster_index, N, s.table);
  // rows per tile: ~100 members on average (the kernel takes at most 256), at most 32 rows
  int64_t rows = N > 0 ? 100 * K / N : GS_ROWS;
  rows = rows < 1 ? 1 : (rows > GS_ROWS ? GS_ROWS : rows);
  const unsigned tiles = static_cast<unsigned>(cdiv(K, rows));
  if (tiles > static_cast<unsigned>(FZ_MAX_TILES)) {  // more tiles than one launch wave holds: the staged pipeline
    hipLaunchKernelGGL(fz_decline_kernel, dim3(1), dim3(1), 0, stream, d_count);
    return check_launch("tgp_connect_coalesce_fused_count");
  }
  if (csr)
    hipLaunchKernelGGL(cr_fused_kernel<int32_t>, dim3(tiles), dim3(256), 0, stream, csr_col, w, E, s.table,
                       assign_row_ptr, assign_perm, node_ptr, K, static_cast<int>(rows), reduce_op, flags, eps, s.bad,
                       s.status, s.tmp_c, w ? out_w : nullptr, s.out_off, s.total, N);
  else
    hipLaunchKernelGGL(cr_fused_kernel<int64_t>, dim3(tiles), dim3(256), 0, stream, col, w, E, s.table,
                       assign_row_ptr, assign_perm, node_ptr, K, static_cast<int>(rows), reduce_op, flags, eps, s.bad,
                       s.status, s.tmp_c, w ? out_w : nullptr, s.out_off, s.total, N);
  hipLaunchKernelGGL(fz_finish_kernel, dim3(1), dim3(1), 0, stream, s.bad, s.total, d_count);
  return check_launch("tgp_connect_coalesce_fused_count");
}

extern "C" int tgp_connect_coalesce_fused_fill(const void* ws, int64_t E, int64_t N, int64_t K, int64_t num_out,
                                               int64_t* out_row, int64_t* out_col, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(ws && num_out >= 0, TGP_ERR_INVALID, "tgp_connect_coalesce_fused_fill: bad argument");
  if (num_out == 0 || E == 0 || K == 0) return TGP_OK;
  TGP_REQUIRE(out_row && out_col, TGP_ERR_INVALID, "tgp_connect_coalesce_fused_fill: null output");
  FzWs s;
  fz_layout(const_cast<void*>(ws), E, N, K, &s);
  hipLaunchKernelGGL(fz_fill_kernel, dim3(cdiv(K, FILL_ROWS)), dim3(256), 0, stream, s.tmp_c, s.out_off, K, out_row,
                     out_col);
  return check_launch("tgp_connect_coalesce_fused_fill");
}

// ------------------------------------------------------------------ grouped path (any edge order)
// Two-level sort for inputs whose rows are NOT sorted: a stable LSD radix sort by SUPERNODE ROW only (log2 K bits:
// 3 passes for K = 550 000, where the (row, col) key of the general path needs 5), carrying (cluster column,
// weight) as one 64-bit payload, then the in-row sort / merge of the row-local path above.  Stable => duplicates
// keep their input order, so the result equals the other two paths bit for bit.
namespace tgp {
__global__ __launch_bounds__(256) void cg_keys_kernel(const int64_t* __restrict__ row, const int64_t* __restrict__ col,
                                                      const float* __restrict__ w, const int32_t* __restrict__ table,
                                                      int64_t E, int64_t n_nodes, int64_t K, int* __restrict__ bad,
                                                      uint32_t* __restrict__ keys,
                                                      unsigned long long* __restrict__ vals) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (e >= E) return;
  const int64_t r = row[e], c = col[e];
  uint32_t kr = 0, kc = 0;
  if (static_cast<uint64_t>(r) >= static_cast<uint64_t>(n_nodes) || static_cast<uint64_t>(c) >= static_cast<uint64_t>(n_nodes)) {
    *bad = 4;  // node id outside [0, n): decline, the general path reports it
  } else {
    kr = static_cast<uint32_t>(table[r]);
    kc = static_cast<uint32_t>(table[c]);
    if (kr >= static_cast<uint64_t>(K) || kc >= static_cast<uint64_t>(K)) { *bad = 4; kr = kc = 0; }
  }
  keys[e] = kr;
  vals[e] = static_cast<unsigned long long>(kc) |
            (static_cast<unsigned long long>(__float_as_uint(w ? w[e] : 1.0f)) << 32);
}

// Four consecutive edges per thread: row / col / w arrive as 16-byte loads, the eight table look-ups of a thread are
// in flight together, keys and payloads leave as 16-byte stores (all arrays 16-byte aligned: the host checks).
__global__ __launch_bounds__(256) void cg_keys4_kernel(const int64_t* __restrict__ row, const int64_t* __restrict__ col,
                                                       const float* __restrict__ w, const int32_t* __restrict__ table,
                                                       int64_t groups, int64_t n_nodes, int64_t K,
                                                       int* __restrict__ bad, uint32_t* __restrict__ keys,
                                                       unsigned long long* __restrict__ vals) {
  const int64_t gi = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (gi >= groups) return;
  const int64_t e = gi * 4;
  typedef long long ll2 __attribute__((ext_vector_type(2)));
  typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
  const ll2 r01 = __builtin_nontemporal_load(reinterpret_cast<const ll2*>(row + e));
  const ll2 r23 = __builtin_nontemporal_load(reinterpret_cast<const ll2*>(row + e + 2));
  const ll2 c01 = __builtin_nontemporal_load(reinterpret_cast<const ll2*>(col + e));
  const ll2 c23 = __builtin_nontemporal_load(reinterpret_cast<const ll2*>(col + e + 2));
  float4 wv = make_float4(1.f, 1.f, 1.f, 1.f);
  if (w) wv = *reinterpret_cast<const float4*>(w + e);
  const int64_t r[4] = {r01.x, r01.y, r23.x, r23.y}, c[4] = {c01.x, c01.y, c23.x, c23.y};
  const float ww[4] = {wv.x, wv.y, wv.z, wv.w};
  uint32_t kr[4], kc[4];
  bool ok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    ok[j] = static_cast<uint64_t>(r[j]) < static_cast<uint64_t>(n_nodes) &&
            static_cast<uint64_t>(c[j]) < static_cast<uint64_t>(n_nodes);
    kr[j] = static_cast<uint32_t>(table[ok[j] ? r[j] : 0]);
    kc[j] = static_cast<uint32_t>(table[ok[j] ? c[j] : 0]);
  }
  unsigned long long v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (!ok[j] || kr[j] >= static_cast<uint64_t>(K) || kc[j] >= static_cast<uint64_t>(K)) {
      *bad = 4;  // node / cluster id out of range: decline, the general path reports it
      kr[j] = kc[j] = 0;
    }
    v[j] = static_cast<unsigned long long>(kc[j]) | (static_cast<unsigned long long>(__float_as_uint(ww[j])) << 32);
  }
  *reinterpret_cast<uint4*>(keys + e) = make_uint4(kr[0], kr[1], kr[2], kr[3]);
  ull2 v01 = {v[0], v[1]}, v23 = {v[2], v[3]};
  *reinterpret_cast<ull2*>(vals + e) = v01;
  *reinterpret_cast<ull2*>(vals + e + 2) = v23;
}

// first slot of every supernode row from the sorted keys (rows without edges get the next row's slot)
__global__ __launch_bounds__(256) void cg_row_off_kernel(const uint32_t* __restrict__ keys, int64_t E, int64_t K,
                                                         uint32_t* __restrict__ raw_off) {
  const int64_t p = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (p > E) return;
  if (p == E) {
    for (int64_t c = E > 0 ? static_cast<int64_t>(keys[E - 1]) + 1 : 0; c < K; ++c) raw_off[c] = static_cast<uint32_t>(E);
    return;
  }
  const int64_t cur = keys[p], prev = p > 0 ? static_cast<int64_t>(keys[p - 1]) : -1;
  for (int64_t c = prev + 1; c <= cur; ++c) raw_off[c] = static_cast<uint32_t>(p);
}

struct CgWs {
  uint32_t *k0, *k1;
  unsigned long long *v0, *v1;
  uint32_t* scratch;
};
static size_t cg_layout(void* ws, int64_t E, int64_t N, int64_t K, CrWs* cr, CgWs* cg) {
  const size_t head = cr_layout(ws, E, N, K, cr);  // same prefix as the row-local path: its fill entry is reused
  Carver cv(ws);
  cv.off = align_up(head);
  const size_t e = static_cast<size_t>(E > 0 ? E : 1);
  CgWs g;
  g.v0 = cv.take<unsigned long long>(e);
  g.v1 = cv.take<unsigned long long>(e);
  g.k0 = cv.take<uint32_t>(e);
  g.k1 = cv.take<uint32_t>(e);
  g.scratch = cv.take<uint32_t>(sort_scratch_words());
  if (cg) *cg = g;
  return cv.off;
}
}  // namespace tgp

extern "C" size_t tgp_connect_coalesce_grouped_workspace_bytes(int64_t E, int64_t N, int64_t K) {
  return cg_layout(nullptr, E, N, K, nullptr, nullptr) + 256;
}

// Counting half; the fill half is tgp_connect_coalesce_rows_fill with the same workspace.
extern "C" int tgp_connect_coalesce_grouped_count(const int64_t* row, const int64_t* col, const float* w, int64_t E,
                                                  const int64_t* cluster_index, int64_t N, int64_t K, int reduce_op,
                                                  int flags, float eps, void* ws, size_t ws_bytes, int64_t* d_count,
                                                  void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(E >= 0 && N >= 0 && K >= 0 && d_count, TGP_ERR_INVALID, "tgp_connect_coalesce_grouped_count: bad argument");
  TGP_REQUIRE(E == 0 || (row && col && cluster_index), TGP_ERR_INVALID,
              "tgp_connect_coalesce_grouped_count: null pointer");
  TGP_REQUIRE(reduce_op >= TGP_SUM && reduce_op <= TGP_MUL, TGP_ERR_INVALID,
              "tgp_connect_coalesce_grouped_count: unknown reduce_op %d", reduce_op);
  TGP_REQUIRE(E < (1ll << 31) && K < (1ll << 26) && N < (1ll << 31), TGP_ERR_RANGE,
              "tgp_connect_coalesce_grouped_count: E/N >= 2^31 or K >= 2^26");
  TGP_REQUIRE(ws && ws_bytes >= tgp_connect_coalesce_grouped_workspace_bytes(E, N, K), TGP_ERR_WORKSPACE,
              "tgp_connect_coalesce_grouped_count: workspace too small");
  if (E == 0 || K == 0) {
    (void)hipMemsetAsync(d_count, 0, sizeof(int64_t), stream);
    return check_launch("tgp_connect_coalesce_grouped_count");
  }
  CrWs s;
  CgWs g;
  cg_layout(ws, E, N, K, &s, &g);
  float* tmp_w = w ? s.tmp_w : nullptr;
  (void)hipMemsetAsync(s.bad, 0, 2 * sizeof(int), stream);
  hipLaunchKernelGGL(cr_table_kernel, dim3(cdiv(N, 256)), dim3(256), 0, stream, cluster_index, N, s.table);
  {
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const int64_t groups = (al16(row) && al16(col) && (!w || al16(w))) ? E / 4 : 0;
    if (groups > 0)
      hipLaunchKernelGGL(cg_keys4_kernel, dim3(cdiv(groups, 256)), dim3(256), 0, stream, row, col, w, s.table, groups, N,
                         K, s.bad, g.k0, g.v0);
    const int64_t done = groups * 4;
    if (done < E)
      hipLaunchKernelGGL(cg_keys_kernel, dim3(cdiv(E - done, 256)), dim3(256), 0, stream, row + done, col + done,
                         w ? w + done : nullptr, s.table, E - done, N, K, s.bad, g.k0 + done, g.v0 + done);
  }
  bool first = true;
  const int rc = radix_sort_pairs<uint32_t, unsigned long long>(g.k0, g.v0, g.k1, g.v1, E,
                                                                bits_for(static_cast<uint64_t>(K - 1)), g.scratch,
                                                                stream, &first);
  if (rc != TGP_OK) return rc;
  const uint32_t* keys = first ? g.k0 : g.k1;
  const unsigned long long* vals = first ? g.v0 : g.v1;
  hipLaunchKernelGGL(cg_row_off_kernel, dim3(cdiv(E + 1, 256)), dim3(256), 0, stream, keys, E, K, s.raw_off);
  hipLaunchKernelGGL(cr_gather_sort_kernel<true>, dim3(cdiv(K, GS_ROWS)), dim3(256), 0, stream,
                     static_cast<const int64_t*>(nullptr), static_cast<const float*>(nullptr), E,
                     static_cast<const int32_t*>(nullptr), static_cast<const int32_t*>(nullptr),
                     static_cast<const uint32_t*>(nullptr), static_cast<const uint32_t*>(nullptr), vals, s.raw_off, K,
                     reduce_op, flags, eps, s.bad, s.tmp_c, tmp_w, s.n_out, N, s.long_list);
  hipLaunchKernelGGL(cr_rows_long_kernel<float>, dim3(cr_long_grid(K)), dim3(256), 0, stream, s.tmp_c, tmp_w, s.raw_off, K, E,
                     reduce_op, flags, eps, s.bad, s.long_list, s.n_out);
  device_scan_u32(s.n_out, K, s.out_off, s.total, s.scan_scratch, stream, s.bad, d_count);
  return check_launch("tgp_connect_coalesce_grouped_count");
}

extern "C" int tgp_connect_coalesce_rows_fill(const void* ws, int64_t E, int64_t N, int64_t K, int has_weight,
                                              int64_t num_out, int64_t* out_row, int64_t* out_col, float* out_w,
                                              void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(ws && num_out >= 0, TGP_ERR_INVALID, "tgp_connect_coalesce_rows_fill: bad argument");
  if (num_out == 0 || E == 0 || K == 0) return TGP_OK;
  TGP_REQUIRE(out_row && out_col && (!(has_weight & 1) || out_w), TGP_ERR_INVALID,
              "tgp_connect_coalesce_rows_fill: null output");
  CrWs s;
  const size_t cr_end = cr_layout(const_cast<void*>(ws), E, N, K, &s);
  const bool weights = (has_weight & 1) != 0;
  hipLaunchKernelGGL(cr_fill_kernel<float>, dim3(cdiv(K, FILL_ROWS)), dim3(256), 0, stream, s.tmp_c,
                     weights ? s.tmp_w : nullptr, s.raw_off, s.out_off, s.total, K, static_cast<const int*>(nullptr),
                     out_row, out_col, weights ? out_w : nullptr);
  if (has_weight & 2) {  // the count call ran with TGP_HUGE_ROWS: hub rows are copied by the whole grid
    HugeWs h;
    huge_layout(const_cast<void*>(ws), align_up(cr_end), E, &h);
    hipLaunchKernelGGL(cr_fill_huge_kernel, dim3(1024), dim3(256), 0, stream, s.tmp_c, weights ? s.tmp_w : nullptr,
                       s.raw_off, s.out_off, s.n_out, K, h.list, s.bad, out_row, out_col, weights ? out_w : nullptr);
  }
  return check_launch("tgp_connect_coalesce_rows_fill");
}

extern "C" int tgp_connect_coalesce_rows_fill_f64(const void* ws, int64_t E, int64_t N, int64_t K, int64_t num_out,
                                                  int64_t* out_row, int64_t* out_col, double* out_w, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(ws && num_out >= 0, TGP_ERR_INVALID, "tgp_connect_coalesce_rows_fill_f64: bad argument");
  if (num_out == 0 || E == 0 || K == 0) return TGP_OK;
  TGP_REQUIRE(out_row && out_col && out_w, TGP_ERR_INVALID, "tgp_connect_coalesce_rows_fill_f64: null output");
  CrWs s;
  cr_layout(const_cast<void*>(ws), E, N, K, &s, sizeof(double));
  hipLaunchKernelGGL(cr_fill_kernel<double>, dim3(cdiv(K, FILL_ROWS)), dim3(256), 0, stream, s.tmp_c,
                     reinterpret_cast<const double*>(s.tmp_w), s.raw_off, s.out_off, s.total, K,
                     static_cast<const int*>(nullptr), out_row, out_col, out_w);
  return check_launch("tgp_connect_coalesce_rows_fill_f64");
}

#ifdef TGP_GEMM_STAMPS
extern "C" int tgp_debug_set_gs_ablate(int mode) {
  return hipMemcpyToSymbol(HIP_SYMBOL(tgp::g_gs_ablate), &mode, sizeof(mode)) == hipSuccess ? 0 : -3;
}
extern "C" int tgp_debug_set_gs_stamps(unsigned long long* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(tgp::g_gs_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -3;
}
#endif

extern "C" int tgp_count_publish(const int64_t* d_count, uint64_t* result, uint32_t epoch, void* stream_) {
  TGP_REQUIRE(d_count && result && epoch > 0 && epoch < (1u << 30), TGP_ERR_INVALID, "tgp_count_publish: bad argument");
  hipLaunchKernelGGL(count_publish_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream_), d_count,
                     reinterpret_cast<unsigned long long*>(result),
                     static_cast<unsigned long long>(epoch) << SPS_EPOCH_SHIFT);
  return check_launch("tgp_count_publish");
}
