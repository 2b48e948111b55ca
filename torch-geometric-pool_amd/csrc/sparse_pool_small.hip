// r4: sparse Reduce + Connect of a BATCH OF SMALL GRAPHS in ONE launch -- the sparse twin of dense_pool_small_kernel.
//
// The staged operators (sparse_reduce.hip, sparse_connect.hip, coalesce_rows.hip) are built for one large graph:
// device-wide bitmaps, sorts and scans, six to ten launches per Reduce + Connect.  On 2048 PROTEINS-sized graphs
// (80 k nodes, 300 k edges) each of those launches moves a few megabytes and the call is bound by their dispatch
// latency (0.093 ms, 0.026 of the HBM roofline).  Here ONE WAVE owns one graph of a sorted batch (<= 64 nodes):
//   * its node range comes from the batch offsets.  The graphs of a workgroup are consecutive, so ONE wave finds the
//     workgroup's edge range and its slice of the (node-sorted) assignment with 128-ary searches (three dependent rounds;
//     r4 stamps: every wave searching for itself kept the CU's address unit busy for 8 us with stride-N probes), and
//     the per-graph boundaries inside come from one coalesced pass over the workgroup's rows;
//   * A1 + A2 (reduce/base_reduce.py:14-53,141-155): the graph's pooled rows are gathered, scaled, summed and
//     written by the wave (products rounded before the add, members in ascending order: the bits of
//     reduce_sparse_*_kernel and of the reference's sequential scatter);
//   * A5 + A6 (connect/base_conn.py:79-82, utils/ops.py:370-380; TopK): membership = a 64-bit mask in registers, new id
//     = first assignment of the graph + popcount below; survivors keep input order;
//   * A4 + A6 (connect/base_conn.py:83-89; Graclus-style clusterings): the graph's edges are staged in LDS with their
//     column already mapped to its cluster; LANE = supernode ROW walks the edge ranges of its member nodes in input
//     order.  A graph has at most 64 clusters, so a row's set of columns is a 64-bit mask: the slot of column c in the
//     sorted row is a popcount below c -- no sorting network, no searching; weights of duplicates are folded into
//     their slot in input order; filters fused; rows leave in (row, column) order = PyG coalesce's;
//   * the only cross-wave quantity, the number of surviving edges in front of a graph, comes from a workgroup sum and a
//     decoupled look-back over the workgroups (WAVES graphs each: 128 tiles for 2048 graphs, two hops).  Survivors are
//     written ONCE, in their final int64 form at their final offsets of capacity-E buffers; the host reads the total
//     (the one sync the reference's own .item() pays) and narrows the buffers: no fill launch.
// Every structural assumption is CHECKED on the device (graph sizes, ranges that tile the arrays, edges that stay
// inside their graph, ascending assignment / rows, cluster ids contiguous per graph): a violation raises a status word
// and the caller takes the staged operators, so no result depends on an unchecked property of the input.
//
// The look-back state lives in a caller-owned status buffer whose words carry the call's EPOCH: stale words of
// earlier calls read as "not ready", so the buffer is never cleared (no memset launch in front of the kernel).  A
// refusal travels INSIDE those words (bit 31 of the value: a tile that saw a violation, or whose predecessors did, says
// so in what it publishes), so the last tile knows the verdict of the whole call and leaves ONE result word
// {epoch, refused, total}, stored with system scope: the caller may point it at pinned host memory and poll it instead
// of paying a device-to-host copy kernel and a stream synchronise for eight bytes.
#include "lookback.h"

namespace tgp {

constexpr int SPS_CAP = 512;  // cluster mode: edges of one graph staged in LDS

struct SpsArgs {
  const float* x;
  int64_t N, F, x_stride;
  const int64_t* gptr;
  int64_t B;
  const int64_t* row;
  const int64_t* col;
  const float* w;
  int64_t E;
  const int64_t* node_index;
  const int64_t* cluster_index;
  const float* weight;
  int64_t nnz, K;
  int reduce_op, flags;
  float eps;
  float* x_pool;
  int64_t* batch_pool;
  int64_t* out_row;
  int64_t* out_col;
  float* out_w;
  const int64_t* edge_ptr;     // NULL, or [B + 1]: first edge of every graph (lower bounds of gptr in `row`)
  const int64_t* assign_ptr;   // NULL, or [B + 1] (MODE 0): first assignment of every graph
  int64_t* edge_ptr_out;       // NULL, or [B + 1]: r5 -- a call that searched for its edge ranges leaves them here, so
                               // that the caller can hand them back (edge_ptr) when the same list is pooled again
  unsigned long long* status;  // [2 + tile] look-back state, epoch-tagged ([0], [1] reserved)
  unsigned long long* result;  // ONE word {epoch, refused (bit 31), total}; may live in pinned host memory
  unsigned long long tag;      // epoch << SPS_EPOCH_SHIFT
  int ticket;                  // tile = arrival number (status[1]) instead of blockIdx.x (TGP_LOOKBACK_TICKET)
};

__device__ __forceinline__ float sps_reduce(float acc, float v, int op) {
  switch (op) {
    case TGP_MIN: return fminf(acc, v);
    case TGP_MAX: return fmaxf(acc, v);
    case TGP_MUL: return __fmul_rn(acc, v);
    default: return __fadd_rn(acc, v);
  }
}

// per-wave LDS of the cluster mode
struct SpsClusterLds {
  unsigned long long mem[64];  // members (local node bits) of local cluster r
  uint32_t deg[64];            // edges per local node, then (first staged edge << 16 | edges)
  uint8_t ecc[SPS_CAP];        // staged edges: local cluster of the column
  float ew[SPS_CAP];           // staged edges: weight
  uint16_t lcnt[SPS_CAP];      // row slots: merged entries (mean)
  float lval[SPS_CAP];         // row slots: merged weight
};

// diagnostic build (make stamps): wave 0 of every workgroup leaves the constant-rate clock at its phase boundaries in
// status[2 + tiles + 8 * tile + k] (tools/sps_stamps.py reads them)
#ifdef TGP_GEMM_STAMPS
#define SPS_STAMP(k)                                                                                                   \
  do {                                                                                                                 \
    if (threadIdx.x == 0) p.status[2 + gridDim.x + 8 * blockIdx.x + (k)] = wall_clock64();                             \
  } while (0)
#else
#define SPS_STAMP(k) do { } while (0)
#endif

template <int MODE, int WAVES>  // MODE 0: kept-node selection (TopK, NDP-shaped); 1: every node in one cluster (Graclus)
__global__ __launch_bounds__(WAVES * 64) void sparse_pool_small_kernel(SpsArgs p) {
  __shared__ uint32_t s_cnt[WAVES];
  __shared__ uint32_t s_base;
  __shared__ int s_ok;
  __shared__ int64_t s_nb[WAVES + 1];  // node offsets of the workgroup's graphs
  __shared__ int s_eb[WAVES + 1], s_ab[WAVES + 1];
  __shared__ int64_t s_rng[4];  // the workgroup's edge range and assignment slice
  __shared__ SpsClusterLds s_cl_all[MODE == 1 ? WAVES : 1];
  __shared__ int s_tile;
  const int lane = lane_id(), wv = wave_id();
  SPS_STAMP(0);
  const int tile_id = sps_tile_id(p.status + 1, p.tag, p.ticket, &s_tile);
  const int64_t g0 = static_cast<int64_t>(tile_id) * WAVES;
  const int64_t g = g0 + wv;
  const bool live = g < p.B;
  if (threadIdx.x <= WAVES) {
    const int64_t gi = g0 + threadIdx.x < p.B ? g0 + threadIdx.x : p.B;
    s_nb[threadIdx.x] = p.gptr[gi];
    s_eb[threadIdx.x] = INT_MAX;
    s_ab[threadIdx.x] = INT_MAX;
  }
  // r4, late: the caller may hand over the per-graph offsets it already has (the edge offsets memoised per edge list and
  // batch vector, TopkSelect's keep-count prefix): the graph's ranges are then ONE round trip instead of the three
  // dependent search rounds + the boundary pass below (5.8 of the kernel's 13.4 us on 2048 PROTEINS-shaped graphs).
  // Nothing is trusted: the ranges must tile [0, E) / [0, nnz) and every edge / kept node is range-checked as before.
  const bool given = p.edge_ptr != nullptr && (MODE == 1 || p.assign_ptr != nullptr);
  if (given) {
    if (threadIdx.x <= WAVES) {
      const int64_t gi = g0 + threadIdx.x < p.B ? g0 + threadIdx.x : p.B;
      const int64_t eg = p.edge_ptr[gi], ag = MODE == 0 ? p.assign_ptr[gi] : 0;
      if (threadIdx.x == 0) {
        s_rng[0] = eg;
        s_rng[2] = ag;
      }
      if (threadIdx.x == WAVES) {
        s_rng[1] = eg;
        s_rng[3] = ag;
      }
      // offsets relative to the workgroup's first graph, as the boundary pass leaves them (clamped: a corrupt table
      // must fail the checks below, not overflow an int)
      const int64_t e_first = p.edge_ptr[g0 < p.B ? g0 : p.B], a_first = MODE == 0 ? p.assign_ptr[g0 < p.B ? g0 : p.B] : 0;
      const int64_t de = eg - e_first, da = ag - a_first;
      s_eb[threadIdx.x] = (de < 0 || de > INT_MAX - 1) ? -1 : static_cast<int>(de);  // -1: refused below
      s_ab[threadIdx.x] = (da < 0 || da > INT_MAX - 1) ? -1 : static_cast<int>(da);
    }
  } else if (wv == 0) {
    const int64_t N0 = p.gptr[g0], N1 = p.gptr[g0 + WAVES < p.B ? g0 + WAVES : p.B];
    if constexpr (MODE == 0) {
      const int64_t* const arrs[4] = {p.row, p.row, p.node_index, p.node_index};
      const int64_t ns[4] = {p.E, p.E, p.nnz, p.nnz}, keys[4] = {N0, N1, N0, N1};
      int64_t res[4];
      wave_lower_bounds<4>(arrs, ns, keys, res);
      if (lane < 4) s_rng[lane] = lane == 0 ? res[0] : (lane == 1 ? res[1] : (lane == 2 ? res[2] : res[3]));
    } else {
      const int64_t* const arrs[2] = {p.row, p.row};
      const int64_t ns[2] = {p.E, p.E}, keys[2] = {N0, N1};
      int64_t res[2];
      wave_lower_bounds<2>(arrs, ns, keys, res);
      if (lane < 4) s_rng[lane] = lane == 0 ? res[0] : (lane == 1 ? res[1] : 0);
    }
  }
  __syncthreads();
  const int64_t E0 = s_rng[0], E1 = s_rng[1], A0 = s_rng[2], A1 = s_rng[3];
  const int64_t LE = E1 - E0, LA = A1 - A0;
  bool bad = false;
  // the workgroups' ranges tile the arrays: consecutive workgroups search for the same key (a deterministic function of
  // array and key, sorted or not), the first range starts at 0, the last one ends at the end
  if (LE < 0 || LA < 0 || (tile_id == 0 && (E0 != 0 || A0 != 0 || s_nb[0] != 0)) ||
      (tile_id == static_cast<int>(gridDim.x) - 1 && (E1 != p.E || (MODE == 0 && A1 != p.nnz) || s_nb[WAVES] != p.N)))
    bad = true;
  if (!bad && !given) {
    sps_boundaries<WAVES>(p.row, E0, LE, s_nb, s_eb);
    if constexpr (MODE == 0) sps_boundaries<WAVES>(p.node_index, A0, LA, s_nb, s_ab);
  }
  if (!given) __syncthreads();
  SPS_STAMP(1);
  int64_t n0 = s_nb[wv], n1 = s_nb[wv + 1];
  if (n1 < n0 || n1 - n0 > 64 || n0 < 0 || n1 > p.N) {
    bad = true;
    n0 = n1 = 0;
  }
  int64_t e0 = 0, e1 = 0, a0 = 0, a1 = 0;
  if (given && (s_eb[wv] < 0 || s_eb[wv + 1] < 0 || s_ab[wv] < 0 || s_ab[wv + 1] < 0)) bad = true;  // a corrupt table
  if (!bad) {
    const int64_t b0 = s_eb[wv] < LE ? s_eb[wv] : LE, b1 = s_eb[wv + 1] < LE ? s_eb[wv + 1] : LE;
    e0 = E0 + b0;
    e1 = E0 + b1;
    if (b1 < b0 || (wv == 0 && b0 != 0) || (wv == WAVES - 1 && b1 != LE)) bad = true;
    if constexpr (MODE == 0) {
      const int64_t c0 = s_ab[wv] < LA ? s_ab[wv] : LA, c1 = s_ab[wv + 1] < LA ? s_ab[wv + 1] : LA;
      a0 = A0 + c0;
      a1 = A0 + c1;
      if (c1 < c0 || c1 - c0 > 64 || (wv == 0 && c0 != 0) || (wv == WAVES - 1 && c1 != LA)) bad = true;
    }
  }
  if (bad) e0 = e1 = a0 = a1 = 0;
  if (!given && p.edge_ptr_out && live && !bad && lane == 0) {  // (a refused call's table is never used by the caller)
    p.edge_ptr_out[g] = e0;
    if (g == p.B - 1) p.edge_ptr_out[p.B] = e1;
  }
  const bool has_w = p.w != nullptr;
  const bool rsl = (p.flags & TGP_REMOVE_SELF_LOOPS) != 0, epsf = has_w && (p.flags & TGP_EPS_FILTER) != 0;
  const bool vec = (p.F & 3) == 0 && (p.x_stride & 3) == 0;
  const int F4 = static_cast<int>(p.F >> 2);
  uint32_t cnt = 0;

  // ---------------------------------------------------------------------------------------------- MODE 0 state
  unsigned long long M = 0;   // membership of the graph's nodes; a0 = new id of its first kept node
  constexpr int CI = 4;       // edge iterations kept in registers between the count and the write pass
  int64_t er[CI], ec[CI];
  float ewt[CI];
  unsigned long long ekeep[CI];
  // ---------------------------------------------------------------------------------------------- MODE 1 state
  int64_t cmin = 0;
  unsigned long long colmask = 0;  // lane = supernode row: its surviving columns
  int row_lo = 0;                  // first LDS slot of the row
  uint32_t row_base = 0;           // survivors of the wave's earlier rows
  SpsClusterLds& L = s_cl_all[MODE == 1 ? wv : 0];

  int ka = 0;            // MODE 0: the graph's kept nodes, one per lane
  int64_t v = 0, ci = 0;
  float wa = 1.0f;
  int kc = 0;            // MODE 1: the graph's clusters; lane = node: its weight
  float wj = 1.0f;
  if constexpr (MODE == 0) {
    ka = static_cast<int>(a1 - a0);
    const bool act = lane < ka;
    // one round trip: the graph's slice of the assignment and its first CI * 64 edges, all requested before any is used
    const int64_t ai = act ? a0 + lane : 0;
    v = p.nnz > 0 ? p.node_index[ai] : 0;
    ci = p.nnz > 0 ? p.cluster_index[ai] : 0;
    wa = (p.weight && p.nnz > 0) ? p.weight[ai] : 1.0f;
#pragma unroll
    for (int it = 0; it < CI; ++it) {
      const int64_t ee = e0 + it * WAVE + lane;
      const int64_t es = ee < e1 ? ee : 0;
      er[it] = p.E > 0 ? p.row[es] : 0;
      ec[it] = p.E > 0 ? p.col[es] : 0;
      ewt[it] = has_w ? p.w[es] : 1.0f;
    }
    const int64_t vprev = __shfl_up(v, 1, WAVE);
    const bool okv = !act || (v >= n0 && v < n1 && (lane == 0 || vprev < v));
    const int64_t cl = ci - a0;  // one supernode per kept node, numbered graph-major: a permutation of the slice
    const bool okc = !act || (cl >= 0 && cl < ka);
    if (__any(!okv || !okc)) bad = true;
    M = wave_or64((act && okv) ? 1ull << (v - n0) : 0ull);
    const unsigned long long CM = wave_or64((act && okc) ? 1ull << cl : 0ull);
    if (__popcll(CM) != ka) bad = true;
    SPS_STAMP(2);
    // A5 + A6, pass 1: survivors of the graph (the predicate of utils/ops.py:370-380 on the induced subgraph)
    auto survives = [&](bool on, int64_t r, int64_t c, float wt) -> bool {
      const bool inb = r >= n0 && r < n1 && c >= n0 && c < n1;
      if (__any(on && !inb)) bad = true;  // an edge that leaves its graph (or rows that are not grouped by graph)
      bool keep = on && inb && (((M >> ((r - n0) & 63)) & (M >> ((c - n0) & 63)) & 1ull) != 0);
      if (rsl && r == c) keep = false;
      if (epsf && !(fabsf(wt) > p.eps)) keep = false;
      return keep;
    };
#pragma unroll
    for (int it = 0; it < CI; ++it) {
      const bool on = e0 + it * WAVE + lane < e1;
      ekeep[it] = __ballot(survives(on, er[it], ec[it], ewt[it]));
      cnt += __popcll(ekeep[it]);
    }
    for (int64_t e = e0 + CI * WAVE; e < e1; e += WAVE) {  // (graphs of more than CI * 64 edges: re-read in pass 2)
      const int64_t ee = e + lane;
      const bool on = ee < e1;
      const int64_t r = on ? p.row[ee] : n0, c = on ? p.col[ee] : n0;
      const float wt = (on && has_w) ? p.w[ee] : 1.0f;
      cnt += __popcll(__ballot(survives(on, r, c, wt)));
    }
    if (bad) cnt = 0;
  } else {
    // ------------------------------------------------------------------------------------------------ MODE 1
    const int n = static_cast<int>(n1 - n0);
    const bool nact = lane < n;
    // one round trip: the graph's nodes and the previous graph's cluster ids (the usual case: it is not empty)
    const int64_t q1 = n0, q0 = wv > 0 ? s_nb[wv - 1] : (g > 0 ? p.gptr[g - 1] : 0);
    const int64_t ni = nact ? p.node_index[n0 + lane] : n0 + lane;
    const int64_t cj = nact ? p.cluster_index[n0 + lane] : 0;
    wj = (nact && p.weight) ? p.weight[n0 + lane] : 1.0f;
    int64_t pc = (q0 + lane < q1) ? p.cluster_index[q0 + lane] : -1;
    if (__any(nact && ni != n0 + lane)) bad = true;  // every node assigned, in node order (base_select.py:58)
    cmin = wave_min64(nact ? cj : INT64_MAX);
    const int64_t cmax = wave_max64(nact ? cj : -1);
    if (n > 0) {
      if (cmin < 0 || cmax >= p.K || cmax - cmin + 1 > 64) bad = true;
      else kc = static_cast<int>(cmax - cmin + 1);
      // cluster ids are contiguous per graph and ascending over the graphs (what a per-graph selector's unique()
      // relabelling gives): no cluster spans two graphs and the graph-major output is PyG coalesce's global order
      int64_t want = 0;
      if (g > 0) {
        if (q1 == q0) {  // the previous graph is empty: walk back to the last graph that is not
          int64_t gp = g - 1;
          while (gp >= 0 && p.gptr[gp + 1] == p.gptr[gp]) --gp;
          pc = -1;
          if (gp >= 0) {
            const int64_t r0 = p.gptr[gp], r1 = p.gptr[gp + 1];
            pc = (r0 + lane < r1) ? p.cluster_index[r0 + lane] : -1;
          }
        }
        want = wave_max64(pc) + 1;
      }
      if (cmin != want) bad = true;
      if (n1 == p.N && cmax != p.K - 1) bad = true;
    } else {
      cmin = 0;
    }
    if (bad) kc = 0;
    const int nn = bad ? 0 : n;
    // members of every local cluster
    L.mem[lane] = 0ull;
    L.deg[lane] = 0u;
    __builtin_amdgcn_wave_barrier();
    if (lane < nn) atomicOr(&L.mem[cj - cmin], 1ull << lane);
    __builtin_amdgcn_wave_barrier();
    const unsigned long long mymem = lane < kc ? L.mem[lane] : 0ull;
    if (__any(lane < kc && mymem == 0ull)) bad = true;  // an id without a node would keep reduce_batch's arange value
    // A4, first half: the graph's edges into LDS (cluster of the column, weight), edges per local row node
    int ne = static_cast<int>(e1 - e0);
    if (bad || e1 - e0 > SPS_CAP) {
      if (e1 - e0 > SPS_CAP) bad = true;
      ne = 0;
    }
    const int clj = static_cast<int>(cj - cmin) & 63;
    int64_t carry = n0;
    for (int base = 0; base < ne; base += WAVE) {
      const int t = base + lane;
      const bool on = t < ne;
      const int64_t r = on ? p.row[e0 + t] : n1, c = on ? p.col[e0 + t] : n0;
      const float wt = (on && has_w) ? p.w[e0 + t] : 1.0f;
      const bool inb = !on || (r >= n0 && r < n1 && c >= n0 && c < n1);
      int64_t rp = __shfl_up(r, 1, WAVE);
      if (lane == 0) rp = carry;
      if (__any(!inb || (on && r < rp))) bad = true;  // rows ascending inside the graph: a node's edges are one range
      carry = __shfl(r, WAVE - 1, WAVE);
      const int cc = __shfl(clj, static_cast<int>(c - n0) & 63, WAVE);  // the column's cluster sits in lane (c - n0)
      if (on && inb) {
        atomicAdd(&L.deg[r - n0], 1u);
        L.ecc[t] = static_cast<uint8_t>(cc);
        L.ew[t] = wt;
      }
    }
    if (bad) {
      ne = 0;
      kc = 0;
    }
    SPS_STAMP(2);
    __builtin_amdgcn_wave_barrier();
    {
      const uint32_t d = ne > 0 ? L.deg[lane] : 0u;
      const uint32_t ps = wave_incl_scan(d) - d;
      __builtin_amdgcn_wave_barrier();
      L.deg[lane] = (ps << 16) | d;
      __builtin_amdgcn_wave_barrier();
    }
    // A4, second half.  lane = supernode row.  First walk: the row's set of columns (a mask) and its raw length.
    if (ne > 0 && lane < kc) {
      unsigned long long mask = mymem;
      while (mask) {
        const int j = __builtin_ctzll(mask);
        mask &= mask - 1;
        const uint32_t pd = L.deg[j];
        const int t0 = static_cast<int>(pd >> 16), t1 = t0 + static_cast<int>(pd & 0xFFFFu);
        for (int t = t0; t < t1; ++t) colmask |= 1ull << L.ecc[t];
      }
      if (rsl) colmask &= ~(1ull << lane);  // (self loops are their own key: dropping them first changes nothing else)
    }
    const uint32_t width = __popcll(colmask);  // merged entries of the row (<= raw): its LDS slots
    row_lo = static_cast<int>(wave_incl_scan(width) - width);
    if (has_w && width > 0) {
      // second walk, input order (ascending member, then position): the slot of column c is the number of the row's
      // columns below c; the first entry of a slot sets it, later ones fold in
      unsigned long long seen = 0, mask = mymem;
      while (mask) {
        const int j = __builtin_ctzll(mask);
        mask &= mask - 1;
        const uint32_t pd = L.deg[j];
        const int t0 = static_cast<int>(pd >> 16), t1 = t0 + static_cast<int>(pd & 0xFFFFu);
        for (int t = t0; t < t1; ++t) {
          const int cc = L.ecc[t];
          const unsigned long long bit = 1ull << cc;
          if (!(colmask & bit)) continue;
          const int slot = row_lo + __popcll(colmask & (bit - 1ull));
          const float wt = L.ew[t];
          if (seen & bit) {
            L.lval[slot] = sps_reduce(L.lval[slot], wt, p.reduce_op);
            L.lcnt[slot] = static_cast<uint16_t>(L.lcnt[slot] + 1);
          } else {
            seen |= bit;
            L.lval[slot] = wt;
            L.lcnt[slot] = 1;
          }
        }
      }
      // mean, |w| > eps: columns that fall out leave the mask (their slots stay where they are)
      unsigned long long cm = colmask;
      int slot = row_lo;
      while (cm) {
        const int cc = __builtin_ctzll(cm);
        cm &= cm - 1;
        float val = L.lval[slot];
        if (p.reduce_op == TGP_MEAN) {
          val = val / static_cast<float>(L.lcnt[slot]);
          L.lval[slot] = val;
        }
        if (epsf && !(fabsf(val) > p.eps)) {
          colmask &= ~(1ull << cc);
          L.lcnt[slot] = 0;  // marks the slot as dropped for the write pass
        }
        ++slot;
      }
    }
    const uint32_t keepn = __popcll(colmask);
    const uint32_t incl = wave_incl_scan(keepn);
    row_base = incl - keepn;
    cnt = __shfl(incl, WAVE - 1, WAVE);
    if (bad) cnt = 0;
  }

  // ------------------------------------------------------------------------------ survivors in front of this graph
  SPS_STAMP(3);
  if (lane == 0) s_cnt[wv] = cnt | (bad ? 0x80000000u : 0u);
  __syncthreads();
  SPS_STAMP(4);
  uint32_t tile_tot = 0;
  bool tile_refused = false;
  if (wv == 0) {  // the tile's count goes out first: successors only need this word
    const uint32_t mine = lane < WAVES ? s_cnt[lane] : 0u;
    tile_tot = wave_sum32(mine & 0x7FFFFFFFu);
    tile_refused = __any((mine >> 31) != 0u);
    if (lane == 0)
      sps_store(p.status + 2 + tile_id, p.tag | (tile_id == 0 ? SPS_PRE : SPS_AGG) |
                                               (tile_refused ? 0x80000000ull : 0ull) | tile_tot);
  }
  SpsLook look;  // the predecessors' words are requested now and consumed behind the Reduce part
  if (wv == 0 && tile_id > 0) sps_lookback_issue(p.status, tile_id, p.tag, look);
  // ------------------------------------------------------------------- A1 + A2, while the look-back words travel
  if (!bad) {
    if constexpr (MODE == 0) {
      const bool act = lane < ka;
      if (ka > 0) {
        // A1: x_pool[cluster] = 0 + weight * x[node] (one member per supernode); A2: batch_pool[cluster] = graph id
        if (p.batch_pool && act) p.batch_pool[ci] = g;
        if (vec) {
          const int total = ka * F4;
          for (int base = 0; base < total; base += WAVE) {
            const int idx = base + lane;
            const bool on = idx < total;
            const int a = on ? idx / F4 : 0;
            const int f = (idx - a * F4) * 4;
            const int64_t vv = __shfl(v, a, WAVE), cc = __shfl(ci, a, WAVE);
            const float ww = __shfl(wa, a, WAVE);
            if (on) {
              const float4 t = *reinterpret_cast<const float4*>(p.x + vv * p.x_stride + f);
              float4 o;
              o.x = __fadd_rn(0.f, __fmul_rn(t.x, ww));
              o.y = __fadd_rn(0.f, __fmul_rn(t.y, ww));
              o.z = __fadd_rn(0.f, __fmul_rn(t.z, ww));
              o.w = __fadd_rn(0.f, __fmul_rn(t.w, ww));
              *reinterpret_cast<float4*>(p.x_pool + cc * p.F + f) = o;
            }
          }
        } else {
          const int Fi = static_cast<int>(p.F);
          const int total = ka * Fi;
          for (int base = 0; base < total; base += WAVE) {
            const int idx = base + lane;
            const bool on = idx < total;
            const int a = on ? idx / Fi : 0;
            const int f = idx - a * Fi;
            const int64_t vv = __shfl(v, a, WAVE), cc = __shfl(ci, a, WAVE);
            const float ww = __shfl(wa, a, WAVE);
            if (on) p.x_pool[cc * p.F + f] = __fadd_rn(0.f, __fmul_rn(p.x[vv * p.x_stride + f], ww));
          }
        }
      }
    } else {
      if (kc > 0) {
        // A2 + A1: members in ascending node order, products rounded before the add
        if (p.batch_pool && lane < kc) p.batch_pool[cmin + lane] = g;
        const int per = vec ? F4 : static_cast<int>(p.F);
        const int total = kc * per;
        for (int base = 0; base < total; base += WAVE) {
          const int idx = base + lane;
          const bool on = idx < total;
          const int rr = on ? idx / per : 0;
          const int f = (idx - rr * per) * (vec ? 4 : 1);
          unsigned long long mask = on ? L.mem[rr] : 0ull;
          float4 acc = {0.f, 0.f, 0.f, 0.f};
          while (__any(mask != 0ull)) {
            const int j = mask ? __builtin_ctzll(mask) : 0;
            const float ww = __shfl(wj, j, WAVE);
            if (mask) {
              mask &= mask - 1;
              const float* src = p.x + (n0 + j) * p.x_stride + f;
              if (vec) {
                const float4 t = *reinterpret_cast<const float4*>(src);
                acc.x = __fadd_rn(acc.x, __fmul_rn(t.x, ww));
                acc.y = __fadd_rn(acc.y, __fmul_rn(t.y, ww));
                acc.z = __fadd_rn(acc.z, __fmul_rn(t.z, ww));
                acc.w = __fadd_rn(acc.w, __fmul_rn(t.w, ww));
              } else {
                acc.x = __fadd_rn(acc.x, __fmul_rn(*src, ww));
              }
            }
          }
          if (on) {
            float* dst = p.x_pool + (cmin + rr) * p.F + f;
            if (vec) *reinterpret_cast<float4*>(dst) = acc;
            else *dst = acc.x;
          }
        }
      }
    }
  }
  if (wv == 0) {
    const int tile = tile_id;
    uint32_t excl = 0;
    bool refused = tile_refused;
    if (tile > 0) {
      bool before = false;
      sps_lookback_finish(p.status, tile, p.tag, look, &excl, &before);
      refused = refused || before;
      if (lane == 0)
        sps_store(p.status + 2 + tile, p.tag | SPS_PRE | (refused ? 0x80000000ull : 0ull) |
                                           static_cast<unsigned long long>((excl + tile_tot) & 0x7FFFFFFFu));
    }
    if (lane == 0) {
      s_base = excl;
      s_ok = refused ? 0 : 1;
      if (tile == static_cast<int>(gridDim.x) - 1)  // the verdict and the size of the whole call, in one word
        __hip_atomic_store(p.result, p.tag | (refused ? 0x80000000ull : 0ull) |
                                         static_cast<unsigned long long>((excl + tile_tot) & 0x7FFFFFFFu),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  __syncthreads();
  SPS_STAMP(5);
  if (!s_ok || bad) return;  // (a refusal seen so far: the outputs will be discarded anyway)
  uint32_t base = s_base;
  for (int w2 = 0; w2 < wv; ++w2) base += s_cnt[w2] & 0x7FFFFFFFu;

  if constexpr (MODE == 0) {
    // pass 2: survivors written once at their final offsets (input order kept); the first CI iterations from registers
    uint32_t pos = base;
    auto emit = [&](unsigned long long km, int64_t r, int64_t c, float wt) {
      if ((km >> lane) & 1ull) {
        const int lr = static_cast<int>(r - n0) & 63, lc = static_cast<int>(c - n0) & 63;
        const uint32_t o = pos + __popcll(km & lanemask_lt());
        p.out_row[o] = a0 + __popcll(M & ((1ull << lr) - 1ull));
        p.out_col[o] = a0 + __popcll(M & ((1ull << lc) - 1ull));
        if (has_w) p.out_w[o] = wt;
      }
      pos += __popcll(km);
    };
#pragma unroll
    for (int it = 0; it < CI; ++it) emit(ekeep[it], er[it], ec[it], ewt[it]);
    for (int64_t e = e0 + CI * WAVE; e < e1; e += WAVE) {
      const int64_t ee = e + lane;
      const bool on = ee < e1;
      const int64_t r = on ? p.row[ee] : n0, c = on ? p.col[ee] : n0;
      const float wt = (on && has_w) ? p.w[ee] : 1.0f;
      bool keep = on && (((M >> ((r - n0) & 63)) & (M >> ((c - n0) & 63)) & 1ull) != 0);
      if (rsl && r == c) keep = false;
      if (epsf && !(fabsf(wt) > p.eps)) keep = false;
      emit(__ballot(keep), r, c, wt);
    }
  } else {
    // the row's surviving columns in ascending order; a weighted row skips the slots the filters emptied
    uint32_t o = base + row_base;
    int slot = row_lo;
    unsigned long long cm = colmask;
    while (cm) {
      const int cc = __builtin_ctzll(cm);
      cm &= cm - 1;
      if (has_w) {
        while (L.lcnt[slot] == 0) ++slot;
        p.out_w[o] = L.lval[slot];
        ++slot;
      }
      p.out_row[o] = cmin + lane;
      p.out_col[o] = cmin + cc;
      ++o;
    }
  }
  SPS_STAMP(6);
}

}  // namespace tgp

using namespace tgp;

constexpr int SPS_WAVES_TOPK = 8, SPS_WAVES_CLUSTER = 8;

/* graphs of at most this many nodes are pooled by one wave */
extern "C" int tgp_sparse_pool_small_max_graph_nodes(void) { return 64; }

extern "C" int64_t tgp_sparse_pool_small_status_words(int64_t num_graphs, int mode) {
  const int waves = mode == 0 ? SPS_WAVES_TOPK : SPS_WAVES_CLUSTER;
  const int64_t tiles = (num_graphs + waves - 1) / waves;
#ifdef TGP_GEMM_STAMPS
  return 2 + 9 * tiles;
#else
  return 2 + tiles;
#endif
}

// out[g] = first position of `values` (ascending, n entries) that is >= graph_ptr[g], g = 0..B: the per-graph offsets of a
// row-sorted edge list (values = its row array) or of a node-sorted assignment (values = node_index) of a sorted batch.
// A caller that pools the same edge list / batch vector again keeps them (tgp_sparse_pool_small_f32's edge_ptr).
__global__ __launch_bounds__(256) void graph_lower_bounds_kernel(const int64_t* __restrict__ values, int64_t n,
                                                                 const int64_t* __restrict__ gptr, int64_t B,
                                                                 int64_t* __restrict__ out) {
  const int64_t g = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (g > B) return;
  const int64_t key = gptr[g];
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (values[mid] < key) lo = mid + 1; else hi = mid;
  }
  out[g] = lo;
}

extern "C" int tgp_graph_lower_bounds_i64(const int64_t* values, int64_t n, const int64_t* graph_ptr, int64_t B,
                                          int64_t* out, void* stream_) {
  TGP_REQUIRE(n >= 0 && B >= 0 && graph_ptr && out && (n == 0 || values), TGP_ERR_INVALID,
              "tgp_graph_lower_bounds_i64: bad argument");
  hipLaunchKernelGGL(graph_lower_bounds_kernel, dim3(cdiv(B + 1, 256)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     values, n, graph_ptr, B, out);
  return check_launch("tgp_graph_lower_bounds_i64");
}

extern "C" int tgp_sparse_pool_small_f32(const float* x, int64_t N, int64_t F, int64_t x_stride, const int64_t* graph_ptr,
                                         int64_t B, const int64_t* edge_ptr, const int64_t* assign_ptr,
                                         int64_t* edge_ptr_out,
                                         const int64_t* row, const int64_t* col, const float* w, int64_t E,
                                         const int64_t* node_index, const int64_t* cluster_index, const float* weight,
                                         int64_t nnz, int64_t K, int mode, int reduce_op, int flags, float eps,
                                         float* x_pool, int64_t* batch_pool, int64_t* out_row, int64_t* out_col,
                                         float* out_w, uint64_t* status, int64_t status_words, uint64_t* result,
                                         uint32_t epoch, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(x && graph_ptr && node_index && cluster_index && x_pool && status && result && N > 0 && F > 0 && B > 0 && E >= 0 &&
                  nnz >= 0 && K >= 0 && x_stride >= F && (mode == 0 || mode == 1),
              TGP_ERR_INVALID, "tgp_sparse_pool_small_f32: bad argument");
  TGP_REQUIRE(E == 0 || (row && col && out_row && out_col && (!w || out_w)), TGP_ERR_INVALID,
              "tgp_sparse_pool_small_f32: null edge pointer");
  TGP_REQUIRE(reduce_op >= TGP_SUM && reduce_op <= TGP_MUL, TGP_ERR_INVALID, "tgp_sparse_pool_small_f32: reduce_op");
  TGP_REQUIRE(E < (1ll << 31) && N < (1ll << 31) && B < (1ll << 24) && epoch != 0 && epoch < (1u << 29), TGP_ERR_RANGE,
              "tgp_sparse_pool_small_f32: size or epoch out of range");
  TGP_REQUIRE(status_words >= tgp_sparse_pool_small_status_words(B, mode), TGP_ERR_WORKSPACE,
              "tgp_sparse_pool_small_f32: status buffer too small");
  SpsArgs a{x, N, F, x_stride, graph_ptr, B, row, col, w, E, node_index, cluster_index, weight, nnz, K, reduce_op, flags,
            eps, x_pool, batch_pool, out_row, out_col, out_w, edge_ptr, assign_ptr, edge_ptr_out,
            reinterpret_cast<unsigned long long*>(status),
            reinterpret_cast<unsigned long long*>(result), static_cast<unsigned long long>(epoch) << SPS_EPOCH_SHIFT,
            kLookbackTicket};
  if (mode == 0) {
    hipLaunchKernelGGL((sparse_pool_small_kernel<0, SPS_WAVES_TOPK>), dim3(cdiv(B, SPS_WAVES_TOPK)),
                       dim3(SPS_WAVES_TOPK * 64), 0, stream, a);
  } else {
    hipLaunchKernelGGL((sparse_pool_small_kernel<1, SPS_WAVES_CLUSTER>), dim3(cdiv(B, SPS_WAVES_CLUSTER)),
                       dim3(SPS_WAVES_CLUSTER * 64), 0, stream, a);
  }
  return check_launch("tgp_sparse_pool_small_f32");
}
