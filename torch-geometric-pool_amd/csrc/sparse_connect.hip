// A4 / A5 / A6 / A10: sparse Connect (edge-list coarsening) and its post-processing.
//
// Everything here is HBM-bound int64 / fp32 stream processing.  Output order is part of the
// contract (SURVEY.md 7 "ordering contracts"): PyG `subgraph` keeps input edge order, PyG
// `coalesce` returns row-major sorted unique edges, `nonzero` returns (b,row,col) order — so all
// compactions are scan-based (ballot ranks + block offsets), never atomic-append.
#include <stdlib.h>

#include "lookback.h"

namespace tgp {

constexpr int kCompactItems = 4;
constexpr int kCompactTile = 256 * kCompactItems;

// =====================================================================================
// A5 + A6 filters: induced subgraph with relabelling (connect/base_conn.py:79-82)
// =====================================================================================
// relabel[node] = position in node_index, and a 1-bit-per-node membership map.  The predicate of both passes
// tests the bitmap (N/8 bytes: 125 KB for a million nodes, cache resident) instead of gathering from the 4 N-byte
// relabel table; the table is read only for the edges that survive (fill pass).
__global__ __launch_bounds__(256) void relabel_scatter_kernel(const int64_t* __restrict__ node_index, int64_t k,
                                                              int64_t n, int32_t* __restrict__ relabel,
                                                              uint32_t* __restrict__ member_bits,
                                                              int* __restrict__ unsorted) {
  const int64_t j = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  bool live = j < k;
  const int64_t v = live ? node_index[j] : -1;
  if (live && static_cast<uint64_t>(v) >= static_cast<uint64_t>(n)) {  // a kept node outside [0, n)
    unsorted[1] = 1;
    live = false;
  }
  if (live) {
    relabel[v] = static_cast<int32_t>(j);  // only ever read for member nodes: no fill of the other N - k entries
    if (j > 0 && node_index[j - 1] >= v) *unsorted = 1;  // then position != rank: the table is the only way
  }
  // membership bitmap (zeroed by the caller).  An ascending node_index puts the ~16 members of a 32-node word on
  // neighbouring lanes: fold their bits with a segmented suffix-OR over the wave and let the first lane of every run
  // issue ONE atomicOr (one atomic per lane onto a handful of words measured 28 us for 500 k nodes).  Runs are only
  // an optimisation: for an unsorted list equal words need not be adjacent, and every run head still ORs its bits in.
  const int lane = lane_id();
  const int64_t word = live ? (v >> 5) : -1 - lane;  // dead lanes: distinct keys, no bits
  uint32_t acc = live ? 1u << (v & 31) : 0u;
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    const uint32_t t = __shfl_down(acc, d, WAVE);
    const int64_t tw = __shfl_down(word, d, WAVE);
    if (lane + d < WAVE && tw == word) acc |= t;
  }
  const int64_t pw = __shfl_up(word, 1, WAVE);
  if (live && (lane == 0 || pw != word)) atomicOr(member_bits + word, acc);
}

// Relabelling by rank: with an ascending node_index the new id of node v is its rank among the members,
//   rank128[v >> 7] + popcount(bitmap words of the 128-node block before v's word) + popcount(v's word below bit v),
// so the staging pass needs no gather from the 4 N-byte relabel table; the directory rank128 (exclusive scan of the
// blocks' popcounts) is built by the stage kernel itself from its LDS copy of the bitmap.
// WT: the weight type of the list (float, or double for model.double() inputs: r4) -- weights only pass through and
// meet the |w| > eps test, in their own precision
template <typename WT>
struct SubgraphPredT {
  const int64_t* row;
  const int64_t* col;
  const WT* w;
  const int32_t* relabel;       // nullptr = no node filter
  const uint32_t* member_bits;  // set with relabel
  const int* unsorted;          // set with relabel: node_index is not ascending
  int flags;
  WT eps;                       // the caller's eps at call time (reference ops.py:377 reads the module global)
  int64_t n;                    // number of nodes: endpoints outside [0, n) set *bad_ids (the reference's index ops raise)
  int* bad_ids;
  int bad_value;                // what is stored there: 1 into a zeroed word (count -> fill pair), or the call's epoch
                                // into the low half of word [1] of the caller's never-cleared status buffer (single pass)
  // keep / drop only; r, c are the ORIGINAL endpoints (relabelling is injective, so r == c decides self loops)
  __device__ __forceinline__ bool operator()(int64_t e, int64_t& r, int64_t& c) const {
    r = row[e];
    c = col[e];
    if (static_cast<uint64_t>(r) >= static_cast<uint64_t>(n) || static_cast<uint64_t>(c) >= static_cast<uint64_t>(n)) {
      *bad_ids = bad_value;
      return false;
    }
    if (relabel) {
      const uint32_t br = member_bits[r >> 5] >> (r & 31), bc = member_bits[c >> 5] >> (c & 31);
      if (!(br & bc & 1u)) return false;
    }
    if ((flags & TGP_REMOVE_SELF_LOOPS) && r == c) return false;
    if (w && (flags & TGP_EPS_FILTER) && !(fabs(w[e]) > eps)) return false;
    return true;
  }
};
using SubgraphPred = SubgraphPredT<float>;

// The edge list is walked ONCE, in chunks of SG_CHUNK = 4096 edges, by persistent 1024-thread workgroups (one per
// CU).  A thread owns 4 CONSECUTIVE edges of a chunk, so row / col arrive as 16-byte loads, and the
// order-preserving rank of an edge is (survivors of earlier chunks) + (survivors of earlier threads) + (survivors
// among the thread's earlier edges).  The node-membership bitmap is copied into LDS once per workgroup when it fits
// (N <= 1.2 M nodes): the two membership tests per edge are then LDS reads instead of 64 scattered L2 requests per
// wave instruction, which is what bounded the first version of these kernels.
constexpr int SG_PER = 4;
constexpr int SG_THREADS = 1024;
constexpr int SG_CHUNK = SG_THREADS * SG_PER;
constexpr int SG_LDS_WORDS_MAX = 38 * 1024;  // 152 KB of the 160 KB LDS

template <typename WT>
struct SgEdgesT {
  int64_t r[SG_PER], c[SG_PER];
  WT w[SG_PER];
  bool keep[SG_PER];
};
using SgEdges = SgEdgesT<float>;

// the streaming part: the thread's 4 consecutive (row, col) pairs; keep[] = "edge exists"
template <typename WT>
__device__ __forceinline__ void sg_fetch(const SubgraphPredT<WT>& pred, int64_t e0, int64_t E, SgEdgesT<WT>& t) {
  if (e0 >= E) {
#pragma unroll
    for (int j = 0; j < SG_PER; ++j) { t.keep[j] = false; t.r[j] = 0; t.c[j] = 0; t.w[j] = WT(0); }
    return;
  }
  if (pred.w) {  // streamed with the indices (16-byte loads) rather than fetched sparsely for the survivors
    if (e0 + SG_PER <= E && (reinterpret_cast<uintptr_t>(pred.w + e0) & 15) == 0) {
      if constexpr (sizeof(WT) == 4) {
#pragma unroll
        for (int g = 0; g < SG_PER; g += 4) {
          const float4 wv = *reinterpret_cast<const float4*>(pred.w + e0 + g);
          t.w[g] = wv.x; t.w[g + 1] = wv.y; t.w[g + 2] = wv.z; t.w[g + 3] = wv.w;
        }
      } else {
#pragma unroll
        for (int g = 0; g < SG_PER; g += 2) {
          const double2 wv = *reinterpret_cast<const double2*>(pred.w + e0 + g);
          t.w[g] = wv.x; t.w[g + 1] = wv.y;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < SG_PER; ++j) t.w[j] = e0 + j < E ? pred.w[e0 + j] : WT(0);
    }
  }
  if (e0 + SG_PER <= E && ((reinterpret_cast<uintptr_t>(pred.row + e0) | reinterpret_cast<uintptr_t>(pred.col + e0)) & 15) == 0) {
    typedef long long ll2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int g = 0; g < SG_PER; g += 2) {
      const ll2 rr = *reinterpret_cast<const ll2*>(pred.row + e0 + g), cc = *reinterpret_cast<const ll2*>(pred.col + e0 + g);
      t.r[g] = rr.x; t.r[g + 1] = rr.y;
      t.c[g] = cc.x; t.c[g + 1] = cc.y;
    }
#pragma unroll
    for (int j = 0; j < SG_PER; ++j) t.keep[j] = true;
  } else {
#pragma unroll
    for (int j = 0; j < SG_PER; ++j) {
      t.keep[j] = e0 + j < E;
      t.r[j] = t.keep[j] ? pred.row[e0 + j] : 0;
      t.c[j] = t.keep[j] ? pred.col[e0 + j] : 0;
    }
  }
}

// the predicate: membership of both endpoints (bitmap), self loops, |w| > eps
template <bool LDSB, typename WT>
__device__ __forceinline__ bool sg_eval(const SubgraphPredT<WT>& pred, const uint32_t* s_bits, int64_t e0,
                                        SgEdgesT<WT>& t) {
  bool met_bad = false;
#pragma unroll
  for (int j = 0; j < SG_PER; ++j) {  // endpoints outside [0, n): flagged, never used as an index
    if (t.keep[j] && (static_cast<uint64_t>(t.r[j]) >= static_cast<uint64_t>(pred.n) ||
                      static_cast<uint64_t>(t.c[j]) >= static_cast<uint64_t>(pred.n))) {
      *pred.bad_ids = pred.bad_value;
      met_bad = true;
      t.keep[j] = false;
      t.r[j] = 0;
      t.c[j] = 0;
    }
  }
  if (pred.relabel) {  // all bitmap words are requested before any is tested
    uint32_t br[SG_PER], bc[SG_PER];
#pragma unroll
    for (int j = 0; j < SG_PER; ++j) {
      if constexpr (LDSB) {
        br[j] = s_bits[t.r[j] >> 5];
        bc[j] = s_bits[t.c[j] >> 5];
      } else {
        br[j] = pred.member_bits[t.r[j] >> 5];
        bc[j] = pred.member_bits[t.c[j] >> 5];
      }
    }
#pragma unroll
    for (int j = 0; j < SG_PER; ++j)
      t.keep[j] = t.keep[j] && (((br[j] >> (t.r[j] & 31)) & (bc[j] >> (t.c[j] & 31)) & 1u) != 0);
  }
#pragma unroll
  for (int j = 0; j < SG_PER; ++j) {
    if ((pred.flags & TGP_REMOVE_SELF_LOOPS) && t.r[j] == t.c[j]) t.keep[j] = false;
    if (pred.w && (pred.flags & TGP_EPS_FILTER) && !(fabs(t.w[j]) > pred.eps)) t.keep[j] = false;
  }
  return met_bad;
}

// exclusive scan of one value per thread over a 1024-thread workgroup (16 waves); s_w: 16 words
__device__ __forceinline__ uint32_t block_excl_scan_1024(uint32_t v, uint32_t* s_w, uint32_t* total) {
  const uint32_t inc = wave_incl_scan(v);
  if (lane_id() == WAVE - 1) s_w[wave_id()] = inc;
  __syncthreads();
  uint32_t off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const uint32_t c = s_w[w];
    if (w < wave_id()) off += c;
    tot += c;
  }
  if (total) *total = tot;
  __syncthreads();
  return off + inc - v;
}

// Pass 1 ("count"): ONE pass over the edge list.  Per chunk of 4096 edges a persistent 1024-thread workgroup evaluates
// the predicate, ranks the survivors (workgroup scan) and writes them -- already relabelled -- compacted at the
// chunk's own base in a staging area (new ids as int32, weight, offset inside the chunk as uint16), plus the chunk's
// survivor count.  Pass 2 ("fill", after the host has read the total and allocated the outputs) only moves the
// staged survivors to their final places: it reads 14 bytes and writes 20 per SURVIVOR instead of streaming the
// whole edge list a second time (at ratio 0.5 three quarters of the edges die: 200 MB -> 35 MB of reads).
// LDSB: 0 = bitmap in global memory, 1 = bitmap in LDS, 2 = bitmap + rank128 in LDS (relabel by rank)
template <typename WT>
struct SgStageT {
  int32_t* r;
  int32_t* c;
  WT* w;
  uint16_t* off;
};
using SgStage = SgStageT<float>;

// r4, SINGLE = true: the same pass writes the survivors ONCE, in their final int64 form at their final offsets of
// capacity-E outputs -- no staging area, no count kernel, no copy pass behind the host read (r3: 12 B read + 20 B written
// per survivor, a quarter of the call with its launch and the device-to-host copy).  The number of survivors in front of
// a chunk comes from the epoch-tagged decoupled look-back of lookback.h: the persistent workgroups take chunks in
// increasing order (chunk = workgroup + round * grid), so the predecessors of a chunk belong to the same round of
// workgroups that are all resident, and the next chunk's loads are already in flight while the words travel.  Edge ids
// outside [0, n) ride along as the refusal bit; the last chunk leaves {epoch, refused, total} in *result (pinned host
// memory: the caller polls it).
template <typename WT>
struct SgSingleT {
  int64_t* out_row;
  int64_t* out_col;
  WT* out_w;
  int64_t* out_eid;
  unsigned long long* status;
  unsigned long long* result;
  unsigned long long tag;
  // r5: the rank directory rank128 (members with id < 128 b) when the selector made it together with the bitmap
  // (TopkSelect's compaction pass writes both as by-products): the workgroups copy it instead of scanning for it,
  // and the call needs neither the memset nor the scatter kernel in front of this one
  const uint32_t* rank128;
};
using SgSingle = SgSingleT<float>;

// PRE (r5, with LDSB == 2 and SINGLE): bitmap AND rank directory come from the selector (sg.rank128): no scan, and the
// relabel table / sortedness flag are never read -- a compile-time switch, the kernel sits at the 128-register limit of
// a 1024-thread workgroup and a run-time branch here cost two spilled registers and 10 us
template <int LDSB, bool SINGLE, typename WT = float, bool PRE = false>
__global__ __launch_bounds__(SG_THREADS) void subgraph_stage_kernel(SubgraphPredT<WT> pred, int64_t E, int nchunks,
                                                                    int nwords, SgStageT<WT> st,
                                                                    uint32_t* __restrict__ block_counts,
                                                                    SgSingleT<WT> sg) {
  using SgEdges = SgEdgesT<WT>;
  extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
  __shared__ uint32_t s_w[16];
  __shared__ uint32_t s_base;
  uint32_t* s_rank = s_dyn + 4 * ((nwords + 3) / 4);  // behind the bitmap, padded to whole 128-node blocks
  bool by_rank = false;
  if constexpr (LDSB >= 1) {
    lds_copy_words<SG_THREADS>(s_dyn, pred.member_bits, nwords);
    if constexpr (LDSB == 2) {
      if constexpr (PRE) {  // the selector handed the directory over with the bitmap
        by_rank = true;
        lds_copy_words<SG_THREADS>(s_rank, sg.rank128, (nwords + 3) / 4);
      } else {
      // rank directory rank128[b] = members with id < 128 b, computed here from the LDS copy of the bitmap by every
      // workgroup for itself, in parallel on every CU, instead of a 10 us single-workgroup kernel in front of this one
      by_rank = *pred.unsorted == 0;
      __syncthreads();
      const int nblocks = (nwords + 3) / 4;
      // rounds of 1024 consecutive blocks, one block per thread (a 16-byte LDS read: conflict-free; a run of blocks
      // per thread would put all lanes on one bank), carry between rounds
      uint32_t carry = 0;
      for (int base = 0; base < nblocks; base += SG_THREADS) {
        const int bb = base + static_cast<int>(threadIdx.x);
        uint32_t v = 0;
        if (bb < nblocks) {  // (words past nwords in the last block are padding: masked)
          const uint4 q = *reinterpret_cast<const uint4*>(s_dyn + 4 * bb);
          const int left = nwords - 4 * bb;
          v = __popc(q.x) + (left > 1 ? __popc(q.y) : 0) + (left > 2 ? __popc(q.z) : 0) + (left > 3 ? __popc(q.w) : 0);
        }
        uint32_t total;
        const uint32_t off = block_excl_scan_1024(v, s_w, &total);
        if (bb < nblocks) s_rank[bb] = carry + off;
        carry += total;
      }
      }
    }
    __syncthreads();
  }
  auto new_id = [&](int64_t v) -> int32_t {
    if constexpr (LDSB == 2) {
      if (PRE || by_rank) {  // rank of the 128-node block + the set bits below v inside it: one 16-byte LDS read, no loop
        const int word = static_cast<int>(v >> 5), sub = word & 3;
        const uint4 b = *reinterpret_cast<const uint4*>(s_dyn + (word & ~3));
        const uint32_t below = (1u << (v & 31)) - 1u;
        const uint32_t m0 = sub > 0 ? 0xFFFFFFFFu : (sub == 0 ? below : 0u);
        const uint32_t m1 = sub > 1 ? 0xFFFFFFFFu : (sub == 1 ? below : 0u);
        const uint32_t m2 = sub > 2 ? 0xFFFFFFFFu : (sub == 2 ? below : 0u);
        const uint32_t m3 = sub == 3 ? below : 0u;
        return static_cast<int32_t>(s_rank[word >> 2] + __popc(b.x & m0) + __popc(b.y & m1) + __popc(b.z & m2) +
                                    __popc(b.w & m3));
      }
    }
    if constexpr (PRE) return 0;  // (unreachable)
    else return pred.relabel[v];
  };
  // the (row, col) stream of the next chunk is requested before this chunk's dependent work starts
  SgEdges nxt;
  sg_fetch(pred, static_cast<int64_t>(blockIdx.x) * SG_CHUNK + static_cast<int64_t>(threadIdx.x) * SG_PER, E, nxt);
  if constexpr (!SINGLE) {
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
      const int64_t e0 = static_cast<int64_t>(chunk) * SG_CHUNK + static_cast<int64_t>(threadIdx.x) * SG_PER;
      SgEdges t = nxt;
      if (chunk + static_cast<int>(gridDim.x) < nchunks)
        sg_fetch(pred, e0 + static_cast<int64_t>(gridDim.x) * SG_CHUNK, E, nxt);
      sg_eval<(LDSB >= 1), WT>(pred, s_dyn, e0, t);
      uint32_t mine = 0;
#pragma unroll
      for (int j = 0; j < SG_PER; ++j) mine += t.keep[j] ? 1u : 0u;
      uint32_t total;
      int64_t pos = static_cast<int64_t>(chunk) * SG_CHUNK + block_excl_scan_1024(mine, s_w, &total);
      if (threadIdx.x == 0) block_counts[chunk] = total;
      // (computing the new ids ahead of the scan, to overlap their LDS look-ups with its barriers, measured 68 vs 64 us)
#pragma unroll
      for (int j = 0; j < SG_PER; ++j) {
        if (t.keep[j]) {
          st.r[pos] = pred.relabel ? new_id(t.r[j]) : static_cast<int32_t>(t.r[j]);
          st.c[pos] = pred.relabel ? new_id(t.c[j]) : static_cast<int32_t>(t.c[j]);
          if (st.w) st.w[pos] = t.w[j];
          if (st.off) st.off[pos] = static_cast<uint16_t>(threadIdx.x * SG_PER + j);
          ++pos;
        }
      }
    }
  } else {
    // Software pipeline over the workgroup's chunks: chunk k + 1 is evaluated and its count PUBLISHED before chunk k
    // waits for its prefix, so the words chunk k needs (chunks of the same round, on other CUs) have had a whole chunk's
    // time to arrive -- with one workgroup per CU nothing else would hide that round trip (r4: 100 us with the wait
    // in line against 64 us for the staging pass of the count -> fill pair).
    const int G = static_cast<int>(gridDim.x);
    auto eval_publish = [&](int chunk, SgEdges& t, uint32_t& rank, uint32_t& tot, bool& refused) {
      const int64_t e0 = static_cast<int64_t>(chunk) * SG_CHUNK + static_cast<int64_t>(threadIdx.x) * SG_PER;
      const bool met_bad = sg_eval<(LDSB >= 1), WT>(pred, s_dyn, e0, t);
      uint32_t mine = met_bad ? 0x10000u : 0u;  // (survivors of a chunk fit 13 bits: the flag rides above)
#pragma unroll
      for (int j = 0; j < SG_PER; ++j) mine += t.keep[j] ? 1u : 0u;
      uint32_t total;
      rank = block_excl_scan_1024(mine, s_w, &total) & 0xFFFFu;
      tot = total & 0xFFFFu;
      refused = (total >> 16) != 0u;
      if (threadIdx.x == 0)
        sps_store(sg.status + 2 + chunk, sg.tag | (chunk == 0 ? SPS_PRE : SPS_AGG) | (refused ? 0x80000000ull : 0ull) | tot);
    };
    int chunk = blockIdx.x;
    if (chunk < nchunks) {
      SgEdges cur = nxt;
      if (chunk + G < nchunks)
        sg_fetch(pred, static_cast<int64_t>(chunk + G) * SG_CHUNK + static_cast<int64_t>(threadIdx.x) * SG_PER, E, nxt);
      uint32_t rank_c, tot_c;
      bool ref_c;
      eval_publish(chunk, cur, rank_c, tot_c, ref_c);
      for (;;) {
        const bool has_next = chunk + G < nchunks;
        SgEdges tn;
        uint32_t rank_n = 0, tot_n = 0;
        bool ref_n = false;
        // the words of `chunk`'s predecessors (published one chunk ago, like its own) are requested now and consumed
        // behind the evaluation of the next chunk: their ~1 us round trip is not on the critical path
        SpsLook look;
        if (wave_id() == 0 && chunk > 0) sps_lookback_issue(sg.status, chunk, sg.tag, look);
        if (has_next) {
          tn = nxt;
          if (chunk + 2 * G < nchunks)
            sg_fetch(pred, static_cast<int64_t>(chunk + 2 * G) * SG_CHUNK + static_cast<int64_t>(threadIdx.x) * SG_PER, E,
                     nxt);
          eval_publish(chunk + G, tn, rank_n, tot_n, ref_n);
        }
        if (wave_id() == 0) {  // the prefix of `chunk`
          const int lane = lane_id();
          uint32_t excl = 0;
          bool refused = ref_c;
#ifdef TGP_GEMM_STAMPS  // diagnostic build: TGP_SG_ABLATE=1 skips the wait (wrong offsets, timing only)
          if (chunk > 0 && (pred.flags & (1 << 30))) {
            excl = static_cast<uint32_t>(chunk) * 1024u;
          } else
#endif
          if (chunk > 0) {
            bool before = false;
            sps_lookback_finish(sg.status, chunk, sg.tag, look, &excl, &before);
            refused = refused || before;
            if (lane == 0)
              sps_store(sg.status + 2 + chunk, sg.tag | SPS_PRE | (refused ? 0x80000000ull : 0ull) |
                                                   static_cast<unsigned long long>((excl + tot_c) & 0x7FFFFFFFu));
          }
          if (lane == 0) {
            s_base = excl;
            if (chunk == nchunks - 1)
              __hip_atomic_store(sg.result, sg.tag | (refused ? 0x80000000ull : 0ull) |
                                                static_cast<unsigned long long>((excl + tot_c) & 0x7FFFFFFFu),
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        }
        __syncthreads();
        int64_t pos = static_cast<int64_t>(s_base) + rank_c;
        const int64_t e0 = static_cast<int64_t>(chunk) * SG_CHUNK + static_cast<int64_t>(threadIdx.x) * SG_PER;
#pragma unroll
        for (int j = 0; j < SG_PER; ++j) {
          if (cur.keep[j]) {
            sg.out_row[pos] = pred.relabel ? static_cast<int64_t>(new_id(cur.r[j])) : cur.r[j];
            sg.out_col[pos] = pred.relabel ? static_cast<int64_t>(new_id(cur.c[j])) : cur.c[j];
            if (sg.out_w) sg.out_w[pos] = cur.w[j];
            if (sg.out_eid) sg.out_eid[pos] = e0 + j;
            ++pos;
          }
        }
        if (!has_next) break;
        cur = tn;
        rank_c = rank_n;
        tot_c = tot_n;
        ref_c = ref_n;
        chunk += G;
      }
    }
  }
}

// Pass 2: staged survivors of chunk c (counts[c] entries at c * SG_CHUNK) -> outputs at offsets[c]
__global__ __launch_bounds__(256) void subgraph_copy_kernel(SgStage st, const uint32_t* __restrict__ counts,
                                                            const uint32_t* __restrict__ offsets, int nchunks,
                                                            int64_t* __restrict__ out_row, int64_t* __restrict__ out_col,
                                                            float* __restrict__ out_w, int64_t* __restrict__ out_eid) {
  for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const uint32_t n = counts[chunk];
    const int64_t dst = offsets[chunk], src = static_cast<int64_t>(chunk) * SG_CHUNK;
    for (uint32_t t = threadIdx.x; t < n; t += 256) {
      out_row[dst + t] = st.r[src + t];
      out_col[dst + t] = st.c[src + t];
      if (out_w) out_w[dst + t] = st.w[src + t];
      if (out_eid) out_eid[dst + t] = src + st.off[src + t];  // (staged only under TGP_WANT_EDGE_ID)
    }
  }
}

// =====================================================================================
// A4: relabel by cluster + coalesce (connect/base_conn.py:83-89)
// =====================================================================================
// int64 cluster ids -> int32 table: half the bytes, so the 2*E random gathers below hit a table that
// fits the per-XCD L2 for graphs up to ~1M nodes.
__global__ __launch_bounds__(256) void cluster_table_kernel(const int64_t* __restrict__ cluster, int64_t n,
                                                            int32_t* __restrict__ table) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) table[i] = static_cast<int32_t>(cluster[i]);
}

// arithmetic of the weight type VT (float; double for model.double() inputs, r4): products / sums rounded one by one
__device__ __forceinline__ float vt_add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ double vt_add(double a, double b) { return __dadd_rn(a, b); }
__device__ __forceinline__ float vt_mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ double vt_mul(double a, double b) { return __dmul_rn(a, b); }
template <typename VT>
__device__ __forceinline__ VT vt_reduce(VT acc, VT v, int op) {
  switch (op) {
    case TGP_MIN: return fmin(acc, v);
    case TGP_MAX: return fmax(acc, v);
    case TGP_MUL: return vt_mul(acc, v);
    default: return vt_add(acc, v);
  }
}

template <typename VT>
__global__ __launch_bounds__(256) void coalesce_keys_kernel(const int64_t* __restrict__ row,
                                                            const int64_t* __restrict__ col,
                                                            const VT* __restrict__ w,
                                                            const int32_t* __restrict__ cluster, int64_t E,
                                                            int64_t n, uint64_t K, uint64_t* __restrict__ keys,
                                                            VT* __restrict__ vals, int* __restrict__ bad_ids) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (e < E) {
    const int64_t r = row[e], c = col[e];
    uint64_t key = 0;
    if (static_cast<uint64_t>(r) >= static_cast<uint64_t>(n) || static_cast<uint64_t>(c) >= static_cast<uint64_t>(n)) {
      *bad_ids = 1;  // node id outside [0, n): the reference's cluster_index[edge_index] would raise
    } else {
      const uint64_t cr = static_cast<uint32_t>(cluster[r]), cc = static_cast<uint32_t>(cluster[c]);
      if (cr >= K || cc >= K) *bad_ids = 1;  // cluster id outside [0, K)
      else key = cr * K + cc;
    }
    keys[e] = key;
    vals[e] = w ? w[e] : VT(1);
  }
}

// Head of every run of equal keys reduces its run in sorted (= stable input) order, decides
// whether the merged edge survives the filters, and the block counts survivors.
// seg[i] holds the merged weight at run heads; keepflag[i] = 1 at surviving heads.
template <typename VT>
__global__ __launch_bounds__(256) void coalesce_segment_kernel(const uint64_t* __restrict__ keys,
                                                               const VT* __restrict__ vals, int64_t E,
                                                               uint64_t K, int has_weight, int reduce_op,
                                                               int flags, VT eps, VT* __restrict__ seg,
                                                               uint8_t* __restrict__ keepflag,
                                                               uint32_t* __restrict__ block_counts) {
  __shared__ uint32_t s_cnt[4];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kCompactTile;
  uint32_t mine = 0;
#pragma unroll
  for (int it = 0; it < kCompactItems; ++it) {
    const int64_t i = base + it * 256 + threadIdx.x;
    bool keep = false;
    if (i < E) {
      const uint64_t key = keys[i];
      const bool head = (i == 0) || keys[i - 1] != key;
      if (head) {
        keep = true;
        const uint64_t r = key / K, c = key - r * K;
        if ((flags & TGP_REMOVE_SELF_LOOPS) && r == c) keep = false;
        if (has_weight) {
          VT acc = vals[i];
          int64_t n = 1;
          for (int64_t j = i + 1; j < E && keys[j] == key; ++j, ++n) acc = vt_reduce(acc, vals[j], reduce_op);
          if (reduce_op == TGP_MEAN) acc = acc / static_cast<VT>(n);
          seg[i] = acc;
          if ((flags & TGP_EPS_FILTER) && !(fabs(acc) > eps)) keep = false;
        }
      }
      keepflag[i] = keep ? 1 : 0;
    }
    mine += __popcll(__ballot(keep));
  }
  if (lane_id() == 0) s_cnt[wave_id()] = mine;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

template <typename VT>
__global__ __launch_bounds__(256) void coalesce_fill_kernel(const uint64_t* __restrict__ keys,
                                                            const VT* __restrict__ seg,
                                                            const uint8_t* __restrict__ keepflag, int64_t E,
                                                            uint64_t K, const uint32_t* __restrict__ block_offsets,
                                                            int64_t* __restrict__ out_row,
                                                            int64_t* __restrict__ out_col,
                                                            VT* __restrict__ out_w) {
  __shared__ uint32_t s_cnt[kCompactItems * 4];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kCompactTile;
  bool keep[kCompactItems];
  uint32_t rank[kCompactItems];
#pragma unroll
  for (int it = 0; it < kCompactItems; ++it) {
    const int64_t i = base + it * 256 + threadIdx.x;
    keep[it] = i < E && keepflag[i] != 0;
  }
  uint32_t total;
  block_compact_ranks<kCompactItems>(keep, rank, total, s_cnt);
  const int64_t off = block_offsets[blockIdx.x];
#pragma unroll
  for (int it = 0; it < kCompactItems; ++it) {
    if (keep[it]) {
      const int64_t i = base + it * 256 + threadIdx.x;
      const uint64_t key = keys[i];
      const uint64_t r = key / K;
      out_row[off + rank[it]] = static_cast<int64_t>(r);
      out_col[off + rank[it]] = static_cast<int64_t>(key - r * K);
      if (out_w) out_w[off + rank[it]] = seg[i];
    }
  }
}

// =====================================================================================
// A6 normalisations (utils/ops.py:383-417) on a pooled edge list, in place
// =====================================================================================
__global__ __launch_bounds__(256) void fill_f32_kernel(float* p, int64_t n, float v) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}

// Degree of the pooled graph.  When the edge list is sorted by row (coalesce output, PyG edge
// lists) the first edge of each row sums its run in list order -- deterministic and in the
// order of the CPU scatter_add_.  An unsorted list falls back to float atomics.
__global__ __launch_bounds__(256) void check_sorted_rows_kernel(const int64_t* __restrict__ row, int64_t E,
                                                                int* __restrict__ unsorted) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (e + 1 < E && row[e] > row[e + 1]) *unsorted = 1;
}

// Sorted rows: segmented sums without atomics.  One wave owns a slab of kDegSlab consecutive edges
// and walks it in 64-edge chunks: a segmented inclusive scan over the lanes (rows are sorted, so
// "same row `off` lanes back" implies the whole span is that row) plus a carry from the previous
// chunk.  A run that lies inside the slab is written by its last edge.  Runs cut by a slab boundary
// leave partial sums in head[] / tail[] and degree_stitch_kernel adds them up in slab order, so the
// result is deterministic.  Skipped (no-op) when *unsorted is set.
constexpr int kDegChunks = 16;
constexpr int kDegSlab = 64 * kDegChunks;
constexpr int kDegEndsLeft = 1;   // the slab's leading run started in an earlier slab and ends here
constexpr int kDegThrough = 2;    // the whole slab is the middle of one run
__global__ __launch_bounds__(256) void degree_sorted_kernel(const int64_t* __restrict__ row,
                                                            const float* __restrict__ w, int64_t E,
                                                            const int* __restrict__ unsorted,
                                                            float* __restrict__ deg, float* __restrict__ head,
                                                            float* __restrict__ tail, int* __restrict__ slab_flags) {
  if (*unsorted) return;
  const int64_t slab = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int64_t start = slab * kDegSlab;
  if (start >= E) return;
  const int64_t row_first = row[start];
  const bool open_left = start > 0 && row[start - 1] == row_first;
  float carry_sum = 0.f;
  int64_t carry_row = -1;
  int flags = 0;
  for (int c = 0; c < kDegChunks; ++c) {
    const int64_t e = start + c * 64 + lane;
    if (start + c * 64 >= E) break;  // wave-uniform
    const bool valid = e < E;
    const int64_t r = valid ? row[e] : -2;
    const int64_t rn = (e + 1 < E) ? row[e + 1] : -3;
    float v = valid ? w[e] : 0.f;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const float pv = __shfl_up(v, off);
      const int64_t pr = __shfl_up(r, off);
      if (lane >= off && pr == r) v = __fadd_rn(pv, v);
    }
    if (r == carry_row) v = __fadd_rn(carry_sum, v);
    if (valid && rn != r) {
      if (open_left && r == row_first) {
        head[slab] = v;
        flags = kDegEndsLeft;
      } else {
        deg[r] = v;
      }
    }
    carry_sum = __shfl(v, 63);
    carry_row = __shfl(r, 63);
  }
  // The run still open at the slab's right edge (carry_row == -2 when the list ended inside it).
  const int64_t end = start + kDegSlab;
  const bool open_right = end < E && row[end] == carry_row;
  if (open_right) {
    if (open_left && carry_row == row_first) {
      head[slab] = carry_sum;
      flags = kDegThrough;
    } else {
      tail[slab] = carry_sum;
    }
  }
  if (__any(flags & kDegEndsLeft)) flags |= kDegEndsLeft;
  if (lane == 0) slab_flags[slab] = flags;
}

// One wave per slab whose leading run ends there: walk back over the "through" slabs to the slab
// the run started in and add the partial sums (fixed reduction tree -> deterministic).
__global__ __launch_bounds__(256) void degree_stitch_kernel(const int64_t* __restrict__ row, int64_t E,
                                                            const int* __restrict__ unsorted,
                                                            const float* __restrict__ head,
                                                            const float* __restrict__ tail,
                                                            const int* __restrict__ slab_flags,
                                                            float* __restrict__ deg) {
  if (*unsorted) return;
  const int64_t slab = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (slab * kDegSlab >= E || !(slab_flags[slab] & kDegEndsLeft)) return;
  float acc = 0.f;
  int64_t t = slab - 1;
  for (;;) {
    const int64_t mine = t - lane;
    const bool through = mine >= 0 && (slab_flags[mine] & kDegThrough);
    const unsigned long long stop = __ballot(!through);
    const int first_stop = stop ? __ffsll(static_cast<long long>(stop)) - 1 : 64;
    float part = 0.f;
    if (lane < first_stop) part = head[mine];
    else if (lane == first_stop) part = tail[mine];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part = __fadd_rn(part, __shfl_xor(part, off));
    acc = __fadd_rn(part, acc);
    if (stop) break;
    t -= 64;
  }
  if (lane == 0) deg[row[slab * kDegSlab]] = __fadd_rn(acc, head[slab]);
}

__global__ __launch_bounds__(256) void degree_fallback_kernel(const int64_t* __restrict__ row,
                                                              const float* __restrict__ w, int64_t E,
                                                              const int* __restrict__ unsorted,
                                                              float* __restrict__ deg) {
  if (!*unsorted) return;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; e < E;
       e += static_cast<int64_t>(gridDim.x) * 256)
    atomicAdd(&deg[row[e]], w[e]);
}

__global__ __launch_bounds__(256) void degree_scale_kernel(const int64_t* __restrict__ row,
                                                           const int64_t* __restrict__ col,
                                                           float* __restrict__ w, int64_t E,
                                                           const float* __restrict__ deg, float eps) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (e < E) {
    // deg.clamp(min=eps).pow(-0.5); w * dis[row] * dis[col]  (ops.py:395-401)
    const float dr = 1.0f / sqrtf(fmaxf(deg[row[e]], eps));
    const float dc = 1.0f / sqrtf(fmaxf(deg[col[e]], eps));
    w[e] = __fmul_rn(__fmul_rn(w[e], dr), dc);
  }
}

// |w| is non-negative, so its IEEE bit pattern orders like an unsigned int: exact, order-free max.
// A batch has few graphs, so one atomic per edge would serialise on a handful of addresses
// (113 ms at E = 10M, 8 graphs).  Each thread instead walks a contiguous slice of the list keeping a
// running (graph, max) pair -- rows are normally sorted, so the graph id changes a few times per
// slice at most -- and a wave whose lanes all ended on the same graph issues a single atomic.
constexpr int kGraphMaxPerThread = 16;
__global__ __launch_bounds__(256) void graph_max_kernel(const int64_t* __restrict__ row,
                                                        const float* __restrict__ w, int64_t E,
                                                        const int64_t* __restrict__ batch_pooled,
                                                        uint32_t* __restrict__ gmax) {
  __shared__ int64_t s_g[4];
  __shared__ uint32_t s_m[4];
  // Lane-interleaved inside a wave-sized slab so loads stay coalesced.
  const int wave_in_block = threadIdx.x >> 6;
  const int64_t wave = static_cast<int64_t>(blockIdx.x) * 4 + wave_in_block;
  const int lane = threadIdx.x & 63;
  const int64_t base = wave * (64 * kGraphMaxPerThread) + lane;
  // All loads first (independent), then the running (graph, max) fold.
  int64_t g[kGraphMaxPerThread];
  uint32_t v[kGraphMaxPerThread];
#pragma unroll
  for (int i = 0; i < kGraphMaxPerThread; ++i) {
    const int64_t e = base + static_cast<int64_t>(i) * 64;
    g[i] = e < E ? row[e] : -1;
    v[i] = e < E ? __float_as_uint(fabsf(w[e])) : 0u;
  }
#pragma unroll
  for (int i = 0; i < kGraphMaxPerThread; ++i) g[i] = g[i] >= 0 ? batch_pooled[g[i]] : -1;
  int64_t g_cur = -1;
  uint32_t m_cur = 0;
#pragma unroll
  for (int i = 0; i < kGraphMaxPerThread; ++i) {
    if (g[i] < 0) continue;
    if (g[i] != g_cur) {
      if (g_cur >= 0 && m_cur > gmax[g_cur]) atomicMax(&gmax[g_cur], m_cur);
      g_cur = g[i];
      m_cur = v[i];
    } else {
      m_cur = m_cur > v[i] ? m_cur : v[i];
    }
  }
  // Wave, then block: when everyone ended on the same graph a single atomic covers 4096 edges.
  // gmax only grows, so a (possibly stale) plain read that already covers the candidate lets us
  // skip the atomic altogether.
  const int64_t g0 = __shfl(g_cur, 0);
  const bool wave_uniform = __all(g_cur == g0 || g_cur < 0);
  if (wave_uniform) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const uint32_t o = __shfl_xor(m_cur, off);
      m_cur = m_cur > o ? m_cur : o;
    }
  } else if (g_cur >= 0 && m_cur > gmax[g_cur]) {
    atomicMax(&gmax[g_cur], m_cur);
  }
  if (lane == 0) {
    s_g[wave_in_block] = wave_uniform ? g0 : -1;
    s_m[wave_in_block] = wave_uniform ? m_cur : 0u;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const bool block_uniform = s_g[0] >= 0 && (s_g[1] == s_g[0] || s_g[1] < 0) &&
                               (s_g[2] == s_g[0] || s_g[2] < 0) && (s_g[3] == s_g[0] || s_g[3] < 0);
    if (block_uniform) {
      uint32_t m = s_m[0];
      for (int k = 1; k < 4; ++k) m = m > s_m[k] ? m : s_m[k];
      if (m > gmax[s_g[0]]) atomicMax(&gmax[s_g[0]], m);
    } else {
      for (int k = 0; k < 4; ++k)
        if (s_g[k] >= 0 && s_m[k] > gmax[s_g[k]]) atomicMax(&gmax[s_g[k]], s_m[k]);
    }
  }
}

__global__ __launch_bounds__(256) void graph_max_scale_kernel(const int64_t* __restrict__ row,
                                                              float* __restrict__ w, int64_t E,
                                                              const int64_t* __restrict__ batch_pooled,
                                                              const uint32_t* __restrict__ gmax) {
  const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (e < E) {
    float m = __uint_as_float(gmax[batch_pooled[row[e]]]);
    if (m == 0.f) m = 1.f;
    w[e] = w[e] / m;
  }
}

// =====================================================================================
// A10: dense [B,K,K] -> block-diagonal edge list (utils/ops.py:53-82, src.py:526-552)
// =====================================================================================
template <typename VT>
struct BlockDiagPredT {
  const VT* adj;
  const int64_t* relabel;  // [B*K] new id or -1; nullptr = identity
  int64_t K;
  int flags;
  VT eps;
  __device__ __forceinline__ bool operator()(int64_t i, int64_t& r, int64_t& c, VT& v) const {
    v = adj[i];
    const int64_t kk = K * K;
    const int64_t b = i / kk, rem = i - b * kk;
    const int64_t rr = rem / K;
    r = b * K + rr;
    c = b * K + (rem - rr * K);
    if (!(fabs(v) > eps)) return false;
    if (relabel) {
      r = relabel[r];
      c = relabel[c];
      if ((r | c) < 0) return false;
    }
    if ((flags & TGP_REMOVE_SELF_LOOPS) && r == c) return false;
    return true;
  }
};

using BlockDiagPred = BlockDiagPredT<float>;

template <typename VT>
__global__ __launch_bounds__(256) void blockdiag_count_kernel(BlockDiagPredT<VT> pred, int64_t total,
                                                              uint32_t* __restrict__ block_counts) {
  __shared__ uint32_t s_cnt[4];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kCompactTile;
  uint32_t mine = 0;
#pragma unroll
  for (int it = 0; it < kCompactItems; ++it) {
    const int64_t i = base + it * 256 + threadIdx.x;
    int64_t r, c;
    VT v;
    const bool keep = i < total && pred(i, r, c, v);
    mine += __popcll(__ballot(keep));
  }
  if (lane_id() == 0) s_cnt[wave_id()] = mine;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

template <typename VT>
__global__ __launch_bounds__(256) void blockdiag_fill_kernel(BlockDiagPredT<VT> pred, int64_t total,
                                                             const uint32_t* __restrict__ block_offsets,
                                                             int64_t* __restrict__ out_row,
                                                             int64_t* __restrict__ out_col,
                                                             VT* __restrict__ out_w) {
  __shared__ uint32_t s_cnt[kCompactItems * 4];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kCompactTile;
  bool keep[kCompactItems];
  int64_t r[kCompactItems], c[kCompactItems];
  VT v[kCompactItems];
  uint32_t rank[kCompactItems];
#pragma unroll
  for (int it = 0; it < kCompactItems; ++it) {
    const int64_t i = base + it * 256 + threadIdx.x;
    keep[it] = i < total && pred(i, r[it], c[it], v[it]);
  }
  uint32_t tot;
  block_compact_ranks<kCompactItems>(keep, rank, tot, s_cnt);
  const int64_t off = block_offsets[blockIdx.x];
#pragma unroll
  for (int it = 0; it < kCompactItems; ++it) {
    if (keep[it]) {
      out_row[off + rank[it]] = r[it];
      out_col[off + rank[it]] = c[it];
      out_w[off + rank[it]] = v[it];
    }
  }
}


// =====================================================================================
// A7' helper: CSR SpMM  T[i,:] = sum_e w[e] * S[col[e],:]   (connect/dense_conn.py:165,204)
// one group of G lanes per row, float4 per lane; rows are summed in CSR (= sorted) order
// =====================================================================================
__global__ __launch_bounds__(256) void rowptr_from_sorted_kernel(const int64_t* __restrict__ rows, int64_t n,
                                                                 int64_t num_rows, int32_t* __restrict__ row_ptr,
                                                                 int* __restrict__ unsorted = nullptr) {
  // Ids are CLAMPED to [-1, num_rows]: a list with ids outside [0, num_rows) never writes outside row_ptr[0 ..
  // num_rows]; it shows as row_ptr[0] != 0 (negative ids in front) or row_ptr[num_rows] != n (ids >= num_rows behind),
  // which the consumers check before they trust the offsets.
  const int64_t p = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (p > n) return;
  auto clamp = [num_rows](int64_t v) { return v < -1 ? -1 : (v > num_rows ? num_rows : v); };
  if (p == n) {
    const int64_t first = n > 0 ? clamp(rows[n - 1]) + 1 : 0;
    for (int64_t c = first; c <= num_rows; ++c) row_ptr[c] = static_cast<int32_t>(n);
    return;
  }
  const int64_t cur = clamp(rows[p]);
  const int64_t prev = p > 0 ? clamp(rows[p - 1]) : -1;
  for (int64_t c = prev + 1; c <= cur && c <= num_rows; ++c) row_ptr[c] = static_cast<int32_t>(p);
  // (a list that is NOT sorted still gets every entry of row_ptr written, with values in [0, n]: any id is crossed by
  //  an ascent, or lies below the first / above the last row -- offsets a kernel can walk without leaving the arrays)
  if (unsorted && p > 0 && rows[p] < rows[p - 1]) *unsorted = 1;
}

__global__ __launch_bounds__(256) void spmm_csr_kernel(const int32_t* __restrict__ row_ptr,
                                                       const int64_t* __restrict__ col,
                                                       const float* __restrict__ w, int64_t num_rows,
                                                       const float* __restrict__ S, int64_t K,
                                                       float* __restrict__ T) {
  const int64_t total = num_rows * K;
  for (int64_t o = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; o < total;
       o += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t i = o / K, f = o - i * K;
    float acc = 0.f;
    for (int32_t e = row_ptr[i]; e < row_ptr[i + 1]; ++e)
      acc = __fadd_rn(acc, __fmul_rn(w ? w[e] : 1.0f, S[col[e] * K + f]));
    T[o] = acc;
  }
}


// r6: T = A S for the dense poolers' rows route (K >= 64 floats per row of S, every row of S read by ~deg rows of A).
// The sparse Reduce's gather-sum (tgp_reduce_sparse_f32, the r5 route) streams its rows with non-temporal loads and
// hands consecutive row chunks to consecutive workgroups, i.e. round-robin over the eight XCDs: right for a Reduce (every
// row read once), wrong here -- the rows of one graph re-read that graph's block of S (N_g x K floats), and spread over
// all XCDs every private L2 sees all of S (16.8 MB at C2 against 4 MB of L2).  This kernel keeps a contiguous range of
// rows -- whole graphs of a sorted batch -- on ONE XCD (the guide's bijective T1 remap of blockIdx: blocks with equal
// blockIdx % 8 share an XCD; a speed choice only), loads S with the default cache policy, and has four row gathers of
// a lane in flight.  Same products, same order of adds as spmm_csr_kernel / the Reduce route: bit-identical.
// STATS (MinCut's degree term, utils/losses.py:73-127): also deg[i] = the row's weight sum (entry count without weights)
// and q[i] = |S_i|^2 -- the row's entries are in hand anyway, S_i is 16 more bytes per lane (tgp_edge_row_stats_f32 was a
// launch of its own for these).
// STATS == 2 (DiffPool's entropy loss, utils/losses.py:476-483): deg[logical block] = the block's share of
// sum(-S log(S + eps)) over its rows of S (q unused): the loss' pass over S rides along as well.
template <int G, int STATS>
__global__ __launch_bounds__(256) void spmm_rows_vec4_kernel(const int32_t* __restrict__ row_ptr,
                                                             const int64_t* __restrict__ col,
                                                             const float* __restrict__ w, int64_t num_rows,
                                                             const float* __restrict__ S, int64_t K,
                                                             float* __restrict__ T, int rows_per_block,
                                                             float* __restrict__ deg, float* __restrict__ qout,
                                                             float ent_eps) {
  constexpr int RPB = 256 / G;
  const int nwg = static_cast<int>(gridDim.x), orig = static_cast<int>(blockIdx.x);
  const int xcd = orig % 8, qq = nwg / 8, rr = nwg % 8;
  const int bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + orig / 8;
  const int g = threadIdx.x % G, sub = threadIdx.x / G;
  const int64_t r0 = static_cast<int64_t>(bid) * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > num_rows) r1 = num_rows;
  [[maybe_unused]] float esum = 0.f;
  for (int64_t i = r0 + sub; i < r1; i += RPB) {
    const int32_t beg = row_ptr[i], end = row_ptr[i + 1];
    [[maybe_unused]] float dsum = 0.f, qsum = 0.f;
    for (int64_t f = 4 * g; f < K; f += 4 * G) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (STATS == 1) {
        const float4 si = *reinterpret_cast<const float4*>(S + i * K + f);
        qsum = fmaf(si.x, si.x, qsum); qsum = fmaf(si.y, si.y, qsum);
        qsum = fmaf(si.z, si.z, qsum); qsum = fmaf(si.w, si.w, qsum);
      }
      if constexpr (STATS == 2) {
        const float4 si = *reinterpret_cast<const float4*>(S + i * K + f);
        esum += (-si.x * logf(si.x + ent_eps) - si.y * logf(si.y + ent_eps)) +
                (-si.z * logf(si.z + ent_eps) - si.w * logf(si.w + ent_eps));
      }
      for (int32_t e0 = beg; e0 < end; e0 += 4) {
        int64_t c[4];
        float wv[4];
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int32_t e = e0 + q < end ? e0 + q : end - 1;  // (clamped: what it returns is dropped below)
          c[q] = col[e];
          wv[q] = w ? w[e] : 1.0f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float4*>(S + c[q] * K + f);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool on = e0 + q < end;  // product rounded before the add, as the reference's two-step form
          const float px = __fadd_rn(acc.x, __fmul_rn(wv[q], v[q].x)), py = __fadd_rn(acc.y, __fmul_rn(wv[q], v[q].y));
          const float pz = __fadd_rn(acc.z, __fmul_rn(wv[q], v[q].z)), pw = __fadd_rn(acc.w, __fmul_rn(wv[q], v[q].w));
          acc.x = on ? px : acc.x;
          acc.y = on ? py : acc.y;
          acc.z = on ? pz : acc.z;
          acc.w = on ? pw : acc.w;
          if constexpr (STATS == 1) {
            if (f == 4 * g && on) dsum += wv[q];  // (every lane of the group holds the same sum)
          }
        }
      }
      *reinterpret_cast<float4*>(T + i * K + f) = acc;
    }
    if constexpr (STATS == 1) {
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) qsum += __shfl_xor(qsum, o, 64);
      if (g == 0) {
        deg[i] = dsum;
        qout[i] = qsum;
      }
    }
  }
  if constexpr (STATS == 2) {
    __shared__ float s_e[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) esum += __shfl_xor(esum, o, 64);
    if ((threadIdx.x & 63) == 0) s_e[threadIdx.x >> 6] = esum;
    __syncthreads();
    if (threadIdx.x == 0) deg[bid] = (s_e[0] + s_e[1]) + (s_e[2] + s_e[3]);
  }
}

}  // namespace tgp

using namespace tgp;

// ------------------------------------------------------------------------------------- subgraph
struct SubgraphWs {
  int32_t* relabel;
  uint32_t* member_bits;
  int* unsorted;
  uint32_t* counts;
  uint32_t* offsets;
  SgStage st;
};
static size_t subgraph_layout(void* ws, int64_t E, int64_t N, SubgraphWs* out) {
  Carver cv(ws);
  const size_t nb = static_cast<size_t>(cdiv(E > 0 ? E : 1, SG_CHUNK));
  const size_t cap = nb * SG_CHUNK;  // staging capacity: every chunk compacts at its own base
  SubgraphWs s;
  s.relabel = cv.take<int32_t>(N > 0 ? N : 1);
  s.member_bits = cv.take<uint32_t>((N > 0 ? N : 1) / 32 + 1);
  s.unsorted = cv.take<int>(4);
  s.counts = cv.take<uint32_t>(nb);
  s.offsets = cv.take<uint32_t>(nb);
  s.st.r = cv.take<int32_t>(cap);
  s.st.c = cv.take<int32_t>(cap);
  s.st.w = cv.take<float>(cap);
  s.st.off = cv.take<uint16_t>(cap);
  if (out) *out = s;
  return cv.off;
}
static SubgraphWs carve_subgraph(void* ws, int64_t E, int64_t N) {
  SubgraphWs s;
  subgraph_layout(ws, E, N, &s);
  return s;
}

extern "C" size_t tgp_connect_subgraph_workspace_bytes(int64_t E, int64_t N) {
  return subgraph_layout(nullptr, E, N, nullptr) + 512;
}

extern "C" int tgp_connect_subgraph_count(const int64_t* row, const int64_t* col, const float* w, int64_t E,
                                          const int64_t* node_index, int64_t k, int64_t N, int flags, float eps,
                                          void* ws, size_t ws_bytes, int64_t* d_count, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(E >= 0 && N >= 0 && k >= 0 && d_count && (E == 0 || (row && col)), TGP_ERR_INVALID,
              "tgp_connect_subgraph_count: bad argument");
  TGP_REQUIRE(E < (1ll << 31) && N < (1ll << 31), TGP_ERR_RANGE, "tgp_connect_subgraph_count: E/N >= 2^31");
  TGP_REQUIRE(ws && ws_bytes >= tgp_connect_subgraph_workspace_bytes(E, N), TGP_ERR_WORKSPACE,
              "tgp_connect_subgraph_count: workspace too small");
  SubgraphWs s = carve_subgraph(ws, E, N);
  const int nwords = static_cast<int>((N > 0 ? N : 1) / 32 + 1);
  (void)hipMemsetAsync(s.unsorted, 0, 4 * sizeof(int), stream);  // [0] node_index not ascending, [1] bad node ids
  if (node_index) {
    (void)hipMemsetAsync(s.member_bits, 0, static_cast<size_t>(nwords) * sizeof(uint32_t), stream);
    if (k > 0)
      hipLaunchKernelGGL(relabel_scatter_kernel, dim3(cdiv(k, 256)), dim3(256), 0, stream, node_index, k, N, s.relabel,
                         s.member_bits, s.unsorted);
    // (the rank directory of the relabel-by-rank route is computed inside the stage kernel from its LDS bitmap; the
    //  other routes relabel through the table and need none)
  }
  const int nb = cdiv(E > 0 ? E : 1, SG_CHUNK);
  SubgraphPred pred{row, col, w, node_index ? s.relabel : nullptr, s.member_bits, s.unsorted, flags, eps,
                    N, s.unsorted + 1, 1};
  SgStage st = s.st;
  if (!w) st.w = nullptr;
  if (!(flags & TGP_WANT_EDGE_ID)) st.off = nullptr;  // input positions are only staged for callers that will ask for them
  const int grid = nb < 256 ? nb : 256;  // persistent: one 1024-thread workgroup per CU
  const int nblocks = (nwords + 3) / 4;
  if (node_index && 5 * nblocks <= SG_LDS_WORDS_MAX + 1984) {  // bitmap (whole blocks) + rank128 <= 159.75 KB
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(subgraph_stage_kernel<2, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (SG_LDS_WORDS_MAX + 1984 + 4) * 4);
    hipLaunchKernelGGL((subgraph_stage_kernel<2, false>), dim3(grid), dim3(SG_THREADS),
                       (5 * nblocks + 4) * sizeof(uint32_t), stream, pred, E, nb, nwords, st, s.counts, SgSingle{});
  } else if (node_index && nwords <= SG_LDS_WORDS_MAX) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(subgraph_stage_kernel<1, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, SG_LDS_WORDS_MAX * 4);
    hipLaunchKernelGGL((subgraph_stage_kernel<1, false>), dim3(grid), dim3(SG_THREADS), nwords * sizeof(uint32_t), stream,
                       pred, E, nb, nwords, st, s.counts, SgSingle{});
  } else {
    hipLaunchKernelGGL((subgraph_stage_kernel<0, false>), dim3(grid), dim3(SG_THREADS), 0, stream, pred, E, nb, nwords,
                       st, s.counts, SgSingle{});
  }
  hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, stream, s.counts, nb, s.offsets, d_count,
                     static_cast<const int*>(s.unsorted + 1));
  return check_launch("tgp_connect_subgraph_count");
}

extern "C" int tgp_connect_subgraph_fill(const int64_t* /*row*/, const int64_t* /*col*/, const float* w, int64_t E,
                                         int64_t N, int flags, float /*eps*/, const void* ws, int64_t num_out,
                                         int64_t* out_row, int64_t* out_col, float* out_w, int64_t* out_edge_id,
                                         void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(E >= 0 && num_out >= 0 && ws, TGP_ERR_INVALID, "tgp_connect_subgraph_fill: bad argument");
  if (num_out == 0 || E == 0) return TGP_OK;
  TGP_REQUIRE(out_row && out_col && (!w || out_w), TGP_ERR_INVALID, "tgp_connect_subgraph_fill: null output");
  TGP_REQUIRE(!out_edge_id || (flags & TGP_WANT_EDGE_ID), TGP_ERR_INVALID,
              "tgp_connect_subgraph_fill: out_edge_id needs TGP_WANT_EDGE_ID in the flags of BOTH calls");
  SubgraphWs s = carve_subgraph(const_cast<void*>(ws), E, N);
  const int nb = cdiv(E, SG_CHUNK);
  hipLaunchKernelGGL(subgraph_copy_kernel, dim3(nb < 4096 ? nb : 4096), dim3(256), 0, stream, s.st, s.counts, s.offsets,
                     nb, out_row, out_col, w ? out_w : nullptr, out_edge_id);
  return check_launch("tgp_connect_subgraph_fill");
}

// ------------------------------------------------------------------------------------- subgraph, single pass (r4)
static size_t subgraph_single_layout(void* ws, int64_t N, SubgraphWs* out) {
  Carver cv(ws);
  SubgraphWs s{};
  s.relabel = cv.take<int32_t>(N > 0 ? N : 1);
  s.member_bits = cv.take<uint32_t>((N > 0 ? N : 1) / 32 + 1);
  s.unsorted = cv.take<int>(4);
  if (out) *out = s;
  return cv.off;
}

extern "C" size_t tgp_connect_subgraph_single_workspace_bytes(int64_t N) {
  return subgraph_single_layout(nullptr, N, nullptr) + 512;
}

extern "C" int64_t tgp_connect_subgraph_single_status_words(int64_t E) { return 2 + cdiv(E > 0 ? E : 1, SG_CHUNK); }

template <typename WT>
static int subgraph_single_impl(const int64_t* row, const int64_t* col, const WT* w, int64_t E,
                                const int64_t* node_index, int64_t k, int64_t N, int flags, WT eps, void* ws,
                                size_t ws_bytes, int64_t* out_row, int64_t* out_col, WT* out_w, int64_t* out_edge_id,
                                const uint32_t* member_bits_in, const uint32_t* rank128_in,
                                uint64_t* status, int64_t status_words, uint64_t* result, uint32_t epoch,
                                void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(E > 0 && N >= 0 && k >= 0 && row && col && out_row && out_col && (!w || out_w) && status && result,
              TGP_ERR_INVALID, "tgp_connect_subgraph_single: bad argument");
  TGP_REQUIRE(E < (1ll << 31) && N < (1ll << 31) && epoch != 0 && epoch < (1u << 29), TGP_ERR_RANGE,
              "tgp_connect_subgraph_single: E/N >= 2^31 or epoch out of range");
  TGP_REQUIRE(ws && ws_bytes >= tgp_connect_subgraph_single_workspace_bytes(N), TGP_ERR_WORKSPACE,
              "tgp_connect_subgraph_single: workspace too small");
  TGP_REQUIRE(status_words >= tgp_connect_subgraph_single_status_words(E), TGP_ERR_WORKSPACE,
              "tgp_connect_subgraph_single: status buffer too small");
  TGP_REQUIRE(!out_edge_id || (flags & TGP_WANT_EDGE_ID), TGP_ERR_INVALID,
              "tgp_connect_subgraph_single: out_edge_id needs TGP_WANT_EDGE_ID");
  SubgraphWs s;
  subgraph_single_layout(ws, N, &s);
  const int nwords = static_cast<int>((N > 0 ? N : 1) / 32 + 1);
  const int nblocks_dir = (nwords + 3) / 4;
  // r5: the selector's own bitmap + rank directory (TopkSelect writes them in its compaction pass): nothing to clear,
  // nothing to scatter, the stage kernel is the whole call
  const bool pre = node_index && member_bits_in && rank128_in && 5 * nblocks_dir <= SG_LDS_WORDS_MAX + 1984;
  // the membership bitmap and the four status words ([0] node_index not ascending, [1] unused since r5) sit next to each
  // other in the workspace: ONE memset (a memset is a launch of its own: ~4 us of this 120 us call each)
  if (pre) {
    // (no launch in front of the stage kernel)
  } else if (node_index) {
    (void)hipMemsetAsync(s.member_bits, 0,
                         static_cast<size_t>(reinterpret_cast<char*>(s.unsorted + 4) - reinterpret_cast<char*>(s.member_bits)),
                         stream);
    if (k > 0)
      hipLaunchKernelGGL(relabel_scatter_kernel, dim3(cdiv(k, 256)), dim3(256), 0, stream, node_index, k, N, s.relabel,
                         s.member_bits, s.unsorted);
  } else {
    (void)hipMemsetAsync(s.unsorted, 0, 4 * sizeof(int), stream);
  }
  const int nb = cdiv(E, SG_CHUNK);
#ifdef TGP_GEMM_STAMPS
  if (getenv("TGP_SG_ABLATE")) flags |= (1 << 30);
#endif
  // a thread that meets an endpoint outside [0, N) stores the call's epoch into the low half of word [1] of the caller's
  // status buffer (never cleared: a word of an earlier call holds another epoch); after a refused call the host tells
  // bad ids (the reference's index ops raise) from a look-back spin bound by that word, read with a copy on this stream
  SubgraphPredT<WT> pred{row, col, w, node_index ? s.relabel : nullptr, pre ? member_bits_in : s.member_bits, s.unsorted,
                         flags, eps, N, reinterpret_cast<int*>(status + 1), static_cast<int>(epoch)};
  SgSingleT<WT> sg{out_row, out_col, w ? out_w : nullptr, out_edge_id, reinterpret_cast<unsigned long long*>(status),
              reinterpret_cast<unsigned long long*>(result), static_cast<unsigned long long>(epoch) << SPS_EPOCH_SHIFT,
              pre ? rank128_in : nullptr};
  // persistent, every workgroup resident (the look-back waits for chunks of the same round): one per CU
  int cus = tgp_device_cu_count();
  if (cus <= 0) cus = 256;
  const int grid = nb < cus ? nb : cus;
  const int nblocks = (nwords + 3) / 4;
  if (pre) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(subgraph_stage_kernel<2, true, WT, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (SG_LDS_WORDS_MAX + 1984 + 4) * 4);
    hipLaunchKernelGGL((subgraph_stage_kernel<2, true, WT, true>), dim3(grid), dim3(SG_THREADS),
                       (5 * nblocks + 4) * sizeof(uint32_t), stream, pred, E, nb, nwords, SgStageT<WT>{}, nullptr, sg);
  } else if (node_index && 5 * nblocks <= SG_LDS_WORDS_MAX + 1984) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(subgraph_stage_kernel<2, true, WT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (SG_LDS_WORDS_MAX + 1984 + 4) * 4);
    hipLaunchKernelGGL((subgraph_stage_kernel<2, true, WT>), dim3(grid), dim3(SG_THREADS),
                       (5 * nblocks + 4) * sizeof(uint32_t), stream, pred, E, nb, nwords, SgStageT<WT>{}, nullptr, sg);
  } else if (node_index && nwords <= SG_LDS_WORDS_MAX) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(subgraph_stage_kernel<1, true, WT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, SG_LDS_WORDS_MAX * 4);
    hipLaunchKernelGGL((subgraph_stage_kernel<1, true, WT>), dim3(grid), dim3(SG_THREADS), nwords * sizeof(uint32_t), stream,
                       pred, E, nb, nwords, SgStageT<WT>{}, nullptr, sg);
  } else {
    hipLaunchKernelGGL((subgraph_stage_kernel<0, true, WT>), dim3(grid), dim3(SG_THREADS), 0, stream, pred, E, nb, nwords,
                       SgStageT<WT>{}, nullptr, sg);
  }
  return check_launch("tgp_connect_subgraph_single");
}

extern "C" int tgp_connect_subgraph_single(const int64_t* row, const int64_t* col, const float* w, int64_t E,
                                           const int64_t* node_index, int64_t k, int64_t N, int flags, float eps,
                                           void* ws, size_t ws_bytes, int64_t* out_row, int64_t* out_col, float* out_w,
                                           int64_t* out_edge_id, const uint32_t* member_bits_in,
                                           const uint32_t* rank128_in, uint64_t* status, int64_t status_words,
                                           uint64_t* result, uint32_t epoch, void* stream_) {
  return subgraph_single_impl<float>(row, col, w, E, node_index, k, N, flags, eps, ws, ws_bytes, out_row, out_col, out_w,
                                     out_edge_id, member_bits_in, rank128_in, status, status_words, result, epoch,
                                     stream_);
}

// float64 weights (model.double(): the reference's ATen ops keep them in fp64): they pass through and meet the
// |w| > eps test in double, eps as a double
extern "C" int tgp_connect_subgraph_single_f64(const int64_t* row, const int64_t* col, const double* w, int64_t E,
                                               const int64_t* node_index, int64_t k, int64_t N, int flags, double eps,
                                               void* ws, size_t ws_bytes, int64_t* out_row, int64_t* out_col,
                                               double* out_w, int64_t* out_edge_id, const uint32_t* member_bits_in,
                                               const uint32_t* rank128_in, uint64_t* status,
                                               int64_t status_words, uint64_t* result, uint32_t epoch, void* stream_) {
  return subgraph_single_impl<double>(row, col, w, E, node_index, k, N, flags, eps, ws, ws_bytes, out_row, out_col,
                                      out_w, out_edge_id, member_bits_in, rank128_in, status, status_words, result,
                                      epoch, stream_);
}

// ------------------------------------------------------------------------------------- coalesce
template <typename VT>
struct CoalesceWsT {
  uint64_t *k0, *k1;
  VT *v0, *v1;
  VT* seg;
  uint8_t* keep;
  uint32_t *counts, *offsets, *scratch;
  int32_t* table;
  int* bad_ids;
};
using CoalesceWs = CoalesceWsT<float>;
template <typename VT>
static size_t coalesce_layout(void* ws, int64_t E, int64_t N, CoalesceWsT<VT>* out) {
  Carver cv(ws);
  const size_t n = static_cast<size_t>(E > 0 ? E : 1);
  const size_t nb = static_cast<size_t>(cdiv(E > 0 ? E : 1, kCompactTile));
  CoalesceWsT<VT> s;
  s.k0 = cv.take<uint64_t>(n);
  s.k1 = cv.take<uint64_t>(n);
  s.v0 = cv.take<VT>(n);
  s.v1 = cv.take<VT>(n);
  s.seg = cv.take<VT>(n);
  s.keep = cv.take<uint8_t>(n);
  s.counts = cv.take<uint32_t>(nb);
  s.offsets = cv.take<uint32_t>(nb);
  s.scratch = cv.take<uint32_t>(sort_scratch_words());
  s.table = cv.take<int32_t>(static_cast<size_t>(N > 0 ? N : 1));
  s.bad_ids = cv.take<int>(4);
  if (out) *out = s;
  return cv.off;
}

extern "C" size_t tgp_connect_coalesce_workspace_bytes(int64_t E, int64_t N, int64_t /*K*/) {
  return coalesce_layout<float>(nullptr, E, N, nullptr) + 256;
}
extern "C" size_t tgp_connect_coalesce_workspace_bytes_f64(int64_t E, int64_t N, int64_t /*K*/) {
  return coalesce_layout<double>(nullptr, E, N, nullptr) + 256;
}

template <typename VT>
static int coalesce_count_impl(const int64_t* row, const int64_t* col, const VT* w, int64_t E,
                               const int64_t* cluster_index, int64_t N, int64_t K, int reduce_op, int flags, VT eps,
                               void* ws, size_t ws_bytes, int64_t* d_count, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(E >= 0 && N >= 0 && K >= 0 && d_count && (E == 0 || (row && col && cluster_index)),
              TGP_ERR_INVALID, "tgp_connect_coalesce_count: bad argument");
  TGP_REQUIRE(reduce_op >= TGP_SUM && reduce_op <= TGP_MUL, TGP_ERR_INVALID,
              "tgp_connect_coalesce_count: unknown reduce_op %d", reduce_op);
  TGP_REQUIRE(E < (1ll << 31) && K < (1ll << 31) && N < (1ll << 31), TGP_ERR_RANGE,
              "tgp_connect_coalesce_count: E/N/K >= 2^31");
  TGP_REQUIRE(ws && ws_bytes >= coalesce_layout<VT>(nullptr, E, N, nullptr) + 256, TGP_ERR_WORKSPACE,
              "tgp_connect_coalesce_count: workspace too small");
  CoalesceWsT<VT> s;
  coalesce_layout<VT>(ws, E, N, &s);
  if (E == 0) {
    (void)hipMemsetAsync(d_count, 0, sizeof(int64_t), stream);
    return check_launch("tgp_connect_coalesce_count");
  }
  const uint64_t Ku = static_cast<uint64_t>(K > 0 ? K : 1);
  hipLaunchKernelGGL(cluster_table_kernel, dim3(cdiv(N, 256)), dim3(256), 0, stream, cluster_index, N, s.table);
  int* bad_ids = s.bad_ids;
  (void)hipMemsetAsync(bad_ids, 0, sizeof(int), stream);
  hipLaunchKernelGGL(coalesce_keys_kernel<VT>, dim3(cdiv(E, 256)), dim3(256), 0, stream, row, col, w, s.table, E, N,
                     Ku, s.k0, s.v0, bad_ids);
  bool first = true;
  const int rc = radix_sort_pairs<uint64_t, VT>(s.k0, s.v0, s.k1, s.v1, E, bits_for(Ku * Ku - 1), s.scratch, stream,
                                                &first);
  if (rc != TGP_OK) return rc;
  const int nb = cdiv(E, kCompactTile);
  hipLaunchKernelGGL(coalesce_segment_kernel<VT>, dim3(nb), dim3(256), 0, stream, first ? s.k0 : s.k1,
                     first ? s.v0 : s.v1, E, Ku, w ? 1 : 0, reduce_op, flags, eps, s.seg, s.keep, s.counts);
  hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, stream, s.counts, nb, s.offsets, d_count,
                     static_cast<const int*>(bad_ids));
  return check_launch("tgp_connect_coalesce_count");
}

extern "C" int tgp_connect_coalesce_count(const int64_t* row, const int64_t* col, const float* w, int64_t E,
                                          const int64_t* cluster_index, int64_t N, int64_t K, int reduce_op,
                                          int flags, float eps, void* ws, size_t ws_bytes, int64_t* d_count,
                                          void* stream_) {
  return coalesce_count_impl<float>(row, col, w, E, cluster_index, N, K, reduce_op, flags, eps, ws, ws_bytes, d_count,
                                    stream_);
}
// float64 weights: merged in fp64 (products / sums rounded one by one, input order), eps as a double
extern "C" int tgp_connect_coalesce_count_f64(const int64_t* row, const int64_t* col, const double* w, int64_t E,
                                              const int64_t* cluster_index, int64_t N, int64_t K, int reduce_op,
                                              int flags, double eps, void* ws, size_t ws_bytes, int64_t* d_count,
                                              void* stream_) {
  return coalesce_count_impl<double>(row, col, w, E, cluster_index, N, K, reduce_op, flags, eps, ws, ws_bytes, d_count,
                                     stream_);
}

template <typename VT>
static int coalesce_fill_impl(const void* ws, int64_t E, int64_t N, int64_t K, int has_weight, int64_t num_out,
                              int64_t* out_row, int64_t* out_col, VT* out_w, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(ws && E >= 0 && num_out >= 0, TGP_ERR_INVALID, "tgp_connect_coalesce_fill: bad argument");
  if (num_out == 0 || E == 0) return TGP_OK;
  TGP_REQUIRE(out_row && out_col && (!has_weight || out_w), TGP_ERR_INVALID,
              "tgp_connect_coalesce_fill: null output");
  CoalesceWsT<VT> s;
  coalesce_layout<VT>(const_cast<void*>(ws), E, N, &s);
  const uint64_t Ku = static_cast<uint64_t>(K > 0 ? K : 1);
  // The ping-pong parity is a pure function of (E, K): recompute it instead of reading it back.
  const bool first = (sort_passes(E, bits_for(Ku * Ku - 1)) % 2) == 0;
  const int nb = cdiv(E, kCompactTile);
  hipLaunchKernelGGL(coalesce_fill_kernel<VT>, dim3(nb), dim3(256), 0, stream, first ? s.k0 : s.k1, s.seg, s.keep, E,
                     Ku, s.offsets, out_row, out_col, has_weight ? out_w : static_cast<VT*>(nullptr));
  return check_launch("tgp_connect_coalesce_fill");
}

extern "C" int tgp_connect_coalesce_fill(const void* ws, int64_t E, int64_t N, int64_t K, int has_weight, int /*flags*/,
                                         int64_t num_out, int64_t* out_row, int64_t* out_col, float* out_w,
                                         void* stream_) {
  return coalesce_fill_impl<float>(ws, E, N, K, has_weight, num_out, out_row, out_col, out_w, stream_);
}
extern "C" int tgp_connect_coalesce_fill_f64(const void* ws, int64_t E, int64_t N, int64_t K, int has_weight,
                                             int /*flags*/, int64_t num_out, int64_t* out_row, int64_t* out_col,
                                             double* out_w, void* stream_) {
  return coalesce_fill_impl<double>(ws, E, N, K, has_weight, num_out, out_row, out_col, out_w, stream_);
}

// ------------------------------------------------------------------------------------- norms
extern "C" size_t tgp_postprocess_sparse_workspace_bytes(int64_t E, int64_t num_nodes, int64_t num_graphs) {
  const size_t slabs = static_cast<size_t>(cdiv(E > 0 ? E : 1, kDegSlab));
  return align_up((num_nodes > 0 ? num_nodes : 1) * sizeof(float)) +
         align_up((num_graphs > 0 ? num_graphs : 1) * sizeof(uint32_t)) + 3 * align_up(slabs * sizeof(float)) + 512;
}

extern "C" int tgp_postprocess_sparse_norm_f32(const int64_t* row, const int64_t* col, float* w, int64_t E,
                                               int64_t num_nodes, int flags, float eps, const int64_t* batch_pooled,
                                               int64_t num_graphs, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(E >= 0 && num_nodes >= 0, TGP_ERR_INVALID, "tgp_postprocess_sparse_norm_f32: bad size");
  if (E == 0) return TGP_OK;
  TGP_REQUIRE(row && col && w, TGP_ERR_INVALID, "tgp_postprocess_sparse_norm_f32: null pointer");
  TGP_REQUIRE(ws && ws_bytes >= tgp_postprocess_sparse_workspace_bytes(E, num_nodes, num_graphs),
              TGP_ERR_WORKSPACE, "tgp_postprocess_sparse_norm_f32: workspace too small");
  Carver cv(ws);
  float* deg = cv.take<float>(num_nodes > 0 ? num_nodes : 1);
  uint32_t* gmax = cv.take<uint32_t>(num_graphs > 0 ? num_graphs : 1);
  const int64_t slabs = cdiv(E, kDegSlab);
  float* head = cv.take<float>(slabs);
  float* tail = cv.take<float>(slabs);
  int* slab_flags = cv.take<int>(slabs);
  int* unsorted = cv.take<int>(4);
  const int nb = cdiv(E, 256);
  if (flags & TGP_DEGREE_NORM) {
    (void)hipMemsetAsync(deg, 0, static_cast<size_t>(num_nodes) * sizeof(float), stream);
    (void)hipMemsetAsync(unsorted, 0, sizeof(int), stream);
    hipLaunchKernelGGL(check_sorted_rows_kernel, dim3(nb), dim3(256), 0, stream, row, E, unsorted);
    const int slab_blocks = cdiv(slabs, 4);
    hipLaunchKernelGGL(degree_sorted_kernel, dim3(slab_blocks), dim3(256), 0, stream, row, w, E, unsorted, deg, head,
                       tail, slab_flags);
    hipLaunchKernelGGL(degree_stitch_kernel, dim3(slab_blocks), dim3(256), 0, stream, row, E, unsorted, head, tail,
                       slab_flags, deg);
    hipLaunchKernelGGL(degree_fallback_kernel, dim3(nb < 2048 ? nb : 2048), dim3(256), 0, stream, row, w, E,
                       unsorted, deg);
    hipLaunchKernelGGL(degree_scale_kernel, dim3(nb), dim3(256), 0, stream, row, col, w, E, deg, eps);
  }
  if (flags & TGP_EDGE_WEIGHT_NORM) {
    TGP_REQUIRE(batch_pooled && num_graphs > 0, TGP_ERR_INVALID,
                "tgp_postprocess_sparse_norm_f32: batch_pooled required for edge_weight_norm");
    (void)hipMemsetAsync(gmax, 0, static_cast<size_t>(num_graphs) * sizeof(uint32_t), stream);
    hipLaunchKernelGGL(graph_max_kernel, dim3(cdiv(E, 256 * kGraphMaxPerThread)), dim3(256), 0, stream, row, w, E,
                       batch_pooled, gmax);
    hipLaunchKernelGGL(graph_max_scale_kernel, dim3(nb), dim3(256), 0, stream, row, w, E, batch_pooled, gmax);
  }
  return check_launch("tgp_postprocess_sparse_norm_f32");
}

// ------------------------------------------------------------------------------------- block diag
extern "C" size_t tgp_block_diag_workspace_bytes(int64_t B, int64_t K) {
  const int64_t total = B * K * K;
  const size_t nb = static_cast<size_t>(cdiv(total > 0 ? total : 1, kCompactTile));
  return 2 * align_up(nb * sizeof(uint32_t)) + 256;
}

template <typename VT>
static int block_diag_count_impl(const VT* adj, int64_t B, int64_t K, const int64_t* relabel, int flags, VT eps, void* ws,
                                 size_t ws_bytes, int64_t* d_count, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(B >= 0 && K >= 0 && d_count, TGP_ERR_INVALID, "tgp_block_diag_count: bad argument");
  const int64_t total = B * K * K;
  TGP_REQUIRE(total < (1ll << 31), TGP_ERR_RANGE, "tgp_block_diag_count: B*K*K >= 2^31");
  TGP_REQUIRE(ws && ws_bytes >= tgp_block_diag_workspace_bytes(B, K), TGP_ERR_WORKSPACE,
              "tgp_block_diag_count: workspace too small");
  if (total == 0) {
    (void)hipMemsetAsync(d_count, 0, sizeof(int64_t), stream);
    return check_launch("tgp_block_diag_count");
  }
  TGP_REQUIRE(adj, TGP_ERR_INVALID, "tgp_block_diag_count: null adjacency");
  Carver cv(ws);
  const int nb = cdiv(total, kCompactTile);
  uint32_t* counts = cv.take<uint32_t>(nb);
  uint32_t* offsets = cv.take<uint32_t>(nb);
  BlockDiagPredT<VT> pred{adj, relabel, K, flags, eps};
  hipLaunchKernelGGL(blockdiag_count_kernel<VT>, dim3(nb), dim3(256), 0, stream, pred, total, counts);
  hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, stream, counts, nb, offsets, d_count,
                     static_cast<const int*>(nullptr));
  return check_launch("tgp_block_diag_count");
}
extern "C" int tgp_block_diag_count(const float* adj, int64_t B, int64_t K, const int64_t* relabel, int flags,
                                    float eps, void* ws, size_t ws_bytes, int64_t* d_count, void* stream_) {
  return block_diag_count_impl<float>(adj, B, K, relabel, flags, eps, ws, ws_bytes, d_count, stream_);
}
extern "C" int tgp_block_diag_count_f64(const double* adj, int64_t B, int64_t K, const int64_t* relabel, int flags,
                                        double eps, void* ws, size_t ws_bytes, int64_t* d_count, void* stream_) {
  return block_diag_count_impl<double>(adj, B, K, relabel, flags, eps, ws, ws_bytes, d_count, stream_);
}

template <typename VT>
static int block_diag_fill_impl(const VT* adj, int64_t B, int64_t K, const int64_t* relabel, int flags, VT eps,
                                const void* ws, int64_t num_out, int64_t* out_row, int64_t* out_col, VT* out_w,
                                void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int64_t total = B * K * K;
  TGP_REQUIRE(ws && num_out >= 0, TGP_ERR_INVALID, "tgp_block_diag_fill: bad argument");
  if (num_out == 0 || total == 0) return TGP_OK;
  TGP_REQUIRE(adj && out_row && out_col && out_w, TGP_ERR_INVALID, "tgp_block_diag_fill: null pointer");
  Carver cv(const_cast<void*>(ws));
  const int nb = cdiv(total, kCompactTile);
  cv.take<uint32_t>(nb);
  uint32_t* offsets = cv.take<uint32_t>(nb);
  BlockDiagPredT<VT> pred{adj, relabel, K, flags, eps};
  hipLaunchKernelGGL(blockdiag_fill_kernel<VT>, dim3(nb), dim3(256), 0, stream, pred, total, offsets, out_row,
                     out_col, out_w);
  return check_launch("tgp_block_diag_fill");
}
extern "C" int tgp_block_diag_fill(const float* adj, int64_t B, int64_t K, const int64_t* relabel, int flags,
                                   float eps, const void* ws, int64_t num_out, int64_t* out_row, int64_t* out_col,
                                   float* out_w, void* stream_) {
  return block_diag_fill_impl<float>(adj, B, K, relabel, flags, eps, ws, num_out, out_row, out_col, out_w, stream_);
}
extern "C" int tgp_block_diag_fill_f64(const double* adj, int64_t B, int64_t K, const int64_t* relabel, int flags,
                                       double eps, const void* ws, int64_t num_out, int64_t* out_row, int64_t* out_col,
                                       double* out_w, void* stream_) {
  return block_diag_fill_impl<double>(adj, B, K, relabel, flags, eps, ws, num_out, out_row, out_col, out_w, stream_);
}

// ------------------------------------------------------------------------------------- debug sort
extern "C" size_t tgp_debug_sort_workspace_bytes(int64_t n) {
  const size_t m = static_cast<size_t>(n > 0 ? n : 1);
  return 2 * align_up(m * sizeof(uint64_t)) + 2 * align_up(m * sizeof(uint32_t)) +
         align_up(sort_scratch_words() * sizeof(uint32_t));
}

extern "C" int tgp_debug_sort_pairs_u64(const uint64_t* keys_in, const uint32_t* vals_in, int64_t n,
                                        int key_bits, uint64_t* keys_out, uint32_t* vals_out, void* ws,
                                        size_t ws_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(n >= 0 && key_bits >= 0 && key_bits <= 64, TGP_ERR_INVALID, "tgp_debug_sort_pairs_u64: bad argument");
  TGP_REQUIRE(ws && ws_bytes >= tgp_debug_sort_workspace_bytes(n), TGP_ERR_WORKSPACE,
              "tgp_debug_sort_pairs_u64: workspace too small");
  if (n == 0) return TGP_OK;
  Carver cv(ws);
  uint64_t* k0 = cv.take<uint64_t>(n);
  uint64_t* k1 = cv.take<uint64_t>(n);
  uint32_t* v0 = cv.take<uint32_t>(n);
  uint32_t* v1 = cv.take<uint32_t>(n);
  uint32_t* scratch = cv.take<uint32_t>(sort_scratch_words());
  (void)hipMemcpyAsync(k0, keys_in, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream);
  (void)hipMemcpyAsync(v0, vals_in, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
  bool first = true;
  const int rc = radix_sort_pairs<uint64_t, uint32_t>(k0, v0, k1, v1, n, key_bits, scratch, stream, &first);
  if (rc != TGP_OK) return rc;
  (void)hipMemcpyAsync(keys_out, first ? k0 : k1, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream);
  (void)hipMemcpyAsync(vals_out, first ? v0 : v1, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
  return check_launch("tgp_debug_sort_pairs_u64");
}

// The same offsets for a list whose order has not been checked yet: *d_unsorted (zeroed here) becomes 1 when a row id
// descends.  The caller goes on optimistically and reads the flag with its next read-back (the offsets of an unsorted
// list are meaningless but in range).
extern "C" int tgp_rowptr_from_sorted_flag_i64(const int64_t* rows, int64_t n, int64_t num_rows, int32_t* row_ptr,
                                               int* d_unsorted, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(n >= 0 && num_rows >= 0 && row_ptr && d_unsorted && (n == 0 || rows), TGP_ERR_INVALID,
              "tgp_rowptr_from_sorted_flag_i64: bad argument");
  TGP_REQUIRE(n < (1ll << 31), TGP_ERR_RANGE, "tgp_rowptr_from_sorted_flag_i64: n >= 2^31");
  (void)hipMemsetAsync(d_unsorted, 0, sizeof(int), stream);
  hipLaunchKernelGGL(rowptr_from_sorted_kernel, dim3(cdiv(n + 1, 256)), dim3(256), 0, stream, rows, n, num_rows,
                     row_ptr, d_unsorted);
  return check_launch("tgp_rowptr_from_sorted_flag_i64");
}

// ------------------------------------------------------------------------------------- SpMM (A7')
extern "C" int tgp_rowptr_from_sorted_i64(const int64_t* rows, int64_t n, int64_t num_rows, int32_t* row_ptr,
                                          void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(n >= 0 && num_rows >= 0 && row_ptr && (n == 0 || rows), TGP_ERR_INVALID,
              "tgp_rowptr_from_sorted_i64: bad argument");
  TGP_REQUIRE(n < (1ll << 31), TGP_ERR_RANGE, "tgp_rowptr_from_sorted_i64: n >= 2^31");
  hipLaunchKernelGGL(rowptr_from_sorted_kernel, dim3(cdiv(n + 1, 256)), dim3(256), 0, stream, rows, n, num_rows,
                     row_ptr, static_cast<int*>(nullptr));
  return check_launch("tgp_rowptr_from_sorted_i64");
}

extern "C" int tgp_edge_row_stats_f32(const int32_t* row_ptr, const float* w, const float* S, int64_t N, int64_t K,
                                      float* deg, float* q, void* stream_);

// deg / q: optional outputs (both or none) of the STATS form
// entropy (mode 2): deg = the per-block partial array (>= num_rows floats of room), *n_partial = entries written, or -1 when
// this shape does not take the row kernel (the caller then runs the loss' own pass)
static int spmm_csr_impl(const int32_t* row_ptr, const int64_t* col, const float* w, int64_t num_rows, int64_t nnz,
                         const float* S, int64_t K, float* T, float* deg, float* q, void* stream_, const char* what,
                         float ent_eps = 0.f, int* n_partial = nullptr) {
  const bool entropy = n_partial != nullptr;
  if (entropy) *n_partial = -1;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(num_rows >= 0 && nnz >= 0 && K >= 0, TGP_ERR_INVALID, "%s: negative size", what);
  if (num_rows == 0 || K == 0) return TGP_OK;
  TGP_REQUIRE(row_ptr && T && (nnz == 0 || (col && S)) && (entropy || (!deg == !q)) && (!deg || S), TGP_ERR_INVALID,
              "%s: null pointer", what);
  // T[i,:] = sum over row i of w[e] S[col[e],:]: rows of 16-byte vectors (K % 4 == 0, K >= 16) take the XCD-grouped row
  // kernel (r6: C2, K = 128: 40 -> 21 us against the r5 route through the sparse Reduce's gather-sum; 2048 graphs of 40
  // nodes, K = 20 / 32: 20.9 / 31.1 -> 12.3 / 11.5 us against one lane per output element), bit-identical; other shapes
  // one lane per output element
  static const int kMinK = getenv("TGP_SPMM_ROWS_MIN_K") ? atoi(getenv("TGP_SPMM_ROWS_MIN_K")) : 16;
  if (K >= kMinK && K % 4 == 0 && reinterpret_cast<uintptr_t>(S) % 16 == 0 && reinterpret_cast<uintptr_t>(T) % 16 == 0) {
    static const int legacy = getenv("TGP_SPMM_REDUCE_ROUTE") ? atoi(getenv("TGP_SPMM_REDUCE_ROUTE")) : 0;  // A/B switch
    if (legacy) {
      const int rc = tgp_reduce_sparse_f32(S, 0, K, K, col, w, row_ptr, nullptr, nnz, num_rows, T, stream_);
      return (rc == TGP_OK && deg && !entropy) ? tgp_edge_row_stats_f32(row_ptr, w, S, num_rows, K, deg, q, stream_) : rc;
    }
    TGP_REQUIRE(num_rows < (1ll << 31), TGP_ERR_RANGE, "%s: num_rows >= 2^31", what);
    // r6: rows in contiguous chunks, a chunk = one workgroup, workgroups that share an XCD take neighbouring chunks
    const int G = K <= 32 ? 8 : (K <= 64 ? 16 : (K <= 128 ? 32 : 64));
    const int rpb_unit = 256 / G;
    static const int kIter = getenv("TGP_SPMM_ROWS_ITER") ? atoi(getenv("TGP_SPMM_ROWS_ITER")) : 1;
    int rows_per_block = rpb_unit * (kIter > 0 ? kIter : 1);
    int64_t blocks = (num_rows + rows_per_block - 1) / rows_per_block;
    while (blocks > 256 * 32) {  // (very long lists: longer chunks instead of more workgroups)
      rows_per_block *= 2;
      blocks = (num_rows + rows_per_block - 1) / rows_per_block;
    }
    const dim3 grid(static_cast<unsigned>(blocks)), block(256);
#define TGP_SPMM_ROWS(GG)                                                                                              \
  do {                                                                                                                 \
    if (entropy)                                                                                                       \
      hipLaunchKernelGGL((spmm_rows_vec4_kernel<GG, 2>), grid, block, 0, stream, row_ptr, col, w, num_rows, S, K, T,   \
                         rows_per_block, deg, q, ent_eps);                                                             \
    else if (deg)                                                                                                      \
      hipLaunchKernelGGL((spmm_rows_vec4_kernel<GG, 1>), grid, block, 0, stream, row_ptr, col, w, num_rows, S, K, T,   \
                         rows_per_block, deg, q, 0.f);                                                                 \
    else                                                                                                               \
      hipLaunchKernelGGL((spmm_rows_vec4_kernel<GG, 0>), grid, block, 0, stream, row_ptr, col, w, num_rows, S, K, T,   \
                         rows_per_block, deg, q, 0.f);                                                                 \
  } while (0)
    if (entropy) *n_partial = static_cast<int>(blocks);
    if (G == 8) TGP_SPMM_ROWS(8);
    else if (G == 16) TGP_SPMM_ROWS(16);
    else if (G == 32) TGP_SPMM_ROWS(32);
    else TGP_SPMM_ROWS(64);
#undef TGP_SPMM_ROWS
    return check_launch(what);
  }
  int64_t blocks = (num_rows * K + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(spmm_csr_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, row_ptr, col, w,
                     num_rows, S, K, T);
  const int rc = check_launch(what);
  return (rc == TGP_OK && deg && !entropy) ? tgp_edge_row_stats_f32(row_ptr, w, S, num_rows, K, deg, q, stream_) : rc;
}

extern "C" int tgp_spmm_csr_f32(const int32_t* row_ptr, const int64_t* col, const float* w, int64_t num_rows,
                                int64_t nnz, const float* S, int64_t K, float* T, void* stream_) {
  return spmm_csr_impl(row_ptr, col, w, num_rows, nnz, S, K, T, nullptr, nullptr, stream_, "tgp_spmm_csr_f32");
}

extern "C" int tgp_spmm_csr_stats_f32(const int32_t* row_ptr, const int64_t* col, const float* w, int64_t num_rows,
                                      int64_t nnz, const float* S, int64_t K, float* T, float* deg, float* q,
                                      void* stream_) {
  TGP_REQUIRE(deg && q, TGP_ERR_INVALID, "tgp_spmm_csr_stats_f32: null output");
  return spmm_csr_impl(row_ptr, col, w, num_rows, nnz, S, K, T, deg, q, stream_, "tgp_spmm_csr_stats_f32");
}

// r6: T = A S with DiffPool's entropy sum riding along: partial[0 .. *n_partial) = per-workgroup shares of
// sum(-S log(S + eps)) over all of S (S has num_rows rows); *n_partial = -1: this shape does not take the row kernel
// (T is computed, the partials are not: the caller runs tgp_entropy_partials_f32).  partial: >= num_rows floats.
extern "C" int tgp_spmm_csr_entropy_f32(const int32_t* row_ptr, const int64_t* col, const float* w, int64_t num_rows,
                                        int64_t nnz, const float* S, int64_t K, float* T, float eps, float* partial,
                                        int* n_partial, void* stream_) {
  TGP_REQUIRE(partial && n_partial, TGP_ERR_INVALID, "tgp_spmm_csr_entropy_f32: null output");
  return spmm_csr_impl(row_ptr, col, w, num_rows, nnz, S, K, T, partial, nullptr, stream_, "tgp_spmm_csr_entropy_f32", eps,
                       n_partial);
}
