// Epoch-tagged decoupled look-back, shared by the one-launch kernels that write their survivors at FINAL offsets
// (sparse_pool_small.hip, the single-pass subgraph Connect of sparse_connect.hip).
//
// A tile publishes {epoch, state, refused, count} in one 64-bit word of a caller-owned status buffer: state AGG = the
// tile's own count, PRE = the inclusive prefix up to it.  Words of earlier calls carry another epoch and read as "not
// ready", so the buffer is never cleared (no memset launch in front of the kernel).  A refusal (bit 31) travels inside
// the words: the last tile knows the verdict of the whole call and leaves ONE result word {epoch, refused, total},
// stored with system scope -- the caller may point it at pinned host memory and poll it.
#pragma once
#include <stdlib.h>

#include "primitives.h"

namespace tgp {

constexpr int SPS_EPOCH_SHIFT = 34;
constexpr unsigned long long SPS_AGG = 1ull << 32, SPS_PRE = 2ull << 32;
__device__ __forceinline__ unsigned long long sps_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sps_store(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool sps_current(unsigned long long word, unsigned long long tag) {
  return (word >> SPS_EPOCH_SHIFT) == (tag >> SPS_EPOCH_SHIFT);
}

// Tile ids.  The look-back needs every predecessor tile of a workgroup to be running (or done) when it spins on their
// words.  By default a workgroup's tile is its blockIdx.x: the dispatcher hands a 1-D grid out in index order, so the
// predecessors of a resident workgroup were dispatched before it (and a bounded spin turns anything else into a refusal
// + the caller's fallback).  TGP_LOOKBACK_TICKET=1 (read once by the library) makes the order explicit instead: the tile is
// the workgroup's ARRIVAL number, taken from an epoch-tagged ticket word ([1] of the look-back's status region) as the
// first thing the workgroup does -- forward progress then holds by construction, at the price of one device-scope
// atomic round trip in front of every workgroup (profiles/r06_lookback_ticket.txt).  A word of another epoch is stale
// (the buffer is never cleared) and is claimed with a compare-and-swap.
static const int kLookbackTicket = getenv("TGP_LOOKBACK_TICKET") ? atoi(getenv("TGP_LOOKBACK_TICKET")) : 0;

__device__ __forceinline__ int sps_tile_id(unsigned long long* ticket_word, unsigned long long tag, int use_ticket,
                                           int* s_tile) {
  if (!use_ticket) return static_cast<int>(blockIdx.x);
  if (threadIdx.x == 0) {
    const unsigned long long ep = (tag >> SPS_EPOCH_SHIFT) << SPS_EPOCH_SHIFT;
    int got = -1;
    while (got < 0) {
      const unsigned long long cur = __hip_atomic_load(ticket_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((cur >> SPS_EPOCH_SHIFT) == (tag >> SPS_EPOCH_SHIFT)) {
        got = static_cast<int>(atomicAdd(ticket_word, 1ull) & 0xFFFFFFFFull);
      } else if (atomicCAS(ticket_word, cur, ep | 1ull) == cur) {
        got = 0;
      }
    }
    *s_tile = got;
  }
  __syncthreads();
  return *s_tile;
}

__device__ __forceinline__ uint32_t wave_sum32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}
// The look-back in two halves, so that the words' round trip (~1 us: they live behind the fabric) overlaps work that does
// not depend on them: `issue` requests the 256 nearest predecessor words, `finish` consumes them (re-reading only those
// that were not published yet) and walks further back if none of them held a prefix.
// WIN = words per lane and round: 4 (256 predecessors at once) where a round of persistent workgroups publishes together
// and no prefix can be nearer than that (sparse_pool_small, the single-pass subgraph Connect); 1 for grids of thousands of
// short tiles, where a prefix is almost always among the 64 nearest (every word is an uncached read behind the fabric:
// 256 of them per tile made the look-back fill of the coalesce Connect 47 us slower than its scan kernels).
template <int WIN = 4>
struct SpsLookT {
  unsigned long long st[WIN];
  int idx[WIN];
};
using SpsLook = SpsLookT<4>;

template <int WIN>
__device__ __forceinline__ void sps_lookback_issue(const unsigned long long* status, int tile, unsigned long long tag,
                                                   SpsLookT<WIN>& lk) {
  const int lane = lane_id();
#pragma unroll
  for (int k = 0; k < WIN; ++k) {
    lk.idx[k] = tile - 1 - lane - 64 * k;
    lk.st[k] = lk.idx[k] >= 0 ? sps_load(status + 2 + lk.idx[k]) : (tag | SPS_PRE);
  }
}

template <int WIN>
__device__ __forceinline__ void sps_lookback_finish(unsigned long long* status, int tile, unsigned long long tag,
                                                    SpsLookT<WIN>& lk, uint32_t* excl_out, bool* refused) {
  const int lane = lane_id();
  uint32_t excl = 0;
  bool bad = false;
  int j = tile - 1;
  bool first_window = true;
  while (j >= 0) {
    if (!first_window) {
#pragma unroll
      for (int k = 0; k < WIN; ++k) {
        lk.idx[k] = j - lane - 64 * k;
        lk.st[k] = lk.idx[k] >= 0 ? sps_load(status + 2 + lk.idx[k]) : (tag | SPS_PRE);
      }
    }
    first_window = false;
    int spins = 0;
    for (;;) {
      bool wait = false;
#pragma unroll
      for (int k = 0; k < WIN; ++k) wait = wait || !sps_current(lk.st[k], tag) || ((lk.st[k] >> 32) & 3ull) == 0;
      if (!__any(wait)) break;
      if (++spins > (1 << 20)) {  // every spin is bounded
        *excl_out = 0;
        *refused = true;
        return;
      }
      if (spins > 4) __builtin_amdgcn_s_sleep(1);
#pragma unroll
      for (int k = 0; k < WIN; ++k)
        if (lk.idx[k] >= 0 && (!sps_current(lk.st[k], tag) || ((lk.st[k] >> 32) & 3ull) == 0))
          lk.st[k] = sps_load(status + 2 + lk.idx[k]);
    }
    bool done = false;
#pragma unroll
    for (int k = 0; k < WIN; ++k) {
      if (!done) {  // (uniform: `done` comes from ballots)
        const unsigned long long pre = __ballot(((lk.st[k] >> 32) & 3ull) == 2);
        const int first = pre ? __builtin_ctzll(pre) : 64;  // nearest predecessor that already knows its prefix
        excl += wave_sum32(lane <= first ? static_cast<uint32_t>(lk.st[k]) & 0x7FFFFFFFu : 0u);
        bad = bad || __any(lane <= first && ((lk.st[k] >> 31) & 1ull));
        done = pre != 0ull;
      }
    }
    if (done) break;
    j -= 64 * WIN;
  }
  *excl_out = excl;
  *refused = bad;
}

// Exclusive prefix of `tile` over the tiles' survivor counts (bits 0..30 of the published values) and whether any of
// them refused (bit 31); every lane of ONE wave calls it.  A spin bound turns a tile that never shows up into a refusal.
// (256 predecessors per round, four words per lane, all requested at once: everybody publishes at about the same time --
// r4 stamps: 64 per round made the last of 256 tiles wait four dependent rounds.)
template <int WIN = 4>
__device__ __forceinline__ void sps_lookback(unsigned long long* status, int tile, unsigned long long tag,
                                             uint32_t* excl_out, bool* refused) {
  SpsLookT<WIN> lk;
  sps_lookback_issue<WIN>(status, tile, tag, lk);
  sps_lookback_finish<WIN>(status, tile, tag, lk, excl_out, refused);
}

// ---------------------------------------------------------------------------------------------------------------------
// Finding a sorted batch's per-graph ranges inside a kernel (sparse_pool_small.hip, the one-launch Graclus matching).
__device__ __forceinline__ unsigned long long wave_or64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o, WAVE);
  return v;
}
__device__ __forceinline__ int64_t wave_min64(int64_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int64_t t = __shfl_xor(v, o, WAVE);
    v = t < v ? t : v;
  }
  return v;
}
__device__ __forceinline__ int64_t wave_max64(int64_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int64_t t = __shfl_xor(v, o, WAVE);
    v = t > v ? t : v;
  }
  return v;
}

// NK lower bounds by ONE wave, together: 128 probes per round and key (two per lane), the loads of all keys in
// flight at once -- three dependent rounds for 300 k entries instead of a log2 chain of nineteen.  On an array that is
// not ascending the result is still a deterministic function of (array, key), which is all the tiling check of the
// caller needs.
template <int NK>
__device__ __forceinline__ void wave_lower_bounds(const int64_t* const (&arr)[NK], const int64_t (&n)[NK],
                                                  const int64_t (&key)[NK], int64_t (&res)[NK]) {
  const int lane = lane_id();
  int64_t lo[NK], hi[NK];
#pragma unroll
  for (int q = 0; q < NK; ++q) {
    lo[q] = 0;
    hi[q] = n[q];
  }
  for (;;) {
    bool more = false;
#pragma unroll
    for (int q = 0; q < NK; ++q) more = more || lo[q] < hi[q];
    if (!more) break;
    int64_t step[NK], va[NK], vb[NK];
    bool oa[NK], ob[NK];
#pragma unroll
    for (int q = 0; q < NK; ++q) {  // unconditional, clamped loads (a finished key re-reads its last element)
      step[q] = (hi[q] - lo[q] + 127) >> 7;
      const int64_t ia = lo[q] + lane * step[q], ib = lo[q] + (lane + 64) * step[q];
      oa[q] = ia < hi[q];
      ob[q] = ib < hi[q];
      const int64_t last = n[q] > 0 ? n[q] - 1 : 0;
      va[q] = n[q] > 0 ? arr[q][oa[q] ? ia : last] : 0;
      vb[q] = n[q] > 0 ? arr[q][ob[q] ? ib : last] : 0;
    }
#pragma unroll
    for (int q = 0; q < NK; ++q) {
      const int c = __popcll(__ballot(oa[q] && va[q] < key[q])) + __popcll(__ballot(ob[q] && vb[q] < key[q]));
      if (lo[q] < hi[q]) {
        if (c == 0) {
          hi[q] = lo[q];
        } else {
          const int64_t last = lo[q] + static_cast<int64_t>(c - 1) * step[q];
          lo[q] = last + 1;
          hi[q] = last + step[q] < hi[q] ? last + step[q] : hi[q];
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NK; ++q) res[q] = lo[q];
}

// boundaries of WAVES + 1 ascending keys s_key[] inside arr[lo, lo + len): s_out[t] = first i with arr[lo + i] >= key
// (left at INT_MAX when no element is that large).  One coalesced pass by the whole workgroup; on an array that is not
// ascending the minimum of the candidates is taken (deterministic; the caller's range checks then refuse the input).
template <int WAVES>
__device__ __forceinline__ void sps_boundaries(const int64_t* __restrict__ arr, int64_t lo, int64_t len,
                                               const int64_t* s_key, int* s_out) {
  for (int64_t i = threadIdx.x; i < len; i += WAVES * WAVE) {
    const int64_t r = arr[lo + i];
    const int64_t rp = i > 0 ? arr[lo + i - 1] : INT64_MIN;
    if (r != rp) {
      for (int t = 0; t <= WAVES; ++t) {
        const int64_t nb = s_key[t];
        if (rp < nb && nb <= r) atomicMin(&s_out[t], static_cast<int>(i));
      }
    }
  }
}

}  // namespace tgp
