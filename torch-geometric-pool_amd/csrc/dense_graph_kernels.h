// Whole graphs per wave / workgroup (small and medium graph batches) and batched products of small matrices.
#pragma once
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "gemm_mfma.h"  // f32x16, stamps

namespace tgp {

// ------------------------------------------------------------------------------------------
// Small graphs (N <= 64, K <= 32, F <= 32; e.g. the PROTEINS-shaped batch of BASELINE configs[2]):
// one WAVE owns one graph.  S and X go straight from HBM into the MFMA operand registers (row-coalesced:
// lane = column, one node row per half-wave); only A, whose operand layout is the transpose of its
// memory layout, is staged through LDS (zero padded 64 x 65 per wave, so two workgroups fit a CU).
// Then X' = S^T X (32 MFMAs), U = A S (64) and A' = S^T U (32) run back to back.  U never leaves the
// accumulators: register r of the 32x32 C/D layout holds rows (rho(r), rho(r)+4) for the two half-waves,
// which is exactly a k-pair of the next MFMA's B operand, so every product walks the node dimension in
// that order (node(q) below) and all three share one register copy of S.  The post-processing
// (utils/ops.py:282-335) happens in registers + wave shuffles.  Each graph crosses HBM once: HBM-bound.
// ------------------------------------------------------------------------------------------
constexpr int SG_N = 64, SG_K = 32, SG_LDA = 65;
constexpr int SG_WAVE_FLOATS = SG_N * SG_LDA + SG_N;  // the A tile, then the node degrees of the fused MinCut terms

struct SmallArgs {
  const float* S; const float* A; const float* X;
  int B, N, K, F, flags;
  float eps;
  float* x_pool; float* adj_raw; float* adj_pool;
  // optional [2,B] (r3): the per-graph tails of MinCut's two auxiliary losses (utils/losses.py:39-70), taken where S,
  // A and the raw S^T A S already sit in registers / LDS: terms[b] = -trace(S^T A S) / (trace(S^T D S) + loss_eps),
  // terms[B + b] = || S^T S / ||S^T S||_F - I / sqrt(K) ||_F  (five more launches and a second pass over A and S when
  // computed behind the pooling kernel)
  float* mincut_terms;
  float loss_eps;
  // optional (r3): the selector folded in -- S = softmax(X W^T + b) * mask (select/mlp_select.py:105-147 for a single
  // Linear) is formed by the wave from the X it has just loaded, written ONCE to s_out [B,N,K] (it is an output:
  // SelectOutput.s) and used from registers; p.S is ignored then.  sel_w [K,F], sel_b [K] or NULL, sel_mask [B,N] bytes
  // or NULL.
  const float* sel_w; const float* sel_b; const unsigned char* sel_mask; float* s_out;
  // optional (r5) [B * K]: the pooled batch vector arange(B).repeat_interleave(K) (utils/ops.py:152-169), written by
  // the graph's wave: one copy launch less per pooler call
  long long* batch_pool;
  // optional (r5, selector folded in, dense_pool_small_kernel<true>): the batch as a PyG loader hands it over -- x
  // [Ntot,F] un-padded in `X`, a ROW-SORTED edge list -- instead of the padded X and the dense A: graph b owns node rows
  // node_ptr[b] .. node_ptr[b+1] and edges edge_ptr[b] .. edge_ptr[b+1] (the caller has checked the order and computed
  // the ranges).  The wave zeroes its LDS tile and adds its edges there (duplicates summed, entries beyond N dropped,
  // a column outside the graph placed by its local id in its own graph: to_dense_adj, src.py:434-443); e_transposed:
  // adj_transpose (the logical A is the transposed scatter).  Neither to_dense_batch nor to_dense_adj runs, and no
  // [B,N,N] tensor exists.  mask_out [B,N]: to_dense_batch's node mask.
  const long long* e_row; const long long* e_col; const float* e_w;
  const long long* node_ptr; const long long* edge_ptr; const long long* e_batch;
  int e_transposed;
  unsigned char* mask_out;
  long long e_count, n_total;  // E and Ntot: what is read through edge_ptr / e_col is clamped / range-checked against them
  // optional side outputs of the sparse form (training: the backward kernels read the padded tensors): the adjacency
  // tile as it stands in LDS -> a_dense_out [B,N,N], the graph's rows of x zero-padded -> x_dense_out [B,N,F]
  float* a_dense_out; float* x_dense_out;
  // optional (r6) [B,4]: what DiffPool's two losses need of this graph, taken where A, S and S^T A S sit in LDS /
  // registers: (sum of A_ij^2, trace(S^T A S), |S^T S|_F^2, sum of -S log(S + loss_eps)).  With
  //   |A - S S^T|_F^2 = sum A^2 - 2 trace(S^T A S) + |S^T S|_F^2      (utils/losses.py:644-658, 476-483)
  // the link-prediction residual needs neither the dense adjacency outside this kernel nor a product of its own: the
  // inference call no longer writes [B,N,N] and the four launches of the loss tail become one (tgp_diffpool_stats_tail_f32).
  float* diff_stats;
};

__device__ __forceinline__ float sg_wave_sum(float v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, WAVE);
  return v;
}

__device__ __forceinline__ int rho(int r) { return (r & 3) + 8 * (r >> 2); }
// Reductions over the 32 lanes of a half-wave, the result in every lane: four DPP rotations inside the 16-lane rows
// (register moves) and ONE ds_swizzle across the two rows (a __shfl_xor ladder is five trips through the LDS pipe: 320
// of them per wave made the folded selector cost 20 us instead of 3)
template <int CTRL>
__device__ __forceinline__ float sg_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float sg_swz16(float v) {
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x1F | (16 << 10)));
}
__device__ __forceinline__ float sg_half_sum(float v) {
  v += sg_dpp<0x121>(v);  // row_ror:1
  v += sg_dpp<0x122>(v);
  v += sg_dpp<0x124>(v);
  v += sg_dpp<0x128>(v);
  return v + sg_swz16(v);
}
__device__ __forceinline__ float sg_half_max(float v) {
  v = fmaxf(v, sg_dpp<0x121>(v));
  v = fmaxf(v, sg_dpp<0x122>(v));
  v = fmaxf(v, sg_dpp<0x124>(v));
  v = fmaxf(v, sg_dpp<0x128>(v));
  return fmaxf(v, sg_swz16(v));
}
// exp(x) for x <= 0 on v_exp_f32 with the rounding error of x * log2(e) carried along (as csrc/mlp_select.hip: relative
// error ~2e-7; -inf and anything below 2^-150 give 0)
__device__ __forceinline__ float sg_exp_neg(float x) {
  const float L = 1.44269504088896341f, Llo = 1.92596299112661746e-8f;
  x = fmaxf(x, -104.f);
  const float y = x * L;
  float r = fmaf(x, L, -y);
  r = fmaf(x, Llo, r);
  const float e = __builtin_amdgcn_exp2f(y);
  return fmaf(e, r * 0.693147180559945309f, e);
}
// uniform base + 32-bit per-lane byte offset: lets the load use the SGPR-base addressing form
__device__ __forceinline__ const float* byte_off(const float* base, int bytes) {
  return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + static_cast<unsigned>(bytes));
}

#ifdef TGP_GEMM_STAMPS
// per-WAVE stamps of the small-graph kernel (slot s of graph b at stamps[b*16 + s])
#define TGP_WSTAMP(slot)                                                                                  \
  do {                                                                                                    \
    if (g_gemm_stamps && lane_id() == 0)                                                                  \
      g_gemm_stamps[static_cast<long>(blockIdx.x * 8 + wave_id()) * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define TGP_WSTAMP(slot) do {} while (0)
#endif

// Workgroup = 8 waves = 8 graphs.  Waves w and w + 4 share a SIMD (a workgroup's waves are dealt to the SIMDs
// cyclically).  Waves 4-7 queue their load requests BEHIND those of waves 0-3 (an LDS counter the first group bumps
// once its loads are issued), so that the first group's operands land first and its MFMA phases overlap the tail of
// the second group's loads.  Measured (r2 stamps): a wave cannot keep more than ~3.8 GB/s of these loads in flight and
// a CU needs ~7 loading waves to reach its ~26 GB/s, so with 8 resident waves (187 VGPRs) the overlap is small:
// 22.0 -> 21.3 us.  Real overlap needs 16 lighter waves per CU (two waves per graph); see DESIGN.md known gaps.
constexpr int SG_WAVES = 8;
constexpr int SG_EDGE_ROUNDS = 8;  // edges of a graph requested up front by the sparse form (64 per round)
// LATE_X (r6, the form without a folded selector: S is an input): the chain A, S -> U -> A' -> post-processing does not
// need X, whose 32 requests are the last a wave makes.  They are made on every path (a call without X reads element 0 of
// S), so that the compiler can COUNT them and wait for A and S with `vmcnt(32)` instead of `vmcnt(0)`; X' = S^T X runs
// last, on operands that landed while the adjacency phases ran.  A separate instantiation: with the selector folded in X
// is needed first, and carrying both orders behind a run-time test costs the folded form 3 us (profiles/r06_c3_experiments.md).
template <bool SPARSE, bool LATE_X = false>
__global__ __launch_bounds__(64 * SG_WAVES, 2) void dense_pool_small_kernel(SmallArgs p) {
  static_assert(!(SPARSE && LATE_X), "the sparse form builds its operands from X's rows first");
  const float* const sel_w = LATE_X ? nullptr : p.sel_w;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int s_issued;
  const int lane = lane_id();
  const int w = __builtin_amdgcn_readfirstlane(wave_id());  // wave-uniform => graph bases stay in SGPRs
  const int lm = lane & 31, lk = lane >> 5;
  float* As = smem + w * SG_WAVE_FLOATS;
  const int N = p.N, K = p.K, F = p.F;
  const bool at = p.flags & TGP_ADJ_TRANSPOSED;
  const int b = blockIdx.x * SG_WAVES + w;  // one graph per wave, no loop (keeps the 64 + 64 load offsets transient)
  if (threadIdx.x == 0) s_issued = 0;
  __syncthreads();
  if (b >= p.B) return;
  if (p.batch_pool && lane < K) p.batch_pool[static_cast<long>(b) * K + lane] = b;
  [[maybe_unused]] long long n0 = 0, e0 = 0, e1 = 0;
  int nb = N;  // real nodes of the graph (sparse form: from node_ptr; dense form: the padding is zero / masked)
  if constexpr (SPARSE) {
    n0 = p.node_ptr[b];
    const long long nn = p.node_ptr[b + 1] - n0;
    nb = nn < N ? static_cast<int>(nn) : N;
    // (clamped: the ranges may come from a facts kernel whose verdict the host reads only after this launch)
    e0 = p.edge_ptr[b];
    e1 = p.edge_ptr[b + 1];
    e0 = e0 < 0 ? 0 : (e0 > p.e_count ? p.e_count : e0);
    e1 = e1 < e0 ? e0 : (e1 > p.e_count ? p.e_count : e1);
    if (p.mask_out && lane < N) p.mask_out[static_cast<long>(b) * N + lane] = lane < nb ? 1 : 0;
  }
  // second group: wait until the first group's requests are in the queue.  (Not with the selector folded in: there a wave
  // starts with a softmax pass over X, which staggers the groups on its own -- measured r6, 0.044 -> 0.042 ms per forward.)
  if (w >= SG_WAVES / 2 && !sel_w) {
    int first = p.B - static_cast<int>(blockIdx.x) * SG_WAVES;
    first = first < SG_WAVES / 2 ? first : SG_WAVES / 2;
    while (__hip_atomic_load(&s_issued, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < first)
      __builtin_amdgcn_s_sleep(2);
  }
  TGP_WSTAMP(0);
  {
    // ---- request everything up front: A (float4 rows), then S and X in operand order ------------
    // out-of-range elements read element 0 of the graph (always valid) and are replaced by 0 afterwards, so
    // the loads stay unconditional and are issued back to back
    float4 v[SPARSE ? 1 : 16];
    [[maybe_unused]] long long er[SPARSE ? SG_EDGE_ROUNDS : 1], ec[SPARSE ? SG_EDGE_ROUNDS : 1];
    [[maybe_unused]] float ew[SPARSE ? SG_EDGE_ROUNDS : 1];
    if constexpr (SPARSE) {  // the graph's first 512 edges (clamped loads: edge 0 of the list when out of range)
#pragma unroll
      for (int q = 0; q < SG_EDGE_ROUNDS; ++q) {
        const long long e = e0 + q * 64 + lane;
        const long long ee = (p.e_row && e < e1) ? e : 0;
        er[q] = p.e_row ? p.e_row[ee] : 0;
        ec[q] = p.e_row ? p.e_col[ee] : 0;
        ew[q] = (p.e_row && p.e_w) ? p.e_w[ee] : 1.0f;
      }
    }
    if constexpr (!SPARSE)
    if (p.A) {  // 16 lanes per row (64 floats), 4 rows per wave-instruction, 16 instructions
      const float* Ab = p.A + static_cast<long>(b) * N * N;
      const int q = lane & 15;
      const bool qfull = 4 * q + 3 < N;   // whole vector inside the row (rows need dword alignment only)
      const int qrem = N - 4 * q;         // 1..3 on the lane that holds a row's ragged tail (N % 4 != 0)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int i = (lane >> 4) + 4 * t;
        const bool ok = qfull && i < N;
        const float4 r = *reinterpret_cast<const float4*>(byte_off(Ab, ok ? (i * N + 4 * q) * 4 : 0));
        v[t] = ok ? r : make_float4(0.f, 0.f, 0.f, 0.f);
        if (!qfull && qrem > 0 && i < N) {  // never read past the end of the row (= of the tensor for the last one)
          const float* tail = byte_off(Ab, (i * N + 4 * q) * 4);
          v[t].x = tail[0];
          if (qrem > 1) v[t].y = tail[1];
          if (qrem > 2) v[t].z = tail[2];
        }
      }
    }
    // step q of every product contracts node rows node(q) = 32*(q>>4) + rho(q&15) + 4*lk
    float sr[32], xr[32];
    if (!sel_w) {
      const float* Sb = p.S + static_cast<long>(b) * N * K;
      const bool cok = lm < K;
#pragma unroll
      for (int q = 0; q < 32; ++q) {
        const int node = 32 * (q >> 4) + rho(q & 15) + 4 * lk;
        const bool ok = cok && node < N;
        const float r = *byte_off(Sb, ok ? (node * K + lm) * 4 : 0);
        sr[q] = ok ? r : 0.f;
      }
    }
    if (LATE_X || p.X) {
      const float* Xb = (LATE_X && !p.X) ? p.S : SPARSE ? p.X + n0 * F : p.X + static_cast<long>(b) * N * F;  // (sparse: the graph's own rows of x)
      const bool cok = (!LATE_X || p.X) && lm < F;
#pragma unroll
      for (int q = 0; q < 32; ++q) {
        const int node = 32 * (q >> 4) + rho(q & 15) + 4 * lk;
        const bool ok = cok && node < nb;
        const float r = *byte_off(Xb, ok ? (node * F + lm) * 4 : 0);
        if constexpr (LATE_X) xr[q] = r;  // (left as loaded; x_at() zeroes the out-of-range lanes where X' consumes them)
        else xr[q] = ok ? r : 0.f;
      }
    }
    [[maybe_unused]] auto x_at = [&](int q) {
      return (p.X && lm < F && 32 * (q >> 4) + rho(q & 15) + 4 * lk < nb) ? xr[q] : 0.f;
    };
    float wl[16];
    if (sel_w) {  // W [K,F] row by row (lane = feature: coalesced), two rows per load instruction; staged in LDS below
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int kr = 2 * t + lk;
        const bool ok = kr < K && lm < F;
        const float v0 = *byte_off(sel_w, ok ? (kr * F + lm) * 4 : 0);
        wl[t] = ok ? v0 : 0.f;
      }
    }
    if (w < SG_WAVES / 2 && lane == 0) atomicAdd(&s_issued, 1);  // this wave's loads are all issued
    if (sel_w) {
      // ---- S = softmax(X W^T + b) * mask, straight into the operand registers sr[] ------------------------------
      // X is in registers with lane = feature; the product needs it with lane = node: through the (still unused) A tile
      // in LDS.  The accumulators of Z_T = X_T W^T (lane = cluster, register r = node 32 T + rho(r) + 4 lk) ARE the
      // layout sr[16 T + r] of every product below; the softmax of a node is a reduction over the 32 lanes of a half.
      float* Xs = As;            // [node][33]
      float* Ws = As + 64 * 33;  // [cluster][33]
#pragma unroll
      for (int q = 0; q < 32; ++q) Xs[(32 * (q >> 4) + rho(q & 15) + 4 * lk) * 33 + lm] = xr[q];
#pragma unroll
      for (int t = 0; t < 16; ++t) Ws[(2 * t + lk) * 33 + lm] = wl[t];
      __builtin_amdgcn_wave_barrier();
      float wr[16];  // W as the B operand of Z = X W^T: lane = cluster, k-step r = features rho(r) + 4 lk
#pragma unroll
      for (int r = 0; r < 16; ++r) wr[r] = Ws[lm * 33 + rho(r) + 4 * lk];
      const float bz = lm < K ? (p.sel_b ? p.sel_b[lm] : 0.f) : -__builtin_inff();
      const unsigned char* mk = p.sel_mask ? p.sel_mask + static_cast<long>(b) * N : nullptr;
      float* So = p.s_out ? p.s_out + static_cast<long>(b) * N * K : nullptr;
      // the graph's mask bytes as ONE load (lane = node) and a ballot (r5, late): tested as `mk[node] != 0` inside the
      // loop below, each of the 32 tests compiled to a byte load + `s_waitcnt vmcnt(0)` in front of the store of S --
      // 32 dependent round trips per graph
      unsigned long long mbits = ~0ull;
      if (mk) mbits = __ballot(mk[lane < N ? lane : 0] != 0);
#pragma unroll
      for (int T = 0; T < 2; ++T) {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = bz;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          z = __builtin_amdgcn_mfma_f32_32x32x2f32(Xs[(32 * T + lm) * 33 + rho(r) + 4 * lk], wr[r], z, 0, 0, 0);
        // (the 16 reductions of a pass are written side by side: one after the other they are 16 dependent chains of
        //  four DPP moves and an LDS swizzle each)
        float mx[16], ex[16], sm[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) mx[r] = sg_half_max(z[r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) ex[r] = sg_exp_neg(z[r] - mx[r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) sm[r] = sg_half_sum(ex[r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int node = 32 * T + rho(r) + 4 * lk;
          const bool on = node < nb && ((mbits >> node) & 1ull) != 0;
          const float sv = on ? ex[r] * __builtin_amdgcn_rcpf(sm[r]) : 0.f;  // (1 ulp reciprocal: far inside 1e-5)
          sr[16 * T + r] = sv;
          if (So && node < N && lm < K) So[node * K + lm] = sv;
        }
      }
      __builtin_amdgcn_wave_barrier();  // (the A tile overwrites Xs below)
    }
    if constexpr (SPARSE) {
      // the adjacency tile is BUILT here: zeroed, then the graph's edges added in LDS (to_dense_adj, src.py:434-443)
      for (int i = lane * 4; i < SG_N * SG_LDA; i += 256) *reinterpret_cast<float4*>(As + i) = make_float4(0.f, 0.f, 0.f, 0.f);
      __builtin_amdgcn_wave_barrier();
      auto add_edge = [&](long long r, long long c, float wv) {
        const long long lr = r - n0;
        if (c < 0 || c >= p.n_total) return;  // (never used as an index)
        // a column of another graph keeps the local id it has THERE (PyG subtracts ptr[batch[col]])
        const long long lc = (c >= n0 && c < n0 + nb) ? c - n0 : c - p.node_ptr[p.e_batch[c]];
        if (lr >= 0 && lr < N && lc >= 0 && lc < N) {
          const int i = static_cast<int>(p.e_transposed ? lc : lr), j = static_cast<int>(p.e_transposed ? lr : lc);
          __hip_atomic_fetch_add(As + i * SG_LDA + j, wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      };
#pragma unroll
      for (int q = 0; q < SG_EDGE_ROUNDS; ++q)
        if (e0 + q * 64 + lane < e1) add_edge(er[q], ec[q], ew[q]);
      for (long long e = e0 + SG_EDGE_ROUNDS * 64 + lane; e < e1; e += 64)  // a graph of more than 512 edges
        add_edge(p.e_row[e], p.e_col[e], p.e_w ? p.e_w[e] : 1.0f);
    } else if (p.A) {
      const int q = lane & 15;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int i = (lane >> 4) + 4 * t;
        if (!at) {
          float* d = As + i * SG_LDA + 4 * q;
          d[0] = v[t].x; d[1] = v[t].y; d[2] = v[t].z; d[3] = v[t].w;
        } else {  // memory holds A^T: element (row i, cols 4q..4q+3) of memory is A[4q+j][i]
          As[(4 * q + 0) * SG_LDA + i] = v[t].x; As[(4 * q + 1) * SG_LDA + i] = v[t].y;
          As[(4 * q + 2) * SG_LDA + i] = v[t].z; As[(4 * q + 3) * SG_LDA + i] = v[t].w;
        }
      }
    }
    // the tile belongs to this wave alone and LDS serves a wave's requests in order: no workgroup barrier
    __builtin_amdgcn_wave_barrier();
    TGP_WSTAMP(1);
    if constexpr (SPARSE) {
      if (p.a_dense_out) {  // what to_dense_adj would have written for this graph (rows of N floats, coalesced)
        float* o = p.a_dense_out + static_cast<long>(b) * N * N;
        for (int e = lane; e < N * N; e += 64) {
          const int i = e / N, j = e - i * N;
          o[e] = As[i * SG_LDA + j];
        }
      }
      if (p.x_dense_out && lm < F) {  // to_dense_batch's rows of this graph (zero behind its nodes)
        float* o = p.x_dense_out + static_cast<long>(b) * N * F;
#pragma unroll
        for (int q = 0; q < 32; ++q) {
          const int node = 32 * (q >> 4) + rho(q & 15) + 4 * lk;
          if (node < N) o[node * F + lm] = xr[q];
        }
      }
    }

    // ---- X' = S^T X (in front of the adjacency phases, or behind them: LATE_X) -------------------------------
    auto x_prime = [&]() {
      if (p.X && p.x_pool) {
        f32x16 ax;
  #pragma unroll
        for (int r = 0; r < 16; ++r) ax[r] = 0.f;
  #pragma unroll
        for (int q = 0; q < 32; ++q) ax = __builtin_amdgcn_mfma_f32_32x32x2f32(sr[q], LATE_X ? x_at(q) : xr[q], ax, 0, 0, 0);
        if (lm < F) {
          float* o = p.x_pool + static_cast<long>(b) * K * F;
  #pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c = rho(r) + 4 * lk;
            if (c < K) o[c * F + lm] = ax[r];
          }
        }
      }
    };
    if constexpr (!LATE_X) x_prime();

    TGP_WSTAMP(2);
    // ---- U = A S (kept in accumulators), A' = S^T U -----------------------------------------
    if ((SPARSE || p.A) && (p.adj_raw || p.adj_pool)) {
      f32x16 u[2], aa;
#pragma unroll
      for (int r = 0; r < 16; ++r) { u[0][r] = 0.f; u[1][r] = 0.f; aa[r] = 0.f; }
#pragma unroll
      for (int q = 0; q < 32; ++q) {
        const int node = 32 * (q >> 4) + rho(q & 15) + 4 * lk;
        u[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(As[lm * SG_LDA + node], sr[q], u[0], 0, 0, 0);
        u[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(As[(32 + lm) * SG_LDA + node], sr[q], u[1], 0, 0, 0);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          aa = __builtin_amdgcn_mfma_f32_32x32x2f32(sr[mt * 16 + r], u[mt][r], aa, 0, 0, 0);

      TGP_WSTAMP(3);
      // aa[r] = A'[row = rho(r) + 4*lk][col = lm]
      if (p.mincut_terms) {
        float tr = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (rho(r) + 4 * lk == lm) tr += aa[r];
        tr = sg_wave_sum(tr);
        // degrees: lane i sums row i of the (logical) adjacency tile; rows / columns beyond N are zero
        float* s_deg = As + SG_N * SG_LDA;
        {
          float dsum = 0.f;
          const float* rowp = As + lane * SG_LDA;
#pragma unroll 16
          for (int j = 0; j < SG_N; ++j) dsum += rowp[j];
          s_deg[lane] = dsum;
        }
        __builtin_amdgcn_wave_barrier();
        float den = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) {
          const int node = 32 * (q >> 4) + rho(q & 15) + 4 * lk;
          den = fmaf(s_deg[node], sr[q] * sr[q], den);
        }
        den = sg_wave_sum(den);
        f32x16 gg;  // S^T S: lane = column, register r = row rho(r) + 4 lk
#pragma unroll
        for (int r = 0; r < 16; ++r) gg[r] = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) gg = __builtin_amdgcn_mfma_f32_32x32x2f32(sr[q], sr[q], gg, 0, 0, 0);
        float fro = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) fro = fmaf(gg[r], gg[r], fro);
        const float nrm = sqrtf(sg_wave_sum(fro));
        const float tdiag = 1.0f / sqrtf(static_cast<float>(K));
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rho(r) + 4 * lk;
          if (row < K && lm < K) {
            const float y = gg[r] / nrm - (row == lm ? tdiag : 0.f);
            acc = fmaf(y, y, acc);
          }
        }
        acc = sg_wave_sum(acc);
        if (lane == 0) {
          p.mincut_terms[b] = -(tr / (den + p.loss_eps));
          p.mincut_terms[p.B + b] = sqrtf(acc);
        }
      }
      if (p.diff_stats) {
        float tr = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (rho(r) + 4 * lk == lm) tr += aa[r];
        tr = sg_wave_sum(tr);
        float a2 = 0.f;  // lane i sweeps row i of the adjacency tile (rows / columns beyond N are zero)
        {
          const float* rowp = As + lane * SG_LDA;
#pragma unroll 16
          for (int j = 0; j < SG_N; ++j) a2 = fmaf(rowp[j], rowp[j], a2);
        }
        a2 = sg_wave_sum(a2);
        f32x16 gg;  // S^T S: lane = column, register r = row rho(r) + 4 lk
#pragma unroll
        for (int r = 0; r < 16; ++r) gg[r] = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) gg = __builtin_amdgcn_mfma_f32_32x32x2f32(sr[q], sr[q], gg, 0, 0, 0);
        float fro = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) fro = fmaf(gg[r], gg[r], fro);
        fro = sg_wave_sum(fro);
        float ent = 0.f;  // (padded nodes / columns hold S = 0: -0 log(eps) = 0)
#pragma unroll
        for (int q = 0; q < 32; ++q) ent -= sr[q] * logf(sr[q] + p.loss_eps);
        ent = sg_wave_sum(ent);
        if (lane == 0) {
          float* o = p.diff_stats + static_cast<long>(b) * 4;
          o[0] = a2; o[1] = tr; o[2] = fro; o[3] = ent;
        }
      }
      if (p.adj_raw && lm < K) {
        float* o = p.adj_raw + static_cast<long>(b) * K * K;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = rho(r) + 4 * lk;
          if (i < K) o[i * K + lm] = aa[r];
        }
      }
      if (p.adj_pool) {
        if (p.flags & TGP_REMOVE_SELF_LOOPS) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (rho(r) + 4 * lk == lm) aa[r] = 0.f;
        }
        if (p.flags & TGP_DEGREE_NORM) {
          float dcol;  // degree of index `lm`, identical on both half-waves
          if (p.flags & TGP_SUM_AXIS_ROWS) {  // sum over rows (axis -2): per-lane column sum
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += aa[r];
            s += __shfl_xor(s, 32, WAVE);
            dcol = s;
          } else {                            // sum over columns (axis -1): reduce each row over lanes
            float mine = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float s = aa[r];
#pragma unroll
              for (int d = 16; d > 0; d >>= 1) s += __shfl_xor(s, d, WAVE);
              // row (rho(r) + 4*lk) total now on every lane of this half-wave; hand it to lane = row
              const int row = rho(r) + 4 * lk;
              if (lm == row) mine = s;
            }
            // lanes of the other half-wave own the other 16 rows: merge
            const float other = __shfl_xor(mine, 32, WAVE);
            bool own = false;
#pragma unroll
            for (int r = 0; r < 16; ++r) own |= (rho(r) + 4 * lk == lm);
            dcol = own ? mine : other;
          }
          const float d = sqrtf(fmaxf(dcol, p.eps));  // d[lm]
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = rho(r) + 4 * lk;
            const float drow = __shfl(d, row, WAVE);      // d[row]
            const float first = (p.flags & TGP_SUM_AXIS_ROWS) ? d : drow;
            const float second = (p.flags & TGP_SUM_AXIS_ROWS) ? drow : d;
            aa[r] = (aa[r] / first) / second;
          }
        }
        if (p.flags & TGP_EDGE_WEIGHT_NORM) {
          float m = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (lm < K && rho(r) + 4 * lk < K) m = fmaxf(m, fabsf(aa[r]));
#pragma unroll
          for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, WAVE));
          if (m == 0.f) m = 1.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) aa[r] = aa[r] / m;
        }
        TGP_WSTAMP(4);
        if (lm < K) {
          float* o = p.adj_pool + static_cast<long>(b) * K * K;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int i = rho(r) + 4 * lk;
            if (i < K) o[i * K + lm] = aa[r];
          }
        }
      }
    }
    if constexpr (LATE_X) x_prime();
  }
  TGP_WSTAMP(5);
}

// ------------------------------------------------------------------------------------------
// Backward of the small-graph kernel (r3): ONE launch for the gradients of everything dense_pool_small_kernel produces.
// Given the upstream gradients of X' = S^T X, of the post-processed A' (utils/ops.py:282-335), optionally of the raw
// R = S^T A S and of the two per-graph MinCut terms (utils/losses.py:39-70), a wave recomputes R, the node degrees and
// S^T S from S and A in registers / LDS exactly as the forward kernel does and forms
//     gR  = d post / d R (gA')  +  gRaw  -  g_cut / (den + eps) * I
//     gS  = A S gR^T + A^T S gR + X gX'^T + g_cut * 2 tr(R) / (den + eps)^2 * D S + 2 S gG ,     gX = S gX'
// with gG the gradient of || S^T S / ||S^T S|| - I / sqrt(K) || with respect to S^T S (symmetric); DiffPool's link and
// entropy losses add  g_link * link_scale / ||A - S S^T|| * (2 S S^T S - (A + A^T) S)  and
// -g_ent * ent_scale * (log(S + eps) + S / (S + eps)).  Every product with a
// node dimension is computed TRANSPOSED (clusters / features as the MFMA row index, nodes as the column index = lane),
// so that the accumulators of one product are the B operand of the next, as in the forward kernel:
//     U^T = S^T A^T, V^T = S^T A   (lane = node; B operand = the LDS tile of A read by row, resp. by column)
//     gS^T = gR U^T + gR^T V^T + 2 gG S^T + gX' X^T ,   gX^T = gX'^T S^T
// The adjacency gets no gradient here (callers whose A requires one keep the operator-by-operator path), and
// edge_weight_norm is not differentiated here either.  Under autograd the reference runs ~110 launches for a MinCut
// training step on a PROTEINS-shaped batch (profiles/r02_e2e_poolers.txt); this is the Reduce + Connect + loss share of
// the backward in one.
// ------------------------------------------------------------------------------------------
struct SmallBwdArgs {
  const float* S; const float* A; const float* X;
  int B, N, K, F, flags;
  float eps, loss_eps;
  const float* g_x_pool;    // [B,K,F] or NULL
  const float* g_adj_pool;  // [B,K,K] or NULL
  const float* g_adj_raw;   // [B,K,K] or NULL
  const float* g_terms;     // [2,B] or NULL: upstream gradients of the per-graph cut / orthogonality terms
  // ... or of their MEANS over the batch (what MinCutPooling hands out: two scalars, each may be NULL): a graph's
  // term then receives *g_mean_* / B (r5: the mean's and the two selects' backward were seven launches of a few bytes)
  const float* g_mean_cut; const float* g_mean_ortho;
  // bit 0 / bit 1: g_x_pool / g_adj_pool is ONE value that stands for every element (the gradient of a plain sum
  // arrives as an expanded scalar: no [B,K,F] copy of it is made)
  int grad_bcast;
  // DiffPool's two batch-wide losses (utils/losses.py:644-658; both NULL = not part of this backward):
  //   link = link_scale * || A - S S^T ||_F over the whole batch,  ent = ent_scale * sum(-S log(S + ent_eps))
  const float* g_link; const float* g_ent;  // upstream gradients of (link, ent): two scalars, each may be NULL
  const float* diff_losses; // [2] the forward values (link = link_scale * norm gives the norm back)
  float link_scale, ent_scale, ent_eps;
  float* gS;                // [B,N,K]
  float* gX;                // [B,N,F] or NULL
};

__global__ __launch_bounds__(64 * SG_WAVES) void dense_pool_small_bwd_kernel(SmallBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = lane_id();
  const int w = __builtin_amdgcn_readfirstlane(wave_id());
  const int lm = lane & 31, lk = lane >> 5;
  float* As = smem + w * SG_WAVE_FLOATS;
  float* s_deg = As + SG_N * SG_LDA;
  const int N = p.N, K = p.K, F = p.F;
  const bool at = p.flags & TGP_ADJ_TRANSPOSED;
  const int b = blockIdx.x * SG_WAVES + w;
  if (b >= p.B) return;  // (no workgroup barrier below: waves are independent)
  // ---- A -> LDS (logical orientation), S in the operand order of the node contraction (as the forward kernel) -------
  float sr[32];
  {
    float4 v[16];
    const float* Ab = p.A + static_cast<long>(b) * N * N;
    const int q4 = lane & 15;
    const bool qfull = 4 * q4 + 3 < N;
    const int qrem = N - 4 * q4;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int i = (lane >> 4) + 4 * t;
      const bool ok = qfull && i < N;
      const float4 r = *reinterpret_cast<const float4*>(byte_off(Ab, ok ? (i * N + 4 * q4) * 4 : 0));
      v[t] = ok ? r : make_float4(0.f, 0.f, 0.f, 0.f);
      if (!qfull && qrem > 0 && i < N) {
        const float* tail = byte_off(Ab, (i * N + 4 * q4) * 4);
        v[t].x = tail[0];
        if (qrem > 1) v[t].y = tail[1];
        if (qrem > 2) v[t].z = tail[2];
      }
    }
    const float* Sb = p.S + static_cast<long>(b) * N * K;
    const bool cok = lm < K;
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      const int node = 32 * (q >> 4) + rho(q & 15) + 4 * lk;
      const bool ok = cok && node < N;
      const float r = *byte_off(Sb, ok ? (node * K + lm) * 4 : 0);
      sr[q] = ok ? r : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int i = (lane >> 4) + 4 * t;
      if (!at) {
        float* d = As + i * SG_LDA + 4 * q4;
        d[0] = v[t].x; d[1] = v[t].y; d[2] = v[t].z; d[3] = v[t].w;
      } else {
        As[(4 * q4 + 0) * SG_LDA + i] = v[t].x; As[(4 * q4 + 1) * SG_LDA + i] = v[t].y;
        As[(4 * q4 + 2) * SG_LDA + i] = v[t].z; As[(4 * q4 + 3) * SG_LDA + i] = v[t].w;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();

  // ---- forward recomputation: R = S^T (A S); C/D layout: lane = column lm, register r = row rho(r) + 4 lk ------------
  f32x16 R;
#pragma unroll
  for (int r = 0; r < 16; ++r) R[r] = 0.f;
  {
    f32x16 u0, u1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { u0[r] = 0.f; u1[r] = 0.f; }
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      const int node = 32 * (q >> 4) + rho(q & 15) + 4 * lk;
      u0 = __builtin_amdgcn_mfma_f32_32x32x2f32(As[lm * SG_LDA + node], sr[q], u0, 0, 0, 0);
      u1 = __builtin_amdgcn_mfma_f32_32x32x2f32(As[(32 + lm) * SG_LDA + node], sr[q], u1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) R = __builtin_amdgcn_mfma_f32_32x32x2f32(sr[r], u0[r], R, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) R = __builtin_amdgcn_mfma_f32_32x32x2f32(sr[16 + r], u1[r], R, 0, 0, 0);
  }
  // ---- the loss terms' coefficients: gR[i][i] += cdiag, gS[n][j] += c1 deg[n] S[n][j], gS += S W (W = 2 gG) ---------
  float cdiag = 0.f, c1 = 0.f;
  f32x16 W;
#pragma unroll
  for (int r = 0; r < 16; ++r) W[r] = 0.f;
  bool have_w = false;
  {
    // degrees of the logical adjacency: lane i sums row i (rows / columns beyond N are zero)
    float dsum = 0.f;
    const float* rowp = As + lane * SG_LDA;
#pragma unroll 16
    for (int j = 0; j < SG_N; ++j) dsum += rowp[j];
    s_deg[lane] = dsum;
  }
  __builtin_amdgcn_wave_barrier();
  float c_link = 0.f, c_ent = 0.f;
  if (p.diff_losses) {
    const float lv = p.diff_losses[0];
    if (p.g_link) c_link = lv != 0.f ? *p.g_link * p.link_scale * p.link_scale / lv : 0.f;  // g * link_scale / ||A - S S^T||
    if (p.g_ent) c_ent = *p.g_ent * p.ent_scale;
  }
  float gt_cut = p.g_terms ? p.g_terms[b] : 0.f, gt_ortho = p.g_terms ? p.g_terms[p.B + b] : 0.f;
  if (p.g_mean_cut) gt_cut += *p.g_mean_cut / static_cast<float>(p.B);
  if (p.g_mean_ortho) gt_ortho += *p.g_mean_ortho / static_cast<float>(p.B);
  if (p.g_terms || p.g_mean_cut) {
    float tr = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (rho(r) + 4 * lk == lm) tr += R[r];
    tr = sg_wave_sum(tr);
    float den = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      const int node = 32 * (q >> 4) + rho(q & 15) + 4 * lk;
      den = fmaf(s_deg[node], sr[q] * sr[q], den);
    }
    den = sg_wave_sum(den) + p.loss_eps;
    cdiag = -gt_cut / den;
    c1 = 2.0f * gt_cut * tr / (den * den);
  }
  if (gt_ortho != 0.f || c_link != 0.f) {  // (uniform) both need S^T S
    f32x16 gg;
#pragma unroll
    for (int r = 0; r < 16; ++r) gg[r] = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) gg = __builtin_amdgcn_mfma_f32_32x32x2f32(sr[q], sr[q], gg, 0, 0, 0);
    if (gt_ortho != 0.f) {
      float fro = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) fro = fmaf(gg[r], gg[r], fro);
      const float ng2 = sg_wave_sum(fro), ng = sqrtf(ng2);
      const float tdiag = 1.0f / sqrtf(static_cast<float>(K));
      float y[16], ny2 = 0.f, gy = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rho(r) + 4 * lk;
        y[r] = (row < K && lm < K) ? gg[r] / ng - (row == lm ? tdiag : 0.f) : 0.f;
        ny2 = fmaf(y[r], y[r], ny2);
        gy = fmaf(gg[r], y[r], gy);
      }
      ny2 = sg_wave_sum(ny2);
      gy = sg_wave_sum(gy);
      const float ny = sqrtf(ny2);
      const float coef = ny > 0.f ? 2.0f * gt_ortho / (ny * ng) : 0.f;  // (the factor 2: gG + gG^T, gG symmetric)
#pragma unroll
      for (int r = 0; r < 16; ++r) W[r] = coef * (y[r] - gg[r] * (gy / ng2));
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) W[r] = fmaf(2.0f * c_link, gg[r], W[r]);
    have_w = true;
  }

  // ---- U^T = S^T A^T and V^T = S^T A: lane = node (two tiles of 32), register r = cluster rho(r) + 4 lk -------------
  f32x16 ut0, ut1, vt0, vt1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { ut0[r] = 0.f; ut1[r] = 0.f; vt0[r] = 0.f; vt1[r] = 0.f; }
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const int node = 32 * (q >> 4) + rho(q & 15) + 4 * lk;
    ut0 = __builtin_amdgcn_mfma_f32_32x32x2f32(sr[q], As[lm * SG_LDA + node], ut0, 0, 0, 0);
    ut1 = __builtin_amdgcn_mfma_f32_32x32x2f32(sr[q], As[(32 + lm) * SG_LDA + node], ut1, 0, 0, 0);
    vt0 = __builtin_amdgcn_mfma_f32_32x32x2f32(sr[q], As[node * SG_LDA + lm], vt0, 0, 0, 0);
    vt1 = __builtin_amdgcn_mfma_f32_32x32x2f32(sr[q], As[node * SG_LDA + 32 + lm], vt1, 0, 0, 0);
  }
  const float deg0 = s_deg[lm], deg1 = s_deg[32 + lm];
  __builtin_amdgcn_wave_barrier();  // the A tile is spent: its LDS is the transposition scratch from here on

  // ---- gR (C/D layout) -------------------------------------------------------------------------------------------
  const int mx = (p.grad_bcast & 1) ? 0 : 1, ma = (p.grad_bcast & 2) ? 0 : 1;  // 0: the gradient is one broadcast value
  f32x16 gR;
  {
    float gP[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = rho(r) + 4 * lk;
      const bool ok = p.g_adj_pool && i < K && lm < K;
      const float t = *byte_off(p.g_adj_pool ? p.g_adj_pool + static_cast<long>(b) * K * K * ma : p.S,
                                ok ? (i * K + lm) * 4 * ma : 0);
      gP[r] = ok ? t : 0.f;
    }
    if (p.g_adj_pool && (p.flags & TGP_DEGREE_NORM)) {
      float r0[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        r0[r] = R[r];
        if ((p.flags & TGP_REMOVE_SELF_LOOPS) && rho(r) + 4 * lk == lm) r0[r] = 0.f;
      }
      // degree of index lm, identical on both half-waves (the forward kernel's code)
      float dcol;
      if (p.flags & TGP_SUM_AXIS_ROWS) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += r0[r];
        s += __shfl_xor(s, 32, WAVE);
        dcol = s;
      } else {
        float mine = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float s = r0[r];
#pragma unroll
          for (int d = 16; d > 0; d >>= 1) s += __shfl_xor(s, d, WAVE);
          if (lm == rho(r) + 4 * lk) mine = s;
        }
        const float other = __shfl_xor(mine, 32, WAVE);
        bool own = false;
#pragma unroll
        for (int r = 0; r < 16; ++r) own |= (rho(r) + 4 * lk == lm);
        dcol = own ? mine : other;
      }
      const float d = sqrtf(fmaxf(dcol, p.eps));
      const bool pass = dcol >= p.eps;  // clamp(min = eps) lets the gradient through where the sum is not below eps
      float qv[16], colq = 0.f, rowmine = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rho(r) + 4 * lk;
        const float drow = __shfl(d, row, WAVE);
        const float first = (p.flags & TGP_SUM_AXIS_ROWS) ? d : drow;
        const float second = (p.flags & TGP_SUM_AXIS_ROWS) ? drow : d;
        const float pv = (r0[r] / first) / second;
        qv[r] = gP[r] * pv;
        colq += qv[r];
        float s = qv[r];
#pragma unroll
        for (int dd = 16; dd > 0; dd >>= 1) s += __shfl_xor(s, dd, WAVE);
        if (lm == row) rowmine = s;
        gP[r] = gP[r] / (drow * d);
      }
      colq += __shfl_xor(colq, 32, WAVE);
      {
        const float other = __shfl_xor(rowmine, 32, WAVE);
        bool own = false;
#pragma unroll
        for (int r = 0; r < 16; ++r) own |= (rho(r) + 4 * lk == lm);
        rowmine = own ? rowmine : other;
      }
      const float gs = pass ? -(rowmine + colq) / (2.0f * d * d) : 0.f;  // d loss / d (degree sum of index lm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rho(r) + 4 * lk;
        const float gsrow = __shfl(gs, row, WAVE);
        const float add = (p.flags & TGP_SUM_AXIS_ROWS) ? gs : gsrow;
        gP[r] = (row < K && lm < K) ? gP[r] + add : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = rho(r) + 4 * lk;
      float g = gP[r];
      if ((p.flags & TGP_REMOVE_SELF_LOOPS) && i == lm) g = 0.f;  // the diagonal of R never reaches A'
      const bool ok = p.g_adj_raw && i < K && lm < K;
      const float t = *byte_off(p.g_adj_raw ? p.g_adj_raw + static_cast<long>(b) * K * K : p.S, ok ? (i * K + lm) * 4 : 0);
      g += ok ? t : 0.f;
      if (i == lm && i < K) g += cdiag;
      gR[r] = g;
    }
  }
  // gRt[r] = gR[lm][rho(r) + 4 lk]: through LDS ([row][col], 33 floats per row)
  f32x16 gRt;
  {
    float* sT = As;
#pragma unroll
    for (int r = 0; r < 16; ++r) sT[(rho(r) + 4 * lk) * 33 + lm] = gR[r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) gRt[r] = sT[lm * 33 + rho(r) + 4 * lk];
    __builtin_amdgcn_wave_barrier();
  }
  // ---- S and X with lane = node, the upstream gradient of X' ---------------------------------------------------------
  float sn0[16], sn1[16];
  {
    const float* Sb = p.S + static_cast<long>(b) * N * K;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = rho(r) + 4 * lk;
      const bool ok0 = lm < N && j < K, ok1 = 32 + lm < N && j < K;
      const float a0 = *byte_off(Sb, ok0 ? (lm * K + j) * 4 : 0), a1 = *byte_off(Sb, ok1 ? ((32 + lm) * K + j) * 4 : 0);
      sn0[r] = ok0 ? a0 : 0.f;
      sn1[r] = ok1 ? a1 : 0.f;
    }
  }
  f32x16 gs0, gs1;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    gs0[r] = c1 * deg0 * sn0[r] - c_link * (ut0[r] + vt0[r]);
    gs1[r] = c1 * deg1 * sn1[r] - c_link * (ut1[r] + vt1[r]);
  }
  if (c_ent != 0.f) {  // every stored element of S, padded rows included (autograd's gradient of -s log(s + eps) at 0)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool jok = rho(r) + 4 * lk < K;
      if (jok && lm < N) gs0[r] -= c_ent * (__logf(sn0[r] + p.ent_eps) + sn0[r] / (sn0[r] + p.ent_eps));
      if (jok && 32 + lm < N) gs1[r] -= c_ent * (__logf(sn1[r] + p.ent_eps) + sn1[r] / (sn1[r] + p.ent_eps));
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    gs0 = __builtin_amdgcn_mfma_f32_32x32x2f32(gRt[r], ut0[r], gs0, 0, 0, 0);
    gs1 = __builtin_amdgcn_mfma_f32_32x32x2f32(gRt[r], ut1[r], gs1, 0, 0, 0);
    gs0 = __builtin_amdgcn_mfma_f32_32x32x2f32(gR[r], vt0[r], gs0, 0, 0, 0);
    gs1 = __builtin_amdgcn_mfma_f32_32x32x2f32(gR[r], vt1[r], gs1, 0, 0, 0);
  }
  if (have_w) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      gs0 = __builtin_amdgcn_mfma_f32_32x32x2f32(W[r], sn0[r], gs0, 0, 0, 0);
      gs1 = __builtin_amdgcn_mfma_f32_32x32x2f32(W[r], sn1[r], gs1, 0, 0, 0);
    }
  }
  if (p.g_x_pool && p.X) {
    const float* Xb = p.X + static_cast<long>(b) * N * F;
    const float* Gb = p.g_x_pool + static_cast<long>(b) * K * F * mx;
    float ga[16], xn0[16], xn1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = rho(r) + 4 * lk;
      const bool okg = lm < K && f < F, ok0 = lm < N && f < F, ok1 = 32 + lm < N && f < F;
      const float tg = *byte_off(Gb, okg ? (lm * F + f) * 4 * mx : 0);
      const float t0 = *byte_off(Xb, ok0 ? (lm * F + f) * 4 : 0), t1 = *byte_off(Xb, ok1 ? ((32 + lm) * F + f) * 4 : 0);
      ga[r] = okg ? tg : 0.f;
      xn0[r] = ok0 ? t0 : 0.f;
      xn1[r] = ok1 ? t1 : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      gs0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[r], xn0[r], gs0, 0, 0, 0);
      gs1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[r], xn1[r], gs1, 0, 0, 0);
    }
  }
  // ---- gS out: accumulators hold gS^T (lane = node): through LDS so that a row leaves as one contiguous run -----------
  {
    float* sT = As;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sT[lm * 33 + rho(r) + 4 * lk] = gs0[r];
      sT[(32 + lm) * 33 + rho(r) + 4 * lk] = gs1[r];
    }
    __builtin_amdgcn_wave_barrier();
    float* o = p.gS + static_cast<long>(b) * N * K;
#pragma unroll 4
    for (int t = 0; t < 32; ++t) {
      const int n = 2 * t + lk;
      if (n < N && lm < K) o[n * K + lm] = sT[n * 33 + lm];
    }
    __builtin_amdgcn_wave_barrier();
  }
  // ---- gX^T = gX'^T S^T (lane = node, register = feature) ----------------------------------------------------------
  if (p.gX) {
    f32x16 gx0, gx1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { gx0[r] = 0.f; gx1[r] = 0.f; }
    if (p.g_x_pool) {
      const float* Gb = p.g_x_pool + static_cast<long>(b) * K * F * mx;
      float gb[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = rho(r) + 4 * lk;
        const bool ok = j < K && lm < F;
        const float t = *byte_off(Gb, ok ? (j * F + lm) * 4 * mx : 0);
        gb[r] = ok ? t : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        gx0 = __builtin_amdgcn_mfma_f32_32x32x2f32(gb[r], sn0[r], gx0, 0, 0, 0);
        gx1 = __builtin_amdgcn_mfma_f32_32x32x2f32(gb[r], sn1[r], gx1, 0, 0, 0);
      }
    }
    float* sT = As;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sT[lm * 33 + rho(r) + 4 * lk] = gx0[r];
      sT[(32 + lm) * 33 + rho(r) + 4 * lk] = gx1[r];
    }
    __builtin_amdgcn_wave_barrier();
    float* o = p.gX + static_cast<long>(b) * N * F;
#pragma unroll 4
    for (int t = 0; t < 32; ++t) {
      const int n = 2 * t + lk;
      if (n < N && lm < F) o[n * F + lm] = sT[n * 33 + lm];
    }
  }
}

// ------------------------------------------------------------------------------------------
// Medium graphs (TU-dataset-sized batches: N up to a few hundred, K <= 64): one WORKGROUP (4 waves) owns one
// graph, every byte of A / X crosses HBM once and nothing intermediate leaves the CU.
//   * S [N,K] is copied to LDS once (zero padded to 32-row / 32-column multiples); every MFMA reads one of its
//     operands from there (lane = cluster: consecutive words, conflict-free).
//   * A and X are read straight from HBM into the MFMA B-operand registers: lane = column, so a half-wave
//     reads 128 contiguous bytes of one row per k-step -- the operand layout IS the memory layout, no staging
//     (that is why the product is associated as (S^T A) S here; (A S) would need A transposed through LDS).
//     Buffer-descriptor loads: rows / columns outside the graph come back as zeros from the range check.
//   * work items = 32-column strips of A (then of X), dealt round-robin to the waves.  An A strip gives
//     P = (S^T A)^T restricted to the strip, [32 nodes x K], with the strip as the MFMA A operand; register r
//     of the C/D layout holds strip rows (rho(r), rho(r)+4) on the two half-waves = a k-pair of a B operand,
//     so P goes straight from the accumulators into A'[c1][c2] += sum_n P[n][c1] S[n][c2] (no LDS round trip).
//     The four waves' partial A' are added in wave order (deterministic), then the workgroup post-processes
//     the K x K result in LDS (utils/ops.py:282-335) and stores it.
// If memory holds A^T (TGP_ADJ_TRANSPOSED) the same program yields (A')^T, which is transposed on the way
// into the post-processing buffer.
// ------------------------------------------------------------------------------------------
struct MediumArgs {
  const float* S; const float* A; const float* X;
  int B, N, K, F, flags;
  float eps;
  float* x_pool; float* adj_raw; float* adj_pool;
  int npad;  // N rounded up to 32
  // optional [B]: graph b's real nodes are its first sizes[b] rows (to_dense_batch layout, src.py:448-450); the rest
  // of the padded tensors is zero by construction, so the loops stop there.  A real batch pads every graph to the
  // longest one (PROTEINS: 39 nodes on average, 620 at most), which is where most of the padded work goes.
  const int64_t* sizes;
};

template <int MT>
static size_t medium_lds_bytes(int64_t npad) {
  constexpr int KP = 32 * MT;
  return (static_cast<size_t>(npad) * KP + KP * (KP + 1) + KP) * sizeof(float);
}

// MINW = waves per SIMD the register allocation is held to.  K <= 32 (MT = 1): 4.  K in (32, 64] (MT = 2): 64 + 32
// accumulator registers plus two operand sets do not fit the 168 registers of 3 waves per SIMD -- that build spilled 34
// VGPRs to scratch inside the strip loop (verdict r4 item 9) -- so MT = 2 is compiled for 2 (256 registers, no scratch);
// S [N][64] + the K x K result in LDS leave room for two or three workgroups per CU anyway.  <2, 3> stays for A/B
// (TGP_MEDIUM_MINW=3).
// WAVES (r5, late): 4, or 8 for graphs with at least eight strips whose S tile leaves one workgroup per CU -- with four
// waves that is ONE wave per SIMD and nothing but the wave's own double buffering hides a load; eight waves put two on
// every SIMD and halve the strips per wave.
template <int MT, int MINW = (MT == 1 ? 4 : 2), int WAVES = 4>
__global__ __launch_bounds__(64 * WAVES, MINW) void dense_pool_medium_kernel(MediumArgs p) {
  constexpr int NT = 64 * WAVES;  // threads
  constexpr int KP = 32 * MT;          // padded K
  constexpr int UNROLL = 8;            // k-pairs whose operands are requested together (two such sets in flight;
                                       // 16 measured no faster for K <= 32 and spills for K <= 64)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, lm = lane & 31, lk = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = p.N, K = p.K, F = p.F, NP = p.npad;
  const int b = blockIdx.x;
  int NV = N;  // valid node rows of this graph
  if (p.sizes) {
    const int64_t sz = p.sizes[b];
    NV = sz < N ? (sz > 0 ? static_cast<int>(sz) : 0) : N;
  }
  const int NPV = (NV + 31) & ~31;
  float* Ss = smem;                       // [NP][KP]
  float* Rs = Ss + NP * KP;               // [KP][KP+1]
  float* ds = Rs + KP * (KP + 1);         // [KP] degrees
  const bool want_a = p.A && (p.adj_raw || p.adj_pool);
  const bool want_x = p.X && p.x_pool;

  TGP_WSTAMP(0);
  // ---- S -> LDS (zero padded) ------------------------------------------------------------
  {
    // eight independent loads in flight per thread: a one-element-per-iteration loop exposes the full load latency
    // NP * KP / 256 times (measured: 22 us of a 92 us workgroup at N = 200, K = 50)
    const float* Sb = p.S + static_cast<long>(b) * N * K;
    constexpr int UB = 8;
    for (int base = 0; base < NPV * KP; base += NT * UB) {
      float v[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int e = base + u * NT + tid;
        const int r = e / KP, c = e - r * KP;
        v[u] = (r < NV && c < K) ? Sb[r * K + c] : 0.f;  // r < NV also covers e beyond the tile (NPV >= NV)
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int e = base + u * NT + tid;
        if (e < NPV * KP) Ss[e] = v[u];
      }
    }
  }
  __syncthreads();

  TGP_WSTAMP(1);
  f32x16 racc[MT][MT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) racc[i][j][r] = 0.f;

  const int nt_a = want_a ? (NV + 31) / 32 : 0;
  const int nt_x = want_x ? (F + 31) / 32 : 0;
  constexpr int OOB = static_cast<int>(0x80000000u);
  // Strips are dealt round-robin, starting at a wave that rotates with the graph index: wave w always runs on SIMD
  // w, so a fixed start would pile every graph's extra strip onto the same SIMD of the CU.
  for (int job = (w + WAVES - (b & (WAVES - 1))) & (WAVES - 1); job < nt_a + nt_x; job += WAVES) {
    const bool is_a = job < nt_a;
    const int n0 = (is_a ? job : job - nt_a) * 32;
    const int ld = is_a ? N : F;
    const float* src = is_a ? p.A + static_cast<long>(b) * N * N : p.X + static_cast<long>(b) * N * F;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, N * ld * 4, 0x00020000);
    const int voff = (n0 + lm < ld) ? (lk * ld + n0 + lm) * 4 : OOB;
    // strip element (node row k + lk, column n0 + lm) of A or X; two register sets: the next batch of k-pairs
    // is requested before the MFMAs of the current one
    float gv[2][UNROLL];
    auto request = [&](int set, int k0) {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
        gv[set][u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (k0 + 2 * u) * ld * 4, 0));
    };
    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // X strip:  X'[c][f]  = sum_k S[k][c] X[k][f]    (S = A operand from LDS, the strip = B operand)
    // A strip:  P [n][c]  = sum_k A[k][n] S[k][c]    (the strip = A operand, S = B operand from LDS) = T^T
    auto consume = [&](auto is_a_c, int set, int k0) {
      constexpr bool IS_A = decltype(is_a_c)::value;
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int k = k0 + 2 * u + lk;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const float sv = Ss[k * KP + i * 32 + lm];
          if constexpr (IS_A) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv[set][u], sv, acc[i], 0, 0, 0);
          else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv, gv[set][u], acc[i], 0, 0, 0);
        }
      }
    };
    // NPV is a multiple of 32 = 4 * UNROLL node rows: every round consumes both register sets.  All rounds but the last
    // request the next round's first set UNCONDITIONALLY (r5, late): with `if (more) request(...)` in the loop, as r1-r5
    // had it, the wait in front of a set's MFMAs could only be written as vmcnt(0) -- it also waited for the set that
    // had just been requested, so the two sets were never in flight together (the k-loop finding of gemm_mfma.h).
    auto k_loop = [&](auto is_a_c) {
      if (NPV <= 0) return;
      // (the scheduling barriers keep a set's loads IN FRONT of the other set's MFMAs: left alone, the scheduler sinks
      //  each load to the MFMA that frees its register, i.e. one set of registers and a quarter of the time in flight)
      request(0, 0);
      int k0 = 0;
      for (; k0 + 4 * UNROLL < NPV; k0 += 4 * UNROLL) {
        request(1, k0 + 2 * UNROLL);
        __builtin_amdgcn_sched_barrier(0);
        consume(is_a_c, 0, k0);
        request(0, k0 + 4 * UNROLL);
        __builtin_amdgcn_sched_barrier(0);
        consume(is_a_c, 1, k0 + 2 * UNROLL);
      }
      request(1, k0 + 2 * UNROLL);
      __builtin_amdgcn_sched_barrier(0);
      consume(is_a_c, 0, k0);
      consume(is_a_c, 1, k0 + 2 * UNROLL);
    };
    if (is_a) k_loop(std::true_type{});   // wave-uniform branch: one operand order per loop body
    else k_loop(std::false_type{});
    if (!is_a) {  // X' strip: rows = clusters, cols = features n0 .. n0+31
      if (n0 + lm < F) {
        float* o = p.x_pool + static_cast<long>(b) * K * F;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c = i * 32 + rho(r) + 4 * lk;
            if (c < K) o[c * F + n0 + lm] = acc[i][r];
          }
      }
      continue;
    }
    // R[c1][c2] = sum_n P[n][c1] S[n][c2]: register r of the C/D layout holds strip rows (rho(r), rho(r)+4) on the
    // two half-waves, which is exactly a k-pair of a B operand, so P never leaves the accumulators:
    //   D[row = c2][col = c1] += S[n0 + rho(r) + 4 lk][c2]  x  P_r[c1]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float sv[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) sv[i] = Ss[(n0 + rho(r) + 4 * lk) * KP + i * 32 + lm];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
          racc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[i], acc[j][r], racc[i][j], 0, 0, 0);
    }
  }
  TGP_WSTAMP(2);
  if (!want_a) return;

  // ---- A' = sum of the four partial products, in wave order ---------------------------------
  const bool at = p.flags & TGP_ADJ_TRANSPOSED;
  for (int turn = 0; turn < WAVES; ++turn) {
    if (w == turn) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c2 = i * 32 + rho(r) + 4 * lk, c1 = j * 32 + lm;  // the accumulators hold R[c1][c2]
            float* d = at ? &Rs[c2 * (KP + 1) + c1] : &Rs[c1 * (KP + 1) + c2];
            *d = turn == 0 ? racc[i][j][r] : __fadd_rn(*d, racc[i][j][r]);
          }
    }
    __syncthreads();
  }

  TGP_WSTAMP(3);
  // ---- post-processing on the K x K result (utils/ops.py:282-335) ---------------------------
  // element loops run over the padded [K][KP] index space: row / column come from shifts, not divisions
  const long obase = static_cast<long>(b) * K * K;
  if (p.adj_raw) {
    for (int e = tid; e < K * KP; e += NT) {
      const int i = e / KP, j = e % KP;
      if (j < K) p.adj_raw[obase + i * K + j] = Rs[i * (KP + 1) + j];
    }
    __syncthreads();  // the diagonal is cleared next
  }
  if (!p.adj_pool) return;
  if (p.flags & TGP_REMOVE_SELF_LOOPS) {
    if (tid < K) Rs[tid * (KP + 1) + tid] = 0.f;
    __syncthreads();
  }
  if (p.flags & TGP_DEGREE_NORM) {
    const bool rows = p.flags & TGP_SUM_AXIS_ROWS;
    if (tid < K) {
      float t = 0.f;
      for (int q = 0; q < K; ++q) t = __fadd_rn(t, rows ? Rs[q * (KP + 1) + tid] : Rs[tid * (KP + 1) + q]);
      ds[tid] = sqrtf(fmaxf(t, p.eps));
    }
    __syncthreads();
    for (int e = tid; e < K * KP; e += NT) {
      const int i = e / KP, j = e % KP;
      if (j < K) {
        const float first = rows ? ds[j] : ds[i], second = rows ? ds[i] : ds[j];
        Rs[i * (KP + 1) + j] = (Rs[i * (KP + 1) + j] / first) / second;
      }
    }
    __syncthreads();
  }
  float scale = 1.f;
  if (p.flags & TGP_EDGE_WEIGHT_NORM) {
    float m = 0.f;
    for (int e = tid; e < K * KP; e += NT) {
      const int i = e / KP, j = e % KP;
      if (j < K) m = fmaxf(m, fabsf(Rs[i * (KP + 1) + j]));
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, WAVE));
    __syncthreads();
    if (lane == 0) ds[w] = m;
    __syncthreads();
    scale = 0.f;
#pragma unroll
    for (int q = 0; q < WAVES; ++q) scale = fmaxf(scale, ds[q]);
    if (scale == 0.f) scale = 1.f;
  }
  for (int e = tid; e < K * KP; e += NT) {
    const int i = e / KP, j = e % KP;
    if (j < K) {
      const float v = Rs[i * (KP + 1) + j];
      p.adj_pool[obase + i * K + j] = (p.flags & TGP_EDGE_WEIGHT_NORM) ? v / scale : v;
    }
  }
  TGP_WSTAMP(4);
}


// Batched products of SMALL matrices (the backward of the dense poolers on TU-dataset-sized graphs: [N x F][F x K],
// [N x N][N x K], ... with N up to a few hundred and at most 64 output columns): the LDS-tiled kernel above spends
// such a launch on 64 / 128-wide tiles that are mostly padding.  Here one WAVE owns a 32-row strip of one batch
// element and keeps 32 x Nc of C in its accumulators; both operands go straight from memory into the MFMA operand
// registers (B: lane = column, coalesced; A: lane = row, each lane walks its own row, so every 64-byte line is
// fetched once and serves 16 k-steps from L1), eight k-pairs requested at a time.  No LDS, no barriers.
struct SmallBmmArgs {
  const float* A; const float* Bm; float* C;
  int M, Nc, Kd, trans_a;
  long lda, ldb, ldc, sA, sB, sC;
  int strips;  // 32-row strips per batch element
};

template <int NT>
__global__ __launch_bounds__(256) void small_bmm_kernel(SmallBmmArgs p, long total_strips) {
  const long strip = static_cast<long>(blockIdx.x) * 4 + wave_id();
  if (strip >= total_strips) return;
  const int lane = lane_id(), lm = lane & 31, lk = lane >> 5;
  const long b = strip / p.strips;
  const int m0 = static_cast<int>(strip - b * p.strips) * 32;
  const float* A = p.A + b * p.sA;
  const float* Bm = p.Bm + b * p.sB;
  float* C = p.C + b * p.sC;
  const bool row_ok = m0 + lm < p.M;
  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  constexpr int U = 8;
  for (int k0 = 0; k0 < p.Kd; k0 += 2 * U) {
    float av[U], bv[U][NT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + 2 * u + lk;
      const bool k_ok = k < p.Kd;
      const long a_off = p.trans_a ? static_cast<long>(k) * p.lda + m0 + lm : static_cast<long>(m0 + lm) * p.lda + k;
      av[u] = (k_ok && row_ok) ? A[a_off] : 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = j * 32 + lm;
        bv[u][j] = (k_ok && n < p.Nc) ? Bm[static_cast<long>(k) * p.ldb + n] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u][j], acc[j], 0, 0, 0);
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = j * 32 + lm;
    if (n >= p.Nc) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + rho(r) + 4 * lk;
      if (m < p.M) C[static_cast<long>(m) * p.ldc + n] = acc[j][r];
    }
  }
}

}  // namespace tgp
