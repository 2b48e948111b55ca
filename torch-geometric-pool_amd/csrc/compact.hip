// Output contract of the single-pass sparse operators (r5).  tgp_sparse_pool_small_f32 and tgp_connect_subgraph_single
// write survivors at their final offsets of CAPACITY-sized buffers (the count is only known afterwards).  The reference
// hands out new tensors of exactly the pooled size (connect/base_conn.py:103-112, SURVEY 8(b) "Ownership"), so by default
// the host mirror now allocates exact outputs once the count has arrived and this one launch moves the first n entries
// of the capacity arrays there: edge_index contiguous [2, n], nothing pins E-sized scratch.  Callers that opt into views
// of the capacity buffers (tgp.kernels.output_views) skip it.
#include "common.h"
#include "lookback.h"
#include "primitives.h"

namespace tgp {

// 16-byte moves where source and destination allow it, scalar otherwise; one grid for all arrays (blockIdx.y = array)
constexpr int COMPACT_MAX = 8;
struct CompactArgs {
  const char* src[COMPACT_MAX];
  char* dst[COMPACT_MAX];
  int64_t bytes[COMPACT_MAX];
};

__global__ __launch_bounds__(256) void edges_compact_kernel(CompactArgs a) {
  const int k = blockIdx.y;
  const char* __restrict__ s = a.src[k];
  char* __restrict__ d = a.dst[k];
  const int64_t nb = a.bytes[k];
  if (!s || !d || nb <= 0) return;
  const int64_t tid = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x, nthr = static_cast<int64_t>(gridDim.x) * 256;
  if (((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0) {
    const int64_t nv = nb >> 4;
    const float4* s4 = reinterpret_cast<const float4*>(s);
    float4* d4 = reinterpret_cast<float4*>(d);
    for (int64_t i = tid; i < nv; i += nthr) d4[i] = s4[i];
    for (int64_t i = (nv << 4) + tid * 4; i < nb; i += nthr * 4)  // tail: sizes are multiples of 4 bytes
      *reinterpret_cast<uint32_t*>(d + i) = *reinterpret_cast<const uint32_t*>(s + i);
  } else if (((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d) | static_cast<uintptr_t>(nb)) & 7) == 0) {
    const int64_t nv = nb >> 3;
    const uint64_t* s8 = reinterpret_cast<const uint64_t*>(s);
    uint64_t* d8 = reinterpret_cast<uint64_t*>(d);
    for (int64_t i = tid; i < nv; i += nthr) d8[i] = s8[i];
  } else {
    const int64_t nv = nb >> 2;
    const uint32_t* s4 = reinterpret_cast<const uint32_t*>(s);
    uint32_t* d4 = reinterpret_cast<uint32_t*>(d);
    for (int64_t i = tid; i < nv; i += nthr) d4[i] = s4[i];
  }
}

}  // namespace tgp

using namespace tgp;

extern "C" int tgp_edges_compact(const int64_t* row, const int64_t* col, const void* weight, int weight_bytes,
                                 const int64_t* edge_id, int64_t n, int64_t* out_row, int64_t* out_col, void* out_weight,
                                 int64_t* out_edge_id, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(n >= 0 && (weight_bytes == 0 || weight_bytes == 4 || weight_bytes == 8), TGP_ERR_INVALID,
              "tgp_edges_compact: bad argument");
  if (n == 0) return TGP_OK;
  TGP_REQUIRE(row && col && out_row && out_col && (!weight == !out_weight) && (!edge_id == !out_edge_id) &&
                  (!weight || weight_bytes),
              TGP_ERR_INVALID, "tgp_edges_compact: null pointer");
  CompactArgs a{};
  a.src[0] = reinterpret_cast<const char*>(row); a.dst[0] = reinterpret_cast<char*>(out_row); a.bytes[0] = n * 8;
  a.src[1] = reinterpret_cast<const char*>(col); a.dst[1] = reinterpret_cast<char*>(out_col); a.bytes[1] = n * 8;
  a.src[2] = static_cast<const char*>(weight); a.dst[2] = static_cast<char*>(out_weight); a.bytes[2] = n * weight_bytes;
  a.src[3] = reinterpret_cast<const char*>(edge_id); a.dst[3] = reinterpret_cast<char*>(out_edge_id); a.bytes[3] = n * 8;
  int64_t blocks = (n * 8 / 16 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(edges_compact_kernel, dim3(static_cast<unsigned>(blocks), 4), dim3(256), 0, stream, a);
  return check_launch("tgp_edges_compact");
}

// Up to eight independent arrays in ONE launch (byte counts multiples of 4): the merged outputs of a gathered step leave
// the receive buffer as exact-size tensors this way (tgp.distributed.SparseGather: four clone launches per step before).
extern "C" int tgp_copy_arrays(const void* const* src, void* const* dst, const int64_t* bytes, int count, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(src && dst && bytes && count >= 0 && count <= COMPACT_MAX, TGP_ERR_INVALID, "tgp_copy_arrays: bad argument");
  CompactArgs a{};
  int64_t longest = 0;
  for (int i = 0; i < count; ++i) {
    TGP_REQUIRE(bytes[i] >= 0 && (bytes[i] & 3) == 0 && (bytes[i] == 0 || (src[i] && dst[i])), TGP_ERR_INVALID,
                "tgp_copy_arrays: array %d: null pointer or a size that is not a multiple of 4 bytes", i);
    a.src[i] = static_cast<const char*>(src[i]); a.dst[i] = static_cast<char*>(dst[i]); a.bytes[i] = bytes[i];
    if (bytes[i] > longest) longest = bytes[i];
  }
  if (longest == 0) return TGP_OK;
  int64_t blocks = (longest / 16 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(edges_compact_kernel, dim3(static_cast<unsigned>(blocks), static_cast<unsigned>(count)), dim3(256), 0,
                     stream, a);
  return check_launch("tgp_copy_arrays");
}

// ---- r6: a byte mask as the sorted list of its set positions (NDPSelect's kept nodes: the reference's `nonzero` of
// select/ndp_select.py:257-262 and the [2, k] index / value arrays of S around it).  Two launches with the host's size
// read between them: (1) per-tile counts, the LAST tile to arrive adds them up and stores {epoch, count} in a pinned word
// (count = -3 when `*declined` is non-zero: the producer of the mask refused its input); (2) every tile re-adds the counts
// in front of it (at most n / 4096 numbers) and writes its positions.  Fixed order, no atomics on the data path.
constexpr int MI_TILE = 4096;  // mask bytes per 256-thread workgroup (16 per thread)

__device__ __forceinline__ uint32_t mi_nonzero_bytes(uint32_t w) {  // one bit per non-zero byte, at the byte's bit 0
  uint32_t t = w | (w >> 4);
  t |= t >> 2;
  t |= t >> 1;
  return t & 0x01010101u;
}
// the 16 mask bytes of thread `tid` of tile `tile` as 16 flag bits (bit j = byte j is non-zero)
__device__ __forceinline__ uint32_t mi_flags16(const uint8_t* __restrict__ mask, int64_t n, int64_t base) {
  uint32_t bits = 0;
  if (base + 16 <= n && (reinterpret_cast<uintptr_t>(mask + base) & 15) == 0) {
    const uint4 v = *reinterpret_cast<const uint4*>(mask + base);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t f = mi_nonzero_bytes(w[q]);  // bits 0, 8, 16, 24
      bits |= ((f & 1u) | ((f >> 7) & 2u) | ((f >> 14) & 4u) | ((f >> 21) & 8u)) << (4 * q);
    }
  } else {
    for (int j = 0; j < 16; ++j)
      if (base + j < n && mask[base + j] != 0) bits |= 1u << j;
  }
  return bits;
}

__global__ __launch_bounds__(256) void mask_index_count_kernel(const uint8_t* __restrict__ mask, int64_t n, int ntiles,
                                                               uint32_t* __restrict__ tile_counts, uint32_t* ticket,
                                                               const int32_t* __restrict__ declined,
                                                               unsigned long long* result, unsigned long long tag) {
  __shared__ uint32_t s_w[4];
  __shared__ bool s_last;
  const int64_t base = static_cast<int64_t>(blockIdx.x) * MI_TILE + static_cast<int64_t>(threadIdx.x) * 16;
  const uint32_t mine = base < n ? __popc(mi_flags16(mask, n, base)) : 0u;
  uint32_t total;
  (void)block_excl_scan_256(mine, s_w, &total);
  if (threadIdx.x == 0) {
    tile_counts[blockIdx.x] = total;
    __threadfence();
    s_last = atomicAdd(ticket, 1u) == static_cast<uint32_t>(ntiles - 1);
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  uint32_t part = 0;
  for (int t = threadIdx.x; t < ntiles; t += 256) part += __hip_atomic_load(tile_counts + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint32_t sum;
  (void)block_excl_scan_256(part, s_w, &sum);
  if (threadIdx.x == 0) {
    *ticket = 0;  // (the next call on this stream finds it cleared)
    const bool refused = declined && *declined != 0;
    const unsigned long long count = refused ? ((1ull << SPS_EPOCH_SHIFT) - 3ull) : static_cast<unsigned long long>(sum);
    __hip_atomic_store(result, tag | count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ __launch_bounds__(256) void mask_index_fill_kernel(const uint8_t* __restrict__ mask, int64_t n,
                                                              const uint32_t* __restrict__ tile_counts, int64_t k,
                                                              int64_t* __restrict__ pos_out, int64_t* __restrict__ rank_out,
                                                              float* __restrict__ ones_out, int32_t* __restrict__ perm_out,
                                                              uint2* __restrict__ pack_out,
                                                              uint32_t* __restrict__ node_rank_out) {
  __shared__ uint32_t s_w[4];
  uint32_t part = 0;
  for (int t = threadIdx.x; t < static_cast<int>(blockIdx.x); t += 256) part += tile_counts[t];
  uint32_t before;
  (void)block_excl_scan_256(part, s_w, &before);
  const int64_t base = static_cast<int64_t>(blockIdx.x) * MI_TILE + static_cast<int64_t>(threadIdx.x) * 16;
  const uint32_t bits = base < n ? mi_flags16(mask, n, base) : 0u;
  int64_t at = static_cast<int64_t>(before) + block_excl_scan_256(__popc(bits), s_w, nullptr);
  if (node_rank_out && base < n) {  // set positions in front of every position of this thread's 16, total at [n]
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (base + j < n) node_rank_out[base + j] = static_cast<uint32_t>(at) + __popc(bits & ((1u << j) - 1u));
    if (base + 16 >= n) node_rank_out[n] = static_cast<uint32_t>(at) + __popc(bits);
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if ((bits >> j) & 1u) {
      if (at < k) {  // (k is what the count pass published for this mask: never exceeded)
        pos_out[at] = base + j;
        if (rank_out) rank_out[at] = at;
        if (ones_out) ones_out[at] = 1.0f;
        // the one-to-one inverted index of the assignment (supernode `at` owns node base + j with weight 1): what
        // tgp_one_to_one_index_build would make of (pos_out, rank_out, ones_out) in a launch of its own
        if (perm_out) perm_out[at] = static_cast<int32_t>(at);
        if (pack_out) pack_out[at] = make_uint2(static_cast<uint32_t>(base + j), 0x3F800000u);
      }
      ++at;
    }
  }
}

extern "C" int64_t tgp_mask_index_scratch_words(int64_t n) { return 2 + (n > 0 ? (n + MI_TILE - 1) / MI_TILE : 1); }

extern "C" int tgp_mask_index_count(const uint8_t* mask, int64_t n, const int32_t* declined, uint32_t* scratch,
                                    uint64_t* result, uint32_t epoch, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(n >= 0 && result && epoch > 0 && epoch < (1u << 30) && scratch && (n == 0 || mask), TGP_ERR_INVALID,
              "tgp_mask_index_count: bad argument");
  TGP_REQUIRE(n < (1ll << 31), TGP_ERR_RANGE, "tgp_mask_index_count: more than 2^31 - 1 mask bytes");
  const int ntiles = static_cast<int>(n > 0 ? (n + MI_TILE - 1) / MI_TILE : 1);
  // scratch: word 0 the arrival ticket (zero between calls: the caller clears the buffer once), words 2.. the tile counts
  hipLaunchKernelGGL(mask_index_count_kernel, dim3(ntiles), dim3(256), 0, stream, mask, n, ntiles, scratch + 2, scratch,
                     declined, reinterpret_cast<unsigned long long*>(result),
                     static_cast<unsigned long long>(epoch) << SPS_EPOCH_SHIFT);
  return check_launch("tgp_mask_index_count");
}

extern "C" int tgp_mask_index_fill(const uint8_t* mask, int64_t n, const uint32_t* scratch, int64_t k, int64_t* pos_out,
                                   int64_t* rank_out, float* ones_out, int32_t* perm_out, uint64_t* pack_out,
                                   uint32_t* node_rank_out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(n >= 0 && k >= 0 && k <= n && scratch && (n == 0 || mask) && (k == 0 || pos_out), TGP_ERR_INVALID,
              "tgp_mask_index_fill: bad argument");
  if (n == 0 || (k == 0 && !node_rank_out)) return TGP_OK;
  const int ntiles = static_cast<int>((n + MI_TILE - 1) / MI_TILE);
  hipLaunchKernelGGL(mask_index_fill_kernel, dim3(ntiles), dim3(256), 0, stream, mask, n, scratch + 2, k, pos_out, rank_out,
                     ones_out, perm_out, reinterpret_cast<uint2*>(pack_out), node_rank_out);
  return check_launch("tgp_mask_index_fill");
}

// r6 (fresh mini-batches: host time).  The host wait of a single-pass operator and the launch that makes its edge_index
// contiguous, in ONE call: spin on the pinned result word until call `epoch` has stored it, then -- unless the kernel
// refused the input (bit 31) -- move the n = word & 0x7fffffff columns the kernel left in `col_scratch` behind the n
// rows it wrote at the front of `out_rows` (capacity 2 E: [2, n] contiguous afterwards; rows and weights are already
// where they belong).  The caller reads the word itself from the pinned memory afterwards.
extern "C" int tgp_result_wait_pack_cols(const uint64_t* result, uint32_t epoch, const int64_t* col_scratch,
                                         int64_t* out_rows, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(result && epoch != 0, TGP_ERR_INVALID, "tgp_result_wait_pack_cols: bad argument");
  uint64_t word = 0;
  for (int64_t spins = 0;; ++spins) {
    word = __atomic_load_n(result, __ATOMIC_ACQUIRE);
    if ((word >> 34) == epoch) break;
    if (spins > (1ll << 24)) {  // a stream that is busy with something long: wait for it properly, then look once more
      TGP_REQUIRE(hipStreamSynchronize(stream) == hipSuccess, TGP_ERR_LAUNCH, "tgp_result_wait_pack_cols: the stream faulted");
      word = __atomic_load_n(result, __ATOMIC_ACQUIRE);
      TGP_REQUIRE((word >> 34) == epoch, TGP_ERR_LAUNCH,
                  "tgp_result_wait_pack_cols: the kernel finished without storing its result word");
      break;
    }
  }
  if (word & 0x80000000ull) return TGP_OK;
  const int64_t n = static_cast<int64_t>(word & 0x7fffffffull);
  if (n == 0) return TGP_OK;
  TGP_REQUIRE(col_scratch && out_rows, TGP_ERR_INVALID, "tgp_result_wait_pack_cols: null pointer");
  CompactArgs a{};
  a.src[0] = reinterpret_cast<const char*>(col_scratch); a.dst[0] = reinterpret_cast<char*>(out_rows + n); a.bytes[0] = n * 8;
  int64_t blocks = (n * 8 / 16 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(edges_compact_kernel, dim3(static_cast<unsigned>(blocks), 1), dim3(256), 0, stream, a);
  return check_launch("tgp_result_wait_pack_cols");
}
