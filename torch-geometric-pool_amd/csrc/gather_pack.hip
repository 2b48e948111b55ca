// the variable-size all-gather of pooled SPARSE outputs as ONE payload collective.
//
// A rank's pooled graphs -- x [K,F], batch [K] int64, edge_index [2,E] int64, edge_weight [E] -- are packed
// by one launch into one byte buffer behind a 128-byte header {magic, K, E, B, F, w_words, needed_bytes, x_words}; the
// values travel as 4-byte WORDS (r5: F = words per feature row, x_words / w_words = words per element: 1 = fp32 / int32,
// 2 = fp64 / int64 -- a pure copy, so float64 features and weights cross ranks bit for bit); the buffers of
// all ranks travel in ONE all_gather_into_tensor (capacity-padded: the capacity is agreed between the ranks from the
// headers themselves, no count exchange in front of the payload), and one launch unpacks the gathered buffer into the
// merged tensors, shifting the node ids / graph ids of rank r by the totals of the ranks before it -- the merge rule the
// reference uses when it collates pooled graphs on the CPU (tgp/data/collate.py:144-153).  (r3: a count exchange with
// a host read + four padded collectives + torch offset ops: 0.35 ms on a one-rank group against 0.09 ms of compute.)
#include "primitives.h"

namespace tgp {

constexpr int64_t GP_MAGIC = 0x7467705f67617468ll;  // "tgp_gath"
constexpr int GP_HEADER_WORDS = 16;                 // int64 words: 128 bytes

struct GpLayout {
  int64_t x, batch, row, col, w, end;  // byte offsets of the segments (16-byte aligned)
};
__host__ __device__ inline int64_t gp_align(int64_t v) { return (v + 15) & ~int64_t(15); }
__host__ __device__ inline GpLayout gp_layout(int64_t K, int64_t E, int64_t F, int w_words) {
  GpLayout l;
  l.x = GP_HEADER_WORDS * 8;
  l.batch = gp_align(l.x + K * F * 4);
  l.row = gp_align(l.batch + K * 8);
  l.col = gp_align(l.row + E * 8);
  l.w = gp_align(l.col + E * 8);
  l.end = gp_align(l.w + E * 4 * w_words);
  return l;
}

__device__ __forceinline__ void gp_pack_body(const float* __restrict__ x, int64_t x_stride,
                                             const int64_t* __restrict__ batch, const int64_t* __restrict__ row,
                                             const int64_t* __restrict__ col, const float* __restrict__ w, int64_t K,
                                             int64_t E, int64_t B, int64_t F, int w_words, int x_words,
                                             int64_t capacity, char* __restrict__ out) {
  if (!w) w_words = 0;
  const GpLayout l = gp_layout(K, E, F, w_words);
  const int64_t tid = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x, nthr = static_cast<int64_t>(gridDim.x) * 256;
  if (tid < GP_HEADER_WORDS) {
    const int64_t h[GP_HEADER_WORDS] = {GP_MAGIC, K, E, B, F, w_words, l.end, x_words, 0, 0, 0, 0, 0, 0, 0, 0};
    reinterpret_cast<int64_t*>(out)[tid] = h[tid];
  }
  if (l.end > capacity) return;  // does not fit: the header alone tells every rank how much room is needed
  float* ox = reinterpret_cast<float*>(out + l.x);
  for (int64_t i = tid; i < K * F; i += nthr) ox[i] = x[(i / F) * x_stride + (i % F)];
  int64_t* ob = reinterpret_cast<int64_t*>(out + l.batch);
  for (int64_t i = tid; i < K; i += nthr) ob[i] = batch ? batch[i] : 0;
  int64_t* orow = reinterpret_cast<int64_t*>(out + l.row);
  int64_t* ocol = reinterpret_cast<int64_t*>(out + l.col);
  for (int64_t i = tid; i < E; i += nthr) {
    orow[i] = row[i];
    ocol[i] = col[i];
  }
  if (w) {
    float* ow = reinterpret_cast<float*>(out + l.w);
    for (int64_t i = tid; i < E * w_words; i += nthr) ow[i] = w[i];
  }
}

__global__ __launch_bounds__(256) void gp_pack_kernel(const float* __restrict__ x, int64_t x_stride,
                                                      const int64_t* __restrict__ batch, const int64_t* __restrict__ row,
                                                      const int64_t* __restrict__ col, const float* __restrict__ w,
                                                      int64_t K, int64_t E, int64_t B, int64_t F, int w_words,
                                                      int x_words, int64_t capacity, char* __restrict__ out) {
  gp_pack_body(x, x_stride, batch, row, col, w, K, E, B, F, w_words, x_words, capacity, out);
}

// r4, late: a whole BUCKET of steps in one launch (blockIdx.y = step): on a one-rank group, where nothing hides host time,
// the per-step pack and unpack launches were a third of what the gather added to a step.
constexpr int GP_MAX_STEPS = 8;
struct GpPackStep {
  const float* x;
  const int64_t *batch, *row, *col;
  const float* w;
  int64_t x_stride, K, E, B, F;
  int w_words, x_words;
};
struct GpPackArgs {
  GpPackStep s[GP_MAX_STEPS];
};
__global__ __launch_bounds__(256) void gp_pack_bucket_kernel(GpPackArgs a, int64_t capacity, char* __restrict__ out) {
  const GpPackStep& t = a.s[blockIdx.y];
  gp_pack_body(t.x, t.x_stride, t.batch, t.row, t.col, t.w, t.K, t.E, t.B, t.F, t.w_words, t.x_words, capacity,
               out + static_cast<int64_t>(blockIdx.y) * capacity);
}

// blockIdx.y = source rank; node ids += supernodes of the ranks before it, graph ids += their graphs
// `rank_stride`: bytes between two ranks' buffers in `gathered` (a bucket of several steps is gathered at once: the
// slot of this step sits at the same offset of every rank's bucket).  `result` (optional, pinned host memory): block
// (0, 0) leaves {tag, K total, E total, largest needed_bytes, status} there -- the caller polls word 0 for its tag instead
// of copying the headers back.  status bits: 1 = every header carries the magic, 2 = every rank packed the same row
// width / value sizes as the caller expects (`F`, `w_words`, `x_words`: ranks that disagree on the feature width, on
// having weights or on a dtype would otherwise be merged into views of uninitialised memory), 4 = the totals fit k_cap /
// e_cap.  Anything short of all three, or a payload that did not fit (needed > capacity), makes the launch a no-op apart
// from that report.
__device__ __forceinline__ void gp_unpack_body(const char* __restrict__ gathered, int64_t capacity,
                                               int64_t rank_stride, int world, int64_t k_cap, int64_t e_cap,
                                               int64_t F_want, int w_words_want, int x_words_want,
                                               float* __restrict__ x_out, int64_t* __restrict__ batch_out,
                                               int64_t* __restrict__ row_out, int64_t* __restrict__ col_out,
                                               float* __restrict__ w_out, unsigned long long* __restrict__ result,
                                               unsigned long long tag) {
  const int r = blockIdx.y;
  {
    int64_t kt = 0, et = 0, need = 0;
    bool ok = true, agree = true;
    for (int q = 0; q < world; ++q) {
      const int64_t* hq = reinterpret_cast<const int64_t*>(gathered + static_cast<int64_t>(q) * rank_stride);
      ok = ok && hq[0] == GP_MAGIC;
      agree = agree && hq[4] == F_want && hq[5] == w_words_want && hq[7] == x_words_want;
      kt += hq[1];
      et += hq[2];
      need = hq[6] > need ? hq[6] : need;
    }
    const bool room = kt <= k_cap && et <= e_cap;
    const bool fits = ok && agree && need <= capacity && room;
    if (result && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
      result[1] = static_cast<unsigned long long>(kt);
      result[2] = static_cast<unsigned long long>(et);
      result[3] = static_cast<unsigned long long>(need);
      result[4] = (ok ? 1ull : 0ull) | (ok && agree ? 2ull : 0ull) | (room ? 4ull : 0ull);
      __threadfence_system();
      __hip_atomic_store(result, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (!fits) return;
  }
  int64_t koff = 0, eoff = 0, goff = 0;
  for (int q = 0; q < r; ++q) {
    const int64_t* h = reinterpret_cast<const int64_t*>(gathered + static_cast<int64_t>(q) * rank_stride);
    koff += h[1];
    eoff += h[2];
    goff += h[3];
  }
  const char* src = gathered + static_cast<int64_t>(r) * rank_stride;
  const int64_t* h = reinterpret_cast<const int64_t*>(src);
  const int64_t K = h[1], E = h[2], F = h[4];
  const int w_words = static_cast<int>(h[5]);
  const GpLayout l = gp_layout(K, E, F, w_words);
  const int64_t tid = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x, nthr = static_cast<int64_t>(gridDim.x) * 256;
  const float* sx = reinterpret_cast<const float*>(src + l.x);
  for (int64_t i = tid; i < K * F; i += nthr) x_out[koff * F + i] = sx[i];
  const int64_t* sb = reinterpret_cast<const int64_t*>(src + l.batch);
  if (batch_out)
    for (int64_t i = tid; i < K; i += nthr) batch_out[koff + i] = sb[i] + goff;
  const int64_t* sr = reinterpret_cast<const int64_t*>(src + l.row);
  const int64_t* sc = reinterpret_cast<const int64_t*>(src + l.col);
  for (int64_t i = tid; i < E; i += nthr) {
    row_out[eoff + i] = sr[i] + koff;
    col_out[eoff + i] = sc[i] + koff;
  }
  if (w_out && w_words) {
    const float* sw = reinterpret_cast<const float*>(src + l.w);
    for (int64_t i = tid; i < E * w_words; i += nthr) w_out[eoff * w_words + i] = sw[i];
  }
}

__global__ __launch_bounds__(256) void gp_unpack_kernel(const char* __restrict__ gathered, int64_t capacity,
                                                        int64_t rank_stride, int world, int64_t k_cap, int64_t e_cap,
                                                        int64_t F, int w_words, int x_words,
                                                        float* __restrict__ x_out, int64_t* __restrict__ batch_out,
                                                        int64_t* __restrict__ row_out, int64_t* __restrict__ col_out,
                                                        float* __restrict__ w_out,
                                                        unsigned long long* __restrict__ result,
                                                        unsigned long long tag) {
  gp_unpack_body(gathered, capacity, rank_stride, world, k_cap, e_cap, F, w_words, x_words, x_out, batch_out, row_out,
                 col_out, w_out, result, tag);
}

struct GpUnpackStep {
  float* x;
  int64_t *batch, *row, *col;
  float* w;
  unsigned long long* result;
  int64_t k_cap, e_cap, F;
  int w_words, x_words;
  unsigned long long tag;
};
struct GpUnpackArgs {
  GpUnpackStep s[GP_MAX_STEPS];
};
// blockIdx.z = step of the bucket (its slot sits at z * capacity of every rank's bucket)
__global__ __launch_bounds__(256) void gp_unpack_bucket_kernel(const char* __restrict__ gathered, int64_t capacity,
                                                               int64_t rank_stride, int world, GpUnpackArgs a) {
  const GpUnpackStep& t = a.s[blockIdx.z];
  gp_unpack_body(gathered + static_cast<int64_t>(blockIdx.z) * capacity, capacity, rank_stride, world, t.k_cap, t.e_cap,
                 t.F, t.w_words, t.x_words, t.x, t.batch, t.row, t.col, t.w, t.result, t.tag);
}

}  // namespace tgp

using namespace tgp;

extern "C" int64_t tgp_gather_pack_bytes(int64_t K, int64_t E, int64_t F, int w_words) {
  return gp_layout(K, E, F, w_words).end;
}

extern "C" int tgp_gather_pack_f32(const float* x, int64_t x_stride, const int64_t* batch, const int64_t* row,
                                   const int64_t* col, const float* w, int64_t K, int64_t E, int64_t B, int64_t F,
                                   int w_words, int x_words, int64_t capacity, void* out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(K >= 0 && E >= 0 && B >= 0 && F >= 0 && capacity >= GP_HEADER_WORDS * 8 && out && w_words >= 0 &&
                  w_words <= 2 && x_words >= 1 && x_words <= 2,
              TGP_ERR_INVALID, "tgp_gather_pack_f32: bad argument");
  TGP_REQUIRE((K == 0 || F == 0 || x) && (E == 0 || (row && col)), TGP_ERR_INVALID, "tgp_gather_pack_f32: null pointer");
  const int64_t words = K * F + 2 * K + 6 * E + GP_HEADER_WORDS;
  int64_t blocks = (words + 256 * 8 - 1) / (256 * 8);
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(gp_pack_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, x, x_stride, batch, row,
                     col, w, K, E, B, F, w_words, x_words, capacity, static_cast<char*>(out));
  return check_launch("tgp_gather_pack_f32");
}

extern "C" int tgp_gather_unpack_f32(const void* gathered, int64_t capacity, int64_t rank_stride, int world,
                                     int64_t max_words, int64_t k_cap, int64_t e_cap, int64_t F, int w_words,
                                     int x_words, float* x_out, int64_t* batch_out, int64_t* row_out, int64_t* col_out,
                                     float* w_out, uint64_t* result, uint64_t tag, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(gathered && capacity >= GP_HEADER_WORDS * 8 && rank_stride >= capacity && world >= 1 && world <= 65535 &&
                  k_cap >= 0 && e_cap >= 0,
              TGP_ERR_INVALID, "tgp_gather_unpack_f32: bad argument");
  int64_t blocks = (max_words + 256 * 8 - 1) / (256 * 8);  // max_words: 4-byte words of the largest rank's payload
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(gp_unpack_kernel, dim3(static_cast<unsigned>(blocks), static_cast<unsigned>(world)), dim3(256), 0,
                     stream, static_cast<const char*>(gathered), capacity, rank_stride, world, k_cap, e_cap, F, w_words,
                     x_words, x_out, batch_out, row_out, col_out, w_out, reinterpret_cast<unsigned long long*>(result),
                     static_cast<unsigned long long>(tag));
  return check_launch("tgp_gather_unpack_f32");
}

extern "C" int tgp_gather_max_bucket_steps(void) { return GP_MAX_STEPS; }

// ptrs [n][5] = {x, batch, row, col, w} (batch / w: NULL ok), dims [n][7] = {x_stride, K, E, B, F, w_words, x_words}
// (x_stride and F in 4-byte words); step j is packed into out + j * capacity.
extern "C" int tgp_gather_pack_bucket_f32(const void* const* ptrs, const int64_t* dims, int n, int64_t capacity,
                                          void* out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(ptrs && dims && n >= 1 && n <= GP_MAX_STEPS && capacity >= GP_HEADER_WORDS * 8 && out, TGP_ERR_INVALID,
              "tgp_gather_pack_bucket_f32: bad argument");
  GpPackArgs a{};
  int64_t words = 0;
  for (int j = 0; j < n; ++j) {
    GpPackStep& t = a.s[j];
    t.x = static_cast<const float*>(ptrs[5 * j]);
    t.batch = static_cast<const int64_t*>(ptrs[5 * j + 1]);
    t.row = static_cast<const int64_t*>(ptrs[5 * j + 2]);
    t.col = static_cast<const int64_t*>(ptrs[5 * j + 3]);
    t.w = static_cast<const float*>(ptrs[5 * j + 4]);
    t.x_stride = dims[7 * j];
    t.K = dims[7 * j + 1];
    t.E = dims[7 * j + 2];
    t.B = dims[7 * j + 3];
    t.F = dims[7 * j + 4];
    t.w_words = static_cast<int>(dims[7 * j + 5]);
    t.x_words = static_cast<int>(dims[7 * j + 6]);
    TGP_REQUIRE(t.K >= 0 && t.E >= 0 && t.B >= 0 && t.F >= 0 && (t.K == 0 || t.F == 0 || t.x) &&
                    (t.E == 0 || (t.row && t.col)) && t.w_words >= 0 && t.w_words <= 2 && t.x_words >= 1 &&
                    t.x_words <= 2,
                TGP_ERR_INVALID, "tgp_gather_pack_bucket_f32: bad step %d", j);
    const int64_t wj = t.K * t.F + 2 * t.K + 6 * t.E + GP_HEADER_WORDS;
    words = wj > words ? wj : words;
  }
  int64_t blocks = (words + 256 * 8 - 1) / (256 * 8);
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(gp_pack_bucket_kernel, dim3(static_cast<unsigned>(blocks), static_cast<unsigned>(n)), dim3(256), 0,
                     stream, a, capacity, static_cast<char*>(out));
  return check_launch("tgp_gather_pack_bucket_f32");
}

// ptrs [n][6] = {x, batch, row, col, w, result} outputs of step j (batch / w / result: NULL ok), dims [n][6] = {k_cap,
// e_cap, tag, F, w_words, x_words} (what this rank packed: every rank's header must say the same); `rank_stride`: bytes
// between two ranks' buckets in `gathered`.
extern "C" int tgp_gather_unpack_bucket_f32(const void* gathered, int64_t capacity, int64_t rank_stride, int world,
                                            int64_t max_words, int n, void* const* ptrs, const int64_t* dims,
                                            void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TGP_REQUIRE(gathered && ptrs && dims && n >= 1 && n <= GP_MAX_STEPS && capacity >= GP_HEADER_WORDS * 8 &&
                  rank_stride >= static_cast<int64_t>(n) * capacity && world >= 1 && world <= 65535,
              TGP_ERR_INVALID, "tgp_gather_unpack_bucket_f32: bad argument");
  GpUnpackArgs a{};
  for (int j = 0; j < n; ++j) {
    GpUnpackStep& t = a.s[j];
    t.x = static_cast<float*>(ptrs[6 * j]);
    t.batch = static_cast<int64_t*>(ptrs[6 * j + 1]);
    t.row = static_cast<int64_t*>(ptrs[6 * j + 2]);
    t.col = static_cast<int64_t*>(ptrs[6 * j + 3]);
    t.w = static_cast<float*>(ptrs[6 * j + 4]);
    t.result = static_cast<unsigned long long*>(ptrs[6 * j + 5]);
    t.k_cap = dims[6 * j];
    t.e_cap = dims[6 * j + 1];
    t.tag = static_cast<unsigned long long>(dims[6 * j + 2]);
    t.F = dims[6 * j + 3];
    t.w_words = static_cast<int>(dims[6 * j + 4]);
    t.x_words = static_cast<int>(dims[6 * j + 5]);
    TGP_REQUIRE(t.k_cap >= 0 && t.e_cap >= 0, TGP_ERR_INVALID, "tgp_gather_unpack_bucket_f32: bad step %d", j);
  }
  int64_t blocks = (max_words + 256 * 8 - 1) / (256 * 8);
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(gp_unpack_bucket_kernel,
                     dim3(static_cast<unsigned>(blocks), static_cast<unsigned>(world), static_cast<unsigned>(n)),
                     dim3(256), 0, stream, static_cast<const char*>(gathered), capacity, rank_stride, world, a);
  return check_launch("tgp_gather_unpack_bucket_f32");
}
