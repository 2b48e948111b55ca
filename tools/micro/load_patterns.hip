// Micro-benchmark for the C3 small-graph kernel's load phase: how fast can a CU pull one graph's operands
// (A [60,60], S [60,20], X [60,32] fp32, 26.9 KB, contiguous per graph) with different instruction shapes and
// wave counts?  Each variant only loads and folds the values into one float per lane (stored, so nothing is elided).
//   build: hipcc -O3 --offload-arch=gfx950 tools/micro/load_patterns.hip -o gpurun_out/load_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

constexpr int N = 60, K = 20, F = 32;
constexpr int GA = N * N, GS = N * K, GX = N * F;  // floats per graph

__device__ __forceinline__ int rho(int r) { return (r & 3) + 8 * (r >> 2); }

// V0: the kernel's own shapes: A as 16 float4 row loads (16 lanes per row), S and X as 32 dword loads each
__global__ void v0(const float* A, const float* S, const float* X, float* out, int B, int waves) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.x * waves + w;
  if (b >= B) return;
  const int lm = lane & 31, lk = lane >> 5, q = lane & 15;
  const float* Ab = A + (long)b * GA; const float* Sb = S + (long)b * GS; const float* Xb = X + (long)b * GX;
  float acc = 0.f;
  float4 v[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int i = (lane >> 4) + 4 * t;
    const bool ok = 4 * q + 3 < N && i < N;
    v[t] = *reinterpret_cast<const float4*>(Ab + (ok ? i * N + 4 * q : 0));
  }
  float sr[32], xr[32];
#pragma unroll
  for (int qq = 0; qq < 32; ++qq) {
    const int node = 32 * (qq >> 4) + rho(qq & 15) + 4 * lk;
    const bool ok = lm < K && node < N;
    sr[qq] = Sb[ok ? node * K + lm : 0];
  }
#pragma unroll
  for (int qq = 0; qq < 32; ++qq) {
    const int node = 32 * (qq >> 4) + rho(qq & 15) + 4 * lk;
    const bool ok = lm < F && node < N;
    xr[qq] = Xb[ok ? node * F + lm : 0];
  }
#pragma unroll
  for (int t = 0; t < 16; ++t) acc += v[t].x + v[t].y + v[t].z + v[t].w;
#pragma unroll
  for (int qq = 0; qq < 32; ++qq) acc += sr[qq] + xr[qq];
  out[(long)b * 64 + lane] = acc;
}

// V1: everything as lane-linear float4 loads (A 15, S 5, X 8 instructions per graph, one wave per graph)
template <int SPLIT>  // SPLIT waves share a graph
__global__ void v1(const float* A, const float* S, const float* X, float* out, int B, int waves) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int unit = blockIdx.x * waves + w;
  const int b = unit / SPLIT, part = unit % SPLIT;
  if (b >= B) return;
  const float4* Ab = reinterpret_cast<const float4*>(A + (long)b * GA);
  const float4* Sb = reinterpret_cast<const float4*>(S + (long)b * GS);
  const float4* Xb = reinterpret_cast<const float4*>(X + (long)b * GX);
  float4 v[32];
  int n = 0;
#pragma unroll
  for (int t = 0; t < 15; ++t) { const int i = (t * SPLIT + part) * 64 + lane; if (t * SPLIT + part < 15) { v[n++] = Ab[i < GA / 4 ? i : 0]; } }
#pragma unroll
  for (int t = 0; t < 5; ++t) { const int i = (t * SPLIT + part) * 64 + lane; if (t * SPLIT + part < 5) { v[n++] = Sb[i < GS / 4 ? i : 0]; } }
#pragma unroll
  for (int t = 0; t < 8; ++t) { const int i = (t * SPLIT + part) * 64 + lane; if (t * SPLIT + part < 8) { v[n++] = Xb[i < GX / 4 ? i : 0]; } }
  float acc = 0.f;
#pragma unroll
  for (int t = 0; t < 32; ++t) if (t < n) acc += v[t].x + v[t].y + v[t].z + v[t].w;
  out[(long)unit * 64 + lane] = acc;
}


// V2: v0's loads + an MFMA phase of the real kernel's length (128 x v_mfma_f32_32x32x2_f32 on the loaded values),
// 8 waves per workgroup; STAGGER: waves 4-7 start loading when waves 0-3 have ISSUED (EARLY = after the first
// quarter of) their loads.
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int STAGGER>
__global__ __launch_bounds__(512, 2) void v2(const float* A, const float* S, const float* X, float* out, int B) {
  __shared__ int s_issued;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x * 8 + w;
  if (threadIdx.x == 0) s_issued = 0;
  __syncthreads();
  if (b >= B) return;
  if (STAGGER && w >= 4) {
    while (__hip_atomic_load(&s_issued, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) __builtin_amdgcn_s_sleep(2);
  }
  const int lm = lane & 31, lk = lane >> 5, q = lane & 15;
  const float* Ab = A + (long)b * GA; const float* Sb = S + (long)b * GS; const float* Xb = X + (long)b * GX;
  float4 v[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int i = (lane >> 4) + 4 * t;
    const bool ok = 4 * q + 3 < N && i < N;
    v[t] = *reinterpret_cast<const float4*>(Ab + (ok ? i * N + 4 * q : 0));
    if (STAGGER == 2 && t == 3 && w < 4 && lane == 0) atomicAdd(&s_issued, 1);
  }
  float sr[32], xr[32];
#pragma unroll
  for (int qq = 0; qq < 32; ++qq) {
    const int node = 32 * (qq >> 4) + rho(qq & 15) + 4 * lk;
    sr[qq] = Sb[(lm < K && node < N) ? node * K + lm : 0];
  }
#pragma unroll
  for (int qq = 0; qq < 32; ++qq) {
    const int node = 32 * (qq >> 4) + rho(qq & 15) + 4 * lk;
    xr[qq] = Xb[(lm < F && node < N) ? node * F + lm : 0];
  }
  if (STAGGER == 1 && w < 4 && lane == 0) atomicAdd(&s_issued, 1);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int rep = 0; rep < 2; ++rep) {
#pragma unroll
    for (int qq = 0; qq < 32; ++qq) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sr[qq], xr[qq], acc, 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[t].x, v[t].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[t].z, v[t].w, acc, 0, 0, 0);
    }
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += acc[i];
  out[(long)b * 64 + lane] = r;
}

int main() {
  const int B = 2048;
  float *A, *S, *X, *out;
  hipMalloc(&A, (size_t)B * GA * 4); hipMalloc(&S, (size_t)B * GS * 4); hipMalloc(&X, (size_t)B * GX * 4);
  hipMalloc(&out, (size_t)B * 4 * 64 * 4);
  hipMemset(A, 0, (size_t)B * GA * 4); hipMemset(S, 0, (size_t)B * GS * 4); hipMemset(X, 0, (size_t)B * GX * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 5; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 200;
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, mb = B * (GA + GS + GX) * 4.0 / 1e6;
    printf("%-52s %7.2f us  %6.2f TB/s\n", name, us, mb / us / 1e6 * 1e6 / 1e6);
  };
  for (int waves : {4, 8, 16}) {
    char nm[128];
    snprintf(nm, sizeof nm, "v0 kernel-shaped loads, %2d waves/WG, all graphs", waves);
    timeit(nm, [&] { hipLaunchKernelGGL(v0, dim3((B + waves - 1) / waves), dim3(64 * waves), 0, 0, A, S, X, out, B, waves); });
    snprintf(nm, sizeof nm, "v1 lane-linear float4, %2d waves/WG, wave per graph", waves);
    timeit(nm, [&] { hipLaunchKernelGGL(v1<1>, dim3((B + waves - 1) / waves), dim3(64 * waves), 0, 0, A, S, X, out, B, waves); });
  }
  timeit("v1 float4, 2 waves per graph, 16 waves/WG", [&] { hipLaunchKernelGGL(v1<2>, dim3((2 * B + 15) / 16), dim3(1024), 0, 0, A, S, X, out, B, 16); });
  timeit("v1 float4, 4 waves per graph, 16 waves/WG", [&] { hipLaunchKernelGGL(v1<4>, dim3((4 * B + 15) / 16), dim3(1024), 0, 0, A, S, X, out, B, 16); });
  timeit("v2 loads + 128 MFMAs, 8 waves/WG, no stagger", [&] { hipLaunchKernelGGL(v2<0>, dim3(B / 8), dim3(512), 0, 0, A, S, X, out, B); });
  timeit("v2 loads + 128 MFMAs, stagger after all issued", [&] { hipLaunchKernelGGL(v2<1>, dim3(B / 8), dim3(512), 0, 0, A, S, X, out, B); });
  timeit("v2 loads + 128 MFMAs, stagger after 4 of 80 loads", [&] { hipLaunchKernelGGL(v2<2>, dim3(B / 8), dim3(512), 0, 0, A, S, X, out, B); });
  // half the graphs only (4 waves per CU active): per-wave rate
  {
    const int Bh = 1024;
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(v0, dim3(Bh / 4), dim3(256), 0, 0, A, S, X, out, Bh, 4);
    hipDeviceSynchronize(); hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(v0, dim3(Bh / 4), dim3(256), 0, 0, A, S, X, out, Bh, 4);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s %7.2f us\n", "v0, 1024 graphs only (4 waves per CU)", ms * 1e3 / 200);
    hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(v1<1>, dim3(Bh / 4), dim3(256), 0, 0, A, S, X, out, Bh, 4);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s %7.2f us\n", "v1, 1024 graphs only (4 waves per CU)", ms * 1e3 / 200);
  }
  return 0;
}
