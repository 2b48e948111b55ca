// Back-to-back issue rate of v_mfma_f64_16x16x4_f64 on gfx950 (the guide's matrix-core table has no fp64 row):
// every wave of every CU runs a loop of independent MFMAs on register operands; prints TFLOP/s and cycles per MFMA.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f64_rate.hip -o tools/micro/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void loop(double* out, int iters, unsigned long long* cyc) {
  f64x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f64x4{0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-4;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
void run(int waves_per_simd) {
  const int blocks = 256 * waves_per_simd, iters = 20000;
  double* out; unsigned long long* cyc;
  hipMalloc(&out, blocks * 256 * sizeof(double));
  hipMalloc(&cyc, blocks * sizeof(unsigned long long));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  loop<NACC><<<blocks, 256>>>(out, 100, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  loop<NACC><<<blocks, 256>>>(out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
  const double flop = 2.0 * 16 * 16 * 4 * double(NACC) * iters * blocks * 4;
  printf("acc=%d waves/SIMD=%d: %.1f TFLOP/s, %.3f ms, %.1f shader cycles per MFMA per wave (wg 0)\n", NACC,
         waves_per_simd, flop / ms * 1e-9, ms, double(c0) / (double(iters) * NACC));
  hipFree(out); hipFree(cyc);
}

int main() {
  run<1>(1); run<4>(1); run<4>(2); run<8>(1);
  return 0;
}
