// Does the raw-buffer range check of gfx950 include the scalar offset?  A descriptor over the first 1024 bytes of a
// 1 MB allocation filled with 7.0f; loads at (voffset, soffset) pairs inside and outside the record range.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/buffer_range_soffset.hip -o /tmp/brs && /tmp/brs
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void probe(float* base, float* out) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 1024, 0x00020000);
  const int cases[6][2] = {{0, 0}, {1020, 0}, {1024, 0}, {0, 1024}, {512, 512}, {512, 508}};
  for (int c = 0; c < 6; ++c) {
    const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(r, cases[c][0], cases[c][1], 0);
    out[c] = __uint_as_float(v);
  }
}

int main() {
  float *d, *o;
  hipMalloc(&d, 1 << 20);
  hipMalloc(&o, 64);
  float* h = (float*)malloc(1 << 20);
  for (int i = 0; i < (1 << 18); ++i) h[i] = 7.0f;
  hipMemcpy(d, h, 1 << 20, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(1), 0, 0, d, o);
  float r[6];
  hipMemcpy(r, o, 24, hipMemcpyDeviceToHost);
  const char* names[6] = {"v=0 s=0 (in)", "v=1020 s=0 (in)", "v=1024 s=0 (out by voffset)", "v=0 s=1024 (out by soffset)",
                          "v=512 s=512 (sum out)", "v=512 s=508 (sum in)"};
  for (int c = 0; c < 6; ++c) printf("%-32s -> %g\n", names[c], r[c]);
  return 0;
}
