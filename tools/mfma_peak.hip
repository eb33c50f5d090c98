// Pure-MFMA throughput probe (no memory traffic): what the matrix pipes sustain on this box at its power-managed clock.
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int NACC>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(threadIdx.x * 0.002f - i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 256 * 8 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs : {256, 512}) {
    for (int rep = 0; rep < 3; ++rep) {
      const int iters = 20000;
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<8>, dim3(wgs), dim3(512), 0, 0, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flops = (double)wgs * 8 /*waves*/ * iters * 8 * 2.0 * 32 * 32 * 16;
      printf("wgs %d (8 waves each): %.2f ms  %.1f TFLOP/s\n", wgs, ms, flops / ms / 1e9);
    }
  }
  return 0;
}
