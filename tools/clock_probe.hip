// What does s_memtime count on this part, and what clock do the kernels actually get?  (GPU box)
//   hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o build/clock_probe && build/clock_probe
// Each mode runs one kernel for a fixed number of loop iterations on every CU; ticks of s_memtime across the kernel (wave 0 of
// block 0) are divided by the HIP-event wall time: a constant ratio over idle / VALU / MFMA / MFMA + LDS loads means a fixed-rate
// counter, a ratio that drops under load is the shader clock being pulled down by the power limit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>
__global__ __launch_bounds__(512) void probe(uint64_t* out, float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  union { uint32_t u[4]; bf16x8 v; } a, b;
  for (int j = 0; j < 4; ++j) { a.u[j] = 0x3c003c00u + threadIdx.x; b.u[j] = 0x3c003c00u + blockIdx.x; }
  float f = threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) { __builtin_amdgcn_s_sleep(8); }
    if constexpr (MODE == 1) { _Pragma("unroll") for (int k = 0; k < 32; ++k) f = __builtin_fmaf(f, 1.0001f, 0.5f); }
    if constexpr (MODE >= 2) {
      _Pragma("unroll") for (int k = 0; k < 4; ++k) {
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc[k], 0, 0, 0);
        if constexpr (MODE == 3) {                       // plus one ds_read_b128 per MFMA (the GEMM's ratio)
          const bf16x8 t = *reinterpret_cast<const bf16x8*>(lds + ((threadIdx.x * 16 + k * 8192 + it * 64) & 65520));
          asm volatile("" :: "v"(t));
        }
      }
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  float s = f;
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  if (s == 12345.678f) sink[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = t1 - t0; }
}

template <int MODE>
static void run(const char* name, int grid, int iters, uint64_t* dout, float* sink) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(512), 0, 0, dout, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint64_t t; hipMemcpy(&t, dout, 8, hipMemcpyDeviceToHost);
    double tf = MODE >= 2 ? 2.0 * 32 * 32 * 16 * 4.0 * iters * 8 * grid / (ms * 1e-3) / 1e12 : 0.0;
    printf("%-34s grid %4d: %9.1f us wall, %12llu ticks -> %7.1f ticks/us%s", name, grid, ms * 1e3, (unsigned long long)t, t / (ms * 1e3), MODE >= 2 ? "" : "\n");
    if (MODE >= 2) printf("   %7.1f TFLOP/s\n", tf);
  }
}

int main() {
  uint64_t* dout; float* sink; hipMalloc(&dout, 64); hipMalloc(&sink, 64);
  run<0>("s_sleep, one workgroup", 1, 20000, dout, sink);
  run<1>("dependent v_fma, one workgroup", 1, 20000, dout, sink);
  run<1>("dependent v_fma, every CU", 256, 20000, dout, sink);
  run<2>("MFMA 32x32x16 bf16, one workgroup", 1, 40000, dout, sink);
  run<2>("MFMA 32x32x16 bf16, every CU", 256, 40000, dout, sink);
  run<3>("MFMA + ds_read_b128, every CU", 256, 40000, dout, sink);
  return 0;
}
