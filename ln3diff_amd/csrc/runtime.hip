// Device queries of the C ABI (include/ln3d.h).  r5's CU-masked stream registry (ln3d_stream_create_cu_mask, ABI 9) served an
// experiment that measured neutral (profiles/r5_lanes.md) and is gone with ABI 10: every launch sizes its tiling for the whole device.
#include "common.h"
#include "../../include/ln3d.h"

static int device_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    n = v;
  }
  return n;
}

// compute units a launch on `s` can occupy (kept as a function of the stream: the launchers' tile selection calls it)
int ln3d_stream_cus(hipStream_t) { return device_cus(); }

extern "C" int ln3d_device_cus(void) { return device_cus(); }

// ------------------------------------------------------------------ diagnostic: what the matrix pipes sustain on THIS box (bench.py's roofline)
// `iters` x 8 independent v_mfma_f32_32x32x16_bf16 per wave, no memory traffic: 8 waves per workgroup, `wgs` workgroups.  out: wgs * 512 floats.
// The rate is the part's power-managed one for a pure-MFMA stream on fixed non-zero operands (profiles/r6_power.md), not the datasheet's.
__global__ __launch_bounds__(512) void mfma_probe_kernel(float* out, int iters) {
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(threadIdx.x * 0.002f - i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
extern "C" int ln3d_probe_mfma_bf16(float* out, int wgs, int iters, void* stream) {
  if (!out || wgs <= 0 || iters <= 0) return LN3D_ERR_BAD_ARG;
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(wgs), dim3(512), 0, (hipStream_t)stream, out, iters);
  return ln3d_check_launch();
}
