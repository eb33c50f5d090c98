// Stress probe for profiles/r4_render_spill.md: dependent chains that mix packed-fp32 and scalar VALU instructions back to back
// (inline asm with fixed registers: no compiler-inserted wait states), many waves per SIMD, results compared lane by lane with the
// same arithmetic from single instructions.  Prints mismatches per 16-lane quarter of the wave (the ray-marcher's irreproducible
// lanes are 48-63 = the last quarter pass of a wave64 VALU instruction).
//   hipcc --offload-arch=gfx950 -O3 tools/pk_hazard_probe.hip -o build/pk_hazard_probe && build/pk_hazard_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

// variant 0: v_div_fixup -> v_pk_mul (op_sel) -> v_pk_fma -> v_pk_add -> v_pk_add -> v_pk_add(neg) -> v_pk_mul -> v_add   (make_ray's chain)
// variant 1: the same chain with s_nop 7 between all instructions (reference for the asm itself)
template <int VAR>
__global__ void probe(const float* in, float* out, unsigned long long* bad, int iters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  float a = in[4 * i], b = in[4 * i + 1], c = in[4 * i + 2], d = in[4 * i + 3];
  unsigned long long nbad = 0;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    // reference with single instructions
    const float q = a / b;                                     // v_div_* sequence ending in v_div_fixup (compiler-generated)
    const float m0 = c * q, m1 = d * a;                        // (c * q, d * a)
    const float f0 = __builtin_fmaf(a, d, m0), f1 = __builtin_fmaf(q, c, m1);
    const float s0 = b + f0, s1 = a + f1;
    const float t0 = c + s0, t1 = d + s1;
    const float u0 = t0 - c, u1 = t1 - d;
    const float r = u0 * u0 + u1 * u1;
    float rr;
#define SETUP "v_mov_b32 v0, %1\n\tv_mov_b32 v2, %3\n\tv_mov_b32 v3, %4\n\tv_mov_b32 v4, %2\n\tv_mov_b32 v5, %1\n\tv_mov_b32 v8, %4\n\tv_mov_b32 v9, %3\n\ts_nop 7\n\t"
    if (VAR == 0)
      asm volatile(SETUP
          "v_div_fixup_f32 v1, %5, %2, %1\n\t"
          "v_pk_mul_f32 v[6:7], v[2:3], v[0:1] op_sel:[0,1] op_sel_hi:[1,0]\n\t"      // (c * q, d * a)
          "v_pk_fma_f32 v[6:7], v[0:1], v[8:9], v[6:7]\n\t"                            // (a * d + , q * c + )  with v8 = d, v9 = c
          "v_pk_add_f32 v[6:7], v[4:5], v[6:7]\n\t"                                    // (b + , a + )
          "v_pk_add_f32 v[6:7], v[2:3], v[6:7]\n\t"                                    // (c + , d + )
          "v_pk_add_f32 v[6:7], v[6:7], v[2:3] neg_lo:[0,1] neg_hi:[0,1]\n\t"
          "v_pk_mul_f32 v[6:7], v[6:7], v[6:7]\n\t"
          "v_add_f32 %0, v6, v7\n\t"
          : "=v"(rr) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(q) : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9");
    else
      asm volatile(SETUP
          "v_div_fixup_f32 v1, %5, %2, %1\n\ts_nop 7\n\t"
          "v_pk_mul_f32 v[6:7], v[2:3], v[0:1] op_sel:[0,1] op_sel_hi:[1,0]\n\ts_nop 7\n\t"
          "v_pk_fma_f32 v[6:7], v[0:1], v[8:9], v[6:7]\n\ts_nop 7\n\t"
          "v_pk_add_f32 v[6:7], v[4:5], v[6:7]\n\ts_nop 7\n\t"
          "v_pk_add_f32 v[6:7], v[2:3], v[6:7]\n\ts_nop 7\n\t"
          "v_pk_add_f32 v[6:7], v[6:7], v[2:3] neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 7\n\t"
          "v_pk_mul_f32 v[6:7], v[6:7], v[6:7]\n\ts_nop 7\n\t"
          "v_add_f32 %0, v6, v7\n\t"
          : "=v"(rr) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(q) : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9");
    if (__float_as_uint(rr) != __float_as_uint(r)) nbad += 1;
    acc += rr;
    a = a * 1.0000002f + 1e-7f; c = c * 0.9999999f - 1e-7f;
  }
  out[i] = acc;
  if (nbad) atomicAdd(bad + (lane >> 4), nbad);
}

int main() {
  const int N = 1 << 20;
  std::vector<float> h(4 * N);
  for (int i = 0; i < N; ++i) { h[4 * i] = 0.5f + (i % 977) * 1e-3f; h[4 * i + 1] = 1.3f + (i % 13) * 0.01f; h[4 * i + 2] = -0.2f + (i % 7) * 0.02f; h[4 * i + 3] = 1.f + i % 5; }
  float *in, *out; unsigned long long* bad;
  if (hipMalloc(&in, N * 16) != hipSuccess || hipMalloc(&out, N * 4) != hipSuccess || hipMalloc(&bad, 32) != hipSuccess) return 1;
  (void)hipMemcpy(in, h.data(), N * 16, hipMemcpyHostToDevice);
  for (int var = 0; var < 2; ++var)
    for (int bs : {64, 256, 1024}) {
      (void)hipMemset(bad, 0, 32);
      if (var == 0) hipLaunchKernelGGL(probe<0>, dim3(N / bs), dim3(bs), 0, 0, in, out, bad, 128);
      else hipLaunchKernelGGL(probe<1>, dim3(N / bs), dim3(bs), 0, 0, in, out, bad, 128);
      unsigned long long hb[4]; (void)hipMemcpy(hb, bad, 32, hipMemcpyDeviceToHost);
      printf("%s block %4d: mismatches per lane quarter [0-15] %llu [16-31] %llu [32-47] %llu [48-63] %llu of %d\n", var ? "s_nop 7 between " : "back to back     ", bs,
             hb[0], hb[1], hb[2], hb[3], N / 4 * 128);
    }
  return 0;
}
