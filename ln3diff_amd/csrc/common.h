// Shared device helpers for the ln3diff_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define LN3D_OK 0
#define LN3D_ERR_BAD_ARG (-1)
#define LN3D_ERR_LAUNCH (-2)
#define LN3D_ERR_UNSUPPORTED (-3)

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// round-to-nearest-even, NaN not special-cased (inputs on this path are finite)
__device__ __forceinline__ bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

// two f32 -> packed bf16x2 (RNE) through the native conversion: hipcc selects v_cvt_pk_bf16_f32 on gfx950
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  bf16x2_t v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Sum over the 64 lanes on the DPP cross-lane path (row_shr 1, 2, 4, 8 inside rows of 16, then row_bcast:15 / :31 - LLVM's
// wave64 gfx9 scan), the total broadcast from lane 63 through an SGPR: 6 VALU-rate instructions instead of 6 ds_bpermute round
// trips through the LDS.  Summation order differs from wave_sum's butterfly (last-ulp differences).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add_step(float x) {
  return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_dpp(float x) {
  x = dpp_add_step<0x111, 0xf>(x); x = dpp_add_step<0x112, 0xf>(x); x = dpp_add_step<0x114, 0xf>(x); x = dpp_add_step<0x118, 0xf>(x);
  x = dpp_add_step<0x142, 0xa>(x); x = dpp_add_step<0x143, 0xc>(x);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

// Two elements at a time (v_pk_mul_f32 / v_pk_fma_f32: one instruction per pair).  r4: erf(x / sqrt 2) = u P(u^2) with u = x clamped
// to [-c, c] and P a degree-7 minimax polynomial of erf on [0, 4]: no transcendental is left (12 packed + 2 scalar instructions per
// pair instead of 15 + 2 v_rcp + 6 - the GEMM's GELU epilogue is VALU-throughput bound, 19 us of a 108 us fc1 launch at the
// throttled clock, profiles/r4_gemm.md).  Accuracy, measured against fp64 erf over [-10, 10] (r5): |gelu error| <=
// 1.3e-4 ABSOLUTE everywhere, which is inside the bf16 rounding of the output only for |gelu(x)| >~ 0.03; in the negative tail the
// RELATIVE error is large (2.3 % at x = -3, ~34 bf16 ulps at x = -3.5, and below x ~ -3.9 the result is ~ -1e-5 where the exact
// value decays to 0) - r3's Abramowitz-Stegun 7.1.28 form was 3e-7.  The clamp bound c = 3.9985 (not 4) keeps |u P(u^2)| <= 1 - 2^-22
// in fp32, so the result always has the sign of x (at c = 4 the polynomial overshoots 1 by 2e-6 and x <= -3.9987 came out +2e-6).
typedef float ln3d_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_erf2(float& x0, float& x1) {
  const ln3d_f32x2 x = {x0, x1};
  const ln3d_f32x2 u = {__builtin_amdgcn_fmed3f(x0, -3.9985f, 3.9985f), __builtin_amdgcn_fmed3f(x1, -3.9985f, 3.9985f)};
  const ln3d_f32x2 t = u * u;
  ln3d_f32x2 p = {-2.556055811e-09f, -2.556055811e-09f};
  p = __builtin_elementwise_fma(p, t, ln3d_f32x2{2.088922457e-07f, 2.088922457e-07f});
  p = __builtin_elementwise_fma(p, t, ln3d_f32x2{-7.419432677e-06f, -7.419432677e-06f});
  p = __builtin_elementwise_fma(p, t, ln3d_f32x2{1.523832179e-04f, 1.523832179e-04f});
  p = __builtin_elementwise_fma(p, t, ln3d_f32x2{-2.042111475e-03f, -2.042111475e-03f});
  p = __builtin_elementwise_fma(p, t, ln3d_f32x2{1.916284487e-02f, 1.916284487e-02f});
  p = __builtin_elementwise_fma(p, t, ln3d_f32x2{-1.321282834e-01f, -1.321282834e-01f});
  p = __builtin_elementwise_fma(p, t, ln3d_f32x2{7.976111174e-01f, 7.976111174e-01f});
  const ln3d_f32x2 e = u * p;                   // erf(x / sqrt 2); |e| <= 1 - 2^-22 beyond the clamp
  const ln3d_f32x2 hx = x * 0.5f;
  const ln3d_f32x2 r = __builtin_elementwise_fma(hx, e, hx);
  x0 = r.x; x1 = r.y;
}
__device__ __forceinline__ float gelu_tanh(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  return 0.5f * x * (1.0f + tanhf(k0 * (x + k1 * x * x * x)));
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }
// torch.nn.functional.softplus(beta=1, threshold=20)
__device__ __forceinline__ float softplus20(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

// LDS-DMA issued as inline asm, not through __builtin_amdgcn_global_load_lds.  r4 finding: hipcc's wait-count pass files the
// builtin as a FLAT access that touches both VMEM and LDS ("pending flat"), and from the first one on it answers every later
// LDS dependency with s_waitcnt lgkmcnt(0) instead of a counted wait - the K loop then drains the fragment read it issued one
// MFMA ago at the head of every K substep, with both waves of a SIMD in the same phase (profiles/r4_gemm.md; the attention kernels had 137 lgkmcnt(0) waits and not one counted wait).  The asm
// form is invisible to that pass: fragment reads get counted lgkmcnt(N) waits, and the DMA's own completion is waited for by
// hand (counted s_waitcnt vmcnt) as before.  M0 = LDS byte address of the wave's 1 KB piece; one wait state between the M0
// write and the DMA (LDS-DMA reads M0).
// "m0" is on the clobber lists: the statements overwrite it.  hipcc files M0 as a reserved register and warns that a clobber
// of it is not preserved (-Winline-asm, once per inlined copy); it initialises M0 itself in front of each of its own M0 users, so
// the clobber documents the write rather than changing code - the diagnostic is silenced for these two statements only.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void lds_dma16_s(const void* sbase, uint32_t voff, uint32_t lds) {      // wave-uniform base + 32-bit lane offset
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ void lds_dma16_v(const void* vaddr, uint32_t lds) {                     // per-lane 64-bit address
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(vaddr), "s"(lds) : "memory", "m0");
}
#pragma clang diagnostic pop
// LDS byte address of a pointer into the dynamic shared segment
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(lds_void_t*)p; }

// hipFuncSetAttribute applies to the CURRENT device: remembered per (kernel instantiation, device) so that a process driving several
// GPUs sets it on each of them (one process per GPU is the deployment, but the library must not depend on it)
struct AttrOnce {
  std::atomic<unsigned long long> done{0};
  bool need() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return true;      // unknown / high device id: set the attribute every time
    const unsigned long long bit = 1ull << d;
    return (done.fetch_or(bit) & bit) == 0;                                    // two host threads racing here: at worst both set it
  }
};

// compute units a launch on this stream can occupy (csrc/runtime.hip): the device's, or the CU mask's of a lane stream
int ln3d_stream_cus(hipStream_t s);

static inline int ln3d_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? LN3D_OK : LN3D_ERR_LAUNCH;
}
