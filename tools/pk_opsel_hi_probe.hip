// r6 probe for profiles/r6_render_opsel.md: v_pk_fma_f32 with a source broadcast from the HIGH half of a register pair (op_sel:[0,1,0]) against
// the same arithmetic with the value copied to an even register first (op_sel_hi:[1,0,1]) and against scalar v_fma_f32, next to the instruction
// kinds the ray-marcher has around it (LDS b128 reads producing the pair, transcendentals, MFMA), at several waves per SIMD and at one.
//   hipcc --offload-arch=gfx950 -O3 tools/pk_opsel_hi_probe.hip -o /tmp/pk_opsel_hi_probe && /tmp/pk_opsel_hi_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { if ((x) != hipSuccess) { printf("hip error line %d\n", __LINE__); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MIX>
__global__ __launch_bounds__(256) void probe(const float* in, unsigned long long* bad, float* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int gi = blockIdx.x * blockDim.x + tid;
  for (int i = tid; i < 256 * 4; i += blockDim.x) lds[i] = in[(gi * 7 + i) & 0xfffff] * 0.5f + 0.25f;
  __syncthreads();
  f2 t = {in[gi & 0xfffff], in[(gi + 1) & 0xfffff]}, acc = {0.f, 0.f}, accr = {0.f, 0.f}, acce = {0.f, 0.f};
  unsigned long long nb_hi = 0, nb_ev = 0;
  f32x16 macc;
  for (int r = 0; r < 16; ++r) macc[r] = 0.f;
  float tr = 0.3f;
  for (int it = 0; it < iters; ++it) {
    // the pair comes out of a 16-byte LDS read: (w.x, w.y) = an even / odd register pair
    const f4 w = *reinterpret_cast<const f4*>(lds + ((tid + it) & 255) * 4);
    f2 wp = {w.x, w.y};
    f2 d_hi, d_ev;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(d_hi) : "v"(t), "v"(wp), "v"(acc));     // broadcast of wp.y from the HIGH half
    float we;
    asm volatile("v_mov_b32 %0, %1" : "=v"(we) : "v"(w.y));
    f2 wq = {we, we};
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d_ev) : "v"(t), "v"(wq), "v"(acc)); // the same from the LOW half of another pair
    if (MIX & 8) {
      // the exact shape of the failing gather code: two packed multiplies by the pair's LOW half, then two packed FMAs by its HIGH half, the
      // other sources fresh from memory
      const f4 ta = *reinterpret_cast<const f4*>(in + (((gi + it) * 4) & 0xffffc));
      const f4 tb = *reinterpret_cast<const f4*>(in + (((gi + it) * 4 + 64) & 0xffffc));
      f2 a0 = {ta.x, ta.y}, a1 = {ta.z, ta.w}, b0 = {tb.x, tb.y}, b1 = {tb.z, tb.w}, p0, p1;
      asm volatile("v_pk_mul_f32 %0, %2, %6 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %1, %3, %6 op_sel_hi:[1,0]\n\t"
                   "v_pk_fma_f32 %0, %4, %6, %0 op_sel:[0,1,0]\n\tv_pk_fma_f32 %1, %5, %6, %1 op_sel:[0,1,0]"
                   : "=&v"(p0), "=&v"(p1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(wp));
      const float q0 = __builtin_fmaf(tb.x, w.y, ta.x * w.x), q1 = __builtin_fmaf(tb.y, w.y, ta.y * w.x);
      const float q2 = __builtin_fmaf(tb.z, w.y, ta.z * w.x), q3 = __builtin_fmaf(tb.w, w.y, ta.w * w.x);
      nb_hi += (__float_as_uint(p0.x) != __float_as_uint(q0)) + (__float_as_uint(p0.y) != __float_as_uint(q1)) +
               (__float_as_uint(p1.x) != __float_as_uint(q2)) + (__float_as_uint(p1.y) != __float_as_uint(q3));
    }
    const float r0 = __builtin_fmaf(t.x, w.y, acc.x), r1 = __builtin_fmaf(t.y, w.y, acc.y);
    nb_hi += (__float_as_uint(d_hi.x) != __float_as_uint(r0)) + (__float_as_uint(d_hi.y) != __float_as_uint(r1));
    nb_ev += (__float_as_uint(d_ev.x) != __float_as_uint(r0)) + (__float_as_uint(d_ev.y) != __float_as_uint(r1));
    acc.x = r0 * 0.5f; acc.y = r1 * 0.5f;
    t.x = t.x * 0.999f + w.z * 0.001f; t.y = t.y * 0.998f + w.w * 0.002f;
    if (MIX & 1) { tr = __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(tr)) * 0.7f; }
    if (MIX & 2) {
      union { float f[4]; bf16x8 v; } a, b;
      a.f[0] = t.x; a.f[1] = t.y; a.f[2] = w.x; a.f[3] = w.z; b.f[0] = w.y; b.f[1] = w.w; b.f[2] = t.y; b.f[3] = t.x;
      macc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, macc, 0, 0, 0);
    }
    if (MIX & 4) { lds[1024 + ((tid * 5 + it) & 1023)] = t.x; }
  }
  float s = acc.x + acc.y + tr;
  for (int r = 0; r < 16; ++r) s += macc[r] * 1e-30f;
  sink[gi] = s;
  if (nb_hi) atomicAdd(&bad[lane >> 4], nb_hi);
  if (nb_ev) atomicAdd(&bad[4 + (lane >> 4)], nb_ev);
}

template <int MIX>
static int run(const float* din, unsigned long long* dbad, float* dsink, int lds_bytes, const char* what) {
  CK(hipMemset(dbad, 0, 64));
  CK(hipFuncSetAttribute((const void*)probe<MIX>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
  hipLaunchKernelGGL(probe<MIX>, dim3(256 * 16), dim3(256), lds_bytes, 0, din, dbad, dsink, 4000);
  CK(hipDeviceSynchronize());
  unsigned long long h[8];
  CK(hipMemcpy(h, dbad, 64, hipMemcpyDeviceToHost));
  printf("mix %d  %-34s op_sel:[0,1,0] mismatches by 16-lane pass: %llu %llu %llu %llu | even-register form: %llu %llu %llu %llu\n", MIX, what,
         h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
  return 0;
}

// Second form: the checker waves (the failing gather sequence, sources fresh from LDS / memory) share their SIMDs with waves that run OTHER
// instruction streams out of phase - an MFMA chain, transcendentals + scalar-rate VALU, LDS + global traffic - as in the ray-marcher, where the
// waves of a SIMD are at different points of the ray loop.
__global__ __launch_bounds__(256) void probe_roles(const float* in, unsigned long long* bad, float* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int gi = blockIdx.x * blockDim.x + tid;
  const int role = blockIdx.x & 3;
  for (int i = tid; i < 256 * 4; i += blockDim.x) lds[i] = in[(gi * 7 + i) & 0xfffff] * 0.5f + 0.25f;
  __syncthreads();
  float acc = 0.f;
  if (role == 0) {
    unsigned long long nb = 0;
    for (int it = 0; it < iters; ++it) {
      const f4 w = *reinterpret_cast<const f4*>(lds + ((tid + it) & 255) * 4);
      f2 wp = {w.x, w.y}, wq = {w.z, w.w};
      const f4 ta = *reinterpret_cast<const f4*>(in + (((gi + it * 977) * 4) & 0xffffc));
      const f4 tb = *reinterpret_cast<const f4*>(in + (((gi + it * 977) * 4 + 64) & 0xffffc));
      const f4 tc = *reinterpret_cast<const f4*>(in + (((gi + it * 977) * 4 + 128) & 0xffffc));
      f2 a0 = {ta.x, ta.y}, a1 = {ta.z, ta.w}, b0 = {tb.x, tb.y}, b1 = {tb.z, tb.w}, c0 = {tc.x, tc.y}, c1 = {tc.z, tc.w}, p0, p1;
      asm volatile("v_pk_mul_f32 %0, %2, %8 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %1, %3, %8 op_sel_hi:[1,0]\n\t"
                   "v_pk_fma_f32 %0, %4, %8, %0 op_sel:[0,1,0]\n\tv_pk_fma_f32 %1, %5, %8, %1 op_sel:[0,1,0]\n\t"
                   "v_pk_fma_f32 %0, %6, %9, %0 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %1, %7, %9, %1 op_sel_hi:[1,0,1]"
                   : "=&v"(p0), "=&v"(p1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1), "v"(wp), "v"(wq));
      const float q0 = __builtin_fmaf(tc.x, w.z, __builtin_fmaf(tb.x, w.y, ta.x * w.x)), q1 = __builtin_fmaf(tc.y, w.z, __builtin_fmaf(tb.y, w.y, ta.y * w.x));
      const float q2 = __builtin_fmaf(tc.z, w.z, __builtin_fmaf(tb.z, w.y, ta.z * w.x)), q3 = __builtin_fmaf(tc.w, w.z, __builtin_fmaf(tb.w, w.y, ta.w * w.x));
      nb += (__float_as_uint(p0.x) != __float_as_uint(q0)) + (__float_as_uint(p0.y) != __float_as_uint(q1)) +
            (__float_as_uint(p1.x) != __float_as_uint(q2)) + (__float_as_uint(p1.y) != __float_as_uint(q3));
      acc += p0.x + p1.y;
    }
    if (nb) atomicAdd(&bad[lane >> 4], nb);
  } else if (role == 1) {
    f32x16 m;
    for (int r = 0; r < 16; ++r) m[r] = 0.f;
    union { float f[4]; bf16x8 v; } a, b;
    a.f[0] = in[gi & 0xfffff]; a.f[1] = 1.f; a.f[2] = 2.f; a.f[3] = 3.f; b.f[0] = 0.5f; b.f[1] = 0.25f; b.f[2] = in[(gi + 9) & 0xfffff]; b.f[3] = 1.f;
    for (int it = 0; it < iters * 2; ++it) { m = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, m, 0, 0, 0); if ((it & 7) == 7) { a.f[0] = m[3] * 1e-20f; } }
    for (int r = 0; r < 16; ++r) acc += m[r] * 1e-30f;
  } else if (role == 2) {
    float t = in[gi & 0xfffff] * 0.5f + 1.0f;
    for (int it = 0; it < iters * 4; ++it) { t = __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(t)) * 0.7f + 0.3f; t = t * 0.999f + (float)(it & 3) * 1e-3f; }
    acc = t;
  } else {
    float t = 0.f;
    for (int it = 0; it < iters * 2; ++it) {
      const f4 w = *reinterpret_cast<const f4*>(lds + ((tid * 3 + it) & 255) * 4);
      const f4 g = *reinterpret_cast<const f4*>(in + (((gi * 5 + it * 131) * 4) & 0xffffc));
      t += w.x * g.y + w.w * g.z;
      lds[1024 + ((tid * 5 + it) & 1023)] = t;
    }
    acc = t;
  }
  sink[gi] = acc;
}

static int run_roles(const float* din, unsigned long long* dbad, float* dsink, int lds_bytes, const char* what) {
  CK(hipMemset(dbad, 0, 64));
  CK(hipFuncSetAttribute((const void*)probe_roles, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
  hipLaunchKernelGGL(probe_roles, dim3(256 * 16), dim3(256), lds_bytes, 0, din, dbad, dsink, 4000);
  CK(hipDeviceSynchronize());
  unsigned long long h[8];
  CK(hipMemcpy(h, dbad, 64, hipMemcpyDeviceToHost));
  printf("roles  %-34s gather sequence mismatches by 16-lane pass: %llu %llu %llu %llu\n", what, h[0], h[1], h[2], h[3]);
  return 0;
}

int main() {
  const int N = 1 << 20;
  std::vector<float> h(N);
  unsigned s = 12345u;
  for (int i = 0; i < N; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)(s >> 8) / 16777216.0f * 2.0f - 1.0f; }
  float* din; unsigned long long* dbad; float* dsink;
  CK(hipMalloc(&din, N * 4)); CK(hipMalloc(&dbad, 64)); CK(hipMalloc(&dsink, 256 * 16 * 256 * 4));
  CK(hipMemcpy(din, h.data(), N * 4, hipMemcpyHostToDevice));
  for (int rep = 0; rep < 2; ++rep) {
    if (run<0>(din, dbad, dsink, 16384, "several waves per SIMD")) return 1;
    if (run<1>(din, dbad, dsink, 16384, "several waves per SIMD")) return 1;
    if (run<3>(din, dbad, dsink, 16384, "several waves per SIMD")) return 1;
    if (run<7>(din, dbad, dsink, 16384, "several waves per SIMD")) return 1;
    if (run<7>(din, dbad, dsink, 100 * 1024, "ONE wave per SIMD (100 KB LDS)")) return 1;
    if (run<8>(din, dbad, dsink, 16384, "gather sequence, several waves")) return 1;
    if (run<15>(din, dbad, dsink, 16384, "gather sequence + all, several waves")) return 1;
    if (run<15>(din, dbad, dsink, 42 * 1024, "gather sequence + all, 3 per SIMD")) return 1;
    if (run_roles(din, dbad, dsink, 16384, "mixed roles, several waves per SIMD")) return 1;
    if (run_roles(din, dbad, dsink, 42 * 1024, "mixed roles, 3 workgroups per CU")) return 1;
  }
  return 0;
}
