// Minimal probe for the r4 finding of profiles/r4_render_spill.md: the ray-marcher becomes irreproducible when hipcc's SLP vectoriser
// emits   v_mov_b32 v0, vZ ; v_pk_fma_f32 v[P:P+1], v[0:1], v[D:D+1], v[P:P+1] op_sel_hi:[0,1,1]
// (scalar z broadcast into both halves through op_sel_hi = 0 on src0, one instruction after the write of its low register).
// This kernel issues exactly that pair (inline asm, so the compiler cannot re-schedule it) next to the scalar form of the same
// arithmetic, with 1 - 8 waves per SIMD, and counts lanes whose two results differ.
//   hipcc --offload-arch=gfx950 -O3 tools/pk_opsel_repro.hip -o build/pk_opsel_repro && build/pk_opsel_repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { if ((x) != hipSuccess) { printf("hip error line %d\n", __LINE__); return 1; } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
template <int NOPS>
__global__ void probe(const float* z, const float* d, const float* o, float* out, unsigned long long* bad, int iters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float zz = z[i], d0 = d[2 * i], d1 = d[2 * i + 1], o0 = o[2 * i], o1 = o[2 * i + 1];
  unsigned long long nbad = 0;
  for (int it = 0; it < iters; ++it) {
    float junk = __int_as_float(0x7fc00000 + it);      // what the high half of the src0 pair holds (must be ignored)
    float p0, p1;
    const f2 dd = {d0, d1};
    if (NOPS == 0)
      asm volatile("v_mov_b32 v1, %6\n\tv_mov_b32 v2, %4\n\tv_mov_b32 v3, %5\n\tv_mov_b32 v0, %2\n\t"
                   "v_pk_fma_f32 v[2:3], v[0:1], %3, v[2:3] op_sel_hi:[0,1,1]\n\tv_mov_b32 %0, v2\n\tv_mov_b32 %1, v3"
                   : "=&v"(p0), "=&v"(p1) : "v"(zz), "v"(dd), "v"(o0), "v"(o1), "v"(junk) : "v0", "v1", "v2", "v3");
    else
      asm volatile("v_mov_b32 v1, %6\n\tv_mov_b32 v2, %4\n\tv_mov_b32 v3, %5\n\tv_mov_b32 v0, %2\n\ts_nop 4\n\t"
                   "v_pk_fma_f32 v[2:3], v[0:1], %3, v[2:3] op_sel_hi:[0,1,1]\n\tv_mov_b32 %0, v2\n\tv_mov_b32 %1, v3"
                   : "=&v"(p0), "=&v"(p1) : "v"(zz), "v"(dd), "v"(o0), "v"(o1), "v"(junk) : "v0", "v1", "v2", "v3");
    const float r0 = __builtin_fmaf(zz, d0, o0), r1 = __builtin_fmaf(zz, d1, o1);
    nbad += (__float_as_uint(p0) != __float_as_uint(r0)) + (__float_as_uint(p1) != __float_as_uint(r1));
    zz = zz * 1.0000001f + 1e-7f;
  }
  out[i] = zz;
  if (nbad) atomicAdd(bad, nbad);
}

int main() {
  const int N = 1 << 20;
  std::vector<float> hz(N), hd(2 * N), ho(2 * N);
  for (int i = 0; i < N; ++i) { hz[i] = 0.5f + (i % 977) * 1e-3f; hd[2 * i] = 0.3f + (i % 13) * 0.01f; hd[2 * i + 1] = -0.2f + (i % 7) * 0.02f; ho[2 * i] = 1.f + i % 5; ho[2 * i + 1] = -1.f - i % 3; }
  float *z, *d, *o, *out; unsigned long long* bad;
  hipMalloc(&z, N * 4); hipMalloc(&d, N * 8); hipMalloc(&o, N * 8); hipMalloc(&out, N * 4); hipMalloc(&bad, 8);
  hipMemcpy(z, hz.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(d, hd.data(), N * 8, hipMemcpyHostToDevice); hipMemcpy(o, ho.data(), N * 8, hipMemcpyHostToDevice);
  for (int nops = 0; nops < 2; ++nops)
    for (int bs : {64, 256, 1024}) {
      hipMemset(bad, 0, 8);
      if (nops == 0) hipLaunchKernelGGL(probe<0>, dim3(N / bs), dim3(bs), 0, 0, z, d, o, out, bad, 256);
      else hipLaunchKernelGGL(probe<1>, dim3(N / bs), dim3(bs), 0, 0, z, d, o, out, bad, 256);
      unsigned long long hb = 0; hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
      printf("%s block %4d: %llu mismatching results of %llu\n", nops ? "with s_nop 4 " : "back to back", bs, hb, 2ull * N * 256);
    }
  return 0;
}
