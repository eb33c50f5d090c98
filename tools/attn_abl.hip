// Stand-alone ablation harness for the attention kernel (GPU box):
//   for a in 0 1 2 3; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -DLN3D_ATTN_ABL=$a tools/attn_abl.hip -o /tmp/attn_abl$a && /tmp/attn_abl$a; done
#include "../ln3diff_amd/csrc/attention.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
  const int B = 16, H = 16, N = argc > 1 ? atoi(argv[1]) : 768, Dh = 64;
  const int NK = argc > 2 ? atoi(argv[2]) : N, NKP = (NK + 63) / 64 * 64;
  const size_t n = (size_t)B * H * N * Dh;
  std::vector<uint16_t> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = 0x3c00 + (uint16_t)((i * 2654435761u) >> 24);   // bf16 in [~0.0078, 0.03]
  void *q, *k, *v, *o;
  hipMalloc(&q, n * 2); hipMalloc(&k, n * 2); hipMalloc(&v, n * 2); hipMalloc(&o, n * 2);
  hipMemcpy(q, h.data(), n * 2, hipMemcpyHostToDevice); hipMemcpy(k, h.data(), n * 2, hipMemcpyHostToDevice);
  hipMemcpy(v, h.data(), n * 2, hipMemcpyHostToDevice);
  ln3d_attn_args a{};
  a.Q = q; a.K = k; a.Vt = v; a.O = o; a.B = B; a.H = H; a.Nq = N; a.Nq_pad = N; a.Nk = NK; a.Nk_pad = NKP; a.Dh = Dh;
  a.ldo = H * Dh; a.scale = 0.125f;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) ln3d_attention_bf16(&a, nullptr);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) ln3d_attention_bf16(&a, nullptr);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("N %d Nk %d ABL %d: %.1f us  (%.1f TF/s-equiv)\n", N, NK, LN3D_ATTN_ABL, ms / 20 * 1e3, 4.0 * N * NK * H * Dh * B / (ms / 20 * 1e-3) / 1e12);
  return 0;
}
