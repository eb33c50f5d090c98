// Stand-alone self-checking benchmark of ln3d_attention_bf16 (GPU box; build in the container, the binary ships with gpurun):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/attn_bench.hip -o build/attn_bench && build/attn_bench
// For every case the shipped kernel and - on the shapes it takes - the r5 experiment tools/attn_kres1w.hip (build with
// -DLN3D_K1W_ABL=bits / -DLN3D_K1W_OPT=bits for its ablation / option builds) are timed with
// HIP events on random data and their output is compared, element by element, with a naive fp32 kernel on the SAME bf16
// operands.  Per head one key row is spiked against one query row: x3 (a score ~2^26 above the rest: the deferred-rebase branch
// of the r1 / r2 kernels) or, in every third head, x40 (~2^346: overflows the fixed reference of attn_kres_kernel, so its
// exact recomputation path runs); heads with bh % 5 == 1 carry the spike in the FIRST tile instead (everything else underflows).
#include "../ln3diff_amd/csrc/attention.hip"
#include "attn_kres1w.hip"   // r5 experiment: one wave per SIMD, AGPR-pinned accumulators (not in the library)
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void naive_attn(const bf16_t* Q, const bf16_t* K, const bf16_t* Vt, float* O, int H, int Nq, int Nqp, int Nk, int Nkp, int Dh,
                           float scale) {
  const int q = blockIdx.x, bh = blockIdx.y, d = threadIdx.x;          // one thread per output dim
  extern __shared__ float sc[];                                         // Nk scores
  const bf16_t* qp = Q + ((int64_t)bh * Nqp + q) * Dh;
  for (int k = threadIdx.x; k < Nk; k += blockDim.x) {
    const bf16_t* kp = K + ((int64_t)bh * Nkp + k) * Dh;
    float s = 0.f;
    for (int i = 0; i < Dh; ++i) s += bf2f(qp[i]) * bf2f(kp[i]);
    sc[k] = s * scale;
  }
  __syncthreads();
  float mx = -3e38f;
  for (int k = 0; k < Nk; ++k) mx = fmaxf(mx, sc[k]);
  float l = 0.f, o = 0.f;
  for (int k = 0; k < Nk; ++k) {
    const float pv = expf(sc[k] - mx);
    const int kp = (k & ~12) | ((k & 4) << 1) | ((k & 8) >> 1);         // V^T key order of the ABI
    l += pv;
    o += pv * bf2f(Vt[((int64_t)bh * Dh + d) * Nkp + kp]);
  }
  const int b = bh / H, h = bh - b * H;
  O[((int64_t)b * Nq + q) * (H * Dh) + h * Dh + d] = o / l;
}

static uint16_t f2bf_host(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float frand(uint64_t& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((s >> 40) & 0xffffff) / 8388608.0f - 1.0f; }

struct Case { int B, H, Nq, Nk, ovf; };

int main() {
  const int Dh = 64;
  const Case cases[] = {{16, 16, 768, 768, 0}, {16, 16, 768, 768, 1}, {16, 16, 1024, 1024, 0}, {32, 16, 512, 512, 1}, {32, 16, 512, 512, 0},
                        {2, 16, 768, 768, 1}, {1, 4, 700, 1000, 1}, {2, 3, 300, 832, 1}, {1, 2, 257, 257, 1}};
  const struct { const char* name; int k1w; } variants[] = {{"shipped kernel", 0}, {"kres1w (r5)", 1}};
  const int ncases = getenv("ATTN_BENCH_CASES") ? atoi(getenv("ATTN_BENCH_CASES")) : 100;
  int ci = 0;
  for (const Case& c : cases) {
    if (ci++ >= ncases) break;
    const int Nqp = (c.Nq + 63) / 64 * 64, Nkp = (c.Nk + 63) / 64 * 64, BH = c.B * c.H;
    const size_t nq = (size_t)BH * Nqp * Dh, nk = (size_t)BH * Nkp * Dh, no = (size_t)c.B * c.Nq * c.H * Dh;
    std::vector<uint16_t> hq(nq, 0), hk(nk, 0), hv(nk, 0);
    uint64_t seed = 1234567 + c.Nq * 31 + c.Nk;
    for (int bh = 0; bh < BH; ++bh) {
      for (int r = 0; r < c.Nq; ++r) for (int d = 0; d < Dh; ++d) hq[((size_t)bh * Nqp + r) * Dh + d] = f2bf_host(1.5f * frand(seed));
      for (int r = 0; r < c.Nk; ++r) for (int d = 0; d < Dh; ++d) hk[((size_t)bh * Nkp + r) * Dh + d] = f2bf_host(1.5f * frand(seed));
      for (int d = 0; d < Dh; ++d) for (int r = 0; r < c.Nk; ++r) {
        const int rp = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
        hv[((size_t)bh * Dh + d) * Nkp + rp] = f2bf_host(frand(seed) + (float)d / Dh);
      }
      // spike: a key equal to 3x (every third head: 40x) query row 3
      const int ks = (bh % 5 == 1) ? 7 : c.Nk - 5;
      const float fac = (c.ovf && bh % 3 == 0) ? 40.0f : 3.0f;
      for (int d = 0; d < Dh; ++d) {
        uint32_t u = (uint32_t)hq[((size_t)bh * Nqp + 3) * Dh + d] << 16; float f; memcpy(&f, &u, 4);
        hk[((size_t)bh * Nkp + ks) * Dh + d] = f2bf_host(fac * f);
      }
    }
    void *q, *k, *v, *o; float* oref;
    hipMalloc(&q, nq * 2); hipMalloc(&k, nk * 2); hipMalloc(&v, nk * 2); hipMalloc(&o, no * 2); hipMalloc(&oref, no * 4);
    hipMemcpy(q, hq.data(), nq * 2, hipMemcpyHostToDevice); hipMemcpy(k, hk.data(), nk * 2, hipMemcpyHostToDevice);
    hipMemcpy(v, hv.data(), nk * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(naive_attn, dim3(c.Nq, BH), dim3(Dh), c.Nk * sizeof(float), 0, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                       oref, c.H, c.Nq, Nqp, c.Nk, Nkp, Dh, 0.125f);
    std::vector<float> href(no);
    hipMemcpy(href.data(), oref, no * 4, hipMemcpyDeviceToHost);
    ln3d_attn_args a{};
    a.Q = q; a.K = k; a.Vt = v; a.O = o; a.B = c.B; a.H = c.H; a.Nq = c.Nq; a.Nq_pad = Nqp; a.Nk = c.Nk; a.Nk_pad = Nkp; a.Dh = Dh;
    a.ldo = c.H * Dh; a.scale = 0.125f;
    printf("B %d H %d Nq %d Nk %d%s\n", c.B, c.H, c.Nq, c.Nk, c.ovf ? "  (x40 spikes: kres recomputes those heads)" : "");
    int vi = 0;
    for (const auto& var : variants) {
      if (getenv("ATTN_BENCH_VAR") && atoi(getenv("ATTN_BENCH_VAR")) != vi++) continue;
      const bool k1w_shape = c.Nk >= 512 && c.Nk <= 768 && (c.Nk & 255) == 0 && (c.Nq & 255) == 0 && Nqp == c.Nq && Nkp == c.Nk && BH >= 256;
      if (var.k1w && !k1w_shape) continue;
      auto run = [&]() -> int {
        if (!var.k1w) return ln3d_attention_bf16(&a, nullptr);
        AttnP p;
        p.Q = (const bf16_t*)a.Q; p.K = (const bf16_t*)a.K; p.Vt = (const bf16_t*)a.Vt; p.O = (bf16_t*)a.O;
        p.B = a.B; p.H = a.H; p.Nq = a.Nq; p.Nq_pad = a.Nq_pad; p.Nk = a.Nk; p.Nk_pad = a.Nk_pad; p.ldo = a.ldo;
        p.scale_log2 = a.scale * 1.4426950408889634f; p.causal = 0; p.nsplit = 1;
        return launch_attn_kres1w(p, nullptr);
      };
      hipMemset(o, 0xff, no * 2);
      const int rc = run();
      hipError_t e = hipDeviceSynchronize();
      if (rc != 0 || e != hipSuccess) { printf("  %-18s FAILED rc %d hip %d\n", var.name, rc, (int)e); return 1; }
      std::vector<uint16_t> ho(no), ho2(no);
      hipMemcpy(ho.data(), o, no * 2, hipMemcpyDeviceToHost);
      size_t nd = 0;                                      // determinism: repeated launches must agree bit for bit
      for (int rep = 0; rep < 4; ++rep) {
        run(); hipDeviceSynchronize();
        hipMemcpy(ho2.data(), o, no * 2, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < no; ++i) nd += ho[i] != ho2[i];
      }
      if (nd) printf("  %-18s NONDETERMINISTIC: %zu elements differ over 4 repeats\n", var.name, nd);
      double num = 0, den = 0, mxe = 0;
      for (size_t i = 0; i < no; ++i) {
        uint32_t u = (uint32_t)ho[i] << 16; float f; memcpy(&f, &u, 4);
        const double dlt = (double)f - href[i];
        num += dlt * dlt; den += (double)href[i] * href[i];
        if (!(std::fabs(dlt) <= mxe)) mxe = std::fabs(dlt);
      }
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int i = 0; i < 5; ++i) run();
      float best = 1e30f, sum = 0.f;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) run();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = fminf(best, ms / 20); sum += ms / 20;
      }
      const double fl = 4.0 * c.Nq * c.Nk * c.H * Dh * c.B;
      printf("  %-18s rel-L2 %.2e  max|err| %.2e   %8.1f us avg %8.1f us best  %7.1f TF/s (best)\n", var.name, std::sqrt(num / den), mxe,
             sum / 5 * 1e3, best * 1e3, fl / (best * 1e-3) / 1e12);
    }
    hipFree(q); hipFree(k); hipFree(v); hipFree(o); hipFree(oref);
  }
  return 0;
}
