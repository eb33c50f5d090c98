// Stand-alone bench + self-check of csrc/gemm_bf16.hip (no torch import: a fresh GPU box spends 1-2 minutes on that).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize [-DLN3D_RING_VAR=n] [-DLN3D_RING_ABL=n] tools/gemm_bench.hip -o build/gemm_bench
//   build/gemm_bench [rounds] [case-substring] [launches per round = 20]
// Every case: outputs hashed (FNV-1a over the raw bytes: variants that keep the summation order must agree bit for bit),
// 4096 sampled outputs checked against an fp32 dot product of the same bf16 operands (plain / GELU / gate+residual epilogues),
// then `rounds` timing rounds of 20 launches each (HIP events on the launch stream); min and median over the rounds.
// Environment: LN3D_GEMM_TILE (force a tile), LN3D_GEMM_ABL (bits 8-11 = prefetch distance of the PREFETCH variant).
#include "../ln3diff_amd/csrc/gemm_bf16.hip"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ float urand(uint32_t seed, uint64_t i) { return (hash32((uint32_t)i * 2654435761u ^ hash32(seed + (uint32_t)(i >> 32))) >> 8) * (1.0f / 8388608.0f) - 1.0f; }
__global__ void fill_bf16(bf16_t* p, int64_t n, uint32_t seed, float scale) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = f2bf(urand(seed, i) * scale);
}
__global__ void fill_f32(float* p, int64_t n, uint32_t seed, float scale) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = urand(seed, i) * scale;
}
// fp32 reference of sampled outputs: sample i -> (m, n) pseudo-random; ref[i] = sum_k X[m,k] W[n,k]
__global__ void ref_samples(const bf16_t* X, const bf16_t* W, int M, int N, int K, int ns, int* mi, int* ni, float* ref) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ns) return;
  const int m = hash32(i * 3 + 1) % M, n = hash32(i * 3 + 2) % N;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc += bf2f(X[(int64_t)m * K + k]) * bf2f(W[(int64_t)n * K + k]);
  mi[i] = m; ni[i] = n; ref[i] = acc;
}

static uint64_t fnv(const void* p, size_t n) {
  const uint8_t* b = (const uint8_t*)p; uint64_t h = 1469598103934665603ull;
  // 8 bytes at a time (order-sensitive)
  size_t i = 0;
  for (; i + 8 <= n; i += 8) { uint64_t v; memcpy(&v, b + i, 8); h = (h ^ v) * 1099511628211ull; }
  for (; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
  return h;
}
static float bf2f_h(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static float gelu_h(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

struct Case { const char* name; int M, N, K, epi; int tokens; const char* tile; };

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 3;
  const char* filt = argc > 2 ? argv[2] : "";
  const int per_round = argc > 3 ? atoi(argv[3]) : 20;      // launches per timing round: 20 = a 2 ms burst at boost clocks, 2000+ = the sustained (power-capped) rate
  const Case cases[] = {
      {"fc1 GELU_ERF", 12288, 4096, 1024, LN3D_EPI_GELU_ERF, 768, nullptr},
      {"fc1 plain", 12288, 4096, 1024, LN3D_EPI_BF16, 768, nullptr},
      {"qkv HEADS", 12288, 3072, 1024, LN3D_EPI_HEADS, 768, nullptr},
      {"qkv plain", 12288, 3072, 1024, LN3D_EPI_BF16, 768, nullptr},
      {"proj GATE_RES", 12288, 1024, 1024, LN3D_EPI_GATE_RES, 768, nullptr},
      {"proj plain", 12288, 1024, 1024, LN3D_EPI_BF16, 768, nullptr},
      {"fc2 GATE_RES", 12288, 1024, 4096, LN3D_EPI_GATE_RES, 768, nullptr},
      {"fc2 plain", 12288, 1024, 4096, LN3D_EPI_BF16, 768, nullptr},
      {"half to_out GATE_RES", 6144, 1024, 1024, LN3D_EPI_GATE_RES, 768, nullptr},
      {"half to_q CROSS_ATTN", 6144, 1024, 1024, LN3D_EPI_CROSS_ATTN, 768, nullptr},
      {"square 8192 plain", 8192, 8192, 8192, LN3D_EPI_BF16, 8192, nullptr},
      {"fc1 GELU x7 (r3 tile)", 12288, 4096, 1024, LN3D_EPI_GELU_ERF, 768, "x7"},
      {"fc1 plain x7 (r3 tile)", 12288, 4096, 1024, LN3D_EPI_BF16, 768, "x7"},
      {"qkv HEADS x12 (r3 tile)", 12288, 3072, 1024, LN3D_EPI_HEADS, 768, "x12"},
      {"qkv plain x12 (r3 tile)", 12288, 3072, 1024, LN3D_EPI_BF16, 768, "x12"},
      {"fc1 GELU x9 (256x192)", 12288, 4096, 1024, LN3D_EPI_GELU_ERF, 768, "x9"},
      {"fc1 GELU x12 (384x192)", 12288, 4096, 1024, LN3D_EPI_GELU_ERF, 768, "x12"},
      {"qkv HEADS x9 (256x192)", 12288, 3072, 1024, LN3D_EPI_HEADS, 768, "x9"},
      {"fc1 GELU x14 (128x192 x2)", 12288, 4096, 1024, LN3D_EPI_GELU_ERF, 768, "x14"},
      {"fc1 plain x14", 12288, 4096, 1024, LN3D_EPI_BF16, 768, "x14"},
      {"qkv HEADS x14", 12288, 3072, 1024, LN3D_EPI_HEADS, 768, "x14"},
      {"proj GATE_RES x14", 12288, 1024, 1024, LN3D_EPI_GATE_RES, 768, "x14"},
      {"fc2 GATE_RES x14", 12288, 1024, 4096, LN3D_EPI_GATE_RES, 768, "x14"},
      {"fc2 plain x14", 12288, 1024, 4096, LN3D_EPI_BF16, 768, "x14"},
      {"square 8192 plain x14", 8192, 8192, 8192, LN3D_EPI_BF16, 8192, "x14"},
      {"fc1 GELU x16 (persistent 4 waves)", 12288, 4096, 1024, LN3D_EPI_GELU_ERF, 768, "x16"},
      {"fc1 plain x16", 12288, 4096, 1024, LN3D_EPI_BF16, 768, "x16"},
      {"qkv plain x16", 12288, 3072, 1024, LN3D_EPI_BF16, 768, "x16"},
      {"proj plain x16", 12288, 1024, 1024, LN3D_EPI_BF16, 768, "x16"},
      {"fc2 plain x16", 12288, 1024, 4096, LN3D_EPI_BF16, 768, "x16"},
      {"square 8192 plain x16", 8192, 8192, 8192, LN3D_EPI_BF16, 8192, "x16"},
      {"i23d fc1 GELU M49152 x16", 49152, 4096, 1024, LN3D_EPI_GELU_ERF, 768, "x16"},
      {"i23d fc1 GELU M49152", 49152, 4096, 1024, LN3D_EPI_GELU_ERF, 768, nullptr},
      {"small 512x512x256 plain x16", 512, 512, 256, LN3D_EPI_BF16, 512, "x16"},
      {"odd tiles 1280x768x384 plain x16", 1280, 768, 384, LN3D_EPI_BF16, 1280, "x16"},
      {"xl2 fc1 GELU 12288x4608x1152", 12288, 4608, 1152, LN3D_EPI_GELU_ERF, 768, nullptr},
      {"xl2 fc1 GELU 12288x4608x1152 x16", 12288, 4608, 1152, LN3D_EPI_GELU_ERF, 768, "x16"},
      {"dit2 fc1 GELU 24576x4096x1024", 24576, 4096, 1024, LN3D_EPI_GELU_ERF, 3072, nullptr},
      {"dit2 fc1 GELU 24576x4096x1024 x7", 24576, 4096, 1024, LN3D_EPI_GELU_ERF, 3072, "x7"},
      {"fc1 GELU x13 (256x256, 4 waves)", 12288, 4096, 1024, LN3D_EPI_GELU_ERF, 768, "x13"},
      {"fc1 plain x13", 12288, 4096, 1024, LN3D_EPI_BF16, 768, "x13"},
      {"qkv plain x13", 12288, 3072, 1024, LN3D_EPI_BF16, 768, "x13"},
      {"proj GATE_RES x13", 12288, 1024, 1024, LN3D_EPI_GATE_RES, 768, "x13"},
      {"proj plain x13", 12288, 1024, 1024, LN3D_EPI_BF16, 768, "x13"},
      {"fc2 GATE_RES x13", 12288, 1024, 4096, LN3D_EPI_GATE_RES, 768, "x13"},
      {"fc2 plain x13", 12288, 1024, 4096, LN3D_EPI_BF16, 768, "x13"},
      {"square 8192 plain x13", 8192, 8192, 8192, LN3D_EPI_BF16, 8192, "x13"},
      {"i23d fc1 GELU M65536 x13", 65536, 4096, 1024, LN3D_EPI_GELU_ERF, 1024, "x13"},
      {"i23d fc2 GATE_RES M65536 x13", 65536, 1024, 4096, LN3D_EPI_GATE_RES, 1024, "x13"},
      {"fc2 GATE_RES x9", 12288, 1024, 4096, LN3D_EPI_GATE_RES, 768, "x9"},
      {"fc2 GATE_RES x7", 12288, 1024, 4096, LN3D_EPI_GATE_RES, 768, "x7"},
      {"i23d fc2 GATE_RES M65536 x7", 65536, 1024, 4096, LN3D_EPI_GATE_RES, 1024, "x7"},
      {"i23d fc2 GATE_RES M65536 x9", 65536, 1024, 4096, LN3D_EPI_GATE_RES, 1024, "x9"},
      {"i23d fc2 GATE_RES M49152 x13", 49152, 1024, 4096, LN3D_EPI_GATE_RES, 768, "x13"},
      {"i23d fc2 GATE_RES M49152", 49152, 1024, 4096, LN3D_EPI_GATE_RES, 768, nullptr},
      {"i23d qkv HEADS M49152", 49152, 3072, 1024, LN3D_EPI_HEADS, 768, nullptr},
      {"i23d fc1 GELU M65536", 65536, 4096, 1024, LN3D_EPI_GELU_ERF, 1024, nullptr},
      {"i23d fc2 GATE_RES M65536", 65536, 1024, 4096, LN3D_EPI_GATE_RES, 1024, nullptr},
  };
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("LN3D_RING_D1=%d LN3D_RING_ABL=%d \n", LN3D_RING_D1, LN3D_RING_ABL);
  for (const Case& c : cases) {
    if (!strstr(c.name, filt)) continue;
    if (c.tile) setenv("LN3D_GEMM_TILE", c.tile, 1); else unsetenv("LN3D_GEMM_TILE");
    ln3d_gemm_reload_env();
    const int M = c.M, N = c.N, K = c.K;
    bf16_t *X, *W; float *bias, *gate; void *o0 = nullptr, *o1 = nullptr, *o2 = nullptr; float* o0_init = nullptr;
    CK(hipMalloc(&X, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&gate, (size_t)(M / c.tokens + 1) * 6 * N * 4));
    fill_bf16<<<1024, 256, 0, st>>>(X, (int64_t)M * K, 11, 1.0f);
    fill_bf16<<<1024, 256, 0, st>>>(W, (int64_t)N * K, 12, 0.05f);
    fill_f32<<<64, 256, 0, st>>>(bias, N, 13, 0.05f);
    fill_f32<<<64, 256, 0, st>>>(gate, (int64_t)(M / c.tokens + 1) * 6 * N, 14, 1.0f);
    size_t o0b = 0, o1b = 0, o2b = 0;
    ln3d_gemm_args a; memset(&a, 0, sizeof a);
    a.X = X; a.ldx = K; a.W = W; a.ldw = K; a.bias = bias; a.M = M; a.N = N; a.K = K; a.epilogue = c.epi; a.ldo = N;
    const int B = M / c.tokens, H = 16;
    if (c.epi == LN3D_EPI_GATE_RES) {
      o0b = (size_t)M * N * 4; a.gate = gate; a.gate_rows = c.tokens; a.gate_ld = 6 * N;
      CK(hipMalloc(&o0_init, o0b)); fill_f32<<<1024, 256, 0, st>>>(o0_init, (int64_t)M * N, 15, 1.0f);
    } else if (c.epi == LN3D_EPI_HEADS) {
      o0b = o1b = o2b = (size_t)B * H * c.tokens * 64 * 2;
      a.tokens = c.tokens; a.tok_pad = c.tokens; a.heads = H; a.head_dim = 64; a.transpose_mask = 4;
    } else if (c.epi == LN3D_EPI_CROSS_ATTN) {
      o0b = (size_t)M * N * 2; o1b = o2b = (size_t)B * H * 128 * 64 * 2; a.bias = nullptr;
      a.tokens = c.tokens; a.heads = H; a.head_dim = 64; a.ctx_keys = 77; a.ctx_pad = 128; a.ctx_scale = 0.125f;
    } else o0b = (size_t)M * N * (c.epi == LN3D_EPI_F32 ? 4 : 2);
    const bool timeline = (LN3D_RING_ABL & 8) && (c.epi == LN3D_EPI_BF16 || c.epi == LN3D_EPI_GELU_ERF);
    if (timeline) o2b = (size_t)8192 * 12 * 64 * 4 * 4;       // [block][wave][stage][4] stamps
    CK(hipMalloc(&o0, o0b)); if (o1b) CK(hipMalloc(&o1, o1b)); if (o2b) CK(hipMalloc(&o2, o2b));
    CK(hipMemsetAsync(o0, 0, o0b, st));
    if (c.epi == LN3D_EPI_CROSS_ATTN) { fill_bf16<<<1024, 256, 0, st>>>((bf16_t*)o1, o1b / 2, 16, 1.0f); fill_bf16<<<1024, 256, 0, st>>>((bf16_t*)o2, o2b / 2, 17, 1.0f); }
    a.out0 = o0; a.out1 = o1; a.out2 = o2;
    // ---- one checked run
    if (o0_init) CK(hipMemcpyAsync(o0, o0_init, o0b, hipMemcpyDeviceToDevice, st));
    int rc = ln3d_gemm_bf16(&a, st);
    CK(hipStreamSynchronize(st));
    if (rc != 0) { printf("%-28s rc=%d\n", c.name, rc); continue; }
    std::vector<uint8_t> h0(o0b); CK(hipMemcpy(h0.data(), o0, o0b, hipMemcpyDeviceToHost));
    uint64_t hs = fnv(h0.data(), o0b);
    if (c.epi == LN3D_EPI_HEADS) {
      std::vector<uint8_t> h1(o1b), h2(o2b); CK(hipMemcpy(h1.data(), o1, o1b, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), o2, o2b, hipMemcpyDeviceToHost));
      hs ^= fnv(h1.data(), o1b) * 3 ^ fnv(h2.data(), o2b) * 5;
    }
    double maxerr = -1.0;
    if ((c.epi == LN3D_EPI_BF16 || c.epi == LN3D_EPI_GELU_ERF || c.epi == LN3D_EPI_GATE_RES)) {
      const int NS = 4096; int *mi, *ni; float* ref;
      CK(hipMalloc(&mi, NS * 4)); CK(hipMalloc(&ni, NS * 4)); CK(hipMalloc(&ref, NS * 4));
      ref_samples<<<NS / 64, 64, 0, st>>>(X, W, M, N, K, NS, mi, ni, ref);
      std::vector<int> hm(NS), hn(NS); std::vector<float> hr(NS), hb(N), hg((size_t)(M / c.tokens + 1) * 6 * N), hx;
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(hm.data(), mi, NS * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hn.data(), ni, NS * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hr.data(), ref, NS * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), bias, N * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hg.data(), gate, hg.size() * 4, hipMemcpyDeviceToHost));
      if (o0_init) { hx.resize((size_t)M * N); CK(hipMemcpy(hx.data(), o0_init, o0b, hipMemcpyDeviceToHost)); }
      maxerr = 0.0;
      for (int i = 0; i < NS; ++i) {
        const int m = hm[i], n = hn[i];
        float v = hr[i] + hb[n], got; double tol;
        if (c.epi == LN3D_EPI_GATE_RES) { v = hx[(size_t)m * N + n] + v * hg[(size_t)(m / c.tokens) * 6 * N + n]; got = ((float*)h0.data())[(size_t)m * N + n]; tol = 2e-3 * (1.0 + fabs(v)); }
        else { if (c.epi == LN3D_EPI_GELU_ERF) v = gelu_h(v); got = bf2f_h(((uint16_t*)h0.data())[(size_t)m * N + n]); tol = 1e-2 * (0.05 + fabs(v)); }
        const double err = fabs((double)got - v) / tol;
        maxerr = std::max(maxerr, err);
      }
      CK(hipFree(mi)); CK(hipFree(ni)); CK(hipFree(ref));
    }
    if (timeline && c.tile && !strcmp(c.tile, "x16")) {
      CK(hipMemsetAsync(o2, 0, o2b, st));
      for (int w = 0; w < 50; ++w) ln3d_gemm_bf16(&a, st);            // sustained clock first
      CK(hipMemsetAsync(o2, 0, o2b, st));
      CK(hipEventRecord(e0, st)); ln3d_gemm_bf16(&a, st); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms1; CK(hipEventElapsedTime(&ms1, e0, e1));
      std::vector<uint32_t> tl(o2b / 4); CK(hipMemcpy(tl.data(), o2, o2b, hipMemcpyDeviceToHost));
      double first = 0, loop = 0, epi = 0, gap = 0, whole = 0; long nt = 0, ng = 0, nw = 0;
      for (int b = 0; b < 256; ++b)
        for (int w = 0; w < 4; ++w) {
          const uint32_t* q = &tl[(((size_t)b * 4 + w) * 16) * 4];
          int last = -1;
          for (int k = 0; k < 16; ++k) {
            if (q[k * 4 + 3] == 0) break;
            first += (double)(uint32_t)(q[k * 4 + 1] - q[k * 4 + 0]); loop += (double)(uint32_t)(q[k * 4 + 2] - q[k * 4 + 0]); epi += (double)(uint32_t)(q[k * 4 + 3] - q[k * 4 + 2]); ++nt;
            if (k > 0) { gap += (double)(uint32_t)(q[k * 4 + 0] - q[(k - 1) * 4 + 3]); ++ng; }
            last = k;
          }
          if (last >= 0) { whole += (double)(uint32_t)(q[last * 4 + 3] - q[0]); ++nw; }
        }
      printf("  p4 timeline %-22s: 1 launch %.1f us | per tile per wave (ticks): first stage %.0f, whole K loop %.0f, epilogue %.0f, between tiles %.0f | first tile start -> last epilogue end %.0f ticks, %ld tiles\n",
             c.name, ms1 * 1000.f, first / nt, loop / nt, epi / nt, ng ? gap / ng : 0.0, whole / nw, nt);
    } else if (timeline) {
      // one launch alone, timed, then the stamps: A = before the stage's waits, B = own DMAs landed + own reads retired, C = behind the barrier
      CK(hipMemsetAsync(o2, 0, o2b, st));
      CK(hipEventRecord(e0, st)); ln3d_gemm_bf16(&a, st); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms1; CK(hipEventElapsedTime(&ms1, e0, e1));
      std::vector<uint32_t> tl(o2b / 4); CK(hipMemcpy(tl.data(), o2, o2b, hipMemcpyDeviceToHost));
      const int nst = K / 64;
      double wdma = 0, wbar = 0, seg = 0; long cnt = 0, cseg = 0; uint32_t tmin = 0xffffffffu, tmax = 0; double span = 0; long nspan = 0;
      std::vector<double> segs;
      for (int b = 0; b < 8192; ++b)
        for (int w = 0; w < 12; ++w) {
          const uint32_t* q = &tl[(((size_t)b * 12 + w) * 64) * 4];
          if (q[2] == 0 && q[6] == 0) continue;
          for (int s2 = 0; s2 + 1 < nst && s2 < 63; ++s2) {
            const uint32_t A = q[s2 * 4], B = q[s2 * 4 + 1], C = q[s2 * 4 + 2];
            if (C == 0) continue;
            wdma += (double)(uint32_t)(B - A); wbar += (double)(uint32_t)(C - B); ++cnt;
            if (s2 + 2 < nst && q[(s2 + 1) * 4 + 2] != 0) { const double d = (double)(uint32_t)(q[(s2 + 1) * 4] - C); seg += d; ++cseg; }
          }
          span += (double)(uint32_t)(q[(nst - 2) * 4 + 2] - q[2]); ++nspan;
        }
      printf("  timeline %-24s: 1 launch %.1f us | per stage per wave: wait own DMA+reads %.0f, barrier %.0f, work segment %.0f ticks (stages %ld) | first->last barrier of a tile %.0f ticks over %d stages\n",
             c.name, ms1 * 1000.f, wdma / cnt, wbar / cnt, seg / cseg, cnt, span / nspan, nst - 2);
    }
    // ---- timing
    std::vector<float> ts;
    for (int r = 0; r < rounds; ++r) {
      for (int i = 0; i < 2; ++i) ln3d_gemm_bf16(&a, st);
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < per_round; ++i) ln3d_gemm_bf16(&a, st);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms * 1000.f / per_round);
    }
    std::sort(ts.begin(), ts.end());
    const float tmin = ts[0], tmed = ts[ts.size() / 2];
    printf("%-28s M%-6d N%-5d K%-5d min %8.1f us  med %8.1f us  %7.1f TF/s  hash %016llx  err/tol %s%.3f\n", c.name, M, N, K, tmin, tmed,
           2.0 * M * N * K / tmin / 1e6, (unsigned long long)hs, maxerr > 1.0 ? "FAIL " : "", maxerr);
    fflush(stdout);
    CK(hipFree(X)); CK(hipFree(W)); CK(hipFree(bias)); CK(hipFree(gate)); CK(hipFree(o0)); if (o1) CK(hipFree(o1)); if (o2) CK(hipFree(o2)); if (o0_init) CK(hipFree(o0_init));
  }
  return 0;
}
