// Channel-last helpers for the conv decoder of the tri-plane VAE (ldm Decoder): GroupNorm(+swish),
// im2col for 3x3 convs (optionally fused with the nearest-2x upsample), layout conversion.
// The convolutions themselves run on the MFMA GEMM (gemm_bf16.hip) with fused bias / residual epilogues.
#include "common.h"
#include "../../include/ln3d.h"

// ------------------------------------------------------------------ GroupNorm (+swish) on f32 [N, HW, C] -> bf16
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* x, float* stats, int HW, int C, int groups, int pix_per_block) {
  // grid (chunks, N); thread t -> channel t % C, pixel phase t / C
  const int n = blockIdx.y;
  const int c = threadIdx.x % C, ph = threadIdx.x / C, nph = 256 / C;
  const int p0 = blockIdx.x * pix_per_block;
  const int p1 = min(p0 + pix_per_block, HW);
  float s = 0.f, q = 0.f;
  if (ph < nph)
    for (int p = p0 + ph; p < p1; p += nph) {
      const float v = x[((int64_t)n * HW + p) * C + c];
      s += v; q += v * v;
    }
  __shared__ float sh[2][256];
  sh[0][threadIdx.x] = s; sh[1][threadIdx.x] = q;
  __syncthreads();
  const int cpg = C / groups;
  if (threadIdx.x < groups) {
    float ts = 0.f, tq = 0.f;
    for (int pp = 0; pp < nph; ++pp)
      for (int cc = 0; cc < cpg; ++cc) {
        ts += sh[0][pp * C + threadIdx.x * cpg + cc];
        tq += sh[1][pp * C + threadIdx.x * cpg + cc];
      }
    // per-chunk partial sums; gn_reduce_kernel adds them in chunk order (fp32 atomics here made the planes differ from run to run)
    float* part = stats + 2 * (int64_t)gridDim.y * groups + (((int64_t)n * gridDim.x + blockIdx.x) * groups + threadIdx.x) * 2;
    part[0] = ts; part[1] = tq;
  }
}

__global__ void gn_reduce_kernel(float* stats, int N, int groups, int chunks) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;          // (n, group)
  if (i >= N * groups) return;
  const int n = i / groups, g = i % groups;
  const float* part = stats + 2 * (int64_t)N * groups + ((int64_t)n * chunks * groups + g) * 2;
  float ts = 0.f, tq = 0.f;
  for (int c = 0; c < chunks; ++c) { ts += part[(int64_t)c * groups * 2]; tq += part[(int64_t)c * groups * 2 + 1]; }
  stats[2 * i] = ts; stats[2 * i + 1] = tq;
}

__global__ void gn_apply_kernel(const float* x, const float* stats, const float* w, const float* b, bf16_t* y, int64_t total4,
                                int HW, int C, int groups, float eps, int swish) {
  const int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i4 >= total4) return;
  const int64_t i = i4 * 4;
  const int c = (int)(i % C);
  const int64_t n = i / ((int64_t)HW * C);
  const int cpg = C / groups;
  const float cnt = (float)HW * cpg;
  const float4 v = *reinterpret_cast<const float4*>(x + i);
  float in[4] = {v.x, v.y, v.z, v.w}, out[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int g = (c + k) / cpg;
    const float mean = stats[(n * groups + g) * 2] / cnt;
    const float var = fmaxf(stats[(n * groups + g) * 2 + 1] / cnt - mean * mean, 0.f);
    float t = (in[k] - mean) * rsqrtf(var + eps) * w[c + k] + b[c + k];
    if (swish) t = t / (1.0f + __expf(-t));
    out[k] = t;
  }
  uint2 o; o.x = pack2bf(out[0], out[1]); o.y = pack2bf(out[2], out[3]);
  *reinterpret_cast<uint2*>(y + i) = o;
}

extern "C" int ln3d_groupnorm_swish(const float* x, const float* w, const float* b, void* y, float* stats_scratch, int N, int HW,
                                    int C, int groups, float eps, int swish, void* stream) {
  if (!x || !w || !b || !y || !stats_scratch || C % groups || 256 % C || C % 4) return LN3D_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int ppb = LN3D_GN_PIXELS_PER_CHUNK;
  const int chunks = (HW + ppb - 1) / ppb;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(chunks, N), dim3(256), 0, s, x, stats_scratch, HW, C, groups, ppb);
  hipLaunchKernelGGL(gn_reduce_kernel, dim3((N * groups + 255) / 256), dim3(256), 0, s, stats_scratch, N, groups, chunks);
  const int64_t total4 = (int64_t)N * HW * C / 4;
  hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, x, stats_scratch, w, b, (bf16_t*)y, total4, HW,
                     C, groups, eps, swish);
  return ln3d_check_launch();
}

// ------------------------------------------------------------------ im2col 3x3 pad 1 (channel-last bf16), optional nearest-2x upsample
// col[(n*Ho*Wo + oy*Wo + ox), (ky*3+kx)*C + c] = x[n, (oy+ky-1)/up, (ox+kx-1)/up, c] (0 outside), zero pad to Kpad
__global__ void im2col3x3_kernel(const bf16_t* x, bf16_t* col, int H, int W, int C, int up, int Kpad, int64_t total8) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one 16-B (8 x bf16) piece
  if (i >= total8) return;
  const int k8 = Kpad / 8;
  const int64_t row = i / k8;
  const int kk = (int)(i % k8) * 8;
  const int Ho = H * up, Wo = W * up;
  const int ox = (int)(row % Wo), oy = (int)((row / Wo) % Ho);
  const int64_t n = row / ((int64_t)Wo * Ho);
  uint4 v = make_uint4(0, 0, 0, 0);
  if (kk < 9 * C) {
    const int tap = kk / C, c = kk % C, ky = tap / 3, kx = tap % 3;
    const int yy = oy + ky - 1, xx = ox + kx - 1;
    if (yy >= 0 && yy < Ho && xx >= 0 && xx < Wo)
      v = *reinterpret_cast<const uint4*>(x + (((int64_t)n * H + yy / up) * W + xx / up) * C + c);
  }
  *reinterpret_cast<uint4*>(col + row * Kpad + kk) = v;
}
extern "C" int ln3d_im2col3x3(const void* x, void* col, int N, int H, int W, int C, int upsample, int Kpad, void* stream) {
  if (!x || !col || C % 8 || Kpad % 8 || Kpad < 9 * C || (upsample != 1 && upsample != 2)) return LN3D_ERR_BAD_ARG;
  const int64_t rows = (int64_t)N * H * upsample * W * upsample;
  const int64_t total8 = rows * (Kpad / 8);
  hipLaunchKernelGGL(im2col3x3_kernel, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (bf16_t*)col, H, W, C, upsample, Kpad, total8);
  return ln3d_check_launch();
}

// ------------------------------------------------------------------ [NP, 3, H, W, C] f32 channel-last -> [NP, 3*C, H, W] (reference layout)
__global__ void cl_to_nchw_kernel(const float* src, float* dst, int C, int HW) {
  __shared__ float tile[32][33];
  const int pn = blockIdx.y;
  const int hw0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int hw = hw0 + r;
    tile[r][tx] = (hw < HW && tx < C) ? src[((int64_t)pn * HW + hw) * C + tx] : 0.f;
  }
  __syncthreads();
  for (int c = ty; c < C; c += 8) {
    const int hw = hw0 + tx;
    if (hw < HW) dst[((int64_t)pn * C + c) * HW + hw] = tile[tx][c];
  }
}
extern "C" int ln3d_planes_to_nchw(const float* src, float* dst, int NP, int C, int H, int W, void* stream) {
  if (!src || !dst || C != 32) return LN3D_ERR_BAD_ARG;
  const int HW = H * W;
  hipLaunchKernelGGL(cl_to_nchw_kernel, dim3((HW + 31) / 32, NP * 3), dim3(256), 0, (hipStream_t)stream, src, dst, C, HW);
  return ln3d_check_launch();
}
