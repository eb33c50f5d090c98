// Kernels of the U-Net denoiser (guided_diffusion/unet.py + ldm/modules/attention_compat.py) that the DiT / conv-decoder kernels do
// not already cover: GroupNorm at any channel count with the ResBlock's per-sample embedding folded in, strided im2col for the 3x3
// stride-2 Downsample, GEGLU, attention at arbitrary head sizes over short sequences, NCHW <-> channel-last conversion and the
// mixed-prediction combination.  The convolutions and linears themselves run on the MFMA GEMM (gemm_bf16.hip); all of this is
// HBM / latency bound glue on [N, H*W, C] channel-last activations (32x32 latents: a few MB per tensor).
#include "common.h"
#include "../../include/ln3d.h"

// ------------------------------------------------------------------ GroupNorm, any C % groups == 0
// y = act( GN(x + add_row[n]) * w + b ) optionally followed by * (1 + mod_scale[n]) + mod_shift[n] before the activation
// (ResBlock: h + emb_out in front of out_layers, guided_diffusion/unet.py:272-273; use_scale_shift_norm :267-271).
// One workgroup per (sample, group); sums in a fixed order (bitwise reproducible).
__global__ __launch_bounds__(256) void gn_any_kernel(const float* x, const float* add_row, const float* w, const float* b, const float* mod_scale,
                                                    const float* mod_shift, bf16_t* y, int HW, int C, int groups, float eps, int swish) {
  const int n = blockIdx.x / groups, g = blockIdx.x % groups;
  const int cpg = C / groups, c0 = g * cpg;
  const int total = HW * cpg;
  const float* xb = x + (int64_t)n * HW * C + c0;
  const float* ar = add_row ? add_row + (int64_t)n * C + c0 : nullptr;
  // two centred passes (r6): E[x^2] - mean^2 in fp32 loses the variance when |mean| >> std (h + emb offsets, late U-Net
  // activations); the group is L2-resident, so the extra read costs little.  Fixed-order tree sums: bitwise reproducible.
  __shared__ float sh[256];
  auto block_sum = [&](float v) {
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
      __syncthreads();
    }
    const float r = sh[0];
    __syncthreads();
    return r;
  };
  float s = 0.f;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int px = e / cpg, cc = e - px * cpg;
    float v = xb[(int64_t)px * C + cc];
    if (ar) v += ar[cc];
    s += v;
  }
  const float mean = block_sum(s) / (float)total;
  float q = 0.f;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int px = e / cpg, cc = e - px * cpg;
    float v = xb[(int64_t)px * C + cc];
    if (ar) v += ar[cc];
    q += (v - mean) * (v - mean);
  }
  const float var = block_sum(q) / (float)total;
  const float rs = rsqrtf(var + eps);
  bf16_t* yb = y + (int64_t)n * HW * C + c0;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int px = e / cpg, cc = e - px * cpg;
    float v = xb[(int64_t)px * C + cc];
    if (ar) v += ar[cc];
    float t = (v - mean) * rs * w[c0 + cc] + b[c0 + cc];
    if (mod_scale) t = t * (1.0f + mod_scale[(int64_t)n * C + c0 + cc]) + mod_shift[(int64_t)n * C + c0 + cc];
    if (swish) t = t / (1.0f + __expf(-t));
    yb[(int64_t)px * C + cc] = f2bf(t);
  }
}
extern "C" int ln3d_groupnorm_any(const float* x, const float* add_row, const float* w, const float* b, const float* mod_scale,
                                  const float* mod_shift, void* y_bf16, int N, int HW, int C, int groups, float eps, int swish, void* stream) {
  if (!x || !w || !b || !y_bf16 || N <= 0 || HW <= 0 || C <= 0 || groups <= 0 || C % groups) return LN3D_ERR_BAD_ARG;
  if ((mod_scale == nullptr) != (mod_shift == nullptr)) return LN3D_ERR_BAD_ARG;
  hipLaunchKernelGGL(gn_any_kernel, dim3(N * groups), dim3(256), 0, (hipStream_t)stream, x, add_row, w, b, mod_scale, mod_shift, (bf16_t*)y_bf16, HW,
                     C, groups, eps, swish);
  return ln3d_check_launch();
}

// ------------------------------------------------------------------ im2col 3x3 pad 1 with a stride (Downsample.op, unet.py:150-153)
// col[(n*Ho*Wo + oy*Wo + ox), (ky*3+kx)*C + c] = x[n, oy*stride+ky-1, ox*stride+kx-1, c] (0 outside), zero pad to Kpad
__global__ void im2col3x3_strided_kernel(const bf16_t* x, bf16_t* col, int H, int W, int C, int stride, int Ho, int Wo, int Kpad, int64_t total8) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total8) return;
  const int k8 = Kpad / 8;
  const int64_t row = i / k8;
  const int kk = (int)(i % k8) * 8;
  const int ox = (int)(row % Wo), oy = (int)((row / Wo) % Ho);
  const int64_t n = row / ((int64_t)Wo * Ho);
  uint4 v = make_uint4(0, 0, 0, 0);
  if (kk < 9 * C) {
    const int tap = kk / C, c = kk % C, ky = tap / 3, kx = tap % 3;
    const int yy = oy * stride + ky - 1, xx = ox * stride + kx - 1;
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = *reinterpret_cast<const uint4*>(x + (((int64_t)n * H + yy) * W + xx) * C + c);
  }
  *reinterpret_cast<uint4*>(col + row * Kpad + kk) = v;
}
extern "C" int ln3d_im2col3x3_strided(const void* x, void* col, int N, int H, int W, int C, int stride, int Kpad, void* stream) {
  if (!x || !col || C % 8 || Kpad % 8 || Kpad < 9 * C || stride < 1 || stride > 2 || N <= 0) return LN3D_ERR_BAD_ARG;
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;          // floor((H + 2 - 3) / stride) + 1
  const int64_t total8 = (int64_t)N * Ho * Wo * (Kpad / 8);
  hipLaunchKernelGGL(im2col3x3_strided_kernel, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (bf16_t*)col, H, W, C, stride, Ho, Wo, Kpad, total8);
  return ln3d_check_launch();
}

// ------------------------------------------------------------------ GEGLU (attention_compat.py:45-53): y = a * gelu(gate), [a | gate] = x rows
__global__ void geglu_kernel(const float* x, bf16_t* y, int64_t rows, int inner) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * inner) return;
  const int64_t r = i / inner; const int c = (int)(i - r * inner);
  const float a = x[r * 2 * inner + c], g = x[r * 2 * inner + inner + c];
  y[i] = f2bf(a * (0.5f * g * (1.0f + erff(g * 0.70710678118654752f))));
}
extern "C" int ln3d_geglu(const float* x, void* y_bf16, int64_t rows, int inner, void* stream) {
  if (!x || !y_bf16 || rows <= 0 || inner <= 0) return LN3D_ERR_BAD_ARG;
  const int64_t n = rows * inner;
  hipLaunchKernelGGL(geglu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y_bf16, rows, inner);
  return ln3d_check_launch();
}

// ------------------------------------------------------------------ attention at any head size over short sequences
// q [B, Nq, ldq], k / v [B, Nk, ldk] bf16 token-major (the plain outputs of the projection GEMMs; head h = columns [h*Dh, (h+1)*Dh)
// at the given column offsets), out bf16 [B, Nq, H*Dh].  One wavefront per (batch, head, query): scores with lanes over the keys,
// softmax in fp32, values with lanes over the head dims.  The U-Net attends 16 - 1024 tokens at head sizes 40 - 160
// (CrossAttention, attention_compat.py:161-202; QKVAttentionLegacy, unet.py:359-389) - a few MFLOP per launch.
#define SMALL_ATTN_MAXK 1024
__global__ __launch_bounds__(256) void attn_small_kernel(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* o, int B, int H, int Nq, int Nk, int Dh,
                                                        int64_t ldq, int64_t ldk, int64_t ldv, float scale) {
  __shared__ float ps[4][SMALL_ATTN_MAXK];
  __shared__ float qs[4][256];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t item = (int64_t)blockIdx.x * 4 + wid;
  if (item >= (int64_t)B * H * Nq) return;
  const int qi = (int)(item % Nq), h = (int)((item / Nq) % H), b = (int)(item / ((int64_t)Nq * H));
  const bf16_t* qp = q + ((int64_t)b * Nq + qi) * ldq + h * Dh;
  for (int d = lane; d < Dh; d += 64) qs[wid][d] = bf2f(qp[d]);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float mx = -3.0e38f;
  for (int j = lane; j < Nk; j += 64) {
    const bf16_t* kp = k + ((int64_t)b * Nk + j) * ldk + h * Dh;
    float s = 0.f;
    for (int d = 0; d < Dh; ++d) s += qs[wid][d] * bf2f(kp[d]);
    s *= scale;
    ps[wid][j] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < Nk; j += 64) { const float e = __expf(ps[wid][j] - mx); ps[wid][j] = e; sum += e; }
  sum = wave_sum(sum);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const float inv = 1.0f / sum;
  bf16_t* op = o + ((int64_t)b * Nq + qi) * ((int64_t)H * Dh) + h * Dh;
  for (int d = lane; d < Dh; d += 64) {
    float acc = 0.f;
    const bf16_t* vp = v + (int64_t)b * Nk * ldv + h * Dh + d;
    for (int j = 0; j < Nk; ++j) acc += ps[wid][j] * bf2f(vp[(int64_t)j * ldv]);
    op[d] = f2bf(acc * inv);
  }
}
extern "C" int ln3d_attention_small(const void* q, const void* k, const void* v, void* out, int B, int H, int Nq, int Nk, int Dh, int64_t ldq,
                                    int64_t ldk, int64_t ldv, float scale, void* stream) {
  if (!q || !k || !v || !out || B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0 || Nk > SMALL_ATTN_MAXK || Dh <= 0 || Dh > 256) return LN3D_ERR_BAD_ARG;
  if (ldq < (int64_t)H * Dh || ldk < (int64_t)H * Dh || ldv < (int64_t)H * Dh) return LN3D_ERR_BAD_ARG;
  const int64_t items = (int64_t)B * H * Nq;
  hipLaunchKernelGGL(attn_small_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k,
                     (const bf16_t*)v, (bf16_t*)out, B, H, Nq, Nk, Dh, ldq, ldk, ldv, scale);
  return ln3d_check_launch();
}

// ------------------------------------------------------------------ layout: NCHW f32 -> channel-last bf16 (channels zero-padded to Cpad), and back
__global__ void nchw_to_cl_bf16_kernel(const float* x, bf16_t* y, int C, int HW, int Cpad, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // over N * HW * Cpad
  if (i >= total) return;
  const int c = (int)(i % Cpad);
  const int64_t p = i / Cpad;
  const int64_t n = p / HW; const int hw = (int)(p - n * HW);
  y[i] = c < C ? f2bf(x[(n * C + c) * HW + hw]) : (bf16_t)0;
}
extern "C" int ln3d_nchw_to_cl_bf16(const float* x, void* y_bf16, int N, int C, int HW, int Cpad, void* stream) {
  if (!x || !y_bf16 || N <= 0 || C <= 0 || HW <= 0 || Cpad < C) return LN3D_ERR_BAD_ARG;
  const int64_t total = (int64_t)N * HW * Cpad;
  hipLaunchKernelGGL(nchw_to_cl_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y_bf16, C, HW, Cpad, total);
  return ln3d_check_launch();
}
__global__ void cl_to_nchw_f32_kernel(const float* x, float* y, int C, int HW, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // over N * C * HW (output order)
  if (i >= total) return;
  const int hw = (int)(i % HW);
  const int64_t nc = i / HW;
  const int64_t n = nc / C; const int c = (int)(nc - n * C);
  y[i] = x[(n * HW + hw) * C + c];
}
extern "C" int ln3d_cl_to_nchw_f32(const float* x, float* y, int N, int C, int HW, void* stream) {
  if (!x || !y || N <= 0 || C <= 0 || HW <= 0) return LN3D_ERR_BAD_ARG;
  const int64_t total = (int64_t)N * C * HW;
  hipLaunchKernelGGL(cl_to_nchw_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, C, HW, total);
  return ln3d_check_launch();
}

// ------------------------------------------------------------------ mixed prediction (LSGM): out = (1 - s_c) * sqrt(1 - ab_t) * x + s_c * eps
// s_c = sigmoid(mixing_logit[c]) per channel (continuous_diffusion_utils.py:748-754, gaussian_diffusion.py:336-348); NCHW f32, in place on eps
__global__ void mix_prediction_kernel(float* eps, const float* x, const float* logit, float s1mab, int C, int HW, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)((i / HW) % C);
  const float s = 1.0f / (1.0f + __expf(-logit[c]));
  eps[i] = (1.0f - s) * (s1mab * x[i]) + s * eps[i];
}
extern "C" int ln3d_mix_prediction(float* eps, const float* x, const float* mixing_logit, float sqrt_one_minus_ab, int N, int C, int HW, void* stream) {
  if (!eps || !x || !mixing_logit || N <= 0 || C <= 0 || HW <= 0) return LN3D_ERR_BAD_ARG;
  const int64_t total = (int64_t)N * C * HW;
  hipLaunchKernelGGL(mix_prediction_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, eps, x, mixing_logit,
                     sqrt_one_minus_ab, C, HW, total);
  return ln3d_check_launch();
}
