/* ln3d.h - C ABI of libln3d_hip.so: the MI355X (gfx950) kernels behind LN3Diff's
 * text/image->3D sampling hot path.
 *
 * The reference (NIRVANALAN/LN3Diff) has NO C/FFI seam on this path: its seam is Python
 * module surfaces plus calls into third-party fused ops.  Each entry point
 * below names the reference call it replaces.  Conventions:
 *   - every pointer is a caller-owned DEVICE pointer (HBM), contiguous in the documented
 *     layout; no ownership transfer, no allocation, no host sync inside;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued asynchronously on it;
 *   - return 0 on success, negative LN3D_ERR_* otherwise (ln3d_strerror());
 *   - bf16 tensors are raw uint16 bfloat16; "f32" is IEEE float;
 *   - random numbers are never drawn inside: noise / jitter are input pointers.
 */
#ifndef LN3D_H
#define LN3D_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LN3D_OK 0
#define LN3D_ERR_BAD_ARG (-1)
#define LN3D_ERR_LAUNCH (-2)
#define LN3D_ERR_UNSUPPORTED (-3)

const char* ln3d_strerror(int code);
int ln3d_abi_version(void);
/* The one measurement switch left in the library (environment variable LN3D_GEMM_TILE: force a tile configuration, swept by the tests)
 * is read ONCE per process, at the first launch that consults it.  A harness that changes it afterwards calls this to have it re-read. */
void ln3d_reload_env(void);

/* Compute units of the current device (the tile selection of the GEMM / attention launchers counts them).  ABI 9 also had
 * two entry points for streams restricted to a subset of the CUs (create-with-CU-mask, CU count of a stream), for an experiment that
 * measured neutral: profiles/r5_lanes.md; removed in ABI 10. */
int ln3d_device_cus(void);
/* Diagnostic (ABI 10): a pure-MFMA stream (wgs workgroups x 8 waves x iters x 8 v_mfma_f32_32x32x16_bf16, no memory traffic) whose timing
 * gives the matrix rate this box SUSTAINS under its power management - bench.py prints it beside the datasheet peak.  out: wgs * 512 floats.
 * flop = wgs * 8 * iters * 8 * 2 * 32 * 32 * 16.  No reference counterpart (measurement only). */
int ln3d_probe_mfma_bf16(float* out, int wgs, int iters, void* stream);

/* ---------------------------------------------------------------- GEMM with fused epilogues
 * out[m, n] = epilogue( sum_k X[m,k] * W[n,k] + bias[n] ),  X:[M,K] bf16 (tokens), W:[N,K] bf16
 * (torch.nn.Linear weight layout).  fp32 accumulation on MFMA 32x32x16 bf16.  K % 64 == 0, N % 4 == 0.
 * Replaces: torch.nn.Linear / F.linear inside
 *   vit/vision_transformer.py:112,121 (qkv, proj), ldm/modules/attention.py:279-283,307 (to_q/k/v/out),
 *   xformers FusedMLP at dit/dit_models_xformers.py:278-283, adaLN at :285-286,311-312,
 *   TimestepEmbedder :94-98, CaptionEmbedder :197-201, and the 1x1/3x3 convs of
 *   ldm/modules/diffusionmodules/model.py via im2col (ln3d_im2col3x3).
 */
enum {
  LN3D_EPI_F32 = 0,        /* out0 f32 [M,ldo]                                             */
  LN3D_EPI_BF16 = 1,       /* out0 bf16 [M,ldo]                                            */
  LN3D_EPI_GELU_ERF = 2,   /* out0 bf16 = gelu_erf(.)   (xformers Activation.GeLU)          */
  LN3D_EPI_GELU_TANH = 3,  /* out0 bf16 = gelu_tanh(.)  (CaptionEmbedder approx_gelu)       */
  LN3D_EPI_SILU = 4,       /* out0 bf16 = silu(.)                                          */
  LN3D_EPI_GATE_RES = 5,   /* out0 f32 [M,ldo] += gate * (.) ; optional out1 bf16 copy      */
  LN3D_EPI_HEADS = 6,      /* split columns into heads: out{0,1,2} bf16, see below          */
  LN3D_EPI_F32_SILU = 7,   /* out0 f32 raw and out1 bf16 = silu(.)                           */
  LN3D_EPI_QUICK_GELU = 8, /* out0 bf16 = x * sigmoid(1.702 x)  (CLIP text MLP, hidden_act quick_gelu) */
  LN3D_EPI_CROSS_ATTN = 9  /* the GEMM is a cross-attention query projection (no bias) and the attention over a SHORT cached
                            * context runs in its epilogue: out0 bf16 [M, ldo] = softmax(ctx_scale * q_h K_h^T) V_h per head of
                            * 64 features, q never leaving the accumulators.  out1 = K cache bf16 [M/tokens, heads, ctx_pad, 64]
                            * with the 64 head dims of every row stored in the 16-group order [0-3, 8-11, 4-7, 12-15];
                            * out2 = V^T cache bf16 [M/tokens, heads, 64, ctx_pad] (key order as for ln3d_attention_bf16).
                            * Requires head_dim 64, N = heads*64, tokens % 192 == 0, ctx_keys <= 96, ctx_pad % 64 == 0.
                            * (TextCondDiTBlock cross-attention, dit/dit_models_xformers.py:298-323 + ldm/modules/attention.py:278) */
};

typedef struct {
  const void* X; int64_t ldx;   /* bf16 [M, ldx]  */
  const void* W; int64_t ldw;   /* bf16 [N, ldw]  */
  const float* bias;            /* [N] or NULL    */
  int M, N, K;
  int epilogue;
  void* out0; void* out1; void* out2;
  int64_t ldo;
  /* GATE_RES: gate[(m / gate_rows) * gate_ld + n] (f32), NULL = 1.  */
  const float* gate; int gate_rows; int64_t gate_ld;
  /* HEADS: column n -> which = n / (heads*head_dim), h, d; row m -> b = m / tokens, t = m % tokens.
   *   which w writes out{w}: layout [B, heads, tok_pad, head_dim] if !(transpose_mask>>w & 1)
   *   else [B, heads, head_dim, tok_pad] with tokens key-permuted inside 16-groups (V^T for ln3d_attention_bf16). */
  int tokens; int tok_pad; int heads; int head_dim; int transpose_mask;
  int ctx_keys, ctx_pad;   /* CROSS_ATTN: context length (<= 96) and its padded row count in the K / V^T caches */
  float ctx_scale;         /* CROSS_ATTN: softmax scale (head_dim^-0.5) */
  int head_dim_pad;   /* HEADS: destination head size (>= head_dim; 0 = head_dim).  DiT-XL/2 (head_dim 72) writes into
                         128-wide zero-initialised heads so that ln3d_attention_bf16 (Dh 64/128) serves it */
  /* HEADS: optional qk_norm fused into the epilogue - per-head RMSNorm weights [64] for output 0 (q) / output 1 (k), applied to
   * the fp32 accumulators (+ bias) before the single rounding to bf16: x * rsqrt(mean_64(x^2) + head_norm_eps) * w
   * (vit/vision_transformer.py:81-82,116; ldm/modules/attention.py:264-265,294).  Needs head_dim == 64, tokens % 32 == 0,
   * M % tokens == 0, M >= 1536, N >= 128 (the head-aligned tiles); otherwise LN3D_ERR_UNSUPPORTED - run ln3d_rmsnorm_heads_bf16
   * after the GEMM instead.  NULL = no normalisation. */
  const float* head_norm0; const float* head_norm1; float head_norm_eps;
  /* GATE_RES (ABI 7): optional per-sample row added to the residual AFTER gating, res_bias[(m / gate_rows) * res_bias_ld + n] (f32):
   *   out0[m, n] += gate * (acc + bias) + res_bias.   Carries the cross-attention output of samples whose context rows are all
   *   identical (the zero embeddings of the unconditional CFG branch, sgm_DiffusionEngine.py:448-452): softmax over identical keys
   *   is uniform, so that sub-block is the constant to_out(v) + b per sample and layer, and its two GEMMs are skipped for them. */
  const float* res_bias; int64_t res_bias_ld;
  /* (ABI 8 carried arguments that fused LayerNorm / RMSNorm + modulate into the GEMMs around it; measured slower than the standalone
   * norm kernel in situ - profiles/r4_gemm.md section 5 - and removed in ABI 9.) */
} ln3d_gemm_args;

int ln3d_gemm_bf16(const ln3d_gemm_args* a, void* stream);
/* 1 when ln3d_gemm_bf16's HEADS epilogue will apply head_norm0/1 itself for this problem (the tile the library picks is
 * head-aligned), 0 when the caller has to run ln3d_rmsnorm_heads_bf16 after the GEMM. */
int ln3d_gemm_heads_norm_fusable(int M, int N, int tokens, int head_dim, int head_dim_pad);

/* ---------------------------------------------------------------- fused attention
 * O[b, q, h*Dh + d] = softmax_k(scale * Q.K^T) V, bf16 in/out, fp32 softmax, MFMA 32x32x16.
 *   Q  : bf16 [B, H, Nq_pad, Dh]   K : bf16 [B, H, Nk_pad, Dh]   Vt : bf16 [B, H, Dh, Nk_pad] with the keys of every
 *   16-group stored in the order [0-3, 8-11, 4-7, 12-15] (position p holds key p with bits 2,3 swapped) - the layout
 *   ln3d_gemm_bf16's LN3D_EPI_HEADS epilogue emits for transposed outputs.
 *   O  : bf16 [B, Nq, ldo]  (only rows q < Nq written).  Keys k >= Nk are masked; Nk_pad % 64 == 0 and
 *   the padded K / Vt entries must be finite (zero).  Dh in {64, 80, 128} (80: r6, the stored width of 65 - 80 wide heads).
 * Replaces xformers.ops.memory_efficient_attention at vit/vision_transformer.py:118,
 *   ldm/modules/attention.py:297 and ldm/modules/diffusionmodules/model.py:262.
 */
typedef struct {
  const void* Q; const void* K; const void* Vt; void* O;
  int B, H, Nq, Nq_pad, Nk, Nk_pad, Dh;
  int64_t ldo;
  float scale;
  int causal;              /* 1: query i attends keys <= i (CLIP text tower; Dh 64 and Nk <= 128 only); 0: full attention */
  int Dh_true;             /* ABI 8: true head size when Q / K / Vt rows are stored zero-padded to Dh (0 or Dh = not padded).  72 with
                            * Dh 80 (r6) or Dh 128 (DiT-XL/2, dit/dit_trilatent.py:272: hidden 1152 / 16 heads): the products skip the padding and
                            * O is written COMPACT, O[b, q, h*Dh_true + d] with ldo = H*Dh_true, so the projection behind it
                            * contracts over H*Dh_true.  Other values: LN3D_ERR_UNSUPPORTED. */
} ln3d_attn_args;
int ln3d_attention_bf16(const ln3d_attn_args* a, void* stream);

/* ---------------------------------------------------------------- text conditioner helpers (CLIP-L text tower =
 * sgm/modules/encoders/modules.py:347-405 -> HuggingFace CLIPTextModel)
 * out[b*T + t, :] = tok_emb[ids[b*T + t], :] + pos_emb[t, :]   (f32) */
int ln3d_embed_tokens(const int32_t* ids, const float* tok_emb, const float* pos_emb, float* out, int B, int T, int D, int vocab,
                      void* stream);
/* affine LayerNorm f32 -> f32 (final_layer_norm; last_hidden_state / pooled stay fp32), D % 128 == 0, D <= 1152 */
int ln3d_layernorm_f32(const float* x, const float* w, const float* b, float* y, int64_t rows, int D, float eps, void* stream);

/* ---------------------------------------------------------------- image conditioner helpers (open_clip ViT-L/14 visual tower,
 * DINOv2 ViT-L/14-reg: sgm/modules/encoders/modules.py:578-869)
 * patchify: out bf16 [B*(S/p)^2, Kpad], column c*p*p + i*p + j = img[b, c, gy*p+i, gx*p+j], columns >= C*p*p zero
 * assemble: x f32 [B, 1+R+L, D]: cls + pos[0] ; R register tokens ; patch[b, n] + pos[1+n] */
int ln3d_vit_patchify(const float* img, void* out_bf16, int B, int S, int p, int Kpad, int C, void* stream);   /* ABI 8: C input channels (3; 9 = RGB + Pluecker rays) */
/* Pluecker ray maps of V posed views for the multi-view conditioner (FrozenDinov2ImageEmbedderMVPlucker.get_plucker_ray,
 * sgm/modules/encoders/modules.py:958-1005): c f32 [V, 25] = camera-to-world 4x4 (row-major) + normalised intrinsics 3x3;
 * out f32 [V, 6, S, S] = (origin x direction, direction) per pixel centre of an S x S grid. */
int ln3d_plucker_rays(const float* c, float* out, int V, int S, void* stream);
int ln3d_vit_assemble(const float* patch, const float* cls, const float* reg, const float* pos, float* x, int B, int L, int R, int D,
                      void* stream);
/* kornia.geometry.transform.resize(x, (S, S), 'bicubic', align_corners=True, antialias) -> (x + 1) / 2 -> (x - mean) / std
 * (the embedders' preprocess(), sgm/modules/encoders/modules.py:633-645,802-814).  x f32 [N, C, H, W] in [-1, 1] -> out f32 [N, C, S, S].
 * tmp: caller-owned 2 * N*C*H*W floats for the two Gaussian passes (needed only when antialias and a side shrinks);
 * mean_host / std_host: C floats in HOST memory (read at call time). */
int ln3d_image_preprocess(const float* x, float* out, float* tmp, int N, int C, int H, int W, int S, int antialias,
                          const float* mean_host, const float* std_host, void* stream);

/* per-head RMSNorm of q / k in place: x[row, 0:Dh] * rsqrt(mean(x^2)+eps) * w   (qk_norm,
 * vit/vision_transformer.py:81-82,116; ldm/modules/attention.py:264-265,294; dit/norm.py:27-40).
 * Dh = stored row width (64, 80 or 128); true_dim = the head size the mean is taken over when heads are zero-padded to Dh
 * (DiT-XL: 72 in rows of 80; w then has Dh entries, zero beyond true_dim); 0 = Dh */
int ln3d_rmsnorm_heads_bf16(void* x, const float* w, int64_t rows, int Dh, int true_dim, float eps, void* stream);

/* ---------------------------------------------------------------- norm + modulation
 * y[r, :] = norm(x[r, :]) * (1 + scale) + shift  -> bf16, one wavefront per row, fp32 statistics.
 *   kind 0: LayerNorm(eps, no affine)   (dit/dit_models_xformers.py:249-258, modulate :48)
 *   kind 1: RMSNorm(eps) * weight        (dit/norm.py:27-40, t2i_modulate :52)
 *   shift/scale: f32, element [ (r / mod_rows) * mod_ld + d ]; NULL = no modulation.
 *   table (optional f32 [D] each): added to shift / scale (PixArt scale_shift_table rows).
 *   output row = (r / rows_in) * rows_out + (r % rows_in)   (rows_out > rows_in leaves room for
 *   the appended DINO tokens, dit/dit_models_xformers.py:522-526).  D % 128 == 0, D <= 1152.
 */
typedef struct {
  const float* x; void* y; int64_t rows; int D;
  int kind; float eps; const float* weight;
  const float* shift; const float* scale; int mod_rows; int64_t mod_ld;
  const float* shift_table; const float* scale_table;
  int rows_in; int rows_out;
} ln3d_norm_args;
int ln3d_norm_modulate(const ln3d_norm_args* a, void* stream);

/* ---------------------------------------------------------------- DiT boundary ops */
/* TimestepEmbedder.timestep_embedding (dit/dit_models_xformers.py:101-122): t[B] f32 -> bf16 [B,256] = [cos|sin] */
int ln3d_timestep_embedding(const float* t, void* out_bf16, int B, int dim, void* stream);

/* y_bf16[i] = act(a[i] + b[i])  (b may be NULL); act 0 = identity, 1 = SiLU; optional f32 copy of a+b */
int ln3d_add_act_cast(const float* a, const float* b, void* y_bf16, float* sum_f32, int64_t n, int act, void* stream);
int ln3d_cast_f32_bf16(const float* x, void* y, int64_t n, void* stream);

/* DiT_TriLatent patchify + x_embedder + pos_embed (dit/dit_trilatent.py:93-98):
 *   x f32 [Bx, C*3, S, S] (channel = c*3 + n), network batch Bn >= Bx uses sample b % Bx;
 *   optional in_scale[Bn] multiplies the input (EDM c_in, sgm denoiser.py:36-39);
 *   w f32 [D, C*p*p], bias [D], pos f32 [3*L, D]  ->  tokens f32 [Bn, 3*L, D]                         */
int ln3d_patch_embed(const float* x, const float* in_scale, const float* w, const float* bias,
                     const float* pos, float* tokens, int Bx, int Bn, int C, int S, int p, int D, void* stream);

/* PatchEmbedTriplane of the VAE decoder (vit/vit_triplane.py:58-110): latent f32 [B, 3*Cg, S, S] -> per-token
 * conditioning c [B, 3*L, D]; emitted as bf16 silu(c) (the only form DiTBlock2's adaLN consumes) and
 * optionally raw f32.  w f32 [3*D, Cg, p, p] (groups = 3), bias [3*D].                                  */
int ln3d_patch_embed_triplane(const float* latent, const float* w, const float* bias, void* out_silu_bf16,
                              float* out_raw, int B, int Cg, int S, int p, int D, void* stream);
/* y[r*per + i] = x[i] for r < reps (DiT2 queries start from pos_embed, dit/dit_decoder.py:104-105) */
int ln3d_tile_rows(const float* x, float* y, int64_t per, int reps, void* stream);

/* FinalLayer / T2IFinalLayer + unpatchify (dit/dit_models_xformers.py:655-678, :61-84, :821-835,
 * dit/dit_trilatent.py:128-140): LN(eps 1e-6) -> *(1+scale)+shift -> Linear(D -> p*p*C) -> [Bn, C*3, S, S] f32.
 *   shift/scale f32 per sample at [b*mod_ld + d]; optional tables added.                              */
int ln3d_final_layer(const float* tokens, const float* shift, const float* scale, int64_t mod_ld,
                     const float* shift_table, const float* scale_table,
                     const float* w, const float* bias, float* out,
                     int Bn, int C, int S, int p, int D, void* stream);

/* ---------------------------------------------------------------- sampler steps (elementwise, f32)
 * EulerEDMSampler step with VanillaCFG on the eps-model (sgm sampling.py:93-104, guiders.py:29-42,
 * denoiser.py:36-41, sampling_utils.py:34): eps[2B,...] = [uncond ; cond] network output on x*c_in.   */
int ln3d_edm_euler_step(float* x, const float* eps2, float sigma, float sigma_next, float cfg_scale,
                        int64_t n_per_batch_total, void* stream);
/* GaussianDiffusion.p_sample, EPSILON / FIXED_LARGE (guided_diffusion/gaussian_diffusion.py:422-427,
 * 252-271, 535-545): x <- c1*x0 + c2*x + nonzero*exp(0.5*logvar)*noise, x0 = a*x - b*eps (opt. clip)  */
int ln3d_ddpm_step(float* x, const float* eps, const float* noise, float sqrt_recip, float sqrt_recipm1,
                   float coef1, float coef2, float sigma_t, int clip, int64_t n, void* stream);
/* GaussianDiffusion.ddim_sample (guided_diffusion/gaussian_diffusion.py:729-866): eps = eu + s*(ec-eu) (eps_c NULL = no CFG),
 * x0 = a*x - b*eps, x <- sqrt(ab_prev)*x0 + coef_eps*eps + sigma*noise (noise NULL when sigma == 0 / t == 0)         */
int ln3d_ddim_step(float* x, const float* eps_u, const float* eps_c, const float* noise, float cfg_scale, float sqrt_recip,
                   float sqrt_recipm1, float sqrt_ab_prev, float coef_eps, float sigma, int clip, int64_t n, void* stream);
/* flow matching Euler + CFG (transport/integrators.py:101-120, dit/dit_i23d.py:155-168):
 * v[2B] = [cond ; uncond]; x[2B] (both halves updated identically): x += dt * (vu + s*(vc - vu))       */
int ln3d_flow_euler_step(float* x2, const float* v2, float dt, float cfg_scale, int64_t n_half, void* stream);
/* PixArt shared adaLN (dit/dit_models_xformers.py:518-519): out[l,b,:] = tables[l,:] + t0[b,:], W = 6*D */
int ln3d_add_table_rows(const float* t0, const float* tables, float* out, int layers, int B, int64_t W, void* stream);
/* forward_with_cfg (dit/dit_i23d.py:155-168): v[2B] = [cond ; uncond] -> both halves = uncond + s*(cond-uncond) */
int ln3d_cfg_combine_dup(float* v2, float cfg_scale, int64_t n_half, void* stream);
/* Runge-Kutta stage combination out = y + sum_j cs[j]*ks[j] (y may be NULL; ks/cs are HOST arrays of <= 7 device pointers /
 * coefficients) and acc = sum((err/(atol + rtol*max(|y0|,|y1|)))^2) - the adaptive Dormand-Prince solver behind
 * transport's sampling_method='dopri5' (transport/integrators.py:112-119 -> torchdiffeq, third-party: parity unpinned) */
int ln3d_lincomb(const float* y, const float* const* ks, const float* cs, int nterms, float* out, int64_t n, void* stream);
int ln3d_err_ratio_sq(const float* err, const float* y0, const float* y1, float atol, float rtol, float* acc, int64_t n, void* stream);
/* y = a*x + b*y (axpby, f32) - Heun / generic combinations */
int ln3d_axpby(const float* x, float* y, float a, float b, int64_t n, void* stream);

/* ---------------------------------------------------------------- tri-plane renderer
 * planes: f32 channel-last [NP, 3, H, W, 32]  (reference layout is [NP, 96, H, W] = (n c) h w;
 * ln3d_planes_to_channel_last converts).  One wavefront per ray.
 */
int ln3d_planes_to_channel_last(const float* planes_nchw, float* planes_nhwc, int NP, int C, int H, int W, void* stream);
int ln3d_planes_to_nchw(const float* planes_nhwc, float* planes_nchw, int NP, int C, int H, int W, void* stream);
#define LN3D_RENDER_SCRATCH_FLOATS 16384  /* size of ln3d_render_args.scalars (64 KB): decoder image + per-call range records */

typedef struct {
  const float* planes; int H, W;        /* channel-last tri-planes                                 */
  const int32_t* plane_index;           /* [V] which tri-plane each view renders                     */
  const float* cams;                    /* [V,25] cam2world(16) + intrinsics(9)                      */
  int V, res;                           /* rays per view = res*res                                   */
  const float* dec_w0; const float* dec_b0;   /* OSGDecoder FC 32->64 (raw weights; gain 1/sqrt(32) applied inside) */
  const float* dec_w1; const float* dec_b1;   /* FC 64->4 (gain 1/sqrt(64))                           */
  const float* jitter;                  /* [V, M, S] stratified jitter in [0,1)                      */
  const float* u_fine;                  /* [V*M, S] importance uniforms                              */
  float box_warp, bbox_min, bbox_max;
  int white_back;
  /* outputs */
  float* rgb;    /* [V,3,res,res] in [-1,1] */
  float* depth;  /* [V,1,res,res] */
  float* wsum;   /* [V,1,res,res] */
  /* scratch */
  float* ray_limits;  /* [V*M*2]  */
  float* scalars;     /* [LN3D_RENDER_SCRATCH_FLOATS]: batch-global min/max words + the decoder's MFMA fragment image */
  /* optional sampling-detail outputs (may be NULL): Triplane.forward's `shape_synthesized` dict, nsr/triplane.py:569-573 */
  float* coarse_sigma; /* [V,M,S] */
  float* fine_depths;  /* [V,M,S] */
  /* optional explicit rays [V,M,3] each (world units, unit directions): when non-NULL they replace the camera ray generation and
   * `cams` may be NULL - the seam of ImportanceRenderer.forward(planes, decoder, ray_origins, ray_directions, rendering_options),
   * nsr/volumetric_rendering/renderer.py:133 */
  const float* ray_o; const float* ray_d;
  float* fine_sigma;    /* [V,M,S]   (optional) */
  float* coarse_coords; /* [V,M,S,3] (optional) sample positions of the coarse pass */
  float* fine_coords;   /* [V,M,S,3] (optional) sample positions of the importance pass */
  /* The reference takes three reductions over everything ONE forward() call renders: the ray-limit fix-up (renderer.py:151-155)
   * and the depth clamp to [min, max] of all sample depths (ray_marcher.py:57-61).  views_per_call = how many consecutive views
   * form one such call: 0 (or >= V) = the whole launch is one call (Triplane.forward on a batch); 1 = every view is its own
   * call (the video drivers render one camera per call, nsr/train_util_diffusion.py:262-283).  At most
   * (LN3D_RENDER_SCRATCH_FLOATS - 2656) / 8 calls per launch. */
  int views_per_call;
  /* ---- ABI 9.  All zero = the behaviour of ABI 8 (Objaverse preset, res x res rays per view). */
  int rays_per_view;     /* M of an explicit ray list [V, M, 3] (any M >= 1, the seam's [N, M, 3] rays are not an image); 0 = res * res.
                          * rgb is written as [V, 3, M], depth / wsum / visibility as [V, M] */
  float* visibility;     /* optional [V, M]: T behind the last interval - 'visibility' of ImportanceRenderer.forward's dict (renderer.py:281,
                          * ray_marcher.py:44) */
  /* rendering presets of nsr/script_util.py:433-1000 other than Objaverse 64 + 64 'auto' (served by render_generic_kernel): */
  int depth_resolution, depth_resolution_importance;   /* samples per ray of the two passes, each <= 128 (presets: 48, 64, 80, 96, 128); 0 = 64.
                                                         * jitter is [V, M, depth_resolution], u_fine [V * M, depth_resolution_importance] and the
                                                         * optional sampling-detail outputs follow the same counts */
  int ray_mode;          /* 0: ray_start = ray_end = 'auto' (ray / AABB limits, renderer.py:145-156); 1: the numbers below (renderer.py:157-163,
                          * ShapeNet / FFHQ presets) */
  float ray_start, ray_end;
  int no_bbox_filter;    /* 1: rendering_options without 'filter_out_of_bbox' (renderer.py:343-352: plain _run_model) */
  /* return_meta outputs (renderer.py:283-300), all optional: the merged, depth-sorted per-sample tensors */
  float* weights;        /* [V, M, S + NI - 1]   'weights'        */
  float* all_coords;     /* [V, M, S + NI, 3]    'all_coords'     */
  float* feature_volume; /* [V, M, S + NI, 3]    'feature_volume' (the sorted colours) */
} ln3d_render_args;
/* Triplane.forward -> ImportanceRenderer.forward -> MipRayMarcher2 (nsr/triplane.py:505-750,
 * nsr/volumetric_rendering/renderer.py:133-307, ray_marcher.py:26-68, ray_sampler.py:262-331).  The Objaverse preset
 * (depth_resolution = depth_resolution_importance = 64, 'auto' limits, bbox filter: nsr/script_util.py:761-798) runs the
 * lane = sample kernel; the other presets and the return_meta outputs the generic one (same gather + MFMA decoder). */
int ln3d_render_triplane(const ln3d_render_args* a, void* stream);

/* triplane_decode_grid / forward_points (vit/vit_triplane.py:2009-2112): points f32 [P,3] -> sigma[P], rgb[P,3].
 * scalars: caller-owned scratch of >= 4096 floats (the decoder's fragment image is built into it) */
int ln3d_query_points(const float* planes, int H, int W, const float* points, int64_t P,
                      const float* dec_w0, const float* dec_b0, const float* dec_w1, const float* dec_b1,
                      float box_warp, float* sigma, float* rgb, float* scalars, void* stream);

/* ---------------------------------------------------------------- iso-surface of the sigma grid (mesh export)
 * Replaces mcubes.marching_cubes(sigma[G,G,G], thr) at nsr/train_util_diffusion.py:221 (PyMCubes, third-party, absent:
 * parity unpinned) with marching tetrahedra (ln3d_mesh_*) or classic marching cubes (ln3d_mcubes_*, the default of the drivers).  Pass 1: triangles per cell -> counts[(G-1)^3] (cell = (x*(G-1)+y)*(G-1)+z);
 * caller takes the inclusive prefix sum; pass 2 writes for triangle k its 3 vertices (grid coordinates) to
 * tri_pos[k*9..] and the ids of the grid edges they lie on to tri_key[k*3..] (for welding).                          */
int ln3d_mesh_count(const float* sigma, int G, float thr, int32_t* counts, void* stream);
/* Same two passes for CLASSIC marching cubes (Lorensen & Cline - the algorithm of mcubes.marching_cubes; 256-case table
 * csrc/mc_table.h, <= 5 triangles per cell, corner value > thr = inside).  Pinned against scikit-image's classic implementation
 * (tests/golden/mcubes_classic.npz); PyMCubes itself is absent. */
int ln3d_mcubes_count(const float* sigma, int G, float thr, int32_t* counts, void* stream);
int ln3d_mcubes_emit(const float* sigma, int G, float thr, const int64_t* offsets_inclusive, float* tri_pos, int64_t* tri_key,
                     void* stream);
int ln3d_mesh_emit(const float* sigma, int G, float thr, const int64_t* offsets_inclusive, float* tri_pos, int64_t* tri_key,
                   void* stream);

/* ---------------------------------------------------------------- conv decoder pieces (channel-last f32/bf16)
 * GroupNorm(32, eps 1e-6, affine) + optional swish over x f32 [N, HW, C] -> bf16 (ldm model.py:45-51)        */
#define LN3D_GN_PIXELS_PER_CHUNK 256
/* stats_scratch: [N*groups*2 * (1 + ceil(HW / LN3D_GN_PIXELS_PER_CHUNK))] floats - the sums, then one partial pair per pixel chunk
 * (reduced in chunk order: the result is bitwise reproducible) */
int ln3d_groupnorm_swish(const float* x, const float* w, const float* b, void* y_bf16, float* stats_scratch,
                         int N, int HW, int C, int groups, float eps, int swish, void* stream);
/* im2col for 3x3 pad 1 convs on channel-last bf16 [N,H,W,C] with optional nearest 2x upsample of the input
 * (ldm model.py:54-70): out bf16 [N*Ho*Wo, Kpad], column = (ky*3+kx)*C + c, zero padded to Kpad             */
int ln3d_im2col3x3(const void* x_bf16, void* col_bf16, int N, int H, int W, int C, int upsample, int Kpad, void* stream);

/* ---------------------------------------------------------------- U-Net denoiser pieces (ABI 9; csrc/unet_ops.hip)
 * The ShapeNet / FFHQ entry point's denoiser (guided_diffusion/unet.py:427-791: ResBlock, Downsample / Upsample, AttentionBlock,
 * ldm/modules/attention_compat.py SpatialTransformer) on channel-last activations [N, H*W, C]; convolutions and linears are
 * ln3d_gemm_bf16 calls (3x3 through ln3d_im2col3x3 / ln3d_im2col3x3_strided), the rest is here.
 * GroupNorm(groups, C) for any C % groups == 0 over x f32 [N, HW, C] -> bf16: y = act(GN(x + add_row[n]) * w + b), optionally
 * modulated per sample before the activation, t * (1 + mod_scale[n]) + mod_shift[n]  (ResBlock: `h + emb_out` :272-273 /
 * use_scale_shift_norm :267-271).  add_row / mod_* f32 [N, C] or NULL; swish = SiLU. */
int ln3d_groupnorm_any(const float* x, const float* add_row, const float* w, const float* b, const float* mod_scale, const float* mod_shift,
                       void* y_bf16, int N, int HW, int C, int groups, float eps, int swish, void* stream);
/* im2col for 3x3 pad 1 convs with a stride (Downsample.op, unet.py:150-153): out bf16 [N*Ho*Wo, Kpad], Ho = (H - 1) / stride + 1 */
int ln3d_im2col3x3_strided(const void* x_bf16, void* col_bf16, int N, int H, int W, int C, int stride, int Kpad, void* stream);
/* GEGLU (attention_compat.py:45-53): x f32 [rows, 2 * inner] = [a | gate] -> bf16 [rows, inner] = a * gelu_erf(gate) */
int ln3d_geglu(const float* x, void* y_bf16, int64_t rows, int inner, void* stream);
/* softmax(scale q k^T) v at any head size <= 256 over <= 1024 keys: q [B, Nq, ldq], k [B, Nk, ldk], v [B, Nk, ldv] bf16 token-major
 * (head h = columns [h * Dh, (h + 1) * Dh) of the given pointers), out bf16 [B, Nq, H * Dh]  (CrossAttention.forward,
 * attention_compat.py:179-202; QKVAttentionLegacy, unet.py:371-389) */
int ln3d_attention_small(const void* q, const void* k, const void* v, void* out, int B, int H, int Nq, int Nk, int Dh, int64_t ldq,
                         int64_t ldk, int64_t ldv, float scale, void* stream);
/* NCHW f32 -> channel-last bf16 [N, HW, Cpad] (channels >= C zero) and channel-last f32 [N, HW, C] -> NCHW f32 */
int ln3d_nchw_to_cl_bf16(const float* x, void* y_bf16, int N, int C, int HW, int Cpad, void* stream);
int ln3d_cl_to_nchw_f32(const float* x, float* y, int N, int C, int HW, void* stream);
/* LSGM mixed prediction of an eps model (continuous_diffusion_utils.py:748-754, gaussian_diffusion.py:336-348), in place on eps
 * (NCHW f32): eps <- (1 - s_c) * sqrt(1 - alpha_bar_t) * x + s_c * eps, s_c = sigmoid(mixing_logit[c]) */
int ln3d_mix_prediction(float* eps, const float* x, const float* mixing_logit, float sqrt_one_minus_ab, int N, int C, int HW, void* stream);

#ifdef __cplusplus
}
#endif
#endif
