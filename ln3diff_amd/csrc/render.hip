// Fused tri-plane volumetric renderer for gfx950 (one wavefront per ray).
//
// Replaces, in ONE launch per batch of views (plus two tiny pre/post passes for the reference's
// batch-global reductions): ray generation, ray/AABB limits, stratified sampling, tri-plane bilinear
// gather (F.grid_sample x3), OSGDecoder MLP, MipRayMarcher2 compositing, importance sampling
// (max/avg pool, cdf, searchsorted), the coarse+fine merge (torch.sort + gathers) and the final
// compositing.  The reference materialises [V,3,M*S,32] features (1.6 GB per 256^2 view-pass) and
// ~6 GB of HBM traffic per view; here nothing but the tri-plane texels and 5 floats per ray touch HBM.
//
// Wave mapping (64 lanes = the 64 coarse / 64 fine samples of one ray):
//  * gather phase : 8 lanes per sample point, lane (g = lane>>3, c4 = lane&7) fetches channels 4c4..4c4+3
//    of the 12 taps of point 8*it+g from CHANNEL-LAST planes [3][H][W][32] -> every wave-level load
//    instruction touches 8 fully-used 128-B texels (coalesced), instead of 64 partially-used lines.  The point's tap offsets /
//    weights (computed once by its owner lane) reach the 8 lanes through the wave's LDS tile (r6), the loads take a scalar base.
//    The interpolated 32-vector goes to a per-wave LDS tile as TWO bf16 rows (hi = truncated fp32, lo = bf16(f - hi)),
//    128 bytes per point, 16-byte chunks XOR-swizzled ...
//  * MLP phase    : ... which is the B operand of v_mfma_f32_32x32x16_bf16: the 32 -> 64 -> 4 decoder of the wave's 64 points
//    runs on the matrix pipe (2 x (12 + 12) MFMAs per pass instead of 2 304 scalar-operand FMAs per lane).  Products are
//    split hi*hi + hi*lo + lo*hi (error ~2^-16 relative: the fp32 parity of the renderer, 1e-5, is kept); weights are split
//    once per launch into A fragments (render_init_kernel) and staged in the workgroup's LDS; the hidden bias is the C
//    operand of the first MFMA; softplus runs in the log2 domain (log2 e folded into layer 1, ln 2 into layer 2).
//  * compositing  : wavefront-level: neighbours by DPP/shuffle, transmittance by a wave prefix product,
//    cdf by a wave prefix sum, searchsorted by a 6-step binary search on the per-wave LDS copy of the
//    cdf, the coarse+fine merge by rank counting (no sort), final sums by wave reductions.
#include <stdlib.h>
#include "common.h"
#include "../../include/ln3d.h"

#define NS 64            // samples per pass (coarse == fine == 64, Objaverse preset)
#define WAVE_LDS_BYTES 8192           // per wave: 64 points x (64 B hi + 64 B lo) feature rows; reused for cdf / merge arrays
#define WAVE_LDS_FLOATS (WAVE_LDS_BYTES / 4)
// scalars[]: [0..15] unused, then the decoder image that every workgroup copies into its LDS:
//   A1  : 8 fragments (split s, hidden tile jt, k-step ks) x 64 lanes x 16 B           at DEC_A1  (8 KB)
//   A2  : rows 0-3 only: [split s][k-step s2][hi][output o] x 16 B (other rows are zero)  at DEC_A2  (1 KB)
//   Z   : 1 KB of zeros (the A2 fragments of the lanes whose row is >= 4, same immediates)  at DEC_Z
//   B0  : hidden bias as C fragments [jt][hi][16] f32 (pre-multiplied by log2 e)         at DEC_B0  (256 B)
//   B1  : output bias [4] f32                                                            at DEC_B1  (16 B)
#define DEC_OFF 16
#define DEC_A1 0
#define DEC_A2 8192
#define DEC_Z (8192 + 1024)
#define DEC_B0 (8192 + 1024 + 1024)
#define DEC_B1 (DEC_B0 + 256)
#define DEC_BYTES (DEC_B1 + 16)
#define DEC_FLOATS (DEC_BYTES / 4)
// then one 8-word record per "reference call" (group of views_per_call consecutive views): [0] min ray start, [1] max ray start,
// [2] min depth, [3] max depth, [4] any-valid-ray flag.  The reference takes these over whatever one forward() call renders
// (renderer.py:151-155, ray_marcher.py:57-61); its video drivers call once per camera, Triplane.forward callers once per batch.
#define GRP_OFF (DEC_OFF + DEC_FLOATS)
#define GRP_WORDS 8
#define MAX_GROUPS ((LN3D_RENDER_SCRATCH_FLOATS - GRP_OFF) / GRP_WORDS)
static_assert(MAX_GROUPS >= 1024, "room for the per-call range records");

// 4 wavefronts per workgroup, 3 per SIMD.  r6 re-measured 8-wave workgroups at 4 per SIMD (-DRENDER_WPB=8 -DRENDER_OCC=4): with the r6 gather and
// the decoder behind the gather the kernel fits 128 VGPRs with 6 spilled loop invariants (r4: 28 spills, 8 % slower) and the isolated loop gains
// 2 - 4 % (0.630 -> 0.62 ms per 256^2 view, 2.37 -> 2.25 at 512^2) - but inside the pipelines it is level on configs[1] (+0.2 %) and 0.8 % BEHIND on
// configs[2] (768 views of 32 plane sets per launch), so the no-spill form ships (profiles/r6_render_occ4.log, r6_render_insitu.log).
#ifndef RENDER_OCC
#define RENDER_OCC 3
#endif
#ifndef RENDER_WPB
#define RENDER_WPB 4          // wavefronts (= rays in flight) per workgroup of render_kernel; they share one LDS copy of the decoder image
#endif
#ifndef LN3D_RENDER_ABL   // bench-only ablations (tools/render_bench.hip): 1 = no decoder MLP, 2 = no texel loads, 4 = no compositing
#define LN3D_RENDER_ABL 0
#endif

__device__ __forceinline__ uint32_t enc_f(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec_f(uint32_t u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ float softplus_fast(float x) {   // torch softplus(beta=1, threshold=20)
  return x > 20.0f ? x : __logf(1.0f + __expf(x));
}

struct RenderP {
  const float* planes; int H, W;
  const int32_t* plane_index; const float* cams; int V, res;
  const float* jitter; const float* u_fine;
  float coord_scale, bbox_min, bbox_max; int white_back;
  float* rgb; float* depth; float* wsum;
  float* ray_limits; uint32_t* scal_u; const float* dec; int vpc;    // scal_u -> the group records
  float* coarse_sigma; float* fine_depths;
  const float* ray_o; const float* ray_d;          // optional explicit rays [V, M, 3] (ImportanceRenderer.forward seam)
  float* fine_sigma; float* coarse_coords; float* fine_coords;
  int M;                                           // rays per view (res * res, or the length of an explicit ray list)
  float* vis;                                      // optional [V, M]: transmittance behind the last interval (ray_marcher.py:44)
  // ---- generic kernel only (render_generic_kernel): the presets other than Objaverse 64 + 64 'auto', and the return_meta outputs
  int S, NI;                                       // coarse / importance samples per ray (<= 128 each)
  int numeric; float t_start, t_end;               // ray_start / ray_end given as numbers (renderer.py:157-163) instead of 'auto'
  float* w_all; float* coords_all; float* colors_all;   // optional [V,M,S+NI-1], [V,M,S+NI,3], [V,M,S+NI,3] (renderer.py:283-300)
};

// ------------------------------------------------------------------ ray generation + AABB limits
__device__ __forceinline__ void make_ray(const float* cam, int res, int pix, float o[3], float d[3]) {
  const float inv_res = 1.0f / (float)res, half = 0.5f / (float)res;
  const int i = pix / res, j = pix - i * res;
  const float x_cam = (float)j * inv_res + half, y_cam = (float)i * inv_res + half;
  const float fx = cam[16], sk = cam[17], cx = cam[18], fy = cam[20], cy = cam[21];
  const float xl = (x_cam - cx + cy * sk / fy - sk * y_cam / fy) / fx;
  const float yl = (y_cam - cy) / fy;
  float w[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    w[r] = cam[4 * r + 0] * xl + cam[4 * r + 1] * yl + cam[4 * r + 2] + cam[4 * r + 3];
    o[r] = cam[4 * r + 3];
    d[r] = w[r] - o[r];
  }
  const float nrm = fmaxf(sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), 1e-12f);
  d[0] /= nrm; d[1] /= nrm; d[2] /= nrm;
}

__device__ __forceinline__ void ray_box(const float o[3], const float d[3], float half, float& tmin_o, float& tmax_o) {
  float inv[3]; int sg[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) { inv[a] = 1.0f / d[a]; sg[a] = inv[a] < 0.f; }
  auto bnd = [&](int s) { return s ? half : -half; };
  float tmin = (bnd(sg[0]) - o[0]) * inv[0], tmax = (bnd(1 - sg[0]) - o[0]) * inv[0];
  const float tymin = (bnd(sg[1]) - o[1]) * inv[1], tymax = (bnd(1 - sg[1]) - o[1]) * inv[1];
  bool valid = !(tmin > tymax || tymin > tmax);
  tmin = fmaxf(tmin, tymin); tmax = fminf(tmax, tymax);
  const float tzmin = (bnd(sg[2]) - o[2]) * inv[2], tzmax = (bnd(1 - sg[2]) - o[2]) * inv[2];
  valid = valid && !(tmin > tzmax || tzmin > tmax);
  tmin = fmaxf(tmin, tzmin); tmax = fminf(tmax, tzmax);
  tmin_o = valid ? tmin : -1.0f;
  tmax_o = valid ? tmax : -2.0f;
}

// r5: the ray's origin / direction leave make_ray as six separate scalar registers.  With hipcc's SLP vectoriser on, ONE tree that keeps
// (o0, o1) / (d0, d1) as 64-bit register tuples across the per-ray loop of the marcher (7 v_pk_* instructions of the ray generation + the
// fine position's v_pk_fma) made 3 - 7 rays of 262 144 differ from launch to launch, always in lanes 48 - 63 of the coarse pass
// (profiles/r4_render_spill.md; not root-caused - single-instruction hazards, modifiers, waits and scratch are excluded).  The library is built
// with -fno-slp-vectorize; this fence additionally makes the tuple impossible whatever the flags (it is what turned that build green
// in the bisect), and tests/test_render_gpu.py sweeps 100 scenes for bitwise repeatability.
#define RAY_FENCE(o_, d_) asm volatile("" : "+v"((o_)[0]), "+v"((o_)[1]), "+v"((o_)[2]), "+v"((d_)[0]), "+v"((d_)[1]), "+v"((d_)[2]))

// hidden index that accumulator register r of lane-half hi holds in hidden tile jt (MFMA 32x32 C/D layout)
__device__ __forceinline__ int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
// hi = fp32 truncated to bf16 (exact), lo = bf16(f - hi) (RNE): f ~ hi + lo to 2^-16 relative
__device__ __forceinline__ void split_bf16(float f, bf16_t& h, bf16_t& l) {
  const uint32_t u = __float_as_uint(f) & 0xffff0000u;
  h = (bf16_t)(u >> 16);
  l = f2bf(f - __uint_as_float(u));
}

// One launch per render call: resets the batch-global words and builds the decoder's LDS image (see DEC_* above).
//   FullyConnectedLayer weight_gain 1/sqrt(fan_in) (nsr/networks_stylegan2.py:122-157) is applied here; layer 1 (and its bias)
//   are pre-multiplied by log2(e) and layer 2 by ln(2): softplus(x) = ln2 * log2(1 + 2^(x log2 e)).
__global__ void render_init_kernel(uint32_t* scal_u, int groups, float* dec, const float* w0, const float* b0, const float* w1, const float* b1) {
  const int t = threadIdx.x + blockIdx.x * blockDim.x, nt = blockDim.x * gridDim.x;
  for (int g = t; g < groups; g += nt) {
    uint32_t* r = scal_u + g * GRP_WORDS;
    r[0] = 0xffffffffu; r[1] = 0u; r[2] = 0xffffffffu; r[3] = 0u; r[4] = 0u;
  }
  const float g0 = 1.0f / sqrtf(32.0f) * 1.4426950408889634f, g1 = 1.0f / sqrtf(64.0f) * 0.6931471805599453f;
  char* img = reinterpret_cast<char*>(dec);
  // A1 fragment (s, jt, ks), lane (row = l31, hi): W0[jt*32 + l31][ks*16 + 8*hi + e], e = 0..7
  for (int i = t; i < 8 * 64 * 8; i += nt) {
    const int e = i & 7, lane = (i >> 3) & 63, f = i >> 9;
    const int sp = f >> 2, jt = (f >> 1) & 1, ks = f & 1, l31 = lane & 31, hi = lane >> 5;
    bf16_t h, l;
    split_bf16(w0[(jt * 32 + l31) * 32 + ks * 16 + 8 * hi + e] * g0, h, l);
    reinterpret_cast<bf16_t*>(img + DEC_A1)[i] = sp ? l : h;
  }
  // A2 [s][s2][hi][o][e]: W1[o][hidden of k-slot (hi, e) in k-step s2] = the hidden index whose softplus sits in accumulator
  // register 8*(s2&1)+e of hidden tile s2>>1 of that lane-half
  for (int i = t; i < 2 * 4 * 2 * 4 * 8; i += nt) {
    const int e = i & 7, o = (i >> 3) & 3, hi = (i >> 5) & 1, s2 = (i >> 6) & 3, sp = i >> 8;
    const int hid = (s2 >> 1) * 32 + acc_row(8 * (s2 & 1) + e, hi);
    bf16_t h, l;
    split_bf16(w1[o * 64 + hid] * g1, h, l);
    reinterpret_cast<bf16_t*>(img + DEC_A2)[i] = sp ? l : h;
  }
  for (int i = t; i < 256; i += nt) reinterpret_cast<uint32_t*>(img + DEC_Z)[i] = 0u;
  for (int i = t; i < 2 * 2 * 16; i += nt) {
    const int r = i & 15, hi = (i >> 4) & 1, jt = i >> 5;
    reinterpret_cast<float*>(img + DEC_B0)[i] = b0[jt * 32 + acc_row(r, hi)] * 1.4426950408889634f;
  }
  for (int i = t; i < 4; i += nt) reinterpret_cast<float*>(img + DEC_B1)[i] = b1[i];
}

__global__ __launch_bounds__(256) void ray_limits_kernel(RenderP p, float box_half) {
  const int M = p.M;
  const int64_t nr = (int64_t)p.V * M;
  const int64_t ray = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float tmn = 3.0e38f, tmx = -3.0e38f; int any = 0, grp = -1;
  if (ray < nr) {
    const int v = (int)(ray / M), pix = (int)(ray % M);
    grp = v / p.vpc;
    float o[3], d[3], a, b;
    if (p.ray_o) {
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) { o[ax] = p.ray_o[3 * ray + ax]; d[ax] = p.ray_d[3 * ray + ax]; }
    } else {
      make_ray(p.cams + 25 * v, p.res, pix, o, d);
    }
    ray_box(o, d, box_half, a, b);
    p.ray_limits[2 * ray] = a; p.ray_limits[2 * ray + 1] = b;
    if (b > a) { tmn = a; tmx = a; any = 1; }
  }
  const int g0 = __shfl(grp, 0);
  if (__all(grp == g0 || grp < 0) && g0 >= 0) {      // the usual case: the wave's 64 rays belong to one call
    tmn = wave_min(tmn); tmx = wave_max(tmx);
    any = __any(any);
    if ((threadIdx.x & 63) == 0 && any) {
      uint32_t* r = p.scal_u + (int64_t)g0 * GRP_WORDS;
      atomicMin(&r[0], enc_f(tmn)); atomicMax(&r[1], enc_f(tmx)); atomicOr(&r[4], 1u);
    }
  } else if (any) {                                   // a wave straddling two calls (res^2 not a multiple of 64)
    uint32_t* r = p.scal_u + (int64_t)grp * GRP_WORDS;
    atomicMin(&r[0], enc_f(tmn)); atomicMax(&r[1], enc_f(tmx)); atomicOr(&r[4], 1u);
  }
}

// ------------------------------------------------------------------ gather + decoder for the wave's 64 points
// in : this lane's point (px,py,pz) in world units.  out: rgb[3], sigma (bbox filter applied).
// wl = the wave's 8 KB of LDS, cimg = the workgroup's copy of the decoder image (DEC_*).
typedef __attribute__((address_space(3))) const bf16x8 lds_bfrag_t;
__device__ __forceinline__ float softplus_log2(float x) {   // log2(1 + 2^x); x carries log2(e), the caller's next layer ln(2)
  const float y = __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(x));
  // torch's threshold (x > 20 -> x) is only NEEDED where 2^x overflows: below it y >= x holds to the last ulp or two and for
  // x >= 25 fl(1 + 2^x) = 2^x, so y is x already.  median(y, x, 128) = y while x <= y <= 128 and x once y = +inf: one instruction
  // instead of a compare + select, 128 times per ray (r6)
  return __builtin_amdgcn_fmed3f(y, x, 128.0f);
}

// torch softplus (beta 1, threshold 20) and exp(-t) of the ray-marcher on the transcendental unit (r4).  The libm forms
// (log1pf(expf(x)), expf) were ~130 + ~25 dependent vector instructions each, three times per ray and lane, all on the per-ray
// critical path.  softplus(x) = max(x, 0) + log1p(e), e = exp(-|x|); log1p with Kahan's correction for the rounding of 1 + e
// (log(u) * e / (u - 1), u = fl(1 + e)): ~2 ulp, the images move by <= 1e-6 against the fp32 oracle (tests/test_render_gpu.py).
__device__ __forceinline__ float softplus20_hw(float x) {
  const float e = __builtin_amdgcn_exp2f(-fabsf(x) * 1.4426950408889634f);
  const float u = 1.0f + e, d = u - 1.0f;
  const float l = __builtin_amdgcn_logf(u) * 0.6931471805599453f;
  const float l1p = d == 0.f ? e : l * (e * __builtin_amdgcn_rcpf(d));
  return x > 20.0f ? x : fmaxf(x, 0.f) + l1p;
}
__device__ __forceinline__ float exp_neg_hw(float t) { return __builtin_amdgcn_exp2f(t * -1.4426950408889634f); }   // exp(-t)

// -DLN3D_RENDER_OPSEL_REPRO (tools/r6_opsel_repro.sh, never the library): without the copy - the reproducer of profiles/r6_render_opsel.md
__device__ __forceinline__ float even_reg(float x) {
#ifdef LN3D_RENDER_OPSEL_REPRO
  return x;
#else
  float r;
  asm("v_mov_b32 %0, %1" : "=v"(r) : "v"(x));
  return r;
#endif
}

__device__ __forceinline__ void shade64(const RenderP& p, const float* __restrict__ planes, char* wl, const char* cimg,
                                        float px, float py, float pz, int lane, float rgb[3], float& sigma) {
  const int g = lane >> 3, c4 = lane & 7;
  const bool inb = px >= p.bbox_min && px <= p.bbox_max && py >= p.bbox_min && py <= p.bbox_max && pz >= p.bbox_min &&
                   pz <= p.bbox_max;
  const float sg_fill = -3.4028234663852886e38f / 3.0f;   // nan_to_num(-inf) / SAFE_GUARD
  // r3: a ray that misses the box keeps ALL of its 64 points outside it (its depth range is the call-wide fallback), and every
  // one of them gets the fill values below whatever the planes hold: skip the gather and the decoder for the whole wave
  // (wave-uniform branch; 15-40 % of the rays of an orbit view at the reference's camera distance).  Same outputs, bit for bit.
  if (__builtin_amdgcn_ballot_w64(inb) == 0ull) {
    sigma = sg_fill; rgb[0] = 0.f; rgb[1] = 0.f; rgb[2] = 0.f;
    return;
  }
  const float sx = px * p.coord_scale, sy = py * p.coord_scale, sz = pz * p.coord_scale;
  const int plane_stride = p.H * p.W * 32;
  // ---- bilinear setup, ONCE per point (this lane's own point): per plane 4 clamped tap offsets (bytes, channel 0 of the texel)
  // and 4 tap weights with the zero-padding mask folded in.  The gather below has 8 lanes cooperate on one point's texels (one
  // 128-B line per tap per point); they fetch the owner's setup (through the LDS tile, below; r2 - r5: ds_bpermute) instead of recomputing it
  // 8 times - the address / weight arithmetic was 200 of the 252 VALU instructions of a gather iteration (profiles/r2_render_pmc.md).
  int toff[12];
  float tw[12];
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    const float gx = pl == 0 ? sx : (pl == 1 ? sy : sz);     // (x,y) (y,z) (z,x)
    const float gy = pl == 0 ? sy : (pl == 1 ? sz : sx);
    const float ix = ((gx + 1.f) * p.W - 1.f) * 0.5f, iy = ((gy + 1.f) * p.H - 1.f) * 0.5f;
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0;
    const float w_nw = (fx0 + 1.f - ix) * (fy0 + 1.f - iy), w_ne = (ix - fx0) * (fy0 + 1.f - iy);
    const float w_sw = (fx0 + 1.f - ix) * (iy - fy0), w_se = (ix - fx0) * (iy - fy0);
    // branch-free taps: out-of-range texels are read at a clamped (valid) address and weighted by 0, so the 12 loads of an
    // iteration are issued back to back (conditional loads cost one L2 round trip each: the compiler waits per branch)
    const bool xin0 = x0 >= 0 && x0 < p.W, xin1 = x0 + 1 >= 0 && x0 + 1 < p.W;
    const bool yin0 = y0 >= 0 && y0 < p.H, yin1 = y0 + 1 >= 0 && y0 + 1 < p.H;
    const int xc0 = min(max(x0, 0), p.W - 1), xc1 = min(max(x0 + 1, 0), p.W - 1);
    const int yc0 = min(max(y0, 0), p.H - 1), yc1 = min(max(y0 + 1, 0), p.H - 1);
    tw[4 * pl + 0] = (xin0 && yin0) ? w_nw : 0.f; tw[4 * pl + 1] = (xin1 && yin0) ? w_ne : 0.f;
    tw[4 * pl + 2] = (xin0 && yin1) ? w_sw : 0.f; tw[4 * pl + 3] = (xin1 && yin1) ? w_se : 0.f;
    const int pb = pl * plane_stride;
    toff[4 * pl + 0] = (pb + (yc0 * p.W + xc0) * 32) * 4; toff[4 * pl + 1] = (pb + (yc0 * p.W + xc1) * 32) * 4;
    toff[4 * pl + 2] = (pb + (yc1 * p.W + xc0) * 32) * 4; toff[4 * pl + 3] = (pb + (yc1 * p.W + xc1) * 32) * 4;
  }
  // ---- r6: the setup hand-off goes through the wave's feature tile instead of 24 ds_bpermute per 8-point iteration.  Row s of the tile
  // (128 B, free until the features of point s land in it - in the very iteration that consumes its setup, behind the reads: a
  // wave's LDS operations execute in program order) takes point s's 12 tap offsets + 12 tap weights as six 16-byte chunks, chunk c at
  // position c ^ (s & 7): the 8 rows a gather iteration reads (one per lane group g, every lane of the group the same address =
  // broadcast) sit in 8 different chunk positions, so both the b128 writes and the b128 reads are conflict-free.  6 + 8 x 6 LDS
  // instructions per pass instead of 8 x 24, and the 24 setup registers are not live across the loop any more.
  const uint32_t wb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)wl;
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) i32x4 lds_i4;
  typedef __attribute__((address_space(3))) f32x4 lds_f4;
  {
    lds_i4* const row = reinterpret_cast<lds_i4*>((uintptr_t)wb) + lane * 8;       // 16-byte chunk units
    const int k7 = lane & 7;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      row[c ^ k7] = i32x4{toff[4 * c], toff[4 * c + 1], toff[4 * c + 2], toff[4 * c + 3]};
      reinterpret_cast<lds_f4*>(row)[(3 + c) ^ k7] = f32x4{tw[4 * c], tw[4 * c + 1], tw[4 * c + 2], tw[4 * c + 3]};
    }
  }
  wave_sync();
  // the plane base as a scalar pair + a 32-bit per-lane byte offset: global_load_dwordx4 v, v_off, s[base] (no 64-bit address arithmetic
  // per tap: it was 24 of the ~90 vector instructions of an iteration).  planes is wave-uniform (one ray = one wave).
  typedef __attribute__((address_space(1))) const char gchar_t;
  const uint64_t pl64 = reinterpret_cast<uint64_t>(planes);
  gchar_t* const pbase = reinterpret_cast<gchar_t*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pl64 >> 32)) << 32) |
                                                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pl64));
  const uint32_t c4b = c4 * 16;
  const lds_i4* const srow = reinterpret_cast<const lds_i4*>((uintptr_t)wb) + g * 8;      // this lane group's setup row of iteration 0
  auto fetch = [&](int it, float4 (&t)[12]) {
    int off[12];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const i32x4 o4 = srow[it * 64 + (c ^ g)];
      off[4 * c] = o4.x; off[4 * c + 1] = o4.y; off[4 * c + 2] = o4.z; off[4 * c + 3] = o4.w;
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      if constexpr (!(LN3D_RENDER_ABL & 2))
      {
        const f32x4 tv = *reinterpret_cast<__attribute__((address_space(1))) const f32x4*>(pbase + ((uint32_t)off[k] + c4b));
        t[k] = make_float4(tv.x, tv.y, tv.z, tv.w);
      }
      else t[k] = make_float4((float)k, (float)off[k], 1.f, 2.f);
    }
  };
  // the tap weights are read where they are used (behind whatever was placed under the loads): 12 registers less across that code
  char* const wrow = wl + g * 128 + (c4 & 1) * 8 + (((c4 >> 1) ^ (g >> 1)) << 4);
  auto reduce = [&](int it, const float4 (&t)[12]) {
    float a[12];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const f32x4 w4 = reinterpret_cast<const lds_f4*>(srow)[it * 64 + ((3 + c) ^ g)];
      // .y / .w arrive in ODD registers.  Broadcasting an odd register into a packed-fp32 source needs op_sel = 1 (the low result lane
      // reads the high half of the pair), and that form is NOT reliable on gfx950: lanes 48 - 63 come out wrong from time to time as soon
      // as a second wave shares the SIMD (profiles/r6_render_opsel.md - the root cause of the r3 - r6 "SLP irreproducibility").  A copy
      // into a fresh register lets hipcc place the value in the low half of a pair (op_sel_hi-only forms, like every other packed
      // instruction of the library); tools/check_isa.py rejects a build that contains the op_sel form anywhere.
      a[4 * c] = w4.x; a[4 * c + 1] = even_reg(w4.y); a[4 * c + 2] = w4.z; a[4 * c + 3] = even_reg(w4.w);
    }
    // explicit packed fp32 (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32, r3): a texel's float4 is two even-aligned register pairs and
    // the tap weight is broadcast by op_sel, so the 48 + 12 + 4 scalar operations of an iteration are 24 + 6 + 2 packed ones; per
    // channel the evaluation order is the reference's grid_sample order ((nw + ne) + sw) + se, then the plane sum, unchanged
    ln3d_f32x2 accA = {0.f, 0.f}, accB = {0.f, 0.f};
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const ln3d_f32x2 w0 = {a[4 * pl], a[4 * pl]};
      ln3d_f32x2 sA = ln3d_f32x2{t[4 * pl].x, t[4 * pl].y} * w0, sB = ln3d_f32x2{t[4 * pl].z, t[4 * pl].w} * w0;
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        const ln3d_f32x2 wk = {a[4 * pl + k], a[4 * pl + k]};
        sA = __builtin_elementwise_fma(ln3d_f32x2{t[4 * pl + k].x, t[4 * pl + k].y}, wk, sA);
        sB = __builtin_elementwise_fma(ln3d_f32x2{t[4 * pl + k].z, t[4 * pl + k].w}, wk, sB);
      }
      accA += sA; accB += sB;
    }
    // mean over the 3 planes.  torch divides (sum / 3); a multiply by the fp32 reciprocal differs by <= 1 ulp - far inside the
    // renderer's 1e-5 parity - and saves four IEEE division sequences (~40 VALU) per iteration
    const float third = 0.333333343267440796f;
    accA *= third; accB *= third;
    float4 acc;
    acc.x = accA.x; acc.y = accA.y; acc.z = accB.x; acc.w = accB.y;
    // feature row of point src: chunks 0-3 = bf16 hi of channels 0-31, chunks 4-7 = bf16 lo; 16-byte chunk c at c ^ ((src >> 1) & 7)
    const uint32_t u0 = __float_as_uint(acc.x) & 0xffff0000u, u1 = __float_as_uint(acc.y) & 0xffff0000u;
    const uint32_t u2 = __float_as_uint(acc.z) & 0xffff0000u, u3 = __float_as_uint(acc.w) & 0xffff0000u;
    uint2 hv, lv;
    hv.x = (u0 >> 16) | u1; hv.y = (u2 >> 16) | u3;
    lv.x = pack2bf(acc.x - __uint_as_float(u0), acc.y - __uint_as_float(u1));
    lv.y = pack2bf(acc.z - __uint_as_float(u2), acc.w - __uint_as_float(u3));
    // key = (src >> 1) & 7 = (g >> 1) + 4 * (it & 1): the hi half sits at chunk p = (c4 >> 1) ^ (g >> 1) on even iterations and p + 4 on odd
    // ones, the lo half at the other - ONE lane-constant address + immediates for the whole pass
    *reinterpret_cast<uint2*>(wrow + it * 1024 + ((it & 1) ? 64 : 0)) = hv;
    *reinterpret_cast<uint2*>(wrow + it * 1024 + ((it & 1) ? 0 : 64)) = lv;
  };
  // ---- the 32 -> 64 -> 4 decoder of one 32-point tile in four pieces (hidden tile jt = 0, 1: layer 1 + softplus, then its two k-steps of
  // layer 2): the LN3D_RENDER_SEQ=0 bench build places the pieces of point tile 0 between the load issue and the load use of gather
  // iterations 4 - 7; the shipped build runs them back to back behind the gather
  const int l31 = lane & 31, hi = lane >> 5;
  const uint32_t cb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)cimg;
  auto mlp_l1 = [&](int pt, int jt, f32x16& hacc) {
    const int prow = pt * 32 + l31, key = (prow >> 1) & 7;
    // hidden tile jt: 32 hidden units x the 32 points, bias as the C operand
    const float4* bp = reinterpret_cast<const float4*>(cimg + DEC_B0 + (jt * 2 + hi) * 64);
#pragma unroll
    for (int q = 0; q < 4; ++q) { const float4 b = bp[q]; hacc[4 * q] = b.x; hacc[4 * q + 1] = b.y; hacc[4 * q + 2] = b.z; hacc[4 * q + 3] = b.w; }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 bh = *(lds_bfrag_t*)(uintptr_t)(wb + prow * 128 + (((2 * ks + hi) ^ key) << 4));
      const bf16x8 bl = *(lds_bfrag_t*)(uintptr_t)(wb + prow * 128 + (((4 + 2 * ks + hi) ^ key) << 4));
      const bf16x8 ah = *(lds_bfrag_t*)(uintptr_t)(cb + DEC_A1 + ((0 * 2 + jt) * 2 + ks) * 1024 + lane * 16);
      const bf16x8 al = *(lds_bfrag_t*)(uintptr_t)(cb + DEC_A1 + ((1 * 2 + jt) * 2 + ks) * 1024 + lane * 16);
      hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, hacc, 0, 0, 0);
      hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, hacc, 0, 0, 0);
      hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, hacc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) hacc[r] = softplus_log2(hacc[r]);
  };
  // layer-2 A fragment (split s, k-step s2) of lane (row l31 < 4, hi) at DEC_A2 + (s * 4 + s2) * 128 + hi * 64 + l31 * 16; the lanes of the
  // 28 padding rows read zeros at the same immediates from DEC_Z (1 KB of zeros): one lane-constant address for all eight fragments
  const uint32_t a2base = l31 < 4 ? cb + DEC_A2 + hi * 64 + l31 * 16 : cb + DEC_Z;
  auto mlp_l2 = [&](int jt, const f32x16& hacc, f32x16& oacc) {
    if (jt == 0) {                                     // the output accumulator starts here (an inline-zero C operand), not a gather iteration earlier
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
    }
#pragma unroll
    for (int sh = 0; sh < 2; ++sh) {
      const int s2 = 2 * jt + sh;
      union { uint32_t u[4]; bf16x8 v; } ph, pl_;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float f0 = hacc[8 * sh + 2 * e], f1 = hacc[8 * sh + 2 * e + 1];
        const uint32_t w0 = __float_as_uint(f0) & 0xffff0000u, w1 = __float_as_uint(f1) & 0xffff0000u;
        ph.u[e] = (w0 >> 16) | w1;
        pl_.u[e] = pack2bf(f0 - __uint_as_float(w0), f1 - __uint_as_float(w1));
      }
      const bf16x8 ah = *(lds_bfrag_t*)(uintptr_t)(a2base + (0 * 4 + s2) * 128);
      const bf16x8 al = *(lds_bfrag_t*)(uintptr_t)(a2base + (1 * 4 + s2) * 128);
      oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ph.v, oacc, 0, 0, 0);
      oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, pl_.v, oacc, 0, 0, 0);
      oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, ph.v, oacc, 0, 0, 0);
    }
  };
  float o[4] = {0.f, 0.f, 0.f, 0.f};
  // rows 0-3 = (sigma, r, g, b) of point pt*32 + l31, in registers 0-3 of lanes 0-31
  auto mlp_out = [&](int pt, const f32x16& oacc) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // point tile 0 lives in the lanes that hold its results already; tile 1's results move up by 32 lanes
      const float v = pt == 0 ? oacc[k] : __shfl(oacc[k], l31, 64);
      if (hi == pt) o[k] = v;
    }
  };
  constexpr bool kMlp = !(LN3D_RENDER_ABL & 1);
  float4 tA[12];
  // one iteration (8 points) at a time: double-buffering the 12 loads measured slower at 2 waves/SIMD (r2: 0.863 vs 0.838 ms per 256^2 view),
  // and r6's ablations say why more loads in flight do not help: the texel loads are an L1-path THROUGHPUT term (profiles/r6_render_abl.log)
  // LN3D_RENDER_SEQ 1 (shipped): the decoder of both point tiles behind the whole gather.  0 (bench builds) = the decoder of tile 0 in four
  // pieces under the texel loads of gather iterations 4 - 7: built first this round, bit-identical, and measured NO faster - isolated 0.636 vs
  // 0.630 ms per 256^2 view, 0.212 vs 0.198 at 128^2, 2.34 vs 2.37 at 512^2; in the pipelines configs[1] level, configs[2] 0.5 % behind - at 23 more
  // VGPRs: the decoder is 0.05 ms of the kernel and the texel loads it would hide under are an L1-path throughput term
  // (profiles/r6_render_abl.log, r6_render_insitu.log).
#ifndef LN3D_RENDER_SEQ
#define LN3D_RENDER_SEQ 1
#endif
#pragma unroll 1
  for (int it = 0; it < (LN3D_RENDER_SEQ ? 8 : 4); ++it) {
    fetch(it, tA);
    reduce(it, tA);
  }
  wave_sync();
  // LN3D_RENDER_SEQ 0 only: points 0 - 31 are complete and their decoder runs UNDER the texel loads of points 32 - 63 (a piece per gather
  // iteration, between the issue of the iteration's 12 loads and their first use).  Same arithmetic in the same order per point: bit-identical.
  f32x16 hacc, oacc;
#define SB0_ __builtin_amdgcn_sched_barrier(0)
  if constexpr (LN3D_RENDER_SEQ) {
    if constexpr (kMlp) { mlp_l1(0, 0, hacc); mlp_l2(0, hacc, oacc); SB0_; mlp_l1(0, 1, hacc); mlp_l2(1, hacc, oacc); mlp_out(0, oacc); }
  } else {
    fetch(4, tA); SB0_; if constexpr (kMlp) mlp_l1(0, 0, hacc); SB0_; reduce(4, tA); SB0_;
    fetch(5, tA); SB0_; if constexpr (kMlp) mlp_l2(0, hacc, oacc); SB0_; reduce(5, tA); SB0_;
    fetch(6, tA); SB0_; if constexpr (kMlp) mlp_l1(0, 1, hacc); SB0_; reduce(6, tA); SB0_;
    fetch(7, tA); SB0_; if constexpr (kMlp) mlp_l2(1, hacc, oacc); SB0_; reduce(7, tA); SB0_;
    if constexpr (kMlp) mlp_out(0, oacc);
    wave_sync();
  }
  if constexpr (kMlp) {
    mlp_l1(1, 0, hacc); mlp_l2(0, hacc, oacc);
    SB0_;                                              // keep the two hidden tiles apart: one set of fragments live at a time
    mlp_l1(1, 1, hacc); mlp_l2(1, hacc, oacc);
    mlp_out(1, oacc);
    const float4 b1 = *reinterpret_cast<const float4*>(cimg + DEC_B1);
    o[0] += b1.x; o[1] += b1.y; o[2] += b1.z; o[3] += b1.w;
  } else {
    o[0] = sx; o[1] = sy; o[2] = sz; o[3] = sx + sy;
  }
#undef SB0_
  wave_sync();   // the wave's LDS may be overwritten by the caller / next pass
  sigma = inb ? o[0] : sg_fill;
  rgb[0] = inb ? (1.0f / (1.0f + __expf(-o[1]))) * 1.002f - 0.001f : 0.f;
  rgb[1] = inb ? (1.0f / (1.0f + __expf(-o[2]))) * 1.002f - 0.001f : 0.f;
  rgb[2] = inb ? (1.0f / (1.0f + __expf(-o[3]))) * 1.002f - 0.001f : 0.f;
}

// ---- wave-wide scans / reductions / neighbour shifts on the DPP cross-lane path (gfx9 row_shr / row_bcast / wave_shr controls):
// a VALU-rate instruction each instead of a ds_bpermute round trip through the LDS (the per-ray code has ~85 of them in
// dependent chains).  Scan recipe = LLVM's wave64 gfx9 one: row_shr 1, 2, 4, 8 inside the rows of 16, then lane 15 of row r
// into row r+1 (row_bcast:15, rows 1 and 3) and lane 31 into rows 2-3 (row_bcast:31).
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_WAVE_SHL1 0x130     // lane i <- lane i+1
#define DPP_WAVE_SHR1 0x138     // lane i <- lane i-1
#define DPP_BCAST15 0x142
#define DPP_BCAST31 0x143
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_mov(float old, float src) {     // lanes without a source keep `old`
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float lane_prev(float x, float fill) { return dpp_mov<DPP_WAVE_SHR1>(fill, x); }   // x of lane-1 (lane 0: fill)
__device__ __forceinline__ float lane_next(float x, float fill) { return dpp_mov<DPP_WAVE_SHL1>(fill, x); }   // x of lane+1 (lane 63: fill)
__device__ __forceinline__ float wave_incl_prod_dpp(float x) {
  x *= dpp_mov<DPP_ROW_SHR(1)>(1.0f, x); x *= dpp_mov<DPP_ROW_SHR(2)>(1.0f, x);
  x *= dpp_mov<DPP_ROW_SHR(4)>(1.0f, x); x *= dpp_mov<DPP_ROW_SHR(8)>(1.0f, x);
  x *= dpp_mov<DPP_BCAST15, 0xa>(1.0f, x); x *= dpp_mov<DPP_BCAST31, 0xc>(1.0f, x);
  return x;
}
__device__ __forceinline__ float wave_incl_sum(float x, int) {
  x += dpp_mov<DPP_ROW_SHR(1)>(0.0f, x); x += dpp_mov<DPP_ROW_SHR(2)>(0.0f, x);
  x += dpp_mov<DPP_ROW_SHR(4)>(0.0f, x); x += dpp_mov<DPP_ROW_SHR(8)>(0.0f, x);
  x += dpp_mov<DPP_BCAST15, 0xa>(0.0f, x); x += dpp_mov<DPP_BCAST31, 0xc>(0.0f, x);
  return x;
}
__device__ __forceinline__ float wave_excl_prod(float v, int) { return lane_prev(wave_incl_prod_dpp(v), 1.0f); }
__device__ __forceinline__ float wave_total(float v) {               // sum over the 64 lanes, the same value in every lane
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_incl_sum(v, 0)), 63));
}

// cnt += bit `lane` of mask: the mask goes in as the carry of v_addc (one vector instruction; shifting the 64-bit mask per lane costs four)
__device__ __forceinline__ void add_mask(int& cnt, uint64_t m) {
  asm("v_addc_co_u32_e64 %0, vcc, 0, %0, %1" : "+v"(cnt) : "s"(m) : "vcc");
}

// the workgroup's copy of the decoder image (built once per launch by render_init_kernel) behind the 4 wave regions
__device__ __forceinline__ const char* stage_decoder(char* lds_bytes, const float* dec_img, int nwpb = 4) {
  char* cimg = lds_bytes + nwpb * WAVE_LDS_BYTES;
  const uint4* src = reinterpret_cast<const uint4*>(dec_img);
  for (int i = threadIdx.x; i < DEC_BYTES / 16; i += blockDim.x) reinterpret_cast<uint4*>(cimg)[i] = src[i];
  __syncthreads();
  return cimg;
}
#define RENDER_LDS_BYTES (4 * WAVE_LDS_BYTES + DEC_BYTES)
#ifndef RENDER_LDS_PAD
#define RENDER_LDS_PAD 0        // bench builds: extra dynamic LDS per workgroup (> 80 KB leaves ONE workgroup per CU: profiles/r6_render_slp.md)
#endif
#define RENDER_K_LDS_BYTES (RENDER_WPB * WAVE_LDS_BYTES + DEC_BYTES + RENDER_LDS_PAD)

__device__ __forceinline__ void flush_depth_range(uint32_t* scal_u, int grp, float dmin_l, float dmax_l, int lane) {
  if (grp < 0) return;
  dmin_l = wave_min(dmin_l); dmax_l = wave_max(dmax_l);
  if (lane == 0 && dmin_l <= dmax_l) {
    uint32_t* r = scal_u + (int64_t)grp * GRP_WORDS;
    atomicMin(&r[2], enc_f(dmin_l));
    atomicMax(&r[3], enc_f(dmax_l));
  }
}

__global__ __launch_bounds__(64 * RENDER_WPB, RENDER_OCC) void render_kernel(RenderP p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const char* cimg = stage_decoder(reinterpret_cast<char*>(lds), p.dec, RENDER_WPB);
  float* feat = lds + wid * WAVE_LDS_FLOATS;       // 2048 floats; reused for cdf/bins/merge arrays between shading passes
  char* wl = reinterpret_cast<char*>(feat);
  const int M = p.M;
  const int64_t nrays = (int64_t)p.V * M;
  const int64_t nwaves = (int64_t)gridDim.x * RENDER_WPB;
  float dmin_l = 3.0e38f, dmax_l = -3.0e38f;
  int cur_grp = -1;

  for (int64_t ray = (int64_t)blockIdx.x * RENDER_WPB + wid; ray < nrays; ray += nwaves) {
    const int v = (int)(ray / M), pix = (int)(ray % M);
    float o[3], d[3];
    if (p.ray_o) {
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) { o[ax] = p.ray_o[3 * ray + ax]; d[ax] = p.ray_d[3 * ray + ax]; }
    } else {
      make_ray(p.cams + 25 * v, p.res, pix, o, d);
    }
    RAY_FENCE(o, d);
    const int grp = v / p.vpc;
    if (grp != cur_grp) {                                       // entering another call's rays: flush this wave's depth range
      flush_depth_range(p.scal_u, cur_grp, dmin_l, dmax_l, lane);
      dmin_l = 3.0e38f; dmax_l = -3.0e38f; cur_grp = grp;
    }
    const uint32_t* gr = p.scal_u + (int64_t)grp * GRP_WORDS;
    float t0 = p.ray_limits[2 * ray], t1 = p.ray_limits[2 * ray + 1];
    if (gr[4] != 0u && !(t1 > t0)) { t0 = dec_f(gr[0]); t1 = dec_f(gr[1]); }    // renderer.py:151-155 (sic)
    const float* planes = p.planes + (int64_t)p.plane_index[v] * 3 * p.H * p.W * 32;

    // ---- coarse depths: linspace + jitter * delta
    const float step = (float)lane / (float)(NS - 1);
    const float delta = (t1 - t0) / (float)(NS - 1);
    const float zc = (t0 + step * (t1 - t0)) + p.jitter[ray * NS + lane] * delta;
    float rgbc[3], sigc;
    shade64(p, planes, wl, cimg, o[0] + zc * d[0], o[1] + zc * d[1], o[2] + zc * d[2], lane, rgbc, sigc);
    if (p.coarse_sigma) p.coarse_sigma[ray * NS + lane] = sigc;
    if (p.coarse_coords) {
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) p.coarse_coords[(ray * NS + lane) * 3 + ax] = o[ax] + zc * d[ax];
    }

    // ---- coarse ray-march weights (63 intervals: lane i <-> samples i, i+1)
    const float zn = lane_next(zc, zc), sn = lane_next(sigc, sigc);
    float wgt;
    {
      const float dl = zn - zc;
      const float dm = softplus20_hw((sigc + sn) * 0.5f - 1.0f);
      float alpha = 1.0f - exp_neg_hw(dm * dl);
      if (lane == NS - 1) alpha = 0.f;
      const float T = wave_excl_prod(lane == NS - 1 ? 1.0f : (1.0f - alpha + 1e-10f), lane);
      wgt = alpha * T;                                       // lane 63: 0 (unused)
    }
    // ---- importance sampling (renderer.py:479-552)
    float zf;
    {
      const float wm1 = lane_prev(wgt, wgt);
      // max_pool1d(k=2,s=1,pad=1) over 63 weights -> 64 values
      const float mp = lane == 0 ? wgt : (lane == NS - 1 ? wm1 : fmaxf(wm1, wgt));
      const float mpn = lane_next(mp, mp);
      const float av = (mp + mpn) * 0.5f + 0.01f;           // avg_pool1d(2,1): 63 values (lanes 0..62)
      // pdf over weights[1:-1]  -> 61 values; lane k (0..60) takes av[k+1]
      const float wk = lane_next(av, av) + 1e-5f;
      const float wv = lane < NS - 3 ? wk : 0.f;
      const float tot = wave_total(wv);
      const float pdf = wv / tot;
      const float cdf_incl = wave_incl_sum(pdf, lane);
      const float cdf_excl = lane_prev(cdf_incl, 0.f);
      const float cdf = lane == 0 ? 0.f : cdf_excl;          // cdf[k], k = 0..61 valid
      const float zmid = 0.5f * (zc + zn);                  // bins[k], k = 0..62 valid
      float* cdf_s = feat; float* bin_s = feat + 64;
      cdf_s[lane] = lane <= NS - 3 ? cdf : 3.0e38f;
      bin_s[lane] = zmid;
      wave_sync();
      const float u = p.u_fine[ray * NS + lane];
      // searchsorted(cdf[0..61], u, right=True) = #entries <= u
      int lo = 0, hi = NS - 2;                               // 62 entries
#pragma unroll
      for (int it = 0; it < 6; ++it) {
        const int mid = (lo + hi) >> 1;
        const bool le = lo < hi && cdf_s[mid] <= u;
        lo = le ? mid + 1 : lo;
        hi = (lo < hi && !le) ? mid : hi;
      }
      const int inds = lo;
      const int below = inds - 1 < 0 ? 0 : inds - 1;
      const int above = inds > NS - 3 ? NS - 3 : inds;       // clamp_max(N_samples_=61)
      const float cb = cdf_s[below], ca = cdf_s[above], bb = bin_s[below], ba = bin_s[above];
      float den = ca - cb;
      den = den < 1e-5f ? 1.0f : den;
      zf = bb + (u - cb) / den * (ba - bb);
      wave_sync();
    }
    if (p.fine_depths) p.fine_depths[ray * NS + lane] = zf;
    float rgbf[3], sigf;
    shade64(p, planes, wl, cimg, o[0] + zf * d[0], o[1] + zf * d[1], o[2] + zf * d[2], lane, rgbf, sigf);
    if (p.fine_sigma) p.fine_sigma[ray * NS + lane] = sigf;
    if (p.fine_coords) {
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) p.fine_coords[(ray * NS + lane) * 3 + ax] = o[ax] + zf * d[ax];
    }

    // ---- merge coarse + fine by rank (replaces cat + torch.sort + gathers)
    float* zc_s = feat; float* zf_s = feat + 64; float* srt = feat + 128;   // srt: {z, sigma, r, g, b} x 128 ranks
    zc_s[lane] = zc; zf_s[lane] = zf;
    wave_sync();
    // Total order: by depth; equal depths: coarse before fine, then by lane.  The coarse depths are NOT assumed to be sorted by
    // lane: on a ray that only grazes the box (t1 - t0 ~ 1e-5) rounding can swap two neighbours, and with a fine sample between
    // them "rank = lane + ..." collided with it (garbage weights on ~1 ray per million at 512^2; the reference sorts for real).
    int rc, rf;
    // r4: the rank counting was 1 956 of the 4 860 vector instructions of a ray (four compare classes x 128 elements with tie rules).
    // With the coarse depths in lane order (every ray but the grazing ones) the coarse self-rank IS the lane and #(zc <= zf) is an
    // upper bound in the sorted LDS copy (7 dependent reads); what is left per fine element k is one compare for the coarse lanes
    // and the (<, ==) pair for the fine ones, accumulated through the carry-in of v_addc (the tie rule "k < lane" is a constant
    // lane mask on the scalar side).  Identical ranks by construction; rays with a swapped neighbour pair or a NaN take the full count.
    // r6: two more cuts on the common path.  (a) The coarse ranks need no compares at all: with the coarse depths in order, fine sample k lies
    // behind exactly pos_k coarse ones (zc_j <= zf_k <=> j < pos_k), so #(zf_k < zc_j) = #(k: pos_k <= j) - a 65-bin histogram of the
    // positions (one LDS atomic per lane) and a wave prefix sum instead of 64 compares + 64 carry adds.  (b) Among the fine samples only
    // the strict compare is counted; two bit-equal fine depths (the only case the tie rule exists for) would then share a rank, which
    // shows as a rank total below 0 + 1 + ... + 127 = 8128 - checked with one wave sum, and such a ray takes the full count below like the
    // grazing ones.  Identical ranks by construction on every ray that stays on this path.
    const bool in_order = (zc <= zn) && (zf == zf);
    bool fast = __builtin_amdgcn_ballot_w64(!in_order) == 0ull;
    if (fast) {
      int pos = 0;
#pragma unroll
      for (int s2 = 32; s2 >= 1; s2 >>= 1) pos += (zc_s[pos + s2 - 1] <= zf) ? s2 : 0;
      pos += (zc_s[pos] <= zf) ? 1 : 0;
      int* cnt = reinterpret_cast<int*>(feat + 768);                         // 65 bins behind the five rank planes
      cnt[lane] = 0;
      if (lane == 0) cnt[64] = 0;
      wave_sync();
      atomicAdd(&cnt[pos], 1);
      wave_sync();
      rc = lane + (int)wave_incl_sum((float)cnt[lane], lane);                // counts <= 64: exact in fp32
      rf = pos;
#pragma unroll
      for (int k = 0; k < NS; k += 4) {
        const float4 a = *reinterpret_cast<const float4*>(zf_s + k);
        add_mask(rf, __builtin_amdgcn_ballot_w64(a.x < zf)); add_mask(rf, __builtin_amdgcn_ballot_w64(a.y < zf));
        add_mask(rf, __builtin_amdgcn_ballot_w64(a.z < zf)); add_mask(rf, __builtin_amdgcn_ballot_w64(a.w < zf));
      }
      fast = wave_total((float)(rc + rf)) == 8128.0f;                        // 64 ranks + 64 ranks, all <= 127: exact in fp32
    }
    if (!fast) {
      rc = 0; rf = 0;
#pragma unroll 4
      for (int k = 0; k < NS; k += 4) {
        const float4 a = *reinterpret_cast<const float4*>(zf_s + k);
        const float4 b = *reinterpret_cast<const float4*>(zc_s + k);
        rc += (a.x < zc) + (a.y < zc) + (a.z < zc) + (a.w < zc);
        rc += (b.x < zc || (b.x == zc && k + 0 < lane)) + (b.y < zc || (b.y == zc && k + 1 < lane)) +
              (b.z < zc || (b.z == zc && k + 2 < lane)) + (b.w < zc || (b.w == zc && k + 3 < lane));
        rf += (b.x <= zf) + (b.y <= zf) + (b.z <= zf) + (b.w <= zf);
        rf += (a.x < zf || (a.x == zf && k + 0 < lane)) + (a.y < zf || (a.y == zf && k + 1 < lane)) +
              (a.z < zf || (a.z == zf && k + 2 < lane)) + (a.w < zf || (a.w == zf && k + 3 < lane));
      }
    }
    // five planes of 128 ({z}, {sigma}, {r}, {g}, {b} by rank): lane i then reads its elements 2i, 2i+1 as ONE 8-byte access per plane
    // and takes element 2i+2 from lane i+1 over DPP (r2's [rank][5] records cost 15 reads with 2-way bank conflicts each)
    srt[0 * 128 + rc] = zc; srt[1 * 128 + rc] = sigc; srt[2 * 128 + rc] = rgbc[0]; srt[3 * 128 + rc] = rgbc[1]; srt[4 * 128 + rc] = rgbc[2];
    srt[0 * 128 + rf] = zf; srt[1 * 128 + rf] = sigf; srt[2 * 128 + rf] = rgbf[0]; srt[3 * 128 + rf] = rgbf[1]; srt[4 * 128 + rf] = rgbf[2];
    wave_sync();
    // lane i: elements 2i, 2i+1, 2i+2 -> intervals 2i and 2i+1 (interval 127 does not exist: lane 63 repeats element 127)
    float e[3][5];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const float2 p01 = *reinterpret_cast<const float2*>(srt + c * 128 + 2 * lane);
      e[0][c] = p01.x; e[1][c] = p01.y;
      e[2][c] = lane_next(p01.x, p01.y);                  // lane 63 keeps its own element 127
    }
    wave_sync();
    float a0, a1;
    {
      const float dm0 = softplus20_hw((e[0][1] + e[1][1]) * 0.5f - 1.0f);
      a0 = 1.0f - exp_neg_hw(dm0 * (e[1][0] - e[0][0]));
      const float dm1 = softplus20_hw((e[1][1] + e[2][1]) * 0.5f - 1.0f);
      a1 = lane == NS - 1 ? 0.f : 1.0f - exp_neg_hw(dm1 * (e[2][0] - e[1][0]));
    }
    const float f0 = 1.0f - a0 + 1e-10f, f1 = lane == NS - 1 ? 1.0f : (1.0f - a1 + 1e-10f);
    const float Ts = wave_excl_prod(f0 * f1, lane);
    const float w0 = a0 * Ts, w1 = a1 * (Ts * f0);
    float acc_r = w0 * (e[0][2] + e[1][2]) * 0.5f + w1 * (e[1][2] + e[2][2]) * 0.5f;
    float acc_g = w0 * (e[0][3] + e[1][3]) * 0.5f + w1 * (e[1][3] + e[2][3]) * 0.5f;
    float acc_b = w0 * (e[0][4] + e[1][4]) * 0.5f + w1 * (e[1][4] + e[2][4]) * 0.5f;
    float acc_d = w0 * (e[0][0] + e[1][0]) * 0.5f + w1 * (e[1][0] + e[2][0]) * 0.5f;
    float acc_w = w0 + w1;
    acc_r = wave_total(acc_r); acc_g = wave_total(acc_g); acc_b = wave_total(acc_b);
    acc_d = wave_total(acc_d); acc_w = wave_total(acc_w);
    dmin_l = fminf(dmin_l, fminf(zc, zf)); dmax_l = fmaxf(dmax_l, fmaxf(zc, zf));
    if (lane == 0) {
      if (p.white_back) { acc_r += 1.0f - acc_w; acc_g += 1.0f - acc_w; acc_b += 1.0f - acc_w; }
      const int64_t img = (int64_t)v * 3 * M;
      p.rgb[img + pix] = acc_r * 2.0f - 1.0f;
      p.rgb[img + M + pix] = acc_g * 2.0f - 1.0f;
      p.rgb[img + 2 * M + pix] = acc_b * 2.0f - 1.0f;
      p.depth[ray] = acc_d;          // clamped by render_finalize_kernel
      p.wsum[ray] = acc_w;
    }
    if (p.vis) {                     // T behind the last of the 127 intervals (lane 63: f1 = 1) - 'visibility' of the reference's return dict
      const float tv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Ts * f0 * f1), 63));
      if (lane == 0) p.vis[ray] = tv;
    }
  }
  flush_depth_range(p.scal_u, cur_grp, dmin_l, dmax_l, lane);
}

// depth = clamp(nan_to_num(depth, inf), min(all depths), max(all depths))   (ray_marcher.py:57-61), "all" = the call's group
__global__ void render_finalize_kernel(float* depth, const uint32_t* scal_u, int64_t n, int64_t rays_per_group) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t* r = scal_u + (i / rays_per_group) * GRP_WORDS;
  const float lo = dec_f(r[2]), hi = dec_f(r[3]);
  float dv = depth[i];
  if (dv != dv) dv = INFINITY;
  depth[i] = fminf(fmaxf(dv, lo), hi);
}

// ------------------------------------------------------------------ generic ray-marcher: every preset of nsr/script_util.py:433-1000
// render_kernel above is the Objaverse preset (64 + 64 samples, 'auto' limits, bbox filter) with lane = sample.  The other presets the
// reference's entry points select - depth_resolution / depth_resolution_importance in {48, 64, 80, 96, 128}, numeric ray_start /
// ray_end (ShapeNet, FFHQ), no bbox filter - and the per-sample outputs of return_meta (weights, all_coords, feature_volume:
// renderer.py:283-300) go through this kernel: still one wavefront per ray and the same gather + MFMA decoder (shade64, up to two
// passes of 64 points per sampling stage), but the per-ray arrays live in the wave's LDS (15 KB) and a lane owns CONSECUTIVE
// intervals of the march (2 of the <= 127 coarse ones, 4 of the <= 255 merged ones), so any sample count <= 128 + 128 fits one code
// path.  Evaluation order follows the reference operator by operator like the kernel above; it is not tuned beyond that.
#define GEN_MAXS 128
#define G_ZC 2048               // float offsets inside the wave's region: [0, 2048) is shade64's feature tile
#define G_ZF (G_ZC + 5 * GEN_MAXS)
// r6: the merged, depth-sorted planes {z, sigma, r, g, b} x 256 (1 280 floats) live in the feature tile - they are written behind the ray's
// last shade64 and read until the ray ends, the tile is written again by the next ray's first shade64 (a wave_sync in between).  15 KB per wave
// instead of 20: 70.5 KB per workgroup, so TWO workgroups share a CU (two waves per SIMD at the kernel's 252 VGPRs) where one did.
#define G_SRT 0
#define G_W (G_ZF + 5 * GEN_MAXS)
#define G_CDF (G_W + 2 * GEN_MAXS)
#define G_BIN (G_CDF + GEN_MAXS)
#define GEN_WAVE_FLOATS (G_BIN + GEN_MAXS)
#define GEN_LDS_BYTES (4 * GEN_WAVE_FLOATS * 4 + DEC_BYTES)
static_assert(5 * 2 * GEN_MAXS <= WAVE_LDS_FLOATS, "the merged planes fit the feature tile");
static_assert(2 * GEN_LDS_BYTES <= 160 * 1024, "two workgroups of the generic ray-marcher per CU");

// MipRayMarcher2.run_forward over n samples held as 5 planes {z, sigma, r, g, b} with plane stride PS in LDS: lane owns intervals
// IPL * lane .. IPL * lane + IPL - 1.  Returns the five sums (same value in every lane) and T behind the last interval; writes the
// n - 1 weights to w_lds (LDS, may be null) and w_glb (global, may be null).
template <int IPL>
__device__ __forceinline__ void march_lds(const float* a, int PS, int n, int lane, float* w_lds, float* w_glb, float& acc_r, float& acc_g,
                                          float& acc_b, float& acc_d, float& acc_w, float& vis) {
  float al[IPL], fk[IPL], mr[IPL], mg[IPL], mb[IPL], mz[IPL];
  float prod = 1.0f;
#pragma unroll
  for (int q = 0; q < IPL; ++q) {
    const int i = IPL * lane + q;
    const bool ok = i < n - 1;
    const int i0 = ok ? i : n - 1, i1 = ok ? i + 1 : n - 1;
    const float z0 = a[i0], z1 = a[i1], s0 = a[PS + i0], s1 = a[PS + i1];
    const float dm = softplus20_hw((s0 + s1) * 0.5f - 1.0f);
    const float alpha = ok ? 1.0f - exp_neg_hw(dm * (z1 - z0)) : 0.f;
    al[q] = alpha;
    fk[q] = ok ? (1.0f - alpha + 1e-10f) : 1.0f;
    prod *= fk[q];
    mr[q] = (a[2 * PS + i0] + a[2 * PS + i1]) * 0.5f; mg[q] = (a[3 * PS + i0] + a[3 * PS + i1]) * 0.5f;
    mb[q] = (a[4 * PS + i0] + a[4 * PS + i1]) * 0.5f; mz[q] = (z0 + z1) * 0.5f;
  }
  const float incl = wave_incl_prod_dpp(prod);
  float run = lane_prev(incl, 1.0f);
  vis = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(incl), 63));
  float r = 0.f, g = 0.f, b = 0.f, d = 0.f, w = 0.f;
#pragma unroll
  for (int q = 0; q < IPL; ++q) {
    const int i = IPL * lane + q;
    const float wq = al[q] * run;
    run *= fk[q];
    r += wq * mr[q]; g += wq * mg[q]; b += wq * mb[q]; d += wq * mz[q]; w += wq;
    if (i < n - 1) {
      if (w_lds) w_lds[i] = wq;
      if (w_glb) w_glb[i] = wq;
    }
  }
  acc_r = wave_total(r); acc_g = wave_total(g); acc_b = wave_total(b); acc_d = wave_total(d); acc_w = wave_total(w);
}

__global__ __launch_bounds__(256, 1) void render_generic_kernel(RenderP p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  char* lds_b = reinterpret_cast<char*>(lds);
  char* cimg_w = lds_b + 4 * GEN_WAVE_FLOATS * 4;
  {
    const uint4* src = reinterpret_cast<const uint4*>(p.dec);
    for (int i = threadIdx.x; i < DEC_BYTES / 16; i += blockDim.x) reinterpret_cast<uint4*>(cimg_w)[i] = src[i];
    __syncthreads();
  }
  const char* cimg = cimg_w;
  float* wf = lds + wid * GEN_WAVE_FLOATS;
  char* wl = reinterpret_cast<char*>(wf);
  const int M = p.M, S = p.S, NI = p.NI, NT = S + NI;
  const int64_t nrays = (int64_t)p.V * M;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  float dmin_l = 3.0e38f, dmax_l = -3.0e38f;
  int cur_grp = -1;
  for (int64_t ray = (int64_t)blockIdx.x * 4 + wid; ray < nrays; ray += nwaves) {
    const int v = (int)(ray / M), pix = (int)(ray % M);
    float o[3], d[3];
    if (p.ray_o) {
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) { o[ax] = p.ray_o[3 * ray + ax]; d[ax] = p.ray_d[3 * ray + ax]; }
    } else {
      make_ray(p.cams + 25 * v, p.res, pix, o, d);
    }
    RAY_FENCE(o, d);
    const int grp = v / p.vpc;
    if (grp != cur_grp) {
      flush_depth_range(p.scal_u, cur_grp, dmin_l, dmax_l, lane);
      dmin_l = 3.0e38f; dmax_l = -3.0e38f; cur_grp = grp;
    }
    float t0 = p.t_start, t1 = p.t_end;
    if (!p.numeric) {
      const uint32_t* gr = p.scal_u + (int64_t)grp * GRP_WORDS;
      t0 = p.ray_limits[2 * ray]; t1 = p.ray_limits[2 * ray + 1];
      if (gr[4] != 0u && !(t1 > t0)) { t0 = dec_f(gr[0]); t1 = dec_f(gr[1]); }    // renderer.py:151-155 (sic)
    }
    const float* planes = p.planes + (int64_t)p.plane_index[v] * 3 * p.H * p.W * 32;
    // sentinels: depths past the sample count compare as +inf in the rank counts and searches
    wf[G_ZC + lane] = 3.0e38f; wf[G_ZC + 64 + lane] = 3.0e38f; wf[G_ZF + lane] = 3.0e38f; wf[G_ZF + 64 + lane] = 3.0e38f;
    wave_sync();

    // ---- coarse pass: sample s = lane + 64 j
    const float delta = (t1 - t0) / (float)(S - 1);
    float zc[2];
#pragma unroll 1
    for (int j = 0; j < 2; ++j) {
      const int sidx = lane + 64 * j;
      if (64 * j >= S) { zc[j] = 3.0e38f; continue; }               // wave-uniform
      const bool ok = sidx < S;
      const int sc = ok ? sidx : S - 1;
      float base;
      if (p.numeric) {           // torch.linspace (renderer.py:466-470): from the start below the midpoint, from the end above it
        base = sc < S / 2 ? t0 + delta * (float)sc : t1 - delta * (float)(S - 1 - sc);
      } else {                   // math_utils.linspace (:121-137): start + i / (num - 1) * (stop - start)
        base = t0 + ((float)sc / (float)(S - 1)) * (t1 - t0);
      }
      const float z = base + p.jitter[ray * S + sc] * delta;
      float rgb[3], sg;
      shade64(p, planes, wl, cimg, o[0] + z * d[0], o[1] + z * d[1], o[2] + z * d[2], lane, rgb, sg);
      zc[j] = ok ? z : 3.0e38f;
      if (ok) {
        wf[G_ZC + sidx] = z; wf[G_ZC + GEN_MAXS + sidx] = sg; wf[G_ZC + 2 * GEN_MAXS + sidx] = rgb[0];
        wf[G_ZC + 3 * GEN_MAXS + sidx] = rgb[1]; wf[G_ZC + 4 * GEN_MAXS + sidx] = rgb[2];
        dmin_l = fminf(dmin_l, z); dmax_l = fmaxf(dmax_l, z);
        if (p.coarse_sigma) p.coarse_sigma[ray * S + sidx] = sg;
        if (p.coarse_coords) {
#pragma unroll
          for (int ax = 0; ax < 3; ++ax) p.coarse_coords[(ray * S + sidx) * 3 + ax] = o[ax] + z * d[ax];
        }
      }
    }
    wave_sync();
    // ---- coarse weights -> importance samples (renderer.py:479-552)
    {
      float r_, g_, b_, d_, w_, v_;
      march_lds<2>(wf + G_ZC, GEN_MAXS, S, lane, wf + G_W, nullptr, r_, g_, b_, d_, w_, v_);
    }
    wave_sync();
    {
      const float* w = wf + G_W;            // S - 1 weights
      float wv[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int k = lane + 64 * j;        // pdf entry k takes avg-pooled value k + 1 (weights[:, 1:-1])
        const bool ok = k < S - 3;
        const int kk = ok ? k : 0;
        const float mp0 = fmaxf(w[kk], w[kk + 1]), mp1 = fmaxf(w[kk + 1], w[kk + 2]);
        const float av = (mp0 + mp1) * 0.5f + 0.01f;
        wv[j] = ok ? av + 1e-5f : 0.f;
      }
      const float tot = wave_total(wv[0] + wv[1]);
      const float c0 = wave_incl_sum(wv[0] / tot, lane);
      const float t63 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c0), 63));
      const float c1 = wave_incl_sum(wv[1] / tot, lane) + t63;
      // cdf[0] = 0, cdf[k + 1] = inclusive sum k (k < S - 3): S - 2 entries, +inf behind them
      if (lane == 0) wf[G_CDF] = 0.f;
      wf[G_CDF + lane + 1] = lane < S - 3 ? c0 : 3.0e38f;
      if (lane + 65 < GEN_MAXS) wf[G_CDF + lane + 65] = lane + 64 < S - 3 ? c1 : 3.0e38f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int k = lane + 64 * j;
        const int k0 = k < S - 1 ? k : S - 2;
        wf[G_BIN + k] = 0.5f * (wf[G_ZC + k0] + wf[G_ZC + k0 + 1]);
      }
    }
    wave_sync();
    float zf[2];
#pragma unroll 1
    for (int j = 0; j < 2; ++j) {
      const int fidx = lane + 64 * j;
      if (64 * j >= NI) { zf[j] = 3.0e38f; continue; }
      const bool ok = fidx < NI;
      const float u = p.u_fine[ray * NI + (ok ? fidx : NI - 1)];
      int pos = 0;                                       // searchsorted(cdf[0 .. S-3], u, right=True): entries <= u
#pragma unroll
      for (int s2 = 64; s2 >= 1; s2 >>= 1) pos += (wf[G_CDF + pos + s2 - 1] <= u) ? s2 : 0;
      pos += (wf[G_CDF + pos] <= u) ? 1 : 0;
      const int below = pos - 1 < 0 ? 0 : pos - 1;
      const int above = pos > S - 3 ? S - 3 : pos;         // clamp_max(N_samples_ = S - 3)
      const float cb = wf[G_CDF + below], ca = wf[G_CDF + above], bb = wf[G_BIN + below], ba = wf[G_BIN + above];
      float den = ca - cb;
      den = den < 1e-5f ? 1.0f : den;
      const float z = bb + (u - cb) / den * (ba - bb);
      float rgb[3], sg;
      shade64(p, planes, wl, cimg, o[0] + z * d[0], o[1] + z * d[1], o[2] + z * d[2], lane, rgb, sg);
      zf[j] = ok ? z : 3.0e38f;
      if (ok) {
        wf[G_ZF + fidx] = z; wf[G_ZF + GEN_MAXS + fidx] = sg; wf[G_ZF + 2 * GEN_MAXS + fidx] = rgb[0];
        wf[G_ZF + 3 * GEN_MAXS + fidx] = rgb[1]; wf[G_ZF + 4 * GEN_MAXS + fidx] = rgb[2];
        dmin_l = fminf(dmin_l, z); dmax_l = fmaxf(dmax_l, z);
        if (p.fine_depths) p.fine_depths[ray * NI + fidx] = z;
        if (p.fine_sigma) p.fine_sigma[ray * NI + fidx] = sg;
        if (p.fine_coords) {
#pragma unroll
          for (int ax = 0; ax < 3; ++ax) p.fine_coords[(ray * NI + fidx) * 3 + ax] = o[ax] + z * d[ax];
        }
      }
    }
    wave_sync();
    // ---- unify_samples (renderer.py:422-435): rank of every sample in the stable order of cat([coarse, fine]) by depth
    {
      int rk[4] = {0, 0, 0, 0};
      const float ze[4] = {zc[0], zc[1], zf[0], zf[1]};
      const int ce[4] = {lane, lane + 64, S + lane, S + lane + 64};          // index in the concatenation
      const int S4 = (S + 3) & ~3, N4 = (NI + 3) & ~3;
#pragma unroll 1
      for (int k = 0; k < S4; k += 4) {
        const float4 a = *reinterpret_cast<const float4*>(wf + G_ZC + k);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          rk[e] += (a.x < ze[e] || (a.x == ze[e] && k + 0 < ce[e])) + (a.y < ze[e] || (a.y == ze[e] && k + 1 < ce[e])) +
                   (a.z < ze[e] || (a.z == ze[e] && k + 2 < ce[e])) + (a.w < ze[e] || (a.w == ze[e] && k + 3 < ce[e]));
      }
#pragma unroll 1
      for (int k = 0; k < N4; k += 4) {
        const float4 a = *reinterpret_cast<const float4*>(wf + G_ZF + k);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          rk[e] += (a.x < ze[e] || (a.x == ze[e] && S + k + 0 < ce[e])) + (a.y < ze[e] || (a.y == ze[e] && S + k + 1 < ce[e])) +
                   (a.z < ze[e] || (a.z == ze[e] && S + k + 2 < ce[e])) + (a.w < ze[e] || (a.w == ze[e] && S + k + 3 < ce[e]));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int li = lane + 64 * (e & 1);
        const bool ok = (e < 2) ? li < S : li < NI;
        const float* src = wf + (e < 2 ? G_ZC : G_ZF) + li;
        if (ok && rk[e] < NT) {
#pragma unroll
          for (int c = 0; c < 5; ++c) wf[G_SRT + c * 2 * GEN_MAXS + rk[e]] = src[c * GEN_MAXS];
        }
      }
    }
    wave_sync();
    float acc_r, acc_g, acc_b, acc_d, acc_w, vis;
    march_lds<4>(wf + G_SRT, 2 * GEN_MAXS, NT, lane, nullptr, p.w_all ? p.w_all + ray * (NT - 1) : nullptr, acc_r, acc_g, acc_b, acc_d, acc_w, vis);
    if (p.coords_all || p.colors_all) {
      for (int i = lane; i < NT; i += 64) {
        const float z = wf[G_SRT + i];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
          if (p.coords_all) p.coords_all[(ray * NT + i) * 3 + ax] = o[ax] + z * d[ax];
          if (p.colors_all) p.colors_all[(ray * NT + i) * 3 + ax] = wf[G_SRT + (2 + ax) * 2 * GEN_MAXS + i];
        }
      }
    }
    if (lane == 0) {
      if (p.white_back) { acc_r += 1.0f - acc_w; acc_g += 1.0f - acc_w; acc_b += 1.0f - acc_w; }
      const int64_t img = (int64_t)v * 3 * M;
      p.rgb[img + pix] = acc_r * 2.0f - 1.0f;
      p.rgb[img + M + pix] = acc_g * 2.0f - 1.0f;
      p.rgb[img + 2 * M + pix] = acc_b * 2.0f - 1.0f;
      p.depth[ray] = acc_d;          // clamped by render_finalize_kernel
      p.wsum[ray] = acc_w;
      if (p.vis) p.vis[ray] = vis;
    }
    wave_sync();
  }
  flush_depth_range(p.scal_u, cur_grp, dmin_l, dmax_l, lane);
}

extern "C" int ln3d_render_triplane(const ln3d_render_args* a, void* stream) {
  if (!a || !a->planes || !a->plane_index || !a->jitter || !a->u_fine || !a->rgb || !a->depth || !a->wsum ||
      !a->ray_limits || !a->scalars || !a->dec_w0 || !a->dec_b0 || !a->dec_w1 || !a->dec_b1)
    return LN3D_ERR_BAD_ARG;
  if (a->V <= 0 || (a->res <= 0 && a->rays_per_view <= 0)) return LN3D_ERR_BAD_ARG;
  if (!a->cams && !(a->ray_o && a->ray_d)) return LN3D_ERR_BAD_ARG;       // cameras, or explicit rays
  if ((a->ray_o != nullptr) != (a->ray_d != nullptr)) return LN3D_ERR_BAD_ARG;
  if (a->rays_per_view > 0 && !a->ray_o && a->rays_per_view != a->res * a->res) return LN3D_ERR_BAD_ARG;   // camera rays are a res x res image
  const int S = a->depth_resolution > 0 ? a->depth_resolution : NS, NI = a->depth_resolution_importance > 0 ? a->depth_resolution_importance : NS;
  if (S < 4 || S > GEN_MAXS || NI < 1 || NI > GEN_MAXS) return LN3D_ERR_UNSUPPORTED;
  if (a->ray_mode != 0 && !(a->ray_end > a->ray_start)) return LN3D_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  RenderP p;
  p.planes = a->planes; p.H = a->H; p.W = a->W; p.plane_index = a->plane_index; p.cams = a->cams; p.V = a->V; p.res = a->res;
  p.M = a->rays_per_view > 0 ? a->rays_per_view : a->res * a->res;
  p.jitter = a->jitter; p.u_fine = a->u_fine;
  p.coord_scale = (float)(2.0 / (double)a->box_warp); p.white_back = a->white_back;
  p.bbox_min = a->no_bbox_filter ? -3.0e38f : a->bbox_min; p.bbox_max = a->no_bbox_filter ? 3.0e38f : a->bbox_max;
  p.rgb = a->rgb; p.depth = a->depth; p.wsum = a->wsum; p.ray_limits = a->ray_limits;
  p.scal_u = reinterpret_cast<uint32_t*>(a->scalars) + GRP_OFF; p.dec = a->scalars + DEC_OFF;
  p.vpc = (a->views_per_call <= 0 || a->views_per_call > a->V) ? a->V : a->views_per_call;
  const int groups = (a->V + p.vpc - 1) / p.vpc;
  if (groups > MAX_GROUPS) return LN3D_ERR_BAD_ARG;             // render in chunks of <= MAX_GROUPS calls
  p.coarse_sigma = a->coarse_sigma; p.fine_depths = a->fine_depths;
  p.ray_o = a->ray_o; p.ray_d = a->ray_d; p.fine_sigma = a->fine_sigma; p.coarse_coords = a->coarse_coords; p.fine_coords = a->fine_coords;
  p.vis = a->visibility;
  p.S = S; p.NI = NI; p.numeric = a->ray_mode != 0; p.t_start = a->ray_start; p.t_end = a->ray_end;
  p.w_all = a->weights; p.coords_all = a->all_coords; p.colors_all = a->feature_volume;
  const int64_t nrays = (int64_t)a->V * p.M;
  // the lane = sample kernel serves the Objaverse preset; everything else (and the return_meta outputs) the generic one
  const bool fast = S == NS && NI == NS && !p.numeric && !a->no_bbox_filter && !p.w_all && !p.coords_all && !p.colors_all;
  hipLaunchKernelGGL(render_init_kernel, dim3(8), dim3(256), 0, s, p.scal_u, groups, a->scalars + DEC_OFF, a->dec_w0, a->dec_b0, a->dec_w1, a->dec_b1);
  if (!p.numeric)
    hipLaunchKernelGGL(ray_limits_kernel, dim3((unsigned)((nrays + 255) / 256)), dim3(256), 0, s, p, a->box_warp * 0.5f);
  if (fast) {
    static AttrOnce attr_once;
    if (attr_once.need()) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&render_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RENDER_K_LDS_BYTES);
    }
    int64_t blocks = (nrays + RENDER_WPB - 1) / RENDER_WPB;
    const int64_t cap = 256 * 8 * 4 / RENDER_WPB;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(render_kernel, dim3((unsigned)blocks), dim3(64 * RENDER_WPB), RENDER_K_LDS_BYTES, s, p);
  } else {
    static AttrOnce attr_once;
    if (attr_once.need()) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&render_generic_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, GEN_LDS_BYTES);
    }
    int64_t blocks = (nrays + 3) / 4;
    if (blocks > 256 * 4) blocks = 256 * 4;
    hipLaunchKernelGGL(render_generic_kernel, dim3((unsigned)blocks), dim3(256), GEN_LDS_BYTES, s, p);
  }
  hipLaunchKernelGGL(render_finalize_kernel, dim3((unsigned)((nrays + 255) / 256)), dim3(256), 0, s, a->depth, p.scal_u, nrays, (int64_t)p.vpc * p.M);
  return ln3d_check_launch();
}

// ------------------------------------------------------------------ point query (sigma / rgb grid), no bbox filter
__global__ __launch_bounds__(256) void query_points_kernel(RenderP p, const float* pts, int64_t P, float* sigma, float* rgb) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const char* cimg = stage_decoder(reinterpret_cast<char*>(lds), p.dec);
  char* wl = reinterpret_cast<char*>(lds + wid * WAVE_LDS_FLOATS);
  const int64_t nw = (int64_t)gridDim.x * 4;
  const int64_t ngroups = (P + 63) / 64;
  for (int64_t gidx = (int64_t)blockIdx.x * 4 + wid; gidx < ngroups; gidx += nw) {
    int64_t i = gidx * 64 + lane;
    const bool ok = i < P;
    if (!ok) i = P - 1;
    const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    float c[3], sg;
    shade64(p, p.planes, wl, cimg, x, y, z, lane, c, sg);
    if (ok) { sigma[i] = sg; rgb[3 * i] = c[0]; rgb[3 * i + 1] = c[1]; rgb[3 * i + 2] = c[2]; }
  }
}

extern "C" int ln3d_query_points(const float* planes, int H, int W, const float* points, int64_t P, const float* dec_w0,
                                 const float* dec_b0, const float* dec_w1, const float* dec_b1, float box_warp, float* sigma,
                                 float* rgb, float* scalars, void* stream) {
  if (!planes || !points || !sigma || !rgb || !scalars || P <= 0) return LN3D_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  RenderP p{};
  p.planes = planes; p.H = H; p.W = W; p.coord_scale = (float)(2.0 / (double)box_warp);
  p.bbox_min = -3.0e38f; p.bbox_max = 3.0e38f;
  p.scal_u = reinterpret_cast<uint32_t*>(scalars) + GRP_OFF; p.dec = scalars + DEC_OFF; p.vpc = 1;
  static AttrOnce attr_once;
  if (attr_once.need()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&query_points_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RENDER_LDS_BYTES);
  }
  hipLaunchKernelGGL(render_init_kernel, dim3(8), dim3(256), 0, s, p.scal_u, 0, scalars + DEC_OFF, dec_w0, dec_b0, dec_w1, dec_b1);
  int64_t blocks = ((P + 63) / 64 + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(query_points_kernel, dim3((unsigned)blocks), dim3(256), RENDER_LDS_BYTES, s, p, points, P, sigma, rgb);
  return ln3d_check_launch();
}

// ------------------------------------------------------------------ [NP, 3*C, H, W] -> [NP, 3, H, W, C]
__global__ void planes_to_cl_kernel(const float* src, float* dst, int C, int HW, int64_t total) {
  // one thread per (np*3 + n, hw, c): read src[(pn*C + c)*HW + hw] -> dst[(pn*HW + hw)*C + c]
  __shared__ float tile[32][33];
  const int pn = blockIdx.y;
  const int hw0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int c = ty; c < C; c += 8) {
    const int hw = hw0 + tx;
    tile[c][tx] = hw < HW ? src[((int64_t)pn * C + c) * HW + hw] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int hw = hw0 + r;
    if (hw < HW && tx < C) dst[((int64_t)pn * HW + hw) * C + tx] = tile[tx][r];
  }
}
extern "C" int ln3d_planes_to_channel_last(const float* src, float* dst, int NP, int C, int H, int W, void* stream) {
  if (!src || !dst || C != 32) return LN3D_ERR_BAD_ARG;
  const int HW = H * W;
  hipLaunchKernelGGL(planes_to_cl_kernel, dim3((HW + 31) / 32, NP * 3), dim3(256), 0, (hipStream_t)stream, src, dst, C, HW, (int64_t)NP * 3 * C * HW);
  return ln3d_check_launch();
}
