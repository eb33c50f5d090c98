// Fused (flash-style) attention for gfx950: O = softmax(scale * Q K^T) V, bf16 in/out, fp32 softmax.
//
// Design (MI355X-first):
//  * v_mfma_f32_32x32x16_bf16, wave64.  One workgroup = 4 waves = 128 query rows of one (batch, head);
//    each wave owns 32 query rows, K/V are streamed in 64-key blocks through LDS shared by the 4 waves.
//  * "Swapped" products so that every softmax statistic is lane-local:
//       S^T[key, q] = K . Q^T      (A = K tile rows, B = Q fragment held in VGPRs for the whole kernel)
//       O^T[d,  q]  = V^T . P^T    (A = V^T tile rows, B = P converted to bf16 in registers)
//    In the 32x32 accumulator layout a lane owns ONE query column (lane&31) and 16 of the 32 key rows,
//    so row-max / row-sum are 32 in-register ops + one exchange with lane^32, the online-softmax
//    rescale factor is a per-lane scalar, and P feeds the second MFMA directly as its B operand:
//    the k-index permutation of the accumulator layout is absorbed by reading the V^T A-operand with
//    the same permutation (two ds_read_b64 per fragment) - no LDS round trip for P, no transposes.
//  * V arrives already transposed ([B,H,Dh,Nk_pad]) from the QKV GEMM epilogue, so both LDS tiles are
//    contraction-contiguous.  LDS rows are padded (K: 144 B for Dh=64, V^T: 136 B) which makes the
//    ds_read_b128 / ds_read_b64 fragment reads bank-conflict free.  Tiles are double-buffered; the
//    next block's global loads are issued before the MFMA work of the current block.
#include <type_traits>
#include "common.h"
#include "../../include/ln3d.h"

struct AttnP {
  const bf16_t* Q; const bf16_t* K; const bf16_t* Vt; bf16_t* O;
  int B, H, Nq, Nq_pad, Nk, Nk_pad;
  int64_t ldo;
  float scale_log2;
};

#define KVB 64
#define VROWB 136  // V^T tile row: 64 keys * 2 B + 8 pad

template <int DH>
__global__ __launch_bounds__(256) void attn_kernel(AttnP p) {
  constexpr int KROWB = DH * 2 + 16;          // K tile row bytes (padded)
  constexpr int KTILE = KVB * KROWB;
  constexpr int VTILE = DH * VROWB;
  constexpr int NDS = DH / 16;                // MFMA k-steps over head dim
  constexpr int NDT = DH / 32;                // 32-row d-tiles of O^T
  constexpr int KCH = DH / 8;                 // 16-B chunks per K row
  constexpr int KLD = KVB * KCH / 256;        // K chunks per thread   (2 for DH=64, 4 for 128)
  constexpr int VLD = DH * 8 / 256;           // V^T chunks per thread (2 / 4)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* kbuf = smem;                          // 2 x KTILE
  char* vbuf = smem + 2 * KTILE;              // 2 x VTILE

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int bh = blockIdx.y;
  const int q0 = blockIdx.x * 128 + wid * 32;

  const bf16_t* Qg = p.Q + (int64_t)bh * p.Nq_pad * DH;
  const bf16_t* Kg = p.K + (int64_t)bh * p.Nk_pad * DH;
  const bf16_t* Vg = p.Vt + (int64_t)bh * DH * p.Nk_pad;

  // Q fragments (B operand of S^T = K.Q^T): lane (q = l31, hi) holds d = 16*ds + 8*hi + 0..7
  bf16x8 qf[NDS];
  {
    int qr = q0 + l31; qr = qr < p.Nq ? qr : p.Nq - 1;
    const bf16_t* qp = Qg + (int64_t)qr * DH + hi * 8;
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) qf[ds] = *reinterpret_cast<const bf16x8*>(qp + ds * 16);
  }

  f32x16 oacc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
  float m_run = -3.0e38f, l_run = 0.f;

  const int nkb = (p.Nk + KVB - 1) / KVB;

  // staging maps
  int k_row[KLD], k_c[KLD], v_row[VLD], v_c[VLD];
#pragma unroll
  for (int i = 0; i < KLD; ++i) { const int id = tid + 256 * i; k_row[i] = id / KCH; k_c[i] = id % KCH; }
#pragma unroll
  for (int i = 0; i < VLD; ++i) { const int id = tid + 256 * i; v_row[i] = id >> 3; v_c[i] = id & 7; }

  // in-flight tile lives in named VGPRs (arrays end up in scratch around the sched barriers)
  uint4 kr0, kr1, kr2, kr3, vr0, vr1, vr2, vr3;
  kr2 = kr3 = vr2 = vr3 = make_uint4(0, 0, 0, 0);
#define LN3D_KLOAD(i, dst) dst = *reinterpret_cast<const uint4*>(Kg + (int64_t)(kb_ * KVB + k_row[i]) * DH + k_c[i] * 8)
#define LN3D_VLOAD(i, dst) dst = *reinterpret_cast<const uint4*>(Vg + (int64_t)v_row[i] * p.Nk_pad + kb_ * KVB + v_c[i] * 8)
#define LN3D_KSTORE(i, src) *reinterpret_cast<uint4*>(kdst + k_row[i] * KROWB + k_c[i] * 16) = src
#define LN3D_VSTORE(i, src)                                                        \
  {                                                                                \
    char* d_ = vdst + v_row[i] * VROWB + v_c[i] * 16; /* 8-B aligned only */       \
    *reinterpret_cast<uint2*>(d_) = make_uint2(src.x, src.y);                      \
    *reinterpret_cast<uint2*>(d_ + 8) = make_uint2(src.z, src.w);                  \
  }
#define LN3D_GLOAD(kbv)                                     \
  {                                                         \
    const int kb_ = (kbv);                                  \
    LN3D_KLOAD(0, kr0); LN3D_KLOAD(1, kr1);                 \
    if constexpr (KLD > 2) { LN3D_KLOAD(2, kr2); LN3D_KLOAD(3, kr3); } \
    LN3D_VLOAD(0, vr0); LN3D_VLOAD(1, vr1);                 \
    if constexpr (VLD > 2) { LN3D_VLOAD(2, vr2); LN3D_VLOAD(3, vr3); } \
  }
#define LN3D_LSTORE(bufv)                                   \
  {                                                         \
    char* kdst = kbuf + (bufv) * KTILE; char* vdst = vbuf + (bufv) * VTILE; \
    LN3D_KSTORE(0, kr0); LN3D_KSTORE(1, kr1);               \
    if constexpr (KLD > 2) { LN3D_KSTORE(2, kr2); LN3D_KSTORE(3, kr3); } \
    LN3D_VSTORE(0, vr0); LN3D_VSTORE(1, vr1);               \
    if constexpr (VLD > 2) { LN3D_VSTORE(2, vr2); LN3D_VSTORE(3, vr3); } \
  }

  auto process = [&](int kb, int buf, auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    const char* kt_ = kbuf + buf * KTILE;
    const char* vt_ = vbuf + buf * VTILE;

    // ---- S^T = K . Q^T  : two 32-key tiles
    f32x16 st[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
      for (int ds = 0; ds < NDS; ++ds) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kt_ + (kt * 32 + l31) * KROWB + ds * 32 + hi * 16);
        st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ds], st[kt], 0, 0, 0);
      }
    }

    // ---- online softmax in the log2 domain (lane-local over its 2x16 keys, one exchange with lane^32):
    //      p = exp2(s*c - m) as one FMA + one v_exp per element; the O rescale is skipped (wave-uniformly) when no
    //      row maximum moved, which is exact (alpha == 1), not an approximation.
    if constexpr (TAIL) {
      const int key_base = kb * KVB + 4 * hi;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key_base + kt * 32 + (r & 3) + 8 * (r >> 2);
          st[kt][r] = key < p.Nk ? st[kt][r] : -3.0e38f;
        }
    }
    float mx = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kt][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx * p.scale_log2);     // scale > 0
    if (!__all(m_new == m_run)) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
      m_run = m_new;
    }
    float psum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(st[kt][r], p.scale_log2, -m_new));
        st[kt][r] = pv;
        psum += pv;
      }
    l_run += psum;

    // ---- P^T as B operand: step s uses tile s>>1, regs 8*(s&1)..+7  <->  keys 16s + 4hi + (j&3) + 8(j>>2)
    bf16x8 pb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      union { uint32_t u[4]; bf16x8 v; } cv;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        cv.u[jj] = pack2bf(st[s >> 1][8 * (s & 1) + 2 * jj], st[s >> 1][8 * (s & 1) + 2 * jj + 1]);
      pb[s] = cv.v;
    }

    // ---- O^T += V^T . P^T
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const char* vrow = vt_ + (dt * 32 + l31) * VROWB + hi * 8;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        union { uint2 h[2]; bf16x8 v; } vf;
        vf.h[0] = *reinterpret_cast<const uint2*>(vrow + s * 32);
        vf.h[1] = *reinterpret_cast<const uint2*>(vrow + s * 32 + 16);
        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v, pb[s], oacc[dt], 0, 0, 0);
      }
    }
  };

  LN3D_GLOAD(0);
  LN3D_LSTORE(0);
  __syncthreads();

  for (int kb = 0; kb + 1 < nkb; ++kb) {
    const int buf = kb & 1;
    LN3D_GLOAD(kb + 1);
    __builtin_amdgcn_sched_barrier(0);
    process(kb, buf, std::false_type{});
    __builtin_amdgcn_sched_barrier(0);
    LN3D_LSTORE(buf ^ 1);
    __syncthreads();
  }
  if ((p.Nk & (KVB - 1)) != 0) process(nkb - 1, (nkb - 1) & 1, std::true_type{});
  else process(nkb - 1, (nkb - 1) & 1, std::false_type{});

  // ---- epilogue: O[b, q, h*DH + d] = O^T[d, q] / l
  float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int q = q0 + l31;
  if (q < p.Nq) {
    const int b = bh / p.H, h = bh - b * p.H;
    bf16_t* op = p.O + ((int64_t)b * p.Nq + q) * p.ldo + h * DH + 4 * hi;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 o;
        o.x = pack2bf(oacc[dt][4 * g + 0] * inv, oacc[dt][4 * g + 1] * inv);
        o.y = pack2bf(oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(op + dt * 32 + 8 * g) = o;
      }
  }
}

template <int DH>
static int launch_attn(const AttnP& p, hipStream_t s) {
  constexpr int LDS = 2 * (KVB * (DH * 2 + 16)) + 2 * (DH * VROWB);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<DH>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  dim3 grid((p.Nq + 127) / 128, p.B * p.H);
  hipLaunchKernelGGL(attn_kernel<DH>, grid, dim3(256), LDS, s, p);
  return ln3d_check_launch();
}

extern "C" int ln3d_attention_bf16(const ln3d_attn_args* a, void* stream) {
  if (!a || !a->Q || !a->K || !a->Vt || !a->O) return LN3D_ERR_BAD_ARG;
  if (a->Nk <= 0 || a->Nq <= 0 || a->Nk_pad % 64 != 0 || a->Nk_pad < a->Nk || a->Nq_pad < a->Nq) return LN3D_ERR_BAD_ARG;
  AttnP p;
  p.Q = (const bf16_t*)a->Q; p.K = (const bf16_t*)a->K; p.Vt = (const bf16_t*)a->Vt; p.O = (bf16_t*)a->O;
  p.B = a->B; p.H = a->H; p.Nq = a->Nq; p.Nq_pad = a->Nq_pad; p.Nk = a->Nk; p.Nk_pad = a->Nk_pad;
  p.ldo = a->ldo;
  p.scale_log2 = a->scale * 1.4426950408889634f;
  hipStream_t s = (hipStream_t)stream;
  if (a->Dh == 64) return launch_attn<64>(p, s);
  if (a->Dh == 128) return launch_attn<128>(p, s);
  return LN3D_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------
// per-head RMSNorm on q / k (qk_norm): rows of Dh bf16, in place; one 16-lane group per row (Dh=64)
__global__ void rmsnorm_heads_kernel(bf16_t* x, const float* w, int64_t rows, int Dh, float eps) {
  const int per = Dh / 4;                          // lanes per row, 4 elements each
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = gid / per;
  const int c = (int)(gid % per);
  if (row >= rows) return;
  bf16_t* px = x + row * Dh + c * 4;
  const uint2 raw = *reinterpret_cast<const uint2*>(px);
  float v0 = bf2f(raw.x & 0xffff), v1 = bf2f(raw.x >> 16), v2 = bf2f(raw.y & 0xffff), v3 = bf2f(raw.y >> 16);
  float ss = v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
  for (int o = per >> 1; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  const float rs = rsqrtf(ss / Dh + eps);
  const float4 ww = *reinterpret_cast<const float4*>(w + c * 4);
  uint2 o;
  o.x = pack2bf(v0 * rs * ww.x, v1 * rs * ww.y);
  o.y = pack2bf(v2 * rs * ww.z, v3 * rs * ww.w);
  *reinterpret_cast<uint2*>(px) = o;
}

extern "C" int ln3d_rmsnorm_heads_bf16(void* x, const float* w, int64_t rows, int Dh, float eps, void* stream) {
  if (!x || !w || (Dh != 64 && Dh != 128)) return LN3D_ERR_BAD_ARG;
  const int per = Dh / 4;
  const int64_t threads = rows * per;
  hipLaunchKernelGGL(rmsnorm_heads_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (bf16_t*)x, w, rows, Dh, eps);
  return ln3d_check_launch();
}
