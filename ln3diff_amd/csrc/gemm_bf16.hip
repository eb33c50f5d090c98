// bf16 GEMM with fused epilogues for gfx950 (CDNA4): out[m,n] = epi(sum_k X[m,k] W[n,k] + bias[n]).
//
// Design (MI355X-first):
//  * MFMA v_mfma_f32_32x32x16_bf16, 64-lane wavefronts.  The MFMA "A" operand is the WEIGHT tile
//    (rows = output features) and the "B" operand the TOKEN tile, so the fp32 accumulator fragment
//    of a lane holds 4 consecutive output FEATURES of one token ( row = (r&3)+8(r>>2)+4(lane>>5),
//    col = lane&31 ) -> every epilogue store is a contiguous 8-byte (bf16) / 16-byte (f32) piece of
//    an output row, and per-feature bias / per-(sample,feature) gates are 4-wide vector loads.
//  * 128(features) x 128(tokens) x 64(K) block tile, 4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles
//    (64 accumulator VGPRs).  Both operands are K-contiguous in HBM ([rows, K] row-major), staged
//    HBM -> VGPR (global_load_dwordx4, 8 lanes cover one 128-B row segment) -> LDS rows padded to
//    144 B, which makes the ds_read_b128 fragment reads bank-conflict free (16 distinct rows of a
//    b128 lane-group land on 16 distinct 16-B slots: 144*r mod 256).  Double-buffered LDS, the next
//    tile's global loads are issued before the MFMA block of the current tile (latency hidden under
//    16 MFMAs/wave), one barrier per K-tile.
//  * Epilogues fused: bias, GELU(erf/tanh), SiLU, gate*out+residual (adaLN-zero gating into the fp32
//    residual stream, optional bf16 copy), head split with V^T emission for the attention kernel.
#include <stdlib.h>
#include "common.h"
#include "../../include/ln3d.h"

#define BMF 128  // features per block
#define BTK 128  // tokens per block
#define BK 64
#define ROWB 144                      // LDS bytes per tile row (128 + 16 pad)
#define TILEB (128 * ROWB)            // 18432
#define GEMM_LDS (4 * TILEB)          // 2 buffers x (W tile + X tile)

struct GemmP {
  const bf16_t* X; const bf16_t* W; const float* bias;
  int64_t ldx, ldw, ldo;
  int M, N, K;
  void* out0; void* out1; void* out2;
  const float* gate; int gate_rows; int64_t gate_ld;
  int tokens, tok_pad, heads, head_dim, transpose_mask, head_dim_pad;
  int abl;   // bench-only ablation bits for the 256-wide kernel (LN3D_GEMM_ABL): 1 = skip the epilogue, 2 = 4 K-stages only
};

template <int EPI>
__device__ __forceinline__ void epilogue4(const GemmP& p, int tok, int fb, float v0, float v1, float v2, float v3) {
  // 4 consecutive features fb..fb+3 of token `tok` (all in range, fb % 4 == 0)
  if (p.bias) {
    const float4 b = *reinterpret_cast<const float4*>(p.bias + fb);
    v0 += b.x; v1 += b.y; v2 += b.z; v3 += b.w;
  }
  if constexpr (EPI == LN3D_EPI_F32) {
    *reinterpret_cast<float4*>((float*)p.out0 + (int64_t)tok * p.ldo + fb) = make_float4(v0, v1, v2, v3);
  } else if constexpr (EPI == LN3D_EPI_BF16 || EPI == LN3D_EPI_GELU_ERF || EPI == LN3D_EPI_GELU_TANH ||
                       EPI == LN3D_EPI_SILU) {
    if constexpr (EPI == LN3D_EPI_GELU_ERF) { v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3); }
    if constexpr (EPI == LN3D_EPI_GELU_TANH) { v0 = gelu_tanh(v0); v1 = gelu_tanh(v1); v2 = gelu_tanh(v2); v3 = gelu_tanh(v3); }
    if constexpr (EPI == LN3D_EPI_SILU) { v0 = silu(v0); v1 = silu(v1); v2 = silu(v2); v3 = silu(v3); }
    uint2 o; o.x = pack2bf(v0, v1); o.y = pack2bf(v2, v3);
    *reinterpret_cast<uint2*>((bf16_t*)p.out0 + (int64_t)tok * p.ldo + fb) = o;
  } else if constexpr (EPI == LN3D_EPI_F32_SILU) {
    *reinterpret_cast<float4*>((float*)p.out0 + (int64_t)tok * p.ldo + fb) = make_float4(v0, v1, v2, v3);
    uint2 o; o.x = pack2bf(silu(v0), silu(v1)); o.y = pack2bf(silu(v2), silu(v3));
    *reinterpret_cast<uint2*>((bf16_t*)p.out1 + (int64_t)tok * p.ldo + fb) = o;
  } else if constexpr (EPI == LN3D_EPI_GATE_RES) {
    if (p.gate) {
      const float4 g = *reinterpret_cast<const float4*>(p.gate + (int64_t)(tok / p.gate_rows) * p.gate_ld + fb);
      v0 *= g.x; v1 *= g.y; v2 *= g.z; v3 *= g.w;
    }
    float4* xp = reinterpret_cast<float4*>((float*)p.out0 + (int64_t)tok * p.ldo + fb);
    float4 x = *xp;
    x.x += v0; x.y += v1; x.z += v2; x.w += v3;
    *xp = x;
    if (p.out1) {
      uint2 o; o.x = pack2bf(x.x, x.y); o.y = pack2bf(x.z, x.w);
      *reinterpret_cast<uint2*>((bf16_t*)p.out1 + (int64_t)tok * p.ldo + fb) = o;
    }
  } else if constexpr (EPI == LN3D_EPI_HEADS) {
    const int dm = p.heads * p.head_dim;
    const int which = fb / dm;
    const int rem = fb - which * dm;
    const int h = rem / p.head_dim, d = rem - h * p.head_dim;
    const int b = tok / p.tokens, t = tok - b * p.tokens;
    bf16_t* dst = (bf16_t*)(which == 0 ? p.out0 : (which == 1 ? p.out1 : p.out2));
    const int64_t bh = (int64_t)b * p.heads + h;
    if (!((p.transpose_mask >> which) & 1)) {
      uint2 o; o.x = pack2bf(v0, v1); o.y = pack2bf(v2, v3);
      *reinterpret_cast<uint2*>(dst + (bh * p.tok_pad + t) * p.head_dim_pad + d) = o;
    } else {
      // V^T: tokens of every 16-group stored in the order [0-3, 8-11, 4-7, 12-15] (bits 2 and 3 of t swapped) - the
      // order in which the attention kernel's S^T accumulator hands P to the PV MFMA (csrc/attention.hip)
      const int tp = (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1);
      bf16_t* q = dst + (bh * p.head_dim_pad + d) * p.tok_pad + tp;
      q[0] = f2bf(v0); q[p.tok_pad] = f2bf(v1); q[2 * (int64_t)p.tok_pad] = f2bf(v2); q[3 * (int64_t)p.tok_pad] = f2bf(v3);
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wf = wid >> 1, wt = wid & 1;
  const int l31 = lane & 31, hi = lane >> 5;

  const int nft = (p.N + BMF - 1) / BMF;
  const int ft = blockIdx.x % nft, tt = blockIdx.x / nft;
  const int f0 = ft * BMF, t0 = tt * BTK;

  // staging map: 8 lanes x 16 B cover one 128-B row segment; 32 rows per pass, 4 passes
  const int c = tid & 7, r0 = tid >> 3;
  const bf16_t* wsrc[4]; const bf16_t* xsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int rf = f0 + r0 + 32 * i; rf = rf < p.N ? rf : p.N - 1;
    int rt = t0 + r0 + 32 * i; rt = rt < p.M ? rt : p.M - 1;
    wsrc[i] = p.W + (int64_t)rf * p.ldw + c * 8;
    xsrc[i] = p.X + (int64_t)rt * p.ldx + c * 8;
  }
  const int st_off = r0 * ROWB + c * 16;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  uint4 ra[4], rb[4];
  const int nk = p.K / BK;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ra[i] = *reinterpret_cast<const uint4*>(wsrc[i]);
    rb[i] = *reinterpret_cast<const uint4*>(xsrc[i]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    *reinterpret_cast<uint4*>(smem + st_off + i * 32 * ROWB) = ra[i];
    *reinterpret_cast<uint4*>(smem + TILEB + st_off + i * 32 * ROWB) = rb[i];
  }
  __syncthreads();

  const int a_off = (wf * 64 + l31) * ROWB + hi * 16;
  const int b_off = TILEB + (wt * 64 + l31) * ROWB + hi * 16;

  auto mma_tile = [&](const char* base) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = *reinterpret_cast<const bf16x8*>(base + a_off + i * 32 * ROWB + ks * 32);
        b[i] = *reinterpret_cast<const bf16x8*>(base + b_off + i * 32 * ROWB + ks * 32);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  };

  for (int kt = 0; kt + 1 < nk; ++kt) {
    const int buf = kt & 1;
    const int koff = (kt + 1) * BK;
    // explicit scalars (not arrays): keeps the in-flight tile in VGPRs across the sched_barrier
    const uint4 na0 = *reinterpret_cast<const uint4*>(wsrc[0] + koff);
    const uint4 na1 = *reinterpret_cast<const uint4*>(wsrc[1] + koff);
    const uint4 na2 = *reinterpret_cast<const uint4*>(wsrc[2] + koff);
    const uint4 na3 = *reinterpret_cast<const uint4*>(wsrc[3] + koff);
    const uint4 nx0 = *reinterpret_cast<const uint4*>(xsrc[0] + koff);
    const uint4 nx1 = *reinterpret_cast<const uint4*>(xsrc[1] + koff);
    const uint4 nx2 = *reinterpret_cast<const uint4*>(xsrc[2] + koff);
    const uint4 nx3 = *reinterpret_cast<const uint4*>(xsrc[3] + koff);
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMA block (hipcc sinks it otherwise)
    mma_tile(smem + buf * 2 * TILEB);
    __builtin_amdgcn_sched_barrier(0);
    char* nb = smem + (buf ^ 1) * 2 * TILEB + st_off;
    *reinterpret_cast<uint4*>(nb + 0 * 32 * ROWB) = na0;
    *reinterpret_cast<uint4*>(nb + 1 * 32 * ROWB) = na1;
    *reinterpret_cast<uint4*>(nb + 2 * 32 * ROWB) = na2;
    *reinterpret_cast<uint4*>(nb + 3 * 32 * ROWB) = na3;
    *reinterpret_cast<uint4*>(nb + TILEB + 0 * 32 * ROWB) = nx0;
    *reinterpret_cast<uint4*>(nb + TILEB + 1 * 32 * ROWB) = nx1;
    *reinterpret_cast<uint4*>(nb + TILEB + 2 * 32 * ROWB) = nx2;
    *reinterpret_cast<uint4*>(nb + TILEB + 3 * 32 * ROWB) = nx3;
    __syncthreads();
  }
  mma_tile(smem + ((nk - 1) & 1) * 2 * TILEB);

#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int tok = t0 + wt * 64 + j * 32 + l31;
    if (tok >= p.M) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int fb = f0 + wf * 64 + i * 32 + 8 * g + 4 * hi;
        if (fb < p.N)
          epilogue4<EPI>(p, tok, fb, acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
      }
    }
  }
}

// =================================================================================================
// ------------------------------------------------------------------------------------------------------------------
// Epilogue of the LDS-DMA kernels.  A lane's accumulator quad is 4 consecutive features of ONE token and lanes 0-31 are 32
// different tokens, so storing straight from the accumulators writes 16-byte pieces scattered over 32 output rows per
// instruction (measured: ~1.5 TB/s, a third of a K = 1024 GEMM).  Instead every wave transposes its own sub-tile through a
// private 8 KB fp32 LDS region (the ring is free after the main loop), 32 tokens x 64 features at a time, and comes back
// with 16 consecutive lanes holding the 64 consecutive features of one token: each store instruction then writes four
// complete 128-byte (bf16) / 256-byte (f32) row segments and the bias / gate / residual traffic is row-contiguous too.
// 256-byte staging rows, 16-byte chunk c of row r stored at chunk c ^ (r & 15): conflict-free ds_write_b128 (8-lane
// groups = 8 rows) and ds_read_b128 (16-lane groups) without padding.  DS operations of one wave execute in order, so
// no barrier is needed beyond the caller's one that retires the ring.
template <int EPI, int NI, int NJ>
__device__ __forceinline__ void staged_epilogue(const GemmP& p, f32x16 (&acc)[NI][NJ], char* stg, int fw0, int tw0, int lane) {
  static_assert(NI % 2 == 0, "feature blocks are staged in pairs");
  const int l31 = lane & 31, hi = lane >> 5;
  const int rrow = lane >> 4, rc = lane & 15;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
#pragma unroll
    for (int ih = 0; ih < NI / 2; ++ih) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = ii * 8 + 2 * g + hi;
          *reinterpret_cast<float4*>(stg + l31 * 256 + ((c ^ (l31 & 15)) << 4)) =
              make_float4(acc[2 * ih + ii][j][4 * g + 0], acc[2 * ih + ii][j][4 * g + 1], acc[2 * ih + ii][j][4 * g + 2],
                          acc[2 * ih + ii][j][4 * g + 3]);
        }
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = 4 * it + rrow;
        const float4 v = *reinterpret_cast<const float4*>(stg + row * 256 + ((rc ^ (row & 15)) << 4));
        const int tok = tw0 + j * 32 + row, fb = fw0 + ih * 64 + 4 * rc;
        if (tok < p.M && fb < p.N) epilogue4<EPI>(p, tok, fb, v.x, v.y, v.z, v.w);
      }
    }
  }
}

// Large-tile variant: 128(features) x 384(tokens) x 32(K) stages, 8 waves (2 x 4), wave tile 64f x 96t
// (2 x 3 MFMA 32x32x16 tiles, 96 accumulator registers), 4-deep LDS ring filled by LDS-DMA
// (global_load_lds_dwordx4: HBM/L2 -> LDS without touching VGPRs), counted vmcnt so that the loads of the
// next two stages stay in flight ACROSS the per-stage barrier (raw s_barrier; __syncthreads would drain them).
//  * LDS rows are 64 B (32 bf16) and XOR-swizzled at 16-B granularity: chunk c of row r lives at position
//    c ^ ((r>>2)&3).  The DMA writes lane-linear (base + 16*lane), so the swizzle is applied to the per-lane
//    SOURCE address and again on the ds_read_b128 side; a b128 lane-group (16 distinct rows) then covers all 16
//    16-B slots of the 256-B bank row: conflict-free without padding.
//  * Tile shape chosen so the DiT GEMMs quantise exactly onto 256 CUs at one 8-wave workgroup per CU:
//    tokens 16*768 = 32 x 384; N = 1024 / 3072 / 4096 -> 256 / 768 / 1024 tiles = 1 / 3 / 4 full rounds.
//  * Workgroup -> tile map is XCD-aware: the 8 XCDs (block b runs on XCD b % 8) each take a contiguous range
//    of tile ids, feature-tile fastest, so the workgroups sharing one token panel hit the same L2.
#define L_BF 128
#define L_BT 384
#define L_BK 32
#define L_STAGES 4
#define L_WB (L_BF * 64)                 // 8192
#define L_XB (L_BT * 64)                 // 24576
#define L_STAGEB (L_WB + L_XB)           // 32768
#define L_LDS (L_STAGES * L_STAGEB)      // 131072

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

// ABL (bench-only ablations, tools/kbench.py): 0 = product kernel, 1 = no LDS-DMA (stale LDS), 2 = no MFMA, 3 = no ds_read
template <int EPI, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_bf16_large_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wf = wid >> 2, wt = wid & 3;
  const int l31 = lane & 31, hi = lane >> 5;

  // XCD-aware bijective tile map
  const int nft = (p.N + L_BF - 1) / L_BF, ntt = (p.M + L_BT - 1) / L_BT;
  const int ntiles = nft * ntt;
  int ft, tt;
  {
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    if (r == 0 && (nft & 7) == 0 && q % nft == 0) {
      // 2-D blocking inside the XCD's chunk: the 32 workgroups that run together on one XCD cover 8 feature tiles x
      // 4 token panels (2 MB of W + 3 MB of X at K = 1024: L2-resident), and successive rounds keep the X panels.
      const int rows = q / nft;                       // token panels owned by this XCD
      const int g = slot / (rows * 8), rem = slot - g * rows * 8;
      ft = g * 8 + (rem & 7);
      tt = xcd * rows + (rem >> 3);
    } else {
      const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
      ft = tile % nft; tt = tile / nft;
    }
  }
  const int f0 = ft * L_BF, t0 = tt * L_BT;

  // LDS-DMA source pointers: instruction j of a tile covers rows [16j, 16j+16); lane -> (row 16j + lane/4,
  // stored position lane%4 holding global chunk (lane%4) ^ ((lane>>4)&3))
  const int lrow = lane >> 2;
  const int lchunk = (lane & 3) ^ ((lane >> 4) & 3);
  int rw = f0 + 16 * wid + lrow; rw = rw < p.N ? rw : p.N - 1;
  const bf16_t* wsrc = p.W + (int64_t)rw * p.ldw + lchunk * 8;
  const bf16_t* xsrc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int rx = t0 + 16 * (wid + 8 * i) + lrow; rx = rx < p.M ? rx : p.M - 1;
    xsrc[i] = p.X + (int64_t)rx * p.ldx + lchunk * 8;
  }
  const int wdst = wid * 1024;
  const int xdst0 = L_WB + wid * 1024, xdst1 = L_WB + (wid + 8) * 1024, xdst2 = L_WB + (wid + 16) * 1024;

#define L_ISSUE(s)                                                                                        \
  {                                                                                                       \
    const int koff_ = (s) * L_BK;                                                                         \
    char* sb_ = smem + ((s) & (L_STAGES - 1)) * L_STAGEB;                                                 \
    __builtin_amdgcn_global_load_lds((glb_void_t*)(wsrc + koff_), (lds_void_t*)(sb_ + wdst), 16, 0, 0);    \
    __builtin_amdgcn_global_load_lds((glb_void_t*)(xsrc[0] + koff_), (lds_void_t*)(sb_ + xdst0), 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((glb_void_t*)(xsrc[1] + koff_), (lds_void_t*)(sb_ + xdst1), 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((glb_void_t*)(xsrc[2] + koff_), (lds_void_t*)(sb_ + xdst2), 16, 0, 0); \
  }

  // fragment read offsets (ks = 0; ks = 1 is the same address ^ 32)
  const int key = (l31 >> 2) & 3;
  const int a_off = (wf * 64 + l31) * 64 + ((hi ^ key) << 4);
  const int b_off = L_WB + (wt * 96 + l31) * 64 + ((hi ^ key) << 4);

  f32x16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Software pipeline, one barrier per 32-deep stage, all 8 waves in step:
  //   iteration s :  wait (counted) until THIS wave's DMAs of stage s+1 landed -> s_barrier (now every wave's have,
  //                  and every wave has finished reading ring slot s) -> issue the DMAs of stage s+4 into slot s ->
  //                  MFMAs on the register fragments of stage s, interleaved with the ds_reads of stage s+1 into
  //                  the other fragment set.
  // The fragments of a stage live in registers one iteration before they are multiplied, so a ring slot is free as
  // soon as it has been read: 3 stages (96 KB per CU) of LDS-DMA stay in flight across the barriers, and the matrix
  // pipe of a SIMD always has the 12 MFMAs of one of its two waves to run while the other issues DMA / LDS reads.
  bf16x8 fa0[2][2], fb0[2][3], fa1[2][2], fb1[2][3];
#define L_READ(s, FA, FB)                                                                              \
  {                                                                                                    \
    const char* sb_ = smem + ((s) & (L_STAGES - 1)) * L_STAGEB;                                        \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                 \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                    \
          FA[ks][i] = *reinterpret_cast<const bf16x8*>(sb_ + ((a_off + i * 32 * 64) ^ (ks << 5)));     \
      _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                    \
          FB[ks][j] = *reinterpret_cast<const bf16x8*>(sb_ + ((b_off + j * 32 * 64) ^ (ks << 5)));     \
    }                                                                                                  \
  }
#define L_MMA(FA, FB)                                                                                  \
  {                                                                                                    \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                   \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                  \
            _Pragma("unroll") for (int j = 0; j < 3; ++j)                                              \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[ks][i], FB[ks][j], acc[i][j], 0, 0, 0); \
  }
  // one pipeline iteration: CUR = fragment set holding stage s, NXT = set receiving stage s+1
#define L_ITER(s, CUR_A, CUR_B, NXT_A, NXT_B)                                                          \
  {                                                                                                    \
    if ((s) + 1 < ns) {                                                                                \
      if ((s) + 3 < ns) { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }                           \
      else if ((s) + 2 < ns) { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }                      \
      else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }                                        \
      __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0): fragment reads of stage s done (compiler-visible) */ \
      __builtin_amdgcn_s_barrier();                                                                    \
      if constexpr (ABL != 1) { if ((s) + 4 < ns) L_ISSUE((s) + 4); }                                  \
      if constexpr (ABL != 3) { L_READ((s) + 1, NXT_A, NXT_B); }                                       \
    }                                                                                                  \
    if constexpr (ABL != 2) { L_MMA(CUR_A, CUR_B); }                                                   \
    else { asm volatile("" ::"v"(CUR_A[0][0]), "v"(CUR_B[1][2])); }                                    \
  }

  const int ns = p.K / L_BK;
  L_ISSUE(0);
  if (ns > 1) L_ISSUE(1);
  if (ns > 2) L_ISSUE(2);
  if (ns > 3) L_ISSUE(3);
  if (ns > 3) { asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
  else if (ns > 2) { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
  else if (ns > 1) { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
  else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  __builtin_amdgcn_s_barrier();                       // stage 0 of every wave has landed
  L_READ(0, fa0, fb0);
  // steady state (branch-free body so hipcc can interleave the ds_reads of stage s+1 under the MFMAs of stage s)
#define L_STEADY(s, CUR_A, CUR_B, NXT_A, NXT_B)                                                        \
  {                                                                                                    \
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                   \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                                                \
    __builtin_amdgcn_s_barrier();                                                                      \
    if constexpr (ABL != 1) { L_ISSUE((s) + 4); }                                                      \
    if constexpr (ABL != 3) { L_READ((s) + 1, NXT_A, NXT_B); }                                         \
    if constexpr (ABL != 2) { L_MMA(CUR_A, CUR_B); }                                                   \
    else { asm volatile("" ::"v"(CUR_A[0][0]), "v"(CUR_B[1][2])); }                                    \
  }
  int s = 0;
  for (; s + 5 < ns; s += 2) {
    L_STEADY(s, fa0, fb0, fa1, fb1);
    L_STEADY(s + 1, fa1, fb1, fa0, fb0);
  }
  for (; s < ns; s += 2) {                            // drain: <= 6 stages, counted waits shrink with the queue
    L_ITER(s, fa0, fb0, fa1, fb1);
    if (s + 1 < ns) L_ITER(s + 1, fa1, fb1, fa0, fb0);
  }

  bool direct = false;                                // V^T emission wants token-contiguous stores: direct path
  if constexpr (EPI == LN3D_EPI_HEADS) direct = (p.transpose_mask != 0) && (f0 + L_BF > 2 * p.heads * p.head_dim);
  if (direct) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int tok = t0 + wt * 96 + j * 32 + l31;
      if (tok >= p.M) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int fb = f0 + wf * 64 + i * 32 + 8 * g + 4 * hi;
          if (fb < p.N)
            epilogue4<EPI>(p, tok, fb, acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
        }
      }
    }
    return;
  }
  __builtin_amdgcn_s_barrier();                       // every wave is done reading the ring
  staged_epilogue<EPI, 2, 3>(p, acc, smem + wid * 8192, f0 + wf * 64, t0 + wt * 96, lane);
}

template <int EPI, int ABL = 0>
static int launch_large(const GemmP& p, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_large_kernel<EPI, ABL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, L_LDS);
    attr_set = true;
  }
  const int nft = (p.N + L_BF - 1) / L_BF, ntt = (p.M + L_BT - 1) / L_BT;
  hipLaunchKernelGGL((gemm_bf16_large_kernel<EPI, ABL>), dim3(nft * ntt), dim3(512), L_LDS, s, p);
  return ln3d_check_launch();
}

// ------------------------------------------------------------------------------------------------------------------
// Generic LDS-DMA ring kernel: NW waves as (NW/WGT) x WGT, wave tile 32*NI features x 32*NJ tokens, K stages of 32,
// 4-deep ring, same swizzle / fragment layout as the 128x384 kernel above.  What the configurations trade is the
// byte/flop ratio of the L2 -> LDS fill (the measured limiter, ~12 TB/s chip-wide) against rounds of 256 tiles:
//   NW 8, 2x4 waves, 4x2 blocks : 256f x 256t, 128 flop/B, 2 waves/SIMD  (N = 4096: 768 tiles = 3 rounds)
//   NW 4, 2x2 waves, 4x4 / 4x3  : 256f x 256t / 192t with one wave per SIMD and 256 / 192 accumulator registers
// The pipeline runs at K-substep (16) granularity with two fragment register sets: substep 0 of a stage multiplies while
// substep 1 is read, the stage barrier sits between them, and the reads of stage s+1 and the DMA issues of stage s+4 are
// slotted one per MFMA behind it (sched_barrier pins the interleave).
template <int EPI, int NW, int WGT, int NI, int NJ>
__global__ __launch_bounds__(NW * 64, NW / 4) void gemm_bf16_ring_kernel(GemmP p) {
  constexpr int WGF = NW / WGT, BF = 32 * NI * WGF, BT = 32 * NJ * WGT;
  constexpr int WB = BF * 64, STAGEB = (BF + BT) * 64, NPW = (BF + BT) / 16 / NW;
  static_assert((BF + BT) / 16 % NW == 0, "DMA instructions must divide evenly over the waves");
  constexpr int NM = NI * NJ, NR = NI + NJ, NFREE = NM - NR;
  static_assert(NFREE >= 0, "one fragment read per MFMA slot");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wf = wid / WGT, wt = wid % WGT;
  const int l31 = lane & 31, hi = lane >> 5;

  // XCD-aware tile map: an XCD owns ntt/8 token panels and walks the feature tiles in groups of 4, so the <= 32 tiles
  // that run together on it share 4 W panels and its own X panels and X is fetched once per XCD.
  const int nft = (p.N + BF - 1) / BF, ntt = (p.M + BT - 1) / BT;
  const int ntiles = nft * ntt;
  int ft, tt;
  {
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
    if ((ntt & 7) == 0 && (nft & 3) == 0) {
      const int rows = ntt >> 3;
      const int g = slot / (rows * 4), rem = slot - g * rows * 4;
      ft = g * 4 + (rem & 3);
      tt = xcd * rows + (rem >> 2);
    } else {
      const int q = ntiles >> 3, r = ntiles & 7;
      const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
      ft = tile % nft; tt = tile / nft;
    }
  }
  const int f0 = ft * BF, t0 = tt * BT;

  // DMA instruction idx (16 rows x 64 B each; W rows first, then X rows) -> LDS bytes [idx*1024, +1024) of the stage;
  // wave w issues idx = w*NPW .. w*NPW+NPW-1
  const int lrow = lane >> 2;
  const int lchunk = (lane & 3) ^ ((lane >> 4) & 3);
  const bf16_t* src[NPW];
#pragma unroll
  for (int q = 0; q < NPW; ++q) {
    const int idx = wid * NPW + q;
    if (idx < BF / 16) {
      int r = f0 + 16 * idx + lrow; r = r < p.N ? r : p.N - 1;
      src[q] = p.W + (int64_t)r * p.ldw + lchunk * 8;
    } else {
      int r = t0 + 16 * (idx - BF / 16) + lrow; r = r < p.M ? r : p.M - 1;
      src[q] = p.X + (int64_t)r * p.ldx + lchunk * 8;
    }
  }
  const int dst0 = wid * NPW * 1024;
#define X_ISSUE1(s, q)                                                                                       \
  __builtin_amdgcn_global_load_lds((glb_void_t*)(src[q] + (s) * L_BK),                                        \
      (lds_void_t*)(smem + ((s) & 3) * STAGEB + dst0 + (q) * 1024), 16, 0, 0)

  const int key = (l31 >> 2) & 3;
  const int a_off = (wf * 32 * NI + l31) * 64 + ((hi ^ key) << 4);
  const int b_off = WB + (wt * 32 * NJ + l31) * 64 + ((hi ^ key) << 4);
#define X_RDA(s, ks, i) (*reinterpret_cast<const bf16x8*>(smem + ((s) & 3) * STAGEB + ((a_off + (i) * 2048) ^ ((ks) << 5))))
#define X_RDB(s, ks, j) (*reinterpret_cast<const bf16x8*>(smem + ((s) & 3) * STAGEB + ((b_off + (j) * 2048) ^ ((ks) << 5))))

  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  bf16x8 a0[NI], b0[NJ], a1[NI], b1[NJ];
  const int ns = (p.abl & 2) ? 4 : p.K / L_BK;

  // prologue: up to 4 stages in flight, wait for the first
#pragma unroll
  for (int q = 0; q < NPW; ++q) X_ISSUE1(0, q);
  if (ns > 1) { _Pragma("unroll") for (int q = 0; q < NPW; ++q) X_ISSUE1(1, q); }
  if (ns > 2) { _Pragma("unroll") for (int q = 0; q < NPW; ++q) X_ISSUE1(2, q); }
  if (ns > 3) { _Pragma("unroll") for (int q = 0; q < NPW; ++q) X_ISSUE1(3, q); }
  if (ns > 3) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NPW) : "memory"); }
  else if (ns > 2) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPW) : "memory"); }
  else if (ns > 1) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory"); }
  else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < NI; ++i) a0[i] = X_RDA(0, 0, i);
#pragma unroll
  for (int j = 0; j < NJ; ++j) b0[j] = X_RDB(0, 0, j);

  // MFMA n of a substep is block (n / NJ, n % NJ); behind MFMA n one or more memory instructions are slotted in
#define X_MMA(FA, FB, n) acc[(n) / NJ][(n) % NJ] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[(n) / NJ], FB[(n) % NJ], acc[(n) / NJ][(n) % NJ], 0, 0, 0)
#define X_PIN() __builtin_amdgcn_sched_barrier(0)
  // one stage; FILL / MORE / WAITN are literals so the body is branch-free
#define X_STAGE(s, FILL, MORE, WAITN)                                                                     \
  {                                                                                                       \
    /* substep 0: multiply set 0, read substep 1 of this stage into set 1 */                              \
    _Pragma("unroll") for (int n = 0; n < NM; ++n) {                                                      \
      X_MMA(a0, b0, n);                                                                                   \
      if (n < NI) a1[n] = X_RDA(s, 1, n);                                                                 \
      else if (n < NR) b1[n - NI] = X_RDB(s, 1, n - NI);                                                  \
      X_PIN();                                                                                            \
    }                                                                                                     \
    if (MORE) {                                                                                           \
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");                                        \
      __builtin_amdgcn_s_waitcnt(0xC07F); /* set-1 fragments in registers; this wave is done with slot s */ \
      __builtin_amdgcn_s_barrier();                                                                       \
    }                                                                                                     \
    /* substep 1: multiply set 1; behind it substep 0 of stage s+1 and the DMAs of stage s+4 (slot just released) */ \
    _Pragma("unroll") for (int n = 0; n < NM; ++n) {                                                      \
      X_MMA(a1, b1, n);                                                                                   \
      if (n < NI) { if (MORE) a0[n] = X_RDA((s) + 1, 0, n); }                                             \
      else if (n < NR) { if (MORE) b0[n - NI] = X_RDB((s) + 1, 0, n - NI); }                              \
      if (FILL) {                                                                                         \
        _Pragma("unroll") for (int d = 0; d < NPW; ++d)                                                   \
            if ((NFREE > 0 ? NR + d * NFREE / NPW : NM - 1) == n) X_ISSUE1((s) + 4, d);                   \
      }                                                                                                   \
      X_PIN();                                                                                            \
    }                                                                                                     \
  }
  int s = 0;
  for (; s + 4 < ns; ++s) X_STAGE(s, true, true, 2 * NPW);
  if (ns >= 4) { X_STAGE(s, false, true, 2 * NPW); ++s; }
  if (ns >= 3) { X_STAGE(s, false, true, NPW); ++s; }
  if (ns >= 2) { X_STAGE(s, false, true, 0); ++s; }
  X_STAGE(s, false, false, 0);

  if ((p.abl & 1) && acc[0][0][0] != 12345.f) return;
  bool direct = false;                                // V^T emission wants token-contiguous stores: direct path
  if constexpr (EPI == LN3D_EPI_HEADS) direct = (p.transpose_mask != 0) && (f0 + BF > 2 * p.heads * p.head_dim);
  if (direct) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int tok = t0 + wt * 32 * NJ + j * 32 + l31;
      if (tok >= p.M) continue;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int fb = f0 + wf * 32 * NI + i * 32 + 8 * g + 4 * hi;
          if (fb < p.N)
            epilogue4<EPI>(p, tok, fb, acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
        }
      }
    }
    return;
  }
  __builtin_amdgcn_s_barrier();                       // every wave is done reading the ring
  staged_epilogue<EPI, NI, NJ>(p, acc, smem + wid * 8192, f0 + wf * 32 * NI, t0 + wt * 32 * NJ, lane);
}

template <int EPI, int NW, int WGT, int NI, int NJ>
static int launch_ring(const GemmP& p, hipStream_t s) {
  constexpr int BF = 32 * NI * (NW / WGT), BT = 32 * NJ * WGT, LDSB = 4 * (BF + BT) * 64;
  static_assert(LDSB >= NW * 8192, "staging regions live in the ring");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_ring_kernel<EPI, NW, WGT, NI, NJ>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
    attr_set = true;
  }
  const int nft = (p.N + BF - 1) / BF, ntt = (p.M + BT - 1) / BT;
  hipLaunchKernelGGL((gemm_bf16_ring_kernel<EPI, NW, WGT, NI, NJ>), dim3(nft * ntt), dim3(NW * 64), LDSB, s, p);
  return ln3d_check_launch();
}

// ------------------------------------------------------------------------------------------------------------------
// Same tile family with K stages of 64: an LDS row is a full 128-byte cache line of the operand (a 64-byte row makes every
// DMA request touch half a line, and the other half is requested again one stage later), two 64 KB slots.  Swizzle for
// 128-byte rows: 16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 7) (the attention kernel's K layout).
// Stage = 4 K-substeps; the fragment sets alternate per substep; the barrier sits before substep 3, behind which the
// first fragments of stage s+1 are read and the DMAs of stage s+2 are issued into the slot that substep 2 finished reading.
template <int EPI, int NW, int WGT, int NI, int NJ>
__global__ __launch_bounds__(NW * 64, NW / 4) void gemm_bf16_ring64_kernel(GemmP p) {
  constexpr int WGF = NW / WGT, BF = 32 * NI * WGF, BT = 32 * NJ * WGT;
  constexpr int WB = BF * 128, STAGEB = (BF + BT) * 128, NPW = (BF + BT) / 8 / NW;
  static_assert((BF + BT) / 8 % NW == 0, "DMA instructions must divide evenly over the waves");
  constexpr int NM = NI * NJ, NR = NI + NJ;
  static_assert(NM >= NR, "one fragment read per MFMA slot");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wf = wid / WGT, wt = wid % WGT;
  const int l31 = lane & 31, hi = lane >> 5;

  const int nft = (p.N + BF - 1) / BF, ntt = (p.M + BT - 1) / BT;
  const int ntiles = nft * ntt;
  int ft, tt;
  {
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
    if ((ntt & 7) == 0 && (nft & 3) == 0) {
      const int rows = ntt >> 3;
      const int g = slot / (rows * 4), rem = slot - g * rows * 4;
      ft = g * 4 + (rem & 3);
      tt = xcd * rows + (rem >> 2);
    } else {
      const int q = ntiles >> 3, r = ntiles & 7;
      const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
      ft = tile % nft; tt = tile / nft;
    }
  }
  const int f0 = ft * BF, t0 = tt * BT;

  // DMA instruction idx = 8 rows x 128 B (W rows first, then X rows) -> LDS bytes [idx*1024, +1024) of the slot
  const int r8 = lane >> 3;
  const bf16_t* src[NPW];
#pragma unroll
  for (int q = 0; q < NPW; ++q) {
    const int idx = wid * NPW + q;
    const int rt = 8 * (idx < BF / 8 ? idx : idx - BF / 8) + r8;          // row inside its own tile
    const int chunk = (lane & 7) ^ ((rt >> 1) & 7);
    if (idx < BF / 8) {
      int r = f0 + rt; r = r < p.N ? r : p.N - 1;
      src[q] = p.W + (int64_t)r * p.ldw + chunk * 8;
    } else {
      int r = t0 + rt; r = r < p.M ? r : p.M - 1;
      src[q] = p.X + (int64_t)r * p.ldx + chunk * 8;
    }
  }
  const int dst0 = wid * NPW * 1024;
#define Y_ISSUE1(s, q)                                                                                       \
  __builtin_amdgcn_global_load_lds((glb_void_t*)(src[q] + (s) * 64),                                          \
      (lds_void_t*)(smem + ((s) & 1) * STAGEB + dst0 + (q) * 1024), 16, 0, 0)

  const int key = (l31 >> 1) & 7;
  const int a_row = (wf * 32 * NI + l31) * 128;
  const int b_row = WB + (wt * 32 * NJ + l31) * 128;
#define Y_RDA(s, ks, i) (*reinterpret_cast<const bf16x8*>(smem + ((s) & 1) * STAGEB + a_row + (i) * 4096 + (((2 * (ks) + hi) ^ key) << 4)))
#define Y_RDB(s, ks, j) (*reinterpret_cast<const bf16x8*>(smem + ((s) & 1) * STAGEB + b_row + (j) * 4096 + (((2 * (ks) + hi) ^ key) << 4)))

  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  bf16x8 a0[NI], b0[NJ], a1[NI], b1[NJ];
  const int ns = (p.abl & 2) ? 2 : p.K / 64;

#pragma unroll
  for (int q = 0; q < NPW; ++q) Y_ISSUE1(0, q);
  if (ns > 1) { _Pragma("unroll") for (int q = 0; q < NPW; ++q) Y_ISSUE1(1, q); }
  if (ns > 1) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory"); }
  else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < NI; ++i) a0[i] = Y_RDA(0, 0, i);
#pragma unroll
  for (int j = 0; j < NJ; ++j) b0[j] = Y_RDB(0, 0, j);

#define Y_MMA(FA, FB, n) acc[(n) / NJ][(n) % NJ] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[(n) / NJ], FB[(n) % NJ], acc[(n) / NJ][(n) % NJ], 0, 0, 0)
  // substep: multiply (CA, CB) while (s2, ks2) is read into (NA, NB)
#define Y_PHASE(CA, CB, NA, NB, s2, ks2, RD)                                                              \
  _Pragma("unroll") for (int n = 0; n < NM; ++n) {                                                        \
    Y_MMA(CA, CB, n);                                                                                     \
    if (RD) {                                                                                             \
      if (n < NI) NA[n] = Y_RDA(s2, ks2, n);                                                              \
      else if (n < NR) NB[n - NI] = Y_RDB(s2, ks2, n - NI);                                               \
    }                                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
  }
#define Y_STAGE(s, FILL, MORE)                                                                            \
  {                                                                                                       \
    Y_PHASE(a0, b0, a1, b1, s, 1, true);                                                                  \
    Y_PHASE(a1, b1, a0, b0, s, 2, true);                                                                  \
    Y_PHASE(a0, b0, a1, b1, s, 3, true);                                                                  \
    if (MORE) {                                                                                           \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   /* stage s+1 (the only DMAs in flight) landed */  \
      __builtin_amdgcn_s_waitcnt(0xC07F);                                                                 \
      __builtin_amdgcn_s_barrier();                                                                       \
    }                                                                                                     \
    _Pragma("unroll") for (int n = 0; n < NM; ++n) {                                                      \
      Y_MMA(a1, b1, n);                                                                                   \
      if (MORE) {                                                                                         \
        if (n < NI) a0[n] = Y_RDA((s) + 1, 0, n);                                                         \
        else if (n < NR) b0[n - NI] = Y_RDB((s) + 1, 0, n - NI);                                          \
      }                                                                                                   \
      if (FILL) {                                                                                         \
        _Pragma("unroll") for (int d = 0; d < NPW; ++d)                                                   \
            if (d * NM / NPW == n) Y_ISSUE1((s) + 2, d);                                                  \
      }                                                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                  \
    }                                                                                                     \
  }
  int s = 0;
  for (; s + 2 < ns; ++s) Y_STAGE(s, true, true);
  if (ns >= 2) { Y_STAGE(s, false, true); ++s; }
  Y_STAGE(s, false, false);

  if ((p.abl & 1) && acc[0][0][0] != 12345.f) return;
  bool direct = false;
  if constexpr (EPI == LN3D_EPI_HEADS) direct = (p.transpose_mask != 0) && (f0 + BF > 2 * p.heads * p.head_dim);
  if (direct) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int tok = t0 + wt * 32 * NJ + j * 32 + l31;
      if (tok >= p.M) continue;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int fb = f0 + wf * 32 * NI + i * 32 + 8 * g + 4 * hi;
          if (fb < p.N)
            epilogue4<EPI>(p, tok, fb, acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
        }
      }
    }
    return;
  }
  __builtin_amdgcn_s_barrier();
  staged_epilogue<EPI, NI, NJ>(p, acc, smem + wid * 8192, f0 + wf * 32 * NI, t0 + wt * 32 * NJ, lane);
}

template <int EPI, int NW, int WGT, int NI, int NJ>
static int launch_ring64(const GemmP& p, hipStream_t s) {
  constexpr int BF = 32 * NI * (NW / WGT), BT = 32 * NJ * WGT, LDSB = 2 * (BF + BT) * 128;
  static_assert(LDSB >= NW * 8192, "staging regions live in the ring");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_ring64_kernel<EPI, NW, WGT, NI, NJ>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
    attr_set = true;
  }
  const int nft = (p.N + BF - 1) / BF, ntt = (p.M + BT - 1) / BT;
  hipLaunchKernelGGL((gemm_bf16_ring64_kernel<EPI, NW, WGT, NI, NJ>), dim3(nft * ntt), dim3(NW * 64), LDSB, s, p);
  return ln3d_check_launch();
}

// cfg: 3 / 4 = one wave per SIMD 256x192 / 256x256; 5 = 8 waves 256x256; 6 = 8 waves 128x384 (the hand-scheduled kernel's tile)
template <int EPI>
static int launch_xl_any(const GemmP& p, hipStream_t s, int cfg) {
  switch (cfg) {
    case 3: return launch_ring<EPI, 4, 2, 4, 3>(p, s);
    case 4: return launch_ring<EPI, 4, 2, 4, 4>(p, s);
    case 5: return launch_ring<EPI, 8, 4, 4, 2>(p, s);
    case 7: return launch_ring64<EPI, 8, 4, 4, 2>(p, s);
    case 8: return launch_ring64<EPI, 8, 4, 2, 3>(p, s);
    case 9: return launch_ring64<EPI, 8, 2, 2, 3>(p, s);
    case 10: return launch_ring64<EPI, 8, 4, 4, 3>(p, s);
    default: return launch_ring<EPI, 8, 4, 2, 3>(p, s);
  }
}

template <int EPI>
static int launch(const GemmP& p, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<EPI>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
    attr_set = true;
  }
  const int nft = (p.N + BMF - 1) / BMF, ntt = (p.M + BTK - 1) / BTK;
  hipLaunchKernelGGL(gemm_bf16_kernel<EPI>, dim3(nft * ntt), dim3(256), GEMM_LDS, s, p);
  return ln3d_check_launch();
}

extern "C" int ln3d_gemm_bf16(const ln3d_gemm_args* a, void* stream) {
  if (!a || !a->X || !a->W || !a->out0) return LN3D_ERR_BAD_ARG;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0 || (a->K % BK) != 0 || (a->N % 4) != 0) return LN3D_ERR_BAD_ARG;
  if ((a->ldx % 8) != 0 || (a->ldw % 8) != 0) return LN3D_ERR_BAD_ARG;
  GemmP p;
  p.X = (const bf16_t*)a->X; p.W = (const bf16_t*)a->W; p.bias = a->bias;
  p.ldx = a->ldx; p.ldw = a->ldw; p.ldo = a->ldo;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.out0 = a->out0; p.out1 = a->out1; p.out2 = a->out2;
  p.gate = a->gate; p.gate_rows = a->gate_rows > 0 ? a->gate_rows : 1; p.gate_ld = a->gate_ld;
  p.tokens = a->tokens; p.tok_pad = a->tok_pad; p.heads = a->heads; p.head_dim = a->head_dim;
  p.transpose_mask = a->transpose_mask;
  p.head_dim_pad = a->head_dim_pad > 0 ? a->head_dim_pad : a->head_dim;
  { const char* e = getenv("LN3D_GEMM_ABL"); p.abl = e ? atoi(e) : 0; }
  hipStream_t s = (hipStream_t)stream;
  // tile selection: the 128x384 LDS-DMA kernel when the problem fills its tiles, the 128x128 kernel otherwise
  // (small M such as the per-sample adaLN / timestep GEMMs, narrow N such as the conv decoder's 32/64 channels)
  const char* force = getenv("LN3D_GEMM_TILE");
  bool large = a->M >= 1536 && a->N >= 128;
  if (force && force[0] == 's') large = false;
  if (force && force[0] == 'l') large = true;
  int xl = 0;                                        // 0 = no, 3 / 4 = 256 x 192 / 256 x 256 tile
  if (force && force[0] == 'x') xl = atoi(force + 1);
  if (xl) {
    switch (a->epilogue) {
      case LN3D_EPI_F32: return launch_xl_any<LN3D_EPI_F32>(p, s, xl);
      case LN3D_EPI_BF16: return launch_xl_any<LN3D_EPI_BF16>(p, s, xl);
      case LN3D_EPI_GELU_ERF: return launch_xl_any<LN3D_EPI_GELU_ERF>(p, s, xl);
      case LN3D_EPI_GELU_TANH: return launch_xl_any<LN3D_EPI_GELU_TANH>(p, s, xl);
      case LN3D_EPI_SILU: return launch_xl_any<LN3D_EPI_SILU>(p, s, xl);
      case LN3D_EPI_GATE_RES: return launch_xl_any<LN3D_EPI_GATE_RES>(p, s, xl);
      case LN3D_EPI_F32_SILU:
        if (!a->out1) return LN3D_ERR_BAD_ARG;
        return launch_xl_any<LN3D_EPI_F32_SILU>(p, s, xl);
      case LN3D_EPI_HEADS:
        if (a->tokens <= 0 || a->heads <= 0 || a->head_dim <= 0 || (a->head_dim % 4) != 0 || a->tok_pad < a->tokens)
          return LN3D_ERR_BAD_ARG;
        return launch_xl_any<LN3D_EPI_HEADS>(p, s, xl);
      default: return LN3D_ERR_UNSUPPORTED;
    }
  }
  if (large) {
    switch (a->epilogue) {
      case LN3D_EPI_F32: return launch_large<LN3D_EPI_F32>(p, s);
      case LN3D_EPI_BF16: {
        const char* abl = getenv("LN3D_GEMM_ABL");
        if (abl && abl[0] == '1') return launch_large<LN3D_EPI_BF16, 1>(p, s);
        if (abl && abl[0] == '2') return launch_large<LN3D_EPI_BF16, 2>(p, s);
        if (abl && abl[0] == '3') return launch_large<LN3D_EPI_BF16, 3>(p, s);
        return launch_large<LN3D_EPI_BF16>(p, s);
      }
      case LN3D_EPI_GELU_ERF: return launch_large<LN3D_EPI_GELU_ERF>(p, s);
      case LN3D_EPI_GELU_TANH: return launch_large<LN3D_EPI_GELU_TANH>(p, s);
      case LN3D_EPI_SILU: return launch_large<LN3D_EPI_SILU>(p, s);
      case LN3D_EPI_GATE_RES: return launch_large<LN3D_EPI_GATE_RES>(p, s);
      case LN3D_EPI_F32_SILU:
        if (!a->out1) return LN3D_ERR_BAD_ARG;
        return launch_large<LN3D_EPI_F32_SILU>(p, s);
      case LN3D_EPI_HEADS:
        if (a->tokens <= 0 || a->heads <= 0 || a->head_dim <= 0 || (a->head_dim % 4) != 0 || a->tok_pad < a->tokens)
          return LN3D_ERR_BAD_ARG;
        return launch_large<LN3D_EPI_HEADS>(p, s);
      default: return LN3D_ERR_UNSUPPORTED;
    }
  }
  switch (a->epilogue) {
    case LN3D_EPI_F32: return launch<LN3D_EPI_F32>(p, s);
    case LN3D_EPI_BF16: return launch<LN3D_EPI_BF16>(p, s);
    case LN3D_EPI_GELU_ERF: return launch<LN3D_EPI_GELU_ERF>(p, s);
    case LN3D_EPI_GELU_TANH: return launch<LN3D_EPI_GELU_TANH>(p, s);
    case LN3D_EPI_SILU: return launch<LN3D_EPI_SILU>(p, s);
    case LN3D_EPI_GATE_RES: return launch<LN3D_EPI_GATE_RES>(p, s);
    case LN3D_EPI_F32_SILU:
      if (!a->out1) return LN3D_ERR_BAD_ARG;
      return launch<LN3D_EPI_F32_SILU>(p, s);
    case LN3D_EPI_HEADS:
      if (a->tokens <= 0 || a->heads <= 0 || a->head_dim <= 0 || (a->head_dim % 4) != 0 || a->tok_pad < a->tokens)
        return LN3D_ERR_BAD_ARG;
      return launch<LN3D_EPI_HEADS>(p, s);
    default: return LN3D_ERR_UNSUPPORTED;
  }
}
