// Fused (flash-style) attention for gfx950: O = softmax(scale * Q K^T) V, bf16 in/out, fp32 softmax.
//
// Design (MI355X-first):
//  * v_mfma_f32_32x32x16_bf16, wave64, "swapped" products so that every softmax statistic is lane-local:
//       S^T[key, q] = K . Q^T      (A = K tile rows, B = Q fragment held in VGPRs for the whole kernel)
//       O^T[d,  q]  = V^T . P^T    (A = V^T tile rows, B = P converted to bf16 in registers)
//    In the 32x32 accumulator layout a lane owns ONE query column (lane&31) and 16 of the 32 key rows, so row-max /
//    row-sum are 32 in-register ops + one exchange with lane^32, the online-softmax rescale factor is a per-lane
//    scalar, and P feeds the second MFMA directly as its B operand (no LDS round trip, no transposes): the key
//    permutation of the accumulator layout is baked into the V^T memory layout (see attn_kernel).
//  * softmax in the log2 domain: one FMA + one v_exp_f32 per score; the O rescale is deferred, wave-uniformly, until a
//    row maximum outgrows the running reference by more than 2^8 (mathematically exact: softmax is shift-invariant).
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "../../include/ln3d.h"

struct AttnP {
  const bf16_t* Q; const bf16_t* K; const bf16_t* Vt; bf16_t* O;
  int B, H, Nq, Nq_pad, Nk, Nk_pad;
  int64_t ldo;
  float scale_log2;
  int causal;
  int nsplit;   // attn_stream_kernel: workgroups per (batch, head)
};

#define KVB 64


// One workgroup = 8 waves = 256 query rows of one (batch, head): K / V^T stream from L2 ONCE per 256 queries
// (the stream, not the MFMA, bounds this kernel: K+V of one head is 196 KB against 50 MFLOP per 256 queries).
// K / V^T blocks of 64 keys are filled by LDS-DMA (global_load_lds_dwordx4) into an NST-deep ring; counted vmcnt +
// raw s_barrier keep NST-2 blocks in flight across the per-block barrier.  LDS rows are unpadded and XOR-swizzled at
// 16-B granularity (swizzle on the DMA source address and on the ds_read_b128 side): conflict-free b128 reads.
// V^T arrives from the QKV GEMM epilogue with the keys of every 16-group permuted to [0-3, 8-11, 4-7, 12-15], which is
// the order the S^T accumulator layout hands P to the second MFMA: the PV A-operand is one aligned 16-B read.
// OCC = waves per SIMD the register allocation is bounded for: 4 (128 VGPRs, two workgroups per CU) for long key
// sequences where latency hiding matters; 2 (256 VGPRs, no spills in the masked tail block) for short ones (cross-attention).
typedef float f32x2 __attribute__((ext_vector_type(2)));

// bench-only ablations (tools/attn_abl.hip builds this file with -DLN3D_ATTN_ABL=n): 1 = no v_exp, 2 = no MFMA, 4 = no barrier/DMA wait
#ifndef LN3D_ATTN_ABL
#define LN3D_ATTN_ABL 0
#endif


// DT = true head size when the heads are stored zero-padded to DH (DiT-XL/2: 72 in 128-wide rows, r4): the S^T products run over
// ceil(DT / 16) k-steps and the PV products over ceil(DT / 32) blocks of output rows instead of DH / 16 and DH / 32 (5 + 3 of
// 8 + 4 for 72: the padding contributes exact zeros), and O is written COMPACT, DT dims per head (row stride ldo = H * DT), so the
// projection GEMM behind it contracts over H * DT instead of H * DH.  K / V^T tiles are still fetched whole (the DMA map and
// its counted waits stay those of the DH-wide layout).
template <int DH, int OCC, int DT = DH>
__global__ __launch_bounds__(512, OCC) void attn_kernel(AttnP p) {
#ifndef LN3D_ATTN_NST128
#define LN3D_ATTN_NST128 3       // bench-only: ring depth of the 128-wide instantiations (4 x 32 KB still fits the CU: one workgroup either way)
#endif
#ifndef LN3D_ATTN_PRIO
#define LN3D_ATTN_PRIO 0         // bench-only: static priority for waves 4-7 (what attn_kres_kernel does) so that the two waves of a SIMD drift apart
#endif
  constexpr int NST = DH == 64 ? 4 : (DH == 80 ? 3 : LN3D_ATTN_NST128);           // ring depth
  // DH = 80 (r6: DiT-XL/2's 72-wide and the U-Net's 80-wide heads, stored 80 wide instead of 128): the K rows (160 B) sit in LDS at a
  // 176-byte pitch - 11 chunks, an odd count, so that the 8 rows a b128 read phase touches fall into 8 different bank groups without a
  // swizzle (XOR needs a power-of-two row) - and the V^T tile keeps room for 96 rows (3 MFMA row blocks; rows 80 - 95 are never loaded and
  // only feed accumulator rows that are never stored).  What the narrower storage buys is L2 / fabric traffic: the K / V^T stream of a launch
  // (every query block re-reads its head) is what bounds this kernel at these shapes - 128-wide tiles of 72-wide heads cost 77 us whether
  // 22 or 32 MFMAs per block run on them, with a 3- or a 4-deep ring (profiles/r6_attn.md).
  constexpr int KROWB = DH == 80 ? 176 : DH * 2;  // K tile row pitch in LDS
  constexpr int KTILE = KVB * KROWB;
  constexpr int VTILE = (DH == 80 ? 96 : DH) * 128;   // V^T tile: DH rows x 64 keys
  constexpr int STAGEB = KTILE + VTILE;
  constexpr int NDS = (DT + 15) / 16, NDT = (DT + 31) / 32;
  static_assert(DT <= DH && DT % 8 == 0, "true head size: a multiple of 8 within the stored row");
  static_assert(DH == 64 || DH == 80 || DH == 128, "stored head width");
  constexpr int NKI = KTILE / 1024, NVI = DH * 128 / 1024;    // wave-level DMA instructions per K / V^T tile (1024 LDS bytes each)
  constexpr int LPW = (NKI + NVI + 7) / 8;        // DMA instruction slots per wave per stage: slot s = wid + 8 i is K instruction s (s < NKI) or V^T instruction s - NKI
  constexpr int KCH = DH / 8;                     // 16-B chunks per K row in memory (8, 10 or 16)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  if constexpr (LN3D_ATTN_PRIO != 0) { if (wid >= 4) __builtin_amdgcn_s_setprio(LN3D_ATTN_PRIO); }
  // block -> (head, query block): the query blocks of a head run on ONE XCD (block b runs on XCD b % 8), next to each other in
  // dispatch order, so that the head's K / V^T stream is fetched into that XCD's L2 once (r4; a (query block, head) grid spread
  // them over three XCDs)
  int bh, qblk;
  {
    const int nqb_ = (p.Nq + 255) / 256, BH_ = p.B * p.H, b_ = blockIdx.x;
    if ((BH_ & 7) == 0) { const int xcd = b_ & 7, slot = b_ >> 3; bh = (slot / nqb_) * 8 + xcd; qblk = slot % nqb_; }
    else { bh = b_ / nqb_; qblk = b_ % nqb_; }
  }
  const int q0 = qblk * 256 + wid * 32;

  const bf16_t* Qg = p.Q + (int64_t)bh * p.Nq_pad * DH;
  const bf16_t* Kg = p.K + (int64_t)bh * p.Nk_pad * DH;
  const bf16_t* Vg = p.Vt + (int64_t)bh * DH * p.Nk_pad;

  bf16x8 qf[NDS];
  {
    int qr = q0 + l31; qr = qr < p.Nq ? qr : p.Nq - 1;
    const bf16_t* qp = Qg + (int64_t)qr * DH + hi * 8;
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) qf[ds] = *reinterpret_cast<const bf16x8*>(qp + ds * 16);
  }

  // DMA source addressing: wave-instruction j of a tile writes LDS bytes [1024 j, 1024 j + 1024).  A wave owns the slots wid, wid + 8, ...
  // of the stage's NKI + NVI instructions (64 / 128 wide: one or two K and as many V^T instructions each, as before; 80 wide: 11 + 10
  // instructions, three for waves 0 - 4 and two for waves 5 - 7 - the counted waits below use the wave's own count).
  const bf16_t* dsrc[LPW];
  int ddst[LPW], dstep[LPW];
  int nslot = 0;                                   // wave-uniform: this wave's DMA instructions per stage
#pragma unroll
  for (int i = 0; i < LPW; ++i) {
    const int sl = wid + 8 * i;
    if (sl < NKI) {
      const int j = sl;
      int krow, kcp;
      if constexpr (DH == 80) { const int n = j * 64 + lane; krow = n / 11; kcp = n - krow * 11; kcp = kcp < 10 ? kcp : 9; }   // pitch chunk 10 = padding: any valid address
      else { krow = j * (1024 / KROWB) + lane / KCH; kcp = lane % KCH; }
      const int kkey = DH == 64 ? ((krow >> 1) & 7) : (DH == 128 ? (krow & 15) : 0);
      dsrc[i] = Kg + (int64_t)krow * DH + ((kcp ^ kkey) * 8);
      ddst[i] = j * 1024;
      dstep[i] = KVB * DH;
    } else {
      const int j = sl - NKI < NVI ? sl - NKI : NVI - 1;
      const int vrow = j * 8 + (lane >> 3), vcp = lane & 7;
      dsrc[i] = Vg + (int64_t)vrow * p.Nk_pad + ((vcp ^ ((vrow >> 1) & 7)) * 8);
      ddst[i] = KTILE + j * 1024;
      dstep[i] = KVB;
    }
    if (sl < NKI + NVI) nslot = i + 1;
  }
#define A_ISSUE(kbv)                                                                                          \
  {                                                                                                           \
    const int kb_i_ = (kbv);                                                                                  \
    char* sb_ = smem + (kb_i_ % NST) * STAGEB;                                                                \
    _Pragma("unroll") for (int ii_ = 0; ii_ < LPW; ++ii_)                                                     \
      if (ii_ < nslot) lds_dma16_v((dsrc[ii_] + (int64_t)kb_i_ * dstep[ii_]), lds_addr((sb_ + ddst[ii_])));  \
  }

  f32x16 oacc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
  float m_run = -3.0e38f, l_run = 0.f;
  const int nkb = (p.Nk + KVB - 1) / KVB;

  // fragment read offsets
  const int kkey_r = DH == 64 ? ((l31 >> 1) & 7) : (DH == 128 ? (l31 & 15) : 0);      // key rows kt*32 + l31: kt*32 does not change the key
  const int k_off = l31 * KROWB;
  const int vkey_r = (l31 >> 1) & 7;
  const int v_off = KTILE + l31 * 128;

  auto process = [&](int kb, auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    const char* sb = smem + (kb % NST) * STAGEB;
    f32x16 st[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sb + k_off + kt * 32 * KROWB + (((2 * ds + hi) ^ kkey_r) << 4));
        if constexpr (LN3D_ATTN_ABL & 2) { st[kt][ds] += (float)kf[0] * (float)qf[ds][0]; } else
        st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ds], st[kt], 0, 0, 0);
      }
    }
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
    // Deferred rebase: the running reference m_run only has to bound the scores loosely (softmax is invariant to it), so
    // O / l are rescaled - wave-uniformly - only when some row's block maximum outgrows it by more than 2^8; until then
    // P = exp2(s - m_run) <= 256, well inside bf16 / fp32 range.  After the first blocks this branch is almost never taken.
    const float m_blk = mx * p.scale_log2;
    if (!__all(m_blk <= m_run + 8.0f)) {
      const float m_new = fmaxf(m_run, m_blk);
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
        const float a = fmaf(st[kt][r], p.scale_log2, -m_run);
        const float pv = (LN3D_ATTN_ABL & 1) ? a : __builtin_amdgcn_exp2f(a);
        st[kt][r] = pv;
        psum += pv;
      }
    l_run += psum;
    bf16x8 pb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      union { uint32_t u[4]; bf16x8 v; } cv;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        cv.u[jj] = pack2bf(st[s >> 1][8 * (s & 1) + 2 * jj], st[s >> 1][8 * (s & 1) + 2 * jj + 1]);
      pb[s] = cv.v;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sb + v_off + dt * 32 * 128 + (((2 * s + hi) ^ vkey_r) << 4));
        if constexpr (LN3D_ATTN_ABL & 2) { oacc[dt][s] += (float)vf[0] * (float)pb[s][0]; } else
        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[s], oacc[dt], 0, 0, 0);
      }
    }
  };

  // prologue: up to NST-1 blocks in flight
#pragma unroll
  for (int i = 0; i < NST - 1; ++i)
    if (i < nkb) A_ISSUE(i);
#define A_WAIT(kbv)                                                                                         \
  {                                                                                                           \
    const int ahead_ = min(NST - 2, nkb - 1 - (kbv));   /* blocks that may stay in flight while block kb is consumed */ \
    const int infl_ = ahead_ * nslot;                   /* this wave's DMA instructions behind block kb's: 0, 2, 3, 4 or 6 */ \
    if (infl_ >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                        \
    else if (infl_ == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                   \
    else if (infl_ == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                                   \
    else if (infl_ == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                   \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                     \
    __builtin_amdgcn_s_waitcnt(0xC07F);                 /* this wave's LDS reads of block kb-1 are complete */  \
    __builtin_amdgcn_s_barrier();                                                                             \
    if ((kbv) + NST - 1 < nkb) A_ISSUE((kbv) + NST - 1); /* ring slot of block kb-1: every wave is past it */   \
  }
  for (int kb = 0; kb + 1 < nkb; ++kb) {
    if constexpr (!(LN3D_ATTN_ABL & 4)) { A_WAIT(kb); }
    process(kb, std::false_type{});
  }
  A_WAIT(nkb - 1);
  if ((p.Nk & (KVB - 1)) != 0) process(nkb - 1, std::true_type{});
  else process(nkb - 1, std::false_type{});

  float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  // O goes out through LDS: a lane's accumulator quad is 4 head dims of ONE query and lanes 0-31 are 32 queries, so direct
  // stores write 16-byte pieces scattered over 32 rows per instruction (partial-line writes, ~1.5 TB/s).  Each wave
  // transposes 32 queries x 64 dims through its own 8 KB of the retired ring (fp32, 16-byte chunk c of row r at c ^ (r & 15):
  // conflict-free both ways) and stores complete 128-byte rows of its head.
  __builtin_amdgcn_s_barrier();                       // every wave is done with the K/V ring
  {
    char* stg = smem + wid * 8192;
    const int b = bh / p.H, h = bh - b * p.H;
    const int rrow = lane >> 4, rc = lane & 15;
#pragma unroll
    for (int dp = 0; dp < (DT + 63) / 64; ++dp) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = ii * 8 + 2 * g + hi;
          const int dt = 2 * dp + ii;
          if (dt < NDT) {                            // compile-time after unrolling
            const int dc = dt < NDT ? dt : 0;
            *reinterpret_cast<float4*>(stg + l31 * 256 + ((c ^ (l31 & 15)) << 4)) =
                make_float4(oacc[dc][4 * g + 0] * inv, oacc[dc][4 * g + 1] * inv, oacc[dc][4 * g + 2] * inv, oacc[dc][4 * g + 3] * inv);
          }
        }
#pragma unroll
      for (int it = 0; it < 4; ++it) {               // 8 dims per lane: 16-byte stores, 8 lanes per 128-byte row
        const int row = 8 * it + (lane >> 3), c8 = lane & 7;
        const float4 v0 = *reinterpret_cast<const float4*>(stg + row * 256 + (((2 * c8) ^ (row & 15)) << 4));
        const float4 v1 = *reinterpret_cast<const float4*>(stg + row * 256 + (((2 * c8 + 1) ^ (row & 15)) << 4));
        const int q = q0 + row;
        if (q < p.Nq && dp * 64 + 8 * c8 < DT) {     // compact heads: only the true dims leave
          uint4 o;
          o.x = pack2bf(v0.x, v0.y); o.y = pack2bf(v0.z, v0.w); o.z = pack2bf(v1.x, v1.y); o.w = pack2bf(v1.z, v1.w);
          *reinterpret_cast<uint4*>(p.O + ((int64_t)b * p.Nq + q) * p.ldo + h * DT + dp * 64 + 8 * c8) = o;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Streaming kernel for the DiT self-attention shapes: Dh = 64, Nk a multiple of 256 (768 / 1024 latent tokens, the 256-token
// planes of the VAE decoder).
//
// What bounds attention on this part is INSTRUCTION ISSUE, not the matrix pipe: a SIMD issues about one instruction per 4-5
// cycles whatever its wave count (profiles/r2_attn_pmc.md: attn_kernel above and the first versions of this kernel all ran
// at 5.5-6.5 cycles per issued instruction per SIMD; removing every MFMA changed nothing, removing any other instruction
// class helped in proportion to its count).  A v_mfma_f32_32x32x16_bf16 occupies the pipe for 32 cycles, so the matrix pipe
// is only kept busy if there are <= ~7 other instructions per MFMA.  attn_kernel issues ~15 (200 VALU per 16 MFMAs plus
// address / counter / branch overhead).  This kernel is written against that budget:
//  * softmax VALU per score: ONE v_exp_f32, half a v_pk_add_f32, half a v_cvt_pk_bf16_f32, half a v_max3_f32.  The scale
//    is folded into the query fragment once per query block (q * scale * log2 e) and the running reference -m is the C
//    operand of the first S^T MFMA of every tile (a persistent 16-register broadcast), so the accumulators ARE
//    s*scale*log2e - m: no multiply-add per score.  The row-maximum test of the deferred rebase needs no cross-lane
//    exchange (the wave-wide vote covers both half-rows).
//  * no address arithmetic in the loop: the step loop is unrolled over the 4 ring slots, every ds_read_b128 is one of four
//    lane-constant offset registers plus an immediate, the DMA source is a scalar base plus a lane-constant offset.
//  * fragments are read one half-step ahead into a second register set; every wait in steady state is a counted one.
//  * ONE workgroup (8 waves x 32 queries, 2 waves per SIMD, <= 256 VGPRs) walks ALL query blocks of its (batch, head) - or a
//    contiguous share of them when there are fewer heads than CUs - while the K / V^T ring keeps streaming across
//    query-block boundaries: one prologue per workgroup, the K / V^T re-reads of a head stay in the L2 of the XCD that owns
//    it, and the step loop is software-pipelined across tiles (the S^T MFMAs of the next 32-key tile are issued under the
//    exp2 / sum / pack work of the current one, whose PV MFMAs run under the row-maximum test of the next).
//  * O leaves through a wave-private staging region beside the ring (the ring never retires here); the next query block's
//    queries arrive in the same region by LDS-DMA one step ahead.
// max of three (the compiler selects v_max3_f32).  NOT inline asm: hipcc pads no hazards inside an asm statement, and a
// v_max3 placed by hand right behind the S^T MFMAs read their accumulators before the matrix pipe had written them back -
// results stayed within tolerance but differed from run to run (tools/attn_bench.hip now checks repeated launches bitwise).
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

__global__ __launch_bounds__(512, 2) void attn_stream_kernel(AttnP p) {
  constexpr int DH = 64, NDS = 4, NDT = 2;
  constexpr int NST = 8;                                  // ring slots: 6 stages (96 KB) in flight per CU cover the L2 / MALL latency
  constexpr int KTILE = 8192, STAGEB = 16384, RING = NST * STAGEB, QB = 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;

  // work item: (head, share of its query blocks).  Block b runs on XCD b % 8: the shares of a head stay on one XCD.
  const int nqb = (p.Nq + QB - 1) / QB;
  const int BH = p.B * p.H;
  int bh, sp;
  {
    const int b = blockIdx.x;
    if ((BH & 7) == 0) { const int xcd = b & 7, slot = b >> 3; bh = (slot / p.nsplit) * 8 + xcd; sp = slot % p.nsplit; }
    else { bh = b / p.nsplit; sp = b % p.nsplit; }
  }
  const int qper = (nqb + p.nsplit - 1) / p.nsplit;
  const int qb0 = sp * qper, qb1 = min(nqb, qb0 + qper);
  if (qb0 >= qb1) return;
  const int nkb = p.Nk >> 6;                              // key blocks of 64: a multiple of 4 (launcher)

  const bf16_t* Qg = p.Q + (int64_t)bh * p.Nq_pad * DH;
  const char* Kg = reinterpret_cast<const char*>(p.K + (int64_t)bh * p.Nk_pad * DH);
  const char* Vg = reinterpret_cast<const char*>(p.Vt + (int64_t)bh * DH * p.Nk_pad);
  const int b_smp = bh / p.H, h_idx = bh - b_smp * p.H;

  // Tile image shared by K rows, V^T rows and the wave's own Q rows: 128-byte rows, 16-byte chunk c of row r at
  // c ^ ((r >> 1) & 7).  fo[j] = LDS ADDRESS of chunk 2j + hi of row l31 - the lane's fragment of contraction step j - in the
  // ring half the current steps consume, fo_hi[j] the same in the other half (DS instructions carry 16-bit immediates: the
  // two sets swap every 4 steps instead of adding a slot base per read).
  typedef __attribute__((address_space(3))) const bf16x8 lds_frag_t;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;
  uint32_t fo[4], fo_hi[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { fo[j] = lds0 + l31 * 128 + (((2 * j + hi) ^ ((l31 >> 1) & 7)) << 4); fo_hi[j] = fo[j] + 65536; }
  // DMA: wave w fills rows 8w..8w+7 of the K tile and of the V^T tile of a stage (1 KB each)
  const int drow = 8 * wid + (lane >> 3), dchunk = ((lane & 7) ^ ((drow >> 1) & 7)) << 4;
  const uint32_t koffs = drow * 128 + dchunk;
  const uint32_t voffs = (uint32_t)drow * (uint32_t)p.Nk_pad * 2u + dchunk;
  const char* kq = Kg; const char* vq = Vg;               // scalar: source of the next stage to issue
  int kb_issue = 0;
  int half_base = 0;                                      // LDS byte offset of the ring half (4 slots) the current steps consume
  auto issue_stage = [&](auto slot_tag) __attribute__((always_inline)) {   // slot relative to the current half: 0..7
    constexpr int SL = decltype(slot_tag)::value & (NST - 1);
    char* dst = smem + ((SL >= 4 ? half_base ^ 65536 : half_base) + (SL & 3) * STAGEB + wid * 1024);
    lds_dma16_s(kq, koffs, lds_addr(dst));
    lds_dma16_s(vq, voffs, lds_addr(dst + KTILE));
    ++kb_issue; kq += 64 * DH * 2; vq += 64 * 2;          // the stream wraps around the head's keys at every query block
    if (kb_issue == nkb) { kb_issue = 0; kq = Kg; vq = Vg; }
  };
  // Queries of a query block: the wave's 32 rows x 128 B are contiguous in HBM and go through the wave's own 4 KB staging
  // region by LDS-DMA, issued at the first step of the previous query block: no VGPRs are held for them.
  bf16x8 qf[NDS];
  char* const wstage = smem + RING + wid * 4096;
  auto dma_q = [&](int qb) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = j * 8 + (lane >> 3);
      int qr = qb * QB + wid * 32 + row; qr = qr < p.Nq_pad ? qr : p.Nq_pad - 1;
      lds_dma16_v((Qg + (int64_t)qr * DH + (((lane & 7) ^ ((row >> 1) & 7)) * 8)), lds_addr((wstage + j * 1024)));
    }
  };
  // (the caller has waited for the DMA)  fragments, scaled by scale * log2(e) and rounded to bf16 once more
  auto read_q = [&]() __attribute__((always_inline)) {
    const uint32_t qrow = lds0 + RING + wid * 4096 + l31 * 128;
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) {
      union { uint32_t u[4]; bf16x8 v; } cv;
      cv.v = *(lds_frag_t*)(uintptr_t)(qrow + (((2 * ds + hi) ^ ((l31 >> 1) & 7)) << 4));
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        cv.u[jj] = pack2bf(bf2f((bf16_t)(cv.u[jj] & 0xffffu)) * p.scale_log2, bf2f((bf16_t)(cv.u[jj] >> 16)) * p.scale_log2);
      qf[ds] = cv.v;
    }
  };

  f32x16 oacc[NDT], negm, stA, stB;
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) negm[r] = 0.f;
  float l_run = 0.f;

  // Fragment registers.  All LDS reads of a half-step are issued at its top: the V^T columns of the current tile (multiplied
  // in its last third) and the K rows of the tile AFTER next into the other of two K sets (used by the S^T MFMAs that open the
  // next half-step), so no MFMA ever waits on a read that was just issued.  Addresses are fo[j] / fo_hi[j] + an immediate.
  bf16x8 kfA[NDS], kfB[NDS], vf[2 * NDT];
  auto load_kf = [&](bf16x8 (&kf)[NDS], auto slot_tag, auto kt_tag) __attribute__((always_inline)) {   // K rows of tile kt of a slot
    constexpr int SL = decltype(slot_tag)::value;           // 0..3 in the current half, 4 = slot 0 of the other half
    constexpr int OFF = (SL & 3) * STAGEB + decltype(kt_tag)::value * 4096;
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) kf[ds] = *(lds_frag_t*)(uintptr_t)((SL >= 4 ? fo_hi[ds] : fo[ds]) + OFF);
  };
  auto load_vf = [&](bf16x8 (&vf)[2 * NDT], auto slot_tag, auto kt_tag) __attribute__((always_inline)) {   // V^T of the tile's 32 keys
    constexpr int SL = decltype(slot_tag)::value;
    constexpr int OFF = (SL & 3) * STAGEB + KTILE;
    constexpr int KT = decltype(kt_tag)::value;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
        vf[s * NDT + dt] = *(lds_frag_t*)(uintptr_t)((SL >= 4 ? fo_hi[2 * KT + s] : fo[2 * KT + s]) + OFF + dt * 4096);
  };
  // S^T of a 32-key tile from its K fragments, C operand = cinit (0 for the first tile of a query block)
  auto qk_tile = [&](f32x16& st, const bf16x8 (&kf)[NDS], const f32x16& cinit) __attribute__((always_inline)) {
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) {
      if constexpr (LN3D_ATTN_ABL & 2) { if (ds == 0) st = cinit; asm volatile("" :: "v"(kf[ds])); } else
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ds], qf[ds], ds == 0 ? cinit : st, 0, 0, 0);
    }
  };
  auto tile_max = [&](const f32x16& st) __attribute__((always_inline)) {
    float mx = max3f(st[0], st[1], st[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = max3f(mx, st[r], st[r + 1]);
    return max3f(mx, st[15], st[15]);
  };
  // first tile of a query block: the reference becomes its row maximum (the accumulators hold s*scale*log2e, C was 0)
  auto fresh_reference = [&](f32x16& st) __attribute__((always_inline)) {
    float mx = tile_max(st);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = -mx;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] -= mx;
  };
  // O: bf16 through the wave's 4 KB (8-byte chunk c of row r at c ^ (r & 15)), out as 16 bytes per lane = whole 128-byte rows
  auto store_o = [&](int qb) __attribute__((always_inline)) {
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int q0 = qb * QB + wid * 32;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        uint2 o;
        o.x = pack2bf(oacc[dt][4 * gq + 0] * inv, oacc[dt][4 * gq + 1] * inv);
        o.y = pack2bf(oacc[dt][4 * gq + 2] * inv, oacc[dt][4 * gq + 3] * inv);
        const int c8 = dt * 8 + 2 * gq + hi;
        *reinterpret_cast<uint2*>(wstage + l31 * 128 + ((c8 ^ (l31 & 15)) << 3)) = o;
      }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 8 * i + (lane >> 3), c16 = lane & 7;
      // chunks 2*c16 and 2*c16+1 of row r sit in one aligned 16-byte slot, swapped when r is odd
      uint4 v = *reinterpret_cast<const uint4*>(wstage + r * 128 + ((c16 ^ ((r & 15) >> 1)) << 4));
      if (r & 1) { const uint32_t t0 = v.x, t1 = v.y; v.x = v.z; v.y = v.w; v.z = t0; v.w = t1; }
      const int q = q0 + r;
      if (q < p.Nq) *reinterpret_cast<uint4*>(p.O + ((int64_t)b_smp * p.Nq + q) * p.ldo + h_idx * DH + c16 * 8) = v;
    }
  };

  // One pipelined half-step: P and PV of the 32-key tile `cur`, while the S^T MFMAs of the NEXT tile (whose K rows are in kf)
  // run into `nxt`; `loads` reads the V^T columns of the current tile and the K rows of the tile after next.
  // MODE 0: next tile with C = -m ; 2: next tile opens the NEXT query block (C = 0, qf already holds its queries) ;
  //      3: nothing follows.
  // The whole half-step is ONE scheduling region; the re-base vote on the next tile's row maximum closes it.  (Voting on the
  // SUM of P instead - no row maximum on the common path, 9 instructions fewer - measured 10 % slower: the vote then sits in the
  // middle of the half-step and the PV MFMAs can no longer be issued under the exps.)
  auto half_step = [&](f32x16& cur, f32x16& nxt, auto mode_tag, const bf16x8 (&kf)[NDS], auto&& loads) __attribute__((always_inline)) {
    constexpr int MODE = decltype(mode_tag)::value;
    if constexpr (!(LN3D_ATTN_ABL & 32)) loads();
    if constexpr (MODE == 0) qk_tile(nxt, kf, negm);
    if constexpr (MODE == 2) { f32x16 z; _Pragma("unroll") for (int r = 0; r < 16; ++r) z[r] = 0.f; qk_tile(nxt, kf, z); }
    float ps0 = 0.f, ps1 = 0.f, ps2 = 0.f, ps3 = 0.f;
    bf16x8 pb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      union { uint32_t u[4]; bf16x8 v; } cv;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float e0 = (LN3D_ATTN_ABL & 1) ? cur[8 * s + 2 * jj] : __builtin_amdgcn_exp2f(cur[8 * s + 2 * jj]);
        const float e1 = (LN3D_ATTN_ABL & 1) ? cur[8 * s + 2 * jj + 1] : __builtin_amdgcn_exp2f(cur[8 * s + 2 * jj + 1]);
        if (jj & 1) { ps2 += e0; ps3 += e1; } else { ps0 += e0; ps1 += e1; }
        cv.u[jj] = pack2bf(e0, e1);
      }
      pb[s] = cv.v;
    }
    l_run += (ps0 + ps1) + (ps2 + ps3);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        if constexpr (LN3D_ATTN_ABL & 2) { asm volatile("" :: "v"(vf[s * NDT + dt]), "v"(pb[s])); } else
        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s * NDT + dt], pb[s], oacc[dt], 0, 0, 0);
      }
    if constexpr (MODE == 0 && !(LN3D_ATTN_ABL & 8)) {
      // deferred re-base (wave-uniform, rare after the first tiles): scores of the next tile outgrow the reference by > 2^8
      const float mx = tile_max(nxt);
      if (!__all(mx <= 8.0f)) {
        const float mxx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float delta = fmaxf(mxx, 0.f);
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] -= delta;
#pragma unroll
        for (int r = 0; r < 16; ++r) nxt[r] -= delta;
      }
    }
  };

  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  // ---- prologue: queries of the first block, 7 ring stages; stage 0 and the queries are the oldest 6 of the 18 requests
  dma_q(qb0);
#pragma unroll
  for (int i = 0; i < NST - 1; ++i) {
    // (constant slot per unrolled iteration)
    switch (i) {
      case 0: issue_stage(std::integral_constant<int, 0>{}); break; case 1: issue_stage(std::integral_constant<int, 1>{}); break;
      case 2: issue_stage(std::integral_constant<int, 2>{}); break; case 3: issue_stage(std::integral_constant<int, 3>{}); break;
      case 4: issue_stage(std::integral_constant<int, 4>{}); break; case 5: issue_stage(std::integral_constant<int, 5>{}); break;
      default: issue_stage(std::integral_constant<int, 6>{}); break;
    }
  }
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_q();
  {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    load_kf(kfB, I0{}, I0{});
    qk_tile(stA, kfB, z);
    fresh_reference(stA);
  }
  load_kf(kfA, I0{}, I1{});

  // ---- the stream.  Step g consumes the stage in slot g & 7: V^T of its key block, K of the next block from slot g+1.  At
  // its top the stage in slot g+1 must have landed for everybody; the five stages behind it may stay in flight (10 requests,
  // plus the 4 of a query DMA issued within the last 5 steps).  Behind the barrier every wave has left step g-1, whose slot is
  // refilled with stage g+7.  A step = two half-steps with STATIC accumulator names: (stA -> stB) on tile 0, (stB -> stA) on
  // tile 1.  The loop is unrolled over the 4 slots of a ring half (slot offsets are immediates); nkb is a multiple of 4, so a
  // query block ends at the end of a half and the two address sets swap there.
  if constexpr (LN3D_ATTN_ABL & 64) { store_o(qb0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }   // bench: prologue + one store only
  // The wait is always "at most 10 requests outstanding": they are younger than stage g+1 whether or not the 4 requests of a
  // query DMA (issued at the first step of a query block) are among them, and the queries are older than anything a boundary
  // 6 or more steps later allows to be outstanding; only a 4-step query block (256 keys) has to ask for them by count.
  auto top_of_step = [&](auto slot_tag, bool first, bool boundary, bool more, int qb) __attribute__((always_inline)) {
    constexpr int SL = decltype(slot_tag)::value;
    if constexpr (!(LN3D_ATTN_ABL & 4)) {
      if (boundary && nkb < 8) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (first && more) dma_q(qb + 1);
    if constexpr (!(LN3D_ATTN_ABL & 16)) issue_stage(std::integral_constant<int, SL + 7>{});
  };
  auto swap_halves = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { const uint32_t t = fo[j]; fo[j] = fo_hi[j]; fo_hi[j] = t; }
    half_base ^= 65536;
  };
  // half-step A (tile 0): S^T of tile 1 from kfA, reads V^T(tile 0) and K(next block, tile 0) -> kfB; half-step B the mirror
#define HS_A(SL_, MODE_) half_step(stA, stB, MODE_{}, kfA, [&]() __attribute__((always_inline)) {                \
    load_vf(vf, std::integral_constant<int, SL_>{}, I0{}); load_kf(kfB, std::integral_constant<int, SL_ + 1>{}, I0{}); })
#define HS_B(SL_, MODE_) half_step(stB, stA, MODE_{}, kfB, [&]() __attribute__((always_inline)) {                \
    load_vf(vf, std::integral_constant<int, SL_>{}, I1{}); if constexpr (MODE_::value != 3) load_kf(kfA, std::integral_constant<int, SL_ + 1>{}, I1{}); })
#define STEP(SL_, FIRST_) { top_of_step(std::integral_constant<int, SL_>{}, FIRST_, false, more, qb); HS_A(SL_, I0); HS_B(SL_, I0); }
  for (int qb = qb0; qb < qb1; ++qb) {
    const bool more = qb + 1 < qb1;
    bool first = true;
    for (int grp = 4; grp < nkb; grp += 4) {
      STEP(0, first); STEP(1, false); STEP(2, false); STEP(3, false);
      swap_halves();
      first = false;
    }
    STEP(0, first); STEP(1, false); STEP(2, false);
    top_of_step(I3{}, false, more, more, qb);
    HS_A(3, I0);
    if (more) {
      read_q();
      HS_B(3, I2);
      store_o(qb);
      l_run = 0.f;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
      fresh_reference(stA);
      swap_halves();
    } else {
      HS_B(3, I3);
      store_o(qb);
    }
  }
#undef STEP
#undef HS_A
#undef HS_B
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the ring's read-ahead must not outlive the workgroup's LDS
}

// ---------------------------------------------------------------------------------------------
// K-resident kernel (r3) for the DiT-L/2 self-attention shape: Dh = 64, 512 <= Nk <= 768 (a multiple of 256), Nq a multiple of
// 256, at least one (batch, head) per CU.  Three changes against attn_stream_kernel, each aimed at a measured limiter
// (profiles/r2_attn_pmc.md):
//  * K of the head (<= 96 KB) stays in LDS for all query blocks of the workgroup; only V^T streams (4 slots x 64 keys).  The
//    fabric traffic of a launch drops from K + V^T re-read per query block (150 MB at 16 x 16 heads x 768, which alone takes
//    ~26 us) to K once + V^T whose re-reads now fit the XCD's L2 (32 heads x 96 KB = 3 MB).  K blocks arrive during the first
//    query block, two steps ahead of the V^T block they are used with.
//  * The row sums of P come out of the matrix pipe: one extra MFMA per 16 keys with an all-ones A operand accumulates
//    sum_k P[k, q] (of the bf16-rounded P that also multiplies V) for every query column.  16 v_add per 32-key tile leave the
//    VALU stream, whose issue rate - not the matrix pipe - bounded the r2 kernel.
//  * No running maximum.  The reference of a query block is the row maximum of its FIRST 32-key tile; softmax is invariant to
//    the reference and floating point is scale-free, so the only thing a poor reference can do is overflow, when some later
//    score exceeds it by > 2^127.  That is detected after the fact (the row sum is not finite / > 1e30), flagged per query
//    block, and such blocks are recomputed at the end of the workgroup by a plain exact online-softmax loop (exact_block).
//    The fast path carries no v_max3 chain, no vote and no branch per tile.
// Per 32-key tile and wave: 10 MFMAs (4 S^T, 4 PV, 2 row-sum) against 16 v_exp + 8 v_cvt_pk + 8 ds_read_b128.
// Counted waits: vmcnt counts LDS-DMA loads and global stores in issue order (gfx9 has one counter for both); every wait below
// is "at most W younger requests outstanding", W derived in top_of_step from what was issued behind the request it needs.
#ifndef LN3D_KRES_ORDER   // 0: S^T chain spread between the PV MFMAs, 1: S^T chain back to back, 2: PV of keys 0-15 first (r3: 54.8 / 50.3 / 51.3 us)
#define LN3D_KRES_ORDER 1
#endif
#ifndef LN3D_KRES_PIN     // 1: pin the order of the pieces with sched_barrier; 0: hipcc's own interleave of the same pieces (r3: 50.3 vs 49.0 us)
#define LN3D_KRES_PIN 0
#endif
#ifndef LN3D_KRES_ABL   // tools/attn_bench.hip builds with -DLN3D_KRES_ABL=bits (wrong results by construction): 1 no v_exp, 2 no MFMA,
#define LN3D_KRES_ABL 0  // 4 no barrier / DMA wait, 8 no fragment reads, 16 no DMA issue in the stream, 32 hipcc's own instruction order
#endif
__device__ __forceinline__ f32x16 abl_nomfma(const bf16x8& a, const bf16x8& b, const f32x16& c) { asm volatile("" :: "v"(a), "v"(b)); return c; }
template <bool PRIO, bool ST_INORDER>
__global__ __launch_bounds__(512, 2) void attn_kres_kernel(AttnP p) {
  constexpr int DH = 64, NDS = 4, NDT = 2, QB = 256;
  constexpr int KREG = 768 * 128;                         // resident K rows (128 B each)
  constexpr int VSLOT = 8192, VRING = KREG;               // V^T ring: 4 slots x (64 dims x 64 keys)
  constexpr int WST = KREG + 4 * VSLOT;                   // 8 x 4 KB: per-wave staging (queries in by LDS-DMA, O out)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  if constexpr (PRIO) { if (wid >= 4) __builtin_amdgcn_s_setprio(1); }   // the later-dispatched half loses every arbitration otherwise

  const int nqb = (p.Nq + QB - 1) / QB;
  const int BH = p.B * p.H;
  int bh, sp;
  {
    const int b = blockIdx.x;
    if ((BH & 7) == 0) { const int xcd = b & 7, slot = b >> 3; bh = (slot / p.nsplit) * 8 + xcd; sp = slot % p.nsplit; }
    else { bh = b / p.nsplit; sp = b % p.nsplit; }
  }
  const int qper = (nqb + p.nsplit - 1) / p.nsplit;
  const int qb0 = sp * qper, qb1 = min(nqb, qb0 + qper);
  if (qb0 >= qb1) return;
  const int nkb = p.Nk >> 6;                              // 8 or 12 key blocks of 64 (launcher)
  const int ngrp = nkb >> 2;

  const bf16_t* Qg = p.Q + (int64_t)bh * p.Nq_pad * DH;
  const char* Kg = reinterpret_cast<const char*>(p.K + (int64_t)bh * p.Nk_pad * DH);
  const char* Vg = reinterpret_cast<const char*>(p.Vt + (int64_t)bh * DH * p.Nk_pad);
  const int b_smp = bh / p.H, h_idx = bh - b_smp * p.H;

  // fragment addresses: fo[j] = chunk 2j + hi of row l31 of the tile image at LDS offset 0 (K block 0); fk[j] = the same in the
  // K group (4 blocks = 32 KB) the current steps read; fv[j] in the V^T ring.  DS immediates (16 bit) cover a group / the ring.
  typedef __attribute__((address_space(3))) const bf16x8 lds_frag_t;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;
  uint32_t fo[4], fk[4], fv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { fo[j] = lds0 + l31 * 128 + (((2 * j + hi) ^ ((l31 >> 1) & 7)) << 4); fk[j] = fo[j]; fv[j] = fo[j] + VRING; }
  // DMA pieces: wave w fills rows 8w..8w+7 (1 KB) of a K block / of a V^T stage
  const int drow = 8 * wid + (lane >> 3), dchunk = ((lane & 7) ^ ((drow >> 1) & 7)) << 4;
  const uint32_t koffs = drow * 128 + dchunk;
  const uint32_t voffs = (uint32_t)drow * (uint32_t)p.Nk_pad * 2u + dchunk;
  const char* kq = Kg; const char* vq = Vg;               // scalar: source of the next K block / V^T stage to issue
  int kb_k = 0, kb_v = 0;
  auto issue_k = [&]() __attribute__((always_inline)) {
    lds_dma16_s(kq, koffs, lds0 + kb_k * 8192 + wid * 1024);
    ++kb_k; kq += 64 * DH * 2;
  };
  auto issue_v = [&](auto slot_tag) __attribute__((always_inline)) {
    constexpr int SL = decltype(slot_tag)::value & 3;
    lds_dma16_s(vq, voffs, lds0 + VRING + SL * VSLOT + wid * 1024);
    ++kb_v; vq += 64 * 2;                                 // the stream wraps around the head's keys at every query block
    if (kb_v == nkb) { kb_v = 0; vq = Vg; }
  };
  bf16x8 qf[NDS];
  char* const wstage = smem + WST + wid * 4096;
  auto dma_q = [&](int qb) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = j * 8 + (lane >> 3);
      int qr = qb * QB + wid * 32 + row; qr = qr < p.Nq_pad ? qr : p.Nq_pad - 1;
      lds_dma16_v((Qg + (int64_t)qr * DH + (((lane & 7) ^ ((row >> 1) & 7)) * 8)), lds_addr((wstage + j * 1024)));
    }
  };
  auto scale_q = [&](uint32_t (&u)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
      u[jj] = pack2bf(bf2f((bf16_t)(u[jj] & 0xffffu)) * p.scale_log2, bf2f((bf16_t)(u[jj] >> 16)) * p.scale_log2);
  };
  auto read_q = [&]() __attribute__((always_inline)) {
    const uint32_t qrow = lds0 + WST + wid * 4096 + l31 * 128;
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) {
      union { uint32_t u[4]; bf16x8 v; } cv;
      cv.v = *(lds_frag_t*)(uintptr_t)(qrow + (((2 * ds + hi) ^ ((l31 >> 1) & 7)) << 4));
      scale_q(cv.u);
      qf[ds] = cv.v;
    }
  };

  f32x16 oacc[NDT], lacc, negm, stA, stB;
#pragma unroll
  for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.f; oacc[1][r] = 0.f; lacc[r] = 0.f; negm[r] = 0.f; }
  bf16x8 ones;
  { union { uint32_t u[4]; bf16x8 v; } cv; cv.u[0] = cv.u[1] = cv.u[2] = cv.u[3] = 0x3F803F80u; ones = cv.v; }
  uint32_t ovf = 0;                                       // bit i: query block qb0 + i overflowed its reference (wave-uniform)

  bf16x8 kf[NDS], vf[2 * NDT];
  auto load_kf = [&](auto off_tag) __attribute__((always_inline)) {                       // K rows at fk + OFF
    constexpr int OFF = decltype(off_tag)::value;
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) kf[ds] = *(lds_frag_t*)(uintptr_t)(fk[ds] + OFF);
  };
  auto load_vf = [&](auto slot_tag, auto kt_tag) __attribute__((always_inline)) {         // V^T of 32 keys of a ring slot
    constexpr int OFF = decltype(slot_tag)::value * VSLOT;
    constexpr int KT = decltype(kt_tag)::value;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
        vf[s * NDT + dt] = *(lds_frag_t*)(uintptr_t)(fv[2 * KT + s] + OFF + dt * 4096);
  };
  auto qk_tile = [&](f32x16& st, const f32x16& cinit) __attribute__((always_inline)) {
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds)
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ds], qf[ds], ds == 0 ? cinit : st, 0, 0, 0);
  };
  auto tile_max = [&](const f32x16& st) __attribute__((always_inline)) {
    float mx = max3f(st[0], st[1], st[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = max3f(mx, st[r], st[r + 1]);
    return fmaxf(mx, st[15]);
  };
  // first tile of a query block: the reference of the whole block is its row maximum (the accumulators hold s*scale*log2e)
  auto fresh_reference = [&](f32x16& st) __attribute__((always_inline)) {
    float mx = tile_max(st);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = -mx;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] -= mx;
  };
  // O: bf16 through the wave's 4 KB (8-byte chunk c of row r at c ^ (r & 15)), out as 16 bytes per lane = whole 128-byte rows
  auto store_o = [&](int qb, float inv) __attribute__((always_inline)) {
    const int q0 = qb * QB + wid * 32;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        uint2 o;
        o.x = pack2bf(oacc[dt][4 * gq + 0] * inv, oacc[dt][4 * gq + 1] * inv);
        o.y = pack2bf(oacc[dt][4 * gq + 2] * inv, oacc[dt][4 * gq + 3] * inv);
        const int c8 = dt * 8 + 2 * gq + hi;
        // The LDS write is asm and the reads below go through INTEGER addresses (like the fragment reads): in front of an LDS
        // store, and of an LDS load whose pointer derives from smem, hipcc puts s_waitcnt vmcnt(0) while LDS-DMA requests are
        // pending (it cannot tell the staging region from the ring) - that would drain the V^T read-ahead at every query
        // block.  The data comes from VALU conversions (no MFMA -> LDS hazard to pad); DS operations of a wave execute in order.
        typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
        const u32x2_t ov = {o.x, o.y};
        const uint32_t oaddr = lds0 + WST + wid * 4096 + l31 * 128 + ((c8 ^ (l31 & 15)) << 3);
        asm volatile("ds_write_b64 %0, %1" :: "v"(oaddr), "v"(ov) : "memory");
      }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 8 * i + (lane >> 3), c16 = lane & 7;
      typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
      typedef __attribute__((address_space(3))) const u32x4_t lds_u4_t;
      uint32_t wbase = lds0 + WST + wid * 4096;
      asm volatile("" : "+s"(wbase));                     // opaque: the address must not be folded back into a pointer into smem
      const u32x4_t vr = *(lds_u4_t*)(uintptr_t)(wbase + r * 128 + ((c16 ^ ((r & 15) >> 1)) << 4));
      uint4 v = make_uint4(vr.x, vr.y, vr.z, vr.w);
      if (r & 1) { const uint32_t t0 = v.x, t1 = v.y; v.x = v.z; v.y = v.w; v.z = t0; v.w = t1; }
      const int q = q0 + r;
      *reinterpret_cast<uint4*>(p.O + ((int64_t)b_smp * p.Nq + q) * p.ldo + h_idx * DH + c16 * 8) = v;   // Nq % 256 == 0 (launcher): always 4 stores
    }
  };
  // end of a query block: every lane's lacc[*] is sum_k P[k, q] of its query column
  auto finish_block = [&](int qb) __attribute__((always_inline)) {
    const float l = lacc[0];
    if (__builtin_amdgcn_ballot_w64(!(l < 1.0e30f)) != 0) ovf |= 1u << (qb - qb0);
    store_o(qb, 1.0f / l);
#pragma unroll
    for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.f; oacc[1][r] = 0.f; lacc[r] = 0.f; }
  };

  // One pipelined half-step: P, PV and row sums of the 32-key tile `cur`, while the S^T MFMAs of the NEXT tile run into `nxt`.
  // `loads` reads the V^T columns of `cur` and the K rows of the next tile at the top; the exps of `cur` (available since the
  // previous half-step) cover the LDS latency, so ONE K fragment set suffices (a second one, read a half-step ahead as in
  // attn_stream_kernel, does not fit 256 VGPRs next to the row-sum accumulators).
  // MODE 0: next tile with C = -m ; 2: next tile opens the NEXT query block (C = 0, qf holds its queries) ; 3: none.
  auto half_step = [&](f32x16& cur, f32x16& nxt, auto mode_tag, auto&& loads) __attribute__((always_inline)) {
    constexpr int MODE = decltype(mode_tag)::value;
    // The instruction order is pinned (sched_barrier between the pieces): an in-order wave issues ~7 VALU under one 32-cycle
    // MFMA, and the four S^T MFMAs are one dependent chain, so they alternate with the independent PV / row-sum MFMAs and the
    // exps sit where an MFMA is in flight.  hipcc's own order put the first S^T MFMA (and a wait for all 8 reads) first.
#define SB_ do { if constexpr (LN3D_KRES_PIN && !(LN3D_KRES_ABL & 32)) __builtin_amdgcn_sched_barrier(0); } while (0)
#define EXP_(x_) ((LN3D_KRES_ABL & 1) ? (x_) : __builtin_amdgcn_exp2f(x_))
#define MFMA_(a_, b_, c_) ((LN3D_KRES_ABL & 2) ? abl_nomfma(a_, b_, c_) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b_, c_, 0, 0, 0))
    f32x16 zc;
    if constexpr (MODE == 2) { _Pragma("unroll") for (int r = 0; r < 16; ++r) zc[r] = 0.f; }
    if constexpr (!(LN3D_KRES_ABL & 8)) loads();         // K rows of the next tile (4 reads), then V^T columns of this one (4)
    SB_;
    bf16x8 pb[2];
    auto probs = [&](auto s_tag) __attribute__((always_inline)) {     // 8 exps + 4 packs: P of 16 keys as the PV B operand
      constexpr int S = decltype(s_tag)::value;
      union { uint32_t u[4]; bf16x8 v; } cv;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) cv.u[jj] = pack2bf(EXP_(cur[8 * S + 2 * jj]), EXP_(cur[8 * S + 2 * jj + 1]));
      pb[S] = cv.v;
    };
    auto pv = [&](auto s_tag) __attribute__((always_inline)) {        // 2 PV MFMAs + 1 row-sum MFMA of 16 keys
      constexpr int S = decltype(s_tag)::value;
      oacc[0] = MFMA_(vf[2 * S], pb[S], oacc[0]);
      oacc[1] = MFMA_(vf[2 * S + 1], pb[S], oacc[1]);
      lacc = MFMA_(ones, pb[S], lacc);
    };
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
    if constexpr (LN3D_KRES_ORDER == 0) {
      // S^T chain spread between the independent MFMAs
      float e[8];
      probs(S0{});
      SB_;
      if constexpr (MODE != 3) nxt = MFMA_(kf[0], qf[0], MODE == 2 ? zc : negm);
      SB_;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) e[jj] = EXP_(cur[8 + jj]);
      SB_;
      if constexpr (MODE != 3) nxt = MFMA_(kf[1], qf[1], nxt);
      SB_;
      {
#pragma unroll
        for (int jj = 4; jj < 8; ++jj) e[jj] = EXP_(cur[8 + jj]);
        union { uint32_t u[4]; bf16x8 v; } cv;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) cv.u[jj] = pack2bf(e[2 * jj], e[2 * jj + 1]);
        pb[1] = cv.v;
      }
      SB_;
      oacc[0] = MFMA_(vf[0], pb[0], oacc[0]);
      SB_;
      if constexpr (MODE != 3) nxt = MFMA_(kf[2], qf[2], nxt);
      SB_;
      oacc[1] = MFMA_(vf[1], pb[0], oacc[1]);
      SB_;
      lacc = MFMA_(ones, pb[0], lacc);
      SB_;
      if constexpr (MODE != 3) nxt = MFMA_(kf[3], qf[3], nxt);
      SB_;
      pv(S1{});
      SB_;
    } else if constexpr (LN3D_KRES_ORDER == 1) {
      // S^T chain back to back (accumulator forwarding), every exp block under MFMAs already issued
      probs(S0{});
      SB_;
      if constexpr (MODE != 3) {
        nxt = MFMA_(kf[0], qf[0], MODE == 2 ? zc : negm);
        nxt = MFMA_(kf[1], qf[1], nxt);
        nxt = MFMA_(kf[2], qf[2], nxt);
        nxt = MFMA_(kf[3], qf[3], nxt);
      }
      SB_;
      probs(S1{});
      SB_;
      pv(S0{});
      pv(S1{});
      SB_;
    } else {
      // PV of the first 16 keys first (needs only V^T), the S^T chain back to back behind it
      probs(S0{});
      SB_;
      pv(S0{});
      SB_;
      probs(S1{});
      SB_;
      if constexpr (MODE != 3) {
        nxt = MFMA_(kf[0], qf[0], MODE == 2 ? zc : negm);
        nxt = MFMA_(kf[1], qf[1], nxt);
        nxt = MFMA_(kf[2], qf[2], nxt);
        nxt = MFMA_(kf[3], qf[3], nxt);
      }
      SB_;
      pv(S1{});
      SB_;
    }
#undef SB_
#undef EXP_
#undef MFMA_
  };

  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  // ---- prologue.  Issue order = the order of the stream below (step i issues K block i+5 and V^T stage i+3):
  //      Q x4, K0, K1, K2, V0, K3, V1, K4, V2.  The first S^T tile needs the queries and K0: 7 younger requests.
  dma_q(qb0);
  issue_k(); issue_k(); issue_k(); issue_v(I0{}); issue_k(); issue_v(I1{}); issue_k(); issue_v(I2{});
  asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_q();
  {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    load_kf(std::integral_constant<int, 0>{});
    qk_tile(stA, z);
    fresh_reference(stA);
  }

  // ---- the stream.  Step g (slot SL = g & 3) multiplies V^T stage g and reads K blocks g (tile 1) and g+1 (tile 0).  At its top, stage g (issued at step
  // g-3) and K block g+1 (issued at step g-4, first query block only) must have landed for every wave; behind the barrier every
  // wave has left step g-1, whose ring slot takes stage g+3.  W = requests issued behind stage g:
  //   later query blocks: V(g+1), V(g+2)                                   -> 2
  //   first query block : K(g+3), V(g+1), K(g+4), V(g+2) while they exist  -> 4 ; 3 / 2 in its last group (K blocks run out)
  //   + 4 while the next block's query DMA (issued at step 0) is younger   : steps 1..3 of a block's first group
  //   + 4 while the previous block's O stores (end of its last step) are younger : steps 0..2 of a later block's first group
  auto wait_w = [&](int w) __attribute__((always_inline)) {
    switch (w) {
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };
  auto top_of_step = [&](auto slot_tag, bool first, bool lastgrp, bool p1, bool more, bool stq, int qb) __attribute__((always_inline)) {
    constexpr int SL = decltype(slot_tag)::value;
    int w = p1 ? (lastgrp ? (SL == 0 ? 3 : 2) : 4) : 2;
    if (first && more && SL >= 1) w += 4;
    if (ST_INORDER && first && stq && SL <= 2) w += 4;
    if constexpr (!(LN3D_KRES_ABL & 4)) { wait_w(w); __builtin_amdgcn_s_barrier(); }
    if constexpr (!(LN3D_KRES_ABL & 16)) {
      if (p1 && kb_k < nkb) issue_k();
      issue_v(std::integral_constant<int, SL + 3>{});
    }
    if (SL == 0 && first && more) dma_q(qb + 1);
  };
  auto advance_k = [&](bool wrap) __attribute__((always_inline)) {      // inside step 3 of a group: K block g+1 opens the next group
#pragma unroll
    for (int j = 0; j < 4; ++j) fk[j] = wrap ? fo[j] : fk[j] + 32768;
  };
  // half-step A: tile 0 of block g -> P / PV, S^T of its tile 1 ; half-step B: tile 1 -> P / PV, S^T of tile 0 of block g+1
#define HS_A(SL_, MODE_) half_step(stA, stB, MODE_{}, [&]() __attribute__((always_inline)) {                                             \
    load_kf(std::integral_constant<int, SL_ * 8192 + 4096>{}); load_vf(std::integral_constant<int, SL_>{}, I0{}); })
#define HS_B(SL_, MODE_) half_step(stB, stA, MODE_{}, [&]() __attribute__((always_inline)) {                                             \
    if constexpr (MODE_::value != 3) load_kf(std::integral_constant<int, (SL_ == 3 ? 0 : (SL_ + 1) * 8192)>{});                          \
    load_vf(std::integral_constant<int, SL_>{}, I1{}); })
#define STEP(SL_, LASTG_) { top_of_step(std::integral_constant<int, SL_>{}, first, LASTG_, p1, more, stq, qb); HS_A(SL_, I0); HS_B(SL_, I0); }
  for (int qb = qb0; qb < qb1; ++qb) {
    const bool more = qb + 1 < qb1, p1 = qb == qb0, stq = qb != qb0;
    bool first = true;
    for (int grp = 1; grp < ngrp; ++grp) {
      STEP(0, false); STEP(1, false); STEP(2, false);
      top_of_step(I3{}, first, false, p1, more, stq, qb);
      HS_A(3, I0);
      advance_k(false);
      HS_B(3, I0);
      first = false;
    }
    STEP(0, true); STEP(1, true); STEP(2, true);
    top_of_step(I3{}, first, true, p1, more, stq, qb);
    HS_A(3, I0);
    advance_k(true);
    if (more) {
      read_q();
      HS_B(3, I2);
      finish_block(qb);
      fresh_reference(stA);
    } else {
      HS_B(3, I3);
      finish_block(qb);
    }
  }
#undef STEP
#undef HS_A
#undef HS_B
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the ring's read-ahead must not outlive the workgroup's LDS
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();

  // ---- exact recomputation of query blocks whose reference overflowed (rare; never on the DiT's own activations)
  if (lane == 0) *reinterpret_cast<volatile uint32_t*>(wstage) = ovf;
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();
  uint32_t redo = 0;
#pragma unroll
  for (int w = 0; w < 8; ++w) redo |= *reinterpret_cast<volatile uint32_t*>(smem + WST + w * 4096);
  redo = __builtin_amdgcn_readfirstlane(redo);
  if constexpr (LN3D_KRES_ABL != 0) redo = 0;              // ablated builds produce garbage sums: time the fast path only
  while (redo) {
    const int blk = __builtin_ctz(redo);
    redo &= redo - 1;
    const int qb = qb0 + blk;
    {
      int qr = qb * QB + wid * 32 + l31; qr = qr < p.Nq_pad ? qr : p.Nq_pad - 1;
#pragma unroll
      for (int ds = 0; ds < NDS; ++ds) {
        union { uint32_t u[4]; uint4 q; bf16x8 v; } cv;
        cv.q = *reinterpret_cast<const uint4*>(Qg + (int64_t)qr * DH + (2 * ds + hi) * 8);
        scale_q(cv.u);
        qf[ds] = cv.v;
      }
    }
    float m_run = -3.0e38f, l_run = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.f; oacc[1][r] = 0.f; }
    for (int kb = 0; kb < nkb; ++kb) {
      __builtin_amdgcn_s_barrier();                       // ring slot 0 is free (previous block of keys consumed by every wave)
      lds_dma16_v((Vg + kb * 128 + voffs), lds_addr((smem + VRING + wid * 1024)));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      f32x16 st[2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
      for (int ds = 0; ds < NDS; ++ds)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          const bf16x8 kf = *(lds_frag_t*)(uintptr_t)(fo[ds] + kb * 8192 + kt * 4096);
          st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ds], st[kt], 0, 0, 0);
        }
      float mx = -3.0e38f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kt][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) { oacc[0][r] *= alpha; oacc[1][r] *= alpha; }
      m_run = m_new;
      float psum = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float pv = __builtin_amdgcn_exp2f(st[kt][r] - m_run); st[kt][r] = pv; psum += pv; }
      l_run += psum;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        union { uint32_t u[4]; bf16x8 v; } cv;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          cv.u[jj] = pack2bf(st[s >> 1][8 * (s & 1) + 2 * jj], st[s >> 1][8 * (s & 1) + 2 * jj + 1]);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          const bf16x8 vfr = *(lds_frag_t*)(uintptr_t)(fv[s] + dt * 4096);
          oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr, cv.v, oacc[dt], 0, 0, 0);
        }
      }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    store_o(qb, 1.0f / l_tot);
  }
}

// ---------------------------------------------------------------------------------------------
// Short key sequences (Nk <= 128: the 77-token text context of every cross-attention, the CLIP text tower), Dh = 64.
// The general kernel above is latency-bound there (ring prologue, a barrier per block, 768 workgroups on 512 slots).  Here a
// workgroup is 4 waves = 256 queries of one (batch, head), every wave takes two 32-query tiles in turn; K and V^T of the whole
// head (2 x 16 KB, same tile images and swizzle as the ring stages) are DMA'd once, one barrier, then softmax is a single
// pass over the <= 128 scores a lane holds (no online rescale).  48 KB LDS and <= 168 VGPRs: three workgroups per CU, so
// B*H*ceil(Nq/256) = 768 workgroups of DiT-L/2 are exactly one round on 256 CUs.
template <int QT>   // query tiles (of 32) per wave: the workgroup covers 128 * QT queries
__global__ __launch_bounds__(256, 3) void attn_short_kernel(AttnP p) {
  constexpr int DH = 64, KROWB = 128, KTILE = KVB * KROWB, VTILE = DH * 128, STAGEB = KTILE + VTILE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int bh = blockIdx.y;
  const bf16_t* Qg = p.Q + (int64_t)bh * p.Nq_pad * DH;
  const bf16_t* Kg = p.K + (int64_t)bh * p.Nk_pad * DH;
  const bf16_t* Vg = p.Vt + (int64_t)bh * DH * p.Nk_pad;
  const int nkb = (p.Nk + KVB - 1) / KVB;            // 1 or 2

  // K / V^T of the head: tile image j (1 KB each, 8 per tile) exactly as in attn_kernel; wave w issues j = w and w + 4
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    if (kb < nkb) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int j = wid + 4 * i;
        const int krow = j * 8 + lane / 8, kcp = lane % 8;
        const bf16_t* ks = Kg + (int64_t)(kb * KVB + krow) * DH + ((kcp ^ ((krow >> 1) & 7)) * 8);
        const int vrow = j * 8 + (lane >> 3), vcp = lane & 7;
        const bf16_t* vs = Vg + (int64_t)vrow * p.Nk_pad + kb * KVB + ((vcp ^ ((vrow >> 1) & 7)) * 8);
        lds_dma16_v(ks, lds_addr((smem + kb * STAGEB + j * 1024)));
        lds_dma16_v(vs, lds_addr((smem + kb * STAGEB + KTILE + j * 1024)));
      }
    }
  }
  // Q tiles of this wave: 32 rows x 128 B are contiguous in memory, so they are read with fully coalesced 16-byte loads (a
  // fragment-shaped read would touch 32 rows x 32 B per instruction) and re-shaped through the wave's private 4 KB of LDS
  // (K-style swizzle); the same 4 KB later transposes O so that it leaves as complete 128-byte rows.
  char* wreg = smem + 2 * STAGEB + wid * 4096;
  uint4 qraw[QT][4];
#pragma unroll
  for (int it = 0; it < QT; ++it) {
    const int q0 = blockIdx.x * (128 * QT) + (QT * wid + it) * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int row = q0 + 8 * i + (lane >> 3); row = row < p.Nq_pad ? row : p.Nq_pad - 1;
      qraw[it][i] = *reinterpret_cast<const uint4*>(Qg + (int64_t)row * DH + (lane & 7) * 8);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  const int kkey_r = (l31 >> 1) & 7, k_off = l31 * KROWB;
  const int vkey_r = (l31 >> 1) & 7, v_off = KTILE + l31 * 128;
  const int b = bh / p.H, h = bh - b * p.H;
  if constexpr (LN3D_ATTN_ABL & 8) {                   // bench-only: loads + stores, no compute
#pragma unroll
    for (int it = 0; it < QT; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = blockIdx.x * (128 * QT) + (QT * wid + it) * 32 + 8 * i + (lane >> 3);
        if (q < p.Nq) *reinterpret_cast<uint4*>(p.O + ((int64_t)b * p.Nq + q) * p.ldo + h * DH + (lane & 7) * 8) = qraw[it][i];
      }
    return;
  }
#pragma unroll 1
  for (int it = 0; it < QT; ++it) {
    const int q0 = blockIdx.x * (128 * QT) + (QT * wid + it) * 32;
    if (q0 >= p.Nq) break;                            // wave-uniform
    bf16x8 qc[4];
    {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 8 * i + (lane >> 3);
        const uint4 v = (QT > 1 && it) ? qraw[QT - 1][i] : qraw[0][i];
        *reinterpret_cast<uint4*>(wreg + r * 128 + (((lane & 7) ^ ((r >> 1) & 7)) << 4)) = v;
      }
#pragma unroll
      for (int ds = 0; ds < 4; ++ds)
        qc[ds] = *reinterpret_cast<const bf16x8*>(wreg + l31 * 128 + (((2 * ds + hi) ^ kkey_r) << 4));
    }
    f32x16 st[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[kb][kt][r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (kb < nkb) {
        const char* sb = smem + kb * STAGEB;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds)
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sb + k_off + kt * 32 * KROWB + (((2 * ds + hi) ^ kkey_r) << 4));
            st[kb][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qc[ds], st[kb][kt], 0, 0, 0);
          }
      }
    }
    // mask (tail keys, keys of an absent second block, causal) and the row maximum
    float mx = -3.0e38f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb * KVB + 4 * hi + kt * 32 + (r & 3) + 8 * (r >> 2);
          const bool ok = key < p.Nk && (!p.causal || key <= q0 + l31);
          st[kb][kt][r] = ok ? st[kb][kt][r] : -3.0e38f;
          mx = fmaxf(mx, st[kb][kt][r]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m = mx * p.scale_log2;
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(fmaf(st[kb][kt][r], p.scale_log2, -m));
          st[kb][kt][r] = pv;
          psum += pv;
        }
    const float inv = 1.0f / (psum + __shfl_xor(psum, 32, 64));
    f32x16 oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (kb < nkb) {
        const char* sb = smem + kb * STAGEB;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          union { uint32_t u[4]; bf16x8 v; } cv;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
            cv.u[jj] = pack2bf(st[kb][s >> 1][8 * (s & 1) + 2 * jj], st[kb][s >> 1][8 * (s & 1) + 2 * jj + 1]);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sb + v_off + dt * 32 * 128 + (((2 * s + hi) ^ vkey_r) << 4));
            oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, cv.v, oacc[dt], 0, 0, 0);
          }
        }
      }
    }
    // O: bf16 through the wave's 4 KB (8-byte chunk c of row r at c ^ (r & 15)), out as 16 bytes per lane = whole 128-byte rows
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 o;
        o.x = pack2bf(oacc[dt][4 * g + 0] * inv, oacc[dt][4 * g + 1] * inv);
        o.y = pack2bf(oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv);
        const int c8 = dt * 8 + 2 * g + hi;
        *reinterpret_cast<uint2*>(wreg + l31 * 128 + ((c8 ^ (l31 & 15)) << 3)) = o;
      }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 8 * i + (lane >> 3), c16 = lane & 7;
      // chunks 2*c16 and 2*c16+1 of row r sit in one aligned 16-byte slot, swapped when r is odd
      uint4 v = *reinterpret_cast<const uint4*>(wreg + r * 128 + ((c16 ^ ((r & 15) >> 1)) << 4));
      if (r & 1) { const uint32_t t0 = v.x, t1 = v.y; v.x = v.z; v.y = v.w; v.z = t0; v.w = t1; }
      const int q = q0 + r;
      if (q < p.Nq) *reinterpret_cast<uint4*>(p.O + ((int64_t)b * p.Nq + q) * p.ldo + h * DH + c16 * 8) = v;
    }
  }
}

// r5: the attention kernels have no environment switches left - one kernel per shape class (the r1 - r4 A/B switches LN3D_ATTN_V /
// _KRES / _SHORT / _QT / _SPLIT and the instantiations they selected are gone; their measurements are in profiles/r2 - r4_attn*.md).
extern "C" void ln3d_attn_reload_env(void) {}

static int launch_attn_short(const AttnP& p, hipStream_t s) {
  // one query tile per wave (128 queries per workgroup): twice the workgroups, so load, compute and store phases of
  // different workgroups overlap on a CU (the kernel is a latency-bound stream of Q in / O out)
  hipLaunchKernelGGL(attn_short_kernel<1>, dim3((p.Nq + 127) / 128, p.B * p.H), dim3(256), 2 * (KVB * 128 + 64 * 128) + 4 * 4096, s, p);
  return ln3d_check_launch();
}

template <int DH, int OCC, int DT = DH>
static int launch_attn(const AttnP& p, hipStream_t s) {
  constexpr int NST = DH == 64 ? 4 : (DH == 80 ? 3 : LN3D_ATTN_NST128);
  constexpr int LDS = DH == 80 ? NST * (KVB * 176 + 96 * 128) : NST * (KVB * DH * 2 + DH * 128);
  static AttrOnce attr_once;
  if (attr_once.need()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<DH, OCC, DT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  }
  dim3 grid(((p.Nq + 255) / 256) * p.B * p.H);
  hipLaunchKernelGGL((attn_kernel<DH, OCC, DT>), grid, dim3(512), LDS, s, p);
  return ln3d_check_launch();
}

static int launch_attn_stream(AttnP p, hipStream_t s) {
  constexpr int LDS = 8 * 16384 + 8 * 4096;     // all 160 KiB of the CU: one workgroup per CU
  static AttrOnce attr_once;
  if (attr_once.need()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_stream_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  }
  // one workgroup per (batch, head) walks all its query blocks; with fewer heads than CUs the query blocks of a head are
  // shared out so that every CU has work
  const int nqb = (p.Nq + 255) / 256, BH = p.B * p.H, cus = ln3d_stream_cus(s);
  int nsplit = 1;
  if (BH < cus) { nsplit = (cus + BH - 1) / BH; if (nsplit > nqb) nsplit = nqb; }
  p.nsplit = nsplit;
  hipLaunchKernelGGL(attn_stream_kernel, dim3(BH * nsplit), dim3(512), LDS, s, p);
  return ln3d_check_launch();
}

template <bool PRIO, bool ST_INORDER>
static int launch_attn_kres_t(const AttnP& p, hipStream_t s) {
  constexpr int LDS = 768 * 128 + 4 * 8192 + 8 * 4096;   // all 160 KiB of the CU
  static AttrOnce attr_once;
  if (attr_once.need()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kres_kernel<PRIO, ST_INORDER>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  }
  hipLaunchKernelGGL((attn_kres_kernel<PRIO, ST_INORDER>), dim3(p.B * p.H * p.nsplit), dim3(512), LDS, s, p);
  return ln3d_check_launch();
}
static int launch_attn_kres(AttnP p, hipStream_t s) {
  p.nsplit = 1;                                            // every workgroup walks all query blocks of its head: K is fetched once
  return launch_attn_kres_t<true, false>(p, s);            // static priority for waves 4-7, O stores not counted in the vmcnt waits (r3 A/B)
}

extern "C" int ln3d_attention_bf16(const ln3d_attn_args* a, void* stream) {
  if (!a || !a->Q || !a->K || !a->Vt || !a->O) return LN3D_ERR_BAD_ARG;
  if (a->Nk <= 0 || a->Nq <= 0 || a->Nk_pad % 64 != 0 || a->Nk_pad < a->Nk || a->Nq_pad < a->Nq) return LN3D_ERR_BAD_ARG;
  AttnP p;
  p.Q = (const bf16_t*)a->Q; p.K = (const bf16_t*)a->K; p.Vt = (const bf16_t*)a->Vt; p.O = (bf16_t*)a->O;
  p.B = a->B; p.H = a->H; p.Nq = a->Nq; p.Nq_pad = a->Nq_pad; p.Nk = a->Nk; p.Nk_pad = a->Nk_pad;
  p.ldo = a->ldo;
  p.scale_log2 = a->scale * 1.4426950408889634f;
  p.causal = a->causal ? 1 : 0;
  p.nsplit = 1;
  hipStream_t s = (hipStream_t)stream;
  // causal masking exists in the short-sequence kernel only (its one user is the 77-token CLIP text tower)
  if (a->causal && !(a->Dh == 64 && a->Nk <= 128)) return LN3D_ERR_UNSUPPORTED;
  if (a->Dh_true != 0 && a->Dh_true != a->Dh && a->Dh != 128 && a->Dh != 80) return LN3D_ERR_UNSUPPORTED;   // compact heads: the padded kernels only
  if (a->Dh == 64) {
    // The short-sequence kernel serves the causal text tower; for the non-causal 77-key cross-attention it measures 2 us
    // faster in isolation but slower inside the sampling loop than the ring kernel (and the DiT runs that attention inside
    // the query-projection GEMM anyway, LN3D_EPI_CROSS_ATTN), so the ring kernel serves it.
    if (a->Nk <= 128 && a->causal) return launch_attn_short(p, s);
    if ((a->Nk & 255) == 0 && a->Nk_pad == a->Nk) {
      // K-resident kernel: K of a head fits beside the V^T ring (Nk <= 768) and there is a head for every CU, so that one
      // workgroup per head walks all its query blocks (fewer heads: the query blocks of a head are shared out - attn_stream)
      if (a->Nk >= 512 && a->Nk <= 768 && (a->Nq & 255) == 0 && a->Nq_pad == a->Nq && p.B * p.H >= ln3d_stream_cus(s))
        return launch_attn_kres(p, s);
      return launch_attn_stream(p, s);
    }
    return a->Nk > 128 ? launch_attn<64, 4>(p, s) : launch_attn<64, 2>(p, s);
  }
  if (a->Dh == 80) {                                                        // r6: 65 - 80 wide heads stored 80 wide
    if (a->Dh_true == 0 || a->Dh_true == 80) return launch_attn<80, 2>(p, s);   // the U-Net's 80-wide heads
    if (a->Dh_true == 72) return launch_attn<80, 2, 72>(p, s);                // DiT-XL/2
    return LN3D_ERR_UNSUPPORTED;
  }
  if (a->Dh == 128) {
    if (a->Dh_true == 0 || a->Dh_true == 128) return launch_attn<128, 2>(p, s);
    if (a->Dh_true == 72) return launch_attn<128, 2, 72>(p, s);           // DiT-XL/2
    return LN3D_ERR_UNSUPPORTED;
  }
  return LN3D_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------
// per-head RMSNorm on q / k (qk_norm): rows of Dh bf16, in place; one 16-lane group per row (Dh=64)
__global__ void rmsnorm_heads_kernel(bf16_t* x, const float* w, int64_t rows, int Dh, int true_dim, float eps) {
  const int per = Dh <= 64 ? 16 : 32;              // lanes per row (a power of two for the butterfly), 4 elements each; lanes beyond Dh / 4 idle
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = gid / per;
  const int c = (int)(gid % per);
  const bool act = row < rows && c * 4 < Dh;
  bf16_t* px = x + (act ? row * Dh + c * 4 : 0);
  uint2 raw = make_uint2(0u, 0u);
  if (act) raw = *reinterpret_cast<const uint2*>(px);
  float v0 = bf2f(raw.x & 0xffff), v1 = bf2f(raw.x >> 16), v2 = bf2f(raw.y & 0xffff), v3 = bf2f(raw.y >> 16);
  float ss = v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
  for (int o = per >> 1; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  if (!act) return;
  const float rs = rsqrtf(ss / true_dim + eps);
  const float4 ww = *reinterpret_cast<const float4*>(w + c * 4);
  uint2 o;
  o.x = pack2bf(v0 * rs * ww.x, v1 * rs * ww.y);
  o.y = pack2bf(v2 * rs * ww.z, v3 * rs * ww.w);
  *reinterpret_cast<uint2*>(px) = o;
}

extern "C" int ln3d_rmsnorm_heads_bf16(void* x, const float* w, int64_t rows, int Dh, int true_dim, float eps, void* stream) {
  if (!x || !w || (Dh != 64 && Dh != 80 && Dh != 128) || true_dim < 0 || true_dim > Dh) return LN3D_ERR_BAD_ARG;
  if (true_dim == 0) true_dim = Dh;
  const int per = Dh <= 64 ? 16 : 32;
  const int64_t threads = rows * per;
  hipLaunchKernelGGL(rmsnorm_heads_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (bf16_t*)x, w, rows, Dh, true_dim, eps);
  return ln3d_check_launch();
}
