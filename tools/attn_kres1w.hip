// EXPERIMENT (round 5), not part of libln3d_hip.so: the one-wave-per-SIMD form of the K-resident self-attention kernel with the O^T /
// row-sum accumulators pinned in AGPRs - the kernel the r3 / r4 reviews asked for.  Included by tools/attn_bench.hip behind
// ../ln3diff_amd/csrc/attention.hip (it uses that file's AttnP, fragment layout and DMA helpers).  Result (profiles/r5_attn.md): bit-for-
// bit the shipped kernel's accuracy on every bench case including the overflow-recompute path, 50.2 - 52.5 us against the shipped
// kernel's 48.9 - 49.6 us at 256 heads x 768^2 x 64 - no gain, so it does not ship.  Kept because the file is the evidence: the AGPR
// pinning works (no v_accvgpr traffic in the loop, 0 scratch), the ablation switches below produced the numbers in the profile note.
#pragma once
// ---------------------------------------------------------------------------------------------
// r5: ONE wave per SIMD form of the K-resident kernel, with the O^T and row-sum accumulators resident in AGPRs (the form the r3 / r4
// reviews asked for).  LDS image, DMA ring, counted waits and softmax scheme are attn_kres_kernel's; what changes is the wave:
//  * 4 waves x 64 queries: every wave carries TWO 32-query sets that share each K / V^T fragment read (half the LDS reads per MFMA).
//  * The six accumulators of the two sets (2 x (2 O^T tiles + 1 row-sum tile) = 96 registers) live in AGPRs: their MFMAs are inline
//    asm with "a" constraints (the builtins cannot name AGPR operands, and hipcc's own AGPR split of the r3 attempt put VALU-visible
//    values there: 35 us of v_accvgpr traffic).  They are read once per query block, behind an explicit s_nop (the hazard recogniser
//    does not see into the asm).  Everything the VALU touches (S^T, P, fragments, queries) stays in architectural VGPRs.
//  * The sets run half a tile apart.  Phase A issues the 10 MFMAs of set 0 (S^T chain of its NEXT tile, then PV + row sums of this
//    tile) with set 1's softmax placed behind them one unit (2 v_exp + 1 v_cvt_pk) per MFMA; phase B is the mirror image.  The order
//    is pinned with sched_barrier: one wave per SIMD has nobody else to fill its issue slots.
//  * Fragment reads are placed where their registers die: V^T of a tile at the top of its phase A (first use 4 MFMAs later), K rows
//    of tile t + 2 behind the last S^T MFMA of phase B (first use 6 MFMAs later).
// Every logical DMA request is two instructions per wave here (4 waves fill what 8 did), so the counted waits are twice kres's.
#ifndef LN3D_K1W_ABL   // tools/attn_bench.hip builds with -DLN3D_K1W_ABL=bits (wrong results by construction): 1 no v_exp, 2 no PV / row-sum
#define LN3D_K1W_ABL 0  // MFMAs, 4 no barrier / DMA wait, 8 no fragment reads in the stream, 16 no DMA issue in the stream, 32 no S^T chain, 64 no softmax units, 128 no pins
#endif
#ifndef LN3D_K1W_OPT   // bring-up A/B bits: 1 row-sum MFMAs ahead of the PV MFMAs, 2 asm wait dispatch, 4 DMA pieces issued under the S^T chain
#define LN3D_K1W_OPT 0
#endif
// The 96 accumulator registers of attn_kres1w_kernel are FIXED AGPRs named in the asm text (set 0: a[0:15] a[16:31] O^T, a[32:47] row
// sums; set 1: a[48:95]) and listed as clobbers of every such statement, so hipcc keeps nothing of its own in them across any of
// these statements.  (As C++ values behind "+a" constraints hipcc re-homed them between statements: 16 v_accvgpr_mov per MFMA.)
#define K1W_ACC_CLOBBERS_                                                                                                            \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19",   \
  "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37",       \
  "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55",       \
  "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73",       \
  "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91",       \
  "a92", "a93", "a94", "a95"
// (register numbers are spelled in the text: an "n" operand above 64 prints in hex)
#define K1W_PV_(ACC_, A_, B_) do { if constexpr (LN3D_K1W_ABL & 2) asm volatile("" ::"v"(A_), "v"(B_)); else                            \
    asm volatile("v_mfma_f32_32x32x16_bf16 " ACC_ ", %0, %1, " ACC_ ::"v"(A_), "v"(B_) : K1W_ACC_CLOBBERS_); } while (0)
template <int LO>
__device__ __forceinline__ void k1w_pv(const bf16x8& a, const bf16x8& b) {       // a[LO:LO+15] += A . B
  static_assert(LO % 16 == 0 && LO < 96, "accumulator tile");
  if constexpr (LO == 0) K1W_PV_("a[0:15]", a, b);
  else if constexpr (LO == 16) K1W_PV_("a[16:31]", a, b);
  else if constexpr (LO == 32) K1W_PV_("a[32:47]", a, b);
  else if constexpr (LO == 48) K1W_PV_("a[48:63]", a, b);
  else if constexpr (LO == 64) K1W_PV_("a[64:79]", a, b);
  else K1W_PV_("a[80:95]", a, b);
}
#undef K1W_PV_
template <int LO>
__device__ __forceinline__ void k1w_acc_read16(f32x16& v) {
  static_assert(LO == 0 || LO == 16 || LO == 48 || LO == 64, "O^T tile");
  float t[16];
  if constexpr (LO == 0)
    asm volatile(
               "v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\tv_accvgpr_read_b32 %2, a2\n\tv_accvgpr_read_b32 %3, a3\n\t"
               "v_accvgpr_read_b32 %4, a4\n\tv_accvgpr_read_b32 %5, a5\n\tv_accvgpr_read_b32 %6, a6\n\tv_accvgpr_read_b32 %7, a7\n\t"
               "v_accvgpr_read_b32 %8, a8\n\tv_accvgpr_read_b32 %9, a9\n\tv_accvgpr_read_b32 %10, a10\n\tv_accvgpr_read_b32 %11, a11\n\t"
               "v_accvgpr_read_b32 %12, a12\n\tv_accvgpr_read_b32 %13, a13\n\tv_accvgpr_read_b32 %14, a14\n\tv_accvgpr_read_b32 %15, a15\n\t"
               : "=v"(t[0]), "=v"(t[1]), "=v"(t[2]), "=v"(t[3]), "=v"(t[4]), "=v"(t[5]), "=v"(t[6]), "=v"(t[7]), "=v"(t[8]), "=v"(t[9]), "=v"(t[10]), "=v"(t[11]), "=v"(t[12]), "=v"(t[13]), "=v"(t[14]), "=v"(t[15]) :: K1W_ACC_CLOBBERS_);
  else if constexpr (LO == 16)
    asm volatile(
               "v_accvgpr_read_b32 %0, a16\n\tv_accvgpr_read_b32 %1, a17\n\tv_accvgpr_read_b32 %2, a18\n\tv_accvgpr_read_b32 %3, a19\n\t"
               "v_accvgpr_read_b32 %4, a20\n\tv_accvgpr_read_b32 %5, a21\n\tv_accvgpr_read_b32 %6, a22\n\tv_accvgpr_read_b32 %7, a23\n\t"
               "v_accvgpr_read_b32 %8, a24\n\tv_accvgpr_read_b32 %9, a25\n\tv_accvgpr_read_b32 %10, a26\n\tv_accvgpr_read_b32 %11, a27\n\t"
               "v_accvgpr_read_b32 %12, a28\n\tv_accvgpr_read_b32 %13, a29\n\tv_accvgpr_read_b32 %14, a30\n\tv_accvgpr_read_b32 %15, a31\n\t"
               : "=v"(t[0]), "=v"(t[1]), "=v"(t[2]), "=v"(t[3]), "=v"(t[4]), "=v"(t[5]), "=v"(t[6]), "=v"(t[7]), "=v"(t[8]), "=v"(t[9]), "=v"(t[10]), "=v"(t[11]), "=v"(t[12]), "=v"(t[13]), "=v"(t[14]), "=v"(t[15]) :: K1W_ACC_CLOBBERS_);
  else if constexpr (LO == 48)
    asm volatile(
               "v_accvgpr_read_b32 %0, a48\n\tv_accvgpr_read_b32 %1, a49\n\tv_accvgpr_read_b32 %2, a50\n\tv_accvgpr_read_b32 %3, a51\n\t"
               "v_accvgpr_read_b32 %4, a52\n\tv_accvgpr_read_b32 %5, a53\n\tv_accvgpr_read_b32 %6, a54\n\tv_accvgpr_read_b32 %7, a55\n\t"
               "v_accvgpr_read_b32 %8, a56\n\tv_accvgpr_read_b32 %9, a57\n\tv_accvgpr_read_b32 %10, a58\n\tv_accvgpr_read_b32 %11, a59\n\t"
               "v_accvgpr_read_b32 %12, a60\n\tv_accvgpr_read_b32 %13, a61\n\tv_accvgpr_read_b32 %14, a62\n\tv_accvgpr_read_b32 %15, a63\n\t"
               : "=v"(t[0]), "=v"(t[1]), "=v"(t[2]), "=v"(t[3]), "=v"(t[4]), "=v"(t[5]), "=v"(t[6]), "=v"(t[7]), "=v"(t[8]), "=v"(t[9]), "=v"(t[10]), "=v"(t[11]), "=v"(t[12]), "=v"(t[13]), "=v"(t[14]), "=v"(t[15]) :: K1W_ACC_CLOBBERS_);
  else
    asm volatile(
               "v_accvgpr_read_b32 %0, a64\n\tv_accvgpr_read_b32 %1, a65\n\tv_accvgpr_read_b32 %2, a66\n\tv_accvgpr_read_b32 %3, a67\n\t"
               "v_accvgpr_read_b32 %4, a68\n\tv_accvgpr_read_b32 %5, a69\n\tv_accvgpr_read_b32 %6, a70\n\tv_accvgpr_read_b32 %7, a71\n\t"
               "v_accvgpr_read_b32 %8, a72\n\tv_accvgpr_read_b32 %9, a73\n\tv_accvgpr_read_b32 %10, a74\n\tv_accvgpr_read_b32 %11, a75\n\t"
               "v_accvgpr_read_b32 %12, a76\n\tv_accvgpr_read_b32 %13, a77\n\tv_accvgpr_read_b32 %14, a78\n\tv_accvgpr_read_b32 %15, a79\n\t"
               : "=v"(t[0]), "=v"(t[1]), "=v"(t[2]), "=v"(t[3]), "=v"(t[4]), "=v"(t[5]), "=v"(t[6]), "=v"(t[7]), "=v"(t[8]), "=v"(t[9]), "=v"(t[10]), "=v"(t[11]), "=v"(t[12]), "=v"(t[13]), "=v"(t[14]), "=v"(t[15]) :: K1W_ACC_CLOBBERS_);
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = t[r];
}
template <int R>
__device__ __forceinline__ float k1w_acc_read1() {       // row sums: every register of the tile holds the lane's column sum
  static_assert(R == 32 || R == 80, "row-sum tile");
  float x;
  if constexpr (R == 32) asm volatile("v_accvgpr_read_b32 %0, a32" : "=v"(x) :: K1W_ACC_CLOBBERS_);
  else asm volatile("v_accvgpr_read_b32 %0, a80" : "=v"(x) :: K1W_ACC_CLOBBERS_);
  return x;
}
template <int LO>
__device__ __forceinline__ void k1w_acc_zero48() {
  static_assert(LO == 0 || LO == 48, "set");
  if constexpr (LO == 0)
    asm volatile(
               "v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0\n\t"
               "v_accvgpr_write_b32 a4, 0\n\tv_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\t"
               "v_accvgpr_write_b32 a8, 0\n\tv_accvgpr_write_b32 a9, 0\n\tv_accvgpr_write_b32 a10, 0\n\tv_accvgpr_write_b32 a11, 0\n\t"
               "v_accvgpr_write_b32 a12, 0\n\tv_accvgpr_write_b32 a13, 0\n\tv_accvgpr_write_b32 a14, 0\n\tv_accvgpr_write_b32 a15, 0\n\t"
               "v_accvgpr_write_b32 a16, 0\n\tv_accvgpr_write_b32 a17, 0\n\tv_accvgpr_write_b32 a18, 0\n\tv_accvgpr_write_b32 a19, 0\n\t"
               "v_accvgpr_write_b32 a20, 0\n\tv_accvgpr_write_b32 a21, 0\n\tv_accvgpr_write_b32 a22, 0\n\tv_accvgpr_write_b32 a23, 0\n\t"
               "v_accvgpr_write_b32 a24, 0\n\tv_accvgpr_write_b32 a25, 0\n\tv_accvgpr_write_b32 a26, 0\n\tv_accvgpr_write_b32 a27, 0\n\t"
               "v_accvgpr_write_b32 a28, 0\n\tv_accvgpr_write_b32 a29, 0\n\tv_accvgpr_write_b32 a30, 0\n\tv_accvgpr_write_b32 a31, 0\n\t"
               "v_accvgpr_write_b32 a32, 0\n\tv_accvgpr_write_b32 a33, 0\n\tv_accvgpr_write_b32 a34, 0\n\tv_accvgpr_write_b32 a35, 0\n\t"
               "v_accvgpr_write_b32 a36, 0\n\tv_accvgpr_write_b32 a37, 0\n\tv_accvgpr_write_b32 a38, 0\n\tv_accvgpr_write_b32 a39, 0\n\t"
               "v_accvgpr_write_b32 a40, 0\n\tv_accvgpr_write_b32 a41, 0\n\tv_accvgpr_write_b32 a42, 0\n\tv_accvgpr_write_b32 a43, 0\n\t"
               "v_accvgpr_write_b32 a44, 0\n\tv_accvgpr_write_b32 a45, 0\n\tv_accvgpr_write_b32 a46, 0\n\tv_accvgpr_write_b32 a47, 0\n\t"
               ::: K1W_ACC_CLOBBERS_);
  else
    asm volatile(
               "v_accvgpr_write_b32 a48, 0\n\tv_accvgpr_write_b32 a49, 0\n\tv_accvgpr_write_b32 a50, 0\n\tv_accvgpr_write_b32 a51, 0\n\t"
               "v_accvgpr_write_b32 a52, 0\n\tv_accvgpr_write_b32 a53, 0\n\tv_accvgpr_write_b32 a54, 0\n\tv_accvgpr_write_b32 a55, 0\n\t"
               "v_accvgpr_write_b32 a56, 0\n\tv_accvgpr_write_b32 a57, 0\n\tv_accvgpr_write_b32 a58, 0\n\tv_accvgpr_write_b32 a59, 0\n\t"
               "v_accvgpr_write_b32 a60, 0\n\tv_accvgpr_write_b32 a61, 0\n\tv_accvgpr_write_b32 a62, 0\n\tv_accvgpr_write_b32 a63, 0\n\t"
               "v_accvgpr_write_b32 a64, 0\n\tv_accvgpr_write_b32 a65, 0\n\tv_accvgpr_write_b32 a66, 0\n\tv_accvgpr_write_b32 a67, 0\n\t"
               "v_accvgpr_write_b32 a68, 0\n\tv_accvgpr_write_b32 a69, 0\n\tv_accvgpr_write_b32 a70, 0\n\tv_accvgpr_write_b32 a71, 0\n\t"
               "v_accvgpr_write_b32 a72, 0\n\tv_accvgpr_write_b32 a73, 0\n\tv_accvgpr_write_b32 a74, 0\n\tv_accvgpr_write_b32 a75, 0\n\t"
               "v_accvgpr_write_b32 a76, 0\n\tv_accvgpr_write_b32 a77, 0\n\tv_accvgpr_write_b32 a78, 0\n\tv_accvgpr_write_b32 a79, 0\n\t"
               "v_accvgpr_write_b32 a80, 0\n\tv_accvgpr_write_b32 a81, 0\n\tv_accvgpr_write_b32 a82, 0\n\tv_accvgpr_write_b32 a83, 0\n\t"
               "v_accvgpr_write_b32 a84, 0\n\tv_accvgpr_write_b32 a85, 0\n\tv_accvgpr_write_b32 a86, 0\n\tv_accvgpr_write_b32 a87, 0\n\t"
               "v_accvgpr_write_b32 a88, 0\n\tv_accvgpr_write_b32 a89, 0\n\tv_accvgpr_write_b32 a90, 0\n\tv_accvgpr_write_b32 a91, 0\n\t"
               "v_accvgpr_write_b32 a92, 0\n\tv_accvgpr_write_b32 a93, 0\n\tv_accvgpr_write_b32 a94, 0\n\tv_accvgpr_write_b32 a95, 0\n\t"
               ::: K1W_ACC_CLOBBERS_);
}
__global__ __launch_bounds__(256, 1) void attn_kres1w_kernel(AttnP p) {
  constexpr int DH = 64, QB = 256;
  constexpr int KREG = 768 * 128;                         // resident K rows (128 B each)
  constexpr int VSLOT = 8192, VRING = KREG;               // V^T ring: 4 slots x (64 dims x 64 keys)
  constexpr int WST = KREG + 4 * VSLOT;                   // 4 x 8 KB: per-wave staging (2 sets x 32 queries in by LDS-DMA, O out)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int nqb = p.Nq / QB;                              // launcher: Nq % 256 == 0, one workgroup per (batch, head)
  const int bh = blockIdx.x;
  const int nkb = p.Nk >> 6;                              // 8 or 12 key blocks of 64
  const int T = 2 * nkb;                                  // 32-key tiles per query block

  const bf16_t* Qg = p.Q + (int64_t)bh * p.Nq_pad * DH;
  const char* Kg = reinterpret_cast<const char*>(p.K + (int64_t)bh * p.Nk_pad * DH);
  const char* Vg = reinterpret_cast<const char*>(p.Vt + (int64_t)bh * DH * p.Nk_pad);
  const int b_smp = bh / p.H, h_idx = bh - b_smp * p.H;

  typedef __attribute__((address_space(3))) const bf16x8 lds_frag_t;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;
  uint32_t fo[4], fv[4];                                  // chunk 2j + hi of row l31 of a tile image (K block 0 / ring slot 0)
#pragma unroll
  for (int j = 0; j < 4; ++j) { fo[j] = lds0 + l31 * 128 + (((2 * j + hi) ^ ((l31 >> 1) & 7)) << 4); fv[j] = fo[j] + VRING; }
  // DMA pieces: wave w fills rows 16w .. 16w+15 (two 1 KB pieces) of a K block / of a V^T stage
  uint32_t koffs[2], voffs[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int drow = 16 * wid + 8 * j + (lane >> 3), dchunk = ((lane & 7) ^ ((drow >> 1) & 7)) << 4;
    koffs[j] = drow * 128 + dchunk;
    voffs[j] = (uint32_t)drow * (uint32_t)p.Nk_pad * 2u + dchunk;
  }
  const char* kq = Kg; const char* vq = Vg;
  int kb_k = 0, kb_v = 0;
  auto issue_k = [&]() __attribute__((always_inline)) {
    lds_dma16_s(kq, koffs[0], lds0 + kb_k * 8192 + wid * 2048);
    lds_dma16_s(kq, koffs[1], lds0 + kb_k * 8192 + wid * 2048 + 1024);
    ++kb_k; kq += 64 * DH * 2;
  };
  auto issue_v = [&](int sl) __attribute__((always_inline)) {
    lds_dma16_s(vq, voffs[0], lds0 + VRING + sl * VSLOT + wid * 2048);
    lds_dma16_s(vq, voffs[1], lds0 + VRING + sl * VSLOT + wid * 2048 + 1024);
    ++kb_v; vq += 64 * 2;
    if (kb_v == nkb) { kb_v = 0; vq = Vg; }
  };
  char* const wstage = smem + WST + wid * 8192;
  auto dma_q = [&](int qb) __attribute__((always_inline)) {
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = j * 8 + (lane >> 3);
        int qr = qb * QB + wid * 64 + st * 32 + row; qr = qr < p.Nq_pad ? qr : p.Nq_pad - 1;
        lds_dma16_v((Qg + (int64_t)qr * DH + (((lane & 7) ^ ((row >> 1) & 7)) * 8)), lds_addr((wstage + st * 4096 + j * 1024)));
      }
  };
  auto scale_q = [&](uint32_t (&u)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
      u[jj] = pack2bf(bf2f((bf16_t)(u[jj] & 0xffffu)) * p.scale_log2, bf2f((bf16_t)(u[jj] >> 16)) * p.scale_log2);
  };
  auto read_q = [&](bf16x8 (&qf)[4], int st) __attribute__((always_inline)) {
    const uint32_t qrow = lds0 + WST + wid * 8192 + st * 4096 + l31 * 128;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
      union { uint32_t u[4]; bf16x8 v; } cv;
      cv.v = *(lds_frag_t*)(uintptr_t)(qrow + (((2 * ds + hi) ^ ((l31 >> 1) & 7)) << 4));
      scale_q(cv.u);
      qf[ds] = cv.v;
    }
  };

  union PB { uint32_t u[4]; bf16x8 v; };
  bf16x8 qf0[4], qf1[4], kf[4], vf[4];
  PB pb0[2], pb1[2];
  f32x16 s0, s1, negm0, negm1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { negm0[r] = 0.f; negm1[r] = 0.f; }
  k1w_acc_zero48<0>(); k1w_acc_zero48<48>();             // a[0:95]: O^T / row-sum accumulators of both sets
  bf16x8 ones;
  { union { uint32_t u[4]; bf16x8 v; } cv; cv.u[0] = cv.u[1] = cv.u[2] = cv.u[3] = 0x3F803F80u; ones = cv.v; }
  asm volatile("" : "+v"(ones));                          // opaque: otherwise re-materialised from SGPRs in front of every row-sum MFMA
  uint32_t ovf = 0;

  auto load_kf = [&](uint32_t off) __attribute__((always_inline)) {               // K rows of the 32-key tile at byte offset `off`
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) kf[ds] = *(lds_frag_t*)(uintptr_t)(fo[ds] + off);
  };
  auto load_vf = [&](auto kt_tag, uint32_t slot_off) __attribute__((always_inline)) {   // V^T of 32 keys of a ring slot
    constexpr int KT = decltype(kt_tag)::value;
#pragma unroll
    for (int sg = 0; sg < 2; ++sg)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        vf[sg * 2 + dt] = *(lds_frag_t*)(uintptr_t)(fv[2 * KT + sg] + slot_off + dt * 4096);
  };
  auto tile_max = [&](const f32x16& st) __attribute__((always_inline)) {
    float mx = max3f(st[0], st[1], st[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = max3f(mx, st[r], st[r + 1]);
    return fmaxf(mx, st[15]);
  };
  auto fresh_reference = [&](f32x16& st, f32x16& negm) __attribute__((always_inline)) {
    float mx = tile_max(st);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = -mx;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] -= mx;
  };
  auto store_o = [&](int qb, int st, const f32x16& oa, const f32x16& ob, float inv) __attribute__((always_inline)) {
    const int q0 = qb * QB + wid * 64 + st * 32;
    const uint32_t sbase = lds0 + WST + wid * 8192 + st * 4096;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const f32x16& oc = dt ? ob : oa;
        typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
        const u32x2_t ov = {pack2bf(oc[4 * gq + 0] * inv, oc[4 * gq + 1] * inv), pack2bf(oc[4 * gq + 2] * inv, oc[4 * gq + 3] * inv)};
        const int c8 = dt * 8 + 2 * gq + hi;
        const uint32_t oaddr = sbase + l31 * 128 + ((c8 ^ (l31 & 15)) << 3);
        asm volatile("ds_write_b64 %0, %1" :: "v"(oaddr), "v"(ov) : "memory");      // see attn_kres_kernel::store_o
      }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 8 * i + (lane >> 3), c16 = lane & 7;
      typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
      typedef __attribute__((address_space(3))) const u32x4_t lds_u4_t;
      uint32_t wbase = sbase;
      asm volatile("" : "+s"(wbase));
      const u32x4_t vr = *(lds_u4_t*)(uintptr_t)(wbase + r * 128 + ((c16 ^ ((r & 15) >> 1)) << 4));
      uint4 v = make_uint4(vr.x, vr.y, vr.z, vr.w);
      if (r & 1) { const uint32_t t0 = v.x, t1 = v.y; v.x = v.z; v.y = v.w; v.z = t0; v.w = t1; }
      *reinterpret_cast<uint4*>(p.O + ((int64_t)b_smp * p.Nq + q0 + r) * p.ldo + h_idx * DH + c16 * 8) = v;
    }
  };
  auto finish_set = [&](int qb, auto base_tag) __attribute__((always_inline)) {
    constexpr int BASE = decltype(base_tag)::value;
    // the asm MFMAs are invisible to the hazard recogniser: 8-pass XDL write -> v_accvgpr_read needs <= 18 wait states
    asm volatile("s_nop 15\n\ts_nop 15" ::: K1W_ACC_CLOBBERS_);
    f32x16 oa, ob;
    const float l = k1w_acc_read1<BASE + 32>();
    k1w_acc_read16<BASE>(oa);
    k1w_acc_read16<BASE + 16>(ob);
    if (__builtin_amdgcn_ballot_w64(!(l < 1.0e30f)) != 0) ovf |= 1u << qb;
    store_o(qb, BASE / 48, oa, ob, 1.0f / l);
    k1w_acc_zero48<BASE>();
  };

#define SB1_ do { if constexpr (!(LN3D_K1W_ABL & 128)) __builtin_amdgcn_sched_barrier(0); } while (0)
#define EXP1_(x_) ((LN3D_K1W_ABL & 1) ? (x_) : __builtin_amdgcn_exp2f(x_))
#define UNIT1_(PB_, S_, I_) do { if constexpr (!(LN3D_K1W_ABL & 64)) PB_[(I_) >> 2].u[(I_) & 3] =                                      \
      pack2bf(EXP1_(S_[8 * ((I_) >> 2) + 2 * ((I_) & 3)]), EXP1_(S_[8 * ((I_) >> 2) + 2 * ((I_) & 3) + 1])); } while (0)
  // The S^T chain is asm too, in VGPR form: with AGPRs in the budget hipcc gives the builtin's result an AGPR, and every v_exp then
  // reads its input through v_accvgpr_read (the r3 one-wave kernel's 35 us).  Dependent 8-pass MFMAs on one accumulator issue back to
  // back (hipcc emits the same for the builtin form); their first VALU reader is >= 6 MFMAs away, or behind SNOP_ in the prologue.
#define SCHC_(S_, A_, B_, C_) do { if constexpr (LN3D_K1W_ABL & 32) asm("" : "=v"(S_) : "v"(A_), "v"(B_), "0"(C_)); else                \
      asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(S_) : "v"(A_), "v"(B_), "v"(C_)); } while (0)
#define SCH0_(S_, A_, B_) do { if constexpr (LN3D_K1W_ABL & 32) asm("" : "+v"(S_) : "v"(A_), "v"(B_)); else                               \
      asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(S_) : "v"(A_), "v"(B_)); } while (0)
#define SCH_(S_, A_, B_) do { if constexpr (LN3D_K1W_ABL & 32) asm("" : "+v"(S_) : "v"(A_), "v"(B_)); else                                \
      asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(S_) : "v"(A_), "v"(B_)); } while (0)
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  using YES = std::true_type; using NO = std::false_type;
  using A0 = std::integral_constant<int, 0>; using A48 = std::integral_constant<int, 48>;
  auto nothing = []() __attribute__((always_inline)) {};
  auto none1 = [](auto) __attribute__((always_inline)) {};
  // One phase: the 10 MFMAs of set M (S^T chain of its next tile into sM, then PV / row sums of this tile from pbM) with the softmax
  // of set V (sV -> pbV) one unit behind each MFMA.  MODE 0: chain with C = -m ; 2: chain opens the next query block (C = 0) ; 3: no
  // chain.  UNITS false: set V has no further tile (last phase of the workgroup).
  auto phase = [&](auto mode_tag, auto units_tag, f32x16& sM, const f32x16& negmM, const bf16x8 (&qfM)[4], auto base_tag,
                   const PB (&pbM)[2], const f32x16& sV, PB (&pbV)[2], auto&& top, auto&& mid, auto&& hook, auto&& ldc, auto&& ldp) __attribute__((always_inline)) {
    constexpr int MODE = decltype(mode_tag)::value;
    constexpr int BASE = decltype(base_tag)::value;       // accumulators of set M: a[BASE:BASE+47]
    constexpr bool UNITS = decltype(units_tag)::value;
    if constexpr (!(LN3D_K1W_ABL & (8 | 256))) top();
    SB1_;
    if constexpr (MODE != 3) {
      if constexpr (MODE == 2) SCH0_(sM, kf[0], qfM[0]); else SCHC_(sM, kf[0], qfM[0], negmM);
      SB1_;
      ldc(I0{}); SB1_;
      if constexpr (UNITS) UNIT1_(pbV, sV, 0); SB1_;
      SCH_(sM, kf[1], qfM[1]); SB1_;
      ldc(I1{}); SB1_;
      hook();                                            // (LN3D_K1W_OPT & 4) this step's DMA pieces, issued under the chain
      SB1_;
      if constexpr (UNITS) UNIT1_(pbV, sV, 1); SB1_;
      SCH_(sM, kf[2], qfM[2]); SB1_;
      ldc(I2{}); SB1_;
      if constexpr (UNITS) UNIT1_(pbV, sV, 2); SB1_;
      SCH_(sM, kf[3], qfM[3]); SB1_;
      ldc(I3{}); SB1_;
      if constexpr (UNITS) UNIT1_(pbV, sV, 3); SB1_;
    } else if constexpr (UNITS) {
      UNIT1_(pbV, sV, 0); UNIT1_(pbV, sV, 1); UNIT1_(pbV, sV, 2); UNIT1_(pbV, sV, 3); SB1_;
    }
    if constexpr (!(LN3D_K1W_ABL & (8 | 512))) mid();
    SB1_;
    if constexpr ((LN3D_K1W_OPT & (1 | 8)) != 0) {
      // row sums first: they need P only, which gives the V^T reads of phase A two more MFMAs of cover
      k1w_pv<BASE + 32>(ones, pbM[0].v); SB1_;
      if constexpr (UNITS) UNIT1_(pbV, sV, 4); SB1_;
      k1w_pv<BASE + 32>(ones, pbM[1].v); SB1_;
      if constexpr (UNITS) UNIT1_(pbV, sV, 5); SB1_;
      k1w_pv<BASE>(vf[0], pbM[0].v); SB1_;
      ldp(I0{}); SB1_;
      if constexpr (UNITS) UNIT1_(pbV, sV, 6); SB1_;
      k1w_pv<BASE + 16>(vf[1], pbM[0].v); SB1_;
      ldp(I1{}); SB1_;
      if constexpr (UNITS) UNIT1_(pbV, sV, 7); SB1_;
      k1w_pv<BASE>(vf[2], pbM[1].v); SB1_;
      ldp(I2{}); SB1_;
      k1w_pv<BASE + 16>(vf[3], pbM[1].v); SB1_;
      ldp(I3{}); SB1_;
    } else {
      k1w_pv<BASE>(vf[0], pbM[0].v); SB1_;
      if constexpr (UNITS) UNIT1_(pbV, sV, 4); SB1_;
      k1w_pv<BASE + 16>(vf[1], pbM[0].v); SB1_;
      if constexpr (UNITS) UNIT1_(pbV, sV, 5); SB1_;
      k1w_pv<BASE + 32>(ones, pbM[0].v); SB1_;
      if constexpr (UNITS) UNIT1_(pbV, sV, 6); SB1_;
      k1w_pv<BASE>(vf[2], pbM[1].v); SB1_;
      if constexpr (UNITS) UNIT1_(pbV, sV, 7); SB1_;
      k1w_pv<BASE + 16>(vf[3], pbM[1].v); SB1_;
      k1w_pv<BASE + 32>(ones, pbM[1].v); SB1_;
    }
  };

  // ---- prologue.  Issue order as in attn_kres_kernel: Q (8 pieces), K0, K1, K2, V0, K3, V1, K4, V2 (2 pieces each).  The first
  //      S^T tiles need the queries and K0: 14 younger pieces.
  dma_q(0);
  issue_k(); issue_k(); issue_k(); issue_v(0); issue_k(); issue_v(1); issue_k(); issue_v(2);
  asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_q(qf0, 0);
  read_q(qf1, 1);
  {
    load_kf(0);
    SCH0_(s0, kf[0], qf0[0]); SCH_(s0, kf[1], qf0[1]); SCH_(s0, kf[2], qf0[2]); SCH_(s0, kf[3], qf0[3]);
    SCH0_(s1, kf[0], qf1[0]); SCH_(s1, kf[1], qf1[1]); SCH_(s1, kf[2], qf1[2]); SCH_(s1, kf[3], qf1[3]);
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(s0), "+v"(s1));     // MFMA write -> VALU read, invisible to the hazard recogniser
    load_kf(4096);
    fresh_reference(s0, negm0);
    fresh_reference(s1, negm1);
    UNIT1_(pb0, s0, 0); UNIT1_(pb0, s0, 1); UNIT1_(pb0, s0, 2); UNIT1_(pb0, s0, 3);
    UNIT1_(pb0, s0, 4); UNIT1_(pb0, s0, 5); UNIT1_(pb0, s0, 6); UNIT1_(pb0, s0, 7);
  }

  // ---- the stream.  Step g multiplies V^T stage g (ring slot g & 3) and the S^T chains read K tiles 2g+1, 2g+2; the fragment
  //      reads issued inside step g touch K blocks <= g+1 and stage g only, so kres's invariant holds: at the top of step g, stage g
  //      and (first query block) K block g+1 have landed for every wave, and behind the barrier slot (g+3) & 3 is free.
  //      W (pieces younger than stage g) = 2 x kres's count.
  auto wait_w = [&](int w) __attribute__((always_inline)) {
    if constexpr (LN3D_K1W_OPT & 2) {
      // one compare + two taken branches on the frequent value; hipcc turns an if-chain or a switch into a structured decision tree
      // of 6 - 10 branches per step, which one wave per SIMD has nobody to hide behind
      asm volatile(
          "s_cmp_eq_u32 %0, 4\n\ts_cbranch_scc1 .Lk1w_w4_%=\n\t"
          "s_cmp_eq_u32 %0, 8\n\ts_cbranch_scc1 .Lk1w_w8_%=\n\t"
          "s_cmp_eq_u32 %0, 16\n\ts_cbranch_scc1 .Lk1w_w16_%=\n\t"
          "s_cmp_eq_u32 %0, 12\n\ts_cbranch_scc1 .Lk1w_w12_%=\n\t"
          "s_cmp_eq_u32 %0, 6\n\ts_cbranch_scc1 .Lk1w_w6_%=\n\t"
          "s_waitcnt vmcnt(0)\n\ts_branch .Lk1w_we_%=\n"
          ".Lk1w_w4_%=:\n\ts_waitcnt vmcnt(4)\n\ts_branch .Lk1w_we_%=\n"
          ".Lk1w_w8_%=:\n\ts_waitcnt vmcnt(8)\n\ts_branch .Lk1w_we_%=\n"
          ".Lk1w_w16_%=:\n\ts_waitcnt vmcnt(16)\n\ts_branch .Lk1w_we_%=\n"
          ".Lk1w_w12_%=:\n\ts_waitcnt vmcnt(12)\n\ts_branch .Lk1w_we_%=\n"
          ".Lk1w_w6_%=:\n\ts_waitcnt vmcnt(6)\n"
          ".Lk1w_we_%=:"
          :: "s"(w) : "memory", "scc");
    } else {
      if (w == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (w == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (w == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (w == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (w == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
  for (int qb = 0; qb < nqb; ++qb) {
    const bool more = qb + 1 < nqb, p1 = qb == 0;
    auto step_dma = [&](int g) __attribute__((always_inline)) {
      if constexpr (!(LN3D_K1W_ABL & 16)) {
        if (p1 && kb_k < nkb) issue_k();
        issue_v((g + 3) & 3);
      }
      if (g == 0 && more) dma_q(qb + 1);
    };
    auto top_of_step = [&](int g) __attribute__((always_inline)) {
      int w = 2 + (p1 ? (g + 3 < nkb) + (g + 4 < nkb) : 0);      // V(g+1), V(g+2) [+ K(g+3), K(g+4) while they exist]
      if (g >= 1 && g <= 3 && more) w += 4;                       // the next block's query DMA (issued at step 0) is younger
      if constexpr (!(LN3D_K1W_ABL & 4)) { wait_w(2 * w); __builtin_amdgcn_s_barrier(); }
      if constexpr (!(LN3D_K1W_OPT & 4)) step_dma(g);
    };
    // tile 2g + KT: phase A (set 0's MFMAs, set 1's softmax) then phase B (the mirror image); `ko` = byte offset of the K tile read
    // during this tile (tile 2g + KT + 2 mod T: the chains of the NEXT tile's phases).  Fragment reads, LN3D_K1W_OPT & 8: one read
    // behind the MFMA that frees its register - V^T of an even tile under its own S^T chain (the stage is only known to have landed
    // behind the step's barrier), V^T of the odd tile under the even tile's phase-B PV MFMAs, K rows under phase B's chain; otherwise
    // in two blocks of four (top of phase A, middle of phase B).
    auto tile = [&](int g, auto kt_tag, auto mode_tag, auto units_b_tag, uint32_t ko, auto&& between) __attribute__((always_inline)) {
      constexpr int KT = decltype(kt_tag)::value;
      constexpr int MODE = decltype(mode_tag)::value;
      const uint32_t vo = (uint32_t)(g & 3) * VSLOT;
      auto dma = [&]() __attribute__((always_inline)) { if constexpr (KT == 0 && (LN3D_K1W_OPT & 4) != 0) step_dma(g); };
      auto ldk = [&](auto i_tag) __attribute__((always_inline)) {
        constexpr int i = decltype(i_tag)::value;
        if constexpr (MODE != 3 && !(LN3D_K1W_ABL & 8)) kf[i] = *(lds_frag_t*)(uintptr_t)(fo[i] + ko);
      };
      auto ldv_here = [&](auto i_tag) __attribute__((always_inline)) {       // fragment i of THIS tile's V^T
        constexpr int i = decltype(i_tag)::value;
        if constexpr (!(LN3D_K1W_ABL & 8)) vf[i] = *(lds_frag_t*)(uintptr_t)(fv[2 * KT + (i >> 1)] + vo + (i & 1) * 4096);
      };
      auto ldv_odd = [&](auto i_tag) __attribute__((always_inline)) {        // fragment i of the step's second tile
        constexpr int i = decltype(i_tag)::value;
        if constexpr (!(LN3D_K1W_ABL & 8)) vf[i] = *(lds_frag_t*)(uintptr_t)(fv[2 + (i >> 1)] + vo + (i & 1) * 4096);
      };
      if constexpr ((LN3D_K1W_OPT & 8) != 0) {
        if constexpr (KT == 0) {
          phase(mode_tag, YES{}, s0, negm0, qf0, A0{}, pb0, s1, pb1, nothing, nothing, dma, ldv_here, none1);
          between();
          phase(mode_tag, units_b_tag, s1, negm1, qf1, A48{}, pb1, s0, pb0, nothing, nothing, nothing, ldk, ldv_odd);
        } else {
          phase(mode_tag, YES{}, s0, negm0, qf0, A0{}, pb0, s1, pb1, nothing, nothing, dma, none1, none1);
          between();
          phase(mode_tag, units_b_tag, s1, negm1, qf1, A48{}, pb1, s0, pb0, nothing, nothing, nothing, ldk, none1);
        }
      } else {
        phase(mode_tag, YES{}, s0, negm0, qf0, A0{}, pb0, s1, pb1, [&]() __attribute__((always_inline)) { load_vf(kt_tag, vo); }, nothing, dma,
              none1, none1);
        between();
        phase(mode_tag, units_b_tag, s1, negm1, qf1, A48{}, pb1, s0, pb0, nothing,
              [&]() __attribute__((always_inline)) { if constexpr (MODE != 3) load_kf(ko); }, nothing, none1, none1);
      }
    };
    auto kofs = [&](int u) __attribute__((always_inline)) { return (uint32_t)(u >= T ? u - T : u) * 4096u; };
    // the inner loop has ONE path (the block's last step is peeled): qf / negm / S stay in their registers across iterations
    for (int g = 0; g + 1 < nkb; ++g) {
      top_of_step(g);
      tile(g, I0{}, I0{}, YES{}, kofs(2 * g + 2), nothing);
      tile(g, I1{}, I0{}, YES{}, kofs(2 * g + 3), nothing);
    }
    {
      const int g = nkb - 1;
      top_of_step(g);
      tile(g, I0{}, I0{}, YES{}, 0u, nothing);            // reads K tile 0: the chains of the next block's first tile
      if (more) {
        // last tile of the block: both chains open the next query block (its queries landed at step 4 at the latest)
        read_q(qf0, 0);
        read_q(qf1, 1);
        tile(g, I1{}, I2{}, YES{}, 4096u, [&]() __attribute__((always_inline)) { fresh_reference(s0, negm0); });
        finish_set(qb, A0{});
        finish_set(qb, A48{});
        fresh_reference(s1, negm1);
      } else {
        tile(g, I1{}, I3{}, NO{}, 0u, nothing);
        finish_set(qb, A0{});
        finish_set(qb, A48{});
      }
    }
  }
#undef SB1_
#undef UNIT1_
#undef EXP1_
#undef SCHC_
#undef SCH0_
#undef SCH_
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the ring's read-ahead must not outlive the workgroup's LDS
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();

  // ---- exact recomputation of query blocks whose reference overflowed (rare; never on the DiT's own activations)
  if (lane == 0) *reinterpret_cast<volatile uint32_t*>(wstage) = ovf;
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();
  uint32_t redo = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) redo |= *reinterpret_cast<volatile uint32_t*>(smem + WST + w * 8192);
  redo = __builtin_amdgcn_readfirstlane(redo);
  if constexpr (LN3D_K1W_ABL != 0) redo = 0;              // ablated builds produce garbage sums: time the fast path only
  while (redo) {
    const int qb = __builtin_ctz(redo);
    redo &= redo - 1;
    for (int st = 0; st < 2; ++st) {
      bf16x8 qx[4];
      {
        int qr = qb * QB + wid * 64 + st * 32 + l31; qr = qr < p.Nq_pad ? qr : p.Nq_pad - 1;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
          union { uint32_t u[4]; uint4 q; bf16x8 v; } cv;
          cv.q = *reinterpret_cast<const uint4*>(Qg + (int64_t)qr * DH + (2 * ds + hi) * 8);
          scale_q(cv.u);
          qx[ds] = cv.v;
        }
      }
      float m_run = -3.0e38f, l_run = 0.f;
      f32x16 ox[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) { ox[0][r] = 0.f; ox[1][r] = 0.f; }
      for (int kb = 0; kb < nkb; ++kb) {
        __builtin_amdgcn_s_barrier();                     // ring slot 0 is free (previous block of keys consumed by every wave)
        lds_dma16_v((Vg + kb * 128 + voffs[0]), lds_addr((smem + VRING + wid * 2048)));
        lds_dma16_v((Vg + kb * 128 + voffs[1]), lds_addr((smem + VRING + wid * 2048 + 1024)));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        f32x16 sx[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) sx[kt][r] = 0.f;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds)
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) {
            const bf16x8 kx = *(lds_frag_t*)(uintptr_t)(fo[ds] + kb * 8192 + kt * 4096);
            sx[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kx, qx[ds], sx[kt], 0, 0, 0);
          }
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sx[kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        l_run *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) { ox[0][r] *= alpha; ox[1][r] *= alpha; }
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) { const float pv = __builtin_amdgcn_exp2f(sx[kt][r] - m_run); sx[kt][r] = pv; psum += pv; }
        l_run += psum;
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) {
          union { uint32_t u[4]; bf16x8 v; } cv;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
            cv.u[jj] = pack2bf(sx[sg >> 1][8 * (sg & 1) + 2 * jj], sx[sg >> 1][8 * (sg & 1) + 2 * jj + 1]);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const bf16x8 vx = *(lds_frag_t*)(uintptr_t)(fv[sg] + dt * 4096);
            ox[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vx, cv.v, ox[dt], 0, 0, 0);
          }
        }
      }
      const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
      store_o(qb, st, ox[0], ox[1], 1.0f / l_tot);
    }
  }
}

static int launch_attn_kres1w(const AttnP& p, hipStream_t s) {
  constexpr int LDS = 768 * 128 + 4 * 8192 + 4 * 8192;    // all 160 KiB of the CU
  static AttrOnce attr_once;
  if (attr_once.need())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kres1w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  hipLaunchKernelGGL(attn_kres1w_kernel, dim3(p.B * p.H), dim3(256), LDS, s, p);
  return ln3d_check_launch();
}
